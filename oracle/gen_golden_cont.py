"""Golden fixture for the Rayleigh and aerosol continuum: the REFERENCE's calc_tau_rayleighj / rayleighv / rayleighv2 /
rayleighls (ForwardModel_0.py:5525, :5598, :5647, :5712) and ForwardModel_0.calc_tau_dust (:4790) on seeded inputs, both
spectral units; the aerosol table is built so that the cubic interpolant leaves the physical range in places (the
reference then falls back to the linear one, :4849-4859).   Build container only.   python oracle/gen_golden_cont.py"""
import os
import sys
import importlib
from types import SimpleNamespace
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import_reference()
    FM = importlib.import_module("archnemesis.ForwardModel_0")
    rng = np.random.default_rng(515)
    out = {}
    NLAY = 9
    TOTAM = 10.0 ** rng.uniform(24, 29, NLAY)
    wn = np.linspace(300.0, 25000.0, 157)                    # cm-1
    wl = np.sort(1.0e4 / wn)                                 # micron
    ID = np.array([39, 40, 6, 11, 2, 39]); ISO = np.array([0, 0, 1, 0, 0, 2])      # the second H2 entry (ISO 2) is ignored
    VMR = np.column_stack([rng.uniform(0.80, 0.88, NLAY), rng.uniform(0.10, 0.15, NLAY), rng.uniform(1e-3, 3e-3, NLAY),
                           10.0 ** rng.uniform(-7, -4, NLAY), rng.uniform(0, 1e-5, NLAY), rng.uniform(0, 1e-4, NLAY)])
    VMR[2, 0] = 0.0                                          # a layer without H2: the ratios stay 0 there (:5772-5774)
    out.update(TOTAM=TOTAM, wn=wn, wl=wl, ID=ID, ISO=ISO, VMR=VMR)
    with np.errstate(all="ignore"):
        for isp, w in ((0, wn), (1, wl)):
            for name, fn in (("j", FM.calc_tau_rayleighj), ("v", FM.calc_tau_rayleighv), ("v2", FM.calc_tau_rayleighv2)):
                t, d = fn(isp, w, TOTAM)
                out[f"ray_{name}_{isp}_tau"] = t; out[f"ray_{name}_{isp}_dtau"] = d
            t, d = FM.calc_tau_rayleighls(isp, w, ID, ISO, VMR, TOTAM)
            out[f"ray_ls_{isp}_tau"] = t; out[f"ray_ls_{isp}_dtau"] = d

    # aerosols: 3 populations on 12 tabulated wavelengths; population 1 has a spike (spline undershoots below 0),
    # population 2 has ksca close to kext (the spline crosses: kext < ksca)
    NW, ND = 12, 3
    SW = np.sort(rng.uniform(0.3, 5.0, NW)); SW[0] = 0.3; SW[-1] = 5.0
    KEXT = 10.0 ** rng.uniform(-9, -8, (NW, ND))
    KSCA = KEXT * rng.uniform(0.3, 0.9, (NW, ND))
    KEXT[5, 1] *= 300.0; KSCA[5, 1] *= 280.0
    KSCA[:, 2] = KEXT[:, 2] * (1.0 - 10.0 ** rng.uniform(-4, -1.5, NW))
    CONT = 10.0 ** rng.uniform(3, 9, (NLAY, ND))
    WAVEC = np.linspace(0.3, 5.0, 211)
    self = SimpleNamespace(Scatter=SimpleNamespace(NDUST=ND), AtmosphereX=SimpleNamespace(DUST_RENORMALISATION={}))
    Scat = SimpleNamespace(NDUST=ND, NWAVE=NW, WAVE=SW, KEXT=KEXT, KSCA=KSCA)
    Lay = SimpleNamespace(NLAY=NLAY, CONT=CONT.copy())
    r = FM.ForwardModel_0.calc_tau_dust(self, WAVEC, Scat, Lay)
    out.update(SW=SW, KEXT=KEXT, KSCA=KSCA, CONT=CONT, WAVEC_D=WAVEC)
    for n, a in zip(("TAUDUST", "TAUCLSCAT", "dTAUDUSTdq", "dTAUCLSCATdq"), r):
        out["dust_" + n] = a
    # two tabulated wavelengths: linear
    Scat2 = SimpleNamespace(NDUST=ND, NWAVE=2, WAVE=SW[[0, -1]], KEXT=KEXT[[0, -1]], KSCA=KSCA[[0, -1]])
    r2 = FM.ForwardModel_0.calc_tau_dust(self, WAVEC, Scat2, SimpleNamespace(NLAY=NLAY, CONT=CONT.copy()))
    for n, a in zip(("TAUDUST", "TAUCLSCAT", "dTAUDUSTdq", "dTAUCLSCATdq"), r2):
        out["dust2_" + n] = a
    np.savez_compressed(os.path.join(OUT, "continuum_ray_dust.npz"), **out)
    from scipy import interpolate
    ke = interpolate.interp1d(SW, KEXT[:, 1], kind="cubic")(WAVEC); ks = interpolate.interp1d(SW, KSCA[:, 1], kind="cubic")(WAVEC)
    ke2 = interpolate.interp1d(SW, KEXT[:, 2], kind="cubic")(WAVEC); ks2 = interpolate.interp1d(SW, KSCA[:, 2], kind="cubic")(WAVEC)
    print("fallback points:", int(((ks < 0) & (ke > 0)).sum()), int(((ke < 0) & (ks > 0)).sum()), int((ke < ks).sum()), int((ke2 < ks2).sum()))
    print({k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
