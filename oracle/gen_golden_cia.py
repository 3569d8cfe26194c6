"""Golden fixture for the collision-induced-absorption opacity: the REFERENCE's ForwardModel_0.calc_tau_cia (:4516-4760)
called with explicit arguments on a synthetic CIA table (default 9 pairs incl. the two H2-H2 / H2-He tables that depend on
the ortho/para ratio), a 5-gas atmosphere with H2, He, CH4, N2, CO2 (so the co2cia / n2n2cia / n2h2cia terms are on)
and a layer set that exercises both table clamps.  Cases: wavenumber space, wavelength space, a NORMAL-ratio CIA object.
Also stores co2cia / n2n2cia / n2h2cia(WAVEN) -- wavenumber-only vectors the GPU entry point takes as inputs.
Build container only.   python oracle/gen_golden_cia.py"""
import os
import sys
import importlib
import types
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import_reference()
    fm = importlib.import_module("archnemesis.ForwardModel_0")
    cm = importlib.import_module("archnemesis.CIA_0")
    rng = np.random.default_rng(515)
    NT, NWC, NLAY = 7, 60, 9
    out = {}
    ID = np.array([39, 40, 6, 22, 2]); ISO = np.array([0, 0, 0, 0, 0]); NVMR = 5
    atm = types.SimpleNamespace(ID=ID, ISO=ISO, NVMR=NVMR)
    PRESS = np.logspace(5, 1, NLAY)
    q = rng.uniform(0.01, 0.4, (NLAY, NVMR)); q /= q.sum(1, keepdims=True)
    TEMP = np.linspace(60.0, 420.0, NLAY)                     # below and above the table's 80..400 K
    FRAC = np.linspace(0.2, 0.55, NLAY)
    lay = types.SimpleNamespace(NLAY=NLAY, PP=q * PRESS[:, None], PRESS=PRESS, TEMP=TEMP, FRAC=FRAC,
                                TOTAM=10.0 ** rng.uniform(26, 29, NLAY), DELH=rng.uniform(5e3, 3e4, NLAY))
    for tag, inormal, npara in (("eq", 0, 0), ("normal", 1, 0), ("para", 0, 5)):
        cia = cm.CIA_0(INORMAL=inormal, NT=NT, NWAVE=NWC, NPARA=npara)
        cia.WAVEN = np.linspace(0.0, 6000.0, NWC)
        cia.TEMP = np.linspace(80.0, 400.0, NT)
        npe = max(npara, 1)
        if npara > 0:
            cia.FRAC = np.linspace(0.25, 0.5, npara)
        cia.K_CIA = 10.0 ** rng.uniform(-48, -44, (cia.NPAIR, npe, NT, NWC))
        for space, WAVEC in ((0, np.linspace(100.0, 5900.0, 83)), (1, np.linspace(1.8, 60.0, 71))):
            tau, dtau = fm.ForwardModel_0.calc_tau_cia(None, ISPACE=space, WAVEC=WAVEC, CIA=cia, Atmosphere=atm, Layer=lay)
            key = f"{tag}_{space}"
            out[key + "_WAVEC"] = WAVEC; out[key + "_tau"] = tau; out[key + "_dtau"] = dtau
            WAVEN = WAVEC if space == 0 else np.sort(1e4 / WAVEC)
            out[key + "_kco2"] = cm.co2cia(WAVEN); out[key + "_kn2n2"] = cm.n2n2cia(WAVEN); out[key + "_kn2h2"] = cm.n2h2cia(WAVEN)
        out[tag + "_K_CIA"] = cia.K_CIA; out[tag + "_FRACGRID"] = np.asarray(cia.FRAC, float)
        out[tag + "_meta"] = np.array([inormal, npara])
        out[tag + "_IPAIRG1"] = np.array([int(g) for g in cia.IPAIRG1]); out[tag + "_IPAIRG2"] = np.array([int(g) for g in cia.IPAIRG2])
        out[tag + "_INORMALT"] = np.array([int(g) for g in cia.INORMALT])
        out[tag + "_INORMALD"] = np.array(cia.locate_INORMAL_pairs())
        out["CIA_WAVEN"] = cia.WAVEN; out["CIA_TEMP"] = cia.TEMP
    out.update(ID=ID, ISO=ISO, PP=lay.PP, PRESS=PRESS, TEMP=TEMP, FRAC=FRAC, TOTAM=lay.TOTAM, DELH=lay.DELH)
    np.savez_compressed(os.path.join(OUT, "tau_cia.npz"), **out)
    print({k: np.shape(v) for k, v in out.items() if k.endswith("_tau")}, float(out["eq_0_tau"].max()))


if __name__ == "__main__":
    main()
