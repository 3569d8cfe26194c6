"""C4 seam golden: the REFERENCE's CIRSrad in its multiple-scattering branch (ForwardModel_0.py:4478-4501 ->
calculate_multiple_scattering_spectrum :4343 -> scloud11wave :5018-5165 -> scloud11wave_core) on the reference's own
scattering test inputs (tests/files/Jupiter_CIRS_angled_thermal_emission_scattering: ISCAT = 1, Rayleigh on, one
Henyey-Greenstein haze, 5 zenith angles, sunlight on) with synthetic .kta tables written by the reference's write_ktable
(the real ones are absent, .MISSING_LARGE_BLOBS).  Captured: everything ansfm_cirsrad_ck_scatter takes (layer properties,
continuum arrays, the host-prepared arguments of scloud11wave_core) and what CIRSrad returns.  The measurement is cut to
its first NKEEP convolution points: the un-jitted core costs ~0.1 s per (wavenumber, g).  Build container only.

    python oracle/gen_golden_c4.py      # -> tests/golden/c4_cirsrad_scatter.npz
"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference, REFERENCE_ROOT  # noqa: E402
from oracle.gen_golden_c1 import GASES  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
NKEEP = 9


def main():
    ans = import_reference()
    sp_mod = sys.modules["archnemesis.Spectroscopy_0"]
    fm_mod = sys.modules["archnemesis.ForwardModel_0"]
    import importlib
    ms_mod = importlib.import_module("archnemesis.Multiple_Scattering_Core")
    src = os.path.join(REFERENCE_ROOT, "tests", "files", "Jupiter_CIRS_angled_thermal_emission_scattering")
    work = tempfile.mkdtemp(prefix="ansfm_c4_")
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), os.path.join(work, f))
        os.chmod(os.path.join(work, f), 0o644)
    rng = np.random.default_rng(4)
    x, w = np.polynomial.legendre.leggauss(10)
    g_ord = 0.5 * (x + 1.0); del_g = 0.5 * w
    PRESS = np.logspace(-7, 1.2, 12); TEMP = np.linspace(70.0, 400.0, 8)
    nwave = 599; vmin = 5.0; delv = 2.5
    names = []
    for name, gid, iso in GASES:
        base = 10.0 ** rng.uniform(-26, -22, size=(nwave, 1, 1, 1))
        gs = np.sort(10.0 ** rng.uniform(-2, 2, size=(nwave, 10, 1, 1)), axis=1)
        k = base * gs * PRESS[None, None, :, None] ** 0.1 * (TEMP[None, None, None, :] / 200.0)
        fn = os.path.join(work, f"{name}_synth.kta")
        sp_mod.write_ktable(fn, gid, iso, g_ord, del_g, PRESS, TEMP, nwave, vmin, delv, 0.0, k)
        names.append(fn)
    with open(os.path.join(work, "cirstest.kls"), "w") as f:
        f.write("\n".join(names) + "\n")

    cap = {}
    o_cirs = fm_mod.ForwardModel_0.CIRSrad
    o_core = ms_mod.scloud11wave_core
    o_cia = fm_mod.ForwardModel_0.calculate_vertical_cia_opacity

    def w_cirs(self, return_grad=False):
        res = o_cirs(self, return_grad)
        cap.setdefault("cirs", (self, res))
        return res

    def w_core(**kw):
        res = o_core(**kw)
        cap.setdefault("core", (dict(kw), res))
        return res

    def w_cia(self, return_grad=False):
        r = o_cia(self, return_grad)
        cap.setdefault("cia", r[0])
        return r

    fm_mod.ForwardModel_0.CIRSrad = w_cirs
    ms_mod.scloud11wave_core = w_core
    fm_mod.ForwardModel_0.calculate_vertical_cia_opacity = w_cia
    cwd = os.getcwd()
    os.chdir(work)
    try:
        Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
        Meas.NCONV = np.array([NKEEP], dtype="int32")
        Meas.VCONV = Meas.VCONV[:NKEEP]; Meas.MEAS = Meas.MEAS[:NKEEP]; Meas.ERRMEAS = Meas.ERRMEAS[:NKEEP]
        Meas.NY = NKEEP
        FM = ans.ForwardModel_0(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec,
                                Stellar=Stel, Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
        t = time.time()
        SPECONV = FM.nemesisfm()
        print("nemesisfm (scattering)", time.time() - t, "s", SPECONV.shape)
        self, SPECOUT = cap["cirs"]
        kw, rad = cap["core"]
        S, L, P, A, Sc = self.SpectroscopyX, self.LayerX, self.PathX, self.AtmosphereX, self.ScatterX
        igas = np.array([A.locate_gas(S.ID[i], S.ISO[i]) for i in range(S.NGAS)])
        out = dict(
            SPECONV=SPECONV, WAVE=S.WAVE, K=S.K, TPRESS=S.PRESS, TTEMP=S.TEMP, DELG=S.DELG, ILBL=int(S.ILBL),
            LAY_PRESS=L.PRESS, LAY_TEMP=L.TEMP, LAY_AMOUNT=L.AMOUNT, IGAS=igas,
            TAUCIA=cap["cia"], TAURAY=L.TAURAY, TAUDUST=L.TAUDUST, TAUSCAT=L.TAUSCAT, TAUCLSCAT=L.TAUCLSCAT,
            TAUGAS=L.TAUGAS, TAUTOT=L.TAUTOT, IMOD=np.array(P.IMOD).astype(int),
            SOL_ANG=P.SOL_ANG, EMISS_ANG=P.EMISS_ANG, AZI_ANG=P.AZI_ANG,
            ISPACE=int(self.MeasurementX.ISPACE), IFORM=int(self.MeasurementX.IFORM),
            NMU=int(Sc.NMU), NF=int(Sc.NF), NPHI=int(Sc.NPHI), IRAY=int(Sc.IRAY), IMIE=int(Sc.IMIE), NDUST=int(Sc.NDUST),
            MU=Sc.MU, WTMU=Sc.WTMU, LOWBC=int(self.SurfaceX.LOWBC), GASGIANT=bool(self.SurfaceX.GASGIANT),
            TSURF=float(self.SurfaceX.TSURF),
            core_phasarr=np.ascontiguousarray(kw["phasarr"]), core_radg=kw["radg"], core_solar=kw["solar"],
            core_brdf=kw["brdf_matrix"], core_bnu=kw["bnu"], core_taus=kw["taus"], core_tauray=kw["tauray"],
            core_omegas=kw["omegas_s"], core_lfrac=np.ascontiguousarray(kw["lfrac"]), core_rad=rad,
            SPECOUT=SPECOUT)
    finally:
        os.chdir(cwd)
        fm_mod.ForwardModel_0.CIRSrad = o_cirs
        ms_mod.scloud11wave_core = o_core
        fm_mod.ForwardModel_0.calculate_vertical_cia_opacity = o_cia
        shutil.rmtree(work, ignore_errors=True)
    fn = os.path.join(OUT, "c4_cirsrad_scatter.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn) / 1e6, "MB")
    for k_, v in out.items():
        if hasattr(v, "shape"):
            print(k_, v.shape)


if __name__ == "__main__":
    main()
