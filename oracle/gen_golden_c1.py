"""C1 ("plumbing") golden: run the REFERENCE's nemesisfm end-to-end on the Jupiter CIRS nadir text
inputs (tests/files/Jupiter_CIRS_nadir_thermal_emission, ISCAT=0, ILBL=0) with synthetic .kta
tables written by the reference's own write_ktable, and capture what CIRSrad reads and returns
(ForwardModel_0.py:4376-4511).  The real k-tables are absent from the reference tree
(.MISSING_LARGE_BLOBS), so the spectrum is not physical; the seam contract is what is pinned.

CIRSrad is independent per wavenumber, so the fixture keeps every STRIDE-th wavenumber of the
captured inputs/outputs (small file).  Build container only.

    python oracle/gen_golden_c1.py [--grad]      # -> tests/golden/c1_cirsrad.npz
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

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
STRIDE = 9

GASES = [("c2h2ZERO", 26, 0), ("c2h6ZERO", 27, 0), ("ch4ONE", 6, 1), ("ch4TWO", 6, 2), ("ch4THREE", 6, 3),
         ("ph3ZERO", 28, 0), ("nh3ZERO", 11, 0)]


def main():
    want_grad = "--grad" in sys.argv
    ans = import_reference()
    sp_mod = sys.modules["archnemesis.Spectroscopy_0"]
    fm_mod = sys.modules["archnemesis.ForwardModel_0"]
    src = os.path.join(REFERENCE_ROOT, "tests", "files", "Jupiter_CIRS_nadir_thermal_emission")
    work = tempfile.mkdtemp(prefix="ansfm_c1_")
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), os.path.join(work, f))
        os.chmod(os.path.join(work, f), 0o644)
    # synthetic k-tables: nu 5..1500 step 2.5, G=10 Gauss-Legendre on [0,1],
    # NP=12 logspace(-7,1.2) atm, NT=8 linspace(70,400) K
    rng = np.random.default_rng(1)
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

    captured = {}
    orig = fm_mod.ForwardModel_0.CIRSrad

    def wrapper(self, return_grad=False):
        res = orig(self, return_grad)
        key = "g" if return_grad else "f"
        if key not in captured:
            captured[key] = (self, res)
        return res

    fm_mod.ForwardModel_0.CIRSrad = wrapper
    # continuum gradient terms are locals of calculate_layer_opacity (:3938-3981): capture what the
    # three host routines return on the return_grad=True pass and assemble dTAUCON exactly as :3941-3981
    cont = {}
    o_cia = fm_mod.ForwardModel_0.calculate_vertical_cia_opacity
    o_ray = fm_mod.ForwardModel_0.calc_tau_rayleigh
    o_dust = fm_mod.ForwardModel_0.calc_tau_dust

    def w_cia(self, return_grad=False):
        r = o_cia(self, return_grad)
        if return_grad:
            cont["cia"] = r
        return r

    def w_ray(self, *a, **k):
        r = o_ray(self, *a, **k)
        cont["ray"] = r
        return r

    def w_dust(self, *a, **k):
        r = o_dust(self, *a, **k)
        cont["dust"] = r
        return r

    fm_mod.ForwardModel_0.calculate_vertical_cia_opacity = w_cia
    fm_mod.ForwardModel_0.calc_tau_rayleigh = w_ray
    fm_mod.ForwardModel_0.calc_tau_dust = w_dust
    cwd = os.getcwd()
    os.chdir(work)
    try:
        Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
        FM = ans.ForwardModel_0(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec,
                                Stellar=Stel, Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
        t = time.time()
        SPECONV = FM.nemesisfm()
        print("nemesisfm", time.time() - t, "s", SPECONV.shape)
        out = {"SPECONV": SPECONV}
        self, SPECOUT = captured["f"]
        S, L, P, A = self.SpectroscopyX, self.LayerX, self.PathX, self.AtmosphereX
        sel = np.arange(0, S.NWAVE, STRIDE)
        igas = np.array([A.locate_gas(S.ID[i], S.ISO[i]) for i in range(S.NGAS)])
        TAUCIA = L.TAUCIA if getattr(L, "TAUCIA", None) is not None else np.zeros((S.NWAVE, L.NLAY))
        out.update(dict(
            sel=sel, NWAVE_full=S.NWAVE, WAVE=S.WAVE[sel], K=S.K[sel], TPRESS=S.PRESS, TTEMP=S.TEMP, DELG=S.DELG,
            G_ORD=S.G_ORD, ID=np.array(S.ID), ISO=np.array(S.ISO), ILBL=int(S.ILBL),
            LAY_PRESS=L.PRESS, LAY_TEMP=L.TEMP, LAY_AMOUNT=L.AMOUNT, LAY_TOTAM=L.TOTAM, IGAS=igas,
            TAUCIA=TAUCIA[sel], TAURAY=L.TAURAY[sel], TAUDUST=L.TAUDUST[sel], TAUGAS=L.TAUGAS[sel],
            TAUTOT=L.TAUTOT[sel],
            NLAYIN=P.NLAYIN, LAYINC=P.LAYINC, SCALE=P.SCALE, EMTEMP=P.EMTEMP, IMOD=np.array(P.IMOD).astype(int),
            SOL_ANG=P.SOL_ANG, EMISS_ANG=P.EMISS_ANG, AZI_ANG=P.AZI_ANG,
            TSURF=self.SurfaceX.TSURF, GASGIANT=bool(self.SurfaceX.GASGIANT), LOWBC=int(self.SurfaceX.LOWBC),
            IFORM=int(self.MeasurementX.IFORM), ISPACE=int(self.MeasurementX.ISPACE),
            SOLEXIST=bool(self.StellarX.SOLEXIST), NVMR=A.NVMR, NDUST=self.ScatterX.NDUST,
            SPECOUT=SPECOUT[sel]))
        if want_grad:
            t = time.time()
            SPECONVg, dSPECONV = FM.nemesisfmg()
            print("nemesisfmg", time.time() - t, "s")
            selfg, (SPg, dSP, dTS) = captured["g"]
            Sg, Lg, Ag, Scg = selfg.SpectroscopyX, selfg.LayerX, selfg.AtmosphereX, selfg.ScatterX
            NPAR = Ag.NVMR + 2 + Scg.NDUST
            dTAUCON = np.zeros((Sg.NWAVE, NPAR, Lg.NLAY))
            TAUCIA_g, dTAUCIA = cont["cia"]
            if dTAUCIA is not None:                                                  # :3940-3942
                dTAUCON[:, 0:Ag.NVMR, :] += np.transpose(np.transpose(dTAUCIA[:, :, 0:Ag.NVMR], axes=(2, 0, 1)) / (Lg.TOTAM.T), axes=(1, 0, 2))
                dTAUCON[:, Ag.NVMR, :] += dTAUCIA[:, :, Ag.NVMR]
            TAURAY_g, dTAURAY = cont["ray"]
            if dTAURAY is not None:                                                  # :3955-3957
                for i in range(Ag.NVMR):
                    dTAUCON[:, i, :] += dTAURAY[:, :]
            TAUDUST1, TAUCLSCAT, dTAUDUST1, dTAUCLSCAT = cont["dust"]
            for i in range(Scg.NDUST):                                               # :3978-3980
                dTAUCON[:, Ag.NVMR + 1 + i, :] += dTAUDUST1[:, :, i]
            out.update(dict(SPECOUTg=SPg[sel], dSPECOUT=dSP[sel], dTSURF=dTS[sel], SPECONVg=SPECONVg,
                            dSPECONV=dSPECONV, dTAUCON=dTAUCON[sel]))
    finally:
        os.chdir(cwd)
        fm_mod.ForwardModel_0.CIRSrad = orig
        fm_mod.ForwardModel_0.calculate_vertical_cia_opacity = o_cia
        fm_mod.ForwardModel_0.calc_tau_rayleigh = o_ray
        fm_mod.ForwardModel_0.calc_tau_dust = o_dust
        shutil.rmtree(work, ignore_errors=True)
    fn = os.path.join(OUT, "c1_cirsrad_grad.npz" if want_grad else "c1_cirsrad.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn) / 1e6, "MB")


if __name__ == "__main__":
    main()
