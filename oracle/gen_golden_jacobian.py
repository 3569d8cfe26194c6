"""Jacobian-harness golden: the REFERENCE's own `jacobian_nemesis(analytical_gradient=False)`
(ForwardModel_0.py:2184-2361) on the C1 inputs (tests/files/Jupiter_CIRS_nadir_thermal_emission: one model-0 temperature
profile of 81 levels, hydrostatic re-adjustment on, CIA + aerosol + Rayleigh continuum, 71 layers) with the synthetic
k-tables of gen_golden_c1.py, cut to the first NKEEP convolution points and NFREE free state-vector elements (the others
FIXed, which jacobian_nemesis skips, :2291).  Kept: xnx, ixrun, YN, KK, the per-state measurement vectors YNtot, and what
every one of the nfm forward models handed to CIRSrad / got back (so the GPU box, where the reference does not exist, can
replay the batch and form KK itself).  Build container only.

    python oracle/gen_golden_jacobian.py        # -> tests/golden/jacobian_c1.npz
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
NKEEP = 40                                                    # convolution points kept (5 ... 83 cm-1)
FREE = (2, 9, 16, 23, 30, 37, 44, 51, 58, 65, 72, 79)         # state-vector elements left free (temperature levels)


def setup_c1(ans, work, seed=1, case="Jupiter_CIRS_nadir_thermal_emission"):
    """Copies the inputs of one of the reference's test cases to `work`, writes the synthetic .kta tables of gen_golden_c1.py
    there, points the .kls at them."""
    sp_mod = sys.modules["archnemesis.Spectroscopy_0"]
    src = os.path.join(REFERENCE_ROOT, "tests", "files", case)
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), os.path.join(work, f))
        os.chmod(os.path.join(work, f), 0o644)
    rng = np.random.default_rng(seed)
    x, w = np.polynomial.legendre.leggauss(10)
    PRESS = np.logspace(-7, 1.2, 12); TEMP = np.linspace(70.0, 400.0, 8)
    names = []
    for name, gid, iso in GASES:
        base = 10.0 ** rng.uniform(-26, -22, size=(599, 1, 1, 1))
        gs = np.sort(10.0 ** rng.uniform(-2, 2, size=(599, 10, 1, 1)), axis=1)
        k = base * gs * PRESS[None, None, :, None] ** 0.1 * (TEMP[None, None, None, :] / 200.0)
        fn = os.path.join(work, f"{name}_synth.kta")
        sp_mod.write_ktable(fn, gid, iso, 0.5 * (x + 1.0), 0.5 * w, PRESS, TEMP, 599, 5.0, 2.5, 0.0, k)
        names.append(fn)
    with open(os.path.join(work, "cirstest.kls"), "w") as f:
        f.write("\n".join(names) + "\n")


def cut_case(ans, cls=None, nkeep=NKEEP, free=FREE):
    """read_input_files('cirstest') in the cwd, cut to nkeep convolution points and the `free` state-vector elements."""
    Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
    Meas.NCONV = np.array([nkeep], dtype="int32")
    Meas.VCONV = Meas.VCONV[:nkeep]; Meas.MEAS = Meas.MEAS[:nkeep]; Meas.ERRMEAS = Meas.ERRMEAS[:nkeep]
    Meas.NY = nkeep
    Var.FIX[:] = 1
    Var.FIX[list(free)] = 0
    cls = cls or ans.ForwardModel_0
    return cls(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec, Stellar=Stel, Scatter=Scat,
               CIA=CIA, Layer=Lay, Variables=Var)


def main():
    ans = import_reference()
    fm_mod = sys.modules["archnemesis.ForwardModel_0"]
    work = tempfile.mkdtemp(prefix="ansfm_jac_")
    setup_c1(ans, work)
    calls = []
    orig = fm_mod.ForwardModel_0.CIRSrad

    def wrapper(self, return_grad=False):
        res = orig(self, return_grad)
        S, L, P, A = self.SpectroscopyX, self.LayerX, self.PathX, self.AtmosphereX
        igas = np.array([A.locate_gas(S.ID[i], S.ISO[i]) for i in range(S.NGAS)])
        z = np.zeros((S.NWAVE, L.NLAY))
        cia = L.TAUCIA if getattr(L, "TAUCIA", None) is not None else z
        calls.append(dict(XN=np.array(self.Variables.XN), PRESS=np.array(L.PRESS), TEMP=np.array(L.TEMP),
                          AMOUNT=np.array(L.AMOUNT[:, igas]), TAUCONT=cia + L.TAUDUST + L.TAURAY, SCALE=np.array(P.SCALE),
                          EMTEMP=np.array(P.EMTEMP), SPECOUT=np.array(res), H=np.array(A.H)))
        if len(calls) == 1:
            calls[0]["static"] = dict(
                WAVE=np.array(S.WAVE), K=np.array(S.K), TPRESS=np.array(S.PRESS), TTEMP=np.array(S.TEMP), DELG=np.array(S.DELG),
                NLAYIN=np.array(P.NLAYIN), LAYINC=np.array(P.LAYINC), IMOD=np.array(P.IMOD).astype(int),
                SOL_ANG=np.array(P.SOL_ANG), EMISS_ANG=np.array(P.EMISS_ANG), TSURF=float(self.SurfaceX.TSURF),
                ISPACE=int(self.MeasurementX.ISPACE), IFORM=int(self.MeasurementX.IFORM))
        elif not (np.array_equal(P.LAYINC, calls[0]["static"]["LAYINC"]) and np.array_equal(S.WAVE, calls[0]["static"]["WAVE"])):
            raise RuntimeError("path structure or calculation grid changed between states")
        return res

    cwd = os.getcwd()
    os.chdir(work)
    try:
        # (1) the method as a retrieval runs it: NCores = 2, two loky worker processes, each with a pickled copy of the
        # forward model (the workers import the reference through the same stand-ins: PYTHONPATH)
        os.environ["PYTHONPATH"] = os.pathsep.join([sys.path[0], REFERENCE_ROOT] + [os.environ.get("PYTHONPATH", "")])
        FM = cut_case(ans)
        XN0 = np.array(FM.Variables.XN)
        t = time.time()
        YN, KK = FM.jacobian_nemesis(NCores=2, analytical_gradient=False)
        print("reference jacobian_nemesis(NCores=2): %.1f s" % (time.time() - t))
        assert np.array_equal(FM.Variables.XN, XN0)
        # (2) once more with NCores = 1 -- joblib then calls execute_fm in THIS process -- to see what every forward model
        # hands to CIRSrad.  In that mode execute_fm's `self.Variables.XN = xnx[:, ixrun[ifm]]` (:2154) mutates the
        # caller's state vector: XN is left at the last perturbed state and the quotient of the last column (:2355-2359)
        # is formed with the perturbed value, i.e. that column comes out divided by 1.05.  Kept as KK_ncores1.
        fm_mod.ForwardModel_0.CIRSrad = wrapper
        FM = cut_case(ans)
        t = time.time()
        YN1, KK1 = FM.jacobian_nemesis(NCores=1, analytical_gradient=False)
        print("reference jacobian_nemesis(NCores=1): %.1f s, %d forward models" % (time.time() - t, len(calls)))
        V, M = FM.Variables, FM.Measurement
        inum = np.where((V.NUM == 1) & (V.FIX == 0))[0]
        ixrun = np.concatenate([[0], inum + 1]).astype("int32")
        assert np.array_equal(YN1, YN)
        assert np.array_equal(KK1[:, inum[:-1]], KK[:, inum[:-1]])
        np.testing.assert_allclose(KK1[:, inum[-1]] * 1.05, KK[:, inum[-1]], rtol=1e-12)
        xnx = np.zeros((V.NX, V.NX + 1)); xnx[:, 0] = XN0
        xnx[:, 1:] = np.repeat(XN0[:, None], V.NX, axis=1) + np.diag(0.05 * XN0)
        blk = xnx[:, 1:]; blk[blk == 0] = 0.05
        for c, ix in zip(calls, ixrun):            # NCores = 1: the forward models ran in ixrun order in this process
            assert np.array_equal(c["XN"], xnx[:, ix])
        st = calls[0].pop("static")
        stack = lambda k: np.stack([c[k] for c in calls])
        # every forward model's measurement vector: FWHM = 0 -> conv is scipy interp1d (linear) of the spectrum at VCONV
        VCONV = np.array(M.VCONV[:NKEEP, 0])
        YNtot = np.stack([np.interp(VCONV, st["WAVE"], c["SPECOUT"][:, 0]) for c in calls], axis=1)
        assert np.allclose(YNtot[:, 0], YN, rtol=1e-13, atol=0)
        out = dict(xnx=xnx, ixrun=ixrun, inum=inum, XN=XN0, FIX=np.array(V.FIX), YN=YN, KK=KK, KK_ncores1=KK1, YNtot=YNtot,
                   VCONV=VCONV,
                   LAY_PRESS=stack("PRESS"), LAY_TEMP=stack("TEMP"), LAY_AMOUNT=stack("AMOUNT"), TAUCONT=stack("TAUCONT"),
                   SCALE=stack("SCALE"), EMTEMP=stack("EMTEMP"), SPECOUT=stack("SPECOUT"), ATM_H=stack("H"), **st)
    finally:
        os.chdir(cwd)
        fm_mod.ForwardModel_0.CIRSrad = orig
        shutil.rmtree(work, ignore_errors=True)
    fn = os.path.join(OUT, "jacobian_c1.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, "%.2f MB" % (os.path.getsize(fn) / 1e6), "KK", KK.shape, "free columns", list(inum))


if __name__ == "__main__":
    main()
