"""Jacobian-harness golden for the scattering configuration: the REFERENCE's own `jacobian_nemesis` on its multiple-scattering
test inputs (tests/files/Jupiter_CIRS_angled_thermal_emission_scattering: ISCAT = 1, so NUM[:] = 1 and every free element
costs one multiple-scattering forward model, ForwardModel_0.py:2251-2252), synthetic k-tables (seed 4, as gen_golden_c4.py), cut
to NKEEP convolution points and five free elements: three temperature levels (model 0) and two parameters of the aerosol
profile (model 47) -- the un-jitted core costs ~0.1 s per (wavenumber, g).  Kept: xnx, ixrun, YN, KK (two loky workers),
KK_ncores1, and for every forward model what CIRSrad's scattering branch read (layer properties, continuum arrays, the
host-prepared arguments of scloud11wave_core that have no g axis) and returned.  Build container only.

    python oracle/gen_golden_jacobian_ms.py        # -> tests/golden/jacobian_c4.npz
"""
import importlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference, REFERENCE_ROOT  # noqa: E402
from oracle import gen_golden_jacobian as gj  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CASE = "Jupiter_CIRS_angled_thermal_emission_scattering"
NKEEP = 6
FREE = (12, 30, 48, 81, 82)


def main():
    ans = import_reference()
    fm_mod = sys.modules["archnemesis.ForwardModel_0"]
    ms_mod = importlib.import_module("archnemesis.Multiple_Scattering_Core")
    work = tempfile.mkdtemp(prefix="ansfm_jacms_")
    gj.setup_c1(ans, work, seed=4, case=CASE)
    calls = []
    o_cirs, o_core, o_cia = fm_mod.ForwardModel_0.CIRSrad, ms_mod.scloud11wave_core, fm_mod.ForwardModel_0.calculate_vertical_cia_opacity
    pend = {}

    def w_core(**kw):
        pend["core"] = dict(kw)
        return o_core(**kw)

    def w_cia(self, return_grad=False):
        r = o_cia(self, return_grad)
        pend["cia"] = r[0]
        return r

    def w_cirs(self, return_grad=False):
        res = o_cirs(self, return_grad)
        S, L, P, A, Sc = self.SpectroscopyX, self.LayerX, self.PathX, self.AtmosphereX, self.ScatterX
        igas = np.array([A.locate_gas(S.ID[i], S.ISO[i]) for i in range(S.NGAS)])
        kw = pend["core"]
        calls.append(dict(XN=np.array(self.Variables.XN), PRESS=np.array(L.PRESS), TEMP=np.array(L.TEMP), AMOUNT=np.array(L.AMOUNT[:, igas]),
                          TAUCIA=np.array(pend["cia"]), TAUDUST=np.array(L.TAUDUST), TAURAY=np.array(L.TAURAY), TAUSCAT=np.array(L.TAUSCAT),
                          LFRAC=np.ascontiguousarray(kw["lfrac"]), RADG=np.array(kw["radg"]), PHASARR=np.ascontiguousarray(kw["phasarr"]),
                          SPECOUT=np.array(res)))
        if len(calls) == 1:
            calls[0]["static"] = dict(
                WAVE=np.array(S.WAVE), K=np.array(S.K), TPRESS=np.array(S.PRESS), TTEMP=np.array(S.TEMP), DELG=np.array(S.DELG),
                IMOD=np.array(P.IMOD).astype(int), SOL_ANG=np.array(P.SOL_ANG), EMISS_ANG=np.array(P.EMISS_ANG), AZI_ANG=np.array(P.AZI_ANG),
                ISPACE=int(self.MeasurementX.ISPACE), MU=np.array(Sc.MU), WTMU=np.array(Sc.WTMU), NF=int(Sc.NF), NPHI=int(Sc.NPHI),
                IRAY=int(Sc.IRAY), IMIE=int(Sc.IMIE), LOWBC=int(self.SurfaceX.LOWBC), SOLAR=np.array(kw["solar"]),
                BRDF=np.array(kw["brdf_matrix"]))
        return res

    cwd = os.getcwd()
    os.chdir(work)
    try:
        os.environ["PYTHONPATH"] = os.pathsep.join([sys.path[0], REFERENCE_ROOT] + [os.environ.get("PYTHONPATH", "")])
        FM = gj.cut_case(ans, nkeep=NKEEP, free=FREE)
        XN0 = np.array(FM.Variables.XN)
        t = time.time()
        YN, KK = FM.jacobian_nemesis(NCores=2, analytical_gradient=True)       # ISCAT = 1 forces the numerical route anyway
        print("reference jacobian_nemesis(NCores=2): %.1f s" % (time.time() - t))
        fm_mod.ForwardModel_0.CIRSrad = w_cirs
        ms_mod.scloud11wave_core = w_core
        fm_mod.ForwardModel_0.calculate_vertical_cia_opacity = w_cia
        FM = gj.cut_case(ans, nkeep=NKEEP, free=FREE)
        t = time.time()
        YN1, KK1 = FM.jacobian_nemesis(NCores=1, analytical_gradient=True)
        print("reference jacobian_nemesis(NCores=1): %.1f s, %d forward models" % (time.time() - t, len(calls)))
        V, M = FM.Variables, FM.Measurement
        assert np.all(V.NUM == 1)
        inum = np.where((V.NUM == 1) & (V.FIX == 0))[0]
        ixrun = np.concatenate([[0], inum + 1]).astype("int32")
        assert np.array_equal(YN1, YN) and np.array_equal(KK1[:, inum[:-1]], KK[:, inum[:-1]])
        xnx = np.zeros((V.NX, V.NX + 1)); xnx[:, 0] = XN0
        xnx[:, 1:] = np.repeat(XN0[:, None], V.NX, axis=1) + np.diag(0.05 * XN0)
        blk = xnx[:, 1:]; blk[blk == 0] = 0.05
        for c, ix in zip(calls, ixrun):
            assert np.array_equal(c["XN"], xnx[:, ix])
        st = calls[0].pop("static")
        stack = lambda k: np.stack([c[k] for c in calls])
        VCONV = np.array(M.VCONV[:NKEEP, 0])
        YNtot = np.stack([np.interp(VCONV, st["WAVE"], c["SPECOUT"][:, 0]) for c in calls], axis=1)
        assert np.allclose(YNtot[:, 0], YN, rtol=1e-13, atol=0)
        out = dict(xnx=xnx, ixrun=ixrun, inum=inum, XN=XN0, FIX=np.array(V.FIX), YN=YN, KK=KK, KK_ncores1=KK1, YNtot=YNtot, VCONV=VCONV,
                   LAY_PRESS=stack("PRESS"), LAY_TEMP=stack("TEMP"), LAY_AMOUNT=stack("AMOUNT"), TAUCIA=stack("TAUCIA"),
                   TAUDUST=stack("TAUDUST"), TAURAY=stack("TAURAY"), TAUSCAT=stack("TAUSCAT"), LFRAC=stack("LFRAC"), RADG=stack("RADG"),
                   PHASARR=stack("PHASARR"), SPECOUT=stack("SPECOUT"), **st)
    finally:
        os.chdir(cwd)
        fm_mod.ForwardModel_0.CIRSrad = o_cirs
        ms_mod.scloud11wave_core = o_core
        fm_mod.ForwardModel_0.calculate_vertical_cia_opacity = o_cia
        shutil.rmtree(work, ignore_errors=True)
    fn = os.path.join(OUT, "jacobian_c4.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, "%.2f MB" % (os.path.getsize(fn) / 1e6), "KK", KK.shape, "free columns", [int(i) for i in inum])


if __name__ == "__main__":
    main()
