"""Drop-in plumbing against the REAL reference (build container only; skipped where /root/reference is absent).

The reference's own ForwardModel_0.nemesisfm()/nemesisfmg() -- its host preparation (subprofretg, calc_path,
continuum opacities, convolution) untouched -- is run through `make_gpu_forward_model`, whose CIRSrad seam is what
normally calls libansfm.so.  There is no GPU in this container, so the engine is replaced by a TEST DOUBLE backed by
the CPU oracle: what is checked here is the adapter (argument mapping, units, dtype semantics, side products), not a
kernel.  The same seam with the real engine is covered on the GPU by tests/test_c1_seam.py."""
import os
import shutil
import sys
import tempfile

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = [pytest.mark.needs_reference,
              pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")]


class OracleEngineDouble:
    """Engine test double: the AnsfmEngine methods the mixin uses, answered by the CPU oracle."""
    def __init__(self, orc):
        self.orc = orc

    def upload_ktable(self, K, PRESS, TEMP, WAVE, DELG):
        self.t = (K, PRESS, TEMP, WAVE, DELG)
        self.dims = K.shape

    def upload_ktable_files(self, paths, wavemin, wavemax):
        """what the engine does on the GPU, answered here with the reference's reader"""
        import importlib
        sp = importlib.import_module("archnemesis.Spectroscopy_0")
        self.file_uploads = getattr(self, "file_uploads", 0) + 1
        r = [sp.read_ktable(p, wavemin, wavemax) for p in paths]
        h = sp.read_ktahead(paths[-1])
        K = np.stack([x[12] for x in r], axis=-1)
        self.upload_ktable(K, h[10], h[11], r[0][3], h[9])
        return r[0][3], h[10], h[11], h[9]

    def cirsrad_ck_thermal(self, ISPACE, lp, lt, am, cont, NLAYIN, LAYINC, SCALE, EMTEMP, TSURF, EMISSIVITY=None,
                           SOL_ANG=None, EMISS_ANG=None, xfac=None, **kw):
        K, P, T, W, D = self.t
        if np.ndim(lp) == 2:                         # leading model axis (the engine's batched call): one model at a time
            n = np.shape(lp)[0]
            self.batch_sizes = getattr(self, "batch_sizes", []) + [n]
            per = lambda a, i: None if a is None else (np.asarray(a)[i] if np.ndim(a) == 3 else np.asarray(a))
            return np.stack([self.cirsrad_ck_thermal(ISPACE, lp[i], lt[i], am[i], None if cont is None else np.asarray(cont)[i],
                                                     NLAYIN, LAYINC, per(SCALE, i), per(EMTEMP, i), float(np.atleast_1d(TSURF)[i]),
                                                     EMISSIVITY=EMISSIVITY, SOL_ANG=SOL_ANG, EMISS_ANG=EMISS_ANG, xfac=xfac)
                             for i in range(n)])
        out, self.tg = self.orc.cirsrad_ck_thermal(ISPACE, K, P, T, W, D, lp, lt, am, cont, NLAYIN, LAYINC, SCALE, EMTEMP,
                                                   TSURF, EMISSIVITY=EMISSIVITY, SOL_ANG=SOL_ANG, EMISS_ANG=EMISS_ANG,
                                                   xfac=xfac, return_taugas=True)
        return out

    def cirsradg_ck_thermal(self, ISPACE, lp, lt, am, cont, dcont, NVMR, NPAR, igas, NLAYIN, LAYINC, SCALE, EMTEMP, TSURF,
                            EMISSIVITY=None, xfac=None):
        K, P, T, W, D = self.t
        r = self.orc.cirsradg_ck_thermal(ISPACE, K, P, T, W, D, lp, lt, am, cont, dcont, NVMR, NPAR, igas, NLAYIN, LAYINC,
                                         SCALE, EMTEMP, TSURF, EMISSIVITY=EMISSIVITY, xfac=xfac)
        self.tg = self.orc.cirsrad_ck_thermal(ISPACE, K, P, T, W, D, lp, lt, am, cont, NLAYIN, LAYINC, SCALE, EMTEMP, TSURF,
                                              return_taugas=True)[1]
        return r

    def cirsrad_ck_scatter(self, ISPACE, lp, lt, am, TAUCIA, TAUDUST, TAURAY, TAUSCAT, phasarr, lfrac, radg, sol, emi, aph,
                           solar, lowbc, brdf, mu1, wt1, nf, nphi, iray, imie, xfac=None, return_spec_g=False):
        """ansfm_cirsrad_ck_scatter answered by the oracle's pieces in the reference's order (:3989, :5099-5119, :4504)."""
        K, P, T, W, D = self.t
        self.scatter_calls = getattr(self, "scatter_calls", 0) + 1
        k = self.orc.calc_k(K, P, T, np.asarray(lp) / 101325.0, lt)
        self.tg = self.orc.k_overlap(D, k, am)
        z = np.zeros((len(W), len(lp)))
        TAUCIA, TAUDUST, TAURAY, TAUSCAT = (z if a is None else np.asarray(a) for a in (TAUCIA, TAUDUST, TAURAY, TAUSCAT))
        tautot = self.tg + TAUCIA[:, None, :] + TAUDUST[:, None, :] + TAURAY[:, None, :]
        omega = np.zeros_like(tautot)
        pos = tautot > 0
        omega[pos] = np.broadcast_to((TAURAY + TAUSCAT)[:, None, :], tautot.shape)[pos] / tautot[pos]
        bnu = np.stack([self.orc.planck(ISPACE, W, t) for t in lt], axis=1)
        rad = self.orc.scloud11wave_core(phasarr, radg, sol, emi, solar, aph, lowbc, brdf, mu1, wt1, nf, W, bnu, tautot, TAURAY,
                                         omega, nphi, iray, imie, lfrac)
        spec = np.transpose(rad, (2, 1, 0))
        out = np.tensordot(spec, np.asarray(D, dtype=float), axes=([1], [0]))
        return (out, spec) if return_spec_g else out

    def cirsrad_ck_scatter_batch(self, ISPACE, lp, lt, am, TAUCIA, TAUDUST, TAURAY, TAUSCAT, phasarr, lfrac, radg, sol, emi, aph,
                                 solar, lowbc, brdf, mu1, wt1, nf, nphi, iray, imie, xfac=None):
        """ansfm_cirsrad_ck_scatter_batch: model by model through the single-model answer"""
        n = np.shape(lp)[0]
        self.scatter_batches = getattr(self, "scatter_batches", []) + [n]
        at = lambda a, m: None if a is None else np.asarray(a)[m]
        return np.stack([self.cirsrad_ck_scatter(ISPACE, lp[m], lt[m], am[m], at(TAUCIA, m), at(TAUDUST, m), at(TAURAY, m),
                                                 at(TAUSCAT, m), phasarr, at(lfrac, m), radg[m], sol, emi, aph, solar, lowbc, brdf,
                                                 mu1, wt1, nf, nphi, iray, imie, xfac=xfac) for m in range(n)])

    def cirsrad_ck_singlescatt(self, ISPACE, lp, lt, am, taucont, tausca, phase, NLAYIN, LAYINC, SCALE, EMTEMP, TSURF, EMIS,
                               BRDF, SOLF, sol, emi, xfac=None):
        """ansfm_cirsrad_ck_singlescatt answered by the oracle's pieces (:3989, :4276-4283, :4006, :6509, :4504)."""
        K, P, T, W, D = self.t
        self.ss_calls = getattr(self, "ss_calls", 0) + 1
        k = self.orc.calc_k(K, P, T, np.asarray(lp) / 101325.0, lt)
        self.tg = self.orc.k_overlap(D, k, am)
        tautot = self.tg + np.asarray(taucont)[:, None, :]
        omega = np.where(tautot > 0, np.asarray(tausca)[:, None, :] / np.where(tautot > 0, tautot, 1.0), 0.0)
        NP_ = len(np.atleast_1d(sol))
        out = np.zeros((len(W), NP_))
        xf = np.ones(len(W)) if xfac is None else np.asarray(xfac)
        for ip in range(NP_):
            n = int(NLAYIN[ip]); li = np.asarray(LAYINC)[:n, ip]
            sp = self.orc.calc_singlescatt_plane_spectrum(ISPACE, W, tautot[:, :, li] * np.asarray(SCALE)[:n, ip],
                                                          np.asarray(EMTEMP)[:n, ip], omega[:, :, li], np.asarray(phase)[ip][:, li],
                                                          TSURF, EMIS, np.asarray(BRDF)[:, ip], SOLF, np.atleast_1d(sol)[ip],
                                                          np.atleast_1d(emi)[ip])
            out[:, ip] = np.tensordot(sp * xf[:, None], np.asarray(D, dtype=float), axes=([1], [0]))
        return out

    def _transmission(self, lp, lt, am, taucont, NLAYIN, LAYINC, SCALE, xfac):
        K, P, T, W, D = self.t
        k, dkdT = self.orc.calc_k(K, P, T, np.asarray(lp) / 101325.0, lt, grad=True)
        self.tg, dk = self.orc.k_overlapg(D, k, dkdT, am)
        tautot = self.tg + (0.0 if taucont is None else np.asarray(taucont)[:, None, :])
        LAYINC = np.asarray(LAYINC); SCALE = np.asarray(SCALE, dtype=float)
        sg = np.exp(-np.sum(tautot[:, :, LAYINC] * SCALE, axis=2))                # :4006-4009, :4115
        if xfac is not None:
            sg = sg * np.asarray(xfac)[:, None, None]
        return sg, dk, np.asarray(D, dtype=float)

    def cirsrad_ck_transmission(self, lp, lt, am, taucont, NLAYIN, LAYINC, SCALE, xfac=None):
        if np.ndim(lp) == 2:                         # leading model axis (the engine's batched call): one model at a time
            n = np.shape(lp)[0]
            self.batch_sizes = getattr(self, "batch_sizes", []) + [n]
            sc = np.asarray(SCALE)
            return np.stack([self.cirsrad_ck_transmission(lp[i], lt[i], am[i], None if taucont is None else np.asarray(taucont)[i],
                                                          NLAYIN, LAYINC, sc[i] if sc.ndim == 3 else sc, xfac=xfac) for i in range(n)])
        self.tr_calls = getattr(self, "tr_calls", 0) + 1
        sg, _, D = self._transmission(lp, lt, am, taucont, NLAYIN, LAYINC, SCALE, xfac)
        return np.tensordot(sg, D, axes=([1], [0]))

    def cirsradg_ck_transmission(self, lp, lt, am, taucont, dtaucon, NVMR, NPAR, igas_map, NLAYIN, LAYINC, SCALE, xfac=None):
        """ansfm_cirsradg_ck_transmission answered by the oracle's calc_kg + k_overlapg (:3868-3872, :4012, :4129, :4507)."""
        self.trg_calls = getattr(self, "trg_calls", 0) + 1
        sg, dk, D = self._transmission(lp, lt, am, taucont, NLAYIN, LAYINC, SCALE, xfac)
        W, G, L, S1 = dk.shape
        dtau = np.zeros((W, G, NPAR, L))
        for i in range(S1 - 1):
            dtau[:, :, igas_map[i], :] = dk[:, :, :, i] * 1.0e-4
        dtau[:, :, NVMR, :] = dk[:, :, :, S1 - 1]
        if dtaucon is not None:
            dtau += np.asarray(dtaucon)[:, None, :, :]
        dlay = dtau[:, :, :, np.asarray(LAYINC)] * np.asarray(SCALE, dtype=float)
        dspec = np.nan_to_num(np.tensordot(-sg[:, :, None, None, :] * dlay, D, axes=([1], [0])))
        return np.tensordot(sg, D, axes=([1], [0])), dspec

    def layer_average(self, RADIUS, H, P, T, ID, VMR, DUST, PARAH2, BASEH, BASEP=None, **k):
        self.lay_calls = getattr(self, "lay_calls", 0) + 1
        if np.ndim(H) == 2:                          # leading state axis: the engine's batched launch, one state at a time
            n = np.shape(H)[0]
            pick = lambda a, i: None if a is None else (np.asarray(a)[i] if np.ndim(a) >= 2 and np.shape(a)[0] == n else a)
            xm = k.pop("XMOLWT", None)
            rows = [self.orc.layer_average(RADIUS, H[i], pick(P, i), T[i], ID, VMR[i], pick(DUST, i), pick(PARAH2, i),
                                           np.asarray(BASEH).reshape(-1, np.shape(BASEH)[-1])[i if np.ndim(BASEH) == 2 else 0], None,
                                           XMOLWT=xm, **k) for i in range(n)]
            return tuple(np.stack([r[j] for r in rows]) for j in range(len(rows[0])))
        return self.orc.layer_average(RADIUS, H, P, T, ID, VMR, DUST, PARAH2, BASEH, BASEP, **k)

    # ---- what the batched ("profile") route calls with torch tensors: answered on the CPU ------------------------------
    @property
    def torch_device(self):
        import torch
        return torch.device("cpu")

    device = 0

    def synchronize(self):
        pass

    def last_layer_rows(self):
        return getattr(self, "_rows", (0, 0))

    def calc_tau_rayleigh_batch_dev(self, IRAY, ISPACE, TOTAM, out, ID=None, ISO=None, VMR=None, variant=None):
        import torch
        K, P, T, W, D = self.t
        for i in range(np.shape(TOTAM)[0]):
            tau = self.orc.calc_tau_rayleigh(int(IRAY), int(ISPACE), W, TOTAM[i], ID, ISO, None if VMR is None else VMR[i])[0]
            out[i] = torch.as_tensor(tau)
        return out

    def cirsrad_ck_thermal_dev(self, ISPACE, n, L, lp, lt, am, cont, P_, LIMAX, NLAYIN, LAYINC, SCALE, EMTEMP, TSURF, EMIS, SOLF,
                               REFL, SOL, EMI, xfac, out):
        import torch
        h = lambda a: None if a is None else a.numpy()
        self.dev_batches = getattr(self, "dev_batches", []) + [int(n)]
        for i in range(n):
            sp = self.cirsrad_ck_thermal(ISPACE, h(lp)[i], h(lt)[i], h(am)[i], None if cont is None else h(cont)[i], h(NLAYIN),
                                         h(LAYINC), h(SCALE)[i], h(EMTEMP)[i], float(h(TSURF)[i]), EMISSIVITY=h(EMIS), xfac=h(xfac))
            out[i] = torch.as_tensor(sp)
        self._rows = (n * L, n * L)

    def layer_averageg(self, *a, **k):
        self.lay_calls = getattr(self, "lay_calls", 0) + 1
        return self.orc.layer_averageg(*a, **k)

    def calc_tau_cia(self, *a, **k):
        self.cia_calls = getattr(self, "cia_calls", 0) + 1
        with_grad = k.pop("with_grad", True)
        r = self.orc.calc_tau_cia(*a, **k)
        return r if with_grad else r[0]

    def calc_tau_rayleigh(self, IRAY, ISPACE, WAVEC, TOTAM, ID=None, ISO=None, VMR=None, variant=None):
        self.ray_calls = getattr(self, "ray_calls", 0) + 1
        if int(IRAY) == 0:
            z = np.zeros((len(WAVEC), len(TOTAM)))
            return z, z.copy()
        return self.orc.calc_tau_rayleigh(int(IRAY), int(ISPACE), WAVEC, TOTAM, ID, ISO, VMR)

    def calc_tau_dust(self, *a):
        self.dust_calls = getattr(self, "dust_calls", 0) + 1
        return self.orc.calc_tau_dust(*a)

    def map2pro(self, *a, **k):
        self.map_calls = getattr(self, "map_calls", 0) + 1
        return self.orc.map2pro(*a, **k)

    def map2xvec(self, *a, **k):
        self.map_calls = getattr(self, "map_calls", 0) + 1
        return self.orc.map2xvec(*a, **k)

    def get_taugas(self, L, model=0):
        return self.tg

    def set_gradient_gases(self, gases=None, temperature=True):
        """recorded only: the double returns every gas's gradient whatever is selected"""
        self.gas_selections = getattr(self, "gas_selections", []) + [None if gases is None else (sorted(gases), temperature)]

    # ILS convolution family
    def _conv(self, name, *a, **k):
        self.conv_calls = getattr(self, "conv_calls", []) + [name]
        return a, k

    def lblconv(self, nw, vw, y, nc, vc, ish, fw):
        self._conv("lblconv"); return self.orc.lblconv(nw, vw, y, nc, vc, ish, fw)

    def lblconvg(self, nw, vw, y, dy, nc, vc, ish, fw):
        self._conv("lblconvg"); return self.orc.lblconv(nw, vw, y, nc, vc, ish, fw, dydx=dy)

    def lblconv_fil(self, nw, vw, y, nc, vc, nf, vf, af):
        self._conv("lblconv_fil"); return self.orc.lblconv_fil(nw, vw, y, nc, vc, nf, vf, af)

    def lblconvg_fil(self, nw, vw, y, dy, nc, vc, nf, vf, af):
        self._conv("lblconvg_fil"); return self.orc.lblconv_fil(nw, vw, y, nc, vc, nf, vf, af, dydx=dy)

    def lblconv_ngeom(self, nw, vw, y, nc, vc, ish, fw):
        self._conv("lblconv_ngeom"); return self.orc.lblconv(nw, vw, y, nc, vc, ish, fw, ngeom=True)

    def lblconvg_ngeom(self, nw, vw, y, dy, nc, vc, ish, fw):
        self._conv("lblconvg_ngeom"); return self.orc.lblconv(nw, vw, y, nc, vc, ish, fw, dydx=dy, ngeom=True)

    def lblconv_fil_ngeom(self, nw, vw, y, nc, vc, nf, vf, af):
        self._conv("lblconv_fil_ngeom"); return self.orc.lblconv_fil(nw, vw, y, nc, vc, nf, vf, af, ngeom=True)

    def lblconvg_fil_ngeom(self, nw, vw, y, dy, nc, vc, nf, vf, af):
        self._conv("lblconvg_fil_ngeom"); return self.orc.lblconv_fil(nw, vw, y, nc, vc, nf, vf, af, dydx=dy, ngeom=True)


@pytest.fixture(scope="module")
def c1_run(oracle):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.ref_import import import_reference
    from oracle.gen_golden_c1 import GASES
    ans = import_reference()
    sp_mod = sys.modules["archnemesis.Spectroscopy_0"]
    src = os.path.join(REF, "tests", "files", "Jupiter_CIRS_nadir_thermal_emission")
    work = tempfile.mkdtemp(prefix="ansfm_dropin_")
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), os.path.join(work, f))
        os.chmod(os.path.join(work, f), 0o644)
    rng = np.random.default_rng(1)
    x, w = np.polynomial.legendre.leggauss(10)
    g_ord = 0.5 * (x + 1.0); del_g = 0.5 * w
    PRESS = np.logspace(-7, 1.2, 12); TEMP = np.linspace(70.0, 400.0, 8)
    nwave = 599
    names = []
    for name, gid, iso in GASES:
        base = 10.0 ** rng.uniform(-26, -22, size=(nwave, 1, 1, 1))
        gs = np.sort(10.0 ** rng.uniform(-2, 2, size=(nwave, 10, 1, 1)), axis=1)
        k = base * gs * PRESS[None, None, :, None] ** 0.1 * (TEMP[None, None, None, :] / 200.0)
        fn = os.path.join(work, f"{name}_synth.kta")
        sp_mod.write_ktable(fn, gid, iso, g_ord, del_g, PRESS, TEMP, nwave, 5.0, 2.5, 0.0, k)
        names.append(fn)
    with open(os.path.join(work, "cirstest.kls"), "w") as f:
        f.write("\n".join(names) + "\n")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        yield ans
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


def test_scattering_nemesisfm_through_the_adapter_matches_the_reference(oracle, golden_dir, monkeypatch):
    """ISCAT = 1: the reference's nemesisfm on its own multiple-scattering test inputs (Jupiter CIRS, haze + Rayleigh,
    sunlight) through the adapter -- CIRSrad's scattering branch lands on ansfm_cirsrad_ck_scatter (here: the double)
    with the host-prepared arguments of scloud11wave, and the convolved spectrum is the reference's
    (oracle/gen_golden_c4.py ran the same inputs through the unmodified reference)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.ref_import import import_reference
    from oracle.gen_golden_c1 import GASES
    from oracle.gen_golden_c4 import NKEEP
    ans = import_reference()
    sp_mod = sys.modules["archnemesis.Spectroscopy_0"]
    import archnemesis_dist_amd.forward_model as fmod
    src = os.path.join(REF, "tests", "files", "Jupiter_CIRS_angled_thermal_emission_scattering")
    work = tempfile.mkdtemp(prefix="ansfm_dropin_ms_")
    cwd = os.getcwd()
    try:
        for f in os.listdir(src):
            shutil.copy(os.path.join(src, f), os.path.join(work, f))
            os.chmod(os.path.join(work, f), 0o644)
        rng = np.random.default_rng(4)
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
        os.chdir(work)
        double = OracleEngineDouble(oracle)
        monkeypatch.setattr(fmod, "get_engine", lambda device=0: double)
        fmod.set_strict(True)                        # a delegation to the reference's CIRSrad would raise
        FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
        Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
        Meas.NCONV = np.array([NKEEP], dtype="int32")
        Meas.VCONV = Meas.VCONV[:NKEEP]; Meas.MEAS = Meas.MEAS[:NKEEP]; Meas.ERRMEAS = Meas.ERRMEAS[:NKEEP]
        Meas.NY = NKEEP
        fm = FMGPU(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec, Stellar=Stel,
                   Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
        SPECONV = fm.nemesisfm()
    finally:
        fmod.set_strict(False)
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    assert double.scatter_calls == 1
    z = np.load(os.path.join(golden_dir, "c4_cirsrad_scatter.npz"))
    np.testing.assert_allclose(SPECONV, z["SPECONV"], rtol=1e-8)
    np.testing.assert_allclose(fm.LayerX.TAUGAS, z["TAUGAS"], rtol=2e-7)
    np.testing.assert_allclose(fm.LayerX.TAUTOT, z["TAUTOT"], rtol=2e-7)


def test_single_scattering_nemesisfm_through_the_adapter_matches_the_reference(oracle, monkeypatch):
    """ISCAT = SINGLE_SCATTERING_PLANE_PARALLEL on the same inputs: the unmodified reference's nemesisfm against the
    adapter's (CIRSrad's single-scattering branch -> ansfm_cirsrad_ck_singlescatt, here the double), in one process."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.ref_import import import_reference
    from oracle.gen_golden_c1 import GASES
    ans = import_reference()
    sp_mod = sys.modules["archnemesis.Spectroscopy_0"]
    import archnemesis_dist_amd.forward_model as fmod
    src = os.path.join(REF, "tests", "files", "Jupiter_CIRS_angled_thermal_emission_scattering")
    work = tempfile.mkdtemp(prefix="ansfm_dropin_ss_")
    cwd = os.getcwd()
    NKEEP = 12
    try:
        for f in os.listdir(src):
            shutil.copy(os.path.join(src, f), os.path.join(work, f))
            os.chmod(os.path.join(work, f), 0o644)
        rng = np.random.default_rng(6)
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
        os.chdir(work)

        def run(cls):
            Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
            Scat.ISCAT = 3                                   # ScatteringCalculationModeEnum.SINGLE_SCATTERING_PLANE_PARALLEL
            Meas.NCONV = np.array([NKEEP], dtype="int32")
            Meas.VCONV = Meas.VCONV[:NKEEP]; Meas.MEAS = Meas.MEAS[:NKEEP]; Meas.ERRMEAS = Meas.ERRMEAS[:NKEEP]
            Meas.NY = NKEEP
            fm = cls(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec, Stellar=Stel,
                     Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
            return fm, fm.nemesisfm()

        _, ref = run(ans.ForwardModel_0)
        double = OracleEngineDouble(oracle)
        monkeypatch.setattr(fmod, "get_engine", lambda device=0: double)
        fmod.set_strict(True)
        fm, got = run(fmod.make_gpu_forward_model(ans.ForwardModel_0))
    finally:
        fmod.set_strict(False)
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    assert double.ss_calls == 1 and int(np.asarray(fm.PathX.IMOD)[0]) & 1024
    assert np.all(ref > 0)
    np.testing.assert_allclose(got, ref, rtol=1e-9)


def test_nemesisfm_through_the_adapter_matches_the_reference(c1_run, oracle, golden_dir, monkeypatch):
    ans = c1_run
    import archnemesis_dist_amd.forward_model as fmod
    double = OracleEngineDouble(oracle)
    monkeypatch.setattr(fmod, "get_engine", lambda device=0: double)
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
    fm = FMGPU(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec, Stellar=Stel,
               Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
    SPECONV = fm.nemesisfm()
    z = np.load(os.path.join(golden_dir, "c1_cirsrad.npz"))
    np.testing.assert_allclose(SPECONV, z["SPECONV"], rtol=1e-10)      # the reference's own nemesisfm() result
    sel = z["sel"]
    np.testing.assert_allclose(fm.LayerX.TAUGAS[sel], z["TAUGAS"], rtol=1e-11)
    np.testing.assert_allclose(fm.LayerX.TAUTOT[sel], z["TAUTOT"], rtol=1e-11)


def test_nemesisfmg_through_the_adapter_matches_the_reference(c1_run, oracle, golden_dir, monkeypatch):
    """Analytic-gradient forward model: CIRSrad(return_grad=True) through the adapter, then the reference's own
    map2pro / map2xvec / convg (ForwardModel_0.py:705-771) -> dSPECONV (NCONV, NGEOM, NX)."""
    ans = c1_run
    import archnemesis_dist_amd.forward_model as fmod
    double = OracleEngineDouble(oracle)
    monkeypatch.setattr(fmod, "get_engine", lambda device=0: double)
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
    fm = FMGPU(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec, Stellar=Stel,
               Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
    fm.ansfm_select_gradient_gases = True        # opt-in: only the gases subprofretg's xmap touches (here: none, T only)
    SPECONV, dSPECONV = fm.nemesisfmg()
    assert double.gas_selections == [([], True), None]   # no gas, temperature: selected for the CIRSrad call, reset afterwards
    z = np.load(os.path.join(golden_dir, "c1_cirsrad_grad.npz"))
    np.testing.assert_allclose(SPECONV, z["SPECONVg"], rtol=1e-10)
    ref = z["dSPECONV"]
    assert dSPECONV.shape == ref.shape
    scale = np.abs(ref).max(axis=(0, 1), keepdims=True) + 1e-300
    # (trold - tr) with tr = trold*exp(-tau) cancels for the thin top layers (tau ~ 1e-9), so a 1-ulp
    # difference between libm's and NumPy's exp shows up as ~1e-7 relative in those (tiny) gradient entries:
    # measured 9e-7 of the column maximum; the Jacobian contract is 1e-4
    assert np.max(np.abs(dSPECONV - ref) / scale) < 1e-5


def test_transmission_branch_of_cirsrad_through_the_adapter_matches_the_reference(c1_run, oracle, monkeypatch):
    """CIRSrad's first dispatch branch (:4478-4483, calculate_transmission_spectrum :4110-4131), without and with
    return_grad: the unmodified reference against the adapter on the same prepared forward model (the Jupiter nadir
    run with its path flags cleared to "pure transmission")."""
    ans = c1_run
    import archnemesis_dist_amd.forward_model as fmod
    NKEEP = 40

    def prepared(cls):
        Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
        Meas.NCONV = np.array([NKEEP], dtype="int32")
        Meas.VCONV = Meas.VCONV[:NKEEP]; Meas.MEAS = Meas.MEAS[:NKEEP]; Meas.ERRMEAS = Meas.ERRMEAS[:NKEEP]
        Meas.NY = NKEEP
        fm = cls(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec, Stellar=Stel,
                 Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
        fm.nemesisfm()                                     # host preparation: profiles, layers, path, tables
        fm.PathX.IMOD = np.zeros_like(np.asarray(fm.PathX.IMOD))
        return fm

    fref = prepared(ans.ForwardModel_0)
    ref = fref.CIRSrad(return_grad=False)
    refg = fref.CIRSrad(return_grad=True)
    double = OracleEngineDouble(oracle)
    monkeypatch.setattr(fmod, "get_engine", lambda device=0: double)
    fmod.set_strict(True)
    try:
        fm = prepared(fmod.make_gpu_forward_model(ans.ForwardModel_0))
        got = fm.CIRSrad(return_grad=False)
        gotg = fm.CIRSrad(return_grad=True)
    finally:
        fmod.set_strict(False)
    assert double.tr_calls == 1 and double.trg_calls == 1
    assert np.all(ref >= 0) and np.all(ref <= 1.0) and 1e-3 < ref.max() < 1.0      # a real transmission spectrum
    np.testing.assert_allclose(got, ref, rtol=1e-10)
    np.testing.assert_allclose(gotg[0], refg[0], rtol=1e-10)
    assert gotg[1].shape == refg[1].shape and gotg[2].shape == refg[2].shape
    scale = np.abs(refg[1]).max(axis=(0, 2, 3), keepdims=True) + 1e-300
    assert np.abs(refg[1]).max() > 0
    assert np.max(np.abs(gotg[1] - refg[1]) / scale) < 1e-9
    assert not np.any(gotg[2]) and not np.any(refg[2])


def test_nemesisfmg_with_gradient_maps_routed_through_the_engine(c1_run, oracle, golden_dir, monkeypatch):
    """install_gpu_gradient_maps: nemesisfmg's map2pro / map2xvec calls (:705-711) land on the engine's entry points
    with the reference's arguments -- likewise Layer_0.layer_averageg via install_gpu_layering (calc_pathg) -- and the
    result is still the reference's dSPECONV."""
    ans = c1_run
    import importlib
    import archnemesis_dist_amd.forward_model as fmod
    fm0 = importlib.import_module("archnemesis.ForwardModel_0")
    double = OracleEngineDouble(oracle)
    monkeypatch.setattr(fmod, "get_engine", lambda device=0: double)
    l0 = importlib.import_module("archnemesis.Layer_0")
    orig = (fm0.map2pro, fm0.map2xvec)
    orig_l = (l0.layer_average, l0.layer_averageg)
    try:
        fmod.install_gpu_gradient_maps()
        fmod.install_gpu_layering()
        fmod.install_gpu_continuum()
        FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
        Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
        fm = FMGPU(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec, Stellar=Stel,
                   Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
        SPECONV, dSPECONV = fm.nemesisfmg()
    finally:
        fm0.map2pro, fm0.map2xvec = orig
        l0.layer_average, l0.layer_averageg = orig_l
        for name in ("calc_tau_cia", "calc_tau_rayleigh", "calc_tau_dust"):
            if hasattr(fm0.ForwardModel_0, "_ansfm_reference_" + name):
                setattr(fm0.ForwardModel_0, name, getattr(fm0.ForwardModel_0, "_ansfm_reference_" + name))
                delattr(fm0.ForwardModel_0, "_ansfm_reference_" + name)
        for mod, name in ((fm0, "_ansfm_reference_maps"), (l0, "_ansfm_reference_layering")):
            if hasattr(mod, name):
                delattr(mod, name)
    assert double.map_calls >= 2 and double.lay_calls >= 1 and double.cia_calls >= 1
    assert double.ray_calls >= 1 and double.dust_calls >= 1
    z = np.load(os.path.join(golden_dir, "c1_cirsrad_grad.npz"))
    ref = z["dSPECONV"]
    scale = np.abs(ref).max(axis=(0, 1), keepdims=True) + 1e-300
    assert np.max(np.abs(dSPECONV - ref) / scale) < 1e-5


def test_nemesisfm_with_tables_streamed_from_the_kta_files(c1_run, oracle, golden_dir, monkeypatch):
    """install_gpu_table_reader: Spectroscopy.read_tables keeps the reference's header logic but leaves the k data in
    the files (K is a KtaTableOnDevice); the adapter hands the files to the engine once although nemesisfm re-reads the
    tables on every call, and the spectrum is still the reference's."""
    ans = c1_run
    import importlib
    import archnemesis_dist_amd.forward_model as fmod
    sp = importlib.import_module("archnemesis.Spectroscopy_0")
    double = OracleEngineDouble(oracle)
    monkeypatch.setattr(fmod, "get_engine", lambda device=0: double)
    orig = sp.Spectroscopy_0.read_tables
    try:
        fmod.install_gpu_table_reader()
        FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
        Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
        fm = FMGPU(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec, Stellar=Stel,
                   Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
        SPECONV = fm.nemesisfm()
        assert isinstance(fm.SpectroscopyX.K, fmod.KtaTableOnDevice)
        SPECONV2 = fm.nemesisfm()
    finally:
        sp.Spectroscopy_0.read_tables = orig
        if hasattr(sp.Spectroscopy_0, "_ansfm_reference_read_tables"):
            del sp.Spectroscopy_0._ansfm_reference_read_tables
    z = np.load(os.path.join(golden_dir, "c1_cirsrad.npz"))
    np.testing.assert_allclose(SPECONV, z["SPECONV"], rtol=1e-10)
    assert np.array_equal(SPECONV, SPECONV2) and double.file_uploads == 1
    # a CPU path that needs the numbers still gets them
    assert np.asarray(fm.SpectroscopyX.K).shape == fm.SpectroscopyX.K.shape


def test_measurement_lblconv_methods_routed_through_the_engine(c1_run, oracle, monkeypatch):
    """install_gpu_convolution: Measurement_0.lblconv / lblconvg (:2125, :2191) reach the module-level kernels by global
    name; with IGEOM = int they call lblconv / lblconvg, with IGEOM = 'All' the *_ngeom variants -- all land on the
    engine and return what the reference's own kernels return on the same inputs."""
    ans = c1_run
    import importlib
    import archnemesis_dist_amd.forward_model as fmod
    m0 = importlib.import_module("archnemesis.Measurement_0")
    ref = {n: getattr(m0, n) for n in ("lblconv", "lblconvg", "lblconv_ngeom", "lblconvg_ngeom")}
    double = OracleEngineDouble(oracle)
    monkeypatch.setattr(fmod, "get_engine", lambda device=0: double)
    for n, f in ref.items():                       # undo the module patch after the test
        monkeypatch.setattr(m0, n, f)
    for n in ("lblconv_fil", "lblconvg_fil", "lblconv_fil_ngeom", "lblconvg_fil_ngeom", "integrate_filter",
              "integrate_filter_ngeom", "integrate_filterg", "integrate_filterg_ngeom"):
        monkeypatch.setattr(m0, n, getattr(m0, n))
    monkeypatch.delattr(m0, "_ansfm_reference_intf", raising=False)
    monkeypatch.delattr(m0, "_ansfm_reference_conv", raising=False)
    rconv, rconvg = m0.Measurement_0.conv, m0.Measurement_0.convg
    monkeypatch.setattr(m0.Measurement_0, "conv", rconv)
    monkeypatch.setattr(m0.Measurement_0, "convg", rconvg)
    monkeypatch.delattr(m0.Measurement_0, "_ansfm_reference_conv_methods", raising=False)
    fmod.install_gpu_convolution()
    rng = np.random.default_rng(4)
    nwave, ngeom, nx, nconv = 400, 2, 3, 9
    wave = 100.0 + 0.02 * np.arange(nwave)
    y = rng.uniform(1, 2, (nwave, ngeom)); dy = rng.normal(size=(nwave, ngeom, nx))
    Meas = ans.Measurement_0()
    Meas.NGEOM = ngeom; Meas.FWHM = 0.3; Meas.ISHAPE = 2; Meas.V_DOPPLER = 0.0
    Meas.NCONV = np.array([nconv] * ngeom)
    Meas.VCONV = np.tile(np.linspace(101.0, 107.0, nconv)[:, None], (1, ngeom))
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        a = Meas.lblconv(wave, y[:, 1], IGEOM=1)
        b = Meas.lblconv(wave, y, IGEOM="All")
        c, dc = Meas.lblconvg(wave, y[:, 0], dy[:, 0, :], IGEOM=0)
        d, dd = Meas.lblconvg(wave, y, dy, IGEOM="All")
    assert double.conv_calls == ["lblconv", "lblconv_ngeom", "lblconvg", "lblconvg_ngeom"]
    vc = Meas.VCONV[:, 0]
    np.testing.assert_allclose(a, ref["lblconv"](nwave, wave, y[:, 1], nconv, vc, 2, 0.3), rtol=1e-13)
    np.testing.assert_allclose(b, ref["lblconv_ngeom"](nwave, wave, y, nconv, vc, 2, 0.3), rtol=1e-13)
    rc, rdc = ref["lblconvg"](nwave, wave, y[:, 0], dy[:, 0, :], nconv, vc, 2, 0.3)
    np.testing.assert_allclose(c, rc, rtol=1e-13); np.testing.assert_allclose(dc, rdc, rtol=1e-12, atol=1e-15)
    rd, rdd = ref["lblconvg_ngeom"](nwave, wave, y, dy, nconv, vc, 2, 0.3)
    np.testing.assert_allclose(d, rd, rtol=1e-13); np.testing.assert_allclose(dd, rdd, rtol=1e-12, atol=1e-15)
    # k-table methods: the filter-function branch (FWHM < 0) goes to the engine, FWHM == 0 stays the reference's interp1d
    Meas.FWHM = -1.0
    nf = 7
    Meas.NFIL = np.full(nconv, nf, dtype=np.int32)
    Meas.VFIL = vc[None, :] + np.linspace(-0.2, 0.2, nf)[:, None]
    Meas.AFIL = np.tile(np.array([0.2, 0.5, 0.9, 1.0, 0.9, 0.5, 0.0])[:, None], (1, nconv))

    class FakeEngine(OracleEngineDouble):
        def conv_fil(self, vw, yy, dd_, nc, vcv, nfil, vfil, afil):
            self._conv("conv_fil")
            return self.orc.lblconv_fil(len(vw), vw, yy, nc, vcv, nfil, vfil, afil, dydx=dd_, bracket=True)
    double.__class__ = FakeEngine
    e = Meas.conv(wave, y[:, 0], IGEOM=0)
    f, df = Meas.convg(wave, y[:, 0], dy[:, 0, :], IGEOM=0)
    assert double.conv_calls[-2:] == ["conv_fil", "conv_fil"]
    np.testing.assert_allclose(e, rconv(Meas, wave, y[:, 0], IGEOM=0), rtol=1e-13)
    rf, rdf = rconvg(Meas, wave, y[:, 0], dy[:, 0, :], IGEOM=0)
    np.testing.assert_allclose(f, rf, rtol=1e-13); np.testing.assert_allclose(df, rdf, rtol=1e-12, atol=1e-15)
    Meas.FWHM = 0.0
    ncalls = len(double.conv_calls)
    np.testing.assert_array_equal(Meas.conv(wave, y[:, 0], IGEOM=0), rconv(Meas, wave, y[:, 0], IGEOM=0))
    assert len(double.conv_calls) == ncalls


def test_read_tables_keeps_lbl_tables_in_their_files(c1_run, golden_dir, monkeypatch):
    """install_gpu_table_reader with ILBL = 2: Spectroscopy.read_tables on binary .lta tables leaves K a file
    description (the engine streams the files); the numbers it stands for are the reference's read_tables result."""
    ans = c1_run
    import importlib
    import archnemesis_dist_amd.forward_model as fmod
    sp = importlib.import_module("archnemesis.Spectroscopy_0")
    monkeypatch.setattr(sp.Spectroscopy_0, "read_tables", sp.Spectroscopy_0.read_tables)
    monkeypatch.delattr(sp.Spectroscopy_0, "_ansfm_reference_read_tables", raising=False)
    paths = [os.path.join(golden_dir, "kta", f"lbl_gas{i}.lta") for i in range(2)]

    def make():
        S = ans.Spectroscopy_0(ILBL=2)
        S.NGAS = 2; S.ID = np.array([2, 5]); S.ISO = np.array([1, 0]); S.LOCATION = list(paths)
        S.read_header()
        return S
    Sref = make(); Sref.read_tables(wavemin=2000.9, wavemax=2003.6)
    fmod.install_gpu_table_reader()
    S = make(); S.read_tables(wavemin=2000.9, wavemax=2003.6)
    assert isinstance(S.K, fmod.KtaTableOnDevice) and S.K.ext == ".lta"
    assert S.K.shape == Sref.K.shape and np.array_equal(S.WAVE, Sref.WAVE)
    assert np.array_equal(np.asarray(S.K), Sref.K)


def test_reference_conv_with_positive_fwhm_cannot_run(c1_run):
    """Why the cubic-spline branch of Measurement_0.conv (FWHM > 0, Measurement_0.py:2347-2411) has no GPU counterpart: the
    reference's own branch raises before it computes anything -- `VCONV[NCONV[IGEOM], IGEOM]` (:2357) is one past the last row
    whenever the geometry fills the array (IndexError), and with fewer points `self.NWAVE` (:2372) does not exist
    (AttributeError); where it could run, its 'trapezoid' sums differences, `(yi[j] - yold) * delx / 2` (:2405), i.e. it
    telescopes to (y(x2) - y(x1)) delx / 2.  `install_gpu_convolution` hands the case to the reference, which raises."""
    ans = c1_run
    Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
    Wave = np.linspace(0.0, 1500.0, 601)
    y = 1e-7 * (1.0 + np.sin(Wave / 50.0))
    Meas.FWHM = 2.0
    with pytest.raises(IndexError):
        Meas.conv(Wave, y, IGEOM=0)
    Meas.NCONV = np.array([int(Meas.NCONV[0]) - 5], dtype="int32")
    with pytest.raises(AttributeError):
        Meas.conv(Wave, y, IGEOM=0)
