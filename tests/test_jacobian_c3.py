"""BASELINE configs[2]: the numerical Jacobian of jacobian_nemesis (ForwardModel_0.py:2184-2361) with every
forward model of a rank in one batched engine call -- state vector (T and ln VMR at the profile levels, model 0) ->
layer_average -> Rayleigh continuum -> CIRSrad -> KK.

  CPU : the batching / sharding / quotient logic with a stand-in model (serial and gloo world 2);
  GPU : KK from the engine against KK formed the reference's way from the oracle's forward models (north_star: <= 1e-4
        of each column; held to 1e-7 here), and the de-duplication bookkeeping."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _ToyModel:
    """spectra_batch on the CPU: a smooth non-linear function of the profiles (no engine)."""
    def __init__(self, log):
        from archnemesis_dist_amd import synthetic as syn
        from archnemesis_dist_amd.profile_state import ContinuousProfileState
        pr = syn.synth_profiles(6, 4)
        self.state = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 2)])
        self.state.FIX[3] = 1
        rng = np.random.default_rng(3)
        self.A = rng.normal(size=(9, 6)); self.B = rng.normal(size=(9, 6))
        self.log = log

    def f(self, X):
        T, VMR = self.state.profiles(X)
        return np.tanh(T / 300.0) @ self.A.T + (1.0e3 * VMR[:, :, 2]) @ self.B.T

    def spectra_batch(self, X):
        import torch
        self.log.append(X.shape[0])
        return torch.as_tensor(self.f(X))


def _direct_kk(model):
    from archnemesis_dist_amd.jacobian import perturbed_states
    st = model.state
    XN = st.XN.copy()
    xnx = perturbed_states(XN, 0.05 * XN)
    Y = model.f(xnx.T)
    KK = np.zeros((Y.shape[1], st.NX))
    for i in range(st.NX):
        if st.FIX[i] == 0:
            KK[:, i] = (Y[i + 1] - Y[0]) / (1.05 * XN[i] - XN[i])
    return Y[0], KK


def test_batched_jacobian_serial():
    from archnemesis_dist_amd.jacobian import jacobian_nemesis_batched, jacobian_nemesis_sharded
    log = []
    m = _ToyModel(log)
    YN, KK = jacobian_nemesis_batched(m)
    assert log == [m.state.NX]                      # 12 elements, one FIXed: 1 + 11 forward models in ONE call
    y0, kk = _direct_kk(m)
    np.testing.assert_allclose(YN, y0, rtol=1e-14)
    np.testing.assert_allclose(KK, kk, rtol=1e-12, atol=1e-300)
    assert np.all(KK[:, 3] == 0.0)
    YN2, KK2 = jacobian_nemesis_sharded(m)          # the sharded entry takes the batched route for such a model
    assert np.array_equal(KK2, KK) and len(log) == 2


def test_batched_jacobian_gloo_world2(tmp_path):
    script = textwrap.dedent(f'''
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        import numpy as np, torch.distributed as dist
        from archnemesis_dist_amd.jacobian import jacobian_nemesis_batched, chunk_range
        from test_jacobian_c3 import _ToyModel, _direct_kk
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        log = []
        m = _ToyModel(log)
        YN, KK = jacobian_nemesis_batched(m, rank=rank, world_size=world)
        s, e = chunk_range(12, world, rank)
        assert log == [e - s + (1 if rank else 0)], (rank, log)    # rank 1 leads its batch with the unperturbed state
        y0, kk = _direct_kk(m)
        np.testing.assert_allclose(YN, y0, rtol=1e-14)
        np.testing.assert_allclose(KK, kk, rtol=1e-12, atol=1e-300)
        dist.destroy_process_group()
        print("rank", rank, "ok")
    ''')
    f = tmp_path / "jb.py"
    f.write_text(script)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29621", str(f)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


class _ToyModelSlice(_ToyModel):
    """the toy model on a part of its 9 'wavenumbers' (rows of A / B): what a rank holds in the wavenumber-sharded mode"""
    def __init__(self, log, rank, world):
        from archnemesis_dist_amd.jacobian import chunk_range
        super().__init__(log)
        self.global_waves = 9
        s, e = chunk_range(9, world, rank)
        self.A, self.B = self.A[s:e], self.B[s:e]
        self.world = world

    def ny_local_all(self, world):
        from archnemesis_dist_amd.jacobian import chunk_range
        return [e - s for s, e in (chunk_range(9, world, r) for r in range(world))]


def test_wavenumber_sharded_jacobian_gloo_world2(tmp_path):
    """shard = "wavenumbers": both ranks run all 12 forward models on their part of the spectral axis (5 + 4 of 9 points,
    ragged), one all_gather puts the parts side by side: KK bit-identical to the one-rank result, and a rank count larger
    than the number of forward models no longer deadlocks the state mode (the empty chunk joins the gather)."""
    script = textwrap.dedent(f'''
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        import numpy as np, torch.distributed as dist
        from archnemesis_dist_amd.jacobian import jacobian_nemesis_batched
        from test_jacobian_c3 import _ToyModel, _ToyModelSlice
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        log = []
        YN1, KK1 = jacobian_nemesis_batched(_ToyModel([]))                       # the whole axis on one rank
        m = _ToyModelSlice(log, rank, world)
        YN, KK = jacobian_nemesis_batched(m, rank=rank, world_size=world, shard="wavenumbers")
        assert log == [12], log                                                  # every state on every rank, none twice
        assert YN.shape == (9,) and KK.shape == (9, 12)
        assert np.array_equal(YN, YN1) and np.array_equal(KK, KK1)
        print("rank", rank, "ok")
        dist.destroy_process_group()
    ''')
    f = tmp_path / "jw.py"
    f.write_text(script)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29623", str(f)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_state_sharded_jacobian_with_more_ranks_than_forward_models():
    """chunk arithmetic of a rank that gets no forward model: an empty (0, NY) block, no call into the engine"""
    from archnemesis_dist_amd.jacobian import chunk_range
    assert chunk_range(3, 4, 3) == (3, 3) and chunk_range(3, 4, 2) == (2, 3)


def test_profile_state_model0_semantics():
    from archnemesis_dist_amd import synthetic as syn
    from archnemesis_dist_amd.profile_state import ContinuousProfileState
    pr = syn.synth_profiles(5, 4)
    st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], [("VMR", 3), "T"])
    assert st.NX == 10
    np.testing.assert_allclose(st.XN[:5], np.log(pr["VMR"][:, 3]))          # ln for mixing ratios, T as it is
    np.testing.assert_allclose(st.XN[5:], pr["T"])
    X = np.stack([st.XN, st.XN])
    X[1, 2] += 0.1; X[1, 7] *= 1.05
    T, VMR = st.profiles(X)
    np.testing.assert_allclose(T[0], pr["T"]); np.testing.assert_allclose(VMR[0], pr["VMR"], rtol=1e-15)
    assert T[1, 2] == pr["T"][2] * 1.05 and np.isclose(VMR[1, 2, 3], pr["VMR"][2, 3] * np.exp(0.1))
    assert np.array_equal(VMR[1, :, :3], pr["VMR"][:, :3])


@pytest.mark.gpu
@pytest.mark.parametrize("pointing,f32", [("nadir", True), ("limb", False)])
def test_c3_numerical_jacobian_vs_oracle(oracle, pointing, f32):
    """NX = 40 (T and ln VMR of one absorber at 20 levels, through Curtis-Godson layer_average), 16 layers, Rayleigh
    continuum: KK from ONE batched engine call vs KK from 41 oracle forward models."""
    import archnemesis_dist_amd as pkg
    from archnemesis_dist_amd import synthetic as syn, layering
    from archnemesis_dist_amd.jacobian import jacobian_nemesis_batched
    from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel
    from oracle import jacobian_twin as twin
    W, G, NP, NT, S, NPRO, NLAY = 192, 10, 8, 6, 4, 20, 16
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=77)
    _, delg = syn.gauss_legendre_01(G, as_float32=f32)
    if f32:
        PRESS, TEMP, delg = PRESS.astype(np.float32), TEMP.astype(np.float32), delg.astype(np.float32)
    WAVE = 300.0 + 0.5 * np.arange(W)
    pr = syn.synth_profiles(NPRO, 6, seed=5, p_bottom_bar=5.0, p_top_bar=1.0e-5)
    st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 2)])
    geo = dict(pointing=layering.NADIR, EMISS_ANG=25.0, ANGLE=25.0) if pointing == "nadir" else \
        dict(pointing=layering.LIMB, BOTLAY=3)
    eng = pkg.AnsfmEngine(0)
    try:
        eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
        model = BatchedCKThermalModel(eng, st, pr["RADIUS"], pr["ID"], pr["ISO"], [2, 3, 4, 5],
                                      layering_args=dict(NLAY=NLAY, LAYINT=1, NINT=101), geometry=geo, IRAY=4)
        YN, KK = jacobian_nemesis_batched(model)
        rows, total = model.last_rows
        # the layers left in HBM (layer_average_dev; default) and the layers through host arrays: the same bits
        model.device_layers = False
        YN_h, KK_h = jacobian_nemesis_batched(model)
        assert np.array_equal(YN, YN_h) and np.array_equal(KK, KK_h) and model.last_rows == (rows, total)
        model.device_layers = True
        model.fused_rayleigh = False                                    # the continuum as an (n, NWAVE, NLAY) array of its own
        YN_c, KK_c = jacobian_nemesis_batched(model)
        assert np.array_equal(YN, YN_c) and np.array_equal(KK, KK_c) and model.last_rows == (rows, total)
        model.fused_rayleigh = True
        import torch
        side = torch.cuda.Stream(device=model.torch_device())          # torch on a stream of its own, the engine on another
        with torch.cuda.stream(side):
            YN_s, KK_s = jacobian_nemesis_batched(model)
        assert np.array_equal(YN, YN_s) and np.array_equal(KK, KK_s)
    finally:
        eng.close()
    assert total == (st.NX + 1) * NLAY and NLAY < rows < total // 3        # a level touches a few layers, not all
    y0, kk = twin.jacobian(model, K, PRESS, TEMP, WAVE, delg)
    np.testing.assert_allclose(YN, y0, rtol=1e-10)
    # error relative to each column's maximum; a column without sensitivity (ln VMR at the top levels: |KK| ~ 1e-20, the
    # rounding of the two spectra it is the difference of) is held to 1e-6 of the largest column instead
    scale = np.max(np.abs(kk), axis=0)
    scale = np.maximum(scale, 1.0e-6 * scale.max())
    assert np.max(np.abs(KK - kk) / scale) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("pointing", ["nadir", "limb"])
def test_c3_analytic_jacobian_vs_small_step_differences(pointing):
    """The nemesisfmg route at the state-vector level (layer_averageg -> CIRSrad(return_grad) -> map2pro -> map2xvec)
    against central differences with a small step.  ln(VMR) columns: differences through the full batched forward route
    (layer_average included).  Temperature columns: like the reference's, the analytic gradient is first order in DTE at
    fixed layer amounts, so the differences move the layer temperatures along the DTE column and keep the amounts."""
    import archnemesis_dist_amd as pkg
    from archnemesis_dist_amd import synthetic as syn, layering
    from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel
    W, G, NP, NT, S, NPRO, NLAY = 192, 10, 8, 6, 4, 20, 16
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=78)
    _, delg = syn.gauss_legendre_01(G)
    WAVE = 300.0 + 0.5 * np.arange(W)
    pr = syn.synth_profiles(NPRO, 6, seed=6, p_bottom_bar=5.0, p_top_bar=1.0e-5)
    st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 3)])
    geo = dict(pointing=layering.NADIR, EMISS_ANG=25.0, ANGLE=25.0) if pointing == "nadir" else \
        dict(pointing=layering.LIMB, BOTLAY=3)
    eng = pkg.AnsfmEngine(0)
    try:
        eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
        model = BatchedCKThermalModel(eng, st, pr["RADIUS"], pr["ID"], pr["ISO"], [2, 3, 4, 5],
                                      layering_args=dict(NLAY=NLAY, LAYINT=1, NINT=101), geometry=geo, IRAY=0)
        from archnemesis_dist_amd.jacobian import jacobian_nemesis_sharded
        YN, KK = jacobian_nemesis_sharded(model, analytical_gradient=True)       # -> model.jacobian_analytic()
        y0 = model.spectra_batch(st.XN[None]).cpu().numpy()[0]
        np.testing.assert_allclose(YN, y0, rtol=1e-12)
        assert KK.shape == (y0.size, st.NX) and np.all(np.isfinite(KK))
        # ln VMR columns through the whole forward route
        lv = (2, 8, 13) if pointing == "nadir" else (8, 11, 14)      # the limb path starts at layer 3
        cols = [NPRO + i for i in lv]
        h = 1.0e-4
        X = np.repeat(st.XN[None], 2 * len(cols), 0)
        for k, c in enumerate(cols):
            X[2 * k, c] += h; X[2 * k + 1, c] -= h
        Y = model.spectra_batch(X).cpu().numpy()
        for k, c in enumerate(cols):
            fd = (Y[2 * k] - Y[2 * k + 1]) / (2 * h)
            assert np.max(np.abs(fd)) > 0 and np.max(np.abs(KK[:, c] - fd)) < 1e-5 * np.max(np.abs(fd)), c     # differences of step 1e-4 / 1e-3: contract 1e-4
        # temperature columns: the layer temperatures moved along the column of DTE (what the analytic map assumes), the
        # amounts kept; levels whose layers include the bottom one are left out (the reference's gradient does not carry
        # the ground term B(T_bottom) of a TSURF <= 0 atmosphere)
        base = model.layers(st.XN[None]); path = base["path"]
        DTE = model.last_analytic_layers["DTE"]
        for c in ((6, 10, 14) if pointing == "nadir" else (8, 11, 14)):
            assert DTE[0, c] == 0.0 and np.any(DTE[:, c] != 0.0)
            hT = 1.0e-3
            ys = []
            for sg in (1.0, -1.0):
                T = base["TEMP"][0] + sg * hT * DTE[:, c]
                em = path.EMTEMP[0] + sg * hT * np.where(np.arange(path.LAYINC.shape[0])[:, None] < path.NLAYIN[None, :],
                                                         DTE[path.LAYINC, c], 0.0)
                ys.append(eng.cirsrad_ck_thermal(0, base["PRESS"][0], T, base["amount"][0], None, path.NLAYIN, path.LAYINC,
                                                 path.SCALE, em, -1.0).reshape(-1))
            fd = (ys[0] - ys[1]) / (2 * hT)
            assert np.max(np.abs(fd)) > 0 and np.max(np.abs(KK[:, c] - fd)) < 1e-5 * np.max(np.abs(fd)), c     # differences of step 1e-4 / 1e-3: contract 1e-4
    finally:
        eng.close()


@pytest.mark.gpu
def test_c3_analytic_jacobian_vs_oracle_chain(oracle):
    """The same route with the Rayleigh continuum (IRAY = 4) against the oracle's restatement of every step:
    layer_averageg, calc_tau_rayleigh, CIRSrad(return_grad) with dTAURAY added to every gas parameter (:3952-3957),
    map2pro, map2xvec.  Two gases and temperature in the state vector: the gradient merge runs with those two selected."""
    import archnemesis_dist_amd as pkg
    from archnemesis_dist_amd import synthetic as syn, layering
    from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel
    W, G, NP, NT, S, NPRO, NLAY = 160, 10, 8, 6, 4, 18, 14
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=79)
    _, delg = syn.gauss_legendre_01(G)
    WAVE = 400.0 + 0.5 * np.arange(W)
    pr = syn.synth_profiles(NPRO, 6, seed=7, p_bottom_bar=5.0, p_top_bar=1.0e-5)
    st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], [("VMR", 5), "T", ("VMR", 2)])
    igas = [2, 3, 4, 5]
    eng = pkg.AnsfmEngine(0)
    try:
        eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
        model = BatchedCKThermalModel(eng, st, pr["RADIUS"], pr["ID"], pr["ISO"], igas, layering_args=dict(NLAY=NLAY, LAYINT=1, NINT=101),
                                      geometry=dict(pointing=layering.NADIR, EMISS_ANG=15.0, ANGLE=15.0), IRAY=4)
        YN, KK = model.jacobian_analytic()
        BASEH, BASEP = model.BASEH, model.BASEP
    finally:
        eng.close()
    names = ("HEIGHT", "PRESS", "TEMP", "TOTAM", "AMOUNT", "PP", "CONT", "FRAC", "DELH", "BASET", "LAYSF", "DTE", "DAM", "DCO", "DPH")
    lay = dict(zip(names, oracle.layer_averageg(pr["RADIUS"], st.H, st.P, st.T, pr["ID"], st.VMR, None, None, BASEH, BASEP, LAYINT=1,
                                                NINT=101)))
    path = layering.calc_path(pr["RADIUS"], BASEH, lay["DELH"], lay["TEMP"], float(st.H[-1]), pointing=layering.NADIR,
                              EMISS_ANG=15.0, ANGLE=15.0)
    NVMR = st.VMR.shape[1]; NPAR = NVMR + 2
    tauray, dtauray = oracle.calc_tau_rayleigh(4, 0, WAVE, lay["TOTAM"], pr["ID"], pr["ISO"], lay["PP"] / lay["PRESS"][:, None])
    dcont = np.zeros((W, NPAR, NLAY)); dcont[:, :NVMR, :] = dtauray[:, None, :]
    amount = np.ascontiguousarray(lay["AMOUNT"][:, igas].T) * 1.0e-4
    spec, dspec, _ = oracle.cirsradg_ck_thermal(0, K, PRESS, TEMP, WAVE, delg, lay["PRESS"], lay["TEMP"], amount, tauray, dcont, NVMR,
                                                NPAR, np.array(igas, dtype=np.int32), path.NLAYIN, path.LAYINC, path.SCALE, path.EMTEMP,
                                                -1.0)
    pro = oracle.map2pro(dspec, W, NVMR, 0, NPRO, 1, path.NLAYIN, path.LAYINC, lay["DTE"], lay["DAM"], lay["DCO"])
    xmap = np.zeros((st.NX, NPAR, NPRO)); lev = np.arange(NPRO)
    xmap[lev, 5, lev] = st.VMR[:, 5]; xmap[NPRO + lev, NVMR, lev] = 1.0; xmap[2 * NPRO + lev, 2, lev] = st.VMR[:, 2]
    kk = oracle.map2xvec(pro, W, NVMR, 0, NPRO, 1, st.NX, xmap).reshape(W, st.NX)
    np.testing.assert_allclose(YN, spec.reshape(-1), rtol=1e-10)
    scale = np.max(np.abs(kk), axis=0); scale = np.maximum(scale, 1e-9 * scale.max())
    # thin top layers: (trold - tr) with tr = trold * exp(-tau) cancels, a 1-ulp difference between the two exp() shows up as
    # ~1e-7 of the column maximum in the (tiny) gradient entries there (as in the C1 drop-in test); contract 1e-4
    assert np.max(np.abs(KK - kk) / scale) < 1e-6


# ---- multi-GPU plumbing of the other shardable path: one line-by-line model split by wavenumber ---------------
def _lbl_case(seed=3, nw=4000, N=900):
    rng = np.random.default_rng(seed)
    wn = 2100.0 + 0.05 * np.arange(nw)
    nu = np.sort(rng.uniform(wn[0] - 90.0, wn[-1] + 90.0, N))
    sw = 10.0 ** rng.uniform(-26, -20, N); el = rng.uniform(0, 2000, N)
    sr = 1.0 - np.exp(-1.4387769 * nu / 296.0)
    bp = np.stack([rng.uniform(0.02, 0.1, N), rng.uniform(0.5, 0.8, N), rng.uniform(-0.05, 0.05, N)])
    return wn, nu, sw, el, sr, bp


def _oracle_kernel(orc):
    def kernel(wn, shape, t, t_ref, p, p_ref, q, iso, mass, mmf, bp, nu, sw, el, sr, out, s_floor=0.0, wn_calc_window=25.0,
               wn_approx_window=75.0):
        for l in range(len(t)):
            orc.add_line_set_monochromatic_absorption(wn, shape, t[l], t_ref, p[l], p_ref, q[l], iso, mass, mmf, bp, nu, sw, el,
                                                      sr, out[l], s_floor=s_floor, wn_calc_window=wn_calc_window,
                                                      wn_approx_window=wn_approx_window)
    return kernel


def test_lbl_wavenumber_split_is_bit_identical(oracle):
    """Every rank's slab (its grid points, the lines whose window reaches them) equals the same rows of the unsplit
    result bit for bit -- ragged split (3 ranks over 4000 points), halo including the pressure shift."""
    from archnemesis_dist_amd import lbl_shard
    wn, nu, sw, el, sr, bp = _lbl_case()
    t = np.array([180.0, 260.0]); p = np.array([0.05, 1.5]); q = np.array([1.4, 1.0]); mmf = np.array([1.0])
    kern = _oracle_kernel(oracle)
    full = np.zeros((2, wn.size))
    kern(wn, 0, t, 296.0, p, 1.0, q, 0.9, 28.0, mmf, bp, nu, sw, el, sr, full)
    seen = np.zeros(wn.size, bool)
    for r in range(3):
        slab, (i0, i1) = lbl_shard.add_line_set_sharded(kern, wn, 0, t, 296.0, p, 1.0, q, 0.9, 28.0, mmf, bp, nu, sw, el, sr,
                                                       rank=r, world_size=3, gather=False)
        assert np.array_equal(slab, full[:, i0:i1])
        seen[i0:i1] = True
    assert seen.all()
    sel = lbl_shard.lines_reaching(nu, wn[0], wn[1300], 75.0, lbl_shard.max_pressure_shift(bp, p, 1.0))
    assert isinstance(sel, slice) and 0 < sel.stop - sel.start < nu.size     # a real subset, found by bisection


def test_lbl_wavenumber_split_gloo_world2(tmp_path):
    script = textwrap.dedent(f'''
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        import numpy as np, torch.distributed as dist
        from archnemesis_dist_amd import lbl_shard
        from oracle import oracle as orc
        from test_jacobian_c3 import _lbl_case, _oracle_kernel
        orc.build()
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        wn, nu, sw, el, sr, bp = _lbl_case(nw=1501, N=400)
        t = np.array([200.0]); p = np.array([0.3]); q = np.array([1.2]); mmf = np.array([1.0])
        kern = _oracle_kernel(orc)
        full = np.zeros((1, wn.size))
        kern(wn, 0, t, 296.0, p, 1.0, q, 0.9, 28.0, mmf, bp, nu, sw, el, sr, full)
        out, (i0, i1) = lbl_shard.add_line_set_sharded(kern, wn, 0, t, 296.0, p, 1.0, q, 0.9, 28.0, mmf, bp, nu, sw, el, sr,
                                                      rank=rank, world_size=world)
        assert out.shape == full.shape and np.array_equal(out, full), rank
        dist.destroy_process_group()
        print("rank", rank, "ok", i0, i1)
    ''')
    f = tmp_path / "ls.py"
    f.write_text(script)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29623", str(f)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


@pytest.mark.gpu
def test_wavenumber_sharded_jacobian_parts_equal_the_whole():
    """What the ranks of a wavenumber-sharded Jacobian compute, on one GPU: three engines (contexts) hold the three ragged
    parts chunk_range(W, 3, r) of the k-table, each runs ALL states on its part; the parts side by side are the
    one-engine spectra bit for bit (every wavenumber is independent), and together they computed exactly as many layer
    rows as the one engine -- no row twice (in the state mode every rank repeats the unperturbed state's)."""
    import archnemesis_dist_amd as pkg
    from archnemesis_dist_amd import synthetic as syn
    from archnemesis_dist_amd.jacobian import chunk_range, perturbed_states
    from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel
    W, G, S, NPRO, NLAY = 200, 10, 3, 12, 10
    PRESS, TEMP, K = syn.synth_ktable(W, G, 6, 5, S, seed=19)
    _, delg = syn.gauss_legendre_01(G)
    WAVE = 450.0 + 0.5 * np.arange(W)
    pr = syn.synth_profiles(NPRO, 5, seed=3)
    extra = syn.synth_continuum(W, NLAY, seed=8)[0]

    def build(w0, w1):
        eng = pkg.AnsfmEngine(0)
        eng.upload_ktable(np.ascontiguousarray(K[w0:w1]), PRESS, TEMP, WAVE[w0:w1], delg)
        st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 3)])
        return eng, BatchedCKThermalModel(eng, st, pr["RADIUS"], pr["ID"], pr["ISO"], [2, 3, 4], layering_args=dict(NLAY=NLAY),
                                          IRAY=4, extra_continuum=extra[w0:w1])
    eng0, whole = build(0, W)
    X = perturbed_states(whole.state.XN, 0.05 * whole.state.XN).T
    Y = whole.spectra_batch(X).cpu().numpy()
    rows_whole = whole.last_rows
    parts, rows = [], 0
    for r in range(3):
        w0, w1 = chunk_range(W, 3, r)
        eng, m = build(w0, w1)
        m.global_waves = W
        assert m.ny_local_all(3)[r] == w1 - w0 == m.ny()
        parts.append(m.spectra_batch(X).cpu().numpy())
        assert m.last_rows == rows_whole          # the same rows per rank, each on a third of the wavenumbers
        eng.close()
    eng0.close()
    assert np.array_equal(np.concatenate(parts, axis=1), Y)
    assert rows_whole[0] < rows_whole[1] // 3     # de-duplication is on: most layers are shared with the unperturbed state


@pytest.mark.gpu
def test_rccl_collectives_under_a_one_rank_nccl_group(tmp_path):
    """The RCCL path of the Jacobian gather and of the wavenumber split, on the one GPU of this box: a fresh process
    initialises a 1-rank `nccl` process group BEFORE any other GPU call, then runs jacobian_nemesis_batched with the
    collective forced and the line-by-line split's all_gather on device tensors."""
    script = textwrap.dedent(f'''
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        import numpy as np, torch, torch.distributed as dist
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        import archnemesis_dist_amd as pkg
        from archnemesis_dist_amd import synthetic as syn, lbl_shard
        from archnemesis_dist_amd.jacobian import jacobian_nemesis_batched, gather_columns
        from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel
        x = torch.arange(12, dtype=torch.float64, device=dev).reshape(3, 4)
        y = gather_columns(x, 3, 0, 1, force=True)
        assert torch.equal(x, y) and y.is_cuda
        W, G, S, NPRO, NLAY = 128, 8, 3, 10, 8
        PRESS, TEMP, K = syn.synth_ktable(W, G, 6, 5, S, seed=9)
        _, delg = syn.gauss_legendre_01(G)
        eng = pkg.AnsfmEngine(0)
        eng.upload_ktable(K, PRESS, TEMP, 500.0 + np.arange(W), delg)
        pr = syn.synth_profiles(NPRO, 5, seed=2)
        st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T"])
        model = BatchedCKThermalModel(eng, st, pr["RADIUS"], pr["ID"], pr["ISO"], [2, 3, 4], layering_args=dict(NLAY=NLAY), IRAY=4)
        YN1, KK1 = jacobian_nemesis_batched(model)
        YN2, KK2 = jacobian_nemesis_batched(model, force_collective=True)       # all_gather_into_tensor through RCCL
        assert np.array_equal(KK1, KK2) and np.array_equal(YN1, YN2) and np.abs(KK1).max() > 0
        model.global_waves = W
        YN3, KK3 = jacobian_nemesis_batched(model, force_collective=True, shard="wavenumbers")   # the other gather, same RCCL call
        assert np.array_equal(KK1, KK3) and np.array_equal(YN1, YN3)
        # wavenumber split of a line-by-line model: world 1, the gather still goes through the collective path
        from test_jacobian_c3 import _lbl_case
        wn, nu, sw, el, sr, bp = _lbl_case(nw=3000, N=500)
        t = np.array([200.0, 250.0]); p = np.array([0.3, 1.0]); q = np.array([1.2, 1.0]); mmf = np.array([1.0])
        ref = np.zeros((2, wn.size))
        eng.add_line_set_monochromatic_absorption(wn, 0, t, 296.0, p, 1.0, q, 0.9, 28.0, mmf, bp, nu, sw, el, sr, ref)
        slab, rng_ = lbl_shard.add_line_set_sharded(eng.add_line_set_monochromatic_absorption, wn, 0, t, 296.0, p, 1.0, q, 0.9, 28.0,
                                                   mmf, bp, nu, sw, el, sr, rank=0, world_size=1, device=dev)
        assert np.array_equal(slab, ref)
        pad = torch.as_tensor(slab, device=dev)
        out = torch.empty_like(pad)
        dist.all_gather_into_tensor(out, pad)
        assert torch.equal(out, pad)
        eng.close()
        dist.destroy_process_group()
        print("nccl ok")
    ''')
    f = tmp_path / "nccl1.py"
    f.write_text(script)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    r = subprocess.run([sys.executable, str(f)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "nccl ok" in r.stdout


@pytest.mark.gpu
def test_bench_prints_one_json_line_under_torch_distributed_run():
    """The driver's launch of bench.py for N > 1 (torch.distributed.run, one rank per GPU, RCCL), rehearsed with the one GPU of
    this box: stdout must be exactly ONE JSON line (RCCL prints a version banner on file descriptor 1 at the first communicator;
    bench.py routes everything but its line to stderr), with the contract's keys, the roofline object and the Jacobian gathered
    through the collective."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
                        "--waves", "2048", "--jac-models", "21"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f64" and d["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert "RCCL" in (d["jacobian"].get("collective") or "")


@pytest.mark.gpu
def test_bench_two_ranks_rehearsed_on_one_gpu():
    """bench.py --gpus 2 the way the driver launches it, except that both ranks sit on this box's one GPU and talk through
    gloo (ANSFM_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device): the N > 1 path -- per-rank forward models, the
    Jacobian sharded over the spectral axis on a per-rank slice of the table, one all_gather -- runs end to end and gives
    the KK shape of the whole axis."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ANSFM_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29643", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
                        "--waves", "2000", "--jac-models", "21"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    j = d["jacobian"]
    assert j["kk_shape"] == [2000, 20] and "spectral axis" in j["sharding"] and "REHEARSAL" in j["collective"]
    assert j["dedup_bit_identical"] is True and j["layer_opacities_computed_rank0"] < j["layer_opacities_all_rank0"]
