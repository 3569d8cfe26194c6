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
