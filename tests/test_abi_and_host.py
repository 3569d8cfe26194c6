"""CPU-only: the C-ABI library loads and exports every symbol include/ansfm.h declares (no compute
calls without a GPU), the host-side Jacobian logic, and the N>1 gather over gloo (world_size 2)."""
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import archnemesis_dist_amd as pkg
    pkg.build()
    lib = pkg.load()
    hdr = open(os.path.join(ROOT, "include", "ansfm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(ansfm_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ansfm.h but not exported"
    from archnemesis_dist_amd import _lib
    assert sorted(_lib.EXPORTS) == declared
    assert lib.ansfm_abi_version() == 1


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import archnemesis_dist_amd as pkg
    with pytest.raises(pkg.AnsfmError):
        pkg.AnsfmEngine(0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "archnemesis_dist_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "ansfm_oracle" not in txt, f


def test_chunk_range_matches_reference_arithmetic():
    from archnemesis_dist_amd.jacobian import chunk_range
    for nfm in (1, 7, 8, 201, 202):
        for n in (1, 2, 3, 8):
            n_jobs = min(n, nfm)
            base, rem = nfm // n_jobs, nfm % n_jobs
            got = [chunk_range(nfm, n_jobs, i) for i in range(n_jobs)]
            ref = [(i * base + min(i, rem), (i + 1) * base + min(i + 1, rem)) for i in range(n_jobs)]   # :2322-2330
            assert got == ref
            assert got[0][0] == 0 and got[-1][1] == nfm
            assert all(got[i][1] == got[i + 1][0] for i in range(n_jobs - 1))


def test_perturbed_states_and_finite_difference():
    from archnemesis_dist_amd.jacobian import perturbed_states, finite_difference_jacobian
    XN = np.array([2.0, 0.0, -3.0])
    DSTEP = 0.05 * XN                                   # Variables_0.calc_DSTEP :535
    xnx = perturbed_states(XN, DSTEP)
    assert xnx.shape == (3, 4)
    np.testing.assert_allclose(xnx[:, 0], XN)
    np.testing.assert_allclose(np.diag(xnx[:, 1:]), [2.1, 0.05, -3.15])
    # off-diagonal zeros are replaced too (the reference's mask acts on the whole block, :2241-2242)
    assert xnx[1, 1] == 0.05 and xnx[1, 3] == 0.05
    A = np.array([[1.0, 2.0, 3.0], [0.5, -1.0, 4.0]])
    f = lambda x: A @ x
    Y = np.stack([f(xnx[:, i]) for i in range(4)], axis=1)
    YN, KK = finite_difference_jacobian(Y, XN, inum=[0, 1, 2])
    np.testing.assert_allclose(YN, f(XN))
    # columns 0 and 2 see a clean one-element perturbation; column 1's run also carries xnx[1,:]=0.05
    np.testing.assert_allclose(KK[:, 2], (Y[:, 3] - YN) / (-3.0 * 0.05))


def test_gather_columns_gloo_world2(tmp_path):
    script = textwrap.dedent(f'''
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from archnemesis_dist_amd.jacobian import chunk_range, gather_columns
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        nfm, NY = 7, 5
        full = torch.arange(nfm * NY, dtype=torch.float64).reshape(nfm, NY)
        s, e = chunk_range(nfm, world, rank)
        out = gather_columns(full[s:e].clone(), nfm, rank, world)
        assert out.shape == (nfm, NY) and torch.equal(out, full), (rank, out)
        dist.destroy_process_group()
        print("rank", rank, "ok")
    ''')
    f = tmp_path / "w.py"
    f.write_text(script)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", str(f)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


class _LinearFM:
    """Stand-in forward model with the attributes jacobian_nemesis reads: a linear model y = A x (2 geometries)."""
    def __init__(self, A, XN, ncalls):
        from types import SimpleNamespace
        self.A = A
        self.ncalls = ncalls
        NX = XN.size
        V = SimpleNamespace(XN=XN.copy(), NX=NX, NUM=np.zeros(NX, dtype=int), FIX=np.zeros(NX, dtype=int), DSTEP=None)
        V.FIX[2] = 1
        V.calc_DSTEP = lambda: setattr(V, "DSTEP", 0.05 * V.XN)          # Variables_0.calc_DSTEP :535
        self.Variables = V
        self.Measurement = SimpleNamespace(NY=A.shape[0], NGEOM=2, NCONV=np.array([A.shape[0] // 2, A.shape[0] - A.shape[0] // 2]))

    def nemesisfm(self):
        self.ncalls.append(1)
        y = self.A @ self.Variables.XN
        n0 = self.Measurement.NCONV[0]
        out = np.zeros((max(self.Measurement.NCONV), 2))
        out[:n0, 0] = y[:n0]; out[:self.Measurement.NCONV[1], 1] = y[n0:]
        return out


def test_jacobian_nemesis_sharded_serial():
    from archnemesis_dist_amd.jacobian import jacobian_nemesis_sharded
    rng = np.random.default_rng(0)
    A = rng.normal(size=(7, 5)); XN = np.array([1.0, -2.0, 0.5, 3.0, 4.0])
    calls = []
    YN, KK = jacobian_nemesis_sharded(_LinearFM(A, XN, calls))
    assert len(calls) == 5                       # 1 + 4 free elements (element 2 is FIXed)
    np.testing.assert_allclose(YN, A @ XN)
    free = [0, 1, 3, 4]
    np.testing.assert_allclose(KK[:, free], A[:, free], rtol=1e-12)
    assert np.all(KK[:, 2] == 0.0)


def test_jacobian_nemesis_sharded_gloo_world2(tmp_path):
    script = textwrap.dedent(f'''
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        import numpy as np, torch.distributed as dist
        from archnemesis_dist_amd.jacobian import jacobian_nemesis_sharded
        from test_abi_and_host import _LinearFM
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        rng = np.random.default_rng(0)
        A = rng.normal(size=(7, 5)); XN = np.array([1.0, -2.0, 0.5, 3.0, 4.0])
        calls = []
        YN, KK = jacobian_nemesis_sharded(_LinearFM(A, XN, calls), rank=rank, world_size=world)
        assert len(calls) in (2, 3), calls                      # 5 forward models over 2 ranks: 3 + 2
        np.testing.assert_allclose(YN, A @ XN)
        np.testing.assert_allclose(KK[:, [0, 1, 3, 4]], A[:, [0, 1, 3, 4]], rtol=1e-12)
        dist.destroy_process_group()
        print("rank", rank, "ok", len(calls))
    ''')
    f = tmp_path / "j.py"
    f.write_text(script)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29618", str(f)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_strict_switch_turns_delegations_into_errors():
    """forward_model.set_strict: a case outside the GPU path is announced once and counted (default) or refused (strict)."""
    import archnemesis_dist_amd.forward_model as fmod
    before = dict(fmod.DELEGATED)
    try:
        fmod.set_strict(False)
        fmod.DELEGATED.pop("unit-test case", None)
        with pytest.warns(RuntimeWarning, match="outside the GPU path"):      # said aloud the first time ...
            fmod._delegate("unit-test case")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error")                                      # ... counted silently afterwards
            fmod._delegate("unit-test case")
        assert fmod.DELEGATED["unit-test case"] == 2
        fmod.set_strict(True)
        with pytest.raises(NotImplementedError):
            fmod._delegate("unit-test case")
    finally:
        fmod.set_strict(False)
        fmod.DELEGATED.pop("unit-test case", None)
