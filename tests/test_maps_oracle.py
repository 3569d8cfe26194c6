"""ForwardModel_0.map2pro / map2xvec: CPU oracle vs goldens from the reference."""
import os
import numpy as np
import pytest


def _z(golden_dir):
    z = np.load(os.path.join(golden_dir, "gradient_maps.npz"))
    W, NVMR, NDUST, NPRO, NPATH, NX = (int(v) for v in z["dims"])
    return z, W, NVMR, NDUST, NPRO, NPATH, NX


def _scale_close(a, b, rtol):
    np.testing.assert_allclose(a, b, rtol=0, atol=rtol * np.max(np.abs(b)))


def test_map2pro(oracle, golden_dir):
    z, W, NVMR, NDUST, NPRO, NPATH, NX = _z(golden_dir)
    a = oracle.map2pro(z["dSPECIN"], W, NVMR, NDUST, NPRO, NPATH, z["NLAYIN"], z["LAYINC"], z["DTE"], z["DAM"], z["DCO"])
    # BLAS vs einsum summation order: compare per parameter slot on that slot's scale
    for par in range(NVMR + 2 + NDUST):
        _scale_close(a[:, par], z["pro_all"][:, par], 1e-13)
    b = oracle.map2pro(z["dSPECIN"], W, NVMR, NDUST, NPRO, NPATH, z["NLAYIN"], z["LAYINC"], z["DTE"], z["DAM"], z["DCO"],
                       INCPAR=list(z["incpar"]))
    for par in range(NVMR + 2 + NDUST):
        _scale_close(b[:, par], z["pro_inc"][:, par], 1e-13)
    # the para-H2 slot holds the previous listed parameter's product (stale dSPECOUT1), unlisted slots are zero
    assert np.array_equal(b[:, 6], b[:, 5]) and not np.any(b[:, 1]) and not np.any(b[:, 4])
    with pytest.raises(UnboundLocalError):
        oracle.map2pro(z["dSPECIN"], W, NVMR, NDUST, NPRO, NPATH, z["NLAYIN"], z["LAYINC"], z["DTE"], z["DAM"], z["DCO"], INCPAR=[6, 0])


def test_map2xvec(oracle, golden_dir):
    z, W, NVMR, NDUST, NPRO, NPATH, NX = _z(golden_dir)
    a = oracle.map2xvec(z["pro_all"], W, NVMR, NDUST, NPRO, NPATH, NX, z["xmap"])
    _scale_close(a, z["xvec_all"], 1e-13)
