"""Optimal-estimation linear algebra on the GPU vs the NumPy expressions of OptimalEstimation_0.py:545-563 / :700-716."""
import numpy as np
import pytest


def _case(NY, NX, seed, se_scalar=False):
    rng = np.random.default_rng(seed)
    KK = rng.normal(size=(NY, NX)) * 10.0 ** rng.uniform(-3, 0, (1, NX))
    A = rng.normal(size=(NX, NX)); SA = A @ A.T / NX + np.eye(NX) * 0.1
    SE = np.array([[0.04]]) if se_scalar else np.diag(rng.uniform(0.01, 0.1, NY))
    return KK, SA, SE


def _numpy_gain(KK, SA, SE):
    sa_kt = SA @ KK.T
    M = KK @ sa_kt + SE
    DD = np.linalg.solve(M.T, sa_kt.T).T
    return DD, DD @ KK


@pytest.mark.gpu
@pytest.mark.parametrize("NY,NX,scalar", [(600, 80, False), (150, 200, True)])
def test_gain_matrix_and_errors(NY, NX, scalar):
    from archnemesis_dist_amd import oe_linalg as oe
    KK, SA, SE = _case(NY, NX, NY + NX, scalar)
    DD, AA = oe.calc_gain_matrix(KK, SA, SE)
    rDD, rAA = _numpy_gain(KK, SA, SE)
    np.testing.assert_allclose(DD, rDD, rtol=0, atol=1e-10 * np.abs(rDD).max())
    np.testing.assert_allclose(AA, rAA, rtol=0, atol=1e-10 * np.abs(rAA).max())
    SEf = SE if not scalar else np.eye(NY) * SE[0, 0]
    SM, SN, ST = oe.calc_serr(DD, AA, SA, SEf)
    b = rAA - np.eye(NX)
    rSM = (rDD @ SEf) @ rDD.T; rSN = (b @ SA) @ b.T
    np.testing.assert_allclose(SM, rSM, rtol=0, atol=1e-10 * np.abs(rSM).max())
    np.testing.assert_allclose(ST, rSN + rSM, rtol=0, atol=1e-10 * np.abs(rSN + rSM).max())
    SM2, _, _ = oe.calc_serr(DD, AA, SA, SEf, simple=True)
    np.testing.assert_allclose(SM2, (rDD * SEf[0, 0]) @ rDD.T, rtol=0, atol=1e-10 * np.abs(rSM).max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["scalar", "diag", "full"])
def test_phiret_and_next_xn(kind):
    """cost function (:573-610, the three SE cases) and the state update (:655-677) vs the NumPy expressions"""
    from archnemesis_dist_amd import oe_linalg as oe
    NY, NX = 400, 60
    KK, SA, SE = _case(NY, NX, 77, kind == "scalar")
    rng = np.random.default_rng(5)
    if kind == "full":
        B = rng.normal(size=(NY, NY)) * 0.01
        SE = SE + B @ B.T
    Y = rng.normal(size=NY); YN = Y + rng.normal(size=NY) * 0.1
    XA = rng.normal(size=NX); XN = XA + rng.normal(size=NX) * 0.2
    PHI, CHISQ = oe.calc_phiret(Y, YN, XN, XA, SE, SA)
    b = YN - Y; d = XN - XA
    meas = b @ b / SE[0, 0] if kind == "scalar" else b @ np.linalg.solve(SE, b)
    np.testing.assert_allclose(CHISQ, meas / NY, rtol=1e-11)
    np.testing.assert_allclose(PHI, meas + d @ np.linalg.solve(SA, d), rtol=1e-11)
    DD, AA = _numpy_gain(KK, SA, SE if kind != "scalar" else np.eye(NY) * SE[0, 0])
    np.testing.assert_allclose(oe.calc_next_xn(XA, XN, Y, YN, DD, AA), XA + DD @ (Y - YN) - AA @ (XA - XN), rtol=1e-11, atol=1e-13)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from archnemesis_dist_amd import oe_linalg as oe
    with pytest.raises(RuntimeError):
        oe.calc_gain_matrix(*_case(10, 4, 1))
