"""Rayleigh and aerosol continuum (ForwardModel_0.calc_tau_rayleigh* / calc_tau_dust): oracle vs reference goldens."""
import os
import numpy as np
import pytest

MODES = [("j", 1), ("v", "v"), ("v2", 2), ("ls", 4)]


@pytest.mark.parametrize("ispace", [0, 1])
@pytest.mark.parametrize("name,mode", MODES)
def test_rayleigh(oracle, golden_dir, name, mode, ispace):
    z = np.load(os.path.join(golden_dir, "continuum_ray_dust.npz"))
    w = z["wn"] if ispace == 0 else z["wl"]
    t, d = oracle.calc_tau_rayleigh(mode, ispace, w, z["TOTAM"], z["ID"], z["ISO"], z["VMR"])
    np.testing.assert_allclose(t, z[f"ray_{name}_{ispace}_tau"], rtol=1e-13)
    np.testing.assert_allclose(d, z[f"ray_{name}_{ispace}_dtau"], rtol=1e-13)


@pytest.mark.parametrize("pre,rows", [("dust", slice(None)), ("dust2", [0, -1])])
def test_dust(oracle, golden_dir, pre, rows):
    """cubic interp1d with the reference's fall-back to the linear interpolant where the spline leaves the physical
    range (62 of 211 points in this fixture), and the two-point (linear) table"""
    z = np.load(os.path.join(golden_dir, "continuum_ray_dust.npz"))
    r = oracle.calc_tau_dust(z["WAVEC_D"], z["SW"][rows], z["KEXT"][rows], z["KSCA"][rows], z["CONT"])
    for n, a in zip(("TAUDUST", "TAUCLSCAT", "dTAUDUSTdq", "dTAUCLSCATdq"), r):
        e = z[f"{pre}_{n}"]
        assert np.all(np.abs(a - e) <= 1e-13 * np.abs(e).max(axis=(0, 1), keepdims=True)), n
