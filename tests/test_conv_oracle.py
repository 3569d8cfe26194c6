"""ILS convolution kernels (Measurement_0.lblconv / lblconvg / lblconv_fil / lblconvg_fil): oracle vs reference goldens."""
import os
import numpy as np
import pytest


def close_nan(a, b, rtol):
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(b)
    np.testing.assert_allclose(a[m], b[m], rtol=rtol)


@pytest.mark.parametrize("ishape", range(5))
def test_lblconv_shapes(oracle, golden_dir, ishape):
    z = np.load(os.path.join(golden_dir, "ils_conv.npz"))
    nw, nc, fw = z["vwave"].size, z["vconv"].size, float(z["fwhm"])
    close_nan(oracle.lblconv(nw, z["vwave"], z["y"], nc, z["vconv"], ishape, fw), z[f"conv_{ishape}"], 1e-13)
    yo, go = oracle.lblconv(nw, z["vwave"], z["y"], nc, z["vconv"], ishape, fw, dydx=z["dydx"])
    close_nan(yo, z[f"convg_{ishape}_y"], 1e-13)
    close_nan(go, z[f"convg_{ishape}_g"], 1e-12)


def test_lblconv_filters(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "ils_conv.npz"))
    nw, nc = z["vwave"].size, z["vconv"].size
    close_nan(oracle.lblconv_fil(nw, z["vwave"], z["y"], nc, z["vconv"], z["nfil"], z["vfil"], z["afil"]), z["fil"], 1e-13)
    yo, go = oracle.lblconv_fil(nw, z["vwave"], z["y"], nc, z["vconv"], z["nfil"], z["vfil"], z["afil"], dydx=z["dydx"])
    close_nan(yo, z["filg_y"], 1e-13)
    close_nan(go, z["filg_g"], 1e-12)
