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


@pytest.mark.parametrize("ishape", [0, 1, 2, 3, 4])
def test_lblconv_ngeom_shapes(oracle, golden_dir, ishape):
    """lblconv_ngeom (:3444) / lblconvg_ngeom (:3685): several geometries on one grid; their own Hamming window."""
    z = np.load(os.path.join(golden_dir, "ils_conv.npz"))
    nw, nc, fw = z["vwave"].size, z["vconv"].size, float(z["fwhm"])
    close_nan(oracle.lblconv(nw, z["vwave"], z["y_ngeom"], nc, z["vconv"], ishape, fw, ngeom=True), z[f"ngconv_{ishape}"], 1e-13)
    yo, go = oracle.lblconv(nw, z["vwave"], z["y_ngeom"], nc, z["vconv"], ishape, fw, dydx=z["dydx_ngeom"], ngeom=True)
    close_nan(yo, z[f"ngconvg_{ishape}_y"], 1e-13)
    close_nan(go, z[f"ngconvg_{ishape}_g"], 1e-13)


def test_lblconv_fil_ngeom(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "ils_conv.npz"))
    nw, nc = z["vwave"].size, z["vconv"].size
    close_nan(oracle.lblconv_fil(nw, z["vwave"], z["y_ngeom"], nc, z["vconv"], z["nfil"], z["vfil"], z["afil"], ngeom=True),
              z["ngfil_y"], 1e-13)
    yo, go = oracle.lblconv_fil(nw, z["vwave"], z["y_ngeom"], nc, z["vconv"], z["nfil"], z["vfil"], z["afil"],
                                dydx=z["dydx_ngeom"], ngeom=True)
    close_nan(yo, z["ngfilg_y"], 1e-13)
    close_nan(go, z["ngfilg_g"], 1e-13)


def test_ktable_conv_filter_branch(oracle, golden_dir):
    """FWHM < 0 branch of Measurement_0.conv / convg (:2425-2461, :2655-2691): window bracketing each filter."""
    z = np.load(os.path.join(golden_dir, "ils_conv.npz"))
    nw, nc = z["vwave"].size, z["vconv"].size
    a = (nw, z["vwave"], z["y"], nc, z["vconv"], z["nfil"], z["vfil"], z["afil_k"])
    close_nan(oracle.lblconv_fil(*a, bracket=True), z["kconv_fil"], 1e-13)
    yo, go = oracle.lblconv_fil(*a, dydx=z["dydx"], bracket=True)
    close_nan(yo, z["kconvg_fil_y"], 1e-13)
    close_nan(go, z["kconvg_fil_g"], 1e-13)
    assert np.max(np.abs(z["kconv_fil"] / oracle.lblconv_fil(*a) - 1)) > 1e-6      # the bracketing points do carry weight


def test_integrate_filter_family(oracle, golden_dir):
    """integrate_filter / integrate_filterg and the *_ngeom variants (:4079-4300): np.trapz of filter x spectrum"""
    z = np.load(os.path.join(golden_dir, "ils_conv.npz"))
    nw, nc = z["vwave"].size, z["vconv"].size
    f = (nc, z["vconv"], z["nfil"], z["vfil"], z["afil"])
    np.testing.assert_allclose(oracle.integrate_filter(nw, z["vwave"], z["y"], *f), z["intf"], rtol=1e-13)
    yo, go = oracle.integrate_filter(nw, z["vwave"], z["y"], *f, dydx=z["dydx"])
    np.testing.assert_allclose(yo, z["intfg_y"], rtol=1e-13)
    np.testing.assert_allclose(go, z["intfg_g"], rtol=0, atol=1e-13 * np.abs(z["intfg_g"]).max())
    np.testing.assert_allclose(oracle.integrate_filter(nw, z["vwave"], z["y_ngeom"], *f), z["ngintf"], rtol=1e-13)
    yo, go = oracle.integrate_filter(nw, z["vwave"], z["y_ngeom"], *f, dydx=z["dydx_ngeom"])
    np.testing.assert_allclose(yo, z["ngintfg_y"], rtol=1e-13)
    np.testing.assert_allclose(go, z["ngintfg_g"], rtol=0, atol=1e-13 * np.abs(z["ngintfg_g"]).max())
