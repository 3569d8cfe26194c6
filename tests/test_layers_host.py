"""Host-side layering / path geometry (archnemesis_dist_amd.layering) vs goldens from the reference's
Layer_0.layer_split and AtmCalc_0."""
import os
import numpy as np
import pytest


def test_layer_split(golden_dir):
    from archnemesis_dist_amd import layering as layers
    z = np.load(os.path.join(golden_dir, "layer_average.npz"))
    for typ in range(4):
        bh, bp = layers.layer_split(float(z["RADIUS"]), z["H"], z["P"], LAYANG=20.0, LAYHT=-6.0e4, NLAY=17, LAYTYP=typ)
        np.testing.assert_allclose(bh, z[f"split{typ}_BASEH"], rtol=1e-13, atol=1e-6)
        np.testing.assert_allclose(bp, z[f"split{typ}_BASEP"], rtol=1e-13)
    bh, bp = layers.layer_split(float(z["RADIUS"]), z["H"], z["P"], LAYHT=-6.0e4, LAYTYP=5, H_base=np.linspace(-6e4, 4e5, 9))
    np.testing.assert_allclose(bh, z["split5_BASEH"], rtol=1e-13)
    np.testing.assert_allclose(bp, z["split5_BASEP"], rtol=1e-13)


def test_calc_path(golden_dir):
    from archnemesis_dist_amd import layering as layers
    z = np.load(os.path.join(golden_dir, "path_geometry.npz"))
    for n in z["names"]:
        n = str(n)
        pointing, botlay, angle, emiss, ipzen, pc = z[n + "_args"]
        r = layers.calc_path(float(z["RADIUS"]), z["BASEH"], z["DELH"], z["TEMP"], float(z["H_top"]), pointing=int(pointing),
                             BOTLAY=int(botlay), ANGLE=float(angle), EMISS_ANG=float(emiss), IPZEN=int(ipzen), path_calc=int(pc))
        assert np.array_equal(r["NLAYIN"], z[n + "_NLAYIN"]), n
        assert np.array_equal(r["LAYINC"], z[n + "_LAYINC"]), n
        np.testing.assert_allclose(r["SCALE"], z[n + "_SCALE"], rtol=1e-13, err_msg=n)
        np.testing.assert_allclose(r["EMTEMP"], z[n + "_EMTEMP"], rtol=0, err_msg=n)
        assert np.array_equal(r["IMOD"], z[n + "_IMOD"]), n
