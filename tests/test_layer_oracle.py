"""Layer_0.layer_average (Curtis-Godson and mid-path): CPU oracle vs goldens from the reference."""
import os
import numpy as np
import pytest

NAMES = ["HEIGHT", "PRESS", "TEMP", "TOTAM", "AMOUNT", "PP", "CONT", "FRAC", "DELH", "BASET", "LAYSF"]
CASES = {"cg_nadir": dict(LAYANG=0.0, LAYINT=1), "cg_slant": dict(LAYANG=35.0, LAYINT=1),
         "mid_slant": dict(LAYANG=35.0, LAYINT=0), "cg_dustunits": dict(LAYANG=10.0, LAYINT=1, dust_units=True)}


@pytest.mark.parametrize("case", list(CASES))
def test_layer_average(oracle, golden_dir, case):
    z = np.load(os.path.join(golden_dir, "layer_average.npz"))
    kw = dict(CASES[case])
    du = kw.pop("dust_units", False)
    r = oracle.layer_average(float(z["RADIUS"]), z["H"], z["P"], z["T"], None, z["VMR"], z["DUST"], z["PARAH2"],
                             z["split1_BASEH"], z["split1_BASEP"], LAYHT=-6.0e4, NINT=101,
                             DUST_UNITS=np.array([-1, 0]) if du else None, XMOLWT=z["XMOLWT"] if du else None, **kw)
    for n, v in zip(NAMES, r):
        np.testing.assert_allclose(v, z[f"{case}_{n}"], rtol=1e-11, err_msg=n)
