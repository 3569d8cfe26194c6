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


GNAMES = NAMES + ["DTE", "DAM", "DCO", "DPH"]


@pytest.mark.parametrize("case", list(CASES))
def test_layer_averageg(oracle, golden_dir, case):
    """Layer_0.layer_averageg (:1032): layer properties through `interpg` plus the DTE/DAM/DCO/DPH matrices."""
    z = np.load(os.path.join(golden_dir, "layer_averageg.npz"))
    kw = dict(CASES[case])
    du = kw.pop("dust_units", False)
    r = oracle.layer_averageg(float(z["RADIUS"]), z["H"], z["P"], z["T"], None, z["VMR"], z["DUST"], z["PARAH2"],
                              z["split1_BASEH"], z["split1_BASEP"], LAYHT=-6.0e4, NINT=101,
                              DUST_UNITS=np.array([-1, 0]) if du else None, XMOLWT=z["XMOLWT"] if du else None, **kw)
    for n, v in zip(GNAMES, r):
        ref = z[f"{case}_{n}"]
        np.testing.assert_allclose(v, ref, rtol=1e-11, atol=1e-13 * np.max(np.abs(ref)), err_msg=n)


def test_layer_averageg_reference_failures(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "layer_averageg.npz"))
    a = (float(z["RADIUS"]), z["H"], z["P"], z["T"], None, z["VMR"], z["DUST"], z["PARAH2"], z["split1_BASEH"], z["split1_BASEP"])
    with pytest.raises(ValueError):      # :1188
        oracle.layer_averageg(*a, LAYHT=-6.0e4, NINT=100, LAYINT=1)
    with pytest.raises(ValueError):      # mis-indented else of the MID_PATH branch (:1255-1257)
        oracle.layer_averageg(*a, LAYHT=-6.0e4, NINT=101, LAYINT=0, DUST_UNITS=np.array([-1, 0]), XMOLWT=z["XMOLWT"])


@pytest.mark.parametrize("nint", [100, 2, 4])
def test_layer_average_even_nint(oracle, golden_dir, nint):
    """scipy.integrate.simpson with an even number of points (last-interval correction) / two points (trapezoid)."""
    z = np.load(os.path.join(golden_dir, "layer_average.npz"))
    e = np.load(os.path.join(golden_dir, "layer_average_even_nint.npz"))
    r = oracle.layer_average(float(z["RADIUS"]), z["H"], z["P"], z["T"], None, z["VMR"], z["DUST"], z["PARAH2"],
                             z["split1_BASEH"], z["split1_BASEP"], LAYANG=35.0, LAYINT=1, LAYHT=-6.0e4, NINT=nint)
    for n, v in zip(NAMES, r):
        np.testing.assert_allclose(v, e[f"nint{nint}_{n}"], rtol=1e-11, err_msg=n)
