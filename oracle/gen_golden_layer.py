"""Golden fixtures for the layering step: the REFERENCE's Layer_0.layer_split (:1402) and layer_average (:755)
on a seeded synthetic profile (build container only).   python oracle/gen_golden_layer.py"""
import os
import sys
import importlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import_reference()
    L0 = importlib.import_module("archnemesis.Layer_0")
    rng = np.random.default_rng(77)
    NPRO, NV, ND, NLAY = 45, 4, 2, 17
    RADIUS = 7.1492e7
    H = np.linspace(-8.0e4, 5.2e5, NPRO) + rng.uniform(-2e3, 2e3, NPRO); H = np.sort(H)
    P = 6.0e5 * np.exp(-(H - H[0]) / 2.7e4)
    T = 110.0 + 60.0 * np.exp(-((H - 5e4) / 9e4) ** 2) + 3e-4 * np.maximum(H - 2e5, 0)
    VMR = 10.0 ** rng.uniform(-8, -2, (1, NV)) * (1 + 0.5 * np.sin(np.linspace(0, 3, NPRO))[:, None])
    DUST = 10.0 ** rng.uniform(1, 4, (1, ND)) * np.exp(-(H[:, None] - H[0]) / 4e4)
    PARAH2 = 0.25 + 0.1 * np.cos(np.linspace(0, 2, NPRO))
    XMOLWT = np.full(NPRO, 2.3e-3)
    out = dict(RADIUS=RADIUS, H=H, P=P, T=T, VMR=VMR, DUST=DUST, PARAH2=PARAH2, XMOLWT=XMOLWT)
    for typ in range(4):
        bh, bp = L0.layer_split(RADIUS, H, P, LAYANG=20.0, LAYHT=-6.0e4, NLAY=NLAY, LAYTYP=typ)
        out[f"split{typ}_BASEH"] = bh; out[f"split{typ}_BASEP"] = bp
    bh5, bp5 = L0.layer_split(RADIUS, H, P, LAYHT=-6.0e4, LAYTYP=5, H_base=np.linspace(-6e4, 4e5, 9))
    out["split5_BASEH"] = bh5; out["split5_BASEP"] = bp5
    BASEH, BASEP = out["split1_BASEH"], out["split1_BASEP"]
    names = ["HEIGHT", "PRESS", "TEMP", "TOTAM", "AMOUNT", "PP", "CONT", "FRAC", "DELH", "BASET", "LAYSF"]
    cases = {"cg_nadir": dict(LAYANG=0.0, LAYINT=1), "cg_slant": dict(LAYANG=35.0, LAYINT=1),
             "mid_slant": dict(LAYANG=35.0, LAYINT=0), "cg_dustunits": dict(LAYANG=10.0, LAYINT=1, DUST_UNITS=np.array([-1, 0]), XMOLWT=XMOLWT.copy())}
    for cn, kw in cases.items():
        r = L0.layer_average(RADIUS, H, P, T, np.arange(NV), VMR, DUST, PARAH2, BASEH, BASEP, LAYHT=-6.0e4, NINT=101, **kw)
        for n, v in zip(names, r):
            out[f"{cn}_{n}"] = np.asarray(v)
        print(cn, r[3][:3])
    np.savez_compressed(os.path.join(OUT, "layer_average.npz"), **out)
    # ---- layer_averageg (:1032): same inputs, plus the DTE/DAM/DCO/DPH matrices -------------------------------
    gnames = names + ["DTE", "DAM", "DCO", "DPH"]
    gout = {k: out[k] for k in ("RADIUS", "H", "P", "T", "VMR", "DUST", "PARAH2", "XMOLWT", "split1_BASEH", "split1_BASEP")}
    for cn, kw in cases.items():
        kw = dict(kw)
        if "XMOLWT" in kw:
            kw["XMOLWT"] = XMOLWT.copy()            # the reference scales it in place (:1144, :1394)
        r = L0.layer_averageg(RADIUS, H, P, T, np.arange(NV), VMR, DUST, PARAH2, BASEH, BASEP, LAYHT=-6.0e4, NINT=101, **kw)
        for n, v in zip(gnames, r):
            gout[f"{cn}_{n}"] = np.asarray(v)
        print("g", cn, np.abs(r[11]).sum(), np.abs(r[12]).sum(), np.abs(r[13]).sum())
    np.savez_compressed(os.path.join(OUT, "layer_averageg.npz"), **gout)
    # ---- layer_average with an even number of sub-points (scipy's simpson end correction) and NINT = 2 (trapezoid)
    eout = {}
    for nint in (100, 2, 4):
        r = L0.layer_average(RADIUS, H, P, T, np.arange(NV), VMR, DUST, PARAH2, BASEH, BASEP, LAYANG=35.0, LAYINT=1,
                             LAYHT=-6.0e4, NINT=nint)
        for n, v in zip(names, r):
            eout[f"nint{nint}_{n}"] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "layer_average_even_nint.npz"), **eout)


if __name__ == "__main__":
    main()
