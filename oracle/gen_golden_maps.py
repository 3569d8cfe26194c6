"""Golden fixture for the gradient maps: the REFERENCE's ForwardModel_0.map2pro (:5319) and map2xvec (:5387) on
seeded inputs, two paths of different length, INCPAR default and explicit (with the para-H2 slot listed).
Build container only.   python oracle/gen_golden_maps.py"""
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
    FM = importlib.import_module("archnemesis.ForwardModel_0")
    rng = np.random.default_rng(4242)
    W, NVMR, NDUST, NPRO, NPATH, NLAY, LIMAX, NX = 37, 3, 2, 21, 2, 13, 11, 9
    NPAR = NVMR + 2 + NDUST
    NLAYIN = np.array([11, 7], dtype=np.int32)
    LAYINC = np.zeros((LIMAX, NPATH), dtype=np.int32)
    LAYINC[:, 0] = rng.permutation(NLAY)[:LIMAX]
    LAYINC[:7, 1] = np.arange(7)[::-1]
    dS = rng.normal(size=(W, NPAR, LIMAX, NPATH)) * 10.0 ** rng.uniform(-12, -6, (1, NPAR, 1, 1))
    dS[:, :, 7:, 1] = 0.0                                   # beyond NLAYIN of the short path
    DTE, DAM, DCO = (rng.uniform(0, 1, (NLAY, NPRO)) * 10.0 ** rng.uniform(-2, 2) for _ in range(3))
    xmap = rng.normal(size=(NX, NPAR, NPRO)) * (rng.uniform(size=(NX, NPAR, 1)) < 0.6)
    out = dict(dSPECIN=dS, NLAYIN=NLAYIN, LAYINC=LAYINC, DTE=DTE, DAM=DAM, DCO=DCO, xmap=xmap,
               dims=np.array([W, NVMR, NDUST, NPRO, NPATH, NX]))
    full = FM.map2pro(dS, W, NVMR, NDUST, NPRO, NPATH, NLAYIN, LAYINC, DTE, DAM, DCO)
    out["pro_all"] = full
    inc = [0, 2, 3, 5, 6]                                   # gas, gas, temperature, dust, para-H2 (stale slot)
    out["incpar"] = np.array(inc)
    out["pro_inc"] = FM.map2pro(dS, W, NVMR, NDUST, NPRO, NPATH, NLAYIN, LAYINC, DTE, DAM, DCO, INCPAR=inc)
    out["xvec_all"] = FM.map2xvec(full, W, NVMR, NDUST, NPRO, NPATH, NX, xmap)
    np.savez_compressed(os.path.join(OUT, "gradient_maps.npz"), **out)
    print("wrote gradient_maps.npz", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
