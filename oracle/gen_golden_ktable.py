"""Golden fixture for the k-table generator: the REFERENCE's Spectroscopy_0.calc_ktable_chunk (Spectroscopy_0.py:3558)
driven by the stand-ins of tests/ktable_fakes.py (analytic line-by-line spectrum, no line database), without and with an
instrument function per bin.   Build container only.   python oracle/gen_golden_ktable.py"""
import os
import sys
import importlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.ref_import import import_reference  # noqa: E402
from ktable_fakes import make_case  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    import_reference()
    sp = importlib.import_module("archnemesis.Spectroscopy_0")
    out = {}
    for tag, wf in (("plain", False), ("ils", True)):
        iw, S, L, sf, M = make_case(wf)
        out[tag] = sp.calc_ktable_chunk(iw, S, L, sf, M)
        print(tag, out[tag].shape, out[tag].min(), out[tag].max(), L.NWAVE)
    np.savez_compressed(os.path.join(OUT, "ktable_chunk.npz"), **out)


if __name__ == "__main__":
    main()
