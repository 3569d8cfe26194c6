"""Fixtures for the native .lta reader: small LBL-table files written by the REFERENCE's write_lbltable
(Spectroscopy_0.py:2856) and what its read_ltahead (:2451) / read_lbltable (:2626) return for them (full range and a
sub-range).   Build container only.   python oracle/gen_golden_lta.py  -> tests/golden/kta/*.lta, tests/golden/lta_read.npz"""
import os
import sys
import importlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
HEAD = ["nwave", "vmin", "delv", "npress", "ntemp", "gasID", "isoID", "presslevels", "templevels"]


def main():
    import_reference()
    sp = importlib.import_module("archnemesis.Spectroscopy_0")
    os.makedirs(os.path.join(OUT, "kta"), exist_ok=True)
    rng = np.random.default_rng(77)
    W, NP, NT = 40, 5, 4
    PRESS = np.logspace(-4, 0.5, NP); TEMP = np.linspace(120.0, 330.0, NT)
    vmin, delv = 2000.0, 0.125
    out = {}
    for gi, (gid, iso) in enumerate(((2, 1), (5, 0))):
        k = 10.0 ** rng.uniform(-27, -19, (W, NP, NT))
        k[5, 1, :] = 0.0
        fn = os.path.join(OUT, "kta", f"lbl_gas{gi}.lta")
        sp.write_lbltable(fn, NP, NT, gid, iso, PRESS, TEMP, W, vmin, delv, k)
        h = sp.read_ltahead(fn)
        for n, v in zip(HEAD, h):
            out[f"g{gi}_head_{n}"] = np.asarray(v)
        wv = np.linspace(h[1], h[1] + h[2] * (h[0] - 1), h[0])
        for rn, (lo, hi) in (("all", (0.0, 1e10)), ("sub", (float(wv[7]), float(wv[29])))):
            r = sp.read_lbltable(fn, lo, hi)
            out[f"g{gi}_{rn}_wave"] = np.asarray(r[7]); out[f"g{gi}_{rn}_k"] = np.asarray(r[8])
            out[f"g{gi}_{rn}_range"] = np.array([lo, hi])
        print(fn, os.path.getsize(fn), h[:7], np.abs(out[f"g{gi}_all_k"] / np.where(k > 0, k, 1) - 1)[k > 0].max())
    np.savez_compressed(os.path.join(OUT, "lta_read.npz"), **out)


if __name__ == "__main__":
    main()
