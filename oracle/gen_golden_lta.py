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
    # NT < 0: one grid of |NT| temperatures per pressure level.  The reference's write_lbltable cannot write such a file (its
    # record offset 9 + npress + ntemp goes wrong for a negative ntemp) and its readers take |NT| = 2 only (templevels =
    # zeros((npress, 2)), :2480 / :2685), so the file is laid out here by hand in the format those readers parse -- header,
    # pressure levels, npress x 2 temperatures, then k * 1e20 [wave][press][temp] from record irec0 -- and READ BACK by the
    # reference for the expected arrays.
    import struct
    NT2 = 2
    TEMP2 = np.stack([np.linspace(100.0, 260.0, NP), np.linspace(180.0, 340.0, NP)], axis=1).astype(np.float32)
    k = 10.0 ** rng.uniform(-27, -19, (W, NP, NT2))
    fn = os.path.join(OUT, "kta", "lbl_perlevel.lta")
    with open(fn, "wb") as f:
        irec0 = 9 + NP + NP * NT2
        f.write(struct.pack("i", irec0)); f.write(struct.pack("i", W)); f.write(struct.pack("f", vmin)); f.write(struct.pack("f", delv))
        f.write(struct.pack("i", NP)); f.write(struct.pack("i", -NT2)); f.write(struct.pack("i", 6)); f.write(struct.pack("i", 1))
        f.write(np.asarray(PRESS, np.float32).tobytes()); f.write(TEMP2.tobytes())
        f.write((k * 1.0e20).astype(np.float32).tobytes())
    h = sp.read_ltahead(fn)
    for n, v in zip(HEAD, h):
        out[f"pl_head_{n}"] = np.asarray(v)
    for rn, (lo, hi) in (("all", (0.0, 1e10)), ("sub", (2001.0, 2003.5))):
        r = sp.read_lbltable(fn, lo, hi)
        out[f"pl_{rn}_wave"] = np.asarray(r[7]); out[f"pl_{rn}_k"] = np.asarray(r[8]); out[f"pl_{rn}_range"] = np.array([lo, hi])
    assert out["pl_head_ntemp"] == -2 and out["pl_head_templevels"].shape == (NP, 2)
    print(fn, os.path.getsize(fn), h[:7])
    np.savez_compressed(os.path.join(OUT, "lta_read.npz"), **out)


if __name__ == "__main__":
    main()
