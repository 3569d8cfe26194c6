"""Fixtures for the native .kta reader: small k-table files written by the REFERENCE's write_ktable (Spectroscopy_0.py:2951)
-- a uniform wavenumber grid (delv > 0) and an explicit one (delv <= 0, wave list in the file) -- and what the
reference's read_ktahead (:2492) / read_ktable (:2733) return for them (full range and a sub-range).
Build container only.   python oracle/gen_golden_kta.py    -> tests/golden/kta/*.kta, tests/golden/kta_read.npz"""
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
    sp = importlib.import_module("archnemesis.Spectroscopy_0")
    os.makedirs(os.path.join(OUT, "kta"), exist_ok=True)
    rng = np.random.default_rng(2026)
    W, G, NP, NT = 14, 5, 4, 3
    x, w = np.polynomial.legendre.leggauss(G)
    g_ord = (0.5 * (x + 1)).astype(np.float32); del_g = (0.5 * w).astype(np.float32)
    PRESS = np.logspace(-3, 1, NP); TEMP = np.linspace(100.0, 300.0, NT)
    out = {}
    for tag, vmin, delv, wave in (("uni", 612.5, 2.5, None), ("list", 0.0, -1.0, np.sort(rng.uniform(5.0, 90.0, W)))):
        for gi, (gid, iso) in enumerate(((6, 1), (11, 0))):
            k = np.sort(10.0 ** rng.uniform(-26, -19, (W, G, NP, NT)), axis=1)
            k[2, :2] = 0.0
            fn = os.path.join(OUT, "kta", f"{tag}_gas{gi}.kta")
            if wave is None:
                sp.write_ktable(fn, gid, iso, g_ord, del_g, PRESS, TEMP, W, vmin, delv, 0.0, k)
            else:
                sp.write_ktable(fn, gid, iso, g_ord, del_g, PRESS, TEMP, W, wave[0], delv, 0.0, k, wave=wave)
            h = sp.read_ktahead(fn)
            names = ["nwave", "wave", "fwhm", "npress", "ntemp", "ng", "gasID", "isoID", "g_ord", "del_g", "presslevels", "templevels"]
            for n, v in zip(names, h):
                out[f"{tag}{gi}_head_{n}"] = np.asarray(v)
            wv = np.asarray(h[1])
            for rn, (lo, hi) in (("all", (0.0, 1e10)), ("sub", (float(wv[3]), float(wv[9])))):
                r = sp.read_ktable(fn, lo, hi)
                out[f"{tag}{gi}_{rn}_wave"] = np.asarray(r[3]); out[f"{tag}{gi}_{rn}_k"] = np.asarray(r[12])
                out[f"{tag}{gi}_{rn}_range"] = np.array([lo, hi])
            print(fn, os.path.getsize(fn), h[0], wv[:3])
    np.savez_compressed(os.path.join(OUT, "kta_read.npz"), **out)


if __name__ == "__main__":
    main()
