#!/usr/bin/env python
"""Wall time of the array-level k_overlap / k_overlapg seams on sorted and on unsorted k-distributions (the generic
kernels: per-lane sort + permutation bytes).  python tools/calib/generic_merge_time.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import archnemesis_dist_amd as pkg
from archnemesis_dist_amd import synthetic as syn

eng = pkg.AnsfmEngine(0)
rng = np.random.default_rng(3)
W, G, L, S = 8192, 20, 24, 6
_, delg = syn.gauss_legendre_01(G)
k = np.sort(10.0 ** rng.uniform(-26, -20, (W, G, L, S)), axis=1)
dk = k * rng.uniform(-0.01, 0.01, k.shape)
am = 10.0 ** rng.uniform(20, 23, (S, L))
ku = k.copy(); ku[:, [3, 4]] = ku[:, [4, 3]]          # one inversion per distribution
for name, kk in (("sorted", k), ("unsorted", ku)):
    for fn, f in (("k_overlap", lambda: eng.k_overlap(delg, kk, am)), ("k_overlapg", lambda: eng.k_overlapg(delg, kk, dk, am))):
        f(); ts = []
        for _ in range(3):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        print(f"{name:9s} {fn:11s} wall {np.median(ts)*1e3:8.2f} ms (host arrays in / out)")
