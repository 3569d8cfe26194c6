"""PCIe-inclusive rate of the host-pointer entry point: one C2 forward model per call, NumPy arrays in, spectrum out
(bench.py's `value` keeps its inputs resident in HBM; this is the number a ctypes caller sees)."""
import os, time, numpy as np, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import archnemesis_dist_amd as pkg
from archnemesis_dist_amd import synthetic as syn
W, G, S, L, NP, NT = 10000, 20, 8, 100, 20, 15
_, delg = syn.gauss_legendre_01(G, True)
PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S)
eng = pkg.AnsfmEngine(0)
eng.upload_ktable(K, PRESS, TEMP, 200.0 + 0.1 * np.arange(W), delg); del K
atm = syn.synth_atmosphere(L, S)
NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
EMTEMP = atm["lay_temp"][0][LAYINC[:, 0]][:, None]
cont = syn.synth_continuum(W, L)[0]
f = lambda c: eng.cirsrad_ck_thermal(0, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0], c, NLAYIN, LAYINC, SCALE, EMTEMP, -1.0)
for c, tag in ((cont, "with 8 MB continuum"), (None, "no continuum")):
    f(c); ts = []
    for _ in range(10):
        t = time.perf_counter(); f(c); ts.append(time.perf_counter() - t)
    print(tag, "median ms:", 1e3 * float(np.median(ts)), "->", 1.0 / float(np.median(ts)), "fm/s")
