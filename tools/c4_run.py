#!/usr/bin/env python
"""One full-size C4 call (CIRSrad scattering branch, 1e4 nu x 20 g x 100 layers, 16 streams, NF 8) for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -- python3 tools/c4_run.py [W]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import archnemesis_dist_amd as pkg
from archnemesis_dist_amd import synthetic as syn

W = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
G, S, L, NP, NT, NMU, NF = 20, 8, 100, 8, 6, 16, 8
eng = pkg.AnsfmEngine(0)
PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=3)
_, delg = syn.gauss_legendre_01(G)
WAVE = 200.0 + 0.1 * np.arange(W)
eng.upload_ktable(K, PRESS, TEMP, WAVE, delg); del K
atm = syn.synth_atmosphere(L, S, seed=7)
x, w = np.polynomial.legendre.leggauss(NMU)
MU, WT = 0.5 * (x + 1.0), 0.5 * w
TH = np.linspace(0.0, 180.0, 41); c = np.cos(np.deg2rad(TH))
leg = np.polynomial.legendre.legval(c, 0.6 ** np.arange(36) * (2 * np.arange(36) + 1)) / (4 * np.pi)
ph = np.zeros((1, W, 2, TH.size)); ph[0, :, 0, :] = leg[None, :]; ph[0, :, 1, :] = c[None, :]
ph = np.ascontiguousarray(ph[:, :, :, ::-1])
lay_p, lay_t, am = atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0]
wv = np.linspace(0, 1, W)[:, None]; lv = np.linspace(0, 1, L)[None, :]
TAURAY = 1e-3 * np.exp(-5.0 * lv) * (1.0 + 0.3 * wv)
TAUSCAT = 2e-2 * np.exp(-((lv - 0.35) / 0.1) ** 2) * (1.0 + 0.5 * np.sin(7.0 * wv))
radg = np.repeat((1.1911e-12 * WAVE ** 3 / (np.exp(1.439 * WAVE / lay_t[0]) - 1.0))[:, None], NMU, 1)
f = lambda: eng.cirsrad_ck_scatter(0, lay_p, lay_t, am, None, 1.1 * TAUSCAT, TAURAY, TAUSCAT, ph, np.ones((W, 1, L)), radg, [30.0],
                                   [20.0], [45.0], np.full(W, 1e-8), 0, np.zeros((W, NMU, NMU, NF + 1)), MU, WT, NF, 101, 1, 1)
f()
t0 = time.perf_counter(); out = f(); print("wall_s", time.perf_counter() - t0, "W", W, float(out.mean()))
