#!/usr/bin/env python
"""Repeat the C3 batched Jacobian (bench.py's configuration) and print each call's wall time and torch's reserved memory."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
import archnemesis_dist_amd as pkg
from archnemesis_dist_amd import synthetic as syn
from archnemesis_dist_amd.jacobian import jacobian_nemesis_batched
from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel

dev = torch.device("cuda:0")
W, G, S, L, NP, NT = 10000, 20, 8, 100, 20, 15
eng = pkg.AnsfmEngine(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
eng.set_stream(stream.cuda_stream)
PRESS, TEMP, K = B.torch_ktable(torch, dev, W, G, NP, NT, S, seed=20260704)
_, delg = syn.gauss_legendre_01(G, as_float32=True)
WAVE = 200.0 + 0.1 * np.arange(W)
eng.upload_ktable(K, PRESS.astype(np.float32), TEMP.astype(np.float32), WAVE, delg.astype(np.float32)); del K
pr = syn.synth_profiles(100, S + 2, seed=11)
st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 2)])
model = BatchedCKThermalModel(eng, st, pr["RADIUS"], pr["ID"], pr["ISO"], list(range(2, S + 2)),
                              layering_args=dict(NLAY=L, LAYINT=1, NINT=101), IRAY=4)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    YN, KK = jacobian_nemesis_batched(model, rank=0, world_size=1)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"call {i}: {t*1e3:7.1f} ms  rows {model.last_rows}  reserved {torch.cuda.memory_reserved()/2**30:6.2f} GiB  kernels {eng.last_kernel_ms()}")

# where the host time of a call goes
import archnemesis_dist_amd.profile_state as ps
_orig_layers = BatchedCKThermalModel.layers
def timed_layers(self, X):
    t0 = time.perf_counter(); r = _orig_layers(self, X); timed_layers.t = time.perf_counter() - t0; return r
BatchedCKThermalModel.layers = timed_layers
_orig_sb = BatchedCKThermalModel.spectra_batch
def timed_sb(self, X, device=None):
    t0 = time.perf_counter(); r = _orig_sb(self, X, device); torch.cuda.synchronize(); timed_sb.t = time.perf_counter() - t0; return r
BatchedCKThermalModel.spectra_batch = timed_sb
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    YN, KK = jacobian_nemesis_batched(model, rank=0, world_size=1)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"split {i}: total {t*1e3:6.1f}  layers(host) {timed_layers.t*1e3:6.1f}  spectra_batch {timed_sb.t*1e3:6.1f}  rest {1e3*(t-timed_sb.t):6.1f} ms")
