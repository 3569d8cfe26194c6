"""Wall time of the analytic C3 Jacobian (BatchedCKThermalModel.jacobian_analytic) -- for a kernel trace:
   rocprofv3 --kernel-trace --stats -- python3 tools/jac_analytic_share.py [W]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import archnemesis_dist_amd as pkg
from archnemesis_dist_amd import synthetic as syn
from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel
from bench import torch_ktable
dev = torch.device("cuda", 0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
G, S, L, NP, NT = 20, 8, 100, 20, 15
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
eng = pkg.AnsfmEngine(0); eng.set_stream(stream.cuda_stream)
_, delg = syn.gauss_legendre_01(G, as_float32=True)
PRESS, TEMP, K = torch_ktable(torch, dev, W, G, NP, NT, S, seed=20260704)
eng.upload_ktable(K, PRESS.astype(np.float32), TEMP.astype(np.float32), 200.0 + 0.1 * np.arange(W), delg.astype(np.float32))
del K
pr = syn.synth_profiles(100, S + 2, seed=11)
st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 2)])
model = BatchedCKThermalModel(eng, st, pr["RADIUS"], pr["ID"], pr["ISO"], list(range(2, S + 2)),
                              layering_args=dict(NLAY=L, LAYINT=1, NINT=101), IRAY=4)
ts = []
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    YN, KK = model.jacobian_analytic()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("W", W, "analytic jacobian ms:", " ".join("%.2f" % t for t in ts))
