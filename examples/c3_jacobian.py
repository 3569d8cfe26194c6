#!/usr/bin/env python
"""Jacobian of a correlated-k thermal-emission forward model with respect to a model-0 state vector (temperature and the
ln mixing ratio of one gas at every level), two ways -- no reference needed, synthetic table and atmosphere:

  * numerically, like ForwardModel_0.jacobian_nemesis: NX + 1 forward models in ONE batched call
    (jacobian_nemesis_batched; layers that equal the unperturbed state's are not recomputed);
  * analytically, like nemesisfmg: layer_averageg -> CIRSrad(return_grad) -> map2pro -> map2xvec, the gradient merge
    restricted to the gas the state vector names.

    python examples/c3_jacobian.py [NWAVE]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import archnemesis_dist_amd as pkg
from archnemesis_dist_amd import synthetic as syn
from archnemesis_dist_amd.jacobian import jacobian_nemesis_batched
from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
G, S, L, NPRO = 20, 6, 60, 60
eng = pkg.AnsfmEngine(0)
PRESS, TEMP, K = syn.synth_ktable(W, G, 20, 15, S, seed=1)
_, delg = syn.gauss_legendre_01(G)
eng.upload_ktable(K, PRESS, TEMP, 200.0 + 0.1 * np.arange(W), delg); del K
pr = syn.synth_profiles(NPRO, S + 2, seed=2)
state = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 3)])
model = BatchedCKThermalModel(eng, state, pr["RADIUS"], pr["ID"], pr["ISO"], list(range(2, S + 2)),
                              layering_args=dict(NLAY=L, LAYINT=1, NINT=101), IRAY=4)

jacobian_nemesis_batched(model)                                   # warm-up: buffers
t0 = time.perf_counter(); YN, KK = jacobian_nemesis_batched(model); t_fd = time.perf_counter() - t0
rows, total = model.last_rows
model.jacobian_analytic()
t0 = time.perf_counter(); YA, KA = jacobian_nemesis_batched(model, analytical_gradient=True); t_an = time.perf_counter() - t0
print(f"state vector NX = {state.NX}, spectrum NY = {YN.size}")
print(f"finite differences (5 % steps, {state.NX + 1} forward models, {rows} of {total} layer opacities computed): {t_fd*1e3:7.1f} ms")
print(f"analytic gradients (one forward model):                                                  {t_an*1e3:7.1f} ms")
print("spectra agree:", bool(np.allclose(YN, YA, rtol=1e-12)))
# The two Jacobians are different objects: the reference's numerical step is 5 % of each element -- for ln(VMR) ~ -10 that is a
# factor 1.6 in the mixing ratio, not a derivative -- and the analytic temperature columns hold the layer amounts fixed.
# tests/test_jacobian_c3.py checks the analytic route against small-step central differences.
