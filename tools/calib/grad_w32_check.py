import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import archnemesis_dist_amd as pkg
from archnemesis_dist_amd import synthetic as syn
eng = pkg.AnsfmEngine(0)
W, G, S, L, NP, NT = 10000, 20, 8, 100, 20, 15
PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S)
WAVE = 200.0 + 0.1 * np.arange(W)
atm = syn.synth_atmosphere(L, S)
NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
EMTEMP = atm["lay_temp"][:, LAYINC[:, 0]][:, :, None]
ig = np.arange(S, dtype=np.int32)
for f32 in (False, True):
    _, delg = syn.gauss_legendre_01(G, True)
    if f32: delg = delg.astype(np.float32)
    eng.upload_ktable(K, PRESS.astype(np.float32) if f32 else PRESS, TEMP.astype(np.float32) if f32 else TEMP, WAVE, delg)
    f = lambda: eng.cirsradg_ck_thermal(0, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0], None, None, S, S + 2,
                                        ig, NLAYIN, LAYINC, SCALE, EMTEMP[0], -1.0)
    f(); ts=[]
    for _ in range(3):
        t=time.perf_counter(); f(); ts.append(time.perf_counter()-t)
    print("f32" if f32 else "f64", np.median(ts), eng.last_kernel_ms())
