"""Where the wall time of one batched C3 Jacobian call goes (host phases + kernels), one GPU.
    python tools/jac_phases.py [--waves 10000] [--shard-of 8]   # --shard-of n: this GPU holds 1/n of the spectral axis"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waves", type=int, default=10000)
    ap.add_argument("--shard-of", type=int, default=1)
    args = ap.parse_args()
    import torch
    import archnemesis_dist_amd as pkg
    from archnemesis_dist_amd import synthetic as syn
    from archnemesis_dist_amd import jacobian as jac
    from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import torch_ktable
    dev = torch.device("cuda", 0)
    W, G, S, L, NP, NT = args.waves, 20, 8, 100, 20, 15
    Wl = jac.chunk_range(W, args.shard_of, 0)[1]
    stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
    eng = pkg.AnsfmEngine(0); eng.set_stream(stream.cuda_stream)
    _, delg = syn.gauss_legendre_01(G, as_float32=True)
    PRESS, TEMP, K = torch_ktable(torch, dev, Wl, G, NP, NT, S, seed=20260704)
    eng.upload_ktable(K, PRESS.astype(np.float32), TEMP.astype(np.float32), 200.0 + 0.1 * np.arange(Wl), delg.astype(np.float32))
    del K
    pr = syn.synth_profiles(100, S + 2, seed=11)
    st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 2)])
    model = BatchedCKThermalModel(eng, st, pr["RADIUS"], pr["ID"], pr["ISO"], list(range(2, S + 2)),
                                  layering_args=dict(NLAY=L, LAYINT=1, NINT=101), IRAY=4)
    model.global_waves = W
    ph = {}

    def timed(name, f):
        def g(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = f(*a, **k)
            torch.cuda.synchronize(); ph[name] = ph.get(name, 0.0) + time.perf_counter() - t0
            return r
        return g
    from archnemesis_dist_amd import layering
    # (every wrapper synchronises before and after: the phases add up to more than the untimed call, tools/jac_share.py)
    model.layers_dev = timed("layers in HBM (profiles up, layer_average_dev, gathers)", model.layers_dev)
    eng.layer_average_dev = timed("  of which eng.layer_average_dev", eng.layer_average_dev)
    eng.cirsrad_ck_thermal_ray_dev = timed("cirsrad (rows' Rayleigh continuum + merge + rt)", eng.cirsrad_ck_thermal_ray_dev)
    jac.finite_difference_jacobian_dev = timed("KK quotient + copy", jac.finite_difference_jacobian_dev)
    for it in range(4):
        ph.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        YN, KK = jac.jacobian_nemesis_batched(model)
        torch.cuda.synchronize(); tot = time.perf_counter() - t0
    k = eng.last_kernel_ms()
    print("W_local", Wl, "total %.1f ms" % (tot * 1e3), "rows", model.last_rows, "merge %.1f ms rt %.1f ms" % (k["overlap_ms"], k["rt_ms"]))
    for n, v in ph.items():
        print("  %-45s %.1f ms" % (n, v * 1e3))
    print("  %-45s %.1f ms" % ("unaccounted (uploads, torch, python)", (tot - sum(ph.values())) * 1e3))


if __name__ == "__main__":
    main()
