"""Batched numerical Jacobian of the scattering configuration at BASELINE configs[3] size (ansfm_cirsrad_ck_scatter_batch):
    python tools/c4_jacobian.py [--nx 20] [--waves 10000] [--check]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def c4_case(W, L, NMU, NF):
    x, w = np.polynomial.legendre.leggauss(NMU)
    MU, WT = 0.5 * (x + 1.0), 0.5 * w
    TH = np.linspace(0.0, 180.0, 41); c = np.cos(np.deg2rad(TH))
    leg = np.polynomial.legendre.legval(c, 0.6 ** np.arange(36) * (2 * np.arange(36) + 1)) / (4 * np.pi)
    ph = np.zeros((1, W, 2, TH.size)); ph[0, :, 0, :] = leg[None, :]; ph[0, :, 1, :] = c[None, :]
    ph = np.ascontiguousarray(ph[:, :, :, ::-1])
    wv = np.linspace(0, 1, W)[:, None]; lv = np.linspace(0, 1, L)[None, :]
    TAURAY = 1e-3 * np.exp(-5.0 * lv) * (1.0 + 0.3 * wv)
    TAUSCAT = 2e-2 * np.exp(-((lv - 0.35) / 0.1) ** 2) * (1.0 + 0.5 * np.sin(7.0 * wv))
    return MU, WT, ph, TAURAY, TAUSCAT, 1.1 * TAUSCAT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=20)
    ap.add_argument("--waves", type=int, default=10000)
    ap.add_argument("--check", action="store_true", help="compare every state with a call of its own (slow)")
    ap.add_argument("--forward", type=int, default=0, help="only time this many calls of ONE forward model (C4 size)")
    ap.add_argument("--nmu", type=int, default=16, help="zenith quadrature points (16: the matrix-core chain; the reference's default is 5)")
    ap.add_argument("--nf", type=int, default=8, help="Fourier orders - 1 (the reference's default is 2)")
    args = ap.parse_args()
    import torch
    import archnemesis_dist_amd as pkg
    from archnemesis_dist_amd import synthetic as syn
    from archnemesis_dist_amd.jacobian import perturbed_states
    from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel
    from bench import torch_ktable
    dev = torch.device("cuda", 0)
    W, G, S, L, NP, NT, NMU, NF = args.waves, 20, 8, 100, 20, 15, args.nmu, args.nf
    eng = pkg.AnsfmEngine(0)
    _, delg = syn.gauss_legendre_01(G, as_float32=True)
    PRESS, TEMP, K = torch_ktable(torch, dev, W, G, NP, NT, S, seed=20260704)
    WAVE = 200.0 + 0.1 * np.arange(W)
    eng.upload_ktable(K, PRESS.astype(np.float32), TEMP.astype(np.float32), WAVE, delg.astype(np.float32))
    del K
    npro = max(args.nx // 2, 2)
    pr = syn.synth_profiles(100, S + 2, seed=11)
    st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 2)])
    model = BatchedCKThermalModel(eng, st, pr["RADIUS"], pr["ID"], pr["ISO"], list(range(2, S + 2)), layering_args=dict(NLAY=L, LAYINT=1, NINT=101))
    cols = np.unique(np.concatenate([np.linspace(0, 99, npro).astype(int), 100 + np.linspace(0, 99, args.nx - npro).astype(int)]))
    X = perturbed_states(st.XN, 0.05 * st.XN)[:, np.concatenate([[0], cols + 1])].T
    lay = model.layers(X)
    n = X.shape[0]
    MU, WT, ph, TAURAY, TAUSCAT, TAUDUST = c4_case(W, L, NMU, NF)
    rep = lambda a: np.ascontiguousarray(np.broadcast_to(a[None], (n,) + a.shape))
    c1, c2 = 1.1911e-12, 1.439
    radg = np.stack([np.repeat((c1 * WAVE ** 3 / (np.exp(c2 * WAVE / lay["TEMP"][m, 0]) - 1.0))[:, None], NMU, 1) for m in range(n)])
    a = (0, lay["PRESS"], lay["TEMP"], lay["amount"], None, rep(TAUDUST), rep(TAURAY), rep(TAUSCAT), ph, rep(np.ones((W, 1, L))), radg,
         [30.0], [20.0], [45.0], np.full(W, 1e-8), 0, np.zeros((W, NMU, NMU, NF + 1)), MU, WT, NF, 101, 1, 1)
    if args.forward:
        ts = []
        for it in range(args.forward):
            t0 = time.perf_counter()
            one = eng.cirsrad_ck_scatter(0, lay["PRESS"][0], lay["TEMP"][0], lay["amount"][0], None, TAUDUST, TAURAY, TAUSCAT, ph,
                                         np.ones((W, 1, L)), radg[0], *a[11:])
            ts.append(time.perf_counter() - t0)
        print("one forward model, %d calls: min %.4f median %.4f s   checksum %.17g" % (len(ts), min(ts), float(np.median(ts)), float(one.sum())))
        return
    for it in range(2):
        t0 = time.perf_counter()
        spec = eng.cirsrad_ck_scatter_batch(*a)
        t = time.perf_counter() - t0
        print("n = %d forward models: %.2f s  (%.3f s per model; cache %s, gas rows %s)" % (n, t, t / n, eng.last_scatter_cache(), eng.last_layer_rows()))
    t0 = time.perf_counter()
    one = eng.cirsrad_ck_scatter(0, lay["PRESS"][0], lay["TEMP"][0], lay["amount"][0], None, TAUDUST, TAURAY, TAUSCAT, ph, np.ones((W, 1, L)),
                                 radg[0], *a[11:])
    print("one forward model on its own: %.2f s; equal to model 0 of the batch: %s" % (time.perf_counter() - t0, np.array_equal(one, spec[0])))
    if args.check:
        for m in range(1, n):
            o = eng.cirsrad_ck_scatter(0, lay["PRESS"][m], lay["TEMP"][m], lay["amount"][m], None, TAUDUST, TAURAY, TAUSCAT, ph,
                                       np.ones((W, 1, L)), radg[m], *a[11:])
            assert np.array_equal(o, spec[m]), m
        print("every state equals a call of its own, bit for bit")
    kk = (spec[1:, :, 0] - spec[0:1, :, 0])
    print("max |dY| / |Y| per column:", np.max(np.abs(kk) / np.abs(spec[0:1, :, 0]), axis=1)[:6])


if __name__ == "__main__":
    main()
