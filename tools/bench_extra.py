#!/usr/bin/env python
"""Secondary timings on one MI355X (not the driver's bench contract): analytic-gradient CIRSrad at C2,
the multiple-scattering core on a C4-like stack, runtime line-by-line on a reduced C5, batched layering.
Host-pointer entry points: PCIe staging is included in the wall times (noted per line)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import archnemesis_dist_amd as pkg
from archnemesis_dist_amd import synthetic as syn


def timeit(f, n=3):
    f(); ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return float(np.median(ts))


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None       # "grad" | "ms" | "lbl" | "layer"
    eng = pkg.AnsfmEngine(0)
    out = {}
    rng = np.random.default_rng(0)
    if only in (None, "grad"):
        bench_grad(eng, out)
    if only in (None, "ms"):
        bench_ms(eng, out, rng)
    if only in (None, "lbl"):
        bench_lbl(eng, out, rng)
    if only == "lbl_c5":
        bench_lbl(eng, out, rng, full=True)
    if only in (None, "layer"):
        bench_layer(eng, out)
    if only in (None, "maps"):
        bench_maps(eng, out, rng)
    if only in (None, "next"):
        bench_next(eng, out, rng)
    print(json.dumps(out, indent=1))


def bench_next(eng, out, rng):
    """the "next" rows: ILS convolution of an LBL spectrum with gradients, continuum opacities at C2, the
    k-table generator's binning -- host arrays in / out, beside the NumPy oracle where it finishes in seconds"""
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    # ILS: 1e6-point spectrum + 100 gradient columns, 1000 convolution points, Gaussian FWHM 0.4 cm-1 (~800 points a window)
    nw, nx, nc = 1000000, 100, 1000
    vw = 2000.0 + 1e-3 * np.arange(nw)
    y = rng.uniform(1, 2, nw); dy = rng.normal(size=(nw, nx))
    vc = np.linspace(2001.0, 2999.0, nc)
    t = timeit(lambda: eng.lblconvg(nw, vw, y, dy, nc, vc, 2, 0.4), 2)
    sub = slice(0, 20)
    t0 = time.perf_counter(); orc.lblconv(nw, vw, y, 20, vc[sub], 2, 0.4, dydx=dy); to = (time.perf_counter() - t0) * nc / 20
    out["lblconvg_1e6x100grad_1000conv"] = {"gpu_wall_s_host_arrays": t, "numpy_oracle_s_extrapolated_from_20_points": to}
    # continuum at C2: CIA table 2 pairs, Rayleigh (Jovian air), 2 aerosol populations
    W, L = 10000, 100
    wn = 200.0 + 0.1 * np.arange(W)
    TOTAM = 10.0 ** rng.uniform(24, 28, L)
    ID = np.array([39, 40, 6, 11]); ISO = np.zeros(4, int); VMR = np.tile([0.86, 0.13, 2e-3, 1e-4], (L, 1))
    tr = timeit(lambda: eng.calc_tau_rayleigh(4, 0, wn, TOTAM, ID, ISO, VMR))
    SW = np.linspace(150.0, 1300.0, 40); KE = 10.0 ** rng.uniform(-10, -8, (40, 2)); KS = KE * 0.6
    CONT = 10.0 ** rng.uniform(3, 8, (L, 2))
    td = timeit(lambda: eng.calc_tau_dust(wn, SW, KE, KS, CONT))
    t0 = time.perf_counter(); orc.calc_tau_dust(wn, SW, KE, KS, CONT); tdo = time.perf_counter() - t0
    out["continuum_C2"] = {"rayleigh_ls_gpu_s": tr, "dust_2pop_gpu_s": td, "dust_scipy_oracle_s": tdo}
    # k-table generator: 200 bins of ~2.5e4 line-by-line points (overlapping ILS windows), 20 g-ordinates
    n = 2000000
    w = np.linspace(1000.0, 1100.0, n)
    k = 10.0 ** (-24 + 3 * np.sin(w * 11.0) ** 2 + rng.normal(0, 0.3, n))
    cen = np.linspace(1001.0, 1099.0, 200); half = np.full(200, 0.625)
    x, _ = np.polynomial.legendre.leggauss(20); g = 0.5 * (x + 1)
    tk = timeit(lambda: eng.kdist_bins(w, k, cen - half, cen + half, g), 2)
    t0 = time.perf_counter(); orc.kdist_bins(w, k, cen - half, cen + half, g); tko = time.perf_counter() - t0
    out["kdist_200bins_2.5e4pts"] = {"gpu_wall_s_host_arrays": tk, "numpy_oracle_s": tko}


def bench_maps(eng, out, rng):
    # ---- nemesisfmg tail at C2: map2pro + map2xvec, host arrays in / out vs NumPy on the host ---------------
    W, NVMR, NDUST, Li, NPRO, NX = 10000, 8, 0, 100, 100, 200
    NPAR = NVMR + 2 + NDUST
    dS = rng.normal(size=(W, NPAR, Li, 1))
    LAYINC = np.arange(Li, dtype=np.int32)[::-1].copy()[:, None]
    DTE, DAM, DCO = (rng.uniform(0, 1, (Li, NPRO)) for _ in range(3))
    xmap = rng.normal(size=(NX, NPAR, NPRO))
    NLAYIN = np.array([Li], dtype=np.int32)

    def gpu():
        p = eng.map2pro(dS, W, NVMR, NDUST, NPRO, 1, NLAYIN, LAYINC, DTE, DAM, DCO)
        return eng.map2xvec(p, W, NVMR, NDUST, NPRO, 1, NX, xmap)

    def host():
        p = np.zeros((W, NPAR, NPRO, 1))
        last = None
        for par in range(NPAR):
            M = DAM if par < NVMR else (DTE if par == NVMR else (DCO if par <= NVMR + NDUST else None))
            if M is not None:
                last = np.tensordot(dS[:, par, :, 0], M[LAYINC[:, 0], :], axes=(1, 0))
            p[:, par, :, 0] = last          # para-H2 slot: the reference's stale dSPECOUT1
        return np.tensordot(p, xmap, axes=([1, 2], [1, 2]))
    tg = timeit(gpu)
    th = timeit(host, n=2)
    err = float(np.max(np.abs(gpu() - host())) / np.max(np.abs(host())))
    out["gradient_maps_C2"] = {"gpu_wall_s_host_ptr": tg, "numpy_host_wall_s": th, "max_rel_diff": err,
                               "note": "W=1e4, NPAR=10, Li=100, NPRO=100, NX=200; GPU time includes 80 MB H2D + 96 MB D2H"}


def bench_grad(eng, out):
    # ---- analytic Jacobian at C2 ------------------------------------------------------------------
    W, G, S, L, NP, NT = 10000, 20, 8, 100, 20, 15
    _, delg = syn.gauss_legendre_01(G, True)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S)
    WAVE = 200.0 + 0.1 * np.arange(W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg); del K
    atm = syn.synth_atmosphere(L, S)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
    EMTEMP = atm["lay_temp"][:, LAYINC[:, 0]][:, :, None]
    NVMR, NPAR = S, S + 2
    ig = np.arange(S, dtype=np.int32)
    f = lambda: eng.cirsradg_ck_thermal(0, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0], None, None, NVMR, NPAR,
                                        ig, NLAYIN, LAYINC, SCALE, EMTEMP[0], -1.0)
    t = timeit(f)
    k = eng.last_kernel_ms()
    out["cirsradg_C2"] = {"wall_s_host_ptr": t, "overlapg_kernel_ms": k["overlap_ms"], "rtg_kernel_ms": k["rt_ms"],
                          "note": "W=1e4,L=100,S=8,G=20, NPAR=10: SPECOUT + dSPECOUT(1e4,10,100,1) + dTSURF"}


def bench_ms(eng, out, rng):
    # ---- multiple scattering, C4-like -----------------------------------------------------------------
    Wm, Gm, Lm, M, NF, NC = 256, 20, 100, 16, 8, 1
    x, w = np.polynomial.legendre.leggauss(2 * M)
    mu1 = np.sort(np.abs(x[x > 0])); wt1 = w[x > 0][np.argsort(np.abs(x[x > 0]))]
    TH = np.linspace(0, 180, 41)
    ph = np.zeros((NC, Wm, 2, TH.size)); c = np.cos(np.deg2rad(TH))
    ph[:, :, 0, :] = ((1 - 0.36) / (1 + 0.36 - 1.2 * c) ** 1.5 / (4 * np.pi))[None, None, :]
    ph[:, :, 1, :] = c[None, None, :]
    ph = np.ascontiguousarray(ph[:, :, :, ::-1])
    taus = 10.0 ** rng.uniform(-3, 0.5, (Wm, Gm, Lm)); tauray = 10.0 ** rng.uniform(-6, -3, (Wm, Lm))
    tausc = 10.0 ** rng.uniform(-4, -1, (Wm, Lm)); taus = np.maximum(taus, (tausc + tauray)[:, None, :] * 1.05)
    om = np.broadcast_to((tausc + tauray)[:, None, :], taus.shape) / taus
    args = (ph, np.full((Wm, M), 1e-7), np.array([30.0]), np.array([20.0]), np.full(Wm, 1e-8), np.array([45.0]), 0,
            np.zeros((Wm, M, M, NF + 1)), mu1, wt1, NF, 500.0 + np.arange(Wm), np.full((Wm, Lm), 1e-7), taus, tauray, om, 101, 1, 1,
            np.ones((Wm, NC, Lm)))
    t = timeit(lambda: eng.scloud11wave_core(*args), n=2)
    nn = np.maximum((np.log2(taus) + 12).astype(int), 0)
    flops = float(((nn * 6.67 + 5) * 2 * M ** 3).sum() * (NF + 1))
    out["scloud11wave_C4like"] = {"wall_s": t, "waves": Wm, "g": Gm, "layers": Lm, "nmu": M, "nf": NF,
                                  "approx_flops": flops, "TFLOPs": flops / t / 1e12,
                                  "scaled_to_W1e4_s": t * 1e4 / Wm}


def bench_lbl(eng, out, rng, full=False):
    # ---- runtime LBL: reduced C5 (default) or the full C5 grid (1e6 wavenumbers x 50 layers, 1e5 lines) ---------
    nw, N, Ll = (1000000, 100000, 50) if full else (200000, 20000, 5)
    wn = 2000.0 + 1e-3 * np.arange(nw)
    span = nw * 1e-3
    nu = np.sort(rng.uniform(2000.0 - 75.0, 2000.0 + span + 75.0, N)); sw = 10.0 ** rng.uniform(-28, -19, N); el = rng.uniform(0, 3000, N)
    bp = np.zeros((3, N)); bp[0] = rng.uniform(0.02, 0.1, N); bp[1] = rng.uniform(0.5, 0.8, N); bp[2] = rng.uniform(-0.01, 0.01, N)
    c2 = 2.99792458E10 * 6.62607015E-27 / 1.380649E-16
    sr = 1 - np.exp(-c2 * nu / 296.0)
    tt = np.linspace(150, 300, Ll); pp = np.logspace(-4, 0, Ll); qq = np.ones(Ll)
    o = np.zeros((Ll, nw))
    t = timeit(lambda: eng.add_line_set_monochromatic_absorption(wn, 0, tt, 296.0, pp, 1.0, qq, 1.0, 28.0, np.array([1.0]), bp, nu,
                                                                 sw, el, sr, o), n=2)
    evals = float(N) * (150.0 / 1e-3) * Ll * (span / (span + 150.0))     # lines whose window overlaps the grid, roughly
    out["lbl_runtime_C5" if full else "lbl_runtime_reducedC5"] = {"wall_s": t, "grid": nw, "lines": N, "layers": Ll, "approx_profile_evals": evals,
                                    "Gevals_per_s": evals / t / 1e9}


def bench_layer(eng, out):
    # ---- batched layering ---------------------------------------------------------------------------------------
    n, NPRO, V, D, NL = 201, 120, 8, 1, 100
    H = np.linspace(0, 6e5, NPRO); P = 1e6 * np.exp(-H / 3e4); T = 150 + 50 * np.sin(H / 1e5)
    rep = lambda a: np.repeat(np.asarray(a)[None], n, 0)
    VM = np.full((NPRO, V), 1e-4); DU = np.full((NPRO, D), 10.0)
    BH = np.linspace(0, 5.9e5, NL)
    t = timeit(lambda: eng.layer_average(7.1e7, rep(H), rep(P), rep(T), None, rep(VM), rep(DU), None, BH, None, LAYINT=1, NINT=101))
    out["layer_average_batch"] = {"wall_s": t, "states": n, "layers": NL, "nint": 101, "states_layers_per_s": n * NL / t}


if __name__ == "__main__":
    main()
