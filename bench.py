#!/usr/bin/env python
"""bench.py -- forward-models/s of the archNEMESIS correlated-k hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): synthetic 10 000-wavenumber x 100-layer x
8-gas correlated-k nadir thermal-emission forward model, G=20, table 20 P x 15 T.  One "step" =
one forward model per rank through the fused CIRSrad path (ck_overlap + thermal_rt kernels) with
every input already resident in HBM.  N>1: one process per GPU, independent forward models per
rank (the jacobian_nemesis fan-out), no data-path collective in the timed steps ("weak").

After the timed steps rank 0 also reports (extra keys, outside the timed region):
  jacobian: wall time of a 201-forward-model numerical Jacobian (C3) sharded over the N ranks with
            one all_gather (RCCL) of the spectra.
  cpu_baseline: the CPU oracle ("port") on a bounded sample of the same workload.
  roofline: dominant kernel (ck_overlap) algorithmic bytes / hipEvent-measured duration vs HBM peak.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(W, G, S, L, NP, NT, P, n_models=1, V=None):
    """B_fm = 4*W*G*S*min(NP*NT,4L) + 8*(S+V+6)*L + 8*3*W*L + 8*W*P ;
    batch: table once + n * per-model terms."""
    V = S if V is None else V
    table = 4.0 * W * G * S * min(NP * NT, 4 * L)
    per_model = 8.0 * (S + V + 6) * L + 8.0 * 3 * W * L + 8.0 * W * P
    return table + n_models * per_model


def torch_ktable(torch, dev, W, G, NP, NT, S, seed):
    """Device-side twin of synthetic.synth_ktable (same family, torch RNG): K (W,G,NP,NT,S) f64."""
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    f8 = torch.float64
    PRESS = torch.logspace(-7, 1.3, NP, dtype=f8, device=dev)
    TEMP = torch.linspace(50.0, 500.0, NT, dtype=f8, device=dev)
    K = torch.empty((W, G, NP, NT, S), dtype=f8, device=dev)
    chunk = 1000
    for w0 in range(0, W, chunk):
        n = min(chunk, W - w0)
        u = lambda lo, hi, shape: lo + (hi - lo) * torch.rand(shape, generator=gen, dtype=f8, device=dev)
        base = 10.0 ** u(-28, -21, (n, 1, 1, 1, S))
        gshape = torch.sort(10.0 ** u(-2, 3, (n, G, 1, 1, S)), dim=1).values
        pexp = u(0.0, 0.3, (n, 1, 1, 1, S)); texp = u(-1.0, 2.0, (n, 1, 1, 1, S))
        K[w0:w0 + n] = base * gshape * PRESS.view(1, 1, NP, 1, 1) ** pexp * (TEMP.view(1, 1, 1, NT, 1) / 200.0) ** texp
    return PRESS.cpu().numpy(), TEMP.cpu().numpy(), K


def run_extras(torch, dev, eng, syn, atm, NLAYIN, LAYINC, SCALE, EMTEMP, WAVE, delg, do_cpu, K_sample=None, PRESS=None, TEMP=None, jac_model=None):
    """Numbers for the other BASELINE configs and the north-star variant, OUTSIDE the timed region of the headline
    metric (rank 0, one GPU).  Each entry says what it timed; profiles/README.md names the rocprofv3 run it can be
    reproduced from."""
    ex = {}
    W, G, S, L, NP, NT = 10000, 20, 8, 100, 20, 15
    rng = np.random.default_rng(5)

    def med(f, n=3):
        """median wall time of n calls after one warm-up call"""
        f()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    # ---- analytic-gradient CIRSrad at C2 (nemesisfmg's RT call): SPECOUT + dSPECOUT (1e4, 10, 100, 1) + dTSURF -------
    NVMR, NPAR = S, S + 2
    ig = np.arange(S, dtype=np.int32)
    fg = lambda: eng.cirsradg_ck_thermal(0, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0], None, None, NVMR, NPAR,
                                         ig, NLAYIN, LAYINC, SCALE, EMTEMP[0], -1.0)
    t = med(fg)
    k = eng.last_kernel_ms()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # the engine runs on torch's stream here
    e0.record(); fg(); e1.record(); torch.cuda.synchronize()
    k2 = eng.last_kernel_ms()
    eng.set_gradient_gases([0])                       # what an analytic Jacobian for T + one gas needs (ansfm_set_gradient_gases)
    try:
        fg(); fg()
        k3 = eng.last_kernel_ms()
        eng.set_gradient_gases([0], temperature=False)    # ... and for one gas at fixed temperature
        fg(); fg()
        k4 = eng.last_kernel_ms()
    finally:
        eng.set_gradient_gases(None)
    ex["cirsradg_c2"] = {"gpu_ms_whole_call": e0.elapsed_time(e1), "overlapg_kernel_ms_second_reading": k2["overlap_ms"],"what": "CIRSrad(return_grad=True) at C2: k_ck_overlapg + k_thermal_rtg, host arrays in / out "
                                 "(80 MB of dSPECOUT cross PCIe inside wall_s)", "wall_s": t,
                         "overlapg_kernel_ms": k["overlap_ms"], "rtg_kernel_ms": k["rt_ms"],
                         "overlapg_kernel_ms_one_gas_selected": k3["overlap_ms"],
                         "overlapg_kernel_ms_one_gas_no_temperature": k4["overlap_ms"]}

    # ---- C4: CIRSrad scattering branch at full size on the C2 table: 1e4 nu x G 20 x 100 layers, 16 streams, NF = 8 ---
    NMU, NF = 16, 8
    x, w = np.polynomial.legendre.leggauss(NMU)
    MU, WT = 0.5 * (x + 1.0), 0.5 * w
    TH = np.linspace(0.0, 180.0, 41); c = np.cos(np.deg2rad(TH))
    leg = np.polynomial.legendre.legval(c, 0.6 ** np.arange(36) * (2 * np.arange(36) + 1)) / (4 * np.pi)   # 36 phase moments (HG g = 0.6)
    ph = np.zeros((1, W, 2, TH.size)); ph[0, :, 0, :] = leg[None, :]; ph[0, :, 1, :] = c[None, :]
    ph = np.ascontiguousarray(ph[:, :, :, ::-1])
    lay_p, lay_t, am = atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0]
    wv = np.linspace(0, 1, W)[:, None]; lv = np.linspace(0, 1, L)[None, :]
    TAURAY = 1e-3 * np.exp(-5.0 * lv) * (1.0 + 0.3 * wv)
    TAUSCAT = 2e-2 * np.exp(-((lv - 0.35) / 0.1) ** 2) * (1.0 + 0.5 * np.sin(7.0 * wv))
    TAUDUST = 1.1 * TAUSCAT
    c1, c2 = 1.1911e-12, 1.439
    radg = np.repeat((c1 * WAVE ** 3 / (np.exp(c2 * WAVE / lay_t[0]) - 1.0))[:, None], NMU, 1)
    fs = lambda: eng.cirsrad_ck_scatter(0, lay_p, lay_t, am, None, TAUDUST, TAURAY, TAUSCAT, ph, np.ones((W, 1, L)), radg,
                                        [30.0], [20.0], [45.0], np.full(W, 1e-8), 0, np.zeros((W, NMU, NMU, NF + 1)), MU, WT,
                                        NF, 101, 1, 1)
    t = med(fs, 2)
    ex["c4_scatter"] = {"what": "CIRSrad, multiple-scattering branch (ansfm_cirsrad_ck_scatter) at BASELINE configs[3] size: 1e4 "
                                "wavenumbers x 20 g x 100 layers, 16 streams, 36 phase moments, NF = 8, Rayleigh on, 1 limb-free path; "
                                "gas opacities + TAUTOT/OMEGA/BB formed on the device", "wall_s": t, "ms_per_1000_nu": t * 1e2,
                        "chains_per_s": W * G * (NF + 1) / t}
    # parity samples of the SAME full-size configuration (checker leg, outside any timing):
    # (i) the non-scattering limit over all 1e4 wavenumbers: without scatterers the doubling / adding branch must return
    #     the plane-parallel thermal emission of the same gas opacities under the quadrature angles (k_thermal_rt, SCALE = 1/mu)
    pick = np.array([0, 5, 10, 15])
    emi = np.rad2deg(np.arccos(MU[pick]))
    ns = eng.cirsrad_ck_scatter(0, lay_p, lay_t, am, None, None, None, None, None, None, radg, np.full(pick.size, 40.0), emi,
                                np.zeros(pick.size), np.zeros(W), 0, np.zeros((W, NMU, NMU, NF + 1)), MU, WT, NF, 101, 0, 0)
    nl = np.full(pick.size, L, dtype=np.int32)
    li = np.repeat(np.arange(L - 1, -1, -1, dtype=np.int32)[:, None], pick.size, 1)
    th = eng.cirsrad_ck_thermal(0, lay_p, lay_t, am, None, nl, li, np.repeat((1.0 / MU[pick])[None, :], L, 0),
                                np.repeat(lay_t[::-1][:, None], pick.size, 1), -1.0)
    ex["c4_scatter"]["non_scattering_limit_max_rel_err_vs_thermal_rt"] = float(np.max(np.abs(ns - th) / th))
    # (ii) the scattering configuration itself against the oracle's restatement of scloud11wave_core: the first four
    #     wavenumbers at the first g-ordinate -- the Hansen renormalisation carries its factors from one (g, wavenumber) to
    #     the next in loop order (g outer), so only a leading run of the sequence can be replayed without the whole call
    if do_cpu and K_sample is not None:
        from oracle import oracle as orc
        nq = 4
        t0 = time.perf_counter()
        _, sg = eng.cirsrad_ck_scatter(0, lay_p, lay_t, am, None, TAUDUST, TAURAY, TAUSCAT, ph, np.ones((W, 1, L)), radg, [30.0],
                                       [20.0], [45.0], np.full(W, 1e-8), 0, np.zeros((W, NMU, NMU, NF + 1)), MU, WT, NF, 101, 1, 1,
                                       return_spec_g=True)
        kk = orc.calc_k(K_sample[:nq], PRESS, TEMP, lay_p / 101325.0, lay_t)
        tg = orc.k_overlap(delg, kk, am)
        tautot = tg + TAUDUST[:nq, None, :] + TAURAY[:nq, None, :]
        omega = np.where(tautot > 0, (TAURAY + TAUSCAT)[:nq, None, :] / np.where(tautot > 0, tautot, 1.0), 0.0)
        bnu = c1 * WAVE[:nq, None] ** 3 / (np.exp(c2 * WAVE[:nq, None] / lay_t[None, :]) - 1.0)
        rad = orc.scloud11wave_core(ph[:, :nq], radg[:nq], [30.0], [20.0], np.full(nq, 1e-8), [45.0], 0, np.zeros((nq, NMU, NMU, NF + 1)),
                                    MU, WT, NF, WAVE[:nq], bnu, np.ascontiguousarray(tautot[:, 0:1, :]), TAURAY[:nq],
                                    np.ascontiguousarray(omega[:, 0:1, :]), 101, 1, 1, np.ones((nq, 1, L)))
        ref = rad[0, 0, :]
        ex["c4_scatter"]["max_rel_err_vs_oracle"] = float(np.max(np.abs(sg[:nq, 0, 0] - ref) / np.abs(ref)))
        ex["c4_scatter"]["oracle_sample"] = (f"{nq} (wavenumber, g) chains (first {nq} wavenumbers, first g-ordinate: 16 streams x {L} "
                                             f"layers x {NF + 1} orders each) through oracle/ansfm_oracle_ms.c, {time.perf_counter() - t0:.1f} s")

    # ---- the same at the reference's DEFAULT quadrature (Scatter_0.py:59: NMU = 5, NF = 2): the stream-count-generic chain
    #      (LDS matrices, one block per (wavenumber, g, order)), not the matrix-core one
    n5, f5 = 5, 2
    x5, w5 = np.polynomial.legendre.leggauss(n5)
    MU5, WT5 = 0.5 * (x5 + 1.0), 0.5 * w5
    radg5 = np.ascontiguousarray(radg[:, :n5])
    fs5 = lambda **kw: eng.cirsrad_ck_scatter(0, lay_p, lay_t, am, None, TAUDUST, TAURAY, TAUSCAT, ph, np.ones((W, 1, L)), radg5,
                                              [30.0], [20.0], [45.0], np.full(W, 1e-8), 0, np.zeros((W, n5, n5, f5 + 1)), MU5, WT5,
                                              f5, 101, 1, 1, **kw)
    t5 = med(fs5, 2)
    ex["c4_scatter_default_streams"] = {"what": "the C4 configuration with the reference's default quadrature (5 streams, NF = 2) "
                                                "instead of 16 streams / NF = 8", "wall_s": t5, "chains_per_s": W * G * (f5 + 1) / t5}
    if do_cpu and K_sample is not None:
        from oracle import oracle as orc
        nq = 4
        t0 = time.perf_counter()
        _, sg5 = fs5(return_spec_g=True)
        kk = orc.calc_k(K_sample[:nq], PRESS, TEMP, lay_p / 101325.0, lay_t)
        tg = orc.k_overlap(delg, kk, am)
        tautot = tg + TAUDUST[:nq, None, :] + TAURAY[:nq, None, :]
        omega = np.where(tautot > 0, (TAURAY + TAUSCAT)[:nq, None, :] / np.where(tautot > 0, tautot, 1.0), 0.0)
        bnu = c1 * WAVE[:nq, None] ** 3 / (np.exp(c2 * WAVE[:nq, None] / lay_t[None, :]) - 1.0)
        rad5 = orc.scloud11wave_core(ph[:, :nq], radg5[:nq], [30.0], [20.0], np.full(nq, 1e-8), [45.0], 0, np.zeros((nq, n5, n5, f5 + 1)),
                                     MU5, WT5, f5, WAVE[:nq], bnu, np.ascontiguousarray(tautot[:, 0:1, :]), TAURAY[:nq],
                                     np.ascontiguousarray(omega[:, 0:1, :]), 101, 1, 1, np.ones((nq, 1, L)))
        ref5 = rad5[0, 0, :]
        ex["c4_scatter_default_streams"]["max_rel_err_vs_oracle"] = float(np.max(np.abs(sg5[:nq, 0, 0] - ref5) / np.abs(ref5)))
        ex["c4_scatter_default_streams"]["oracle_sample"] = f"{nq} (wavenumber, g) chains as for c4_scatter, {time.perf_counter() - t0:.1f} s"

    # ---- numerical Jacobian of the scattering configuration at C4 size: jacobian_nemesis forces NX + 1 multiple-scattering
    #      forward models when ISCAT != THERMAL_EMISSION (ForwardModel_0.py:2251-2252) -- its most expensive case.  The state
    #      vector of the C3 row (T and ln VMR of one absorber at 100 levels through Curtis-Godson layer_average), 201 forward
    #      models in ONE ansfm_cirsrad_ck_scatter_batch call: model 0's doubled layers are cached per (wavenumber, g, order,
    #      layer), the others re-run the adding sweep and recompute only the layers their level perturbation changed.
    if jac_model is not None:
        from archnemesis_dist_amd.jacobian import perturbed_states
        stj = jac_model.state
        Xj = perturbed_states(stj.XN, 0.05 * stj.XN).T
        layj = jac_model.layers(Xj)
        nj = Xj.shape[0]
        rep = lambda a: np.ascontiguousarray(np.broadcast_to(a[None], (nj,) + a.shape))
        radg_j = np.stack([np.repeat((c1 * WAVE ** 3 / (np.exp(c2 * WAVE / layj["TEMP"][m, 0]) - 1.0))[:, None], NMU, 1) for m in range(nj)])
        tail = ([30.0], [20.0], [45.0], np.full(W, 1e-8), 0, np.zeros((W, NMU, NMU, NF + 1)), MU, WT, NF, 101, 1, 1)
        aj = (0, layj["PRESS"], layj["TEMP"], layj["amount"], None, rep(TAUDUST), rep(TAURAY), rep(TAUSCAT), ph, rep(np.ones((W, 1, L))), radg_j)
        t0 = time.perf_counter()
        spec_j = eng.cirsrad_ck_scatter_batch(*aj, *tail)
        tj_first = time.perf_counter() - t0          # includes the one-off allocation of the layer cache (tens of GB)
        first = spec_j.copy()
        t0 = time.perf_counter()
        spec_j = eng.cirsrad_ck_scatter_batch(*aj, *tail)
        tj = time.perf_counter() - t0                # as the second and later iterations of a retrieval see it
        repeatable = bool(np.array_equal(first, spec_j))
        del first
        hits, tot = eng.last_scatter_cache()
        rows_g = eng.last_layer_rows()
        fs1 = lambda m: eng.cirsrad_ck_scatter(0, layj["PRESS"][m], layj["TEMP"][m], layj["amount"][m], None, TAUDUST, TAURAY, TAUSCAT, ph,
                                               np.ones((W, 1, L)), radg_j[m], *tail)
        t0 = time.perf_counter(); one0 = fs1(0); t_one = time.perf_counter() - t0
        pick = [0, 37, 101, 163, nj - 1]
        same = all(np.array_equal(fs1(m), spec_j[m]) for m in pick)
        xn1 = stj.XN * 1.05
        KKj = ((spec_j[1:, :, 0] - spec_j[0:1, :, 0]) / (xn1 - stj.XN)[:, None]).T
        ex["c4_jacobian"] = {
            "what": "numerical Jacobian of the C4 scattering configuration (1e4 wavenumbers x 20 g x 100 layers, 16 streams, NF = 8): "
                    "%d multiple-scattering forward models (T and ln VMR of one absorber at 100 levels) in one "
                    "ansfm_cirsrad_ck_scatter_batch call, host arrays in / out" % nj,
            "forward_models": nj, "kk_shape": list(KKj.shape), "wall_s": tj, "wall_s_first_call": tj_first,
            "second_call_bit_identical": repeatable, "s_per_forward_model": tj / nj,
            "one_forward_model_on_its_own_s": t_one, "speedup_vs_separate_calls": t_one * nj / tj,
            "layers_from_cache": int(hits), "layers_of_models_1_to_n": int(tot), "layers_doubled": int(tot - hits + L),
            "layers_total": int(nj * L), "gas_opacity_rows_computed": int(rows_g[0]), "gas_opacity_rows_all": int(rows_g[1]),
            "bit_identical_to_separate_calls": bool(same), "models_compared": pick}
        # ... and at the reference's default quadrature (5 streams, NF = 2): the lane-per-chain kernel with its own layer cache
        radg5_j = np.ascontiguousarray(radg_j[:, :, :n5])
        tail5 = ([30.0], [20.0], [45.0], np.full(W, 1e-8), 0, np.zeros((W, n5, n5, f5 + 1)), MU5, WT5, f5, 101, 1, 1)
        aj5 = aj[:10] + (radg5_j,)
        t0 = time.perf_counter(); spec5 = eng.cirsrad_ck_scatter_batch(*aj5, *tail5); t5_first = time.perf_counter() - t0
        t0 = time.perf_counter(); spec5 = eng.cirsrad_ck_scatter_batch(*aj5, *tail5); t5j = time.perf_counter() - t0
        hits5, tot5 = eng.last_scatter_cache()
        fs5m = lambda m: eng.cirsrad_ck_scatter(0, layj["PRESS"][m], layj["TEMP"][m], layj["amount"][m], None, TAUDUST, TAURAY, TAUSCAT, ph,
                                                np.ones((W, 1, L)), radg5_j[m], *tail5)
        same5 = all(np.array_equal(fs5m(m), spec5[m]) for m in pick)
        ex["c4_jacobian_default_streams"] = {
            "what": "the same numerical Jacobian (%d multiple-scattering forward models) at the reference's default quadrature: 5 streams, "
                    "NF = 2 (k_ms_chain_lane<5, CACHE>)" % nj,
            "wall_s": t5j, "wall_s_first_call": t5_first, "s_per_forward_model": t5j / nj, "layers_from_cache": int(hits5),
            "layers_of_models_1_to_n": int(tot5), "bit_identical_to_separate_calls": bool(same5), "models_compared": pick}
        del spec_j, aj, radg_j, spec5, aj5, radg5_j

    # ---- C5: runtime line-by-line, 1e6 wavenumbers x 50 layers x 1e5 lines (Voigt, windows 25 / 75 cm-1) --------------
    nw, N, Ll = 1000000, 100000, 50
    wn = 2000.0 + 1e-3 * np.arange(nw)
    span = nw * 1e-3
    nu = np.sort(rng.uniform(2000.0 - 75.0, 2000.0 + span + 75.0, N)); sw = 10.0 ** rng.uniform(-28, -19, N)
    el = rng.uniform(0, 3000, N)
    bp = np.zeros((3, N)); bp[0] = rng.uniform(0.02, 0.1, N); bp[1] = rng.uniform(0.5, 0.8, N); bp[2] = rng.uniform(-0.01, 0.01, N)
    sr = 1 - np.exp(-(2.99792458E10 * 6.62607015E-27 / 1.380649E-16) * nu / 296.0)
    tt = np.linspace(150, 300, Ll); pp = np.logspace(-4, 0, Ll); qq = np.ones(Ll)
    o = np.zeros((Ll, nw))
    fl = lambda: eng.add_line_set_monochromatic_absorption(wn, 0, tt, 296.0, pp, 1.0, qq, 1.0, 28.0, np.array([1.0]), bp, nu, sw,
                                                           el, sr, o)
    t = med(fl, 2)
    evals = float(N) * (150.0 / 1e-3) * Ll * (span / (span + 150.0))
    ex["c5_lbl"] = {"what": "add_line_set_monochromatic_absorption at BASELINE configs[4] size (k_lbl_line_params + "
                            "k_lbl_accumulate), host arrays in / out (800 MB of PCIe inside wall_s)", "wall_s": t,
                    "profile_evaluations": evals, "Gevals_per_s": evals / t / 1e9}
    if do_cpu:      # a slab of the same launch against the oracle's line loop (every line whose window reaches the slab)
        from oracle import oracle as orc
        o[:] = 0.0
        fl()
        i0, nsl = 431000, 2000
        grid = wn[i0:i0 + nsl]
        near = (nu > grid[0] - 75.5) & (nu < grid[-1] + 75.5)
        err = 0.0
        for l in (0, 24, 49):
            ref = np.zeros(nsl)
            orc.add_line_set_monochromatic_absorption(grid, 0, tt[l], 296.0, pp[l], 1.0, qq[l], 1.0, 28.0, np.array([1.0]), bp[:, near],
                                                      nu[near], sw[near], el[near], sr[near], ref)
            err = max(err, float(np.max(np.abs(o[l, i0:i0 + nsl] - ref) / ref)))
        ex["c5_lbl"]["max_rel_err_vs_oracle"] = err
        ex["c5_lbl"]["oracle_sample"] = f"{nsl} grid points x 3 layers of the full-size launch, {int(near.sum())} lines in reach"
    del o

    # ---- north-star variant: C2 with 20 gases (replaces the table in HBM: last) ----------------------------------------
    S2 = 20
    PRESS, TEMP, Kdev = torch_ktable(torch, dev, W, G, NP, NT, S2, seed=20260705)
    PRESS, TEMP = PRESS.astype(np.float32), TEMP.astype(np.float32)
    eng.upload_ktable(Kdev, PRESS, TEMP, WAVE, delg)
    Wc = 2000
    K_sample = Kdev[:Wc].cpu().numpy() if do_cpu else None
    del Kdev
    torch.cuda.empty_cache()
    a2 = syn.synth_atmosphere(L, S2, seed=7)
    cont = syn.synth_continuum(W, L)
    f8 = torch.float64
    td = lambda a, dt=f8: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    d = [td(a2["lay_press_pa"][0]), td(a2["lay_temp"][0]), td(a2["amount"][0]), td(cont), td(NLAYIN, torch.int32),
         td(LAYINC, torch.int32), td(SCALE), td(EMTEMP[0]), td(np.full(1, -1.0))]
    out = torch.empty((1, W, 1), dtype=f8, device=dev)
    step = lambda: eng.cirsrad_ck_thermal_dev(0, 1, L, d[0], d[1], d[2], d[3], 1, L, d[4], d[5], d[6], d[7], d[8], None, None,
                                              None, None, None, None, out)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 10
    k = eng.last_kernel_ms()
    e = {"what": "the C2 forward model with 20 gases (north_star target variant), inputs resident in HBM", "value": 1.0 / t,
         "unit": "forward-models/s", "ms_per_step": t * 1e3, "overlap_kernel_ms": k["overlap_ms"],
         "algorithmic_bytes": algorithmic_bytes(W, G, S2, L, NP, NT, 1), }
    e["hbm_frac"] = e["algorithmic_bytes"] / (k["overlap_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if do_cpu:
        from oracle import oracle as orc
        cores = orc.num_threads()
        t0 = time.perf_counter()
        ref = orc.cirsrad_ck_thermal(0, K_sample, PRESS, TEMP, WAVE[:Wc], delg, a2["lay_press_pa"][0], a2["lay_temp"][0],
                                     a2["amount"][0], cont[0][:Wc], NLAYIN, LAYINC, SCALE, EMTEMP[0], -1.0)
        ct = time.perf_counter() - t0
        got = out[0, :Wc].cpu().numpy()
        e["cpu_baseline"] = {"value": (Wc / W) / ct, "unit": "forward-models/s", "cores": int(cores), "kind": "port",
                             "sample": f"first {Wc} of {W} wavenumbers, {ct:.2f} s wall, value scaled by {Wc}/{W}",
                             "gpu_vs_oracle_max_rel_err_on_sample": float(np.max(np.abs(got - ref) / np.abs(ref)))}
        e["gpu_over_cpu"] = e["value"] / e["cpu_baseline"]["value"]
    ex["c2_20_gases"] = e

    # ---- C2 with G = 10 g-ordinates ("G = 20; also report G = 10"), 8 gases ----------------------------------
    G2 = 10
    _, delg2 = syn.gauss_legendre_01(G2, as_float32=True)
    delg2 = delg2.astype(np.float32)
    PRESS, TEMP, Kdev = torch_ktable(torch, dev, W, G2, NP, NT, S, seed=20260706)
    eng.upload_ktable(Kdev, PRESS.astype(np.float32), TEMP.astype(np.float32), WAVE, delg2)
    del Kdev
    torch.cuda.empty_cache()
    d = [td(atm["lay_press_pa"][0]), td(atm["lay_temp"][0]), td(atm["amount"][0]), td(cont), td(NLAYIN, torch.int32),
         td(LAYINC, torch.int32), td(SCALE), td(EMTEMP[0]), td(np.full(1, -1.0))]
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 10
    k = eng.last_kernel_ms()
    e = {"what": "the C2 forward model with G = 10 g-ordinates (8 gases), inputs resident in HBM", "value": 1.0 / t,
         "unit": "forward-models/s", "ms_per_step": t * 1e3, "overlap_kernel_ms": k["overlap_ms"],
         "algorithmic_bytes": algorithmic_bytes(W, G2, S, L, NP, NT, 1)}
    e["hbm_frac"] = e["algorithmic_bytes"] / (k["overlap_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    ex["c2_g10"] = e
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gases", type=int, default=8)
    ap.add_argument("--waves", type=int, default=10000)
    ap.add_argument("--layers", type=int, default=100)
    ap.add_argument("--ng", type=int, default=20)
    ap.add_argument("--jac-models", type=int, default=201)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-jacobian", action="store_true")
    ap.add_argument("--cpu-sample-waves", type=int, default=10000)
    ap.add_argument("--no-extras", action="store_true", help="skip the extra keys (S=20 variant, gradients, C4, C5)")
    args = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: native libraries write there too (RCCL prints a version banner at the
    # first communicator), so file descriptor 1 is pointed at stderr for the run and the line goes out through a copy of
    # the original descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import archnemesis_dist_amd as pkg
    from archnemesis_dist_amd import synthetic as syn
    from archnemesis_dist_amd.jacobian import chunk_range, gather_columns

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # ANSFM_BENCH_REHEARSAL=1 (tests only): every rank on GPU 0 and the gloo backend, so that the N > 1 code path can be run
    # on a one-GPU box (RCCL refuses two ranks on one device).  Never set by the driver; the line says so when it is.
    rehearsal = os.environ.get("ANSFM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)   # launched by torch.distributed.run
    if use_dist:
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    if not os.path.exists(pkg.LIB_PATH):        # source-only checkout: one rank compiles the library, the others wait for it
        if local_rank == 0:
            pkg.build()
        if use_dist:
            dist.barrier()

    W, G, S, L, NP, NT, P = args.waves, args.ng, args.gases, args.layers, 20, 15, 1
    f8 = torch.float64
    eng = pkg.AnsfmEngine(local_rank)
    # one stream for torch and the engine: torch's default stream has a null handle, which the C-ABI reads as "the engine's
    # own stream" -- the two would then run side by side on shared buffers
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)

    # ---- synthetic inputs, resident in HBM -----------------------------------------------------
    _, delg = syn.gauss_legendre_01(G, as_float32=True)
    delg = delg.astype(np.float32)          # a .kta header gives float32 arrays: NumPy then forms del_g[i]*del_g[j] in float32
    WAVE = 200.0 + 0.1 * np.arange(W)
    PRESS, TEMP, Kdev = torch_ktable(torch, dev, W, G, NP, NT, S, seed=20260704)
    PRESS, TEMP = PRESS.astype(np.float32), TEMP.astype(np.float32)      # like read_ktahead (Spectroscopy_0.py:2544-2559)
    torch.cuda.synchronize()
    t0 = time.time()
    eng.upload_ktable(Kdev, PRESS, TEMP, WAVE, delg)
    torch.cuda.synchronize()
    table_relayout_s = time.time() - t0
    Wc = min(args.cpu_sample_waves, W)
    do_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline      # contract: rank 0 at N = 1 only
    K_sample = Kdev[:Wc].cpu().numpy() if do_cpu else None
    # N > 1: the Jacobian is sharded over the spectral axis (jacobian_nemesis_batched, shard = "wavenumbers"): a second
    # context on this GPU holds this rank's slice chunk_range(W, N, rank) of the same table -- 1/N of it, not a replica
    eng_j, wj0, wj1 = eng, 0, W
    if world > 1 and not args.no_jacobian:
        wj0, wj1 = chunk_range(W, world, rank)
        eng_j = pkg.AnsfmEngine(local_rank)
        eng_j.set_stream(stream.cuda_stream)
        eng_j.upload_ktable(Kdev[wj0:wj1].contiguous(), PRESS, TEMP, WAVE[wj0:wj1], delg)
        torch.cuda.synchronize()
    del Kdev
    torch.cuda.empty_cache()

    nj = args.jac_models
    atm = syn.synth_atmosphere(L, S, seed=7 + rank, n_models=nj, perturb=0.05)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
    cont = syn.synth_continuum(W, L)                                   # (1,W,L), shared by all models
    EMTEMP = atm["lay_temp"][:, LAYINC[:, 0]][:, :, None]              # (n,L,1)
    td = lambda a, dt=f8: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    d_lp, d_lt, d_am = td(atm["lay_press_pa"]), td(atm["lay_temp"]), td(atm["amount"])
    d_cont1 = td(cont)
    d_nlayin, d_layinc = td(NLAYIN, torch.int32), td(LAYINC, torch.int32)
    d_scale = td(np.repeat(SCALE[None], nj, 0)); d_emtemp = td(EMTEMP)
    d_tsurf = td(np.full(nj, -1.0))
    d_out1 = torch.empty((1, W, P), dtype=f8, device=dev)

    def step(m):
        """one forward model (model index m of this rank's batch)"""
        eng.cirsrad_ck_thermal_dev(0, 1, L, d_lp[m], d_lt[m], d_am[m], d_cont1, P, L, d_nlayin, d_layinc, d_scale[m],
                                   d_emtemp[m], d_tsurf[m:m + 1], None, None, None, None, None, None, d_out1)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i % nj)
    barrier()
    ov_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i % nj)
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel duration measured live with hipEvents on the stream the kernels run on:
    # a few extra (untimed) steps, one event pair per launch
    rt_ms = []
    for i in range(min(args.steps, 10)):
        step(i % nj)
        k = eng.last_kernel_ms()
        ov_ms.append(k["overlap_ms"]); rt_ms.append(k["rt_ms"])
    if use_dist:
        t = torch.tensor([elapsed], dtype=f8, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * args.steps / elapsed

    # ---- numerical Jacobian (C3): NX = 2 L state vector, nfm = NX + 1 forward models, sharded + one gather -------
    # jacobian_nemesis (ForwardModel_0.py:2184-2361) for a state vector of T at every level and ln VMR of the first
    # absorber at every level (model 0, continuous profiles): every perturbed state goes through Curtis-Godson
    # layer_average, the Rayleigh continuum and CIRSrad -- one batched call per rank (jacobian_nemesis_batched).  A level
    # perturbation changes the few layers whose slant paths cross it; the engine recomputes only those
    # (`layer_opacities_computed`).  Measured twice: with and without that de-duplication (same KK to the last bit).
    jac = None
    model = None
    if not args.no_jacobian:
        from archnemesis_dist_amd.jacobian import jacobian_nemesis_batched
        from archnemesis_dist_amd.profile_state import ContinuousProfileState, BatchedCKThermalModel
        npro = (nj - 1) // 2
        pr = syn.synth_profiles(npro, S + 2, seed=11)
        st = ContinuousProfileState(pr["H"], pr["P"], pr["T"], pr["VMR"], ["T", ("VMR", 2)])
        model = BatchedCKThermalModel(eng_j, st, pr["RADIUS"], pr["ID"], pr["ISO"], list(range(2, S + 2)),
                                      layering_args=dict(NLAY=L, LAYINT=1, NINT=101), IRAY=4)
        model.global_waves = W
        shard = "wavenumbers" if world > 1 else "states"

        def run_jac(dedup):
            eng_j.set_layer_dedup(dedup)
            barrier()
            t0 = time.perf_counter()
            YN, KK = jacobian_nemesis_batched(model, rank=rank, world_size=world, force_collective=use_dist, shard=shard)
            barrier()
            jt = time.perf_counter() - t0
            km = eng_j.last_kernel_ms()
            if use_dist:
                t = torch.tensor([jt], dtype=f8, device="cpu" if rehearsal else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                jt = float(t.item())
            return jt, model.last_rows, YN, KK, km

        run_jac(True)                                   # warm-up: buffers of the batch sizes
        run_jac(False)                                  # ... and of the all-layers call (first-touch allocation of 3.2 GB)
        jt_all, rows_all, YN_a, KK_a, _ = run_jac(False)
        reps = []                                       # median of five: the host side of a call (Python, NumPy, copies on a
        for _ in range(5):                              # shared box) varies 0.057 - 0.08 s while its kernels take the same 47 ms
            reps.append(run_jac(True))
            reps[-1] = reps[-1][:2] + ((None, None) if len(reps) < 5 else reps[-1][2:4]) + reps[-1][4:]   # keep one KK only
        jts = sorted(r[0] for r in reps)
        jt, rows, YN_j, KK_j = jts[2], reps[-1][1], reps[-1][2], reps[-1][3]
        eng_j.set_layer_dedup(True)
        jac = {"forward_models": st.NX + 1, "sharding": ("spectral axis: every rank runs all forward models on its 1/%d of the "
                                                          "wavenumbers (table split, not replicated)" % world) if world > 1 else None, "state_vector": f"T and ln(VMR) of one absorber at {npro} levels (NX = {st.NX}), "
               "through layer_average (Curtis-Godson, NINT 101) and the Rayleigh continuum",
               "wall_s": jt, "wall_s_five_calls": jts, "merge_kernel_ms_rank0": reps[-1][4]["overlap_ms"],
               "rt_kernel_ms_rank0": reps[-1][4]["rt_ms"], "fm_per_s": (st.NX + 1) / jt, "wall_s_all_layers": jt_all,
               "layer_opacities_computed_rank0": int(rows[0]), "layer_opacities_all_rank0": int(rows_all[0]),
               "dedup_bit_identical": bool(np.array_equal(KK_a, KK_j) and np.array_equal(YN_a, YN_j)),
               "kk_shape": list(KK_j.shape),
               "collective": ("one all_gather_into_tensor (%s) of the (nfm, NY / N) blocks" % ("gloo, REHEARSAL on one GPU" if rehearsal else "RCCL"))
               if use_dist else None}
        if rank == 0 and world == 1:
            # the same state vector by analytic gradients (nemesisfmg's route: layer_averageg -> CIRSrad(return_grad) ->
            # map2pro -> map2xvec; jacobian_nemesis(analytical_gradient=True))
            m2 = model
            m2.jacobian_analytic()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ya, ka = m2.jacobian_analytic()
                ts.append(time.perf_counter() - t0)
            jac["analytic_route"] = {"what": "YN, KK (NY x NX) of the same state vector by analytic gradients; the layer- and level-level "
                                             "gradients (80 MB each) stay on the device, KK (16 MB) comes back", "wall_s": sorted(ts)[1], "kk_shape": list(ka.shape)}

    # ---- what one rank of an 8-GPU wavenumber-sharded Jacobian does, measured on THIS GPU: a second context holds the first
    #      1/8 of the spectral axis (chunk_range(W, 8, 0)) and runs all 201 states on it (bench.py --gpus 8 does exactly this on
    #      every rank, plus one all_gather of 2 MB per rank).  Reported beside the one-GPU wall time; not a scaling measurement.
    if jac is not None and world == 1 and rank == 0 and not args.no_extras:
        w1 = chunk_range(W, 8, 0)[1]
        eng8 = pkg.AnsfmEngine(local_rank)
        eng8.set_stream(stream.cuda_stream)
        PRESS8, TEMP8, K8 = torch_ktable(torch, dev, W, G, NP, NT, S, seed=20260704)
        eng8.upload_ktable(K8[:w1].contiguous(), PRESS8.astype(np.float32), TEMP8.astype(np.float32), WAVE[:w1], delg)
        del K8
        torch.cuda.empty_cache()
        m8 = BatchedCKThermalModel(eng8, st, pr["RADIUS"], pr["ID"], pr["ISO"], list(range(2, S + 2)),
                                   layering_args=dict(NLAY=L, LAYINT=1, NINT=101), IRAY=4)
        m8.global_waves = W
        ts8 = []
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            y8, k8 = jacobian_nemesis_batched(m8)
            torch.cuda.synchronize(); ts8.append(time.perf_counter() - t0)
        jac["one_rank_of_eight"] = {
            "what": "the work of ONE rank of an 8-GPU run sharded over the spectral axis, measured on this GPU: all %d states on "
                    "%d of %d wavenumbers (its own slice of the k-table); the 8-GPU wall time is this plus one all_gather of "
                    "%.1f MB per rank and the full-size KK assembly" % (st.NX + 1, w1, W, (st.NX + 1) * w1 * 8 / 1e6),
            "wall_s": sorted(ts8[1:])[2], "same_kk_rows_as_the_whole_axis": bool(np.array_equal(k8, KK_j[:w1])),
            "ratio_to_one_gpu_wall_s": jt / sorted(ts8[1:])[2]}
        eng8.close()

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel -------------------------------------------------------------
    ov = float(np.mean(ov_ms)) * 1e-3
    abytes = algorithmic_bytes(W, G, S, L, NP, NT, P, 1)
    traffic = None
    valu = None
    traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"W{W}_G{G}_S{S}_L{L}"
            traffic = tj.get(key, {}).get("ck_overlap_hbm_bytes_per_launch")
            valu = tj.get(key, {}).get("valu_insts_per_launch")
            traffic_src = ("profiles/pmc_traffic.json: committed rocprofv3 --pmc pass (run %s), NOT measured in this run"
                           % tj.get(key, {}).get("profile", "?"))
        except Exception:
            traffic = None
    roof = {"bound": "hbm", "kernel": "k_ck_overlap", "achieved": abytes / ov / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": abytes / ov / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": abytes, "kernel_ms": ov * 1e3, "rt_kernel_ms": float(np.mean(rt_ms)),
            "note": "second bound (fp64 VALU of the per-lane G-way merge) dominates; see second_bound and DESIGN.md 4.1"}
    if valu:
        # the kernel's own limit: wave64 VALU instructions issued (SQ_INSTS_VALU of the committed PMC pass, per launch)
        # against what the chip can issue -- one wave64 fp64 instruction per SIMD every 4 clocks
        props = torch.cuda.get_device_properties(dev)
        n_simd = props.multi_processor_count * 4
        clk_hz = 2.4e9                                                    # MI355X peak engine clock (MI355X_MICROARCH.md)
        peak = n_simd * clk_hz / 4.0
        roof["second_bound"] = {"bound": "valu_issue", "insts_per_launch": valu, "achieved": valu / ov / 1e9,
                                "peak": peak / 1e9, "unit": "G wave-instructions/s", "frac": valu / ov / peak,
                                "source": "SQ_INSTS_VALU of the committed rocprofv3 --pmc pass (profiles/pmc_traffic.json, not this run) / live kernel time"}

    # ---- CPU baseline: the oracle (port) on a bounded sample, rank 0 only ---------------------------------
    cpu = None
    if do_cpu:
        from oracle import oracle as orc
        orc.build()
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
        # the 1-GPU box exposes every host core but grants a 16-core share: oversubscribing OpenMP
        # makes the baseline slower, not faster
        cores = int(os.environ.get("ANSFM_CPU_THREADS", min(cores, 16)))
        orc.set_num_threads(cores)
        a0 = {k: v[0] for k, v in atm.items()}
        t0 = time.perf_counter()
        ref = orc.cirsrad_ck_thermal(0, K_sample, PRESS, TEMP, WAVE[:Wc], delg, a0["lay_press_pa"], a0["lay_temp"],
                                     a0["amount"], cont[0][:Wc], NLAYIN, LAYINC, SCALE, EMTEMP[0], -1.0)
        ct = time.perf_counter() - t0
        step(0)
        torch.cuda.synchronize()
        got = d_out1[0, :Wc].cpu().numpy()
        perr = float(np.max(np.abs(got - ref) / np.abs(ref)))
        cpu = {"value": (Wc / W) / ct, "unit": "forward-models/s", "cores": int(cores), "kind": "port",
               "sample": (f"{'the whole' if Wc == W else f'first {Wc} of {W} wavenumbers of the'} C2 forward model "
                          f"({Wc} wavenumbers x {L} layers x {S} gases, G={G}) through oracle/ansfm_oracle.c, OpenMP over "
                          f"wavenumbers, {ct:.2f} s wall x {cores} threads" + ("" if Wc == W else f"; value scaled by {Wc}/{W}")),
               "gpu_vs_oracle_max_rel_err_on_sample": perr}

    # ---- C3 check: a sample of KK columns against the reference's recipe run on the oracle's forward models -------
    if do_cpu and jac is not None:
        from oracle import jacobian_twin as twin
        cols = np.unique(np.concatenate([np.linspace(2, npro - 3, 8).astype(int), npro + np.linspace(2, npro - 3, 8).astype(int)]))
        t0 = time.perf_counter()
        y0, kk = twin.jacobian(model, K_sample, PRESS, TEMP, WAVE[:Wc], delg, columns=cols)
        sc = np.max(np.abs(kk), axis=0)
        sc = np.maximum(sc, 1.0e-6 * float(np.max(np.abs(KK_j))))       # a column without sensitivity is rounding noise
        nyc = kk.shape[0]
        jac["kk_max_rel_err_vs_oracle"] = float(np.max(np.abs(KK_j[:nyc, cols] - kk) / sc))
        jac["yn_max_rel_err_vs_oracle"] = float(np.max(np.abs(YN_j[:nyc] - y0) / np.abs(y0)))
        jac["oracle_sample"] = (f"columns {cols.tolist()} of KK (8 temperature levels, 8 ln VMR levels; {len(cols) + 1} oracle forward models at "
                                f"{Wc} wavenumbers, {time.perf_counter() - t0:.1f} s); error relative to each column's maximum")

    extras = None
    if world == 1 and not args.no_extras and (W, G, S, L) == (10000, 20, 8, 100):
        extras = run_extras(torch, dev, eng, syn, atm, NLAYIN, LAYINC, SCALE, EMTEMP, WAVE, delg, do_cpu, K_sample, PRESS, TEMP, model)

    line = {
        "metric": "forward-models/sec (10k nu x 100 layers)", "value": value, "unit": "forward-models/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic correlated-k nadir thermal-emission forward model: {W} wavenumbers x "
                               f"{L} layers x {S} gases, G={G}, k-table {NP}Px{NT}T (BASELINE configs[1], SURVEY C2)",
                   "forward_models_per_step_per_gpu": 1, "parallelism": f"replicas x{world} (independent forward models)"},
        "roofline": roof, "cpu_baseline": cpu, "jacobian": jac,
        "table_relayout_s": table_relayout_s,
        "extras": extras,
    }
    sys.stdout.flush()
    with os.fdopen(json_fd, "w") as out:
        out.write(json.dumps(line) + "\n")
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
