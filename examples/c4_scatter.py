#!/usr/bin/env python
"""Multiple scattering without the reference: CIRSrad's doubling / adding branch on a synthetic atmosphere, at the reference's
default quadrature (5 streams, NF = 2) and at 16 streams, then the forward models of a numerical Jacobian as ONE batched call
(model 0's doubled layers are cached, the perturbed states run the adding sweep over them).

    python examples/c4_scatter.py            # needs an MI355X and a built libansfm.so
"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import archnemesis_dist_amd as pkg                                    # noqa: E402
from archnemesis_dist_amd import synthetic as syn                      # noqa: E402


def scattering_inputs(W, L, WAVE, lay_t, nmu):
    """One aerosol (Henyey-Greenstein-like phase function on 41 angles) + Rayleigh, no surface reflection."""
    x, w = np.polynomial.legendre.leggauss(nmu)
    MU, WT = 0.5 * (x + 1.0), 0.5 * w                                  # Gauss-Legendre on (0, 1): sum(mu w) = 1/2
    TH = np.linspace(0.0, 180.0, 41); c = np.cos(np.deg2rad(TH))
    g = 0.6
    ph = np.zeros((1, W, 2, TH.size))
    ph[0, :, 0, :] = ((1 - g * g) / (1 + g * g - 2 * g * c) ** 1.5 / (4 * np.pi))[None, :]
    ph[0, :, 1, :] = c[None, :]
    ph = np.ascontiguousarray(ph[:, :, :, ::-1])                       # as Scatter_0 hands it to scloud11wave
    wv = np.linspace(0, 1, W)[:, None]; lv = np.linspace(0, 1, L)[None, :]
    TAURAY = 1e-3 * np.exp(-5.0 * lv) * (1.0 + 0.3 * wv)
    TAUSCAT = 2e-2 * np.exp(-((lv - 0.35) / 0.1) ** 2) * (1.0 + 0.5 * np.sin(7.0 * wv))
    TAUDUST = 1.1 * TAUSCAT
    c1, c2 = 1.1911e-12, 1.439
    radg = np.repeat((c1 * WAVE ** 3 / (np.exp(c2 * WAVE / lay_t[0]) - 1.0))[:, None], nmu, 1)
    return MU, WT, ph, TAURAY, TAUSCAT, TAUDUST, radg


def main():
    W, G, S, L, NP, NT = 2000, 20, 4, 60, 12, 10
    eng = pkg.AnsfmEngine(0)
    _, delg = syn.gauss_legendre_01(G, as_float32=True)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S)
    WAVE = 200.0 + 0.1 * np.arange(W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    n = 21                                                             # base state + 20 single-layer perturbations
    atm = syn.synth_atmosphere(L, S, n_models=n, perturb=0.05)
    for nmu, nf in ((5, 2), (16, 8)):
        MU, WT, ph, TAURAY, TAUSCAT, TAUDUST, radg = scattering_inputs(W, L, WAVE, atm["lay_temp"][0], nmu)
        tail = ([30.0], [20.0], [45.0], np.full(W, 1e-8), 0, np.zeros((W, nmu, nmu, nf + 1)), MU, WT, nf, 101, 1, 1)
        one = lambda m: eng.cirsrad_ck_scatter(0, atm["lay_press_pa"][m], atm["lay_temp"][m], atm["amount"][m], None, TAUDUST, TAURAY,
                                               TAUSCAT, ph, np.ones((W, 1, L)), radg, *tail)
        one(0)
        t = time.perf_counter(); spec0 = one(0); t1 = time.perf_counter() - t
        rep = lambda a: np.ascontiguousarray(np.broadcast_to(a[None], (n,) + a.shape))
        args = (0, atm["lay_press_pa"], atm["lay_temp"], atm["amount"], None, rep(TAUDUST), rep(TAURAY), rep(TAUSCAT), ph,
                rep(np.ones((W, 1, L))), rep(radg))
        eng.cirsrad_ck_scatter_batch(*args, *tail)
        t = time.perf_counter(); spec = eng.cirsrad_ck_scatter_batch(*args, *tail); tb = time.perf_counter() - t
        hits, total = eng.last_scatter_cache()
        print(f"{nmu:2d} streams, NF = {nf}: one forward model {t1 * 1e3:7.1f} ms; {n} forward models in one call {tb * 1e3:7.1f} ms "
              f"({hits} of {total} layers from model 0's cache); model 0 of the batch equals the single call: "
              f"{bool(np.array_equal(spec[0], spec0))}")
        print("    spectrum [W cm-2 sr-1 (cm-1)-1]:", spec0[:3, 0], " largest response to a 5 % layer perturbation:",
              float(np.max(np.abs(spec[1:] - spec[0]) / np.abs(spec[0]))))


if __name__ == "__main__":
    main()
