#!/usr/bin/env python
"""Minimal use of the engine without the reference: a synthetic correlated-k nadir atmosphere (reduced),
one forward model, a batch of perturbed states (numerical Jacobian) and the analytic layer gradients.

    python examples/c2_forward.py            # needs an MI355X and a built libansfm.so
"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import archnemesis_dist_amd as pkg                                    # noqa: E402
from archnemesis_dist_amd import synthetic as syn                      # noqa: E402


def main():
    W, G, S, L, NP, NT = 2000, 20, 8, 100, 20, 15
    eng = pkg.AnsfmEngine(0)
    _, delg = syn.gauss_legendre_01(G, as_float32=True)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S)                 # K (NWAVE,NG,NP,NT,NGAS) like Spectroscopy_0.K
    eng.upload_ktable(K, PRESS, TEMP, 200.0 + 0.1 * np.arange(W), delg)

    n = 41                                                             # base state + 40 single-layer perturbations
    atm = syn.synth_atmosphere(L, S, n_models=n, perturb=0.05)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
    EMTEMP = atm["lay_temp"][:, LAYINC[:, 0]][:, :, None]
    cont = np.repeat(syn.synth_continuum(W, L), n, 0)

    t = time.perf_counter()
    spec = eng.cirsrad_ck_thermal(0, atm["lay_press_pa"], atm["lay_temp"], atm["amount"], cont, NLAYIN, LAYINC,
                                  np.repeat(SCALE[None], n, 0), EMTEMP, np.full(n, -1.0))
    dt = time.perf_counter() - t
    rows, total = eng.last_layer_rows()
    print(f"{n} forward models ({W} wavenumbers x {L} layers x {S} gases): {dt * 1e3:.1f} ms, "
          f"{rows} of {total} layer opacities computed (layer de-duplication)")
    print("base spectrum [W cm-2 sr-1 (cm-1)-1]:", spec[0, :3, 0])
    dy = (spec[1:] - spec[0]) / 0.05                                   # d(spectrum)/d(ln x) by finite differences
    print("largest finite-difference response:", float(np.abs(dy).max()))

    NVMR, NPAR = S, S + 2
    s1, dspec, dts = eng.cirsradg_ck_thermal(0, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0], cont[0], None,
                                             NVMR, NPAR, np.arange(S, dtype=np.int32), NLAYIN, LAYINC, SCALE, EMTEMP[0], -1.0)
    print("analytic layer gradients dSPECOUT", dspec.shape, "agree with the forward spectrum:",
          bool(np.allclose(s1, spec[0], rtol=1e-12)))


if __name__ == "__main__":
    main()
