"""Golden fixture for the single-scattering plane-parallel spectrum: the REFERENCE's calc_singlescatt_plane_spectrum
(ForwardModel_0.py:6509-6600) on seeded arrays, both spectral units, with and without a surface (build container only).

    python oracle/gen_golden_singlescatt.py      # -> tests/golden/singlescatt.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import_reference()
    fm = sys.modules["archnemesis.ForwardModel_0"]
    rng = np.random.default_rng(61)
    W, G, Li = 9, 5, 13
    out = {}
    for ispace, tag in ((0, "wn"), (1, "wl")):
        WAVE = (400.0 + 55.0 * np.arange(W)) if ispace == 0 else (0.8 + 0.45 * np.arange(W))
        TAU = 10.0 ** rng.uniform(-5, 0.6, (W, G, Li)); TAU[1, 2, :] = 0.0
        OMEGA = rng.uniform(0.0, 1.0, (W, G, Li)); OMEGA[:, :, 3] = 0.0
        PHASE = 10.0 ** rng.uniform(-2, 0.5, (W, Li))
        TEMP = np.linspace(140.0, 290.0, Li) + rng.uniform(-4, 4, Li)
        EMIS = rng.uniform(0.6, 1.0, W); BRDF = rng.uniform(0.0, 0.2, W); SOL = 10.0 ** rng.uniform(-8, -6, W)
        out.update({f"{tag}_WAVE": WAVE, f"{tag}_TAU": TAU, f"{tag}_OMEGA": OMEGA, f"{tag}_PHASE": PHASE, f"{tag}_TEMP": TEMP,
                    f"{tag}_EMIS": EMIS, f"{tag}_BRDF": BRDF, f"{tag}_SOL": SOL})
        for cn, (TSURF, sa, ea) in {"nosurf": (-1.0, 35.0, 20.0), "surf": (270.0, 62.0, 5.0), "graze": (270.0, 80.0, 70.0)}.items():
            out[f"{tag}_{cn}_args"] = np.array([TSURF, sa, ea])
            out[f"{tag}_{cn}_spec"] = fm.calc_singlescatt_plane_spectrum(ispace, WAVE, TAU, TEMP, OMEGA, PHASE, TSURF, EMIS, BRDF,
                                                                         SOL, sa, ea)
    fn = os.path.join(OUT, "singlescatt.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
