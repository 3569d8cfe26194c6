"""Golden fixtures for the runtime line-by-line kernels: the REFERENCE's
LineData_0.add_line_set_monochromatic_absorption (:280) with its default Voigt
(scipy.special.voigt_profile), Lorentz and Gaussian line shapes on seeded synthetic line lists,
plus a lattice of scipy.special.voigt_profile values (third-party arithmetic the reference calls).

    python oracle/gen_golden_lbl.py       # -> tests/golden/lbl_lines.npz, voigt_lattice.npz
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    ans = import_reference()
    import importlib
    import scipy.special
    ld = importlib.import_module("archnemesis.LineData_0")
    ls = importlib.import_module("archnemesis.lineshape")
    rng = np.random.default_rng(42)
    N, M = 240, 2
    nu = np.sort(rng.uniform(1993.0, 2017.0, N))
    sw = 10.0 ** rng.uniform(-26, -19, N)
    e_lower = rng.uniform(0.0, 3000.0, N)
    bp = np.zeros((3 * M, N))
    bp[0] = rng.uniform(0.05, 0.12, N); bp[1] = rng.uniform(0.5, 0.8, N); bp[2] = rng.uniform(-0.01, 0.01, N)     # self
    bp[3] = rng.uniform(0.02, 0.1, N); bp[4] = rng.uniform(0.5, 0.8, N); bp[5] = rng.uniform(-0.01, 0.01, N)      # air
    mmf = np.array([0.1, 0.9])
    t_ref, p_ref = 296.0, 1.0
    c2 = 2.99792458E10 * 6.62607015E-27 / 1.380649E-16
    stim_ref = 1 - np.exp(-c2 * nu / t_ref)
    wn = 2000.0 + 0.01 * np.arange(1001)
    cases = [("voigt_mid", ls.voigt, 0, 200.0, 0.5, 1.3, 0.0, 25.0, 75.0),
             ("voigt_doppler_regime", ls.voigt, 0, 150.0, 1e-4, 2.1, 0.0, 25.0, 75.0),
             ("voigt_wings", ls.voigt, 0, 296.0, 1.0, 1.0, 1e-24, 2.0, 5.0),
             ("lorentz", ls.lorentz, 4, 250.0, 0.3, 1.1, 0.0, 3.0, 8.0),
             ("gaussian", ls.gaussian, 12, 180.0, 0.01, 1.6, 0.0, 1.0, 4.0)]
    out = dict(wn_grid=wn, nu=nu, sw=sw, e_lower=e_lower, broadening_params=bp, mol_mix_frac=mmf, stim_ref=stim_ref,
               t_ref=t_ref, p_ref=p_ref, isotopic_abundance=0.98654, isotopic_mass=28.0, names=np.array([c[0] for c in cases]))
    for name, fn, lid, t, p, q, sfl, wc, wa in cases:
        k = np.zeros(wn.size); store = np.empty((4, N))
        ld.add_line_set_monochromatic_absorption(wn, fn, t, t_ref, p, p_ref, q, 0.98654, 28.0, mmf, bp, nu, sw, e_lower,
                                                 stim_ref, k, store, sfl, wc, wa)
        out[name + "_k"] = k; out[name + "_store"] = store
        out[name + "_args"] = np.array([lid, t, p, q, sfl, wc, wa])
        print(name, k.min(), k.max())
    np.savez_compressed(os.path.join(OUT, "lbl_lines.npz"), **out)
    # Voigt lattice (scipy.special.voigt_profile is third-party arithmetic on the path)
    x = np.concatenate([[0.0], 10.0 ** np.linspace(-6, 2, 33)])
    sig = 10.0 ** np.linspace(-5, 0, 11)
    gam = np.concatenate([[0.0], 10.0 ** np.linspace(-8, 1, 19)])
    X, S, G = np.meshgrid(x, sig, gam, indexing="ij")
    V = scipy.special.voigt_profile(X, S, G)
    np.savez_compressed(os.path.join(OUT, "voigt_lattice.npz"), x=x, sigma=sig, gamma=gam, V=V)
    print("lattice", V.shape)


if __name__ == "__main__":
    main()
