"""Runtime line-by-line: oracle vs goldens from the reference's add_line_set_monochromatic_absorption and
from scipy.special.voigt_profile (the reference's default Voigt)."""
import os
import numpy as np
import pytest


def lbl_case(z, name):
    lid, t, p, q, sfl, wc, wa = z[name + "_args"]
    return dict(lineshape_id=int(lid), t_calc=float(t), p_calc=float(p), q_ratio=float(q), s_floor=float(sfl),
                wn_calc_window=float(wc), wn_approx_window=float(wa))


def test_voigt_lattice(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "voigt_lattice.npz"))
    X, S, G = np.meshgrid(z["x"], z["sigma"], z["gamma"], indexing="ij")
    V = oracle.voigt_profile(X, S, G)
    ref = z["V"]
    ok = ref > 1e-300
    assert np.max(np.abs(V[ok] - ref[ok]) / ref[ok]) < 5e-13
    assert np.all(V[~ok] < 1e-290)


def test_add_line_set_monochromatic_absorption(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "lbl_lines.npz"))
    for name in z["names"]:
        a = lbl_case(z, str(name))
        out = np.zeros(z["wn_grid"].size); store = np.empty((4, z["nu"].size))
        oracle.add_line_set_monochromatic_absorption(
            z["wn_grid"], a["lineshape_id"], a["t_calc"], float(z["t_ref"]), a["p_calc"], float(z["p_ref"]), a["q_ratio"],
            float(z["isotopic_abundance"]), float(z["isotopic_mass"]), z["mol_mix_frac"], z["broadening_params"], z["nu"],
            z["sw"], z["e_lower"], z["stim_ref"], out, store, a["s_floor"], a["wn_calc_window"], a["wn_approx_window"])
        np.testing.assert_allclose(store, z[str(name) + "_store"], rtol=1e-13, err_msg=str(name))
        np.testing.assert_allclose(out, z[str(name) + "_k"], rtol=1e-11, err_msg=str(name))
