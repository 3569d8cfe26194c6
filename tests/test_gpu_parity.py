"""GPU parity: the HIP path (through the C-ABI, libansfm.so) against
  (a) golden vectors produced by the reference itself (tests/golden/, oracle/gen_golden.py), and
  (b) the CPU oracle on seeded inputs the oracle finishes in seconds, and
  (c) at the BASELINE C2 size, size-independent properties.
Tolerances: north_star asks <=1e-6 relative radiance; these tests hold the kernels to ~1e-11."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CK_CASES = ["ck_g10_s4", "ck_g20_s8", "ck_g16_s2", "ck_g8_s1", "ck_g10_s3_nozero", "ck_g10_s3_f32dtype"]


@pytest.fixture(scope="module")
def eng():
    import archnemesis_dist_amd as pkg
    e = pkg.AnsfmEngine(0)
    yield e
    e.close()


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _relmax(a, b, floor=0.0):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor + 1e-300)))


@pytest.mark.parametrize("name", CK_CASES)
def test_calc_k_golden(eng, golden_dir, name):
    z = _load(golden_dir, name)
    eng.upload_ktable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"])
    k = eng.calc_k(z["press"], z["temp"])
    assert np.array_equal(k == 0.0, z["k"] == 0.0)           # good/bad/mixed mask exactly
    # float32 grids: NumPy's float32 log is not correctly rounded -> reference defined to ~1e-7 only
    rt = 2e-7 if name.endswith("f32dtype") else 1e-12
    np.testing.assert_allclose(k, z["k"], rtol=rt, atol=0)
    kg, dk = eng.calc_k(z["press"], z["temp"], grad=True)
    np.testing.assert_allclose(kg, z["kg"], rtol=rt, atol=0)
    np.testing.assert_allclose(dk, z["dkdT"], rtol=max(rt, 1e-10), atol=0)


@pytest.mark.parametrize("name", CK_CASES)
def test_k_overlap_golden(eng, golden_dir, name):
    z = _load(golden_dir, name)
    tau = eng.k_overlap(z["DELG"], z["k"], z["amount"])
    np.testing.assert_allclose(tau, z["tau"], rtol=1e-11, atol=0)


def test_k_overlap_unsorted_golden(eng, golden_dir):
    """k-distributions NOT sorted in g (generic path: per-lane sort of each gas, weights follow) vs the reference's
    k_overlap (golden): skip rules on the last ordinate, unmerged spectra returned in their original order."""
    z = _load(golden_dir, "ko_unsorted_g8_s4")
    tau = eng.k_overlap(z["DELG"], z["k"], z["amount"])
    np.testing.assert_allclose(tau, z["tau"], rtol=1e-12, atol=0)


def test_k_overlapg_unsorted_golden(eng, golden_dir):
    """The same unsorted k-distributions with gradients: k_overlapg / rankg of the reference (golden).  The gradient merge's
    generic path sorts each gas per lane and stages the gradient rows through the same permutations."""
    z = _load(golden_dir, "ko_unsorted_g8_s4")
    taug, dk = eng.k_overlapg(z["DELG"], z["k"], z["dkdT"], z["amount"])
    np.testing.assert_allclose(taug, z["taug"], rtol=1e-12, atol=0)
    scale = np.abs(z["dk"]).max(axis=1, keepdims=True) + 1e-300
    assert np.max(np.abs(dk - z["dk"]) / scale) < 1e-10


@pytest.mark.parametrize("G,S,f32", [(20, 8, True), (10, 3, False), (7, 1, False), (16, 5, False)])
def test_k_overlapg_unsorted_vs_oracle(eng, oracle, G, S, f32):
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(G * 100 + S + 1)
    W, L = 130, 6
    _, delg = syn.gauss_legendre_01(G, f32)
    k = 10.0 ** rng.uniform(-25, -20, (W, G, L, S))             # no exact ties: rankg's gradient rows follow argsort's order
    k[:, -1, 2, S // 2] = 0.0                                   # a gas skipped in one layer (last ordinate only)
    if S > 1:
        k[:, -1, 4, 0] = 0.0                                    # first gas empty: the second is taken as is
    dkdT = k * rng.uniform(-0.02, 0.02, k.shape)
    amount = 10.0 ** rng.uniform(19, 22, (S, L))
    tau, dk = eng.k_overlapg(delg, k, dkdT, amount)
    rt, rdk = oracle.k_overlapg(delg, k, dkdT, amount)
    np.testing.assert_allclose(tau, rt, rtol=1e-11, atol=0)
    scale = np.abs(rdk).max(axis=1, keepdims=True) + 1e-300
    assert np.max(np.abs(dk - rdk) / scale) < 1e-10


@pytest.mark.parametrize("G,S,f32", [(20, 8, True), (10, 3, False), (7, 1, False)])
def test_k_overlap_unsorted_vs_oracle(eng, oracle, G, S, f32):
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(G * 100 + S)
    W, L = 150, 7
    _, delg = syn.gauss_legendre_01(G, f32)
    k = 10.0 ** rng.uniform(-25, -20, (W, G, L, S))
    k[rng.uniform(size=k.shape) < 0.05] = 0.0                   # zeros anywhere, also in the last ordinate
    amount = 10.0 ** rng.uniform(19, 22, (S, L))
    tau = eng.k_overlap(delg, k, amount)
    ref = oracle.k_overlap(delg, k, amount)
    np.testing.assert_allclose(tau, ref, rtol=1e-11, atol=0)


@pytest.mark.parametrize("G,S", [(32, 3), (24, 2), (12, 4), (9, 5), (2, 3), (1, 4)])
def test_merge_list_lengths_vs_oracle(eng, oracle, G, S):
    """Every instantiated list length of the merge kernels (8, 10, 16, 20, 32), with and without padding entries, up
    to the largest supported G (32: the column tag needs its sixth bit) and down to G = 1 -- forward and gradient."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(1000 + G)
    W, L = 70, 4
    _, delg = syn.gauss_legendre_01(G, G % 2 == 0)
    k = np.sort(10.0 ** rng.uniform(-25, -20, (W, G, L, S)), axis=1)
    dkdT = k * rng.uniform(-0.02, 0.02, k.shape)
    amount = 10.0 ** rng.uniform(19, 22, (S, L))
    np.testing.assert_allclose(eng.k_overlap(delg, k, amount), oracle.k_overlap(delg, k, amount), rtol=1e-11, atol=0)
    tau, dk = eng.k_overlapg(delg, k, dkdT, amount)
    rt, rdk = oracle.k_overlapg(delg, k, dkdT, amount)
    np.testing.assert_allclose(tau, rt, rtol=1e-11, atol=0)
    assert np.array_equal(np.isnan(dk), np.isnan(rdk))          # G = 1: rankg's single bin leaves 0/0 slots in the reference too
    ok = ~np.isnan(rdk)
    scale = np.max(np.where(ok, np.abs(rdk), 0.0), axis=1, keepdims=True) + 1e-300
    assert np.max((np.abs(dk - rdk) / scale)[ok], initial=0.0) < 1e-10


@pytest.mark.parametrize("keys", [32, 64])
def test_nan_and_inf_input_stays_in_its_cells(eng, oracle, keys):
    """NaN / inf absorption coefficients poison only the (wavenumber, layer) cells they are in: the merge kernels never
    index by data beyond what the bin sentinel bounds (a NaN key can displace list entries), so the other cells equal
    the clean run bit for bit.  keys = 32: the float32-key forward kernel (NaN and +inf stay on it; a negative value
    such as -inf sends the call to the generic 64-bit path, whose clean run is the comparison then)."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(99)
    W, G, L, S = 70, 10, 5, 4
    _, delg = syn.gauss_legendre_01(G, True)
    k = np.sort(10.0 ** rng.uniform(-25, -20, (W, G, L, S)), axis=1)
    amount = 10.0 ** rng.uniform(19, 22, (S, L))
    dkdT = k * 0.01
    eng.set_merge_keys(keys)
    try:
        clean = eng.k_overlap(delg, k, amount)
        cleang, cleandk = eng.k_overlapg(delg, k, dkdT, amount)
        bad = k.copy()
        bad[3, 4, 1, 2] = np.nan; bad[10, G - 1, 2, 0] = np.inf; bad[33, :, 4, 3] = np.nan
        cells = [(3, 1), (10, 2), (33, 4)]
        if keys == 64:
            bad[20, 0, 3, 1] = -np.inf
            cells.append((20, 3))
        mask = np.ones((W, L), bool)
        for w, l in cells:
            mask[w, l] = False
        with np.errstate(all="ignore"):
            tau = eng.k_overlap(delg, bad, amount)
            taug, dk = eng.k_overlapg(delg, bad, dkdT, amount)
        assert np.array_equal(tau.transpose(0, 2, 1)[mask], clean.transpose(0, 2, 1)[mask])
        assert np.array_equal(taug.transpose(0, 2, 1)[mask], cleang.transpose(0, 2, 1)[mask])
        assert np.array_equal(dk.transpose(0, 2, 1, 3)[mask], cleandk.transpose(0, 2, 1, 3)[mask])
        if keys == 32:      # a negative value: rerun on the generic path, still confined to its cell
            bad[20, 0, 3, 1] = -np.inf
            mask[20, 3] = False
            with np.errstate(all="ignore"):
                tau = eng.k_overlap(delg, bad, amount)
            np.testing.assert_allclose(tau.transpose(0, 2, 1)[mask], clean.transpose(0, 2, 1)[mask], rtol=1e-12, atol=0)
    finally:
        eng.set_merge_keys(64)


@pytest.mark.parametrize("G,S,f32,kind", [(20, 8, True, "lattice"), (20, 3, False, "flat"), (10, 4, True, "lattice"),
                                           (16, 3, False, "mixed"), (32, 2, False, "lattice")])
def test_k_overlap_float32_key_ties_vs_oracle(eng, oracle, G, S, f32, kind):
    """The float32-key merge (k_ck_overlap32) on input made of heads whose float32 keys coincide: sums that differ by
    1e-9 .. 1e-13 relative (all inside one float32 value), exactly equal sums, and a mix with ordinary columns.  The
    exact-sum tie branch has to reproduce the reference's order: same tolerance as every other k_overlap test; and
    the 64-bit-key kernel gives the same numbers."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(4242 + G + S)
    W, L = 130, 3
    _, delg = syn.gauss_legendre_01(G, f32)
    g = np.arange(G, dtype=np.float64)[None, :, None, None]
    if kind == "lattice":       # k_g = k0 (1 + g eps): every pair sum lies within 2 G eps of every other
        eps = 10.0 ** rng.uniform(-13, -9, (W, 1, L, S))
        k = 10.0 ** rng.uniform(-24, -21, (W, 1, L, S)) * (1.0 + g * eps)
    elif kind == "flat":        # exact ties everywhere
        k = np.repeat(10.0 ** rng.uniform(-24, -21, (W, 1, L, S)), G, axis=1)
    else:
        k = np.sort(10.0 ** rng.uniform(-25, -20, (W, G, L, S)), axis=1)
        flat = rng.uniform(size=(W, 1, L, S)) < 0.4
        k = np.where(flat, k[:, :1] * (1.0 + g * 1e-11), k)
    amount = 10.0 ** rng.uniform(19, 22, (S, L))
    ref = oracle.k_overlap(delg, k, amount)
    eng.set_merge_keys(32)
    try:
        tau32 = eng.k_overlap(delg, k, amount)
    finally:
        eng.set_merge_keys(64)
    eng.set_merge_keys(64)
    tau64 = eng.k_overlap(delg, k, amount)
    np.testing.assert_allclose(tau32, ref, rtol=1e-11, atol=0)
    np.testing.assert_allclose(tau64, ref, rtol=1e-11, atol=0)


def test_cirsrad_unsorted_table_vs_oracle(eng, oracle):
    """A k-table that is not monotone in g goes down the generic path for the whole forward model."""
    from archnemesis_dist_amd import synthetic as syn
    W, G, S, L, NP, NT = 200, 10, 4, 15, 6, 5
    _, delg = syn.gauss_legendre_01(G, True)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=3)
    K = K[:, np.random.default_rng(1).permutation(G)]           # scramble the g axis
    WAVE = 300.0 + np.arange(W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    assert eng.ktable_info()[1] is False
    atm = syn.synth_atmosphere(L, S)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
    EMTEMP = atm["lay_temp"][0][LAYINC[:, 0]][:, None]
    spec = eng.cirsrad_ck_thermal(0, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0], None, NLAYIN, LAYINC, SCALE,
                                  EMTEMP, -1.0)
    ref = oracle.cirsrad_ck_thermal(0, K, PRESS, TEMP, WAVE, delg, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0],
                                    None, NLAYIN, LAYINC, SCALE, EMTEMP, -1.0)
    np.testing.assert_allclose(np.squeeze(spec), np.squeeze(ref), rtol=1e-10)
    # the analytic-gradient forward model on the same table: generic path of the gradient merge
    NPAR = S + 2
    args = (0, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0], None, None, S, NPAR, np.arange(S, dtype=np.int32),
            NLAYIN, LAYINC, SCALE, EMTEMP, -1.0)
    sg, dsg, dts = eng.cirsradg_ck_thermal(*args)
    rs, rds, rdt = oracle.cirsradg_ck_thermal(0, K, PRESS, TEMP, WAVE, delg, *args[1:])
    np.testing.assert_allclose(np.squeeze(sg), np.squeeze(ref), rtol=1e-10)
    sc = np.abs(rds).max(axis=(0, 2, 3), keepdims=True) + 1e-300
    assert np.max(np.abs(np.asarray(dsg).reshape(rds.shape) - rds) / sc) < 1e-9


def test_thermal_emission_golden(eng, golden_dir):
    z = _load(golden_dir, "thermal_g6")
    for ispace, tag in ((0, "wn"), (1, "wl")):
        W = z[f"{tag}_WAVE"]
        for cn in ("nadir_nosurf", "nadir_surf", "nadir_solar", "limb"):
            PR = z[f"{tag}_PRESS_limb"] if cn == "limb" else z[f"{tag}_PRESS_nadir"]
            TSURF, SOLA, EMIA = z[f"{tag}_{cn}_args"]
            s = eng.calc_thermal_emission_spectrum(ispace, W, z[f"{tag}_TAU"], None, z[f"{tag}_TEMP"], PR, TSURF,
                                                   z[f"{tag}_EMIS"], z[f"{tag}_SOL"], z[f"{tag}_REFL"], SOLA, EMIA)
            np.testing.assert_allclose(s, z[f"{tag}_{cn}_spec"], rtol=1e-11, err_msg=f"{tag} {cn}")
        s = eng.calc_thermal_emission_spectrum(ispace, W, z[f"{tag}_TAU"], z[f"{tag}_EMI"], z[f"{tag}_TEMP"],
                                               z[f"{tag}_PRESS_nadir"], 265.0, z[f"{tag}_EMIS"], z[f"{tag}_SOL"],
                                               z[f"{tag}_REFL"], 180.0, 20.0)
        np.testing.assert_allclose(s, z[f"{tag}_emi_spec"], rtol=1e-11)


def test_thermal_emission_gradient_seam_golden(eng, oracle, golden_dir):
    """Array-level calc_thermal_emission_spectrumg (:6380-6504) vs the reference (golden): nadir without / with a surface,
    limb, both ISPACE; then a larger random case vs the oracle's literal O(NPAR Li^2) restatement."""
    z = _load(golden_dir, "thermal_g6")
    for ispace, tag in ((0, "wn"), (1, "wl")):
        for cn in ("nadir_nosurf", "nadir_surf", "limb"):
            PR = z[f"{tag}_PRESS_limb"] if cn == "limb" else z[f"{tag}_PRESS_nadir"]
            TSURF = float(z[f"{tag}_{cn}_args"][0])
            NVMR = z[f"{tag}_dTAU"].shape[2] - 2
            sp, dsp, dts = eng.calc_thermal_emission_spectrumg(ispace, z[f"{tag}_WAVE"], z[f"{tag}_TAU"], z[f"{tag}_dTAU"], NVMR,
                                                               z[f"{tag}_TEMP"], PR, TSURF, z[f"{tag}_EMIS"])
            np.testing.assert_allclose(sp, z[f"{tag}_{cn}_specg"], rtol=1e-11, err_msg=f"{tag} {cn}")
            np.testing.assert_allclose(dts, z[f"{tag}_{cn}_dtsurf"], rtol=1e-11, atol=0, err_msg=f"{tag} {cn}")
            ref = z[f"{tag}_{cn}_dspecg"]
            scale = np.max(np.abs(ref), axis=3, keepdims=True) + 1e-300
            assert np.max(np.abs(dsp - ref) / scale) < 1e-9, f"{tag} {cn}"
    rng = np.random.default_rng(31)
    W, G, NPAR, Li = 37, 5, 7, 40
    WAVE = 300.0 + 11.0 * np.arange(W)
    TAU = 10.0 ** rng.uniform(-4, 0.5, (W, G, Li)); dTAU = rng.normal(size=(W, G, NPAR, Li)) * TAU[:, :, None, :]
    TEMP = np.linspace(120.0, 300.0, Li); PRESS = np.logspace(1, 5, Li); EMIS = rng.uniform(0.5, 1.0, W)
    got = eng.calc_thermal_emission_spectrumg(0, WAVE, TAU, dTAU, 4, TEMP, PRESS, 250.0, EMIS)
    ref = oracle.calc_thermal_emission_spectrumg(0, WAVE, TAU, dTAU, 4, TEMP, PRESS, 250.0, EMIS)
    np.testing.assert_allclose(got[0], ref[0], rtol=1e-11)
    np.testing.assert_allclose(got[2], ref[2], rtol=1e-11)
    scale = np.max(np.abs(ref[1]), axis=3, keepdims=True) + 1e-300
    assert np.max(np.abs(got[1] - ref[1]) / scale) < 1e-9


@pytest.mark.parametrize("W,G,S,L,zero,f32", [(130, 20, 8, 24, False, True), (70, 10, 4, 17, True, True),
                                                (64, 16, 5, 9, True, False), (33, 20, 20, 7, False, False)])
def test_cirsrad_vs_oracle(eng, oracle, W, G, S, L, zero, f32):
    """Fused CIRSrad (ILBL=0, thermal emission) for a 3-model batch vs the CPU oracle."""
    from archnemesis_dist_amd import synthetic as syn
    NP, NT = 9, 7
    _, delg = syn.gauss_legendre_01(G, as_float32=f32)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=5 + W, zero_low_g=zero)
    if f32:
        K = (K * 1e20).astype(np.float32).astype(np.float64) * 1e-20
    WAVE = 200.0 + 0.5 * np.arange(W)
    n = 3
    atm = syn.synth_atmosphere(L, S, seed=11, n_models=n, perturb=0.05)
    atm["amount"][0, 0, 2] = 0.0                      # exercise the skip rules
    atm["amount"][1, 1, 3] = 0.0
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L, emiss_ang=30.0)
    cont = syn.synth_continuum(W, L, n_models=n)
    EMTEMP = atm["lay_temp"][:, LAYINC[:, 0]][:, :, None]
    TSURF = np.array([-1.0, 300.0, 0.0])
    EMIS = np.linspace(0.7, 1.0, W)
    xfac = np.linspace(1.0, 2.0, W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    dims, mono = eng.ktable_info()
    assert dims == (W, G, NP, NT, S) and mono
    out = eng.cirsrad_ck_thermal(0, atm["lay_press_pa"], atm["lay_temp"], atm["amount"], cont, NLAYIN, LAYINC,
                                 SCALE, EMTEMP, TSURF, EMISSIVITY=EMIS, xfac=xfac)
    assert out.shape == (n, W, 1)
    for m in range(n):
        ref, tg = oracle.cirsrad_ck_thermal(0, K, PRESS, TEMP, WAVE, delg, atm["lay_press_pa"][m],
                                            atm["lay_temp"][m], atm["amount"][m], cont[m], NLAYIN, LAYINC, SCALE,
                                            EMTEMP[m], TSURF[m], EMISSIVITY=EMIS, xfac=xfac, return_taugas=True)
        np.testing.assert_allclose(out[m], ref, rtol=1e-10, atol=0, err_msg=f"model {m}")
        np.testing.assert_allclose(eng.get_taugas(L, m), tg, rtol=1e-10, atol=0)


def test_c2_full_size_properties(eng):
    """BASELINE C2 shape (10 000 wavenumbers x 100 layers x 8 gases, G=20): size-independent
    properties of the merge + RT, checked on every cell.
      * tau(g) non-decreasing in g (rank output is a sorted re-binning)
      * first moment conserved by every merge: sum_g delg*tau = sum_s sum_g delg*k_s*amount
        (exact-weight Gauss-Legendre, sum delg = 1 to rounding)
      * optically thick isothermal atmosphere radiates B(T); zero-opacity atmosphere over a
        surface radiates emissivity*B(TSURF)
    """
    from archnemesis_dist_amd import synthetic as syn
    W, G, S, L, NP, NT = 10000, 20, 8, 100, 20, 15
    _, delg = syn.gauss_legendre_01(G, as_float32=False)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S)
    WAVE = 200.0 + 0.1 * np.arange(W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    atm = syn.synth_atmosphere(L, S)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
    EMTEMP = atm["lay_temp"][:, LAYINC[:, 0]][:, :, None]
    spec = eng.cirsrad_ck_thermal(0, atm["lay_press_pa"], atm["lay_temp"], atm["amount"], None, NLAYIN, LAYINC,
                                  SCALE, EMTEMP, np.array([-1.0]))
    assert spec.shape == (1, W, 1) and np.all(np.isfinite(spec)) and np.all(spec > 0)
    tau = eng.get_taugas(L, 0)                                    # (W,G,L)
    assert np.all(np.diff(tau, axis=1) >= 0.0)
    k = eng.calc_k(atm["lay_press_pa"][0] / 101325.0, atm["lay_temp"][0])     # (W,G,L,S)
    mom_in = np.einsum("g,wgls,sl->wl", delg, k, atm["amount"][0])
    mom_out = np.einsum("g,wgl->wl", delg, tau)
    np.testing.assert_allclose(mom_out, mom_in, rtol=1e-11)
    # radiative limits through the array-level RT seam at full spectral size
    Li = 40
    T0 = 250.0
    c1, c2 = 1.1911e-12, 1.439
    bb = c1 * WAVE ** 3 / (np.exp(c2 * WAVE / T0) - 1.0)
    thick = np.full((W, G, Li), 5.0)
    PR = np.logspace(1, 5, Li)
    z = np.zeros(W)
    s = eng.calc_thermal_emission_spectrum(0, WAVE, thick, None, np.full(Li, T0), PR, -1.0, z, z, z, 180.0, 0.0)
    np.testing.assert_allclose(s, np.repeat(bb[:, None], G, 1), rtol=1e-12)
    em = np.linspace(0.5, 1.0, W)
    s = eng.calc_thermal_emission_spectrum(0, WAVE, np.zeros((W, G, Li)), None, np.full(Li, 100.0), PR, T0, em, z,
                                           z, 180.0, 0.0)
    np.testing.assert_allclose(s, np.repeat((em * bb)[:, None], G, 1), rtol=1e-12)


@pytest.mark.parametrize("name", CK_CASES)
def test_k_overlapg_golden(eng, golden_dir, name):
    z = _load(golden_dir, name)
    taug, dk = eng.k_overlapg(z["DELG"], z["kg"], z["dkdT"], z["amount"])
    np.testing.assert_allclose(taug, z["taug"], rtol=1e-11, atol=0)
    # tie-order caveat of rankg gradients (see tests/test_oracle_golden.py): 1e-5 of the per-cell
    # column scale where exact ties exist, 1e-10 otherwise; the Jacobian contract is 1e-4
    scale = np.abs(z["dk"]).max(axis=1, keepdims=True) + 1e-300
    tol = 1e-10 if name.endswith("nozero") else 1e-5
    assert np.max(np.abs(dk - z["dk"]) / scale) < tol


@pytest.mark.parametrize("W,G,S,L,f32,NDUST", [(96, 20, 8, 20, True, 1), (70, 10, 3, 9, False, 1), (40, 16, 12, 6, False, 1),
                                               (24, 8, 24, 5, False, 60)])
def test_cirsradg_vs_oracle(eng, oracle, W, G, S, L, f32, NDUST):
    """Fused CIRSrad(return_grad=True) for a 2-model batch vs the CPU oracle (literal O(Li^2) recursion).  Last case: 24
    spectroscopic gases and NPAR = 88 -- beyond the 20 gases / 64 parameters the gradient path was capped at until round 3."""
    from archnemesis_dist_amd import synthetic as syn
    NP, NT = 8, 6
    _, delg = syn.gauss_legendre_01(G, as_float32=f32)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=9 + W)
    WAVE = 150.0 + 0.7 * np.arange(W)
    n = 2
    atm = syn.synth_atmosphere(L, S, seed=4, n_models=n, perturb=0.05)
    atm["amount"][0, 1, 3] = 0.0
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L, emiss_ang=25.0)
    cont = syn.synth_continuum(W, L, n_models=n)
    NVMR = S + 2                                # more atmospheric gases than spectroscopic ones
    NPAR = NVMR + 2 + NDUST
    rng = np.random.default_rng(3)
    igas_map = rng.permutation(NVMR)[:S].astype(np.int32)
    dcont = cont[:, :, None, :] * rng.uniform(0.0, 1e-22, size=(n, W, NPAR, L))
    EMTEMP = atm["lay_temp"][:, LAYINC[:, 0]][:, :, None]
    TSURF = np.array([-1.0, 280.0])
    EMIS = np.linspace(0.8, 1.0, W)
    xfac = np.linspace(1.0, 1.5, W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    spec, dspec, dts = eng.cirsradg_ck_thermal(0, atm["lay_press_pa"], atm["lay_temp"], atm["amount"], cont, dcont,
                                               NVMR, NPAR, igas_map, NLAYIN, LAYINC, SCALE, EMTEMP, TSURF,
                                               EMISSIVITY=EMIS, xfac=xfac)
    for m in range(n):
        rs, rd, rt = oracle.cirsradg_ck_thermal(0, K, PRESS, TEMP, WAVE, delg, atm["lay_press_pa"][m],
                                                atm["lay_temp"][m], atm["amount"][m], cont[m], dcont[m], NVMR, NPAR,
                                                igas_map, NLAYIN, LAYINC, SCALE, EMTEMP[m], TSURF[m], EMISSIVITY=EMIS,
                                                xfac=xfac)
        np.testing.assert_allclose(spec[m], rs, rtol=1e-10)
        np.testing.assert_allclose(dts[m], rt, rtol=1e-10, atol=0)
        scale = np.abs(rd).max(axis=(0, 2, 3), keepdims=True) + 1e-300
        assert np.max(np.abs(dspec[m] - rd) / scale) < 1e-9, m


def test_analytic_jacobian_agrees_with_finite_differences(eng):
    """The two Jacobian routes of the path against each other (KK vs finite differences to 1e-4): central
    differences of the forward CIRSrad in the gas amounts, the layer temperatures and the surface temperature vs the
    analytic dSPECOUT / dTSURF of CIRSrad(return_grad=True).  Amount slots are d/dAMOUNT in m-2 (:3868-3870): 1e-4 x the
    derivative in the engine's cm-2 columns."""
    from archnemesis_dist_amd import synthetic as syn
    W, G, S, L, NP, NT = 128, 10, 3, 8, 8, 6
    _, delg = syn.gauss_legendre_01(G, as_float32=False)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=21)
    WAVE = 400.0 + 1.5 * np.arange(W)
    atm = syn.synth_atmosphere(L, S, seed=5)
    lp, lt, am = atm["lay_press_pa"][0], atm["lay_temp"][0].copy(), atm["amount"][0].copy()
    am *= 40.0                                                    # optical depths of order 1: every term matters
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L, emiss_ang=30.0)
    TSURF, EMIS = 300.0, np.linspace(0.85, 1.0, W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    NVMR, NPAR = S, S + 2
    igas = np.arange(S, dtype=np.int32)

    def forward(amount, temp, tsurf=TSURF):
        emt = temp[LAYINC[:, 0]][:, None]
        return np.squeeze(eng.cirsrad_ck_thermal(0, lp, temp, amount, None, NLAYIN, LAYINC, SCALE, emt, tsurf, EMISSIVITY=EMIS))

    spec, dspec, dts = eng.cirsradg_ck_thermal(0, lp, lt, am, None, None, NVMR, NPAR, igas, NLAYIN, LAYINC, SCALE,
                                               lt[LAYINC[:, 0]][:, None], TSURF, EMISSIVITY=EMIS)
    dspec = np.asarray(dspec).reshape(W, NPAR, L, 1)[..., 0]       # (W, NPAR, path position)
    pos = {int(l): j for j, l in enumerate(LAYINC[:, 0])}          # path position of each layer (nadir: every layer once)
    top = np.abs(dspec).max()
    worst = 0.0
    for l in (0, 3, L - 1):
        for g in range(S):
            h = 1e-4 * am[g, l]
            ap, an = am.copy(), am.copy(); ap[g, l] += h; an[g, l] -= h
            fd = (forward(ap, lt) - forward(an, lt)) / (2 * h) * 1.0e-4
            ana = dspec[:, g, pos[l]]
            worst = max(worst, np.max(np.abs(fd - ana)) / np.abs(ana).max())
        h = 1e-4 * lt[l]
        tp, tn = lt.copy(), lt.copy(); tp[l] += h; tn[l] -= h
        fd = (forward(am, tp) - forward(am, tn)) / (2 * h)
        ana = dspec[:, NVMR, pos[l]]
        worst = max(worst, np.max(np.abs(fd - ana)) / np.abs(ana).max())
    h = 1e-3
    fd = (forward(am, lt, TSURF + h) - forward(am, lt, TSURF - h)) / (2 * h)
    worst = max(worst, np.max(np.abs(fd - np.squeeze(dts))) / np.abs(dts).max())
    assert top > 0 and worst < 1e-5, worst


@pytest.mark.parametrize("name", ["ms_nmu5_hg_ray", "ms_nmu5_tab_lambert", "ms_nmu16_tab_ray", "ms_nmu5_lookup",
                                  "ms_nmu5_lookup_lambert", "ms_nmu16_lookup_lambert"])
def test_scloud11wave_core_golden(eng, golden_dir, name):
    """Doubling/adding core vs the reference's scloud11wave_core (golden) -- contract 1e-6.  Look-down and look-up
    geometry (the latter without and with a reflecting lower boundary, i.e. through idown)."""
    from test_ms_oracle import ms_args
    z = _load(golden_dir, name)
    rad = eng.scloud11wave_core(*ms_args(z))
    assert rad.shape == z["rad"].shape
    np.testing.assert_allclose(rad, z["rad"], rtol=1e-8)


def test_scloud11wave_core_vs_oracle_random(eng, oracle, golden_dir):
    """A wider random stack (more waves/g/layers than the golden) against the CPU oracle."""
    from test_ms_oracle import ms_args
    z = dict(_load(golden_dir, "ms_nmu5_hg_ray"))
    rng = np.random.default_rng(5)
    W, G, L = 40, 3, 12
    nmu = z["mu1"].size; ncont = z["phasarr"].shape[0]; nth = z["phasarr"].shape[3]
    z["vwaves"] = 400.0 + 10.0 * np.arange(W)
    ph = np.zeros((ncont, W, 2, nth)); ph[:, :, 1, :] = z["phasarr"][0, 0, 1, :]
    ph[:, :, 0, 0] = rng.uniform(0.5, 0.95, (ncont, W)); ph[:, :, 0, 1] = rng.uniform(0.2, 0.85, (ncont, W))
    ph[:, :, 0, 2] = rng.uniform(-0.6, -0.05, (ncont, W))
    z["phasarr"] = ph
    taus = 10.0 ** rng.uniform(-5, 1.3, size=(W, G, L))
    tauray = 10.0 ** rng.uniform(-6, -2, size=(W, L)); tauscat = 10.0 ** rng.uniform(-5, 0, size=(W, L))
    taus = np.maximum(taus, (tauscat + tauray)[:, None, :] * 1.01)
    z["taus"] = taus; z["tauray"] = tauray
    z["omegas_s"] = np.broadcast_to((tauray + tauscat)[:, None, :], taus.shape) / taus
    fr = rng.uniform(0.1, 1.0, size=(W, ncont, L)); z["lfrac"] = fr / fr.sum(axis=1, keepdims=True)
    z["bnu"] = 10.0 ** rng.uniform(-8, -6, size=(W, L)); z["radg"] = 10.0 ** rng.uniform(-8, -6, size=(W, nmu))
    z["solar"] = 10.0 ** rng.uniform(-9, -8, W); z["brdf_matrix"] = np.zeros((W, nmu, nmu, int(z["nf"]) + 1))
    rad = eng.scloud11wave_core(*ms_args(z))
    ref = oracle.scloud11wave_core(*ms_args(z))
    np.testing.assert_allclose(rad, ref, rtol=1e-8)
    # the same stack seen from below, with a reflecting surface under it (idown)
    z["emiss_angs"] = 180.0 - np.asarray(z["emiss_angs"])
    z["lowbc"] = 1
    z["brdf_matrix"] = rng.uniform(0.02, 0.2, size=z["brdf_matrix"].shape) / np.pi
    rad = eng.scloud11wave_core(*ms_args(z))
    ref = oracle.scloud11wave_core(*ms_args(z))
    np.testing.assert_allclose(rad, ref, rtol=1e-8)
    with pytest.raises(ValueError):      # mixed geometries: the reference's ValueError (:776)
        z["emiss_angs"] = np.array([20.0, 160.0])
        eng.scloud11wave_core(*ms_args(z))


def test_scloud11wave_core_nearly_conservative_thick_layers_vs_oracle(eng, oracle, golden_dir):
    """Optically thick, almost conservatively scattering layers at 16 streams (20 and more doublings per layer, reflection
    operators of norm close to 1): the regime where the product-form Neumann series of inv(E - r r) needs most of its 12
    squarings.  The oracle inverts by elimination throughout."""
    from test_ms_oracle import ms_args
    z = dict(_load(golden_dir, "ms_nmu16_tab_ray"))
    rng = np.random.default_rng(8)
    W, G, L = 3, 2, 3
    nmu = z["mu1"].size; ncont = z["phasarr"].shape[0]
    z["vwaves"] = 500.0 + 10.0 * np.arange(W)
    z["phasarr"] = np.ascontiguousarray(np.broadcast_to(z["phasarr"][:, :1], (ncont, W) + z["phasarr"].shape[2:]))
    taus = np.empty((W, G, L)); taus[:, :, 0] = 0.3; taus[:, :, 1] = rng.uniform(300.0, 3000.0, (W, G)); taus[:, :, 2] = 40.0
    tauray = np.full((W, L), 1e-4)
    z["taus"] = taus; z["tauray"] = tauray
    om = np.empty((W, G, L)); om[:, :, 0] = 0.6; om[:, :, 1] = 1.0 - 1e-7; om[:, :, 2] = 0.9999
    z["omegas_s"] = om
    fr = rng.uniform(0.1, 1.0, size=(W, ncont, L)); z["lfrac"] = fr / fr.sum(axis=1, keepdims=True)
    z["bnu"] = 10.0 ** rng.uniform(-8, -6, size=(W, L)); z["radg"] = 10.0 ** rng.uniform(-8, -6, size=(W, nmu))
    z["solar"] = 10.0 ** rng.uniform(-9, -8, W); z["brdf_matrix"] = np.zeros((W, nmu, nmu, int(z["nf"]) + 1))
    rad = eng.scloud11wave_core(*ms_args(z))
    ref = oracle.scloud11wave_core(*ms_args(z))
    assert np.all(np.isfinite(rad))
    np.testing.assert_allclose(rad, ref, rtol=1e-6)      # inv(E - r r) is ill-conditioned here: the contract, not 1e-8


@pytest.mark.parametrize("name", ["lbl_tab", "lbl_tab_t2d_f32"])
def test_calc_klbl_golden(eng, golden_dir, name):
    z = _load(golden_dir, name)
    eng.upload_lbltable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"])
    k = eng.calc_klbl(z["press"], z["temp"])
    np.testing.assert_allclose(k, z["k"], rtol=1e-12, atol=0)
    kg, dk = eng.calc_klbl(z["press"], z["temp"], grad=True)
    np.testing.assert_allclose(kg, z["kg"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(dk, z["dkdT"], rtol=1e-10, atol=0)


def test_cirsrad_lbl_tables_vs_oracle(eng, oracle):
    """Fused CIRSrad for ILBL = LINE_BY_LINE_TABLES (NG=1), forward and analytic gradient, against the oracle
    composed the way calculate_gaseous_line_opacity does it (:3795-3817)."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(12)
    W, NP, NT, S, L = 300, 7, 6, 3, 15
    PRESS = np.logspace(-6, 1.1, NP); TEMP = np.linspace(80.0, 420.0, NT)
    K = 10.0 ** rng.uniform(-27, -20, size=(W, NP, NT, S))
    WAVE = 2000.0 + 0.01 * np.arange(W)
    atm = syn.synth_atmosphere(L, S, seed=3)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L, 15.0)
    cont = syn.synth_continuum(W, L)
    EMTEMP = atm["lay_temp"][:, LAYINC[:, 0]][:, :, None]
    lp, lt, am = atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0]
    eng.upload_lbltable(K, PRESS, TEMP, WAVE)
    out = eng.cirsrad_ck_thermal(0, lp, lt, am, cont[0], NLAYIN, LAYINC, SCALE, EMTEMP[0], 250.0, EMISSIVITY=np.ones(W))
    k, dkdT = oracle.calc_klbl(K, PRESS, TEMP, lp / 101325.0, lt, grad=True)              # (W,L,S)
    tau = np.zeros((W, L))
    for s in range(S):
        tau = tau + k[:, :, s] * am[s][None, :]
    tautot = tau + cont[0]
    path = (tautot[:, LAYINC[:, 0]] * SCALE[:, 0])[:, None, :]
    z = np.zeros(W)
    ref = oracle.calc_thermal_emission_spectrum(0, WAVE, path, None, EMTEMP[0][:, 0], lp[LAYINC[:, 0]], 250.0,
                                                np.ones(W), z, z, 180.0, 180.0)
    np.testing.assert_allclose(out[:, 0], ref[:, 0], rtol=1e-11)
    # gradient
    NVMR, NPAR = S, S + 2
    igas_map = np.arange(S, dtype=np.int32)
    spec, dspec, dts = eng.cirsradg_ck_thermal(0, lp, lt, am, cont[0], None, NVMR, NPAR, igas_map, NLAYIN, LAYINC, SCALE,
                                               EMTEMP[0], 250.0, EMISSIVITY=np.ones(W))
    dtau = np.zeros((W, 1, NPAR, L))
    for s in range(S):
        dtau[:, 0, s, :] = k[:, :, s] * 1.0e-4
        dtau[:, 0, NVMR, :] += dkdT[:, :, s] * am[s][None, :]
    dpath = dtau[:, :, :, LAYINC[:, 0]] * SCALE[:, 0]
    rs, rd, rt = oracle.calc_thermal_emission_spectrumg(0, WAVE, path, dpath, NVMR, EMTEMP[0][:, 0], lp[LAYINC[:, 0]],
                                                        250.0, np.ones(W))
    np.testing.assert_allclose(spec[:, 0], rs[:, 0], rtol=1e-11)
    np.testing.assert_allclose(dts[:, 0], rt[:, 0], rtol=1e-11)
    ref_d = rd[:, 0]                                                       # (W,NPAR,Li)
    scale = np.abs(ref_d).max(axis=(0, 2), keepdims=True) + 1e-300
    assert np.max(np.abs(dspec[:, :, :, 0] - ref_d) / scale) < 1e-10


def test_lbl_runtime_golden(eng, golden_dir):
    """Runtime line-by-line absorption vs the reference's add_line_set_monochromatic_absorption (golden)."""
    from test_lbl_oracle import lbl_case
    z = _load(golden_dir, "lbl_lines")
    N = z["nu"].size
    for name in z["names"]:
        a = lbl_case(z, str(name))
        out = np.zeros(z["wn_grid"].size); store = np.empty((4, N))
        eng.add_line_set_monochromatic_absorption(
            z["wn_grid"], a["lineshape_id"], a["t_calc"], float(z["t_ref"]), a["p_calc"], float(z["p_ref"]), a["q_ratio"],
            float(z["isotopic_abundance"]), float(z["isotopic_mass"]), z["mol_mix_frac"], z["broadening_params"], z["nu"],
            z["sw"], z["e_lower"], z["stim_ref"], out, store, a["s_floor"], a["wn_calc_window"], a["wn_approx_window"])
        np.testing.assert_allclose(store, z[str(name) + "_store"], rtol=1e-12, err_msg=str(name))
        ref = z[str(name) + "_k"]
        np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-300, err_msg=str(name))


def test_lbl_runtime_batched_unsorted_vs_oracle(eng, oracle, golden_dir):
    """L = 3 (T,p) points in one call, lines handed over in random order, accumulation into a non-zero `out`."""
    z = _load(golden_dir, "lbl_lines")
    rng = np.random.default_rng(8)
    perm = rng.permutation(z["nu"].size)
    nu, sw, el, sr, bp = z["nu"][perm], z["sw"][perm], z["e_lower"][perm], z["stim_ref"][perm], z["broadening_params"][:, perm]
    wn = z["wn_grid"][::2].copy()
    t = np.array([160.0, 230.0, 300.0]); p = np.array([1e-3, 0.2, 2.0]); q = np.array([1.9, 1.2, 0.97])
    out = rng.uniform(0, 1e-22, size=(3, wn.size)); ref = out.copy()
    eng.add_line_set_monochromatic_absorption(wn, 0, t, 296.0, p, 1.0, q, 0.9, 28.0, z["mol_mix_frac"], bp, nu, sw, el, sr, out)
    srt = np.argsort(nu, kind="stable")
    for l in range(3):
        oracle.add_line_set_monochromatic_absorption(wn, 0, t[l], 296.0, p[l], 1.0, q[l], 0.9, 28.0, z["mol_mix_frac"],
                                                     bp[:, srt], nu[srt], sw[srt], el[srt], sr[srt], ref[l])
    np.testing.assert_allclose(out, ref, rtol=1e-10)


def test_lbl_runtime_large_grid_properties(eng):
    """C5-like sizes the oracle cannot reach in seconds (2e5 grid points x 2e4 lines x 4 layers): the spectrum of a line
    list is the sum of the spectra of its parts (accumulation into `out` in line order, so exactly: two calls on the same
    `out` vs one call), scales linearly with the isotopic abundance, and is non-negative."""
    rng = np.random.default_rng(55)
    nw, N, L, M = 200000, 20000, 4, 2
    wn = 2000.0 + 1e-3 * np.arange(nw)
    nu = np.sort(rng.uniform(1995.0, 2205.0, N))
    sw = 10.0 ** rng.uniform(-28, -20, N); el = rng.uniform(0, 3000, N)
    sr = 1.0 - np.exp(-1.4387769 * nu / 296.0)
    bp = np.stack([rng.uniform(0.02, 0.1, N), rng.uniform(0.5, 0.8, N), rng.uniform(-0.01, 0.01, N),
                   rng.uniform(0.05, 0.12, N), rng.uniform(0.5, 0.8, N), rng.uniform(-0.01, 0.01, N)])
    mmf = np.array([0.9, 0.1])
    t = np.array([150.0, 200.0, 250.0, 300.0]); p = np.array([1e-4, 1e-2, 0.3, 1.0]); q = np.array([2.1, 1.5, 1.1, 0.98])
    run = lambda sel, out, iso=0.9: eng.add_line_set_monochromatic_absorption(
        wn, 0, t, 296.0, p, 1.0, q, iso, 28.0, mmf, bp[:, sel], nu[sel], sw[sel], el[sel], sr[sel], out)
    full = np.zeros((L, nw)); run(slice(None), full)
    half = N // 2
    parts = np.zeros((L, nw)); run(slice(0, half), parts); run(slice(half, N), parts)
    assert np.array_equal(full, parts)                                # same additions in the same (ascending line) order
    twice = np.zeros((L, nw)); run(slice(None), twice, iso=1.8)
    np.testing.assert_allclose(twice, 2.0 * full, rtol=1e-14)
    assert full.min() >= 0.0 and full.max() > 0.0


@pytest.mark.parametrize("case", ["cg_nadir", "cg_slant", "mid_slant", "cg_dustunits"])
def test_layer_average_golden(eng, golden_dir, case):
    """Layer_0.layer_average (Curtis-Godson / mid-path) vs the reference (golden), single state and a batch."""
    from test_layer_oracle import NAMES, CASES
    z = _load(golden_dir, "layer_average")
    kw = dict(CASES[case]); du = kw.pop("dust_units", False)
    args = dict(LAYHT=-6.0e4, NINT=101, DUST_UNITS=np.array([-1, 0]) if du else None, XMOLWT=z["XMOLWT"] if du else None, **kw)
    r = eng.layer_average(float(z["RADIUS"]), z["H"], z["P"], z["T"], None, z["VMR"], z["DUST"], z["PARAH2"], z["split1_BASEH"],
                          z["split1_BASEP"], **args)
    for n, v in zip(NAMES, r):
        np.testing.assert_allclose(v, z[f"{case}_{n}"], rtol=1e-10, err_msg=n)
    # batch of 3 states: state 1 has a warmer profile; state 0 and 2 must reproduce the golden
    T3 = np.stack([z["T"], z["T"] * 1.05, z["T"]])
    rep = lambda a: np.repeat(np.asarray(a)[None], 3, 0)
    rb = eng.layer_average(float(z["RADIUS"]), rep(z["H"]), rep(z["P"]), T3, None, rep(z["VMR"]), rep(z["DUST"]),
                           rep(z["PARAH2"]), z["split1_BASEH"], None, **args)
    for n, v in zip(NAMES, rb):
        np.testing.assert_allclose(v[0], z[f"{case}_{n}"], rtol=1e-10, err_msg=n)
        np.testing.assert_allclose(v[2], z[f"{case}_{n}"], rtol=1e-10, err_msg=n)
    assert np.all(rb[2][1] > rb[2][0])          # TEMP of the warmed state


@pytest.mark.parametrize("case", ["cg_nadir", "cg_slant", "mid_slant", "cg_dustunits"])
def test_layer_average_batch_shares_unchanged_layers_bit_for_bit(eng, golden_dir, case):
    """A Jacobian's states differ from state 0 at one level: the batched call takes state 0's layers where a layer reads only
    unchanged levels -- every state equals its own single-state call bit for bit, whatever was perturbed (T, P, H, a gas,
    the dust, para-H2, the molecular weight, the layer bases)."""
    from test_layer_oracle import CASES
    z = _load(golden_dir, "layer_average")
    kw = dict(CASES[case]); du = kw.pop("dust_units", False)
    NPRO = z["H"].size
    base = dict(H=z["H"], P=z["P"], T=z["T"], VMR=z["VMR"], DUST=z["DUST"], PARAH2=z["PARAH2"], XMOLWT=z["XMOLWT"],
                BASEH=z["split1_BASEH"])
    states = [dict(base)]
    def bump(key, lev, col=None):
        a = np.array(base[key], dtype=float)
        if col is None: a[lev] *= 1.05
        else: a[lev, col] *= 1.05
        states.append(dict(base, **{key: a}))
    bump("T", 0); bump("T", NPRO // 2); bump("T", NPRO - 1); bump("P", 3); bump("VMR", NPRO // 3, 1); bump("DUST", 5, 0)
    bump("PARAH2", 7); bump("XMOLWT", NPRO // 2); bump("BASEH", 4)
    hs = np.array(base["H"], dtype=float); hs[NPRO - 1] += 10.0; states.append(dict(base, H=hs))     # the top level enters SMAX
    states.append(dict(base))                                                                          # an unperturbed copy
    st = lambda k: np.stack([np.asarray(s_[k], dtype=float) for s_ in states])
    args = dict(LAYHT=-6.0e4, NINT=101, DUST_UNITS=np.array([-1, 0]) if du else None, **kw)
    rb = eng.layer_average(float(z["RADIUS"]), st("H"), st("P"), st("T"), None, st("VMR"), st("DUST"), st("PARAH2"), st("BASEH"),
                           None, XMOLWT=st("XMOLWT") if du else None, **args)
    nchanged = []
    for i, s_ in enumerate(states):
        r1 = eng.layer_average(float(z["RADIUS"]), s_["H"], s_["P"], s_["T"], None, s_["VMR"], s_["DUST"], s_["PARAH2"], s_["BASEH"],
                               None, XMOLWT=s_["XMOLWT"] if du else None, **args)
        for vb, v1 in zip(rb, r1):
            assert np.array_equal(vb[i], v1), (case, i)
        nchanged.append(int(np.sum(np.any(rb[4][i] != rb[4][0], axis=-1) | (rb[2][i] != rb[2][0]) | (rb[8][i] != rb[8][0]))))
    assert nchanged[0] == 0 and nchanged[-1] == 0
    assert 1 <= nchanged[2] <= 8, nchanged            # one temperature level reaches a few layers only


def test_gradient_maps_golden(eng, golden_dir):
    """ForwardModel_0.map2pro / map2xvec (matrix-core GEMM) vs the reference (golden): default and explicit INCPAR
    (para-H2 slot = previous parameter's product), host-pointer and device-chained inputs."""
    z = _load(golden_dir, "gradient_maps")
    W, NVMR, NDUST, NPRO, NPATH, NX = (int(v) for v in z["dims"])
    close = lambda a, b: np.testing.assert_allclose(a, b, rtol=0, atol=1e-13 * np.max(np.abs(b)))
    a = eng.map2pro(z["dSPECIN"], W, NVMR, NDUST, NPRO, NPATH, z["NLAYIN"], z["LAYINC"], z["DTE"], z["DAM"], z["DCO"])
    for par in range(NVMR + 2 + NDUST):
        close(a[:, par], z["pro_all"][:, par])
    x = eng.map2xvec(a, W, NVMR, NDUST, NPRO, NPATH, NX, z["xmap"])          # chained: a's device copy
    close(x, z["xvec_all"])
    x2 = eng.map2xvec(z["pro_all"].copy(), W, NVMR, NDUST, NPRO, NPATH, NX, z["xmap"])   # uploaded
    close(x2, z["xvec_all"])
    b = eng.map2pro(z["dSPECIN"], W, NVMR, NDUST, NPRO, NPATH, z["NLAYIN"], z["LAYINC"], z["DTE"], z["DAM"], z["DCO"],
                    INCPAR=list(z["incpar"]))
    for par in range(NVMR + 2 + NDUST):
        close(b[:, par], z["pro_inc"][:, par])
    assert np.array_equal(b[:, 6], b[:, 5]) and not np.any(b[:, 1])
    with pytest.raises(UnboundLocalError):
        eng.map2pro(z["dSPECIN"], W, NVMR, NDUST, NPRO, NPATH, z["NLAYIN"], z["LAYINC"], z["DTE"], z["DAM"], z["DCO"], INCPAR=[6, 0])


def test_gradient_maps_vs_oracle_chained_from_cirsradg(eng, oracle):
    """nemesisfmg's tail on the device: CIRSrad(return_grad=True) -> map2pro -> map2xvec with the intermediate arrays
    taken from HBM (the arrays handed back are recognised), against the oracle fed the same host arrays."""
    from archnemesis_dist_amd import synthetic as syn
    W, G, S, L, NP, NT = 300, 8, 3, 24, 6, 5
    _, delg = syn.gauss_legendre_01(G, True)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=11)
    eng.upload_ktable(K, PRESS, TEMP, 100.0 + np.arange(W), delg)
    atm = syn.synth_atmosphere(L, S)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
    EMTEMP = atm["lay_temp"][0][LAYINC[:, 0]][:, None]
    NVMR, NDUST = S, 1
    NPAR = NVMR + 2 + NDUST
    spec, dspec, dts = eng.cirsradg_ck_thermal(0, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0], None, None,
                                               NVMR, NPAR, np.arange(S, dtype=np.int32), NLAYIN, LAYINC, SCALE, EMTEMP, -1.0)
    rng = np.random.default_rng(3)
    NPRO, NX = 40, 17
    DTE, DAM, DCO = (rng.uniform(0, 1, (L, NPRO)) for _ in range(3))
    xmap = rng.normal(size=(NX, NPAR, NPRO))
    pro = eng.map2pro(dspec, W, NVMR, NDUST, NPRO, 1, NLAYIN, LAYINC, DTE, DAM, DCO)
    xv = eng.map2xvec(pro, W, NVMR, NDUST, NPRO, 1, NX, xmap)
    pro_o = oracle.map2pro(dspec, W, NVMR, NDUST, NPRO, 1, NLAYIN, LAYINC, DTE, DAM, DCO)
    xv_o = oracle.map2xvec(pro_o, W, NVMR, NDUST, NPRO, 1, NX, xmap)
    for par in range(NPAR):
        if np.any(pro_o[:, par]):
            np.testing.assert_allclose(pro[:, par], pro_o[:, par], rtol=0, atol=1e-12 * np.max(np.abs(pro_o[:, par])))
    np.testing.assert_allclose(xv, xv_o, rtol=0, atol=1e-12 * np.max(np.abs(xv_o)))
    # arrays with a device twin are read-only; a modified copy is NOT mistaken for the device-resident array
    with pytest.raises(ValueError):
        dspec[0, 0, 0, 0] = 1.0
    d2 = dspec.copy(); d2[:, 0] *= 2.0
    pro2 = eng.map2pro(d2, W, NVMR, NDUST, NPRO, 1, NLAYIN, LAYINC, DTE, DAM, DCO)
    np.testing.assert_allclose(pro2[:, 0], 2.0 * pro[:, 0], rtol=1e-12)


@pytest.mark.parametrize("case", ["cg_nadir", "cg_slant", "mid_slant", "cg_dustunits"])
def test_layer_averageg_golden(eng, golden_dir, case):
    """Layer_0.layer_averageg (:1032) vs the reference (golden): layer values through `interpg` and the DTE/DAM/DCO/DPH
    matrices; a batch of states; the reference's two ValueErrors."""
    from test_layer_oracle import GNAMES, CASES
    z = _load(golden_dir, "layer_averageg")
    kw = dict(CASES[case]); du = kw.pop("dust_units", False)
    args = dict(LAYHT=-6.0e4, NINT=101, DUST_UNITS=np.array([-1, 0]) if du else None, XMOLWT=z["XMOLWT"] if du else None, **kw)
    a = (float(z["RADIUS"]), z["H"], z["P"], z["T"], None, z["VMR"], z["DUST"], z["PARAH2"], z["split1_BASEH"], z["split1_BASEP"])
    r = eng.layer_averageg(*a, **args)
    for n, v in zip(GNAMES, r):
        ref = z[f"{case}_{n}"]
        np.testing.assert_allclose(v, ref, rtol=1e-10, atol=1e-13 * np.max(np.abs(ref)), err_msg=n)
    rep = lambda x: np.repeat(np.asarray(x)[None], 2, 0)
    rb = eng.layer_averageg(float(z["RADIUS"]), rep(z["H"]), rep(z["P"]), np.stack([z["T"] * 1.02, z["T"]]), None, rep(z["VMR"]),
                            rep(z["DUST"]), rep(z["PARAH2"]), z["split1_BASEH"], None, **args)
    for n, v in zip(GNAMES, rb):
        ref = z[f"{case}_{n}"]
        np.testing.assert_allclose(v[1], ref, rtol=1e-10, atol=1e-13 * np.max(np.abs(ref)), err_msg=n)
    if case == "cg_nadir":
        with pytest.raises(ValueError):
            eng.layer_averageg(*a, LAYHT=-6.0e4, NINT=100, LAYINT=1)
        with pytest.raises(ValueError):
            eng.layer_averageg(*a, LAYHT=-6.0e4, NINT=101, LAYINT=0, DUST_UNITS=np.array([-1, 0]), XMOLWT=z["XMOLWT"])


@pytest.mark.parametrize("nint", [100, 2, 4])
def test_layer_average_even_nint_golden(eng, golden_dir, nint):
    from test_layer_oracle import NAMES
    z = _load(golden_dir, "layer_average"); e = _load(golden_dir, "layer_average_even_nint")
    r = eng.layer_average(float(z["RADIUS"]), z["H"], z["P"], z["T"], None, z["VMR"], z["DUST"], z["PARAH2"], z["split1_BASEH"],
                          z["split1_BASEP"], LAYANG=35.0, LAYINT=1, LAYHT=-6.0e4, NINT=nint)
    for n, v in zip(NAMES, r):
        np.testing.assert_allclose(v, e[f"nint{nint}_{n}"], rtol=1e-10, err_msg=n)


@pytest.mark.parametrize("ishape", range(5))
def test_lblconv_golden(eng, golden_dir, ishape):
    """ILS convolution vs the reference (golden), every ISHAPE incl. the ones whose result is 0/0 in the reference."""
    from test_conv_oracle import close_nan
    z = _load(golden_dir, "ils_conv")
    nw, nc, fw = z["vwave"].size, z["vconv"].size, float(z["fwhm"])
    close_nan(eng.lblconv(nw, z["vwave"], z["y"], nc, z["vconv"], ishape, fw), z[f"conv_{ishape}"], 1e-12)
    yo, go = eng.lblconvg(nw, z["vwave"], z["y"], z["dydx"], nc, z["vconv"], ishape, fw)
    close_nan(yo, z[f"convg_{ishape}_y"], 1e-12)
    close_nan(go, z[f"convg_{ishape}_g"], 1e-11)


def test_lblconv_fil_golden_and_large_vs_oracle(eng, oracle, golden_dir):
    from test_conv_oracle import close_nan
    z = _load(golden_dir, "ils_conv")
    nw, nc = z["vwave"].size, z["vconv"].size
    close_nan(eng.lblconv_fil(nw, z["vwave"], z["y"], nc, z["vconv"], z["nfil"], z["vfil"], z["afil"]), z["fil"], 1e-12)
    yo, go = eng.lblconvg_fil(nw, z["vwave"], z["y"], z["dydx"], nc, z["vconv"], z["nfil"], z["vfil"], z["afil"])
    close_nan(yo, z["filg_y"], 1e-12)
    close_nan(go, z["filg_g"], 1e-11)
    # a larger case (windows of several hundred points, 200 gradient columns) against the oracle
    rng = np.random.default_rng(8)
    nw, nx, nc = 20000, 200, 40
    vw = 1000.0 + 1e-3 * np.arange(nw)
    y = rng.uniform(1, 2, nw); dy = rng.normal(size=(nw, nx))
    vc = np.linspace(1001.0, 1019.0, nc)
    for ishape in (1, 2, 3):
        yo, go = eng.lblconvg(nw, vw, y, dy, nc, vc, ishape, 0.4)
        yr, gr = oracle.lblconv(nw, vw, y, nc, vc, ishape, 0.4, dydx=dy)
        np.testing.assert_allclose(yo, yr, rtol=1e-12)
        np.testing.assert_allclose(go, gr, rtol=0, atol=1e-12 * np.max(np.abs(gr)))
    with pytest.raises(ValueError):
        eng.lblconv(nw, vw[::-1].copy(), y, nc, vc, 1, 0.4)


@pytest.mark.parametrize("ishape", range(5))
def test_lblconv_ngeom_golden(eng, golden_dir, ishape):
    """The *_ngeom kernels (several geometries on one grid) vs the reference, incl. their own Hamming window and the 0/0
    results."""
    from test_conv_oracle import close_nan
    z = _load(golden_dir, "ils_conv")
    nw, nc, fw = z["vwave"].size, z["vconv"].size, float(z["fwhm"])
    close_nan(eng.lblconv_ngeom(nw, z["vwave"], z["y_ngeom"], nc, z["vconv"], ishape, fw), z[f"ngconv_{ishape}"], 1e-12)
    yo, go = eng.lblconvg_ngeom(nw, z["vwave"], z["y_ngeom"], z["dydx_ngeom"], nc, z["vconv"], ishape, fw)
    close_nan(yo, z[f"ngconvg_{ishape}_y"], 1e-12)
    close_nan(go, z[f"ngconvg_{ishape}_g"], 1e-11)


def test_lblconv_fil_ngeom_golden(eng, golden_dir):
    from test_conv_oracle import close_nan
    z = _load(golden_dir, "ils_conv")
    nw, nc = z["vwave"].size, z["vconv"].size
    close_nan(eng.lblconv_fil_ngeom(nw, z["vwave"], z["y_ngeom"], nc, z["vconv"], z["nfil"], z["vfil"], z["afil"]),
              z["ngfil_y"], 1e-12)
    yo, go = eng.lblconvg_fil_ngeom(nw, z["vwave"], z["y_ngeom"], z["dydx_ngeom"], nc, z["vconv"], z["nfil"], z["vfil"],
                                    z["afil"])
    close_nan(yo, z["ngfilg_y"], 1e-12)
    close_nan(go, z["ngfilg_g"], 1e-11)


def test_ktable_conv_filter_branch_golden(eng, golden_dir):
    """Measurement_0.conv / convg, FWHM < 0 (k-table runs with a filter per convolution point) vs the reference."""
    from test_conv_oracle import close_nan
    z = _load(golden_dir, "ils_conv")
    nc = z["vconv"].size
    a = (nc, z["vconv"], z["nfil"], z["vfil"], z["afil_k"])
    close_nan(eng.conv_fil(z["vwave"], z["y"], None, *a), z["kconv_fil"], 1e-12)
    yo, go = eng.conv_fil(z["vwave"], z["y"], z["dydx"], *a)
    close_nan(yo, z["kconvg_fil_y"], 1e-12)
    close_nan(go, z["kconvg_fil_g"], 1e-11)
    vf = z["vfil"].copy(); vf[0, 0] = z["vwave"][0] - 1.0            # a filter that sticks out of the grid: IndexError there
    with pytest.raises(ValueError):
        eng.conv_fil(z["vwave"], z["y"], None, nc, z["vconv"], z["nfil"], vf, z["afil_k"])


@pytest.mark.parametrize("ispace", [0, 1])
@pytest.mark.parametrize("name,iray,variant", [("j", 1, None), ("v", 2, "v"), ("v2", 2, None), ("ls", 4, None)])
def test_rayleigh_golden(eng, golden_dir, name, iray, variant, ispace):
    """calc_tau_rayleighj / rayleighv / rayleighv2 / rayleighls vs the reference, wavenumber and wavelength grids."""
    z = _load(golden_dir, "continuum_ray_dust")
    w = z["wn"] if ispace == 0 else z["wl"]
    t, d = eng.calc_tau_rayleigh(iray, ispace, w, z["TOTAM"], z["ID"], z["ISO"], z["VMR"], variant=variant)
    np.testing.assert_allclose(t, z[f"ray_{name}_{ispace}_tau"], rtol=1e-13)
    np.testing.assert_allclose(d, z[f"ray_{name}_{ispace}_dtau"], rtol=1e-13)
    t0, d0 = eng.calc_tau_rayleigh(0, ispace, w, z["TOTAM"])
    assert not t0.any() and not d0.any() and t0.shape == t.shape
    with pytest.raises(ValueError):
        eng.calc_tau_rayleigh(3, ispace, w, z["TOTAM"])               # N2-O2: not implemented in the reference either


@pytest.mark.parametrize("pre,rows", [("dust", slice(None)), ("dust2", [0, -1])])
def test_dust_golden(eng, oracle, golden_dir, pre, rows):
    """calc_tau_dust vs the reference: not-a-knot spline of the host side + the linear fall-back; then a larger random
    case against the oracle."""
    z = _load(golden_dir, "continuum_ray_dust")
    r = eng.calc_tau_dust(z["WAVEC_D"], z["SW"][rows], z["KEXT"][rows], z["KSCA"][rows], z["CONT"])
    for n, a in zip(("TAUDUST", "TAUCLSCAT", "dTAUDUSTdq", "dTAUCLSCATdq"), r):
        e = z[f"{pre}_{n}"]
        assert np.all(np.abs(a - e) <= 1e-12 * np.abs(e).max(axis=(0, 1), keepdims=True)), n
    if pre == "dust":
        rng = np.random.default_rng(3)
        SW = np.cumsum(rng.uniform(0.05, 0.4, 60)); W = np.sort(rng.uniform(SW[0], SW[-1], 5000)); W[0], W[-1] = SW[0], SW[-1]
        KE = 10.0 ** rng.uniform(-10, -8, (60, 4)); KS = KE * rng.uniform(0.2, 1.0, KE.shape)
        CONT = 10.0 ** rng.uniform(2, 8, (40, 4))
        for a, e in zip(eng.calc_tau_dust(W, SW, KE, KS, CONT), oracle.calc_tau_dust(W, SW, KE, KS, CONT)):
            assert np.all(np.abs(a - e) <= 1e-12 * np.abs(e).max(axis=(0, 1), keepdims=True))
        with pytest.raises(ValueError):
            eng.calc_tau_dust(np.array([SW[0] - 0.1, SW[1]]), SW, KE, KS, CONT)       # interp1d(bounds_error=True)


def test_integrate_filter_family_golden(eng, golden_dir):
    """integrate_filter / integrate_filterg / *_ngeom vs the reference (trapezoid weights formed per node on the GPU)."""
    z = _load(golden_dir, "ils_conv")
    nw, nc = z["vwave"].size, z["vconv"].size
    f = (nc, z["vconv"], z["nfil"], z["vfil"], z["afil"])
    np.testing.assert_allclose(eng.integrate_filter(nw, z["vwave"], z["y"], *f), z["intf"], rtol=1e-12)
    yo, go = eng.integrate_filter(nw, z["vwave"], z["y"], *f, dydx=z["dydx"])
    np.testing.assert_allclose(yo, z["intfg_y"], rtol=1e-12)
    np.testing.assert_allclose(go, z["intfg_g"], rtol=0, atol=1e-12 * np.abs(z["intfg_g"]).max())
    np.testing.assert_allclose(eng.integrate_filter(nw, z["vwave"], z["y_ngeom"], *f), z["ngintf"], rtol=1e-12)
    yo, go = eng.integrate_filter(nw, z["vwave"], z["y_ngeom"], *f, dydx=z["dydx_ngeom"])
    np.testing.assert_allclose(yo, z["ngintfg_y"], rtol=1e-12)
    np.testing.assert_allclose(go, z["ngintfg_g"], rtol=0, atol=1e-12 * np.abs(z["ngintfg_g"]).max())


def test_batch_layer_dedup_is_bit_identical(eng):
    """A numerical-Jacobian batch (every state differs from the first in two layers): layers identical to the first
    model's share its opacity rows -- same spectra and TAUGAS to the last bit, far fewer rows computed."""
    from archnemesis_dist_amd import synthetic as syn
    W, G, S, L, NP, NT = 260, 10, 4, 30, 8, 6
    _, delg = syn.gauss_legendre_01(G, True)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=21)
    eng.upload_ktable(K, PRESS, TEMP, 150.0 + np.arange(W), delg)
    base = syn.synth_atmosphere(L, S)
    n = 2 * L + 1
    lp = np.repeat(base["lay_press_pa"], n, 0); lt = np.repeat(base["lay_temp"], n, 0); am = np.repeat(base["amount"], n, 0)
    for i in range(L):                            # temperature of layer i (and the density of its neighbour)
        lt[1 + i, i] *= 1.01
        am[1 + i, :, min(i + 1, L - 1)] *= 0.995
        am[1 + L + i, 1, i] *= 1.05               # amount of gas 1 in layer i
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
    EMTEMP = lt[:, LAYINC[:, 0]][:, :, None]
    cont = np.repeat(syn.synth_continuum(W, L), n, 0)
    args = (0, lp, lt, am, cont, NLAYIN, LAYINC, np.repeat(SCALE[None], n, 0), EMTEMP, np.full(n, -1.0))
    eng.set_layer_dedup(True)
    a = eng.cirsrad_ck_thermal(*args)
    rows, total = eng.last_layer_rows()
    tg_a = eng.get_taugas(L, model=7)
    eng.set_layer_dedup(False)
    b = eng.cirsrad_ck_thermal(*args)
    rows_b, _ = eng.last_layer_rows()
    tg_b = eng.get_taugas(L, model=7)
    eng.set_layer_dedup(True)
    assert total == n * L and rows_b == total and rows <= L + 3 * L + 2 and rows < total // 5
    assert np.array_equal(a, b) and np.array_equal(tg_a, tg_b)
    assert not np.array_equal(a[0], a[3])


@pytest.mark.parametrize("geometry", ["nadir", "limb_paths"])
def test_batch_thermal_rt_starts_from_state0_records_bit_for_bit(eng, geometry):
    """The states of a batch start every path from what state 0 left after the last layer they share with it: a change of a
    layer's gas opacity, of its continuum only, of SCALE only, of EMTEMP only, of TSURF, and no change at all -- against
    the same call with de-duplication (and with it the sharing) off."""
    from archnemesis_dist_amd import synthetic as syn
    W, G, S, L, NP, NT = 200, 12, 3, 24, 8, 6
    _, delg = syn.gauss_legendre_01(G, True)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=33)
    eng.upload_ktable(K, PRESS, TEMP, 150.0 + np.arange(W), delg)
    base = syn.synth_atmosphere(L, S)
    n = 9
    lp = np.repeat(base["lay_press_pa"], n, 0); lt = np.repeat(base["lay_temp"], n, 0); am = np.repeat(base["amount"], n, 0)
    if geometry == "nadir":
        NLAYIN, LAYINC, SCALE = syn.nadir_path(L)
    else:       # three tangent paths: down to layer b and up again (AtmCalc_0's limb order), ragged NLAYIN
        bots = [2, 9, 17]
        LIMAX = 2 * (L - min(bots))
        NLAYIN = np.array([2 * (L - b) for b in bots], dtype=np.int32)
        LAYINC = np.zeros((LIMAX, len(bots)), dtype=np.int32); SCALE = np.zeros((LIMAX, len(bots)))
        for ip, b in enumerate(bots):
            down = np.arange(L - 1, b - 1, -1)
            order = np.concatenate([down, down[::-1]])
            LAYINC[:order.size, ip] = order
            SCALE[:order.size, ip] = 1.0 + 3.0 / (1.0 + np.abs(order - b))
    P_ = LAYINC.shape[1]
    inside = np.arange(LAYINC.shape[0])[:, None] < NLAYIN[None, :]
    SC = np.repeat(SCALE[None], n, 0)
    cont = np.repeat(syn.synth_continuum(W, L), n, 0)
    tsurf = np.full(n, -1.0)
    lt[1, 20] *= 1.02                        # near the top of the atmosphere: nearly nothing shared on the way down
    am[2, 1, 3] *= 1.05                      # near the bottom: most of a nadir path shared
    cont[3, :, 11] *= 1.01                   # the continuum alone
    cont[4, 7, 15] *= 1.0000001              # ... at a single wavenumber
    SC[5, 5, 0] *= 1.01                      # SCALE alone
    lt[7, 0] *= 1.0                          # state 7: nothing changed
    tsurf[8] = 180.0                         # the ground term only (applied after the loop)
    EMTEMP = np.where(inside[None], lt[:, LAYINC], 0.0)
    EMTEMP[6, 8, P_ - 1] += 0.5              # EMTEMP alone
    emis = np.full(W, 0.9)
    args = (0, lp, lt, am, cont, NLAYIN, LAYINC, SC, EMTEMP, tsurf)
    eng.set_layer_dedup(True)
    a = eng.cirsrad_ck_thermal(*args, EMISSIVITY=emis)
    assert eng.last_rt_shared()
    eng.set_layer_dedup(False)
    try:
        b = eng.cirsrad_ck_thermal(*args, EMISSIVITY=emis)
        assert not eng.last_rt_shared()
    finally:
        eng.set_layer_dedup(True)
    assert np.array_equal(a, b)
    assert np.array_equal(a[7], a[0])
    for m in (1, 2, 3, 4, 5, 6):
        assert not np.array_equal(a[m], a[0]), m
    if geometry == "nadir":
        assert not np.array_equal(a[8], a[0])


@pytest.mark.parametrize("W,G,S,L", [(1, 1, 1, 1), (3, 1, 3, 2), (63, 2, 2, 3), (65, 3, 4, 2), (130, 32, 3, 2), (5, 31, 2, 1),
                                       (64, 17, 5, 1)])
def test_k_overlap_edge_sizes_vs_oracle(eng, oracle, W, G, S, L):
    """Smallest / largest g-ordinate counts, single gas, single layer, wavenumber counts around the 64-lane tile."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(W * 1000 + G * 10 + S)
    _, delg = syn.gauss_legendre_01(G, False)
    k = np.sort(10.0 ** rng.uniform(-25, -20, (W, G, L, S)), axis=1)
    k[rng.uniform(size=(W, 1, L, S)).repeat(G, 1) < 0.15] = 0.0                # whole (cell, gas) columns empty
    amount = 10.0 ** rng.uniform(19, 22, (S, L))
    tau = eng.k_overlap(delg, k, amount)
    ref = oracle.k_overlap(delg, k, amount)
    np.testing.assert_allclose(tau, ref, rtol=1e-11, atol=0)
    dkdT = k * rng.uniform(-0.01, 0.01, k.shape)
    tg, dk = eng.k_overlapg(delg, k, dkdT, amount)
    rg, rdk = oracle.k_overlapg(delg, k, dkdT, amount)
    np.testing.assert_allclose(tg, rg, rtol=1e-11, atol=0)
    # G = 1 makes rankg divide 0 by 0 in the reference (restated by the oracle): the NaNs must be the same ones
    assert np.array_equal(np.isnan(dk), np.isnan(rdk))
    ok = ~np.isnan(rdk)
    if ok.any():
        scale = np.max(np.abs(np.where(ok, rdk, 0.0)), axis=(0, 1), keepdims=True) + 1e-300
        assert np.max(np.where(ok, np.abs(dk - rdk) / scale, 0.0)) < 1e-9


@pytest.mark.parametrize("ispace", [0, 1])
def test_cirsrad_multi_path_ragged_vs_oracle(eng, oracle, ispace):
    """Three paths of different length in one call (nadir top->bottom, a short one, a limb-like down-and-up path with
    repeated layers), surface + solar-reflection terms, wavenumber and wavelength units, two models."""
    from archnemesis_dist_amd import synthetic as syn
    W, G, S, L, NP, NT = 150, 10, 3, 14, 7, 6
    _, delg = syn.gauss_legendre_01(G, True)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=31)
    WAVE = (400.0 + 2.0 * np.arange(W)) if ispace == 0 else np.linspace(5.0, 25.0, W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    n = 2
    atm = syn.synth_atmosphere(L, S, seed=4, n_models=n, perturb=0.03)
    P = 3
    LIMAX = 2 * 6
    LAYINC = np.zeros((LIMAX, P), dtype=np.int32)
    NLAYIN = np.array([L if L <= LIMAX else LIMAX, 5, 12], dtype=np.int32)
    LAYINC[:NLAYIN[0], 0] = np.arange(L - 1, -1, -1)[:NLAYIN[0]]                # top -> bottom
    LAYINC[:5, 1] = np.arange(L - 1, L - 6, -1)
    LAYINC[:12, 2] = np.concatenate([np.arange(L - 1, L - 7, -1), np.arange(L - 6, L)])   # down, then up again
    rng = np.random.default_rng(12)
    SCALE = rng.uniform(1.0, 3.0, (n, LIMAX, P))
    EMTEMP = np.stack([atm["lay_temp"][m][LAYINC] for m in range(n)])            # (n, LIMAX, P)
    cont = syn.synth_continuum(W, L, n_models=n)
    TSURF = np.array([250.0, -1.0])
    EMIS = np.linspace(0.8, 1.0, W); SOLF = 10.0 ** rng.uniform(-9, -8, W); REFL = rng.uniform(0, 0.3, W)
    SOL = np.array([20.0, 120.0, 45.0]); EMI = np.array([10.0, 30.0, 85.0])
    out = eng.cirsrad_ck_thermal(ispace, atm["lay_press_pa"], atm["lay_temp"], atm["amount"], cont, NLAYIN, LAYINC, SCALE, EMTEMP,
                                 TSURF, EMISSIVITY=EMIS, SOLFLUX=SOLF, REFLECTANCE=REFL, SOL_ANG=SOL, EMISS_ANG=EMI)
    assert out.shape == (n, W, P)
    for m in range(n):
        ref = oracle.cirsrad_ck_thermal(ispace, K, PRESS, TEMP, WAVE, delg, atm["lay_press_pa"][m], atm["lay_temp"][m],
                                        atm["amount"][m], cont[m], NLAYIN, LAYINC, SCALE[m], EMTEMP[m], TSURF[m], EMISSIVITY=EMIS,
                                        SOLFLUX=SOLF, REFLECTANCE=REFL, SOL_ANG=SOL, EMISS_ANG=EMI)
        np.testing.assert_allclose(out[m], ref, rtol=1e-10, atol=0, err_msg=f"model {m}")


def test_cirsradg_multi_path_ragged_vs_oracle(eng, oracle):
    """CIRSrad(return_grad=True) with three paths of different length (one visiting layers twice): dSPECOUT (W,NPAR,LIMAX,P)."""
    from archnemesis_dist_amd import synthetic as syn
    W, G, S, L, NP, NT = 100, 10, 3, 12, 7, 6
    _, delg = syn.gauss_legendre_01(G, True)
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=32)
    WAVE = 300.0 + 1.5 * np.arange(W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    atm = syn.synth_atmosphere(L, S, seed=5)
    P, LIMAX = 3, 12
    LAYINC = np.zeros((LIMAX, P), dtype=np.int32)
    NLAYIN = np.array([12, 4, 10], dtype=np.int32)
    LAYINC[:12, 0] = np.arange(L - 1, -1, -1)
    LAYINC[:4, 1] = np.arange(L - 1, L - 5, -1)
    LAYINC[:10, 2] = np.concatenate([np.arange(L - 1, L - 6, -1), np.arange(L - 5, L)])
    rng = np.random.default_rng(13)
    SCALE = rng.uniform(1.0, 3.0, (LIMAX, P))
    EMTEMP = atm["lay_temp"][0][LAYINC]
    cont = syn.synth_continuum(W, L)[0]
    NVMR, NDUST = S, 1
    NPAR = NVMR + 2 + NDUST
    dcont = cont[:, None, :] * rng.uniform(0.0, 1e-22, size=(W, NPAR, L))
    spec, dspec, dts = eng.cirsradg_ck_thermal(0, atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0], cont, dcont, NVMR, NPAR,
                                               np.arange(S, dtype=np.int32), NLAYIN, LAYINC, SCALE, EMTEMP, 240.0,
                                               EMISSIVITY=np.linspace(0.8, 1.0, W))
    rs, rd, rt = oracle.cirsradg_ck_thermal(0, K, PRESS, TEMP, WAVE, delg, atm["lay_press_pa"][0], atm["lay_temp"][0],
                                            atm["amount"][0], cont, dcont, NVMR, NPAR, np.arange(S, dtype=np.int32), NLAYIN, LAYINC,
                                            SCALE, EMTEMP, 240.0, EMISSIVITY=np.linspace(0.8, 1.0, W))
    assert dspec.shape == (W, NPAR, LIMAX, P)
    np.testing.assert_allclose(spec, rs, rtol=1e-10)
    np.testing.assert_allclose(dts, rt, rtol=1e-10, atol=0)
    scale = np.abs(rd).max(axis=(0, 2), keepdims=True) + 1e-300
    assert np.max(np.abs(dspec - rd) / scale) < 1e-9


@pytest.mark.parametrize("tag,space", [(t, sp) for t in ("eq", "normal", "para") for sp in (0, 1)])
def test_calc_tau_cia_golden(eng, golden_dir, tag, space):
    """Collision-induced absorption vs the reference's calc_tau_cia (golden): ortho/para pair selection, table clamps in
    temperature and para fraction, the co2cia / n2n2cia / n2h2cia terms, wavenumber and wavelength grids, gradients."""
    from test_cia_oracle import cia_args
    z = _load(golden_dir, "tau_cia")
    a, kw = cia_args(z, tag, space)
    tau, dtau = eng.calc_tau_cia(*a, **kw)
    ref_t, ref_d = z[f"{tag}_{space}_tau"], z[f"{tag}_{space}_dtau"]
    np.testing.assert_allclose(tau, ref_t, rtol=1e-12, atol=0)
    scale = np.max(np.abs(ref_d), axis=(0, 1), keepdims=True) + 1e-300
    assert np.max(np.abs(dtau - ref_d) / scale) < 1e-12
    assert np.array_equal(eng.calc_tau_cia(*a, with_grad=False, **kw), tau)


def test_lbl_table_batch_with_dedup(eng):
    """ILBL = LINE_BY_LINE_TABLES in a batch: per-model results equal the single-model calls, with and without layer
    de-duplication."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(14)
    W, NP, NT, S, L, n = 200, 6, 5, 2, 10, 4
    PRESS = np.logspace(-6, 1.1, NP); TEMP = np.linspace(80.0, 420.0, NT)
    K = 10.0 ** rng.uniform(-27, -20, size=(W, NP, NT, S))
    eng.upload_lbltable(K, PRESS, TEMP, 2000.0 + 0.01 * np.arange(W))
    atm = syn.synth_atmosphere(L, S, seed=3, n_models=n, perturb=0.04)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L, 15.0)
    EMTEMP = atm["lay_temp"][:, LAYINC[:, 0]][:, :, None]
    args = (0, atm["lay_press_pa"], atm["lay_temp"], atm["amount"], None, NLAYIN, LAYINC, np.repeat(SCALE[None], n, 0), EMTEMP,
            np.full(n, 200.0))
    a = eng.cirsrad_ck_thermal(*args)
    assert eng.last_layer_rows()[0] < n * L
    eng.set_layer_dedup(False)
    b = eng.cirsrad_ck_thermal(*args)
    eng.set_layer_dedup(True)
    assert np.array_equal(a, b)
    for m in range(n):
        one = eng.cirsrad_ck_thermal(0, atm["lay_press_pa"][m], atm["lay_temp"][m], atm["amount"][m], None, NLAYIN, LAYINC, SCALE,
                                     EMTEMP[m], 200.0)
        assert np.array_equal(one, a[m])


@pytest.mark.parametrize("ishape", [0, 1, 2])
def test_lblconv_reference_known_answer(eng, ishape):
    """Mirror of the reference's own tests/test_measurement_class.py::test_lblconv (:8-79): a Gaussian line on a 0.1 cm-1
    grid convolved with a FWHM = 1 ILS must agree with numpy.convolve of the same kernel where the signal is >= 1 % of
    its maximum (the reference asserts rtol = 3e-2; the kernels agree far better away from the grid ends)."""
    dvx = 0.1
    vwave = np.arange(0, 50. + dvx, dvx)
    sig0 = 0.5 * 1.0 / np.sqrt(np.log(2))
    y = np.exp(-((vwave - 35.) / sig0) ** 2)
    dy = (y * 0.01)[:, None]
    fwhm = 1.0
    yc = eng.lblconv(vwave.size, vwave, y, vwave.size, vwave, ishape, fwhm)
    yc2, g2 = eng.lblconvg(vwave.size, vwave, y, dy, vwave.size, vwave, ishape, fwhm)
    half = {0: 0.5 * fwhm, 1: fwhm, 2: 3. * sig0}[ishape]
    nk = int(round(half / dvx))
    xk = dvx * np.arange(-nk, nk + 1)
    kern = {0: np.ones_like(xk), 1: 1.0 - np.abs(xk) / fwhm, 2: np.exp(-(xk / sig0) ** 2)}[ishape]
    kern = np.where(kern > 0, kern, 0.0); kern /= kern.sum()
    ynp = np.convolve(y, kern, mode="same")
    big = yc / yc.max() >= 1.0e-2
    assert np.allclose(yc[big], ynp[big], rtol=3.0e-2)
    assert np.allclose(yc2[big], ynp[big], rtol=3.0e-2)
    np.testing.assert_allclose(g2[big, 0], 0.01 * yc2[big], rtol=1e-12)


# ---- CIRSrad, scattering branch (BASELINE configs[3]) ---------------------------------------------------------
def _scatter_inputs(rng, W, G, L, S, NMU, NF, ncont, imie, iray, lowbc):
    """Seeded inputs of ansfm_cirsrad_ck_scatter in the reference's layouts (what scloud11wave prepares, :5018-5165)."""
    from archnemesis_dist_amd import synthetic as syn
    NP, NT = 8, 6
    PRESS, TEMP, K = syn.synth_ktable(W, G, NP, NT, S, seed=int(rng.integers(1 << 30)))
    _, delg = syn.gauss_legendre_01(G)
    WAVE = 600.0 + 2.0 * np.arange(W)
    lay_p = np.logspace(5.3, 1.5, L); lay_t = np.linspace(165.0, 110.0, L)
    amount = 10.0 ** rng.uniform(17.5, 20.5, (S, 1)) * (lay_p[None, :] / lay_p[0]) ** 0.9
    TAUCIA = 10.0 ** rng.uniform(-5, -2, (W, L)); TAURAY = (10.0 ** rng.uniform(-6, -3, (W, L))) * (1.0 if iray else 0.0)
    clscat = 10.0 ** rng.uniform(-5, -1.5, (W, L, ncont))
    clscat[:, 2, :] = 0.0                                                   # a layer without aerosol
    TAUSCAT = clscat.sum(axis=2)
    TAUDUST = TAUSCAT * rng.uniform(1.02, 1.5, (W, L))                      # extinction >= scattering
    lfrac = np.zeros((W, ncont, L))
    pos = TAUSCAT > 0
    lfrac[:] = np.transpose(np.where(pos[:, :, None], clscat / np.where(pos, TAUSCAT, 1.0)[:, :, None], 0.0), (0, 2, 1))
    x, w = np.polynomial.legendre.leggauss(2 * NMU)                         # any positive quadrature on (0, 1] serves the test
    MU = 0.5 * (x[NMU:] + 1.0); MU[-1] = 1.0; WT = w[NMU:] * 0.5
    THETA = np.linspace(0.0, 180.0, 41)
    PH = np.zeros((ncont, W, 2, THETA.size))
    if imie == 0:
        PH[:, :, 0, -1] = rng.uniform(0.6, 0.95, (ncont, W)); PH[:, :, 0, -2] = rng.uniform(0.3, 0.8, (ncont, W))
        PH[:, :, 0, -3] = rng.uniform(-0.5, -0.1, (ncont, W))
    else:
        c = np.cos(np.deg2rad(THETA))
        gg = rng.uniform(0.2, 0.7, (ncont, W, 1))
        PH[:, :, 0, :] = (1 - gg * gg) / (1 + gg * gg - 2 * gg * c) ** 1.5 / (4 * np.pi)
    PH[:, :, 1, :] = np.cos(THETA * np.pi / 180)
    phasarr = np.ascontiguousarray(PH[:, :, :, ::-1])
    c1, c2 = 1.1911e-12, 1.439
    radg = np.repeat((c1 * WAVE ** 3 / (np.exp(c2 * WAVE / lay_t[0]) - 1.0))[:, None], NMU, 1)
    solar = 10.0 ** rng.uniform(-9, -8, W)
    brdf = np.zeros((W, NMU, NMU, NF + 1))
    if lowbc:
        brdf[:, :, :, 0] = rng.uniform(0.05, 0.3, (W, 1, 1)) / np.pi
    return dict(K=K, TPRESS=PRESS, TTEMP=TEMP, WAVE=WAVE, DELG=delg, lay_p=lay_p, lay_t=lay_t, amount=amount, TAUCIA=TAUCIA,
                TAUDUST=TAUDUST, TAURAY=TAURAY, TAUSCAT=TAUSCAT, lfrac=lfrac, phasarr=phasarr, radg=radg, solar=solar,
                brdf=brdf, MU=MU, WT=WT)


def _scatter_oracle(oracle, z, sol, emi, azi, lowbc, NF, nphi, iray, imie):
    """The reference's recipe on the oracle's restatements: calc_k + k_overlap, TAUTOT (:3989), OMEGA / BB (:5099-5119),
    scloud11wave_core, g-quadrature (:4504)."""
    W, L = z["TAUCIA"].shape
    k = oracle.calc_k(z["K"], z["TPRESS"], z["TTEMP"], z["lay_p"] / 101325.0, z["lay_t"])
    taugas = oracle.k_overlap(z["DELG"], k, z["amount"])
    tautot = taugas + z["TAUCIA"][:, None, :] + z["TAUDUST"][:, None, :] + z["TAURAY"][:, None, :]
    omega = np.zeros_like(tautot)
    pos = tautot > 0
    omega[pos] = np.broadcast_to((z["TAURAY"] + z["TAUSCAT"])[:, None, :], tautot.shape)[pos] / tautot[pos]
    c1, c2 = 1.1911e-12, 1.439
    bnu = c1 * z["WAVE"][:, None] ** 3 / (np.exp(c2 * z["WAVE"][:, None] / z["lay_t"][None, :]) - 1.0)
    rad = oracle.scloud11wave_core(z["phasarr"], z["radg"], sol, emi, z["solar"], azi, lowbc, z["brdf"], z["MU"], z["WT"], NF,
                                   z["WAVE"], bnu, tautot, z["TAURAY"], omega, nphi, iray, imie, z["lfrac"])
    spec = np.transpose(rad, (2, 1, 0))                                     # (W, G, P), :5164
    return np.tensordot(spec, z["DELG"], axes=([1], [0])), spec, taugas


@pytest.mark.parametrize("NMU,NF,ncont,imie,iray,lowbc,up", [(5, 2, 2, 0, 1, 0, False), (16, 3, 1, 1, 1, 1, False),
                                                             (24, 2, 1, 1, 1, 1, False), (32, 1, 1, 0, 1, 0, True),   # beyond the old cap of 20
                                                             (5, 1, 1, 1, 0, 0, True), (16, 2, 0, 0, 1, 0, False),
                                                             (16, 2, 2, 0, 1, 0, False),      # three components: phase matrices stay in HBM
                                                             (16, 1, 2, 1, 0, 1, True),
                                                             # 7 .. 15 streams: padded to the 16-stream matrix-core kernels
                                                             (7, 2, 1, 1, 1, 0, False), (8, 3, 2, 0, 1, 1, False), (10, 2, 1, 1, 1, 1, True),
                                                             (12, 2, 2, 1, 1, 0, True), (15, 1, 1, 0, 0, 1, False),
                                                             (4, 2, 1, 1, 1, 1, False), (6, 2, 2, 0, 1, 1, True)])
def test_cirsrad_scatter_vs_oracle(eng, oracle, NMU, NF, ncont, imie, iray, lowbc, up):
    """ansfm_cirsrad_ck_scatter (gas opacities, TAUTOT, OMEGA, BB formed on the device, straight into the doubling /
    adding kernels) vs the same chain through the oracle: radiances before and after the g-quadrature and TAUGAS."""
    rng = np.random.default_rng(900 + NMU + NF + ncont)
    W, G, L, S = 24, 6, 7, 3
    z = _scatter_inputs(rng, W, G, L, S, NMU, NF, ncont, imie, iray, lowbc)
    sol = np.array([30.0, 120.0]); emi = np.array([160.0, 130.0]) if up else np.array([20.0, 50.0]); azi = np.array([45.0, 0.0])
    eng.upload_ktable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"])
    ph = z["phasarr"] if ncont else None
    out, spec_g = eng.cirsrad_ck_scatter(0, z["lay_p"], z["lay_t"], z["amount"], z["TAUCIA"], z["TAUDUST"],
                                         z["TAURAY"] if iray else None, z["TAUSCAT"], ph, z["lfrac"] if ncont else None,
                                         z["radg"], sol, emi, azi, z["solar"], lowbc, z["brdf"], z["MU"], z["WT"], NF, 101, iray,
                                         imie, return_spec_g=True)
    ref, ref_g, taugas = _scatter_oracle(oracle, z, sol, emi, azi, lowbc, NF, 101, iray, imie)
    np.testing.assert_allclose(eng.get_taugas(L, 0), taugas, rtol=1e-11, atol=0)
    scale = np.max(np.abs(ref_g))
    assert np.max(np.abs(spec_g - ref_g)) / scale < 1e-8
    assert np.max(np.abs(out - ref)) / np.max(np.abs(ref)) < 1e-8


@pytest.mark.parametrize("ncont,imie,iray,lowbc,up,slab", [(2, 0, 1, 0, False, None), (1, 1, 1, 1, False, 5), (1, 1, 0, 1, True, 7),
                                                           (0, 0, 1, 0, False, 3)])
def test_cirsrad_scatter_batch_equals_separate_calls(eng, monkeypatch, ncont, imie, iray, lowbc, up, slab):
    """ansfm_cirsrad_ck_scatter_batch -- the forward models of a numerical Jacobian of the scattering configuration: model 0
    plus models that differ from it in one or two layers (temperature, one gas amount, the aerosol opacity, the Rayleigh
    opacity of a layer) and one that differs everywhere.  Model 0's doubled layers are cached, the others re-run the
    adding sweep over them -- from the cached stack below their first changed layer where there is one: every spectrum is
    bit-identical to a call of its own, with one slab and with several (ANSFM_MS_SLAB), and the bookkeeping says which layers
    came from the cache."""
    rng = np.random.default_rng(7700 + ncont + 3 * imie + 7 * lowbc)
    W, G, L, S, NMU, NF = 20, 5, 9, 3, 16, 3
    z = _scatter_inputs(rng, W, G, L, S, NMU, NF, max(ncont, 1), imie, iray, lowbc)
    if ncont == 0:
        z["TAUSCAT"] = np.zeros((W, L)); z["TAUDUST"] = 10.0 ** rng.uniform(-5, -3, (W, L))
    sol = np.array([30.0, 120.0]); emi = np.array([160.0, 130.0]) if up else np.array([20.0, 50.0]); azi = np.array([45.0, 0.0])
    eng.upload_ktable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"])
    n = 8
    rep = lambda a: np.repeat(np.asarray(a)[None], n, 0).copy()
    lp, lt, am = rep(z["lay_p"]), rep(z["lay_t"]), rep(z["amount"])
    cia, dust, ray, sca = rep(z["TAUCIA"]), rep(z["TAUDUST"]), rep(z["TAURAY"]), rep(z["TAUSCAT"])
    lf, rg = rep(z["lfrac"]), rep(z["radg"])
    lt[1, 4] *= 1.05                                   # a layer temperature (gas opacity and Planck function of that layer)
    am[2, 1, 6] *= 1.05                                # a gas amount
    sca[3, :, 3] *= 1.05; dust[3, :, 3] *= 1.05        # the aerosol of a layer
    ray[4, :, 0] *= 1.05                               # Rayleigh opacity of the bottom layer
    lt[5, 0] *= 1.02; rg[5] *= 1.1                     # bottom temperature: layer 0 and the lower boundary radiance
    lt[6] *= 1.01; am[6] *= 1.03                       # everything: nothing can come from the cache
    rg[7] *= 1.2                                       # only the lower boundary radiance: every layer cached, but the stack of a
                                                       # look-down sweep over a reflecting surface starts from it (no prefix)
    common = (z["phasarr"] if ncont else None,)
    tail = (sol, emi, azi, z["solar"], lowbc, z["brdf"], z["MU"], z["WT"], NF, 101, iray, imie)
    one = lambda m: eng.cirsrad_ck_scatter(0, lp[m], lt[m], am[m], cia[m], dust[m], ray[m] if iray else None, sca[m], common[0],
                                           lf[m] if ncont else None, rg[m], *tail)
    ref = np.stack([one(m) for m in range(n)])
    if slab is not None:
        monkeypatch.setenv("ANSFM_MS_SLAB", str(slab))
    got = eng.cirsrad_ck_scatter_batch(0, lp, lt, am, cia, dust, ray if iray else None, sca, common[0], lf if ncont else None, rg, *tail)
    assert np.array_equal(got, ref)
    hits, total = eng.last_scatter_cache()
    assert total == (n - 1) * L
    changed = 1 + 1 + 1 + (1 if iray else 0) + 1 + L   # layers that differ from model 0, per model 1..6
    assert hits == total - changed
    rows, allrows = eng.last_layer_rows()
    assert allrows == n * L and rows == L + 1 + 1 + 0 + 0 + 1 + L          # gas opacity rows: T / amount changes only
    eng.set_layer_dedup(False)                         # model by model through the single entry point
    try:
        again = eng.cirsrad_ck_scatter_batch(0, lp, lt, am, cia, dust, ray if iray else None, sca, common[0], lf if ncont else None,
                                             rg, *tail)
    finally:
        eng.set_layer_dedup(True)
    assert np.array_equal(again, ref)


@pytest.mark.parametrize("NMU,lane,dedup", [(5, True, True), (5, False, True), (4, True, True), (8, True, True), (12, True, True),
                                            (20, True, True), (5, True, False), (8, True, False)])
def test_cirsrad_scatter_batch_other_stream_counts_equal_separate_calls(eng, monkeypatch, NMU, lane, dedup):
    """Other stream counts than 16 (5 = the reference's default: the lane-per-chain kernel, or with ANSFM_MS_LANE=0 the
    wavefront-per-chain one; 8: that kernel with compile-time sizes; 12, 20: its run-time build): model 0's doubled layers are
    cached, the other models run the adding sweep over them (k_ms_chain_lane<N, CACHE> / k_ms_chain<N, CACHE>) -- every spectrum
    equal to a call of its own bit for bit, also when a single call came in between and the batch is repeated.  With the
    de-duplication off the models run one after the other on the phase matrices and Hansen factors model 0 left behind."""
    if not lane:
        monkeypatch.setenv("ANSFM_MS_LANE", "0")
    if NMU in (5, 20) and lane and dedup:
        monkeypatch.setenv("ANSFM_MS_SLAB", "64")               # 70 wavenumbers: two slabs of the layer cache, the second a partial tile
    rng = np.random.default_rng(9100 + NMU)
    W, G, L, S, NF = 70, 4, 8, 3, 2
    z = _scatter_inputs(rng, W, G, L, S, NMU, NF, 2, 1, 1, 1)
    sol = np.array([30.0, 75.0]); emi = np.array([20.0, 50.0]); azi = np.array([45.0, 10.0])
    eng.upload_ktable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"])
    n = 4
    rep = lambda a: np.repeat(np.asarray(a)[None], n, 0).copy()
    lp, lt, am = rep(z["lay_p"]), rep(z["lay_t"]), rep(z["amount"])
    cia, dust, ray, sca = rep(z["TAUCIA"]), rep(z["TAUDUST"]), rep(z["TAURAY"]), rep(z["TAUSCAT"])
    lf, rg = rep(z["lfrac"]), rep(z["radg"])
    lt[1, 4] *= 1.05; am[2, 1, 6] *= 1.05; sca[3, :, 3] *= 1.05; dust[3, :, 3] *= 1.05
    tail = (sol, emi, azi, z["solar"], 1, z["brdf"], z["MU"], z["WT"], NF, 101, 1, 1)
    one = lambda m: eng.cirsrad_ck_scatter(0, lp[m], lt[m], am[m], cia[m], dust[m], ray[m], sca[m], z["phasarr"], lf[m], rg[m], *tail)
    ref = np.stack([one(m) for m in range(n)])
    eng.set_layer_dedup(dedup)
    try:
        got = eng.cirsrad_ck_scatter_batch(0, lp, lt, am, cia, dust, ray, sca, z["phasarr"], lf, rg, *tail)
        assert np.array_equal(got, ref)
        hits, total = eng.last_scatter_cache()
        if dedup:
            assert total == (n - 1) * L and hits == total - 3                     # three changed layers in all
        else:
            assert hits == 0
        assert np.array_equal(one(2), ref[2])                   # a single call afterwards starts from its own walk again
        assert np.array_equal(eng.cirsrad_ck_scatter_batch(0, lp, lt, am, cia, dust, ray, sca, z["phasarr"], lf, rg, *tail), ref)
    finally:
        eng.set_layer_dedup(True)
    assert not np.array_equal(ref[1], ref[0]) and not np.array_equal(ref[3], ref[0])


def test_cirsrad_scatter_more_than_sixteen_paths(eng):
    """Twenty paths -- four more than one call of the chain kernels takes (one path per lane of a 16-lane row): the engine
    runs them in groups, each path's spectrum equal to the one it gets in a call of its own group of <= 16."""
    rng = np.random.default_rng(1616)
    W, G, L, S, NMU, NF = 10, 4, 6, 2, 16, 2
    z = _scatter_inputs(rng, W, G, L, S, NMU, NF, 1, 1, 1, 0)
    P = 20
    sol = rng.uniform(10.0, 70.0, P); emi = rng.uniform(5.0, 75.0, P); azi = rng.uniform(0.0, 180.0, P)
    eng.upload_ktable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"])
    run = lambda sel: eng.cirsrad_ck_scatter(0, z["lay_p"], z["lay_t"], z["amount"], z["TAUCIA"], z["TAUDUST"], z["TAURAY"], z["TAUSCAT"],
                                             z["phasarr"], z["lfrac"], z["radg"], sol[sel], emi[sel], azi[sel], z["solar"], 0, z["brdf"],
                                             z["MU"], z["WT"], NF, 101, 1, 1, return_spec_g=True)
    out, spec_g = run(slice(None))
    assert out.shape == (W, P) and spec_g.shape == (W, G, P)
    a, ag = run(slice(0, 16)); b, bg = run(slice(16, 20))
    assert np.array_equal(out, np.concatenate([a, b], axis=1)) and np.array_equal(spec_g, np.concatenate([ag, bg], axis=2))
    with pytest.raises(ValueError):
        emi2 = emi.copy(); emi2[18] = 120.0
        eng.cirsrad_ck_scatter(0, z["lay_p"], z["lay_t"], z["amount"], z["TAUCIA"], z["TAUDUST"], z["TAURAY"], z["TAUSCAT"], z["phasarr"],
                               z["lfrac"], z["radg"], sol, emi2, azi, z["solar"], 0, z["brdf"], z["MU"], z["WT"], NF, 101, 1, 1)


def test_cirsrad_scatter_fine_azimuth_grid_vs_oracle(eng, oracle):
    """NPHI = 701 with NF = 8: the cos(ic phi) table of k_ms_phase (56 KB) does not fit its LDS budget, the cosines are
    evaluated in place -- same sums as the table path and as the oracle."""
    rng = np.random.default_rng(4242)
    W, G, L, S, NMU, NF, nphi = 6, 3, 5, 2, 16, 8, 701
    z = _scatter_inputs(rng, W, G, L, S, NMU, NF, 1, 1, 1, 0)
    sol = np.array([30.0]); emi = np.array([20.0]); azi = np.array([45.0])
    eng.upload_ktable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"])
    out = eng.cirsrad_ck_scatter(0, z["lay_p"], z["lay_t"], z["amount"], z["TAUCIA"], z["TAUDUST"], z["TAURAY"], z["TAUSCAT"],
                                 z["phasarr"], z["lfrac"], z["radg"], sol, emi, azi, z["solar"], 0, z["brdf"], z["MU"], z["WT"],
                                 NF, nphi, 1, 1)
    ref, _, _ = _scatter_oracle(oracle, z, sol, emi, azi, 0, NF, nphi, 1, 1)
    assert np.max(np.abs(out - ref)) / np.max(np.abs(ref)) < 1e-8


def test_cirsrad_scatter_reference_golden(eng, golden_dir):
    """The reference's CIRSrad in its multiple-scattering branch on its own scattering test inputs (Jupiter CIRS, one
    Henyey-Greenstein haze + Rayleigh, sunlight on; oracle/gen_golden_c4.py): what it read and what it returned."""
    z = _load(golden_dir, "c4_cirsrad_scatter")
    assert int(z["IMOD"][0]) & 256 and not int(z["IMOD"][0]) & 64
    eng.upload_ktable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"])
    f_gas = np.ascontiguousarray(z["LAY_AMOUNT"][:, z["IGAS"]].T) * 1.0e-4
    out, spec_g = eng.cirsrad_ck_scatter(int(z["ISPACE"]), z["LAY_PRESS"], z["LAY_TEMP"], f_gas, z["TAUCIA"], z["TAUDUST"],
                                         z["TAURAY"], z["TAUSCAT"], z["core_phasarr"], z["core_lfrac"], z["core_radg"],
                                         z["SOL_ANG"], z["EMISS_ANG"], z["AZI_ANG"], z["core_solar"], int(z["LOWBC"]),
                                         z["core_brdf"], z["MU"], z["WTMU"], int(z["NF"]), int(z["NPHI"]), int(z["IRAY"]),
                                         int(z["IMIE"]), return_spec_g=True)
    rt = 2e-7 if z["TPRESS"].dtype == np.float32 else 1e-11       # float32 grids: NumPy's float32 log (see test_calc_k_golden)
    np.testing.assert_allclose(eng.get_taugas(z["LAY_PRESS"].size, 0), z["TAUGAS"], rtol=rt, atol=0)
    ref_g = np.transpose(z["core_rad"], (2, 1, 0))
    assert np.max(np.abs(spec_g - ref_g)) / np.max(np.abs(ref_g)) < max(1e-8, 10 * rt)
    assert np.max(np.abs(out - z["SPECOUT"]) / np.abs(z["SPECOUT"])) < max(1e-8, 10 * rt)


@pytest.mark.parametrize("name", ["ms_nmu16_deep", "ms_nmu16_deep_lambert"])
def test_scloud11wave_core_deep_golden(eng, golden_dir, name):
    """BASELINE configs[3] depth -- 16 streams, 9 Fourier orders, 50-60 layers of up to 12 doublings each -- against the
    reference's own core: the MFMA chain kernel's inverse (product form of the Neumann series, Gauss-Jordan fallback)
    in place of numpy.linalg.inv over a deep stack."""
    z = _load(golden_dir, name)
    rad = eng.scloud11wave_core(z["phasarr"], z["radg"], z["sol_angs"], z["emiss_angs"], z["solar"], z["aphis"], int(z["lowbc"]),
                                z["brdf_matrix"], z["mu1"], z["wt1"], int(z["nf"]), z["vwaves"], z["bnu"], z["taus"], z["tauray"],
                                z["omegas_s"], int(z["nphi"]), int(z["iray"]), int(z["imie"]), z["lfrac"])
    assert np.max(np.abs(rad - z["rad"])) / np.max(np.abs(z["rad"])) < 1e-8


@pytest.mark.parametrize("nmu,lookup,lowbc", [(5, False, 0), (5, False, 1), (5, True, 0), (5, True, 1), (4, False, 1), (4, True, 1),
                                              (6, False, 1), (6, True, 0)])
def test_few_streams_one_lane_per_chain_equals_one_wavefront_per_chain(eng, monkeypatch, nmu, lookup, lowbc):
    """The reference's default quadrature (5 streams) runs one LANE per (wavenumber, g, order) chain with the matrices in
    registers (k_ms_chain_lane); the wavefront-per-chain kernel (k_ms_chain<N>, LDS matrices; ANSFM_MS_LANE=0) does the same
    operations in the same order: the same radiances -- bit for bit in most configurations, within one unit in the last place
    where the compiler fuses the closing interpolation of the two kernels differently -- look-down and look-up, with and
    without a Lambert surface, two aerosols + Rayleigh, three paths, layers without scattering and without opacity, a
    wavenumber count that does not fill the last tile of 64."""
    rng = np.random.default_rng(50 + nmu + 2 * int(lookup) + lowbc)
    W, G, L, NF, ncont, nth = 150, 3, 14, 3, 2, 31
    MU, WT = _c4_quadrature(nmu)
    TH = np.linspace(0.0, 180.0, nth); cth = np.cos(np.deg2rad(TH))
    ph = np.zeros((ncont, W, 2, nth))
    for c in range(ncont):
        g = 0.3 + 0.35 * c + 0.1 * np.sin(np.arange(W) / 9.0)
        ph[c, :, 0, :] = (1 - g[:, None] ** 2) / (1 + g[:, None] ** 2 - 2 * g[:, None] * cth[None, :]) ** 1.5 / (4 * np.pi)
        ph[c, :, 1, :] = cth[None, :]
    ph = np.ascontiguousarray(ph[:, :, :, ::-1])
    taus = 10.0 ** rng.uniform(-4, 1.2, (W, G, L))
    omegas = rng.uniform(0.05, 0.98, (W, G, L))
    tauray = 0.2 * taus[:, 0, :] * omegas[:, 0, :] * rng.uniform(0, 1, (W, L))
    omegas[:, :, 3] = 0.0; tauray[:, 3] = 0.0              # a layer that only absorbs
    taus[:, :, 7] = 0.0; tauray[:, 7] = 0.0                # an empty layer
    lfrac = rng.uniform(0.1, 1.0, (W, ncont, L)); lfrac /= lfrac.sum(axis=1, keepdims=True)
    WAVE = 500.0 + np.arange(W)
    bnu = 1e-7 * rng.uniform(0.5, 1.5, (W, L))
    radg = 1e-7 * rng.uniform(0.5, 1.5, (W, nmu))
    emi = np.array([20.0, 47.0, 71.0]); emi = 180.0 - emi if lookup else emi
    sol = np.array([30.0, 60.0, 100.0]); aph = np.array([0.0, 45.0, 130.0])
    brdf = np.zeros((W, nmu, nmu, NF + 1)); brdf[:, :, :, 0] = 0.3 / np.pi
    args = (ph, radg, sol, emi, np.full(W, 1e-6), aph, lowbc, brdf, MU, WT, NF, WAVE, bnu, taus, tauray, omegas, 101, 1, 1, lfrac)
    monkeypatch.delenv("ANSFM_MS_LANE", raising=False)
    by_lane = eng.scloud11wave_core(*args)
    monkeypatch.setenv("ANSFM_MS_LANE", "0")
    by_wave = eng.scloud11wave_core(*args)
    monkeypatch.delenv("ANSFM_MS_LANE", raising=False)
    assert by_lane.shape == (3, G, W) and np.all(np.isfinite(by_lane)) and np.abs(by_lane).max() > 0
    np.testing.assert_allclose(by_lane, by_wave, rtol=4 * np.finfo(float).eps, atol=0)
    assert np.mean(by_lane == by_wave) > 0.98


def _c4_quadrature(nmu):
    x, w = np.polynomial.legendre.leggauss(nmu)              # Gauss-Legendre on (0, 1): sum(mu w) = 1/2 exactly
    return 0.5 * (x + 1.0), 0.5 * w


def test_c4_full_depth_non_scattering_limit_equals_thermal_rt(eng):
    """BASELINE configs[3] depth (100 layers, 16 streams, 9 Fourier orders, G = 20, 2000 wavenumbers): with no scatterer
    the doubling / adding branch must return the plane-parallel thermal emission of the SAME gas opacities seen under the
    quadrature angles -- i.e. what the thermal branch (k_thermal_rt) gives on a path with SCALE = 1 / mu."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(404)
    W, G, L, S, NMU, NF = 2000, 20, 100, 4, 16, 8
    PRESS, TEMP, K = syn.synth_ktable(W, G, 8, 6, S, seed=12)
    _, delg = syn.gauss_legendre_01(G)
    WAVE = 400.0 + 0.5 * np.arange(W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    lay_p = np.logspace(5.5, 0.5, L); lay_t = 110.0 + 60.0 * np.linspace(1.0, 0.0, L) ** 2
    amount = 10.0 ** rng.uniform(16.5, 19.0, (S, 1)) * (lay_p[None, :] / lay_p[0]) ** 0.8
    TAUCIA = 1e-3 * (lay_p[None, :] / lay_p[0]) * (1.0 + 0.3 * np.sin(WAVE / 37.0))[:, None]
    MU, WT = _c4_quadrature(NMU)
    pick = np.array([0, 3, 7, 11, 15])
    emi = np.rad2deg(np.arccos(MU[pick]))
    c1, c2 = 1.1911e-12, 1.439
    radg = np.repeat((c1 * WAVE ** 3 / (np.exp(c2 * WAVE / lay_t[0]) - 1.0))[:, None], NMU, 1)
    spec = eng.cirsrad_ck_scatter(0, lay_p, lay_t, amount, TAUCIA, None, None, None, None, None, radg, np.full(pick.size, 40.0),
                                  emi, np.zeros(pick.size), np.zeros(W), 0, np.zeros((W, NMU, NMU, NF + 1)), MU, WT, NF, 101, 0, 0)
    NLAYIN = np.full(pick.size, L, dtype=np.int32)
    LAYINC = np.repeat(np.arange(L - 1, -1, -1, dtype=np.int32)[:, None], pick.size, 1)
    SCALE = np.repeat((1.0 / MU[pick])[None, :], L, 0)
    EMTEMP = np.repeat(lay_t[::-1][:, None], pick.size, 1)
    ref = eng.cirsrad_ck_thermal(0, lay_p, lay_t, amount, TAUCIA, NLAYIN, LAYINC, SCALE, EMTEMP, -1.0)
    assert spec.shape == ref.shape == (W, pick.size) and np.all(ref > 0)
    assert np.max(np.abs(spec - ref) / ref) < 1e-9


def test_c4_full_depth_conservative_scattering_returns_the_sunlight(eng):
    """100 conservative layers (omega = 1: Rayleigh + one aerosol, no absorber, no thermal source) over a Lambert ground of
    albedo 1, 16 streams: every photon of the solar beam comes back out of the top, 2 pi sum_i w_i mu_i I(mu_i) = mu0 F --
    the doubling (up to 12 per layer), the Hansen renormalisation and the adding of 100 layers keep the flux."""
    rng = np.random.default_rng(405)
    W, G, L, S, NMU = 256, 4, 100, 2, 16
    _, delg = np.polynomial.legendre.leggauss(G)
    delg = 0.5 * delg
    WAVE = 5000.0 + 10.0 * np.arange(W)
    eng.upload_ktable(np.zeros((W, G, 4, 3, S)), np.logspace(-6, 1, 4), np.array([50.0, 150.0, 400.0]), WAVE, delg)
    lay_p = np.logspace(5.5, 0.5, L); lay_t = np.full(L, 60.0)              # exp(-c2 nu / T) underflows: no thermal emission
    amount = np.full((S, L), 1.0e18)
    TAURAY = 10.0 ** rng.uniform(-3, -0.5, (W, L))
    TAUSCAT = 10.0 ** rng.uniform(-3, 0.3, (W, L))
    MU, WT = _c4_quadrature(NMU)
    THETA = np.linspace(0.0, 180.0, 41)
    PH = np.zeros((1, W, 2, THETA.size))
    c = np.cos(np.deg2rad(THETA)); gg = 0.6
    PH[0, :, 0, :] = (1 - gg * gg) / (1 + gg * gg - 2 * gg * c) ** 1.5 / (4 * np.pi)
    PH[:, :, 1, :] = c
    phasarr = np.ascontiguousarray(PH[:, :, :, ::-1])
    lfrac = np.ones((W, 1, L))
    solar = 10.0 ** rng.uniform(-7, -6, W)
    brdf = np.zeros((W, NMU, NMU, 1)); brdf[:, :, :, 0] = 1.0 / np.pi        # Lambert, albedo 1
    k0 = 9
    emi = np.rad2deg(np.arccos(MU))                                         # one path per stream
    spec = eng.cirsrad_ck_scatter(0, lay_p, lay_t, amount, None, TAUSCAT, TAURAY, TAUSCAT, phasarr, lfrac, np.zeros((W, NMU)),
                                  np.full(NMU, np.rad2deg(np.arccos(MU[k0]))), emi, np.zeros(NMU), solar, 1, brdf, MU, WT, 0, 101,
                                  1, 1)
    flux_up = 2.0 * np.pi * (spec * (WT * MU)[None, :]).sum(axis=1)
    # the scheme's own accuracy, not rounding: a phase function tabulated at 41 angles, renormalised by Hansen's single
    # factor per stream, first-order starting layers of optical depth 2^-12 -- measured 1e-4 .. 5e-4 over 256 wavenumbers
    assert np.max(np.abs(flux_up / (MU[k0] * solar) - 1.0)) < 1e-3


def test_c5_full_size_lbl_properties_and_slab_vs_oracle(eng, oracle):
    """BASELINE configs[4] at full size -- 1e6 grid points x 50 layers x 1e5 lines (Voigt, windows 25 / 75 cm-1): the
    spectrum of the list is the sum of the spectra of its halves (same additions in the same ascending-line order: bit
    for bit), doubles with the isotopic abundance, is non-negative; and a slab of 2000 grid points x 3 layers of the
    SAME launch agrees with the oracle's line loop (every line whose +-75 cm-1 window reaches the slab)."""
    rng = np.random.default_rng(56)
    nw, N, L = 1000000, 100000, 50
    wn = 2000.0 + 1e-3 * np.arange(nw)
    nu = np.sort(rng.uniform(1925.0, 3075.0, N))
    sw = 10.0 ** rng.uniform(-28, -19, N); el = rng.uniform(0, 3000, N)
    sr = 1.0 - np.exp(-1.4387769 * nu / 296.0)
    bp = np.stack([rng.uniform(0.02, 0.1, N), rng.uniform(0.5, 0.8, N), rng.uniform(-0.01, 0.01, N)])
    mmf = np.array([1.0])
    t = np.linspace(150.0, 300.0, L); p = np.logspace(-4, 0, L); q = np.linspace(2.0, 0.95, L)
    run = lambda sel, out, iso=0.9: eng.add_line_set_monochromatic_absorption(
        wn, 0, t, 296.0, p, 1.0, q, iso, 28.0, mmf, bp[:, sel], nu[sel], sw[sel], el[sel], sr[sel], out)
    full = np.zeros((L, nw)); run(slice(None), full)
    assert full.min() >= 0.0 and full.max() > 0.0 and np.all(np.isfinite(full))
    parts = np.zeros((L, nw)); run(slice(0, N // 2), parts); run(slice(N // 2, N), parts)
    assert np.array_equal(full, parts)
    del parts
    twice = np.zeros((L, nw)); run(slice(None), twice, iso=1.8)
    np.testing.assert_allclose(twice, 2.0 * full, rtol=1e-14)
    del twice
    # slab vs the oracle
    i0, ns, layers = 431000, 2000, [0, 24, 49]
    grid = wn[i0:i0 + ns]
    near = (nu > grid[0] - 75.0 - 0.5) & (nu < grid[-1] + 75.0 + 0.5)      # +- the pressure shift, generously
    for l in layers:
        ref = np.zeros(ns)
        oracle.add_line_set_monochromatic_absorption(grid, 0, t[l], 296.0, p[l], 1.0, q[l], 0.9, 28.0, mmf, bp[:, near], nu[near],
                                                     sw[near], el[near], sr[near], ref)
        np.testing.assert_allclose(full[l, i0:i0 + ns], ref, rtol=1e-9)


def test_gradient_gas_selection_leaves_the_selected_gradients_unchanged(eng):
    """ansfm_set_gradient_gases: with two of four gases selected the temperature and the selected gases' parameters of
    dSPECOUT are bit-identical to the full computation, the others reduce to their continuum part, SPECOUT is unchanged."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(91)
    W, G, S, L = 200, 10, 4, 14
    PRESS, TEMP, K = syn.synth_ktable(W, G, 8, 6, S, seed=31)
    _, delg = syn.gauss_legendre_01(G)
    eng.upload_ktable(K, PRESS, TEMP, 700.0 + 0.5 * np.arange(W), delg)
    atm = syn.synth_atmosphere(L, S, seed=4)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L, 20.0)
    lp, lt, am = atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0]
    EMTEMP = lt[LAYINC[:, 0]][:, None]
    NVMR, NPAR = 6, 9
    igas_map = np.array([4, 0, 2, 5], dtype=np.int32)
    dcont = 10.0 ** rng.uniform(-25, -23, (W, NPAR, L))
    args = (0, lp, lt, am, None, dcont, NVMR, NPAR, igas_map, NLAYIN, LAYINC, SCALE, EMTEMP, 300.0)
    full = eng.cirsradg_ck_thermal(*args, EMISSIVITY=np.ones(W))
    try:
        eng.set_gradient_gases([1, 3])
        part = eng.cirsradg_ck_thermal(*args, EMISSIVITY=np.ones(W))
        eng.set_gradient_gases([])
        none = eng.cirsradg_ck_thermal(*args, EMISSIVITY=np.ones(W))
    finally:
        eng.set_gradient_gases(None)
    again = eng.cirsradg_ck_thermal(*args, EMISSIVITY=np.ones(W))
    assert np.array_equal(part[0], full[0]) and np.array_equal(part[2], full[2])
    for par in (igas_map[1], igas_map[3], NVMR):                       # selected gases and temperature
        assert np.array_equal(part[1][:, par], full[1][:, par])
    for i in (0, 2):                                                   # switched off: no gas part, the continuum part stays
        par = igas_map[i]
        assert not np.array_equal(part[1][:, par], full[1][:, par])
        assert np.array_equal(part[1][:, par], none[1][:, par])
    assert np.array_equal(none[1][:, NVMR], full[1][:, NVMR])          # the temperature gradient needs no gas selected
    try:                                                               # a state vector without temperature elements
        eng.set_gradient_gases([1, 3], temperature=False)
        not_t = eng.cirsradg_ck_thermal(*args, EMISSIVITY=np.ones(W))
    finally:
        eng.set_gradient_gases(None)
    for par in (igas_map[1], igas_map[3]):
        assert np.array_equal(not_t[1][:, par], full[1][:, par])
    assert not np.array_equal(not_t[1][:, NVMR], full[1][:, NVMR]) and np.array_equal(not_t[0], full[0])
    for par in (1, 3, 7, 8):                                           # parameters no table gas maps to
        assert np.array_equal(part[1][:, par], full[1][:, par])
    assert np.array_equal(again[1], full[1])


def test_shared_gas_continuum_gradient_equals_the_full_array(eng):
    """ansfm_set_shared_gas_gradient: one (W, L) array for every gas parameter gives the dSPECOUT of the (W, NPAR, L) dTAUCON
    that repeats it NVMR times, bit for bit; the setting is consumed by one call."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(92)
    W, G, S, L = 150, 10, 3, 11
    PRESS, TEMP, K = syn.synth_ktable(W, G, 8, 6, S, seed=32)
    _, delg = syn.gauss_legendre_01(G)
    eng.upload_ktable(K, PRESS, TEMP, 650.0 + 0.5 * np.arange(W), delg)
    atm = syn.synth_atmosphere(L, S, seed=5)
    NLAYIN, LAYINC, SCALE = syn.nadir_path(L, 10.0)
    lp, lt, am = atm["lay_press_pa"][0], atm["lay_temp"][0], atm["amount"][0]
    EMTEMP = lt[LAYINC[:, 0]][:, None]
    NVMR, NPAR = 5, 8
    igas_map = np.array([3, 0, 4], dtype=np.int32)
    dray = 10.0 ** rng.uniform(-26, -24, (W, L))
    full = np.zeros((W, NPAR, L)); full[:, :NVMR, :] = dray[:, None, :]
    args = (0, lp, lt, am, None)
    rest = (NVMR, NPAR, igas_map, NLAYIN, LAYINC, SCALE, EMTEMP, -1.0)
    a = eng.cirsradg_ck_thermal(*args, full, *rest)
    b = eng.cirsradg_ck_thermal(*args, None, *rest, dtau_every_gas=dray)
    c = eng.cirsradg_ck_thermal(*args, None, *rest)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert not np.array_equal(c[1][:, 0], a[1][:, 0]) and np.array_equal(c[1][:, NVMR], a[1][:, NVMR])


def test_cirsrad_transmission_vs_oracle(eng, oracle):
    """CIRSrad's pure-transmission branch (calculate_transmission_spectrum :4110-4131): exp(-TAUTOT_PATH) of the same
    opacity assembly, g-quadrature, optional solar-flux factor -- two paths of different length, batch of two states."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(77)
    W, G, S, L = 130, 10, 3, 12
    PRESS, TEMP, K = syn.synth_ktable(W, G, 8, 6, S, seed=21)
    _, delg = syn.gauss_legendre_01(G)
    WAVE = 900.0 + 0.7 * np.arange(W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    lp = np.stack([np.logspace(4.5, 0.5, L), np.logspace(4.4, 0.6, L)]); lt = np.stack([np.linspace(200, 140, L), np.linspace(210, 150, L)])
    am = 10.0 ** rng.uniform(17, 19.5, (2, S, L)) * (lp[:, None, :] / lp[:, None, :1])
    cont = 10.0 ** rng.uniform(-4, -1, (2, W, L))
    LAYINC = np.zeros((2 * L, 2), dtype=np.int32)
    LAYINC[:, 0] = np.concatenate([np.arange(L - 1, -1, -1), np.arange(L)])        # a limb path down to layer 0 and up again
    LAYINC[:2 * (L - 4), 1] = np.concatenate([np.arange(L - 1, 3, -1), np.arange(4, L)])
    NLAYIN = np.array([2 * L, 2 * (L - 4)], dtype=np.int32)
    SCALE = np.where(np.arange(2 * L)[:, None] < NLAYIN[None, :], rng.uniform(1.0, 30.0, (2 * L, 2)), 0.0)
    solflux = 10.0 ** rng.uniform(-8, -7, W)
    got = eng.cirsrad_ck_transmission(lp, lt, am, cont, NLAYIN, LAYINC, SCALE, xfac=solflux)
    for m in range(2):
        k = oracle.calc_k(K, PRESS, TEMP, lp[m] / 101325.0, lt[m])
        tautot = oracle.k_overlap(delg, k, am[m]) + cont[m][:, None, :]
        path = np.sum(tautot[:, :, LAYINC] * SCALE, axis=2)                         # TAUTOT_PATH (:4006-4009)
        ref = np.tensordot(np.exp(-path) * solflux[:, None, None], delg, axes=([1], [0]))
        np.testing.assert_allclose(got[m], ref, rtol=1e-11)
    # the states of a solar-occultation Jacobian (jacobian_nemesis(nemesisSO=True), staged route): six states, four of them one
    # layer away from the first -- the batch build of the RT kernel, shared opacity rows; every state equals its own call
    n = 6
    lp6 = np.repeat(lp[:1], n, 0); lt6 = np.repeat(lt[:1], n, 0); am6 = np.repeat(am[:1], n, 0); cont6 = np.repeat(cont[:1], n, 0)
    lt6[1, 3] *= 1.02; am6[2, 1, 7] *= 1.05; cont6[3, :, 5] *= 1.01; lp6[4, 9] *= 1.001
    SC6 = np.repeat(SCALE[None], n, 0); SC6[5, 2, 0] *= 1.01
    batch = eng.cirsrad_ck_transmission(lp6, lt6, am6, cont6, NLAYIN, LAYINC, SC6, xfac=solflux)
    rows, total = eng.last_layer_rows()
    assert total == n * L and rows < total
    for m in range(n):
        one = eng.cirsrad_ck_transmission(lp6[m], lt6[m], am6[m], cont6[m], NLAYIN, LAYINC, SC6[m], xfac=solflux)
        assert np.array_equal(batch[m], one), m
    assert not any(np.array_equal(batch[m], batch[0]) for m in range(1, n))


def test_cirsradg_transmission_vs_oracle(eng, oracle):
    """The same branch with return_grad (:4128-4131, then the g-quadrature and nan_to_num of :4504-4507):
    dSPECOUT = -SPECOUT * dTAUTOT_LAYINC, the opacity gradients assembled as for thermal emission (gas slots x 1e-4 through
    the gas map, temperature slot, continuum gradients, x SCALE).  Restated with the oracle's calc_kg + k_overlapg."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(78)
    W, G, S, L = 130, 10, 3, 12
    PRESS, TEMP, K = syn.synth_ktable(W, G, 8, 6, S, seed=22)
    _, delg = syn.gauss_legendre_01(G)
    WAVE = 900.0 + 0.7 * np.arange(W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    lp = np.stack([np.logspace(4.5, 0.5, L), np.logspace(4.4, 0.6, L)]); lt = np.stack([np.linspace(200, 140, L), np.linspace(210, 150, L)])
    am = 10.0 ** rng.uniform(17, 19.5, (2, S, L)) * (lp[:, None, :] / lp[:, None, :1])
    cont = 10.0 ** rng.uniform(-4, -1, (2, W, L))
    NVMR, NDUST = 4, 1
    NPAR = NVMR + 2 + NDUST
    igas_map = np.array([2, 0, 3], dtype=np.int32)
    dcont = 10.0 ** rng.uniform(-24, -22, (2, W, NPAR, L))
    LAYINC = np.zeros((2 * L, 2), dtype=np.int32)
    LAYINC[:, 0] = np.concatenate([np.arange(L - 1, -1, -1), np.arange(L)])
    LAYINC[:2 * (L - 4), 1] = np.concatenate([np.arange(L - 1, 3, -1), np.arange(4, L)])
    NLAYIN = np.array([2 * L, 2 * (L - 4)], dtype=np.int32)
    SCALE = np.where(np.arange(2 * L)[:, None] < NLAYIN[None, :], rng.uniform(1.0, 30.0, (2 * L, 2)), 0.0)
    solflux = 10.0 ** rng.uniform(-8, -7, W)
    spec, dspec = eng.cirsradg_ck_transmission(lp, lt, am, cont, dcont, NVMR, NPAR, igas_map, NLAYIN, LAYINC, SCALE, xfac=solflux)
    fwd = eng.cirsrad_ck_transmission(lp, lt, am, cont, NLAYIN, LAYINC, SCALE, xfac=solflux)
    np.testing.assert_allclose(spec, fwd, rtol=1e-13)
    inside = (np.arange(2 * L)[:, None] < NLAYIN[None, :])
    for m in range(2):
        k, dkdT = oracle.calc_k(K, PRESS, TEMP, lp[m] / 101325.0, lt[m], grad=True)
        tau, dk = oracle.k_overlapg(delg, k, dkdT, am[m])                        # (W,G,L), (W,G,L,S+1)
        dtau = np.zeros((W, G, NPAR, L))
        for i in range(S):
            dtau[:, :, igas_map[i], :] = dk[:, :, :, i] * 1.0e-4                   # :3868-3870
        dtau[:, :, NVMR, :] = dk[:, :, :, S]                                     # :3872
        dtau += dcont[m][:, None, :, :]
        tautot = tau + cont[m][:, None, :]
        path = np.sum(tautot[:, :, LAYINC] * SCALE, axis=2)                       # (W,G,P)
        sg = np.exp(-path) * solflux[:, None, None]
        dlay = dtau[:, :, :, LAYINC] * SCALE                                      # (W,G,NPAR,LIMAX,P)
        dref = np.nan_to_num(np.tensordot(-sg[:, :, None, None, :] * dlay, delg, axes=([1], [0])))
        dref = dref * inside[None, None, :, :]
        np.testing.assert_allclose(spec[m], np.tensordot(sg, delg, axes=([1], [0])), rtol=1e-11)
        scale = np.max(np.abs(dref), axis=(0, 2), keepdims=True)
        assert np.max(np.abs(dspec[m] - dref) / np.where(scale > 0, scale, 1.0)) < 1e-10


def test_transmission_with_lbl_tables_vs_oracle(eng, oracle):
    """The transmission branch on line-by-line tables (ILBL = LINE_BY_LINE_TABLES, NG = 1: the solar-occultation case),
    forward and with gradients, against calc_klbl / calc_klblg composed as calculate_gaseous_line_opacity does (:3795-3817)."""
    rng = np.random.default_rng(31)
    W, NP, NT, S, L = 260, 7, 6, 3, 14
    PRESS = np.logspace(-6, 1.1, NP); TEMP = np.linspace(80.0, 420.0, NT)
    K = 10.0 ** rng.uniform(-27, -21, size=(W, NP, NT, S))
    WAVE = 2500.0 + 0.005 * np.arange(W)
    lp = np.logspace(4.0, -0.5, L); lt = np.linspace(230.0, 150.0, L)
    am = 10.0 ** rng.uniform(20.5, 23.0, (S, 1)) * (lp[None, :] / lp[0])
    cont = 10.0 ** rng.uniform(-4, -1.5, (W, L))
    LAYINC = np.concatenate([np.arange(L - 1, 2, -1), np.arange(3, L)]).astype(np.int32)[:, None]    # limb path, tangent layer 3
    NLAYIN = np.array([LAYINC.shape[0]], dtype=np.int32)
    SCALE = rng.uniform(1.0, 40.0, LAYINC.shape)
    eng.upload_lbltable(K, PRESS, TEMP, WAVE)
    NVMR, NPAR = S + 1, S + 3
    igas_map = np.array([1, 3, 0], dtype=np.int32)
    fwd = eng.cirsrad_ck_transmission(lp, lt, am, cont, NLAYIN, LAYINC, SCALE)
    spec, dspec = eng.cirsradg_ck_transmission(lp, lt, am, cont, None, NVMR, NPAR, igas_map, NLAYIN, LAYINC, SCALE)
    k, dkdT = oracle.calc_klbl(K, PRESS, TEMP, lp / 101325.0, lt, grad=True)              # (W,L,S)
    tau = cont.copy()
    dtau = np.zeros((W, NPAR, L))
    for i in range(S):
        tau = tau + k[:, :, i] * am[i][None, :]
        dtau[:, igas_map[i], :] = k[:, :, i] * 1.0e-4
        dtau[:, NVMR, :] += dkdT[:, :, i] * am[i][None, :]
    tr = np.exp(-np.sum(tau[:, LAYINC[:, 0]] * SCALE[:, 0], axis=1))
    np.testing.assert_allclose(fwd[:, 0], tr, rtol=1e-11)
    np.testing.assert_allclose(spec[:, 0], tr, rtol=1e-11)
    dref = -tr[:, None, None] * dtau[:, :, LAYINC[:, 0]] * SCALE[None, None, :, 0]
    scale = np.max(np.abs(dref), axis=(0, 2), keepdims=True)
    assert np.max(np.abs(dspec[:, :, :, 0] - dref) / np.where(scale > 0, scale, 1.0)) < 1e-10


def test_singlescatt_plane_spectrum_golden(eng, golden_dir):
    """Array-level calc_singlescatt_plane_spectrum (:6509-6600) vs the reference (golden): both spectral units, without /
    with a surface, a grazing geometry."""
    z = _load(golden_dir, "singlescatt")
    for ispace, tag in ((0, "wn"), (1, "wl")):
        for cn in ("nosurf", "surf", "graze"):
            TSURF, sa, ea = z[f"{tag}_{cn}_args"]
            s = eng.calc_singlescatt_plane_spectrum(ispace, z[f"{tag}_WAVE"], z[f"{tag}_TAU"], z[f"{tag}_TEMP"], z[f"{tag}_OMEGA"],
                                                    z[f"{tag}_PHASE"], TSURF, z[f"{tag}_EMIS"], z[f"{tag}_BRDF"], z[f"{tag}_SOL"], sa, ea)
            np.testing.assert_allclose(s, z[f"{tag}_{cn}_spec"], rtol=1e-11, err_msg=f"{tag} {cn}")


def test_cirsrad_singlescatt_vs_oracle(eng, oracle):
    """CIRSrad's single-scattering branch fused with the opacity assembly (ansfm_cirsrad_ck_singlescatt) vs the reference's
    recipe on the oracle's pieces: calc_k + k_overlap, TAUTOT (:3989), OMEGA (:4276-4283), LAYINC x SCALE (:4006),
    calc_singlescatt_plane_spectrum per path, xfac, g-quadrature (:4504).  Two paths of different length and geometry."""
    from archnemesis_dist_amd import synthetic as syn
    rng = np.random.default_rng(88)
    W, G, S, L = 100, 10, 3, 9
    PRESS, TEMP, K = syn.synth_ktable(W, G, 8, 6, S, seed=31)
    _, delg = syn.gauss_legendre_01(G)
    WAVE = 2000.0 + 3.0 * np.arange(W)
    eng.upload_ktable(K, PRESS, TEMP, WAVE, delg)
    lp = np.logspace(5.0, 1.0, L); lt = np.linspace(230.0, 150.0, L)
    am = 10.0 ** rng.uniform(17, 19.5, (S, L)) * (lp[None, :] / lp[0])
    TAURAY = 10.0 ** rng.uniform(-4, -2, (W, L)); TAUSCAT = 10.0 ** rng.uniform(-3, -1, (W, L)); TAUSCAT[:, 4] = 0.0
    TAUDUST = TAUSCAT * 1.2; TAUCIA = 10.0 ** rng.uniform(-5, -3, (W, L))
    cont = TAUCIA + TAUDUST + TAURAY
    P = 2
    phase = 10.0 ** rng.uniform(-1.5, 0.3, (P, W, L))
    LAYINC = np.zeros((L, P), dtype=np.int32)
    LAYINC[:, 0] = np.arange(L - 1, -1, -1); LAYINC[:L - 2, 1] = np.arange(L - 1, 1, -1)
    NLAYIN = np.array([L, L - 2], dtype=np.int32)
    sol = np.array([30.0, 55.0]); emi = np.array([10.0, 40.0])
    SCALE = np.where(np.arange(L)[:, None] < NLAYIN[None, :], 1.0 / np.cos(np.deg2rad(emi))[None, :], 0.0)
    EMTEMP = np.where(np.arange(L)[:, None] < NLAYIN[None, :], lt[LAYINC], 0.0)
    EMIS = rng.uniform(0.7, 1.0, W); BRDF = rng.uniform(0.0, 0.15, (W, P)); SOLF = 10.0 ** rng.uniform(-8, -7, W)
    xfac = rng.uniform(0.5, 2.0, W)
    for TSURF in (-1.0, 260.0):
        got = eng.cirsrad_ck_singlescatt(0, lp, lt, am, cont, TAURAY + TAUSCAT, phase, NLAYIN, LAYINC, SCALE, EMTEMP, TSURF, EMIS,
                                         BRDF, SOLF, sol, emi, xfac=xfac)
        k = oracle.calc_k(K, PRESS, TEMP, lp / 101325.0, lt)
        tautot = oracle.k_overlap(delg, k, am) + cont[:, None, :]
        omega = np.where(tautot > 0, (TAURAY + TAUSCAT)[:, None, :] / np.where(tautot > 0, tautot, 1.0), 0.0)
        ref = np.zeros((W, P))
        for ip in range(P):
            n = NLAYIN[ip]; li = LAYINC[:n, ip]
            tpath = tautot[:, :, li] * SCALE[:n, ip]
            sp = oracle.calc_singlescatt_plane_spectrum(0, WAVE, tpath, EMTEMP[:n, ip], omega[:, :, li], phase[ip][:, li], TSURF,
                                                        EMIS, BRDF[:, ip], SOLF, sol[ip], emi[ip])
            ref[:, ip] = np.tensordot(sp * xfac[:, None], delg, axes=([1], [0]))
        np.testing.assert_allclose(got, ref, rtol=1e-11)
