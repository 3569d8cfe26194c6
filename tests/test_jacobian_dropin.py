"""`jacobian_nemesis` as a drop-in (archnemesis_dist_amd.jacobian_dropin) against the reference's own Jacobian harness.

tests/golden/jacobian_c1.npz is the output of the REFERENCE's `jacobian_nemesis(analytical_gradient=False)`
(ForwardModel_0.py:2184-2361) on the C1 inputs cut to 40 convolution points and 12 free temperature levels, run with two
loky workers (oracle/gen_golden_jacobian.py), plus what each of its 13 forward models handed to CIRSrad.

  * CPU, no reference: the oracle's CIRSrad on the 13 recorded states -> YNtot, KK vs the fixture (pins the chain the
    other Jacobian tests use as their checker, oracle/jacobian_twin.py's `cirsrad_ck_thermal` leg, to the reference);
  * CPU, reference present: the subclass from `make_gpu_forward_model` (engine double) called the way coreretOE calls it,
    every route, vs the fixture;
  * GPU: the recorded states as ONE batched call through the C-ABI -> KK vs the fixture, de-duplication on / off."""
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
needs_reference = [pytest.mark.needs_reference,
                   pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")]


def _load(golden_dir, name="jacobian_c1.npz"):
    return np.load(os.path.join(golden_dir, name))


def _kk_from_spectra(z, SPECOUT):
    """Measurement vectors and KK from per-state spectra (nfm, NWAVE, 1) the reference's way: conv with FWHM = 0 is a
    linear interpolation onto VCONV (Measurement_0.py:2330-2336), then the quotient of :2348-2359."""
    Y = np.stack([np.interp(z["VCONV"], z["WAVE"], s[:, 0]) for s in SPECOUT], axis=1)
    XN, inum = z["XN"], z["inum"]
    KK = np.zeros((Y.shape[0], XN.size))
    for i, ix in enumerate(inum):
        xn1 = XN[ix] * 1.05
        if xn1 == 0.0:
            xn1 = 0.05
        KK[:, ix] = (Y[:, i + 1] - Y[:, 0]) / (xn1 - XN[ix])
    return Y, KK


def _assert_kk(KK, z, tol, tight=None):
    """Each free column against the fixture, relative to that column's largest element (the contract: 1e-4), plus the
    floor a finite difference cannot go below: a few units of round-off of the spectrum divided by the step (the top
    level's column is ten decades below the largest one -- its whole signal is 1e-12 of the radiance)."""
    eps = np.finfo(float).eps
    for ix in z["inum"]:
        scale = np.abs(z["KK"][:, ix]).max()
        assert scale > 0
        floor = 64 * eps * np.abs(z["YN"]).max() / abs(0.05 * z["XN"][ix])
        err = np.abs(KK[:, ix] - z["KK"][:, ix]).max()
        assert err <= tol * scale + floor, (ix, err / scale)
        if tight is not None and scale >= 1e-6 * np.abs(z["KK"]).max():       # measured: <= 3e-11 on these columns
            assert err <= tight * scale, (ix, err / scale)
    fixed = np.setdiff1d(np.arange(KK.shape[1]), z["inum"])
    assert not KK[:, fixed].any()


def test_fixture_is_what_the_reference_documents(golden_dir):
    """Shape of the harness and the one reference quirk the fixture records: with NCores = 1 joblib runs
    execute_fm in the caller's process, `Variables.XN` is left at the last perturbed state (:2154) and the last column's
    quotient (:2355-2359) is formed with the perturbed value -> that column is the two-worker result / 1.05."""
    z = _load(golden_dir)
    NX = z["XN"].size
    assert z["xnx"].shape == (NX, NX + 1) and z["KK"].shape == (40, NX) and z["YNtot"].shape == (40, 13)
    assert np.array_equal(z["ixrun"], np.concatenate([[0], z["inum"] + 1]))
    assert np.array_equal(z["xnx"][:, 0], z["XN"])
    for i in z["inum"]:
        assert z["xnx"][i, i + 1] == z["XN"][i] + 0.05 * z["XN"][i]
    last = z["inum"][-1]
    assert np.array_equal(z["KK_ncores1"][:, z["inum"][:-1]], z["KK"][:, z["inum"][:-1]])
    np.testing.assert_allclose(z["KK_ncores1"][:, last] * 1.05, z["KK"][:, last], rtol=1e-12)
    from archnemesis_dist_amd.jacobian import perturbed_states
    assert np.array_equal(perturbed_states(z["XN"], 0.05 * z["XN"]), z["xnx"])       # the product's (:2234-2242)
    Y, KK = _kk_from_spectra(z, z["SPECOUT"])             # the reference's own spectra through this file's quotient
    np.testing.assert_allclose(Y, z["YNtot"], rtol=1e-13)
    _assert_kk(KK, z, 1e-12)


def test_oracle_chain_reproduces_the_reference_jacobian(oracle, golden_dir):
    z = _load(golden_dir)
    spec = [oracle.cirsrad_ck_thermal(int(z["ISPACE"]), z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"], z["LAY_PRESS"][m],
                                      z["LAY_TEMP"][m], np.ascontiguousarray(z["LAY_AMOUNT"][m].T) * 1.0e-4, z["TAUCONT"][m],
                                      z["NLAYIN"], z["LAYINC"], z["SCALE"][m], z["EMTEMP"][m], float(z["TSURF"]))
            for m in range(z["LAY_PRESS"].shape[0])]
    np.testing.assert_allclose(np.stack(spec), z["SPECOUT"], rtol=2e-7)       # float32 table grids: NumPy's float32 log
    Y, KK = _kk_from_spectra(z, spec)
    np.testing.assert_allclose(Y[:, 0], z["YN"], rtol=2e-7)
    _assert_kk(KK, z, 1e-4)                                                    # contract; measured below


@pytest.mark.gpu
def test_gpu_batched_replay_of_the_reference_jacobian(golden_dir):
    """The 13 forward models of the reference's run as ONE batched call through the C-ABI (layers shared with the
    unperturbed state are not recomputed) -> KK within 1e-4 of each column (measured ~1e-7: finite differences amplify
    the float32-log 2e-7 of the spectra), identical with de-duplication off."""
    from archnemesis_dist_amd.engine import AnsfmEngine
    z = _load(golden_dir)
    eng = AnsfmEngine(0)
    eng.upload_ktable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"])
    amount = np.ascontiguousarray(np.transpose(z["LAY_AMOUNT"], (0, 2, 1))) * 1.0e-4
    args = (int(z["ISPACE"]), z["LAY_PRESS"], z["LAY_TEMP"], amount, z["TAUCONT"], z["NLAYIN"], z["LAYINC"], z["SCALE"], z["EMTEMP"],
            np.full(z["LAY_PRESS"].shape[0], float(z["TSURF"])))
    spec = eng.cirsrad_ck_thermal(*args)
    rows = eng.last_layer_rows()
    np.testing.assert_allclose(spec, z["SPECOUT"], rtol=2e-7)
    Y, KK = _kk_from_spectra(z, spec)
    _assert_kk(KK, z, 1e-4)
    assert rows[1] == 13 * 71 and 71 <= rows[0] < rows[1]      # hydrostatic re-adjustment moves the layers above a level
    eng.set_layer_dedup(False)
    try:
        spec2 = eng.cirsrad_ck_thermal(*args)
    finally:
        eng.set_layer_dedup(True)
    assert np.array_equal(spec, spec2)


# ---- the scattering configuration: ISCAT = 1 forces the numerical route (ForwardModel_0.py:2251-2252) -------------------------
def _scatter_oracle_spectra(oracle, z):
    """CIRSrad's scattering branch for every recorded forward model through the oracle's restatements (calc_k + k_overlap,
    TAUTOT :3989, OMEGA / BB :5099-5119, scloud11wave_core, g-quadrature :4504)."""
    out = []
    for m in range(z["LAY_PRESS"].shape[0]):
        k = oracle.calc_k(z["K"], z["TPRESS"], z["TTEMP"], z["LAY_PRESS"][m] / 101325.0, z["LAY_TEMP"][m])
        tg = oracle.k_overlap(z["DELG"], k, np.ascontiguousarray(z["LAY_AMOUNT"][m].T) * 1.0e-4)
        tautot = tg + z["TAUCIA"][m][:, None, :] + z["TAUDUST"][m][:, None, :] + z["TAURAY"][m][:, None, :]
        omega = np.zeros_like(tautot)
        pos = tautot > 0
        omega[pos] = np.broadcast_to((z["TAURAY"][m] + z["TAUSCAT"][m])[:, None, :], tautot.shape)[pos] / tautot[pos]
        bnu = np.stack([oracle.planck(int(z["ISPACE"]), z["WAVE"], t) for t in z["LAY_TEMP"][m]], axis=1)
        rad = oracle.scloud11wave_core(z["PHASARR"][m], z["RADG"][m], z["SOL_ANG"], z["EMISS_ANG"], z["SOLAR"], z["AZI_ANG"],
                                       int(z["LOWBC"]), z["BRDF"], z["MU"], z["WTMU"], int(z["NF"]), z["WAVE"], bnu, tautot,
                                       z["TAURAY"][m], omega, int(z["NPHI"]), int(z["IRAY"]), int(z["IMIE"]), z["LFRAC"][m])
        out.append(np.tensordot(np.transpose(rad, (2, 1, 0)), np.asarray(z["DELG"], dtype=float), axes=([1], [0])))
    return np.stack(out)


def test_scattering_fixture_and_oracle_chain(oracle, golden_dir):
    """jacobian_c4.npz: the reference's jacobian_nemesis on its multiple-scattering inputs (three temperature levels, two
    parameters of the aerosol profile; six multiple-scattering forward models).  The oracle's chain on the recorded inputs
    gives the reference's spectra and, through the quotient, its KK."""
    z = _load(golden_dir, "jacobian_c4.npz")
    assert int(z["IMOD"][0]) & 256 and z["KK"].shape == (6, 92) and z["SPECOUT"].shape[0] == 6
    Y, KK = _kk_from_spectra(z, z["SPECOUT"])
    np.testing.assert_allclose(Y, z["YNtot"], rtol=1e-13)
    _assert_kk(KK, z, 1e-12)
    spec = _scatter_oracle_spectra(oracle, z)
    np.testing.assert_allclose(spec, z["SPECOUT"], rtol=5e-7)          # float32 table grids (2e-7 on k) through the core
    _assert_kk(_kk_from_spectra(z, spec)[1], z, 1e-4)


@pytest.mark.gpu
def test_gpu_batched_scattering_jacobian_replays_the_reference(golden_dir):
    """The six multiple-scattering forward models of the reference's run as ONE ansfm_cirsrad_ck_scatter_batch call -> KK within
    1e-4 of each column of the reference's; every spectrum equal to a call of its own, bit for bit."""
    from archnemesis_dist_amd.engine import AnsfmEngine
    z = _load(golden_dir, "jacobian_c4.npz")
    eng = AnsfmEngine(0)
    eng.upload_ktable(z["K"], z["TPRESS"], z["TTEMP"], z["WAVE"], z["DELG"])
    amount = np.ascontiguousarray(np.transpose(z["LAY_AMOUNT"], (0, 2, 1))) * 1.0e-4
    tail = (z["SOL_ANG"], z["EMISS_ANG"], z["AZI_ANG"], z["SOLAR"], int(z["LOWBC"]), z["BRDF"], z["MU"], z["WTMU"], int(z["NF"]),
            int(z["NPHI"]), int(z["IRAY"]), int(z["IMIE"]))
    spec = eng.cirsrad_ck_scatter_batch(int(z["ISPACE"]), z["LAY_PRESS"], z["LAY_TEMP"], amount, z["TAUCIA"], z["TAUDUST"], z["TAURAY"],
                                        z["TAUSCAT"], z["PHASARR"][0], z["LFRAC"], z["RADG"], *tail)
    np.testing.assert_allclose(spec, z["SPECOUT"], rtol=5e-7)
    _assert_kk(_kk_from_spectra(z, spec)[1], z, 1e-4)
    for m in range(spec.shape[0]):
        one = eng.cirsrad_ck_scatter(int(z["ISPACE"]), z["LAY_PRESS"][m], z["LAY_TEMP"][m], amount[m], z["TAUCIA"][m], z["TAUDUST"][m],
                                     z["TAURAY"][m], z["TAUSCAT"][m], z["PHASARR"][0], z["LFRAC"][m], z["RADG"][m], *tail)
        assert np.array_equal(one, spec[m])


@pytest.fixture()
def c1_cut(oracle, monkeypatch):
    """The cut C1 case in a scratch directory, the reference imported, the adapter's engine replaced by the oracle double."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.ref_import import import_reference
    from oracle import gen_golden_jacobian as gj
    from test_dropin_reference import OracleEngineDouble
    import archnemesis_dist_amd.forward_model as fmod
    ans = import_reference()
    work = tempfile.mkdtemp(prefix="ansfm_jacdrop_")
    gj.setup_c1(ans, work)
    cwd = os.getcwd()
    os.chdir(work)
    double = OracleEngineDouble(oracle)
    monkeypatch.setattr(fmod, "get_engine", lambda device=0: double)
    fmod.set_strict(True)
    fmod.reset_summary()
    try:
        yield ans, gj, fmod, double
    finally:
        fmod.set_strict(False)
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


@pytest.mark.parametrize("route", ["auto", "profile", "staged", "loop"])
def test_jacobian_nemesis_through_the_subclass_matches_the_reference(c1_cut, golden_dir, route):
    """What coreretOE does (OptimalEstimation_0.py:1318-1333): build the forward model, call jacobian_nemesis(NCores, ...).
    The subclass takes it without joblib workers.  "auto" lands on the profile route here (one continuous temperature
    profile; hydrostatic re-adjustment, CIA, aerosol and Rayleigh continuum are all inside it): one layer_average launch and
    ONE batched CIRSrad call for the 13 states; "staged" runs the reference's subprofretg / calc_path per state and batches
    CIRSrad; "loop" is execute_fm per column."""
    ans, gj, fmod, double = c1_cut
    z = _load(golden_dir)
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    fm = gj.cut_case(ans, cls=FMGPU)
    fm.ansfm_jacobian_route = route
    XN0 = np.array(fm.Variables.XN)
    with pytest.warns(RuntimeWarning) if route == "loop" else _nullcontext():
        YN, KK = fm.jacobian_nemesis(NCores=4, analytical_gradient=False)
    info = fm.ansfm_last_jacobian
    assert info["route"] == ("profile" if route == "auto" else route) and info["nfm"] == 13
    assert np.array_equal(fm.Variables.XN, XN0)                # never left at a perturbed state
    np.testing.assert_allclose(YN, z["YN"], rtol=2e-7)
    _assert_kk(KK, z, 1e-4, tight=1e-8)
    if route in ("auto", "profile"):
        assert double.dev_batches == [13] and not getattr(double, "batch_sizes", [])
    elif route == "staged":
        assert double.batch_sizes == [13]
    else:
        assert not getattr(double, "batch_sizes", []) and any("NCores" in k for k in fmod.summary()["notes"])
    assert fmod.summary()["delegated"] == {}


class _nullcontext:
    def __enter__(self): return None
    def __exit__(self, *a): return False


test_jacobian_nemesis_through_the_subclass_matches_the_reference = pytest.mark.needs_reference(
    pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")(
        test_jacobian_nemesis_through_the_subclass_matches_the_reference))


def _scattering_dropin(c1_cut, golden_dir, route):
    import shutil
    ans, gj, fmod, double = c1_cut
    from oracle import gen_golden_jacobian_ms as gm
    z = _load(golden_dir, "jacobian_c4.npz")
    work = os.getcwd()
    gj.setup_c1(ans, work, seed=4, case=gm.CASE)                # the scattering inputs over the scratch directory
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    fm = gj.cut_case(ans, cls=FMGPU, nkeep=gm.NKEEP, free=gm.FREE)
    fm.ansfm_jacobian_route = route
    YN, KK = fm.jacobian_nemesis(NCores=1, analytical_gradient=True)      # ISCAT = 1: numerical whatever is asked (:2251)
    info = fm.ansfm_last_jacobian
    assert info["nfm"] == 6 and info["analytic_columns"] == 0
    np.testing.assert_allclose(YN, z["YN"], rtol=5e-7)
    _assert_kk(KK, z, 1e-4, tight=1e-6)
    return info, double


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_scattering_jacobian_through_the_subclass_matches_the_reference(c1_cut, golden_dir):
    """ISCAT = MULTIPLE_SCATTERING: `auto` cannot take the profile route (scattering, an aerosol model) and lands on the staged
    one -- the reference's subprofretg / calc_path / continuum per state, then ONE batched scattering call for the six
    forward models (here: the engine double) -- with the reference's KK."""
    info, double = _scattering_dropin(c1_cut, golden_dir, "auto")
    assert info["route"] == "staged" and double.scatter_batches == [6]


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_routes_agree_on_two_geometries_with_fov_averaging(c1_cut):
    """NGEOM = 2 with different numbers of convolution points, the second geometry a field-of-view average of two emission
    angles (NAV = 2, weights 0.3 / 0.7): the measurement vector packs the geometries one after the other (execute_fm
    :2171-2174), SPEC is the weighted sum over the averaging points (:531).  The profile route (everything batched), the staged
    route (the reference's host code per state) and the loop route (the reference's nemesisfm per column) give the same KK."""
    ans, gj, fmod, double = c1_cut
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    free = (5, 30, 60)
    res = {}
    for route in ("profile", "staged", "loop"):
        fm = gj.cut_case(ans, cls=FMGPU, nkeep=12, free=free)
        M = fm.Measurement
        n0, n1 = 12, 9
        two = lambda a, b: np.stack([a, b], axis=1)
        pad = lambda v: np.concatenate([v[:n1], np.zeros(n0 - n1)])
        M.NGEOM = 2
        M.NCONV = np.array([n0, n1], dtype="int32")
        M.NAV = np.array([1, 2], dtype="int32")
        M.VCONV = two(M.VCONV[:n0, 0], pad(M.VCONV[:n0, 0] + 1.0))
        M.MEAS = two(M.MEAS[:n0, 0], pad(M.MEAS[:n0, 0])); M.ERRMEAS = two(M.ERRMEAS[:n0, 0], pad(M.ERRMEAS[:n0, 0]))
        z2 = np.zeros((2, 2))
        M.FLAT, M.FLON, M.SOL_ANG, M.AZI_ANG = z2.copy(), z2.copy(), z2.copy(), z2.copy()
        M.EMISS_ANG = np.array([[0.0, 0.0], [25.0, 40.0]])
        M.WGEOM = np.array([[1.0, 0.0], [0.3, 0.7]])
        M.NY = n0 + n1
        fm.ansfm_jacobian_route = route
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            YN, KK = fm.jacobian_nemesis(NCores=1, analytical_gradient=False)
        assert fm.ansfm_last_jacobian["route"] == route and YN.shape == (n0 + n1,) and KK.shape == (n0 + n1, 81)
        res[route] = (YN, KK)
    assert np.array_equal(res["staged"][0], res["loop"][0]) and np.array_equal(res["staged"][1], res["loop"][1])
    np.testing.assert_allclose(res["profile"][0], res["staged"][0], rtol=1e-13)
    for ix in free:
        sc = np.abs(res["staged"][1][:, ix]).max()
        assert sc > 0 and np.abs(res["profile"][1][:, ix] - res["staged"][1][:, ix]).max() <= 1e-9 * sc
    assert not np.array_equal(res["staged"][0][:9], res["staged"][0][12:])      # the second geometry is a different spectrum


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_routes_agree_on_a_limb_geometry(c1_cut):
    """EMISS_ANG < 0 is the reference's limb flag (calc_path :2995-2999: LAYHT = tangent height, LAYANG = 90, the ray goes
    down to the tangent layer and up again).  With the hydrostatic re-adjustment on, every state has its own heights, layer
    grid and slant factors; the three routes give the same KK."""
    ans, gj, fmod, double = c1_cut
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    free = (20, 45, 70)
    res = {}
    for route in ("profile", "staged", "loop"):
        fm = gj.cut_case(ans, cls=FMGPU, nkeep=10, free=free)
        M = fm.Measurement
        M.EMISS_ANG = np.array([[-1.0]]); M.TANHE = np.array([[60.0]]); M.SOL_ANG = np.array([[60.0]])
        fm.ansfm_jacobian_route = route
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            YN, KK = fm.jacobian_nemesis(NCores=1, analytical_gradient=False)
        assert fm.ansfm_last_jacobian["route"] == route
        res[route] = (YN, KK)
    assert np.array_equal(res["staged"][1], res["loop"][1])
    np.testing.assert_allclose(res["profile"][0], res["staged"][0], rtol=1e-12)
    XN = np.array(fm.Variables.XN)
    for ix in free:      # 1e-8 of the column, or the floor of a finite difference: a few units of round-off of YN over the step
        sc = np.abs(res["staged"][1][:, ix]).max()
        floor = 64 * np.finfo(float).eps * np.abs(res["staged"][0]).max() / abs(0.05 * XN[ix])
        assert np.abs(res["profile"][1][:, ix] - res["staged"][1][:, ix]).max() <= 1e-8 * sc + floor, ix


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
@pytest.mark.parametrize("flag", ["nemesisL", "nemesisSO"])
def test_limb_forward_model_jacobian_staged_equals_the_reference_loop(c1_cut, flag):
    """jacobian_nemesis(nemesisL=True): every forward model is the reference's nemesisLfm (all tangent paths of a state in one
    CIRSrad call, interpolation to the three tangent heights of the measurement, convolution over all geometries).  The
    staged route -- nemesisLfm's host code per state, ONE batched engine call with NPATH paths per state -- against the loop
    route, which IS the reference's execute_fm / nemesisLfm per column: the same YN and KK to the last bit.  nemesisSO=True:
    the same for nemesisSOfm (solar occultation: calc_path_SO, CIRSrad's transmission branch)."""
    ans, gj, fmod, double = c1_cut
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    free = (20, 45, 70)
    res = {}
    for route in ("auto", "loop"):
        fm = gj.cut_case(ans, cls=FMGPU, nkeep=10, free=free)
        M = fm.Measurement
        n0, ng = 10, 3
        rep = lambda a: np.repeat(np.asarray(a)[:n0, 0:1], ng, axis=1)
        M.NGEOM = ng
        M.NCONV = np.array([n0] * ng, dtype="int32")
        M.NAV = np.ones(ng, dtype="int32")
        M.VCONV = rep(M.VCONV); M.MEAS = rep(M.MEAS); M.ERRMEAS = rep(M.ERRMEAS)
        z = np.zeros((ng, 1))
        M.FLAT, M.FLON, M.AZI_ANG = z.copy(), z.copy(), z.copy()
        M.SOL_ANG = np.full((ng, 1), 60.0)
        M.EMISS_ANG = np.full((ng, 1), -1.0)
        M.TANHE = np.array([[40.0], [80.0], [130.0]])
        M.WGEOM = np.ones((ng, 1))
        M.NY = n0 * ng
        fm.ansfm_jacobian_route = route
        double.batch_sizes = []
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            YN, KK = fm.jacobian_nemesis(NCores=1, analytical_gradient=False, **{flag: True})
        info = fm.ansfm_last_jacobian
        assert info["route"] == ("staged" if route == "auto" else "loop") and info["nfm"] == 4
        res[route] = (YN, KK, list(double.batch_sizes))
    assert np.array_equal(res["auto"][0], res["loop"][0]) and np.array_equal(res["auto"][1], res["loop"][1])
    assert res["auto"][2] == [4] and res["loop"][2] == []                        # one batched call of four states / none
    for ix in free:
        assert np.abs(res["auto"][1][:, ix]).max() > 0
    assert not np.array_equal(res["auto"][0][:10], res["auto"][0][10:20])         # the tangent heights see different spectra


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_disc_forward_model_jacobian_staged_equals_the_reference_loop(c1_cut):
    """jacobian_nemesis(nemesisdisc=True): nemesisdiscfm is nemesisfm with the averaging points of a geometry run through
    process_IAV and summed at once.  Two geometries, the second a field-of-view average of three emission angles: the staged
    route (one batched call per averaging point) equals the loop route (the reference's nemesisdiscfm per column) bit for bit."""
    ans, gj, fmod, double = c1_cut
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    free = (5, 30, 60)
    res = {}
    for route in ("auto", "loop"):
        fm = gj.cut_case(ans, cls=FMGPU, nkeep=12, free=free)
        M = fm.Measurement
        n0, n1 = 12, 9
        two = lambda a, b: np.stack([a, b], axis=1)
        pad = lambda v: np.concatenate([v[:n1], np.zeros(n0 - n1)])
        M.NGEOM = 2
        M.NCONV = np.array([n0, n1], dtype="int32")
        M.NAV = np.array([1, 3], dtype="int32")
        M.VCONV = two(M.VCONV[:n0, 0], pad(M.VCONV[:n0, 0] + 1.0))
        M.MEAS = two(M.MEAS[:n0, 0], pad(M.MEAS[:n0, 0])); M.ERRMEAS = two(M.ERRMEAS[:n0, 0], pad(M.ERRMEAS[:n0, 0]))
        z3 = np.zeros((2, 3))
        M.FLAT, M.FLON, M.SOL_ANG, M.AZI_ANG = z3.copy(), z3.copy(), z3.copy(), z3.copy()
        M.EMISS_ANG = np.array([[0.0, 0.0, 0.0], [15.0, 35.0, 55.0]])
        M.WGEOM = np.array([[1.0, 0.0, 0.0], [0.2, 0.5, 0.3]])
        M.NY = n0 + n1
        fm.ansfm_jacobian_route = route
        double.batch_sizes = []
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            YN, KK = fm.jacobian_nemesis(NCores=1, nemesisdisc=True, analytical_gradient=False)
        assert fm.ansfm_last_jacobian["route"] == ("staged" if route == "auto" else "loop")
        res[route] = (YN, KK, list(double.batch_sizes))
    assert np.array_equal(res["auto"][0], res["loop"][0]) and np.array_equal(res["auto"][1], res["loop"][1])
    assert res["auto"][2] == [4, 4, 4, 4] and res["loop"][2] == []        # one batched call per (geometry, averaging point)
    for ix in free:
        assert np.abs(res["auto"][1][:, ix]).max() > 0


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_several_viewing_angles_jacobian_staged_equals_the_reference_loop(c1_cut):
    """jacobian_nemesis(nemesisC=True) on the scattering case: nemesisCfm puts the three viewing geometries into ONE
    multiple-scattering CIRSrad call per state (calc_path_C, hydrostatic re-adjustment on), applies subspecret to the
    unconvolved spectra and convolves all geometries at once.  Staged (one batched scattering call for the six states) against
    the loop (the reference's nemesisCfm per column): the same YN and KK to the last bit."""
    ans, gj, fmod, double = c1_cut
    from oracle import gen_golden_jacobian_ms as gm
    gj.setup_c1(ans, os.getcwd(), seed=4, case=gm.CASE)
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    res = {}
    for route in ("auto", "loop"):
        fm = gj.cut_case(ans, cls=FMGPU, nkeep=gm.NKEEP, free=gm.FREE)
        M = fm.Measurement
        n0, ng = int(M.NCONV[0]), 3
        rep = lambda a: np.repeat(np.asarray(a)[:n0, 0:1], ng, axis=1)
        M.NGEOM = ng
        M.NCONV = np.array([n0] * ng, dtype="int32")
        M.NAV = np.ones(ng, dtype="int32")
        M.VCONV = rep(M.VCONV); M.MEAS = rep(M.MEAS); M.ERRMEAS = rep(M.ERRMEAS)
        z = np.zeros((ng, 1))
        M.FLAT, M.FLON, M.TANHE = z.copy(), z.copy(), z.copy()
        M.AZI_ANG = np.array([[0.0], [40.0], [90.0]])
        M.SOL_ANG = np.full((ng, 1), 30.0)
        M.EMISS_ANG = np.array([[10.0], [30.0], [50.0]])
        M.WGEOM = np.ones((ng, 1))
        M.NY = n0 * ng
        fm.ansfm_jacobian_route = route
        double.scatter_batches = []
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            YN, KK = fm.jacobian_nemesis(NCores=1, nemesisC=True, analytical_gradient=False)
        info = fm.ansfm_last_jacobian
        assert info["route"] == ("staged" if route == "auto" else "loop") and info["nfm"] == 6
        res[route] = (YN, KK, list(double.scatter_batches))
    assert np.array_equal(res["auto"][0], res["loop"][0]) and np.array_equal(res["auto"][1], res["loop"][1])
    assert res["auto"][2] == [6] and res["loop"][2] == []
    n0 = res["auto"][0].size // 3
    assert not np.array_equal(res["auto"][0][:n0], res["auto"][0][n0:2 * n0]) and np.abs(res["auto"][1]).max() > 0


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_primary_transit_jacobian_staged_equals_the_reference_loop(c1_cut):
    """jacobian_nemesis(nemesisPT=True), IFORM = TransitDepth: nemesisPTfm takes the transmission of every tangent path in one
    CIRSrad call and integrates the absorbing area over the tangent heights.  Staged (one batched transmission call for the
    four states, the reference's integration restated) against the loop (the reference's nemesisPTfm per column): same bits.
    With another measurement unit the staged route declines and the reference's own code raises its error."""
    ans, gj, fmod, double = c1_cut
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    free = (20, 45, 70)
    res = {}
    for route in ("auto", "loop"):
        fm = gj.cut_case(ans, cls=FMGPU, nkeep=10, free=free)
        fm.Measurement.IFORM = 2
        fm.ansfm_jacobian_route = route
        double.batch_sizes = []
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            YN, KK = fm.jacobian_nemesis(NCores=1, nemesisPT=True, analytical_gradient=False)
        assert fm.ansfm_last_jacobian["route"] == ("staged" if route == "auto" else "loop")
        res[route] = (YN, KK, list(double.batch_sizes))
    assert np.array_equal(res["auto"][0], res["loop"][0]) and np.array_equal(res["auto"][1], res["loop"][1])
    assert res["auto"][2] == [4] and res["loop"][2] == []
    assert np.all(res["auto"][0] > 0.5) and np.all(res["auto"][0] < 5.0)          # a transit depth in per cent
    for ix in free:
        assert np.abs(res["auto"][1][:, ix]).max() > 0
    fm = gj.cut_case(ans, cls=FMGPU, nkeep=10, free=free)                          # radiance units: the reference refuses
    with pytest.raises(ValueError, match="TransitDepth"):
        fm.jacobian_nemesis(NCores=1, nemesisPT=True, analytical_gradient=False)


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
@pytest.mark.parametrize("route", ["staged", "loop", "profile"])
def test_dropin_jacobian_sharded_over_two_ranks_gloo(tmp_path, route):
    """`ansfm_jacobian_group = (rank, world, group)`: every route of the drop-in jacobian_nemesis gives rank r the reference's
    contiguous chunk of the forward models (:2322-2330; the staged and profile routes put the unperturbed state in front of a
    chunk that does not start with it) and ONE all_gather puts YNtot together -- two gloo ranks on the CPU (the engine
    answered by the oracle double) return the KK of a single process on every rank."""
    import subprocess, textwrap
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = textwrap.dedent(f'''
        import os, sys, tempfile, warnings
        import numpy as np
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        import torch.distributed as dist
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        from oracle.ref_import import import_reference
        from oracle import gen_golden_jacobian as gj
        from oracle import oracle as orc
        from test_dropin_reference import OracleEngineDouble
        import archnemesis_dist_amd.forward_model as fmod
        orc.build()
        ans = import_reference()
        work = tempfile.mkdtemp(prefix="ansfm_w2_%d_" % rank)
        gj.setup_c1(ans, work)
        os.chdir(work)
        double = OracleEngineDouble(orc)
        fmod.get_engine = lambda device=0: double
        FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
        free = (10, 30, 50, 70, 75)
        def run(group):
            fm = gj.cut_case(ans, cls=FMGPU, nkeep=10, free=free)
            fm.ansfm_jacobian_route = {route!r}
            fm.ansfm_jacobian_group = group
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                YN, KK = fm.jacobian_nemesis(NCores=1, analytical_gradient=False)
            assert fm.ansfm_last_jacobian["route"] == {route!r}
            return YN, KK
        one = run(None)
        two = run((rank, world, None))
        assert np.array_equal(one[0], two[0]) and np.array_equal(one[1], two[1]), rank
        assert all(np.abs(two[1][:, ix]).max() > 0 for ix in free)
        dist.destroy_process_group()
        print("rank", rank, "ok")
    ''')
    f = tmp_path / "w.py"
    f.write_text(script)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    port = {"staged": "29641", "loop": "29642", "profile": "29643"}[route]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(f)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


def test_limb_spectra_are_brought_to_the_tangent_heights_as_the_reference_does():
    """nemesisLfm :1322-1344 restated (JacobianGPU._ansfm_limb_to_tangent_heights): weights between the neighbouring tangent
    paths, the nearest base compared after a second division by 1e3, the lower neighbour -1 wrapping to the last path, the
    lower neighbour's spectrum above the top path."""
    from archnemesis_dist_amd.jacobian_dropin import JacobianGPU
    f = JacobianGPU._ansfm_limb_to_tangent_heights
    base = np.array([0.0, 10.0, 25.0, 45.0])                       # km, one tangent path per layer base
    S = np.arange(12, dtype=float).reshape(3, 4) + 1.0
    out = f(S, base, [np.array([12.0]), np.array([60.0]), np.array([10.0])])
    # 12 km: nearest base 10 km; 10 / 1e3 <= 12 -> between paths 1 and 2
    fhl, fhh = (12.0 - 10.0) / 15.0, (25.0 - 12.0) / 15.0
    np.testing.assert_array_equal(out[:, 0], S[:, 1] * (1 - fhl) + S[:, 2] * (1 - fhh))
    # 60 km: nearest base is the top path, its upper neighbour does not exist -> the top path's spectrum
    np.testing.assert_array_equal(out[:, 1], S[:, 3])
    # exactly on a base
    np.testing.assert_array_equal(out[:, 2], S[:, 1] * 1.0 + S[:, 2] * 0.0)
    # a tangent height below every base: nearest is path 0, 0 / 1e3 <= -5 fails -> neighbours (-1, 0): path -1 is the LAST path
    out = f(S, base, [np.array([-5.0])])
    span = base[0] - base[-1]
    fhl, fhh = (-5.0 - base[-1]) / span, (base[0] + 5.0) / span
    np.testing.assert_array_equal(out[:, 0], S[:, -1] * (1 - fhl) + S[:, 0] * (1 - fhh))


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_routes_agree_with_a_gas_profile_carried_as_logarithm(c1_cut):
    """A second model-0 variable, the NH3 mixing-ratio profile: read_apr carries it as ln(vmr) (LX = 1; model_0.py
    from_apr_to_state_vector), subprofretg un-logs it.  State vector of 162 elements, two temperature levels and two ln(vmr)
    levels free: the profile route (exp where LX = 1, the gas column found from VARIDENT) against the reference's host code."""
    ans, gj, fmod, double = c1_cut
    Atm = ans.Files.read_input_files("cirstest")[0]
    j = int(np.nonzero((np.asarray(Atm.ID) == 11) & (np.asarray(Atm.ISO) == 0))[0][0])
    with open("nh3apr.dat", "w") as f:
        f.write("%d 1.5\n" % Atm.NP)
        for p_, v in zip(Atm.P / 101325.0, Atm.VMR[:, j]):
            v = max(float(v), 1.0e-20)                 # the profile is zero above the cloud: ln needs a positive value
            f.write("%.6e %.6e %.6e\n" % (p_, v, 0.5 * v))
    with open("cirstest.apr", "w") as f:
        f.write("header\n2\n0 0 0\ntestapr.dat\n11 0 0\nnh3apr.dat\n")
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    free = (8, 40, 81 + 3, 81 + 9)
    res = {}
    for route in ("profile", "staged"):
        fm = gj.cut_case(ans, cls=FMGPU, nkeep=14, free=free)
        assert fm.Variables.NX == 162 and np.all(fm.Variables.LX[81:] == 1) and np.all(fm.Variables.LX[:81] == 0)
        fm.ansfm_jacobian_route = route
        YN, KK = fm.jacobian_nemesis(NCores=1, analytical_gradient=False)
        assert fm.ansfm_last_jacobian["route"] == route and fm.ansfm_last_jacobian["nfm"] == 5
        res[route] = (YN, KK, np.array(fm.Variables.XN))
    np.testing.assert_allclose(res["profile"][0], res["staged"][0], rtol=1e-13)
    XN = res["staged"][2]
    for ix in free:
        sc = np.abs(res["staged"][1][:, ix]).max()
        floor = 64 * np.finfo(float).eps * np.abs(res["staged"][0]).max() / abs(0.05 * XN[ix])
        assert sc > 0 and np.abs(res["profile"][1][:, ix] - res["staged"][1][:, ix]).max() <= 1e-8 * sc + floor, ix


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_coreretOE_runs_on_the_installed_subclass_and_retrieves_the_same_state(c1_cut):
    """north_star: "drops into the existing OptimalEstimation_0 retrieval loop".  The reference's own `coreretOE`
    (OptimalEstimation_0.py:1173-1584: forward model per iteration, jacobian_nemesis, gain matrix, Marquardt brake) is run twice
    on the C1 case cut to 30 points, one iteration: unmodified, and after `install_gpu_forward_model()` -- coreretOE imports
    `ForwardModel_0` from the package at call time (:1255), which then names the GPU subclass.  Same YN, KK, retrieved state
    and cost; nothing delegated (strict mode); the engine (here its oracle double) did the radiative transfer."""
    import importlib
    import warnings
    ans, gj, fmod, double = c1_cut
    oe = importlib.import_module("archnemesis.OptimalEstimation_0")

    def case():
        Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
        nk = 30
        Meas.NCONV = np.array([nk], dtype="int32")
        Meas.VCONV = Meas.VCONV[:nk]; Meas.MEAS = Meas.MEAS[:nk]; Meas.ERRMEAS = Meas.ERRMEAS[:nk]
        Meas.NY = nk; Meas.Y = Meas.Y[:nk]; Meas.SE = Meas.SE[:nk, :nk]
        return dict(runname="cirstest", Variables=Var, Measurement=Meas, Atmosphere=Atm, Spectroscopy=Spec, Scatter=Scat, Stellar=Stel,
                    Surface=Surf, CIA=CIA, Layer=Lay, Telluric=None, NITER=1, PHILIMIT=0.1, NCores=1)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = oe.coreretOE(**case())
        assert set(vars(double)) == {"orc"}                      # the reference ran on its own
        cls = fmod.install_gpu_forward_model()
        try:
            assert ans.ForwardModel_0 is cls and issubclass(cls, ans._ansfm_reference_ForwardModel_0)
            got = oe.coreretOE(**case())
        finally:
            fmod.uninstall_gpu_forward_model()
    assert ans.ForwardModel_0 is ans._ansfm_reference_ForwardModel_0
    assert hasattr(double, "t") and getattr(double, "lay_calls", 0) == 0   # the table and CIRSrad went to the engine (layering was not installed)
    np.testing.assert_allclose(got.YN, ref.YN, rtol=5e-7)
    sc = np.abs(ref.KK).max(axis=0)
    big = sc > 1e-6 * sc.max()
    assert np.max(np.abs(got.KK - ref.KK)[:, big] / sc[big]) < 1e-6     # analytic gradients (NUM = 0: nemesisfmg's route)
    np.testing.assert_allclose(got.XN, ref.XN, rtol=1e-7)
    np.testing.assert_allclose(got.PHI, ref.PHI, rtol=1e-5)
    assert fmod.summary()["delegated"] == {}


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_routes_agree_with_scaling_models(c1_cut):
    """Models 2 and 3 of the reference (one state-vector element scaling a profile of the reference atmosphere, carried as it
    is / as its logarithm): a temperature profile (model 0), a CH4 scaling (model 2) and a log scaling of PH3 (model 3).  The
    profile route restates them (`profile *= value`, exp where LX = 1) -- same KK as the reference's own subprofretg."""
    ans, gj, fmod, double = c1_cut
    with open("cirstest.apr", "w") as f:
        f.write("header\n3\n0 0 0\ntestapr.dat\n6 1 2\n1.1 0.3\n28 0 3\n0.9 0.4\n")
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    free = (12, 50, 81, 82)
    res = {}
    for route in ("profile", "staged"):
        fm = gj.cut_case(ans, cls=FMGPU, nkeep=14, free=free)
        assert fm.Variables.NX == 83 and [int(m.id) for m in fm.Variables.models] == [0, 2, 3]
        fm.ansfm_jacobian_route = route
        YN, KK = fm.jacobian_nemesis(NCores=1, analytical_gradient=False)
        assert fm.ansfm_last_jacobian["route"] == route and fm.ansfm_last_jacobian["nfm"] == 5
        res[route] = (YN, KK, np.array(fm.Variables.XN))
    np.testing.assert_allclose(res["profile"][0], res["staged"][0], rtol=1e-13)
    XN = res["staged"][2]
    for ix in free:
        sc = np.abs(res["staged"][1][:, ix]).max()
        floor = 64 * np.finfo(float).eps * np.abs(res["staged"][0]).max() / abs(0.05 * XN[ix] if XN[ix] != 0 else 0.05)
        assert sc > 0 and np.abs(res["profile"][1][:, ix] - res["staged"][1][:, ix]).max() <= 1e-8 * sc + floor, ix


@pytest.mark.needs_reference
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "archnemesis")), reason="reference tree not present")
def test_auto_route_falls_back_when_a_route_stumbles(c1_cut, golden_dir, monkeypatch):
    """`auto` never takes a retrieval down because a batched route met something it did not foresee: the failure is recorded,
    announced once as a RuntimeWarning, and the next route (closer to the reference's own code) gives the same KK."""
    ans, gj, fmod, double = c1_cut
    import archnemesis_dist_amd.profile_dropin as pd
    z = _load(golden_dir)
    monkeypatch.setattr(pd.ReferenceProfileBatch, "spectra_batch", lambda self, X: (_ for _ in ()).throw(KeyError("unforeseen")))
    FMGPU = fmod.make_gpu_forward_model(ans.ForwardModel_0)
    fm = gj.cut_case(ans, cls=FMGPU)
    XN0 = np.array(fm.Variables.XN)
    with pytest.warns(RuntimeWarning, match="profile route failed"):
        YN, KK = fm.jacobian_nemesis(NCores=1, analytical_gradient=False)
    info = fm.ansfm_last_jacobian
    assert info["route"] == "staged" and "KeyError" in info["error_profile"]
    assert np.array_equal(fm.Variables.XN, XN0)
    _assert_kk(KK, z, 1e-4, tight=1e-8)
    fm.ansfm_jacobian_route = "profile"          # forced: the error surfaces
    with pytest.raises(KeyError):
        fm.jacobian_nemesis(NCores=1, analytical_gradient=False)
