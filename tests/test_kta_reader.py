"""Native .kta reader: header parsing (no GPU) and file -> HBM upload (GPU) against what the reference's read_ktahead /
read_ktable returned for the same files (fixtures written by the reference's write_ktable, oracle/gen_golden_kta.py)."""
import os
import numpy as np
import pytest

HEAD = ["nwave", "wave", "fwhm", "npress", "ntemp", "ng", "gasID", "isoID", "g_ord", "del_g", "presslevels", "templevels"]


@pytest.mark.parametrize("tag", ["uni", "list"])
def test_header_matches_reference(golden_dir, tag):
    from archnemesis_dist_amd._lib import read_ktable_header
    z = np.load(os.path.join(golden_dir, "kta_read.npz"))
    for gi in range(2):
        h = read_ktable_header(os.path.join(golden_dir, "kta", f"{tag}_gas{gi}.kta"))
        for n, v in zip(HEAD, h):
            ref = z[f"{tag}{gi}_head_{n}"]
            assert np.array_equal(np.asarray(v, dtype=ref.dtype), ref), n
    h = read_ktable_header(os.path.join(golden_dir, "kta", f"{tag}_gas0"))        # '.kta' appended like the reference
    assert h[0] == int(z[f"{tag}0_head_nwave"])
    with pytest.raises(ValueError):
        read_ktable_header(os.path.join(golden_dir, "kta", "missing.kta"))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["uni", "list"])
@pytest.mark.parametrize("rng_name", ["all", "sub"])
def test_upload_from_files_equals_upload_of_reference_arrays(golden_dir, tag, rng_name):
    """Table built on the GPU straight from the files == table uploaded from the K the reference read from them:
    calc_k on both engines is bit-identical (same ln k entries in HBM)."""
    import archnemesis_dist_amd as pkg
    z = np.load(os.path.join(golden_dir, "kta_read.npz"))
    paths = [os.path.join(golden_dir, "kta", f"{tag}_gas{gi}.kta") for gi in range(2)]
    lo, hi = z[f"{tag}0_{rng_name}_range"]
    e1 = pkg.AnsfmEngine(0)
    WAVE, PRESS, TEMP, DELG = e1.upload_ktable_files(paths, lo, hi)
    ref_wave = z[f"{tag}0_{rng_name}_wave"]
    assert np.array_equal(WAVE, ref_wave)
    assert PRESS.dtype == np.float32 and np.array_equal(PRESS, z[f"{tag}0_head_presslevels"])
    K = np.stack([z[f"{tag}{gi}_{rng_name}_k"] for gi in range(2)], axis=-1)      # (NWAVE,NG,NP,NT,NGAS) like read_tables
    e2 = pkg.AnsfmEngine(0)
    e2.upload_ktable(K, z[f"{tag}0_head_presslevels"], z[f"{tag}0_head_templevels"], ref_wave, z[f"{tag}0_head_del_g"])
    assert e1.dims == e2.dims and e1.ktable_info()[1] == e2.ktable_info()[1]
    press = np.array([2e-3, 0.05, 0.7, 20.0]); temp = np.array([90.0, 150.0, 222.0, 310.0])
    k1, d1 = e1.calc_k(press, temp, grad=True)
    k2, d2 = e2.calc_k(press, temp, grad=True)
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)
    assert k1.max() > 0
