"""Native .lta reader: header parsing (no GPU) and file -> HBM upload (GPU) against what the reference's read_ltahead /
read_lbltable returned for the same files (fixtures written by the reference's write_lbltable, oracle/gen_golden_lta.py)."""
import os
import numpy as np
import pytest

HEAD = ["nwave", "vmin", "delv", "npress", "ntemp", "gasID", "isoID", "presslevels", "templevels"]


def test_header_matches_reference(golden_dir):
    from archnemesis_dist_amd._lib import read_lbltable_header
    z = np.load(os.path.join(golden_dir, "lta_read.npz"))
    for gi in range(2):
        h = read_lbltable_header(os.path.join(golden_dir, "kta", f"lbl_gas{gi}.lta"))
        for n, v in zip(HEAD, h):
            ref = z[f"g{gi}_head_{n}"]
            assert np.array_equal(np.asarray(v, dtype=ref.dtype), ref), n
        assert np.array_equal(h[9], z[f"g{gi}_all_wave"])                 # np.linspace(vmin, vmax, nwave) of read_lbltable
    assert read_lbltable_header(os.path.join(golden_dir, "kta", "lbl_gas0"))[0] == int(z["g0_head_nwave"])
    with pytest.raises(ValueError):
        read_lbltable_header(os.path.join(golden_dir, "kta", "uni_gas0.kta"))


@pytest.mark.gpu
@pytest.mark.parametrize("rng_name", ["all", "sub"])
def test_upload_from_files_equals_upload_of_reference_arrays(golden_dir, rng_name):
    """LBL table built on the GPU straight from the files == table uploaded from the k the reference read from them:
    calc_klbl / calc_klblg on both engines are bit-identical."""
    import archnemesis_dist_amd as pkg
    z = np.load(os.path.join(golden_dir, "lta_read.npz"))
    paths = [os.path.join(golden_dir, "kta", f"lbl_gas{gi}.lta") for gi in range(2)]
    lo, hi = z[f"g0_{rng_name}_range"]
    e1 = pkg.AnsfmEngine(0)
    WAVE, PRESS, TEMP = e1.upload_lbltable_files(paths, lo, hi)
    ref_wave = z[f"g0_{rng_name}_wave"]
    assert np.array_equal(WAVE, ref_wave)
    assert PRESS.dtype == np.float32 and np.array_equal(PRESS, z["g0_head_presslevels"])
    assert np.array_equal(TEMP, z["g0_head_templevels"])
    K = np.stack([z[f"g{gi}_{rng_name}_k"] for gi in range(2)], axis=-1)          # (NWAVE,NP,NT,NGAS) like read_tables
    e2 = pkg.AnsfmEngine(0)
    e2.upload_lbltable(K, z["g0_head_presslevels"], z["g0_head_templevels"], ref_wave)
    assert e1.dims == e2.dims
    press = np.array([3e-4, 0.02, 0.5, 2.0]); temp = np.array([100.0, 180.0, 250.0, 340.0])
    k1, d1 = e1.calc_klbl(press, temp, grad=True)
    k2, d2 = e2.calc_klbl(press, temp, grad=True)
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)
    assert k1.max() > 0


def test_per_level_temperature_grids_header(golden_dir):
    """NT < 0 in the file: one grid of |NT| temperatures per pressure level (read_ltahead :2480-2483)."""
    from archnemesis_dist_amd._lib import read_lbltable_header
    z = np.load(os.path.join(golden_dir, "lta_read.npz"))
    h = read_lbltable_header(os.path.join(golden_dir, "kta", "lbl_perlevel.lta"))
    for n, v in zip(HEAD, h):
        ref = z[f"pl_head_{n}"]
        assert np.array_equal(np.asarray(v, dtype=ref.dtype), ref), n
    assert h[4] == -2 and h[8].shape == (5, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("rng_name", ["all", "sub"])
def test_per_level_table_streamed_equals_reference_arrays(golden_dir, rng_name):
    """The NT < 0 table streamed from the file == the same table uploaded from what the reference's read_lbltable returned
    (calc_klbl / calc_klblg bit-identical, incl. the per-level temperature brackets)."""
    import archnemesis_dist_amd as pkg
    z = np.load(os.path.join(golden_dir, "lta_read.npz"))
    lo, hi = z[f"pl_{rng_name}_range"]
    e1 = pkg.AnsfmEngine(0)
    WAVE, PRESS, TEMP = e1.upload_lbltable_files([os.path.join(golden_dir, "kta", "lbl_perlevel.lta")], lo, hi)
    assert np.array_equal(WAVE, z[f"pl_{rng_name}_wave"]) and TEMP.shape == (5, 2)
    assert np.array_equal(TEMP, z["pl_head_templevels"].astype(np.float32))
    e2 = pkg.AnsfmEngine(0)
    e2.upload_lbltable(z[f"pl_{rng_name}_k"][..., None], z["pl_head_presslevels"], z["pl_head_templevels"].astype(np.float32),
                       z[f"pl_{rng_name}_wave"])
    press = np.array([3e-4, 0.02, 0.5, 2.0]); temp = np.array([90.0, 150.0, 250.0, 360.0])
    k1, d1 = e1.calc_klbl(press, temp, grad=True)
    k2, d2 = e2.calc_klbl(press, temp, grad=True)
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2) and k1.max() > 0
