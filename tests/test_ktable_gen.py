"""k-table generator (Spectroscopy_0.calc_ktable_chunk): the oracle's binning vs the reference golden (CPU) and the GPU
mirror vs the golden and, on larger bins, vs the oracle."""
import os
import numpy as np
import pytest

from ktable_fakes import make_case


@pytest.fixture(scope="module")
def eng():
    import archnemesis_dist_amd as pkg
    e = pkg.AnsfmEngine(0)
    yield e
    e.close()


class OracleBinner:
    """engine stand-in for the CPU test: kdist_bins answered by the oracle"""
    def __init__(self, orc):
        self.orc = orc

    def kdist_bins(self, *a):
        return self.orc.kdist_bins(*a)


@pytest.mark.parametrize("tag,wf", [("plain", False), ("ils", True)])
def test_chunk_mirror_with_oracle_binning(oracle, golden_dir, tag, wf):
    """host mirror (grid sizing, bin limits, filter offsets) + the oracle's binning == the reference's calc_ktable_chunk"""
    from archnemesis_dist_amd.ktable_gen import calc_ktable_chunk
    z = np.load(os.path.join(golden_dir, "ktable_chunk.npz"))
    k = calc_ktable_chunk(*make_case(wf), engine=OracleBinner(oracle))
    np.testing.assert_allclose(k, z[tag], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,wf", [("plain", False), ("ils", True)])
def test_chunk_on_gpu_matches_reference(eng, golden_dir, tag, wf):
    from archnemesis_dist_amd.ktable_gen import calc_ktable_chunk
    z = np.load(os.path.join(golden_dir, "ktable_chunk.npz"))
    k = calc_ktable_chunk(*make_case(wf), engine=eng)
    np.testing.assert_allclose(k, z[tag], rtol=1e-11)


@pytest.mark.gpu
def test_large_bins_vs_oracle(eng, oracle):
    """bins of 2e4..6e4 points (low-pressure line-by-line grids), overlapping, 20 g-ordinates"""
    rng = np.random.default_rng(12)
    n = 400000
    w = np.linspace(2000.0, 2040.0, n)
    k = 10.0 ** (-24 + 3 * np.sin(w * 37.0) ** 2 + rng.normal(0, 0.3, n))
    cen = np.linspace(2003.0, 2037.0, 18)
    half = rng.uniform(1.0, 3.0, cen.size)
    x, _ = np.polynomial.legendre.leggauss(20)
    g = 0.5 * (x + 1)
    np.testing.assert_allclose(eng.kdist_bins(w, k, cen - half, cen + half, g), oracle.kdist_bins(w, k, cen - half, cen + half, g),
                               rtol=1e-10)
    nf = np.full(cen.size, 5, dtype=np.int32)
    dfil = np.linspace(-1, 1, 5)[:, None] * half[None, :]
    afil = np.tile(np.array([0.1, 0.6, 1.0, 0.6, 0.1])[:, None], (1, cen.size))
    fil = (cen, nf, dfil, afil)
    np.testing.assert_allclose(eng.kdist_bins(w, k, cen - half, cen + half, g, fil),
                               oracle.kdist_bins(w, k, cen - half, cen + half, g, fil), rtol=1e-10)
    with pytest.raises(ValueError):
        eng.kdist_bins(w, k, np.array([1000.0]), np.array([1001.0]), g)        # empty bin: np.interp raises in the reference
