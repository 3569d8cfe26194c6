"""Multiple-scattering core: CPU oracle vs golden vectors produced by the reference's
scloud11wave_core (oracle/gen_golden_ms.py)."""
import os
import numpy as np
import pytest

MS_CASES = ["ms_nmu5_hg_ray", "ms_nmu5_tab_lambert", "ms_nmu16_tab_ray", "ms_nmu5_lookup", "ms_nmu5_lookup_lambert",
            "ms_nmu16_lookup_lambert", "ms_nmu16_deep", "ms_nmu16_deep_lambert"]


def ms_args(z):
    return (z["phasarr"], z["radg"], z["sol_angs"], z["emiss_angs"], z["solar"], z["aphis"], int(z["lowbc"]),
            z["brdf_matrix"], z["mu1"], z["wt1"], int(z["nf"]), z["vwaves"], z["bnu"], z["taus"], z["tauray"],
            z["omegas_s"], int(z["nphi"]), int(z["iray"]), int(z["imie"]), z["lfrac"])


@pytest.mark.parametrize("name", MS_CASES)
def test_scloud11wave_core_oracle(oracle, golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    rad = oracle.scloud11wave_core(*ms_args(z))
    np.testing.assert_allclose(rad, z["rad"], rtol=1e-9)


def test_mixed_emission_angles_raise(oracle, golden_dir):
    z = dict(np.load(os.path.join(golden_dir, "ms_nmu5_hg_ray.npz")))
    z["emiss_angs"] = np.array([20.0, 120.0])
    with pytest.raises(ValueError):
        oracle.scloud11wave_core(*ms_args(z))


def test_cirsrad_scatter_chain_oracle_vs_reference_golden(oracle, golden_dir):
    """The scattering branch of CIRSrad restated with the oracle's pieces (calc_k + k_overlap, TAUTOT :3989, OMEGA / BB
    :5099-5119, scloud11wave_core, g-quadrature :4504) against what the reference's CIRSrad read and returned on its own
    scattering test inputs (oracle/gen_golden_c4.py)."""
    z = np.load(os.path.join(golden_dir, "c4_cirsrad_scatter.npz"))
    k = oracle.calc_k(z["K"], z["TPRESS"], z["TTEMP"], z["LAY_PRESS"] / 101325.0, z["LAY_TEMP"])
    f_gas = np.ascontiguousarray(z["LAY_AMOUNT"][:, z["IGAS"]].T) * 1.0e-4
    taugas = oracle.k_overlap(z["DELG"], k, f_gas)
    rt = 2e-7 if z["TPRESS"].dtype == np.float32 else 1e-11
    np.testing.assert_allclose(taugas, z["TAUGAS"], rtol=rt)
    tautot = taugas + z["TAUCIA"][:, None, :] + z["TAUDUST"][:, None, :] + z["TAURAY"][:, None, :]
    np.testing.assert_allclose(tautot, z["core_taus"], rtol=rt)
    omega = np.zeros_like(tautot)
    pos = tautot > 0
    omega[pos] = np.broadcast_to((z["TAURAY"] + z["TAUSCAT"])[:, None, :], tautot.shape)[pos] / tautot[pos]
    np.testing.assert_allclose(omega, z["core_omegas"], rtol=rt)
    rad = oracle.scloud11wave_core(z["core_phasarr"], z["core_radg"], z["SOL_ANG"], z["EMISS_ANG"], z["core_solar"], z["AZI_ANG"],
                                   int(z["LOWBC"]), z["core_brdf"], z["MU"], z["WTMU"], int(z["NF"]), z["WAVE"], z["core_bnu"],
                                   tautot, z["TAURAY"], omega, int(z["NPHI"]), int(z["IRAY"]), int(z["IMIE"]), z["core_lfrac"])
    np.testing.assert_allclose(rad, z["core_rad"], rtol=max(1e-9, 10 * rt))
    spec = np.tensordot(np.transpose(rad, (2, 1, 0)), z["DELG"], axes=([1], [0]))
    np.testing.assert_allclose(spec, z["SPECOUT"], rtol=max(1e-9, 10 * rt))
