"""Multiple-scattering core: CPU oracle vs golden vectors produced by the reference's
scloud11wave_core (oracle/gen_golden_ms.py)."""
import os
import numpy as np
import pytest

MS_CASES = ["ms_nmu5_hg_ray", "ms_nmu5_tab_lambert", "ms_nmu16_tab_ray", "ms_nmu5_lookup", "ms_nmu5_lookup_lambert",
            "ms_nmu16_lookup_lambert"]


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
