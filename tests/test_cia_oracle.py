"""Collision-induced absorption opacity (ForwardModel_0.calc_tau_cia): oracle vs goldens from the reference."""
import os
import numpy as np
import pytest

CASES = [(t, s) for t in ("eq", "normal", "para") for s in (0, 1)]


def cia_args(z, tag, space):
    key = f"{tag}_{space}"
    inormal, npara = (int(v) for v in z[tag + "_meta"])
    return (space, z[key + "_WAVEC"], z["CIA_WAVEN"], z["CIA_TEMP"], z[tag + "_FRACGRID"], npara, z[tag + "_K_CIA"],
            z[tag + "_IPAIRG1"], z[tag + "_IPAIRG2"], z[tag + "_INORMALT"], inormal, z[tag + "_INORMALD"], z["ID"], z["ISO"],
            z["PP"], z["PRESS"], z["TEMP"], z["FRAC"], z["TOTAM"], z["DELH"]), dict(k_co2=z[key + "_kco2"], k_n2n2=z[key + "_kn2n2"],
                                                                                  k_n2h2=z[key + "_kn2h2"])


@pytest.mark.parametrize("tag,space", CASES)
def test_calc_tau_cia(oracle, golden_dir, tag, space):
    z = np.load(os.path.join(golden_dir, "tau_cia.npz"))
    a, kw = cia_args(z, tag, space)
    tau, dtau = oracle.calc_tau_cia(*a, **kw)
    ref_t, ref_d = z[f"{tag}_{space}_tau"], z[f"{tag}_{space}_dtau"]
    np.testing.assert_allclose(tau, ref_t, rtol=1e-12, atol=0)
    scale = np.max(np.abs(ref_d), axis=(0, 1), keepdims=True) + 1e-300
    assert np.max(np.abs(dtau - ref_d) / scale) < 1e-12
