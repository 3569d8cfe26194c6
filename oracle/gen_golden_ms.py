"""Golden fixtures for the multiple-scattering core: the REFERENCE's scloud11wave_core
(Multiple_Scattering_Core.py:651) on seeded synthetic layer stacks (build container only).

    python oracle/gen_golden_ms.py        # -> tests/golden/ms_*.npz
"""
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def make_case(ans, name, seed, NMU, NWAVE, NG, NLAY, NCONT, NF, NPHI, imie, iray, lowbc, geoms):
    import importlib
    msc = importlib.import_module("archnemesis.Multiple_Scattering_Core")
    sc = sys.modules["archnemesis.Scatter_0"]
    rng = np.random.default_rng(seed)
    x, w = sc.gauss_lobatto(2 * NMU, n_digits=12)           # Scatter_0.calc_GAUSS_LOBATTO :547
    MU = np.array(x[NMU:2 * NMU], dtype="float64"); WTMU = np.array(w[NMU:2 * NMU], dtype="float64")
    VW = 500.0 + 40.0 * np.arange(NWAVE)
    THETA = np.array([0, 1, 2, 3, 4, 5, 7.5, 10, 12.5, 15, 17.5, 20, 25, 30, 35, 40, 45, 50, 55, 60, 70, 80, 90, 100, 110, 120,
                      130, 140, 145, 150, 155, 160, 165, 170, 172.5, 175, 176, 177, 178, 179, 180.0])
    NTH = THETA.size
    PH = np.zeros((NCONT, NWAVE, 2, NTH))
    if imie == 0:                                             # HG: f, g1, g2 in the last three slots (:5129-5133)
        PH[:, :, 0, -1] = rng.uniform(0.6, 0.95, (NCONT, NWAVE))
        PH[:, :, 0, -2] = rng.uniform(0.3, 0.8, (NCONT, NWAVE))
        PH[:, :, 0, -3] = rng.uniform(-0.5, -0.1, (NCONT, NWAVE))
    else:                                                     # tabulated phase function on THETA
        for i in range(NCONT):
            for iw in range(NWAVE):
                g = rng.uniform(0.2, 0.7)
                c = np.cos(np.deg2rad(THETA))
                p = (1 - g * g) / (1 + g * g - 2 * g * c) ** 1.5 / (4 * np.pi)
                PH[i, iw, 0, :] = p
    PH[:, :, 1, :] = np.cos(THETA * np.pi / 180)
    phasarr = np.ascontiguousarray(PH[:, :, :, ::-1])        # :5142
    taus = 10.0 ** rng.uniform(-4, 0.8, size=(NWAVE, NG, NLAY))
    tauray = 10.0 ** rng.uniform(-6, -2, size=(NWAVE, NLAY)) * (1.0 if iray else 0.0)
    tauscat = 10.0 ** rng.uniform(-5, -0.5, size=(NWAVE, NLAY))
    taus[:, :, 1] = 0.0                                       # empty layer
    tauscat[:, 2] = 0.0; tauray[:, 2] = 0.0                   # purely absorbing layer
    taus = np.maximum(taus, (tauscat + tauray)[:, None, :] * 1.05)
    taus[:, :, 1] = 0.0
    omegas = np.zeros_like(taus)
    nz = taus > 0
    omegas[nz] = np.broadcast_to((tauray + tauscat)[:, None, :], taus.shape)[nz] / taus[nz]
    fr = rng.uniform(0.1, 1.0, size=(NWAVE, NCONT, NLAY)); lfrac = fr / fr.sum(axis=1, keepdims=True)
    T = np.linspace(160.0, 110.0, NLAY)
    c1, c2 = 1.1911e-12, 1.439
    bnu = c1 * VW[:, None] ** 3 / (np.exp(c2 * VW[:, None] / T[None, :]) - 1.0)
    radg = np.repeat((c1 * VW ** 3 / (np.exp(c2 * VW / 170.0) - 1.0))[:, None], NMU, 1) * (0.9 if lowbc else 1.0)
    solar = 10.0 ** rng.uniform(-9, -8, NWAVE)
    brdf = np.zeros((NWAVE, NMU, NMU, NF + 1))
    if lowbc:
        brdf[:, :, :, 0] = rng.uniform(0.05, 0.3, (NWAVE, 1, 1)) / np.pi * (1 + 0.1 * rng.uniform(size=(NWAVE, NMU, NMU)))
        if NF >= 1:
            brdf[:, :, :, 1] = 0.1 * brdf[:, :, :, 0]
    sol = np.array([g[0] for g in geoms], float); emi = np.array([g[1] for g in geoms], float)
    azi = np.array([g[2] for g in geoms], float)
    t = time.time()
    rad = msc.scloud11wave_core(phasarr, radg, sol, emi, solar, azi, lowbc, brdf, MU, WTMU, NF, VW, bnu, taus, tauray, omegas,
                                NPHI, iray, imie, lfrac)
    print(name, "rad", rad.shape, "%.1f s" % (time.time() - t), float(rad.min()), float(rad.max()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), phasarr=phasarr, radg=radg, sol_angs=sol, emiss_angs=emi,
                        solar=solar, aphis=azi, lowbc=lowbc, brdf_matrix=brdf, mu1=MU, wt1=WTMU, nf=NF, vwaves=VW, bnu=bnu,
                        taus=taus, tauray=tauray, omegas_s=omegas, nphi=NPHI, iray=iray, imie=imie, lfrac=lfrac, rad=rad)


def main():
    ans = import_reference()
    if "--deep-only" in sys.argv:
        # BASELINE configs[3] depth: 16 streams, NF = 8 (9 Fourier orders), 60 layers -- a few (wavenumber, g) points of
        # the un-jitted reference core (minutes), to hold the MFMA chain kernel's inverse products over deep stacks
        make_case(ans, "ms_nmu16_deep", 7, NMU=16, NWAVE=2, NG=2, NLAY=60, NCONT=1, NF=8, NPHI=101, imie=1, iray=1, lowbc=0,
                  geoms=[(30.0, 20.0, 45.0), (65.0, 50.0, 120.0)])
        make_case(ans, "ms_nmu16_deep_lambert", 8, NMU=16, NWAVE=1, NG=2, NLAY=50, NCONT=2, NF=8, NPHI=101, imie=0, iray=1,
                  lowbc=1, geoms=[(20.0, 35.0, 60.0)])
        return
    # upward-looking geometry (emission angle > 90, :900-903, :927-942): without a surface in the stack (lowbc = 0) and
    # with the internal-field formula `idown` (lowbc > 0)
    gu = [(30.0, 160.0, 45.0), (120.0, 130.0, 0.0)]
    make_case(ans, "ms_nmu5_lookup", 4, NMU=5, NWAVE=3, NG=2, NLAY=6, NCONT=2, NF=2, NPHI=101, imie=0, iray=1, lowbc=0, geoms=gu)
    make_case(ans, "ms_nmu5_lookup_lambert", 5, NMU=5, NWAVE=3, NG=2, NLAY=5, NCONT=1, NF=3, NPHI=101, imie=1, iray=0, lowbc=1,
              geoms=[(10.0, 120.0, 130.0), (75.0, 175.0, 10.0), (40.0, 140.0, 180.0)])
    make_case(ans, "ms_nmu16_lookup_lambert", 6, NMU=16, NWAVE=2, NG=2, NLAY=5, NCONT=1, NF=3, NPHI=101, imie=1, iray=1, lowbc=1,
              geoms=[(30.0, 155.0, 45.0)])
    if "--lookup-only" in sys.argv:
        return
    g2 = [(30.0, 20.0, 45.0), (120.0, 50.0, 0.0)]
    make_case(ans, "ms_nmu5_hg_ray", 1, NMU=5, NWAVE=4, NG=2, NLAY=6, NCONT=2, NF=2, NPHI=101, imie=0, iray=1, lowbc=0, geoms=g2)
    make_case(ans, "ms_nmu5_tab_lambert", 2, NMU=5, NWAVE=3, NG=2, NLAY=5, NCONT=1, NF=3, NPHI=101, imie=1, iray=0, lowbc=1,
              geoms=[(10.0, 60.0, 130.0), (75.0, 5.0, 10.0), (40.0, 40.0, 180.0)])
    make_case(ans, "ms_nmu16_tab_ray", 3, NMU=16, NWAVE=2, NG=2, NLAY=5, NCONT=1, NF=4, NPHI=101, imie=1, iray=1, lowbc=0,
              geoms=[(30.0, 20.0, 45.0)])


if __name__ == "__main__":
    main()
