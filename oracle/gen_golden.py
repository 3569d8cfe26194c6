"""Generate golden fixtures for the hot path by running the REFERENCE itself (build container only).

    python oracle/gen_golden.py            # writes tests/golden/*.npz

Inputs are seeded synthetic arrays; outputs are what the reference's own functions return
(un-jitted, see oracle/ref_import.py).  Fixtures are data only (inputs + expected outputs).
Reference seams exercised (paths relative to the reference tree):
  Spectroscopy_0.calc_k / calc_kg                     Spectroscopy_0.py:2298 / :2147
  ForwardModel_0.k_overlap / k_overlapg / rank        ForwardModel_0.py:6029 / :5842 / :6117
  ForwardModel_0.planck / planckg                     ForwardModel_0.py:6183 / :6230
  ForwardModel_0.calc_thermal_emission_spectrum(g)    ForwardModel_0.py:6287 / :6380
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def gauss_legendre_01(ng, as_float32=True):
    x, w = np.polynomial.legendre.leggauss(ng)
    g_ord = 0.5 * (x + 1.0)
    del_g = 0.5 * w
    if as_float32:  # .kta headers store these as float32 (Spectroscopy_0.py:2951-3031)
        g_ord = g_ord.astype(np.float32).astype(np.float64)
        del_g = del_g.astype(np.float32).astype(np.float64)
    return g_ord, del_g


def synth_ktable(rng, W, G, NP, NT, S, zero_low_g=True, as_float32=True):
    """k(ν,g,p,T,gas): log-uniform strength, monotone in g, smooth in (p,T); cm^2."""
    PRESS = np.logspace(-7, 1.2, NP)          # atm
    TEMP = np.linspace(70.0, 400.0, NT)
    base = 10.0 ** rng.uniform(-26, -20, size=(W, 1, 1, 1, S))
    gshape = np.sort(10.0 ** rng.uniform(-3, 3, size=(W, G, 1, 1, S)), axis=1)
    pfac = (PRESS[None, None, :, None, None]) ** rng.uniform(0.0, 0.3, size=(W, 1, 1, 1, S))
    tfac = (TEMP[None, None, None, :, None] / 200.0) ** rng.uniform(-1.0, 2.0, size=(W, 1, 1, 1, S))
    K = base * gshape * pfac * tfac
    if zero_low_g:
        # physical tables have exact zeros at the low-g end for weak bins; make the cut depend on
        # (p,T) so that "mixed-sign corner" cells (-> 0) and all-zero cells (-> linear branch) occur
        ncut = rng.integers(0, 4, size=(W, 1, NP, NT, S))
        gidx = np.arange(G)[None, :, None, None, None]
        K = np.where(gidx < ncut, 0.0, K)
    if as_float32:  # values as read from a .kta: float32(k*1e20)/1e20 in double
        K = (K * 1e20).astype(np.float32).astype(np.float64) * 1e-20
    return PRESS, TEMP, K


def make_spec(ans, WAVE, G, PRESS, TEMP, K, g_ord, del_g):
    sp = ans.Spectroscopy_0(ILBL=0)
    W, G_, NP, NT, S = K.shape
    sp.NGAS = S
    sp.ID = np.arange(1, S + 1); sp.ISO = np.zeros(S, dtype=int)
    sp.NWAVE = W; sp.WAVE = WAVE
    sp.NP = NP; sp.NT = NT; sp.PRESS = PRESS; sp.TEMP = TEMP
    sp.NG = G; sp.G_ORD = g_ord; sp.DELG = del_g
    sp.K = K
    return sp


def case_ck(ans, name, seed, W, G, NP, NT, S, L, fp32=True, zero_low_g=True, special=True, f32_dtype=False):
    fm = sys.modules['archnemesis.ForwardModel_0']
    rng = np.random.default_rng(seed)
    g_ord, del_g = gauss_legendre_01(G, fp32)
    PRESS, TEMP, K = synth_ktable(rng, W, G, NP, NT, S, zero_low_g, fp32)
    WAVE = 200.0 + 2.5 * np.arange(W)
    if f32_dtype:   # what read_tables leaves behind for .kta input: float32 PRESS/TEMP/G_ORD/DELG arrays
        PRESS = PRESS.astype(np.float32); TEMP = (TEMP + 0.123).astype(np.float32)
        g_ord = g_ord.astype(np.float32); del_g = del_g.astype(np.float32)
    sp = make_spec(ans, WAVE, G, PRESS, TEMP, K, g_ord, del_g)
    press = np.logspace(np.log10(5.0), -6.5, L)           # atm, bottom -> top
    temp = 110.0 + 250.0 * (np.linspace(0, 1, L) - 0.4) ** 2 + rng.uniform(-3, 3, L)
    if special and L >= 6:
        press[0] = PRESS[-1] * 3.0      # above the table: clamp to last pressure
        press[-1] = PRESS[0] * 0.5      # below the table
        temp[1] = TEMP[-1] + 25.0       # hotter than the table
        temp[-2] = TEMP[0] - 10.0       # colder than the table
        press[2] = PRESS[3]             # exactly on a grid point
        temp[3] = TEMP[2]
    amount = 10.0 ** rng.uniform(17, 24, size=(S, L))    # cm-2
    if special and S >= 3:
        amount[0, 0] = 0.0              # first gas negligible in layer 0
        amount[1, 1] = 0.0              # second gas negligible in layer 1
        amount[2, 2] = 0.0              # later gas negligible
        if S >= 4 and L >= 5:
            amount[:3, 4] = 0.0         # everything before gas 3 negligible
    k = sp.calc_k(L, press, temp)
    kg, dkdT = sp.calc_kg(L, press, temp)
    tau = fm.k_overlap(del_g, k, amount)
    taug, dk = fm.k_overlapg(del_g, kg, dkdT, amount)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), WAVE=WAVE, G_ORD=g_ord, DELG=del_g,
                        TPRESS=PRESS, TTEMP=TEMP, K=K, press=press, temp=temp, amount=amount,
                        k=k, kg=kg, dkdT=dkdT, tau=tau, taug=taug, dk=dk)
    print(name, "k", k.shape, "tau", tau.shape, "max|k-kg|", np.abs(k - kg).max())
    return dict(WAVE=WAVE, DELG=del_g, tau=tau, taug=taug, dk=dk, press=press, temp=temp, S=S)


def case_rank(ans, name, seed, G):
    fm = sys.modules['archnemesis.ForwardModel_0']
    rng = np.random.default_rng(seed)
    _, del_g = gauss_legendre_01(G)
    n = 6
    weights = np.zeros((n, G * G)); conts = np.zeros((n, G * G)); outs = np.zeros((n, G))
    for i in range(n):
        a = np.sort(10.0 ** rng.uniform(-4, 2, G)); b = np.sort(10.0 ** rng.uniform(-4, 2, G))
        if i == 1:
            b[:3] = 0.0
        if i == 2:
            a = rng.permutation(a)      # unsorted input (generic argsort path)
        conts[i] = (a[:, None] + b[None, :]).ravel()
        weights[i] = (del_g[:, None] * del_g[None, :]).ravel()
        outs[i] = fm.rank(weights[i].copy(), conts[i].copy(), del_g)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), DELG=del_g, weight=weights, cont=conts, k_g=outs)
    print(name, outs.shape)


def case_ko_unsorted(ans, name, seed, W, G, L, S):
    """k_overlap (:6029) on k-distributions that are NOT sorted in g (the reference sorts the G*G products itself,
    :6150) including the skip rules, which look at the LAST g-ordinate only (:6075-6102)."""
    fm = sys.modules['archnemesis.ForwardModel_0']
    rng = np.random.default_rng(seed)
    _, del_g = gauss_legendre_01(G)
    k = 10.0 ** rng.uniform(-24, -20, (W, G, L, S))
    amount = 10.0 ** rng.uniform(19, 22, (S, L))
    k[:, -1, 1, 2] = 0.0          # gas 2 skipped in layer 1 although its other ordinates are non-zero
    k[:, -1, 2, 0] = 0.0          # first gas "empty" in layer 2: the second gas is taken as is (unsorted output)
    k[:, -1, 3, 1:] = 0.0         # only gas 0 left in layer 3: output = its unsorted k * amount
    tau = fm.k_overlap(del_g, k, amount)
    dkdT = k * rng.uniform(-0.02, 0.02, k.shape)            # drawn after everything above: tau is unchanged
    taug, dk = fm.k_overlapg(del_g, k, dkdT, amount)        # the same with gradients (k_overlapg :5842, rankg :5959)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), DELG=del_g, k=k, amount=amount, tau=tau, dkdT=dkdT, taug=taug, dk=dk)
    print(name, tau.shape, dk.shape, np.abs(taug - tau).max())


def case_thermal(ans, name, seed, W, G, Li, NPAR, NVMR):
    fm = sys.modules['archnemesis.ForwardModel_0']
    rng = np.random.default_rng(seed)
    out = {}
    for ispace, tag in ((0, "wn"), (1, "wl")):
        WAVE = (50.0 + 37.0 * np.arange(W)) if ispace == 0 else (1.0 + 0.7 * np.arange(W))
        TAU = 10.0 ** rng.uniform(-6, 1.0, size=(W, G, Li))
        TAU[0, 0, :] = 0.0
        dTAU = TAU[:, :, None, :] * 10.0 ** rng.uniform(-3, 0, size=(W, G, NPAR, Li))
        TEMP = np.linspace(120.0, 300.0, Li) + rng.uniform(-5, 5, Li)
        PRESS_nadir = np.logspace(1, 5, Li)                       # increasing along the path
        PRESS_limb = np.concatenate([np.logspace(1, 4, Li // 2), np.logspace(4, 1, Li - Li // 2)])
        EMIS = rng.uniform(0.6, 1.0, W); SOL = 10.0 ** rng.uniform(-9, -7, W); REFL = rng.uniform(0, 0.3, W)
        out.update({f"{tag}_WAVE": WAVE, f"{tag}_TAU": TAU, f"{tag}_dTAU": dTAU, f"{tag}_TEMP": TEMP,
                    f"{tag}_PRESS_nadir": PRESS_nadir, f"{tag}_PRESS_limb": PRESS_limb,
                    f"{tag}_EMIS": EMIS, f"{tag}_SOL": SOL, f"{tag}_REFL": REFL})
        cfgs = {"nadir_nosurf": (PRESS_nadir, -1.0, 180.0, 20.0),
                "nadir_surf": (PRESS_nadir, 265.0, 180.0, 20.0),
                "nadir_solar": (PRESS_nadir, 265.0, 35.0, 20.0),
                "limb": (PRESS_limb, 265.0, 180.0, 90.0)}
        for cn, (PR, TSURF, SOLA, EMIA) in cfgs.items():
            s = fm.calc_thermal_emission_spectrum(ispace, WAVE, TAU, None, TEMP, PR, TSURF, EMIS, SOL, REFL, SOLA, EMIA)
            out[f"{tag}_{cn}_spec"] = s
            out[f"{tag}_{cn}_args"] = np.array([TSURF, SOLA, EMIA])
            sg, dsg, dts = fm.calc_thermal_emission_spectrumg(ispace, WAVE, TAU, dTAU, NVMR, TEMP, PR, TSURF, EMIS)
            out[f"{tag}_{cn}_specg"] = sg; out[f"{tag}_{cn}_dspecg"] = dsg; out[f"{tag}_{cn}_dtsurf"] = dts
        EMI = 10.0 ** rng.uniform(-12, -9, size=(W, Li))
        out[f"{tag}_EMI"] = EMI
        out[f"{tag}_emi_spec"] = fm.calc_thermal_emission_spectrum(ispace, WAVE, TAU, EMI, TEMP, PRESS_nadir, 265.0, EMIS, SOL, REFL, 180.0, 20.0)
        T = np.array([55.0, 130.0, 288.0, 1500.0])
        out[f"{tag}_planck_T"] = T
        out[f"{tag}_planck"] = np.stack([fm.planck(ispace, WAVE, t) for t in T])
        bg = [fm.planckg(ispace, WAVE, t) for t in T]
        out[f"{tag}_planckg_bb"] = np.stack([b[0] for b in bg]); out[f"{tag}_planckg_db"] = np.stack([b[1] for b in bg])
    out["NVMR"] = np.array(NVMR)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "ok")


def case_lbl(ans, name, seed, W, NP, NTa, S, L, temp2d=False, fp32=False):
    """Spectroscopy_0.calc_klbl / calc_klblg on a synthetic LBL table K (W,NP,NT,S)."""
    rng = np.random.default_rng(seed)
    PRESS = np.logspace(-6, 1.0, NP)
    if temp2d:   # NT < 0: one temperature grid per pressure level (:1664-1669)
        TEMP = np.stack([np.linspace(80.0 + 3 * i, 330.0 + 5 * i, NTa) for i in range(NP)])
    else:
        TEMP = np.linspace(90.0, 350.0, NTa)
    K = 10.0 ** rng.uniform(-27, -19, size=(W, NP, NTa, S))
    K[1, :, :, 0] = 0.0                      # an all-zero line ("bad" branch)
    K[2, 1, 2, 1] = 0.0                      # a mixed-sign corner (-> 0)
    if fp32:
        PRESS = PRESS.astype(np.float32); TEMP = TEMP.astype(np.float32)
        K = (K * 1e20).astype(np.float32).astype(np.float64) * 1e-20
    sp = ans.Spectroscopy_0(ILBL=2)
    sp.NGAS = S; sp.ID = np.arange(1, S + 1); sp.ISO = np.zeros(S, dtype=int)
    sp.NWAVE = W; sp.WAVE = 2000.0 + 0.01 * np.arange(W)
    sp.NP = NP; sp.NT = -NTa if temp2d else NTa; sp.PRESS = PRESS; sp.TEMP = TEMP
    sp.NG = 1; sp.G_ORD = np.array([0.]); sp.DELG = np.array([1.0]); sp.K = K
    press = np.logspace(0.8, -5.5, L)
    temp = np.linspace(120.0, 300.0, L) + rng.uniform(-4, 4, L)
    press[0] = float(PRESS[-1]) * 2.0; press[-1] = float(PRESS[0]) * 0.3      # outside the table
    temp[1] = float(np.max(TEMP)) + 20.0; temp[2] = float(np.min(TEMP)) - 5.0  # clamped (klblg: python [-1] wrap)
    press[3] = float(PRESS[2]); temp[3] = float(np.ravel(TEMP)[1])            # on grid values
    k = sp.calc_klbl(L, press, temp)
    kg, dkdT = sp.calc_klblg(L, press, temp)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), K=K, TPRESS=PRESS, TTEMP=TEMP, press=press, temp=temp,
                        k=k, kg=kg, dkdT=dkdT, WAVE=sp.WAVE)
    print(name, k.shape, "max|k-kg|/k", np.nanmax(np.abs(k - kg) / np.maximum(np.abs(k), 1e-300)))


def main():
    os.makedirs(OUT, exist_ok=True)
    ans = import_reference()
    if "--f32-only" in sys.argv:
        case_ck(ans, "ck_g10_s3_f32dtype", 106, W=6, G=10, NP=6, NT=5, S=3, L=8, f32_dtype=True)
        return
    if "--unsorted-only" in sys.argv:
        case_ko_unsorted(ans, "ko_unsorted_g8_s4", 107, W=5, G=8, L=4, S=4)
        return
    if "--lbl-only" in sys.argv:
        case_lbl(ans, "lbl_tab", 301, W=9, NP=6, NTa=5, S=3, L=9)
        case_lbl(ans, "lbl_tab_t2d_f32", 302, W=7, NP=5, NTa=4, S=2, L=8, temp2d=True, fp32=True)
        return
    case_rank(ans, "rank_g10", 11, 10)
    case_rank(ans, "rank_g20", 12, 20)
    case_ck(ans, "ck_g10_s4", 101, W=10, G=10, NP=7, NT=6, S=4, L=9)
    case_ck(ans, "ck_g20_s8", 102, W=5, G=20, NP=6, NT=5, S=8, L=6)
    case_ck(ans, "ck_g16_s2", 103, W=6, G=16, NP=5, NT=4, S=2, L=6, fp32=False)
    case_ck(ans, "ck_g8_s1", 104, W=6, G=8, NP=5, NT=4, S=1, L=6, special=False)
    case_ck(ans, "ck_g10_s3_nozero", 105, W=6, G=10, NP=5, NT=4, S=3, L=7, zero_low_g=False)
    case_thermal(ans, "thermal_g6", 201, W=7, G=6, Li=11, NPAR=5, NVMR=3)
    case_ko_unsorted(ans, "ko_unsorted_g8_s4", 107, W=5, G=8, L=4, S=4)


if __name__ == "__main__":
    main()
