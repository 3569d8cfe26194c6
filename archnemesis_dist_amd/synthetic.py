"""Seeded synthetic atmospheres / k-tables for bench.py and the parity tests.

No model keys, no real data: the reference's own k-tables are not redistributable blobs, so the
benchmark configuration is defined by these generators.  NumPy only (host); bench.py moves the
arrays to HBM before timing.
"""
import numpy as np


def gauss_legendre_01(ng, as_float32=False):
    """g-ordinates / weights on [0,1] (Gauss-Legendre), optionally rounded to float32 like a
    .kta header (Spectroscopy_0.py:2951-3031)."""
    x, w = np.polynomial.legendre.leggauss(ng)
    g_ord = 0.5 * (x + 1.0)
    del_g = 0.5 * w
    if as_float32:
        g_ord = g_ord.astype(np.float32).astype(np.float64)
        del_g = del_g.astype(np.float32).astype(np.float64)
    return g_ord, del_g


def synth_ktable(W, G, NP, NT, S, seed=20260704, zero_low_g=False, chunk=512, out=None):
    """K (W,G,NP,NT,S): log-uniform 1e-30..1e-18 cm^2, non-decreasing in g, smooth in (p,T).
    PRESS logspace(-7,1.3) atm, TEMP linspace(50,500) K."""
    rng = np.random.default_rng(seed)
    PRESS = np.logspace(-7, 1.3, NP)
    TEMP = np.linspace(50.0, 500.0, NT)
    K = np.empty((W, G, NP, NT, S)) if out is None else out
    for w0 in range(0, W, chunk):
        w1 = min(W, w0 + chunk)
        n = w1 - w0
        base = 10.0 ** rng.uniform(-28, -21, size=(n, 1, 1, 1, S))
        gshape = np.sort(10.0 ** rng.uniform(-2, 3, size=(n, G, 1, 1, S)), axis=1)
        pexp = rng.uniform(0.0, 0.3, size=(n, 1, 1, 1, S))
        texp = rng.uniform(-1.0, 2.0, size=(n, 1, 1, 1, S))
        blk = base * gshape * PRESS[None, None, :, None, None] ** pexp * (TEMP[None, None, None, :, None] / 200.0) ** texp
        if zero_low_g:
            ncut = rng.integers(0, 4, size=(n, 1, NP, NT, S))
            blk = np.where(np.arange(G)[None, :, None, None, None] < ncut, 0.0, blk)
        K[w0:w1] = blk
    return PRESS, TEMP, K


def synth_atmosphere(L, S, seed=7, n_models=1, perturb=0.0):
    """Nadir atmosphere, L equal-log-p layers 10 bar -> 1e-6 bar (bottom -> top, like Layer_0).
    Returns dict of per-model arrays: lay_press_pa (n,L), lay_temp (n,L), amount (n,S,L) in cm^-2."""
    rng = np.random.default_rng(seed)
    p_bar = np.logspace(1, -6, L)
    press_pa = p_bar * 1e5
    z = np.linspace(0, 1, L)
    temp = 110.0 + 290.0 * (1.0 - z) ** 3 + 60.0 * z ** 2        # 110..400 K analytic profile
    vmr = 10.0 ** rng.uniform(-9, -1, size=S)
    # column density of a layer ~ dp/(g*mu): molecules cm^-2
    dlnp = np.log(p_bar[0] / p_bar[1])
    totam = press_pa * dlnp / (24.8 * 2.3 * 1.6605e-27) * 1e-4
    amount = vmr[:, None] * totam[None, :]
    lp = np.repeat(press_pa[None], n_models, 0)
    lt = np.repeat(temp[None], n_models, 0)
    am = np.repeat(amount[None], n_models, 0)
    if perturb and n_models > 1:
        # model 0 = base state; model i>0 perturbs one element by +5 % (jacobian_nemesis :2234-2242):
        # first L perturbations: T of layer i-1; next L: amount of gas 0 in layer i-1-L; then cycle
        for i in range(1, n_models):
            j = (i - 1) % (2 * L)
            if j < L:
                lt[i, j] *= (1.0 + perturb)
            else:
                am[i, 0, j - L] *= (1.0 + perturb)
    return dict(lay_press_pa=lp, lay_temp=lt, amount=am)


def nadir_path(L, emiss_ang=0.0):
    """PathX for a nadir ray: layers used top -> bottom (AtmCalc_0.py:355-375), SCALE = 1/cos(emi)."""
    LAYINC = np.arange(L - 1, -1, -1, dtype=np.int32)[:, None]
    NLAYIN = np.array([L], dtype=np.int32)
    SCALE = np.full((L, 1), 1.0 / np.cos(np.deg2rad(emiss_ang)))
    return NLAYIN, LAYINC, SCALE


def synth_continuum(W, L, seed=3, n_models=1):
    """Smooth seeded (W,L) continuum opacity standing in for TAUCIA+TAUDUST+TAURAY."""
    rng = np.random.default_rng(seed)
    wv = np.linspace(0, 1, W)[:, None]
    lv = np.linspace(0, 1, L)[None, :]
    base = 1e-3 * np.exp(-6.0 * lv) * (1.0 + 0.5 * np.sin(9.0 * wv + rng.uniform(0, 6.28)))
    return np.repeat(base[None], n_models, 0)


# Jovian-like composition of the synthetic profile set: H2 and He carry the bulk (and the Rayleigh opacity, IRAY = 4),
# the k-table gases are CH4, NH3 and trace species.  IDs are the radtran gas numbers the reference uses.
PROFILE_GAS_ID = np.array([39, 40, 6, 11, 26, 27, 28, 32, 33, 41, 1, 2, 4, 5, 7, 8, 9, 10, 12, 13, 14, 15], dtype=np.int32)


def synth_profiles(NPRO, NVMR, seed=11, p_bottom_bar=10.0, p_top_bar=1.0e-6):
    """Reference-atmosphere profiles on NPRO levels (bottom -> top, like Atmosphere_0): H (m) from the hydrostatic
    equation of an H2/He atmosphere (g = 24.8 m s-2), P (Pa), T (K) 110..400 K, VMR (NPRO, NVMR) with H2, He first and
    NVMR - 2 absorbers (log-uniform 1e-9 .. 1e-3, mild vertical gradient).  Returns dict(H, P, T, VMR, ID, ISO, RADIUS)."""
    if not 3 <= NVMR <= PROFILE_GAS_ID.size:
        raise ValueError("synth_profiles: 3 <= NVMR <= %d" % PROFILE_GAS_ID.size)
    rng = np.random.default_rng(seed)
    p_bar = np.logspace(np.log10(p_bottom_bar), np.log10(p_top_bar), NPRO)
    z = np.linspace(0.0, 1.0, NPRO)
    T = 110.0 + 290.0 * (1.0 - z) ** 3 + 60.0 * z ** 2
    scale_height = 8.314462618 * T / (2.3e-3 * 24.8)                       # R T / (mu g), m
    dlnp = -np.diff(np.log(p_bar))
    H = np.concatenate([[0.0], np.cumsum(0.5 * (scale_height[1:] + scale_height[:-1]) * dlnp)])
    VMR = np.empty((NPRO, NVMR))
    trace = 10.0 ** rng.uniform(-9, -3, NVMR - 2)
    slope = rng.uniform(-0.5, 0.5, NVMR - 2)
    VMR[:, 2:] = trace[None, :] * (p_bar[:, None] / p_bar[0]) ** (0.1 * slope[None, :])
    rest = 1.0 - VMR[:, 2:].sum(axis=1)
    VMR[:, 0] = 0.864 * rest
    VMR[:, 1] = 0.136 * rest
    return dict(H=H, P=p_bar * 1.0e5, T=T, VMR=VMR, ID=PROFILE_GAS_ID[:NVMR].copy(), ISO=np.zeros(NVMR, dtype=np.int32),
                RADIUS=7.1492e7)
