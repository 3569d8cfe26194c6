"""ctypes/numpy front-end of the CPU oracle (oracle/ansfm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by the product package `archnemesis_dist_amd`.
Function names/arguments mirror the reference seams they restate (file:line in the C source).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_d = np.float64


def build(force=False):
    so = os.path.join(_HERE, "libansfm_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ansfm_oracle.c", "ansfm_oracle_ms.c", "ansfm_oracle_lbl.c", "ansfm_oracle_layer.c", "Makefile")]
    if force or (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libansfm_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype=_d):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _is_f32(a):
    return a is not None and getattr(a, "dtype", None) == np.float32


def set_f32_semantics(grid_f32, delg_f32):
    """float32 dtype semantics of Spectroscopy_0.PRESS/TEMP and DELG (see ansfm_oracle.c)."""
    lib().orc_set_f32_semantics(int(bool(grid_f32)), int(bool(delg_f32)))


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(int(n)))


def calc_k(K, PRESS, TEMP, press, temp, grad=False):
    """Spectroscopy_0.calc_k / calc_kg.  K (W,G,NP,NT,S) -> k (W,G,L,S) [, dkdT]."""
    set_f32_semantics(_is_f32(PRESS) or _is_f32(TEMP), False)
    K = _c(K); PRESS = _c(PRESS); TEMP = _c(TEMP); press = _c(press); temp = _c(temp)
    W, G, NP, NT, S = K.shape
    L = press.shape[0]
    k = np.zeros((W, G, L, S)); dk = np.zeros((W, G, L, S)) if grad else None
    lib().orc_calc_k(W, G, NP, NT, S, _p(K), _p(PRESS), _p(TEMP), L, _p(press), _p(temp), _p(k), _p(dk))
    return (k, dk) if grad else k


def rank(weight, cont, del_g):
    set_f32_semantics(False, _is_f32(del_g))
    weight = _c(weight); cont = _c(cont); del_g = _c(del_g)
    ng = del_g.shape[0]
    assert weight.size == ng * ng == cont.size
    out = np.zeros(ng)
    lib().orc_rank(ng, _p(weight), _p(cont), _p(del_g), _p(out))
    return out


def k_overlap(del_g, k_w_g_l_gas, amount_layer):
    set_f32_semantics(False, _is_f32(del_g))
    del_g = _c(del_g); k = _c(k_w_g_l_gas); am = _c(amount_layer)
    W, G, L, S = k.shape
    assert am.shape == (S, L)
    tau = np.zeros((W, G, L))
    lib().orc_k_overlap(W, G, L, S, _p(del_g), _p(k), None, _p(am), _p(tau), None)
    return tau


def k_overlapg(del_g, k_w_g_l_gas, dkdT_w_g_l_gas, amount_layer):
    set_f32_semantics(False, _is_f32(del_g))
    del_g = _c(del_g); k = _c(k_w_g_l_gas); dkdT = _c(dkdT_w_g_l_gas); am = _c(amount_layer)
    W, G, L, S = k.shape
    tau = np.zeros((W, G, L)); dk = np.zeros((W, G, L, S + 1))
    lib().orc_k_overlap(W, G, L, S, _p(del_g), _p(k), _p(dkdT), _p(am), _p(tau), _p(dk))
    return tau, dk


def planck(ispace, wave, temp):
    wave, temp = np.broadcast_arrays(_c(wave), _c(temp))
    wave = _c(wave); temp = _c(temp)
    bb = np.zeros(wave.shape)
    lib().orc_planck(int(ispace), wave.size, _p(wave), _p(temp), _p(bb), None)
    return bb


def planckg(ispace, wave, temp):
    wave, temp = np.broadcast_arrays(_c(wave), _c(temp))
    wave = _c(wave); temp = _c(temp)
    bb = np.zeros(wave.shape); db = np.zeros(wave.shape)
    lib().orc_planck(int(ispace), wave.size, _p(wave), _p(temp), _p(bb), _p(db))
    return bb, db


def calc_thermal_emission_spectrum(ISPACE, WAVE, TAUTOT_PATH, EMITOT_PATH, TEMP, PRESS, TSURF,
                                   EMISSIVITY, SOLFLUX, REFLECTANCE, SOL_ANG, EMISS_ANG):
    WAVE = _c(WAVE); TAU = _c(TAUTOT_PATH); EMI = _c(EMITOT_PATH); TEMP = _c(TEMP); PRESS = _c(PRESS)
    W, G, Li = TAU.shape
    out = np.zeros((W, G))
    lib().orc_thermal_emission(int(ISPACE), W, G, Li, _p(WAVE), _p(TAU), _p(EMI), _p(TEMP), _p(PRESS),
                               C.c_double(TSURF), _p(_c(EMISSIVITY)), _p(_c(SOLFLUX)),
                               _p(_c(REFLECTANCE)), C.c_double(SOL_ANG), C.c_double(EMISS_ANG), _p(out))
    return out


def calc_thermal_emission_spectrumg(ISPACE, WAVE, TAUTOT_PATH, dTAUTOT_PATH, NVMR, TEMP, PRESS, TSURF,
                                    EMISSIVITY):
    WAVE = _c(WAVE); TAU = _c(TAUTOT_PATH); dTAU = _c(dTAUTOT_PATH); TEMP = _c(TEMP); PRESS = _c(PRESS)
    W, G, NPAR, Li = dTAU.shape
    spec = np.zeros((W, G)); dspec = np.zeros((W, G, NPAR, Li)); dts = np.zeros((W, G))
    lib().orc_thermal_emissiong(int(ISPACE), W, G, NPAR, Li, _p(WAVE), _p(TAU), _p(dTAU), int(NVMR),
                                _p(TEMP), _p(PRESS), C.c_double(TSURF), _p(_c(EMISSIVITY)), _p(spec),
                                _p(dspec), _p(dts))
    return spec, dspec, dts


def calc_singlescatt_plane_spectrum(ISPACE, WAVE, TAUTOT_PATH, TEMP, OMEGA, PHASE, TSURF, EMISSIVITY, BRDF, SOLFLUX, SOL_ANG,
                                    EMISS_ANG):
    """ForwardModel_0.calc_singlescatt_plane_spectrum (:6509-6600), NumPy over (wavenumber, g), the layer loop sequential
    with the reference's order of operations -> SPECOUT (NWAVE, NG)."""
    WAVE = np.asarray(WAVE, float); TAU = np.asarray(TAUTOT_PATH, float); OMEGA = np.asarray(OMEGA, float)
    PHASE = np.asarray(PHASE, float); SOLFLUX = np.asarray(SOLFLUX, float)
    W, G, Li = TAU.shape
    mu = np.cos(EMISS_ANG / 180. * np.pi); mu0 = np.cos(SOL_ANG / 180. * np.pi)
    ssfac = mu0 / (mu0 + mu)
    taud = np.zeros((W, G)); trold = np.ones((W, G)); spec = np.zeros((W, G))
    for j in range(Li):
        taud = taud + TAU[:, :, j]
        tr = np.exp(-taud)
        spec = spec + (trold - tr) * ssfac * OMEGA[:, :, j] * PHASE[:, j][:, None] * SOLFLUX[:, None] / (4. * np.pi)
        spec = spec + (trold - tr) * planck(ISPACE, WAVE, TEMP[j])[:, None]
        trold = tr
    if TSURF <= 0.0:
        radground = planck(ISPACE, WAVE, TEMP[Li - 1])
    else:
        radground = planck(ISPACE, WAVE, TSURF) * np.asarray(EMISSIVITY, float)
    spec = spec + trold * radground[:, None]
    spec = spec + trold * SOLFLUX[:, None] * mu0 * np.asarray(BRDF, float)[:, None]
    return spec


def cirsrad_ck_thermal(ISPACE, K, TPRESS, TTEMP, WAVE, DELG, lay_press_pa, lay_temp, amount,
                       TAUCONT, NLAYIN, LAYINC, SCALE, EMTEMP, TSURF, EMISSIVITY=None, SOLFLUX=None,
                       REFLECTANCE=None, SOL_ANG=None, EMISS_ANG=None, xfac=None, return_taugas=False):
    """CIRSrad (ILBL=K_TABLES, IMOD=THERMAL_EMISSION).  ForwardModel_0.py:4376-4511."""
    set_f32_semantics(_is_f32(TPRESS) or _is_f32(TTEMP), _is_f32(DELG))
    K = _c(K); W, G, NP, NT, S = K.shape
    lay_press_pa = _c(lay_press_pa); L = lay_press_pa.shape[0]
    patm = _c(lay_press_pa / 101325.0)
    amount = _c(amount); assert amount.shape == (S, L)
    LAYINC = _c(LAYINC, np.int32); SCALE = _c(SCALE); EMTEMP = _c(EMTEMP)
    NLAYIN = _c(NLAYIN, np.int32)
    LIMAX, P = LAYINC.shape
    out = np.zeros((W, P))
    tg = np.zeros((W, G, L)) if return_taugas else None
    lib().orc_cirsrad_ck_thermal(
        int(ISPACE), W, G, NP, NT, S, _p(K), _p(_c(TPRESS)), _p(_c(TTEMP)), _p(_c(WAVE)), _p(_c(DELG)),
        L, _p(patm), _p(_c(lay_temp)), _p(lay_press_pa), _p(amount), _p(_c(TAUCONT)), P, LIMAX,
        _p(NLAYIN), _p(LAYINC), _p(SCALE), _p(EMTEMP), C.c_double(TSURF), _p(_c(EMISSIVITY)),
        _p(_c(SOLFLUX)), _p(_c(REFLECTANCE)), _p(_c(SOL_ANG)), _p(_c(EMISS_ANG)), _p(_c(xfac)),
        _p(out), _p(tg))
    return (out, tg) if return_taugas else out


def cirsradg_ck_thermal(ISPACE, K, TPRESS, TTEMP, WAVE, DELG, lay_press_pa, lay_temp, amount, TAUCONT, dTAUCON,
                        NVMR, NPAR, igas_map, NLAYIN, LAYINC, SCALE, EMTEMP, TSURF, EMISSIVITY=None, xfac=None):
    """CIRSrad(return_grad=True) (ILBL=K_TABLES, THERMAL_EMISSION): SPECOUT (W,P), dSPECOUT (W,NPAR,LIMAX,P),
    dTSURF (W,P).  ForwardModel_0.py:4376-4511 with :3853-3872, :3993, :4012, :4233-4247, :4504-4508."""
    set_f32_semantics(_is_f32(TPRESS) or _is_f32(TTEMP), _is_f32(DELG))
    K = _c(K); W, G, NP, NT, S = K.shape
    lay_press_pa = _c(lay_press_pa); L = lay_press_pa.shape[0]
    patm = _c(lay_press_pa / 101325.0)
    amount = _c(amount); assert amount.shape == (S, L)
    LAYINC = _c(LAYINC, np.int32); SCALE = _c(SCALE); EMTEMP = _c(EMTEMP); NLAYIN = _c(NLAYIN, np.int32)
    LIMAX, P = LAYINC.shape
    dTAUCON = _c(dTAUCON)
    if dTAUCON is not None:
        assert dTAUCON.shape == (W, NPAR, L)
    spec = np.zeros((W, P)); dspec = np.zeros((W, NPAR, LIMAX, P)); dts = np.zeros((W, P))
    lib().orc_cirsradg_ck_thermal(
        int(ISPACE), W, G, NP, NT, S, _p(K), _p(_c(TPRESS)), _p(_c(TTEMP)), _p(_c(WAVE)), _p(_c(DELG)), L, _p(patm),
        _p(_c(lay_temp)), _p(lay_press_pa), _p(amount), _p(_c(TAUCONT)), _p(dTAUCON), int(NVMR), int(NPAR),
        _p(_c(igas_map, np.int32)), P, LIMAX, _p(NLAYIN), _p(LAYINC), _p(SCALE), _p(EMTEMP), C.c_double(TSURF),
        _p(_c(EMISSIVITY)), _p(_c(xfac)), _p(spec), _p(dspec), _p(dts))
    return spec, dspec, dts


def scloud11wave_core(phasarr, radg, sol_angs, emiss_angs, solar, aphis, lowbc, brdf_matrix, mu1, wt1, nf, vwaves, bnu,
                      taus, tauray, omegas_s, nphi, iray, imie, lfrac):
    """Multiple_Scattering_Core.scloud11wave_core (:651) -> rad (NPATH, NG, NWAVE)."""
    phasarr = _c(phasarr); radg = _c(radg); taus = _c(taus)
    ncont, nwave, _, nth = phasarr.shape
    nmu = len(mu1); ngeom = len(emiss_angs)
    _, ng, nlay = taus.shape
    rad = np.zeros((ngeom, ng, nwave))
    rc = lib().orc_scloud11wave_core(
        ncont, nwave, nth, _p(phasarr), _p(radg), ngeom, _p(_c(sol_angs)), _p(_c(emiss_angs)), _p(_c(solar)),
        _p(_c(aphis)), int(lowbc), _p(_c(brdf_matrix)), nmu, _p(_c(mu1)), _p(_c(wt1)), int(nf), _p(_c(bnu)), ng, nlay,
        _p(taus), _p(_c(tauray)), _p(_c(omegas_s)), int(nphi), int(iray), int(imie), _p(_c(lfrac)), _p(rad))
    if rc == 1:
        raise ValueError("Emission angles are a mix of values above and below 90 degrees (or NMU too large).")
    return rad


def calc_klbl(K, PRESS, TEMP, press, temp, grad=False):
    """Spectroscopy_0.calc_klbl / calc_klblg.  K (W,NP,|NT|,S); TEMP (|NT|,) or (NP,|NT|) (the NT<0 form)
    -> k (W,L,S) [, dkdT]."""
    set_f32_semantics(_is_f32(PRESS) or _is_f32(TEMP), False)
    K = _c(K); PRESS = _c(PRESS); TEMP = _c(TEMP); press = _c(press); temp = _c(temp)
    W, NP, NTa, S = K.shape
    temp2d = int(TEMP.ndim == 2)
    L = press.shape[0]
    k = np.zeros((W, L, S)); dk = np.zeros((W, L, S)) if grad else None
    lib().orc_calc_klbl(W, NP, NTa, S, _p(K), _p(PRESS), _p(TEMP), temp2d, L, _p(press), _p(temp), _p(k), _p(dk))
    return (k, dk) if grad else k


LINESHAPE_VOIGT, LINESHAPE_LORENTZ, LINESHAPE_DOPPLER = 0, 4, 12      # SpectroscopicLineProfileEnum values


def rew(x, y):
    """Re wofz(x + i y) (the oracle's own Faddeeva restatement)."""
    f = lib().orc_rew; f.restype = C.c_double
    x, y = np.broadcast_arrays(np.asarray(x, float), np.asarray(y, float))
    return np.array([f(C.c_double(a), C.c_double(b)) for a, b in zip(x.ravel(), y.ravel())]).reshape(x.shape)


def voigt_profile(x, sigma, gamma):
    f = lib().orc_voigt_profile; f.restype = C.c_double
    x, sigma, gamma = np.broadcast_arrays(np.asarray(x, float), np.asarray(sigma, float), np.asarray(gamma, float))
    return np.array([f(C.c_double(a), C.c_double(b), C.c_double(c))
                     for a, b, c in zip(x.ravel(), sigma.ravel(), gamma.ravel())]).reshape(x.shape)


def add_line_set_monochromatic_absorption(wn_grid, lineshape_id, t_calc, t_ref, p_calc, p_ref, q_ratio,
                                          isotopic_abundance, isotopic_mass, mol_mix_frac, broadening_params, nu, sw,
                                          e_lower, stimulated_emission_at_t_ref, out, store=None, s_floor=0.0,
                                          wn_calc_window=25.0, wn_approx_window=75.0):
    """LineData_0.add_line_set_monochromatic_absorption (:280); adds into `out` (float64, contiguous)."""
    wn_grid = _c(wn_grid); mmf = _c(mol_mix_frac); bp = _c(broadening_params); nu = _c(nu); sw = _c(sw)
    el = _c(e_lower); sr = _c(stimulated_emission_at_t_ref)
    assert out.dtype == np.float64 and out.flags.c_contiguous
    N = nu.shape[0]; M = mmf.shape[0]
    assert bp.shape == (3 * M, N)
    lib().orc_add_line_set_monochromatic_absorption(
        wn_grid.shape[0], _p(wn_grid), int(lineshape_id), C.c_double(t_calc), C.c_double(t_ref), C.c_double(p_calc),
        C.c_double(p_ref), C.c_double(q_ratio), C.c_double(isotopic_abundance), C.c_double(isotopic_mass), M, _p(mmf), N,
        _p(bp), _p(nu), _p(sw), _p(el), _p(sr), _p(out), _p(store), C.c_double(s_floor), C.c_double(wn_calc_window),
        C.c_double(wn_approx_window))
    return out


def layer_average(RADIUS, H, P, T, ID, VMR, DUST, PARAH2, BASEH, BASEP, LAYANG=0.0, LAYINT=0, LAYHT=0.0, NINT=101,
                  DUST_UNITS=None, XMOLWT=None):
    """Layer_0.layer_average (:755) -> HEIGHT,PRESS,TEMP,TOTAM,AMOUNT,PP,CONT,FRAC,DELH,BASET,LAYSF."""
    H = _c(H); P = _c(P); T = _c(T); BASEH = _c(BASEH)
    VMR = _c(VMR).reshape(H.size, -1); NV = VMR.shape[1]
    ND = 0 if DUST is None else _c(DUST).reshape(H.size, -1).shape[1]
    D = None if DUST is None else _c(DUST).reshape(H.size, -1)
    NL = BASEH.size
    o = [np.zeros(NL) for _ in range(4)]
    AM = np.zeros((NL, NV)); PPo = np.zeros((NL, NV)); CO = np.zeros((NL, ND)); FR = np.zeros(NL)
    DELH = np.zeros(NL); BASET = np.zeros(NL); LAYSF = np.zeros(NL)
    rc = lib().orc_layer_average(C.c_double(RADIUS), H.size, _p(H), _p(P), _p(T), NV, _p(VMR), ND, _p(D), _p(_c(PARAH2)), NL,
                                 _p(BASEH), C.c_double(LAYANG), int(LAYINT), C.c_double(LAYHT), int(NINT),
                                 _p(_c(DUST_UNITS, np.int32)), _p(_c(XMOLWT)), _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(AM),
                                 _p(PPo), _p(CO), _p(FR), _p(DELH), _p(BASET), _p(LAYSF))
    if rc:
        raise ValueError("layer_average: NINT < 2")
    return o[0], o[1], o[2], o[3], AM, PPo, CO, FR, DELH, BASET, LAYSF


def layer_averageg(RADIUS, H, P, T, ID, VMR, DUST, PARAH2, BASEH, BASEP, LAYANG=0.0, LAYINT=0, LAYHT=0.0, NINT=101,
                   DUST_UNITS=None, XMOLWT=None):
    """Layer_0.layer_averageg (:1032) -> HEIGHT,PRESS,TEMP,TOTAM,AMOUNT,PP,CONT,FRAC,DELH,BASET,LAYSF,DTE,DAM,DCO,DPH."""
    H = _c(H); P = _c(P); T = _c(T); BASEH = _c(BASEH)
    VMR = _c(VMR).reshape(H.size, -1); NV = VMR.shape[1]
    ND = 0 if DUST is None else _c(DUST).reshape(H.size, -1).shape[1]
    D = None if DUST is None else _c(DUST).reshape(H.size, -1)
    NL = BASEH.size
    o = [np.zeros(NL) for _ in range(4)]
    AM = np.zeros((NL, NV)); PPo = np.zeros((NL, NV)); CO = np.zeros((NL, ND)); FR = np.zeros(NL)
    DELH = np.zeros(NL); BASET = np.zeros(NL); LAYSF = np.zeros(NL)
    M = [np.zeros((NL, H.size)) for _ in range(4)]
    rc = lib().orc_layer_averageg(C.c_double(RADIUS), H.size, _p(H), _p(P), _p(T), NV, _p(VMR), ND, _p(D), _p(_c(PARAH2)), NL,
                                  _p(BASEH), C.c_double(LAYANG), int(LAYINT), C.c_double(LAYHT), int(NINT),
                                  _p(_c(DUST_UNITS, np.int32)), _p(_c(XMOLWT)), _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(AM),
                                  _p(PPo), _p(CO), _p(FR), _p(DELH), _p(BASET), _p(LAYSF), _p(M[0]), _p(M[1]), _p(M[2]), _p(M[3]))
    if rc == 5:
        raise ValueError("NINT must be odd for Simpson's rule.")
    if rc:
        raise ValueError("setting an array element with a sequence.")   # the reference's MID_PATH + DUST_UNITS=-1 failure
    return (o[0], o[1], o[2], o[3], AM, PPo, CO, FR, DELH, BASET, LAYSF) + tuple(M)


# ---- gradient maps (ForwardModel_0.map2pro :5319-5383, map2xvec :5387-5424) ----------------------------------
def map2pro(dSPECIN, NWAVE, NVMR, NDUST, NPRO, NPATH, NLAYIN, LAYINC, DTE, DAM, DCO, INCPAR=(-1,)):
    """NumPy restatement: per (path, listed parameter) a (NWAVE x NLAYIN)·(NLAYIN x NPRO) product with the rows
    DAM/DTE/DCO[LAYINC[:,path]] (:5362-5377).  The para-H2 slot is assigned the previous iteration's product (the
    reference sets it to zero and then overwrites it with the stale dSPECOUT1, :5375-5377)."""
    dSPECIN = np.asarray(dSPECIN, float)
    LAYINC = np.asarray(LAYINC).reshape(dSPECIN.shape[2], -1)
    out = np.zeros((NWAVE, NVMR + 2 + NDUST, NPRO, NPATH))
    inc = list(range(NVMR + 2 + NDUST)) if INCPAR[0] == -1 else list(INCPAR)
    last = None
    for ipath in range(NPATH):
        for par in inc:
            if par <= NVMR - 1:
                M = DAM
            elif par <= NVMR:
                M = DTE
            elif NVMR < par <= NVMR + NDUST:
                M = DCO
            else:
                M = None
            if M is not None:
                last = dSPECIN[:, par, :, ipath] @ np.asarray(M, float)[LAYINC[:, ipath], :]
            if last is None:
                raise UnboundLocalError("local variable 'dSPECOUT1' referenced before assignment")
            out[:, par, :, ipath] = last
    return out


def map2xvec(dSPECIN, NWAVE, NVMR, NDUST, NPRO, NPATH, NX, xmap):
    """dSPECOUT[w,p,x] = sum_{par,pro} dSPECIN[w,par,pro,p] * xmap[x,par,pro]   (:5421)."""
    a = np.asarray(dSPECIN, float)
    x = np.asarray(xmap, float)
    W, NPAR, NP_, P = a.shape
    return np.einsum("wkp,xk->wpx", a.reshape(W, NPAR * NP_, P), x.reshape(NX, NPAR * NP_), optimize=True)


# ---- ILS convolution (Measurement_0.lblconv :3335, lblconvg :3799, lblconv_fil :3549, lblconvg_fil :3992) ---------
def _ils_window(ishape, vcen, fwhm, grad, ngeom=False):
    """Window [v1, v2] and the Gaussian sigma of one convolution point.  The reference kernels differ for the
    Hamming shape: lblconv sets v1 = v2 = vcen - 1.1*fwhm (:3391-3393), lblconvg vcen -+ fwhm (:3866-3868), both
    *_ngeom variants v1 = v2 = vcen - fwhm (:3501-3503, :3753-3755)."""
    sig = 0.0
    if ishape == 0:
        v1 = vcen - 0.5 * fwhm; v2 = v1 + fwhm
    elif ishape == 1:
        v1 = vcen - fwhm; v2 = vcen + fwhm
    elif ishape == 2:
        sig = 0.5 * fwhm / np.sqrt(np.log(2.0)); v1 = vcen - 3. * sig; v2 = vcen + 3. * sig
    elif ishape == 3:
        if ngeom:
            v1 = vcen - fwhm; v2 = vcen - fwhm
        elif grad:
            v1 = vcen - fwhm; v2 = vcen + fwhm
        else:
            v1 = vcen - 1.1 * fwhm; v2 = vcen - 1.1 * fwhm
    else:
        v1 = vcen - 3. * fwhm; v2 = vcen + 3. * fwhm
    return v1, v2, sig


def _ils_weight(ishape, dv_plus_vcen, vcen, fwhm, sig):
    v = dv_plus_vcen
    if ishape == 0:
        return np.ones_like(v)
    if ishape == 1:
        return 1.0 - np.abs(v - vcen) / fwhm
    if ishape == 2:
        return np.exp(-((v - vcen) / sig) ** 2.0)
    if ishape == 3:
        a = 0.907 / fwhm
        k = v - vcen
        with np.errstate(all="ignore"):
            num = a * (1.08 - (0.64 * a ** 2 * k ** 2)) * np.sin(2 * np.pi * a * k)
            den = (1 - 4 * a ** 2 * k ** 2) * (2 * np.pi * a * k)
            f = num / den
        return np.where(k != 0.0, f, a * 1.08)
    return np.zeros_like(v)             # Hanning: no weight is ever assigned (`else: pass`) -> 0/0


def _ils_sum(f1, cols):
    """sum_i f1_i * col_i over the window in index order, only where f1 > 0 (:3433-3435)."""
    keep = f1 > 0.0
    out = np.zeros(cols.shape[1]); nor = 0.0
    for fi, row in zip(f1[keep], cols[keep]):
        out = out + fi * row
        nor = nor + fi
    with np.errstate(all="ignore"):
        return out / nor


def _ils_cols(y, dydx, ngeom):
    """columns [gradients..., spectra...] of one call and how to split the result again"""
    y = np.asarray(y, float)
    ycols = y if ngeom else y[:, None]
    ng = ycols.shape[1]
    if dydx is None:
        return ycols, ng, 0
    d = np.asarray(dydx, float)
    nx = d.shape[-1]
    return np.column_stack([d.reshape(d.shape[0], ng * nx), ycols]), ng, nx


def _ils_split(out, ng, nx, grad, ngeom):
    nconv = out.shape[0]
    yo = out[:, ng * nx:]; go = out[:, :ng * nx].reshape(nconv, ng, nx)
    if not ngeom:
        yo = yo[:, 0]; go = go[:, 0, :]
    return (yo, go) if grad else yo


def lblconv(nwave, vwave, y, nconv, vconv, ishape, fwhm, dydx=None, ngeom=False):
    """Measurement_0.lblconv (dydx None) / lblconvg: yout (nconv) [, gradout (nconv, nx)]; ngeom=True: lblconv_ngeom
    (:3444) / lblconvg_ngeom (:3685) with y (nwave, ngeom), dydx (nwave, ngeom, nx)."""
    vwave = np.asarray(vwave, float)
    grad = dydx is not None
    cols, ng, nx = _ils_cols(y, dydx, ngeom)
    out = np.zeros((nconv, cols.shape[1]))
    for j in range(nconv):
        v1, v2, sig = _ils_window(int(ishape), vconv[j], fwhm, grad, ngeom)
        idx = np.where((vwave >= v1) & (vwave <= v2))[0]
        out[j] = _ils_sum(_ils_weight(int(ishape), vwave[idx], vconv[j], fwhm, sig), cols[idx])
    return _ils_split(out, ng, nx, grad, ngeom)


def lblconv_fil(nwave, vwave, y, nconv, vconv, nfil, vfil, afil, dydx=None, ngeom=False, bracket=False):
    """Measurement_0.lblconv_fil / lblconvg_fil (ngeom=True: lblconv_fil_ngeom :3614 / lblconvg_fil_ngeom :3912):
    tabulated filter per convolution point, np.interp weights.  bracket=True: the FWHM < 0 branch of the k-table
    methods Measurement_0.conv / convg (:2425-2461, :2655-2691), whose window runs from the last point below the
    filter to the first above it."""
    vwave = np.asarray(vwave, float)
    grad = dydx is not None
    cols, ng, nx = _ils_cols(y, dydx, ngeom)
    out = np.zeros((nconv, cols.shape[1]))
    for j in range(nconv):
        n = int(nfil[j])
        xp = np.asarray(vfil[:n, j], float); yp = np.asarray(afil[:n, j], float)
        if bracket:
            lo = np.where(vwave < xp[0])[0]; hi = np.where(vwave > xp[-1])[0]
            idx = np.arange(lo[-1], hi[0] + 1)            # IndexError when the filter is not inside the grid, like :2437
        else:
            idx = np.where((vwave >= xp[0]) & (vwave <= xp[-1]))[0]
        out[j] = _ils_sum(np.interp(vwave[idx], xp, yp), cols[idx])
    return _ils_split(out, ng, nx, grad, ngeom)


def integrate_filter(nwave, vwave, y, nconv, vconv, nfil, vfil, afil, dydx=None):
    """Measurement_0.integrate_filter (:4079) / integrate_filterg (:4188) and, for y (nwave, ngeom), their *_ngeom
    variants (:4131, :4251): np.trapz of filter x spectrum inside each filter."""
    vwave = np.asarray(vwave, float)
    ngeom = np.ndim(y) == 2
    grad = dydx is not None
    cols, ng, nx = _ils_cols(y, dydx, ngeom)
    out = np.zeros((nconv, cols.shape[1]))
    for j in range(nconv):
        n = int(nfil[j])
        idx = np.where((vwave >= vfil[0, j]) & (vwave <= vfil[n - 1, j]))[0]
        f = np.interp(vwave[idx], vfil[:n, j], afil[:n, j])
        trapz = getattr(np, "trapezoid", None) or np.trapz
        out[j] = trapz(cols[idx] * f[:, None], vwave[idx], axis=0)
    return _ils_split(out, ng, nx, grad, ngeom)


# ---- collision-induced absorption (ForwardModel_0.calc_tau_cia :4516-4760) -----------------------------------------
def calc_tau_cia(ISPACE, WAVEC, CIA_WAVEN, CIA_TEMP, CIA_FRAC, NPARA, K_CIA, IPAIRG1, IPAIRG2, INORMALT, INORMAL, INORMALD,
                 ID, ISO, PP, PRESS, TEMP, FRAC, TOTAM, DELH, k_co2=None, k_n2n2=None, k_n2h2=None):
    """NumPy restatement with the reference's objects flattened into arrays.  k_co2 / k_n2n2 / k_n2h2 are the reference's
    co2cia / n2n2cia / n2h2cia(WAVEN) (wavenumber-only parametrisations with embedded data tables, taken as inputs).
    Returns TAUCIA (NWAVE, NLAY), dTAUCIA (NWAVE, NLAY, NVMR+2).  Quirks kept: the upper para-fraction clamp overwrites
    temp1 (:4623), the temperature gradient goes to slot NVMR-2 (:4695), dktdT = (kthi - ktlo) * dfhldT (:4667) and the
    wavelength-space result is gathered with isort (:4741-4743)."""
    WAVEC = np.asarray(WAVEC, float); ID = np.asarray(ID); ISO = np.asarray(ISO)
    NVMR = ID.size; NLAY = np.asarray(TEMP).size; NPAIR = len(IPAIRG1)
    CIA_TEMP = np.asarray(CIA_TEMP, float); CIA_FRAC = np.asarray(CIA_FRAC, float); NT = CIA_TEMP.size
    q = (np.asarray(PP, float).T / np.asarray(PRESS, float)).T
    ico2 = ih2 = in2 = -1
    for i in range(NVMR):
        if ID[i] == 39 and ISO[i] in (0, 1): ih2 = i
        if ID[i] == 22: in2 = i
        if ID[i] == 2 and ISO[i] in (0, 1): ico2 = i
    XFAC = (np.asarray(TOTAM, float) * 1.0e-4) ** 2. / (np.asarray(DELH, float) * 1.0e2)
    if ISPACE == 0:
        WAVEN = WAVEC; isort = None
    else:
        WAVEN = 1.e4 / WAVEC; isort = np.argsort(WAVEN); WAVEN = WAVEN[isort]
    NW = WAVEC.size
    tau = np.zeros((NW, NLAY)); dtau = np.zeros((NW, NLAY, NVMR + 2))
    covers = (CIA_WAVEN.min() <= WAVEN.min()) and (CIA_WAVEN.max() >= WAVEN.max())
    # linear interp1d over CIA_WAVEN: idx = clip(searchsorted(x, xn, 'left'), 1, n-1); slope*(xn - x_lo) + y_lo
    iw = np.clip(np.searchsorted(CIA_WAVEN, WAVEN, side="left"), 1, CIA_WAVEN.size - 1)
    xlo = CIA_WAVEN[iw - 1]; dx = CIA_WAVEN[iw] - xlo
    for l in range(NLAY):
        temp1 = float(TEMP[l])
        it = int(np.argmin(np.abs(CIA_TEMP - temp1)))
        if CIA_TEMP[it] >= temp1:
            ithi = it
            if it == 0: temp1 = CIA_TEMP[0]; itl = 0; ithi = 1
            else: itl = it - 1
        else:
            itl = it
            if it == NT - 1: temp1 = CIA_TEMP[it]; ithi = NT - 1; itl = NT - 2
            else: ithi = it + 1
        frac1 = float(FRAC[l])
        ip = int(np.argmin(np.abs(CIA_FRAC - frac1)))
        if CIA_FRAC[ip] >= frac1:
            iphi = ip
            if ip == 0: frac1 = CIA_FRAC[0]; ipl = 0; iphi = 1
            else: ipl = ip - 1
        else:
            ipl = ip
            if ip == NPARA - 1: temp1 = CIA_FRAC[ip]; iphi = NPARA - 1; ipl = NPARA - 2       # sic: temp1
            else: iphi = ip + 1
        if NPARA == 0: ipl = iphi = 0
        fhl_t = (temp1 - CIA_TEMP[itl]) / (CIA_TEMP[ithi] - CIA_TEMP[itl])
        fhh_t = (CIA_TEMP[ithi] - temp1) / (CIA_TEMP[ithi] - CIA_TEMP[itl])
        dfhldT = 1.0 / (CIA_TEMP[ithi] - CIA_TEMP[itl])
        if CIA_FRAC.size > 1:
            fhl_f = (frac1 - CIA_FRAC[ipl]) / (CIA_FRAC[iphi] - CIA_FRAC[ipl])
            fhh_f = (CIA_FRAC[iphi] - frac1) / (CIA_FRAC[iphi] - CIA_FRAC[ipl])
        else:
            fhl_f = fhh_f = 0.5
        ktlo = K_CIA[:, ipl, itl, :] * fhh_t + K_CIA[:, ipl, ithi, :] * fhl_t
        kthi = K_CIA[:, iphi, itl, :] * fhh_t + K_CIA[:, iphi, ithi, :] * fhl_t
        kt = ktlo * fhh_f + kthi * fhl_f
        dktdT = (kthi - ktlo) * dfhldT
        sum1 = np.zeros(NW)
        if covers:
            for ipair in range(NPAIR):
                g1 = np.where(ID == IPAIRG1[ipair])[0]; g2 = np.where(ID == IPAIRG2[ipair])[0]
                if len(g1) > 1: g1 = np.where((ID == IPAIRG1[ipair]) & (ISO == 1))[0]
                if len(g2) > 1: g2 = np.where((ID == IPAIRG2[ipair]) & (ISO == 1))[0]
                if len(g1) == 1 and len(g2) == 1:
                    g1 = int(g1[0]); g2 = int(g2[0])
                    if INORMALD[ipair] and INORMALT[ipair] != INORMAL:
                        continue
                    y = kt[ipair]; k_cia = ((y[iw] - y[iw - 1]) / dx) * (WAVEN - xlo) + y[iw - 1]
                    y = dktdT[ipair]; dk = ((y[iw] - y[iw - 1]) / dx) * (WAVEN - xlo) + y[iw - 1]
                    sum1 = sum1 + k_cia * q[l, g1] * q[l, g2]
                    dtau[:, l, g1] = dtau[:, l, g1] + q[l, g2] * k_cia
                    dtau[:, l, g2] = dtau[:, l, g2] + q[l, g1] * k_cia
                    dtau[:, l, NVMR - 2] = dtau[:, l, NVMR - 2] + dk * q[l, g1] * q[l, g2]
        if ico2 != -1:
            sum1 = sum1 + k_co2 * q[l, ico2] * q[l, ico2]
            dtau[:, l, ico2] = dtau[:, l, ico2] + 2. * q[l, ico2] * k_co2
        if in2 != -1:
            sum1 = sum1 + k_n2n2 * q[l, in2] * q[l, in2]
            dtau[:, l, in2] = dtau[:, l, in2] + 2. * q[l, in2] * k_n2n2
        if in2 != -1 and ih2 != -1:
            sum1 = sum1 + k_n2h2 * q[l, in2] * q[l, ih2]
            dtau[:, l, ih2] = dtau[:, l, ih2] + q[l, in2] * k_n2h2
            dtau[:, l, in2] = dtau[:, l, in2] + q[l, ih2] * k_n2h2
        tau[:, l] = sum1 * XFAC[l]
        dtau[:, l, :] = dtau[:, l, :] * XFAC[l]
    if isort is not None:
        tau = tau[isort, :]; dtau = dtau[isort, :, :]
    return tau, dtau


# ---- Rayleigh scattering (ForwardModel_0.calc_tau_rayleighj :5525, rayleighv :5598, rayleighv2 :5647, rayleighls :5712)
def calc_tau_rayleigh(mode, ISPACE, WAVEC, TOTAM, ID=None, ISO=None, VMR=None):
    """mode = IRAY (1 gas giant, 2 CO2 (v2), 4 Jovian air) or "v" for the older CO2 formula.
    -> TAURAY (NWAVE, NLAY), dTAURAY (NWAVE, NLAY)."""
    WAVEC = np.asarray(WAVEC, float); TOTAM = np.asarray(TOTAM, float)
    if mode == 1:
        AH2 = 13.58E-5; BH2 = 7.52E-3; AHe = 3.48E-5; BHe = 2.30E-3; fH2 = 0.864
        kb = 1.37971e-23; P0 = 1.01325e5; T0 = 273.15
        LAMBDA = 1. / WAVEC * 1.0e-2 if ISPACE == 0 else WAVEC * 1.0e-6
        x = 1.0 / (LAMBDA * 1.0e6)
        nAir = fH2 * (AH2 * (1.0 + BH2 * x * x)) + (1 - fH2) * (AHe * (1.0 + BHe * x * x))
        temp = 32 * (np.pi ** 3.) * nAir ** 2.
        x = (P0 / (kb * T0)) * LAMBDA * LAMBDA
        k = temp * 1.0 / (3. * (x ** 2))
        k = np.repeat(k[:, None], TOTAM.size, axis=1)
    elif mode == "v":
        LAMBDA = 1. / WAVEC * 1.0e4 if ISPACE == 0 else WAVEC
        k = np.repeat((8.8e-28 / LAMBDA ** 4. * 1.0e-4)[:, None], TOTAM.size, axis=1)
    elif mode == 2:
        LAMBDA = 1. / WAVEC * 1.0e4 if ISPACE == 0 else WAVEC
        dens = 2.5475605e+19
        lam = LAMBDA * 1.0e-4
        f_king = 1.14 + (25.3e-12) / (lam * lam)
        nu2 = 1. / lam / lam
        term1 = (5799.3 / (16.618e9 - nu2) + 120.05 / (7.9609e9 - nu2) + 5.3334 / (5.6306e9 - nu2) + 4.3244 / (4.6020e9 - nu2)
                 + 1.218e-5 / (5.84745e6 - nu2))
        n = 1.0 + 1.1427e3 * term1
        factor1 = ((n * n - 1) / (n * n + 2.0)) ** 2.
        k = (24. * np.pi ** 3. / lam ** 4. / dens ** 2.) * factor1 * f_king * 1.0e-4
        k = np.repeat(k[:, None], TOTAM.size, axis=1)
    elif mode == 4:
        ID = np.asarray(ID); ISO = np.asarray(ISO); VMR = np.asarray(VMR, float)
        NLAY = VMR.shape[0]
        f = {g: np.zeros(NLAY) for g in (39, 40, 6, 11)}
        for j in range(ID.size):
            if int(ID[j]) in f and ISO[j] in (0, 1):
                f[int(ID[j])] = VMR[:, j].copy()
        fh2, fhe, fch4, fnh3 = f[39], f[40], f[6], f[11]
        fheh2 = np.zeros(NLAY); fch4h2 = np.zeros(NLAY)
        ok = fh2 > 0.0
        fheh2[ok] = fhe[ok] / fh2[ok]; fch4h2[ok] = fch4[ok] / fh2[ok]
        comp = np.zeros((NLAY, 4))
        comp[:, 0] = (1.0 - fnh3) / (1.0 + fheh2 + fch4h2)
        comp[:, 1] = fheh2 * comp[:, 0]; comp[:, 2] = fch4h2 * comp[:, 0]; comp[:, 3] = fnh3
        losch = 2.687e19 * 1.0E+12
        wl = 1. / WAVEC * 1.0e4 if ISPACE == 0 else WAVEC
        A = (13.58e-5, 3.48e-5, 37.0e-5, 37.0e-5); B = (7.52e-3, 2.3e-3, 12.0e-3, 12.0e-3); Dp = (0.0221, 0.025, .0922, .0922)
        xc1 = np.zeros((NLAY, WAVEC.size)); sumwt = np.zeros(NLAY)
        for j in range(4):
            nr = 1.0 + A[j] * (1.0 + B[j] / wl ** 2.)
            xc1 += np.outer(comp[:, j], (nr ** 2.0 - 1.0) ** 2.0) * (6.0 + 3.0 * Dp[j]) / (6.0 - 7.0 * Dp[j])
            sumwt += comp[:, j]
        fact = 8.0 * (np.pi ** 3.0) / (3.0 * (wl ** 4.0) * (losch ** 2.0))
        k = np.transpose(fact * xc1 * 1.0E-8) / sumwt * 1.0e-4
    else:
        raise ValueError("mode")
    return k * TOTAM, k.copy()


# ---- aerosol opacity (ForwardModel_0.calc_tau_dust :4790-4867) -----------------------------------------------------
def calc_tau_dust(WAVEC, SWAVE, KEXT, KSCA, CONT):
    """-> TAUDUST, TAUCLSCAT, dTAUDUSTdq, dTAUCLSCATdq (NWAVE, NLAY, NDUST); scipy's interp1d does the interpolation, as in
    the reference."""
    from scipy import interpolate
    WAVEC = np.asarray(WAVEC, float); SWAVE = np.asarray(SWAVE, float); CONT = np.asarray(CONT, float)
    W, (L, ND) = WAVEC.size, CONT.shape
    out = [np.zeros((W, L, ND)) for _ in range(4)]
    for i in range(ND):
        kind = "cubic" if SWAVE.size > 2 else "linear"
        kext = interpolate.interp1d(SWAVE, KEXT[:, i], kind=kind)(WAVEC)
        ksca = interpolate.interp1d(SWAVE, KSCA[:, i], kind=kind)(WAVEC)
        m1 = (ksca < 0) & (kext > 0); m2 = (kext < 0) & (ksca > 0); m3 = kext < ksca
        le = interpolate.interp1d(SWAVE, KEXT[:, i], fill_value="extrapolate")(WAVEC)
        ls = interpolate.interp1d(SWAVE, KSCA[:, i], fill_value="extrapolate")(WAVEC)
        ksca = np.where(m1 | m3, ls, ksca); kext = np.where(m2 | m3, le, kext)
        out[0][:, :, i] = np.outer(kext * 1.0e-4, CONT[:, i]); out[1][:, :, i] = np.outer(ksca * 1.0e-4, CONT[:, i])
        out[2][:, :, i] = (kext * 1.0e-4)[:, None]; out[3][:, :, i] = (ksca * 1.0e-4)[:, None]
    return tuple(out)


# ---- k-table generator: k-distribution of an LBL spectrum in bins (Spectroscopy_0.calc_ktable_chunk :3620-3652) -----
def kdist_bins(wavecalc, kabs, vbinmin, vbinmax, g_ord, fil=None):
    """fil = (centres, nfil, dfil (NF, nbin), afil (NF, nbin)) or None.  -> (nbin, NG)."""
    wavecalc = np.asarray(wavecalc, float); kabs = np.asarray(kabs, float)
    out = np.zeros((len(vbinmin), len(g_ord)))
    for b in range(len(vbinmin)):
        mask = (wavecalc >= vbinmin[b]) & (wavecalc <= vbinmax[b])
        idx = np.argsort(kabs[mask], kind="stable")
        wavesel = wavecalc[mask]
        k_sorted = kabs[mask][idx]
        if fil is not None:
            n = int(fil[1][b])
            ils = np.interp(wavesel[idx] - fil[0][b], fil[2][:n, b], fil[3][:n, b])
        else:
            ils = np.ones_like(wavesel)
        dv = np.zeros_like(k_sorted) + (wavecalc[1] - wavecalc[0])
        g_sorted = np.cumsum(ils * dv) / np.sum(ils * dv)
        out[b] = np.interp(g_ord, g_sorted, k_sorted)
    return out
