"""Host-side layering / path geometry of the hot path (tiny arrays: stays on the host).

  layer_split     Layer_0.layer_split  (archnemesis/Layer_0.py:1402-1492)  -> BASEH, BASEP
  calc_path       AtmCalc_0.__init__   (archnemesis/AtmCalc_0.py:330-478) + Path_0: which layers a ray crosses
                  (USELAY), their slant scale factors (SF) and emission temperatures -> NLAYIN, LAYINC, SCALE,
                  EMTEMP, IMOD, i.e. exactly what the RT kernels consume.
The Curtis-Godson integration itself (layer_average) runs on the GPU: AnsfmEngine.layer_average.
Enum values are the reference's IntEnum values (LayerTypeEnum, PathObserverPointingEnum, ZenithAngleOriginEnum,
PathCalcEnum).
"""
import numpy as np

# LayerTypeEnum
EQUAL_PRESSURE, EQUAL_LOG_PRESSURE, EQUAL_HEIGHT, EQUAL_PATH_LENGTH, BASE_PRESSURE, BASE_HEIGHT = range(6)
# PathObserverPointingEnum / ZenithAngleOriginEnum
LIMB, NADIR, DISK = 0, 1, 2
IPZEN_BOTTOM, IPZEN_ALTITUDE_ZERO, IPZEN_TOP = 0, 1, 2
# PathCalcEnum flags used here
WEIGHTING_FUNCTION, UPWARD_FLUX, THERMAL_EMISSION, BROADENING = 1, 4, 64, 16384


def _interp(x, y, xn):
    """scipy.interpolate.interp1d(kind='linear', fill_value='extrapolate') (Layer_0.interp :645)."""
    x = np.asarray(x, float); y = np.asarray(y, float); xn = np.asarray(xn, float)
    idx = np.clip(np.searchsorted(x, xn), 1, len(x) - 1)
    slope = (y[idx] - y[idx - 1]) / (x[idx] - x[idx - 1])
    return slope * (xn - x[idx - 1]) + y[idx - 1]


def layer_split(RADIUS, H, P, LAYANG=0.0, LAYHT=0.0, NLAY=20, LAYTYP=EQUAL_LOG_PRESSURE, H_base=None, P_base=None):
    H = np.asarray(H, float); P = np.asarray(P, float)
    if LAYHT < H[0]:
        LAYHT = H[0]
    if LAYTYP == EQUAL_PRESSURE:
        PBOT = _interp(H, P, LAYHT)
        BASEP = np.linspace(PBOT, P[-1], NLAY + 1)[:-1]
        BASEH = _interp(P[::-1], H[::-1], BASEP)
    elif LAYTYP == EQUAL_LOG_PRESSURE:
        PBOT = _interp(H, P, LAYHT)
        BASEP = np.logspace(np.log10(PBOT), np.log10(P[-1]), NLAY + 1)[:-1]
        BASEH = _interp(P[::-1], H[::-1], BASEP)
    elif LAYTYP == EQUAL_HEIGHT:
        BASEH = np.linspace(LAYHT, H[-1], NLAY + 1)[:-1]
        BASEP = _interp(H, P, BASEH)
    elif LAYTYP == EQUAL_PATH_LENGTH:
        if not (0 <= LAYANG <= 90):
            raise AssertionError('Zennith angle should be in [0,90]')
        sin = np.sin(LAYANG * np.pi / 180); cos = np.cos(LAYANG * np.pi / 180)
        z0 = RADIUS + LAYHT
        zmax = RADIUS + H[-1]
        SMAX = np.sqrt(zmax ** 2 - (z0 * sin) ** 2) - z0 * cos
        BASES = np.linspace(0, SMAX, NLAY + 1)[:-1]
        BASEH = np.sqrt(BASES ** 2 + z0 ** 2 + 2 * BASES * z0 * cos) - RADIUS
        BASEP = np.exp(_interp(H, np.log(P), BASEH))
    elif LAYTYP == BASE_PRESSURE:
        P_base = np.asarray(P_base, float)
        if not ((P_base[-1] >= P[-1]) and (P_base[0] <= P[0])):
            raise AssertionError('Input layer base pressures out of range of atmosphere profile')
        BASEP = P_base
        BASEH = _interp(P[::-1], H[::-1], BASEP)
    elif LAYTYP == BASE_HEIGHT:
        BASEH = np.asarray(H_base, float)
        BASEP = np.exp(_interp(H, np.log(P), BASEH))
    else:
        raise ValueError('Layering scheme not defined')
    return BASEH, BASEP


def calc_path(RADIUS, BASEH, DELH, TEMP, H_top, pointing=NADIR, BOTLAY=0, ANGLE=0.0, EMISS_ANG=0.0, IPZEN=IPZEN_BOTTOM,
              path_calc=THERMAL_EMISSION):
    """Geometry of AtmCalc_0.__init__ (:330-478).  BASEH/DELH/TEMP are the Layer arrays, H_top = Layer.H[-1].
    Returns dict(NPATH, NLAYIN (P,), LAYINC (NUSE,P) int32, SCALE, EMTEMP (NUSE,P), IMOD (P,), ANGLE, BOTLAY)."""
    BASEH = np.asarray(BASEH, float); TEMP = np.asarray(TEMP, float); DELH = np.asarray(DELH, float)
    NLAY = BASEH.size
    if pointing == DISK:
        raise NotImplementedError("PathObserverPointingEnum.DISK is not implemented in the reference either (:175)")
    if pointing == LIMB:
        observer_height = np.inf
        ANGLE = 90.
    elif pointing == NADIR:
        if EMISS_ANG > 90.:
            ANGLE = 180.0 - ANGLE
            observer_height = 0.0
        else:
            observer_height = np.inf
    else:
        raise ValueError(f'path observer pointing "{pointing}" not recognised')
    if IPZEN == IPZEN_ALTITUDE_ZERO:                                            # :292-294
        z0 = RADIUS + BASEH[BOTLAY]
        ANGLE = np.arcsin(RADIUS / z0 * np.sin(ANGLE / 180. * np.pi)) / np.pi * 180.
    elif IPZEN == IPZEN_TOP:                                                    # :295-310
        z0 = RADIUS + BASEH[NLAY - 1] + DELH[NLAY - 1]
        HTAN = z0 * np.sin(ANGLE / 180. * np.pi) - RADIUS
        if HTAN <= BASEH[BOTLAY]:
            ANGLE = np.arcsin(z0 / (RADIUS + BASEH[BOTLAY]) * np.sin(ANGLE / 180. * np.pi)) / np.pi * 180.
        else:
            pointing = LIMB
            ANGLE = 90.
            for ILAY in range(NLAY):
                if BASEH[ILAY] < HTAN:
                    BOTLAY = ILAY
            if BOTLAY < NLAY - 1:
                F = (HTAN - BASEH[BOTLAY]) / (BASEH[BOTLAY + 1] - BASEH[BOTLAY])
                if F > 0.5:
                    BOTLAY = BOTLAY + 1
    Z0 = RADIUS + BASEH[BOTLAY]
    SIN2A = np.sin(ANGLE / 180. * np.pi) ** 2.
    COSA = np.cos(ANGLE / 180. * np.pi)
    if pointing == LIMB:                                                        # :338-347 down then up
        NUSE = int(2 * (NLAY - BOTLAY))
        USELAY = np.zeros(NUSE, dtype='int32')
        for IUSE in range(int(NUSE / 2)):
            USELAY[IUSE] = NLAY - 1 - IUSE
            USELAY[int(NUSE / 2) + IUSE] = BOTLAY + IUSE
    else:                                                                       # :355-375
        NUSE = NLAY - BOTLAY
        USELAY = np.zeros(NUSE, dtype='int32')
        for IUSE in range(NUSE):
            USELAY[IUSE] = IUSE if observer_height == 0.0 else NLAY - 1 - IUSE
    EMITT = TEMP[USELAY]
    SF = np.zeros(NUSE)                                                         # :381-400
    for IUSE in range(NUSE):
        STMP = (RADIUS + BASEH[USELAY[IUSE]]) ** 2. - SIN2A * Z0 ** 2.
        if STMP < 0.0:
            STMP = 0.0
        S0 = np.sqrt(STMP) - Z0 * COSA
        if USELAY[IUSE] < NLAY - 1:
            S1 = np.sqrt((RADIUS + BASEH[USELAY[IUSE] + 1]) ** 2. - SIN2A * Z0 ** 2.) - Z0 * COSA
            SF[IUSE] = (S1 - S0) / (BASEH[USELAY[IUSE] + 1] - BASEH[USELAY[IUSE]])
        if USELAY[IUSE] == NLAY - 1:
            S1 = np.sqrt((RADIUS + H_top) ** 2. - SIN2A * Z0 ** 2.) - Z0 * COSA
            SF[IUSE] = (S1 - S0) / (H_top - BASEH[USELAY[IUSE]])
    NPATH = 1                                                                   # :404-415
    if path_calc & WEIGHTING_FUNCTION:
        NPATH = NUSE
    if (path_calc & THERMAL_EMISSION) and (path_calc & BROADENING):
        NPATH = NUSE
    if path_calc & UPWARD_FLUX:
        NPATH = NUSE
    NLAYIN = np.zeros(NPATH, dtype='int32')                                     # :447-470
    LAYINC = np.zeros([NUSE, NPATH], dtype='int32')
    SCALE = np.zeros([NUSE, NPATH]); EMTEMP = np.zeros([NUSE, NPATH])
    IMOD = np.full((NPATH,), fill_value=int(path_calc), dtype=np.int32)
    for j in range(NPATH):
        NLAYIN[j] = (j + 1) + NUSE - NPATH
        for i in range(NLAYIN[j]):
            LAYINC[i, j] = USELAY[i]; EMTEMP[i, j] = EMITT[i]; SCALE[i, j] = SF[i]
    return dict(NPATH=NPATH, NLAYIN=NLAYIN, LAYINC=LAYINC, SCALE=SCALE, EMTEMP=EMTEMP, IMOD=IMOD, ANGLE=ANGLE, BOTLAY=BOTLAY)
