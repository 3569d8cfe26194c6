"""Host-side mirror of the reference's ForwardModel_0 surface for the hot path.

`make_gpu_forward_model(ForwardModel_0)` returns a subclass of the reference's own class
(ForwardModel_0.py:87) whose `CIRSrad` (ForwardModel_0.py:4376-4511) runs on the MI355X through
libansfm.so for the supported cases (ILBL = K_TABLES or LINE_BY_LINE_TABLES, IMOD = THERMAL_EMISSION, no layer emissions,
with or without analytic gradients) and whose `jacobian_nemesis` fan-out can batch the independent forward models.
Everything else (subprofretg, calc_path, conv, ...) is the reference's own host code.

`CIRSradGPU` is the mixin with the seam itself; it only needs the `*X` attributes CIRSrad reads
(SpectroscopyX, LayerX, PathX, AtmosphereX, SurfaceX, MeasurementX, ScatterX, StellarX, CIAX,
EmissionsX), so tests drive it with plain namespaces rebuilt from the C1 golden fixture.
"""
import hashlib
import os
import sys

import numpy as np

from .engine import AnsfmEngine

# IntEnum / IntFlag values of the reference (archnemesis/enum/*.py) kept as plain ints at the seam
ILBL_K_TABLES = 0                      # SpectralCalculationModeEnum.K_TABLES
ILBL_LBL_TABLES = 2                    # SpectralCalculationModeEnum.LINE_BY_LINE_TABLES
IMOD_THERMAL_EMISSION = 64             # PathCalcEnum.THERMAL_EMISSION
IMOD_MULTIPLE_SCATTERING = 256
IMOD_DOWNWARD_FLUX = 16
IMOD_SINGLE_SCATTERING_PLANE_PARALLEL = 1024
IMOD_ABSORBTION = 4096
IFORM_FLUXRATIO = 1                    # SpectraUnitEnum.FluxRatio
IFORM_TRANSIT_DEPTH = 2                # SpectraUnitEnum.TransitDepth
IFORM_ATMOSPHERIC_TRANSMISSION = 4     # SpectraUnitEnum.Atmospheric_transmission
ATM_TO_PASCAL = 101325.0               # ForwardModel_0.py:61
SQ_CM_TO_SQ_METER = 1.0e-4             # ForwardModel_0.py:66

_ENGINES = {}


def _planck(ispace, wave, temp):
    """planck (ForwardModel_0.py:6183-6227): W cm-2 sr-1 (cm-1)-1 for wavenumbers (ISPACE 0), W cm-2 sr-1 um-1 for
    wavelengths in micron (ISPACE 1); c1 = 1.1911e-12, c2 = 1.439 as there."""
    wave = np.asarray(wave, dtype=np.float64)
    c1, c2 = 1.1911e-12, 1.439
    if int(ispace) == 0:
        y, a = wave, c1 * wave ** 3
    else:
        y = 1.0e4 / wave
        a = c1 * y ** 5 / 1.0e4
    return a / (np.exp(c2 * y / temp) - 1.0)


def get_engine(device=0):
    """One engine (ctx) per GPU per process."""
    if device not in _ENGINES:
        _ENGINES[device] = AnsfmEngine(device)
    return _ENGINES[device]


# What the adapter does with a case the GPU path does not cover (another IMOD, a line shape that is not built, an unsorted
# calculation grid, a non-binary table ...): by default the call goes to the reference's own function -- the reference's
# code in the reference's process, never a CPU re-implementation of this package -- and is counted in DELEGATED; with
# set_strict(True) it raises NotImplementedError instead, so a run can prove that nothing left the GPU path.
STRICT = False
DELEGATED = {}


def set_strict(on=True):
    global STRICT
    STRICT = bool(on)


def _delegate(what):
    if STRICT:
        raise NotImplementedError("ansfm (strict): %s is outside the GPU path and delegation to the reference is off" % what)
    if what not in DELEGATED:            # said once per kind of case, counted every time (DELEGATED)
        import warnings
        warnings.warn("ansfm: %s is outside the GPU path; handed to the reference's own CPU implementation "
                      "(set_strict(True) turns this into an error)" % what, RuntimeWarning, stacklevel=3)
    DELEGATED[what] = DELEGATED.get(what, 0) + 1


NOTES = {}


def _note(what):
    """Something the adapter does differently from the reference on purpose (not a delegation: the work stays on the GPU
    path), said once as a RuntimeWarning and counted in NOTES -- e.g. NCores > 1 without joblib workers."""
    if what not in NOTES:
        import warnings
        warnings.warn("ansfm: " + what, RuntimeWarning, stacklevel=3)
    NOTES[what] = NOTES.get(what, 0) + 1


def summary():
    """What left the GPU path (DELEGATED: case -> count) and what was done differently on purpose (NOTES) since the
    process started or `reset_summary()`; `install_all()` returns the same object, so a caller can print it after a
    retrieval."""
    return {"delegated": dict(DELEGATED), "notes": dict(NOTES), "strict": STRICT}


def reset_summary():
    DELEGATED.clear()
    NOTES.clear()


class KtaTableOnDevice:
    """Stand-in for Spectroscopy.K when the k-table is taken from the .kta files straight into HBM
    (install_gpu_table_reader): knows the files, the wavenumber range and the shape the array would have.  Code that
    really needs the numbers on the host (a CPU path of the reference) gets them through __array__, read by the
    reference's own read_ktable -- slowly, once."""

    def __init__(self, paths, wavemin, wavemax, shape, reader, ext=".kta", kindex=12):
        self.ext, self._kindex = ext, kindex      # ".lta": read_lbltable returns k as its 9th item
        self.paths = [str(p) for p in paths]
        self.wavemin, self.wavemax = float(wavemin), float(wavemax)
        self.shape = tuple(int(x) for x in shape)
        self.ndim = len(self.shape)
        self.dtype = np.dtype(np.float64)
        self._reader = reader
        self._host = None
        st = []
        for p in self.paths:
            q = p if p.endswith(ext) else p + ext
            s = os.stat(q)
            st.append((q, s.st_size, s.st_mtime_ns))
        self.fingerprint = hashlib.blake2b(repr((st, self.wavemin, self.wavemax, self.shape)).encode(), digest_size=16).hexdigest()

    def __array__(self, dtype=None, copy=None):
        if self._host is None:
            k = np.zeros(self.shape)
            for i, p in enumerate(self.paths):
                k[..., i] = self._reader(p, self.wavemin, self.wavemax)[self._kindex]
            self._host = k
        return self._host if dtype is None else self._host.astype(dtype, copy=False)

    def __deepcopy__(self, memo):
        return self                      # immutable description of files; nemesisfm deep-copies Spectroscopy per call


def _table_fingerprint(S):
    """Cheap content fingerprint of the k-table held by a Spectroscopy object.  nemesisfm
    deep-copies Spectroscopy and re-reads the tables for every forward model
    (ForwardModel_0.py:480-482); the table in HBM is re-used when nothing changed."""
    K = S.K
    h = hashlib.blake2b(digest_size=16)
    if isinstance(K, KtaTableOnDevice):
        h.update(K.fingerprint.encode())
        for a in (S.WAVE, S.PRESS, S.TEMP, S.DELG):
            h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        return h.hexdigest()
    h.update(np.asarray(K.shape, dtype=np.int64).tobytes())
    h.update(str(int(S.ILBL)).encode())
    flat = K.reshape(-1)
    step = max(1, flat.size // 8192)
    h.update(np.ascontiguousarray(flat[::step]).tobytes())
    for a in (S.WAVE, S.PRESS, S.TEMP, S.DELG):
        h.update(str(getattr(a, "dtype", "")).encode())          # float32 grids select float32 semantics
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


class CIRSradGPU:
    """Mixin: CIRSrad on the GPU.  Put it in front of the reference class in the MRO."""

    ansfm_device = 0
    ansfm_keep_side_products = True   # fill LayerX.TAUGAS / TAUTOT like the reference (:3925, :3997)
    # Opt-in: CIRSrad(return_grad=True) computes the gas-amount gradients only of the gases the last subprofretg() mapped
    # to the state vector (non-zero xmap rows: what nemesisfmg's own `incpar` keeps for map2pro, :699-702); the other gases'
    # parameters of dSPECOUT then hold their continuum part only.  dSPECONV is unchanged, the gradient merge is not.
    ansfm_select_gradient_gases = False

    def subprofretg(self, *a, **k):
        xmap = super().subprofretg(*a, **k)
        self._ansfm_xmap = xmap
        return xmap

    # ---- what is supported -----------------------------------------------------------------------
    def _ansfm_supported(self, return_grad):
        S = self.SpectroscopyX
        if S.NGAS <= 0 or int(S.ILBL) not in (ILBL_K_TABLES, ILBL_LBL_TABLES) or S.K is None:
            return False
        if return_grad and (self.AtmosphereX.NVMR + 2 + self.ScatterX.NDUST > 256 or S.NGAS > 31):
            return False                   # kMaxPar of the gradient kernels' slot table; one mask bit per gas
        if getattr(self, "EmissionsX", None) is not None:
            return False
        imod = np.unique(np.asarray(self.PathX.IMOD).astype(int))
        if imod.size != 1:
            return False
        imod = int(imod[0])
        # dispatch order of CIRSrad :4478-4501: transmission / absorption come before thermal emission, thermal emission
        # before single scattering, the downward-flux variant before plain multiple scattering
        if self._ansfm_transmission_branch(imod):
            return True
        if imod & IMOD_ABSORBTION:
            return False           # calculate_absorption_spectrum (:4133) lacks `self` in the reference: never callable
        if imod & IMOD_THERMAL_EMISSION:
            return True
        if imod & IMOD_SINGLE_SCATTERING_PLANE_PARALLEL:
            return (not return_grad) and int(S.ILBL) == ILBL_K_TABLES
        if self._ansfm_scatter_branch(imod):
            return (not return_grad) and int(S.ILBL) == ILBL_K_TABLES      # the reference has no gradients there either
        return False

    @staticmethod
    def _ansfm_transmission_branch(imod):
        """the first test of CIRSrad's dispatch (:4478-4483): none of the four flags -> exp(-tau)"""
        return not (imod & (IMOD_ABSORBTION | IMOD_THERMAL_EMISSION | IMOD_MULTIPLE_SCATTERING |
                            IMOD_SINGLE_SCATTERING_PLANE_PARALLEL))

    @staticmethod
    def _ansfm_scatter_branch(imod):
        return bool(imod & IMOD_MULTIPLE_SCATTERING) and not (
            imod & (IMOD_ABSORBTION | IMOD_THERMAL_EMISSION | IMOD_SINGLE_SCATTERING_PLANE_PARALLEL | IMOD_DOWNWARD_FLUX))

    # ---- continuum opacities: the reference's own host routines when present ---------------------
    def _ansfm_continuum(self, return_grad=False):
        """Returns TAUCIA, TAUDUST, TAURAY (NWAVE,NLAY) and, for return_grad, dTAUCON (NWAVE,NPAR,NLAY)
        assembled exactly as calculate_layer_opacity does (:3916-3981)."""
        S, L = self.SpectroscopyX, self.LayerX
        dTAUCON = None
        if hasattr(self, "calculate_vertical_cia_opacity"):
            A, Sc = self.AtmosphereX, self.ScatterX
            TAUCIA, dTAUCIA = self.calculate_vertical_cia_opacity(return_grad)   # :3938 (sets LayerX.TAUCIA)
            TAURAY, dTAURAY = self.calc_tau_rayleigh(MakePlot=False)              # :3952
            L.TAURAY = TAURAY
            TAUDUST1, TAUCLSCAT, dTAUDUST1, dTAUCLSCAT = self.calc_tau_dust()     # :3963
            TAUDUST1 = np.clip(np.nan_to_num(TAUDUST1), 0, 1e20)                 # :3966
            TAUDUST = np.sum(TAUDUST1, 2)
            L.TAUDUST = TAUDUST
            L.TAUSCAT = np.sum(TAUCLSCAT, 2)
            L.TAUCLSCAT = TAUCLSCAT
            if return_grad:
                dTAUCON = np.zeros((S.NWAVE, A.NVMR + 2 + Sc.NDUST, L.NLAY))
                if dTAUCIA is not None:                                           # :3940-3942
                    # d TAUCIA / d amount of every gas: per unit column, with the gas axis ahead of the layer axis
                    dTAUCON[:, 0:A.NVMR, :] += np.moveaxis(dTAUCIA[:, :, 0:A.NVMR], 2, 1) / np.asarray(L.TOTAM)[None, None, :]
                    dTAUCON[:, A.NVMR, :] += dTAUCIA[:, :, A.NVMR]
                if dTAURAY is not None:                                           # :3955-3957
                    for i in range(A.NVMR):
                        dTAUCON[:, i, :] += dTAURAY[:, :]
                for i in range(Sc.NDUST):                                         # :3978-3980
                    dTAUCON[:, A.NVMR + 1 + i, :] += dTAUDUST1[:, :, i]
        else:  # standalone: continuum arrays were provided on LayerX
            dTAUCON = getattr(L, "dTAUCON", None) if return_grad else None
            z = np.zeros((S.NWAVE, L.NLAY))
            TAUCIA = getattr(L, "TAUCIA", None)
            TAUCIA = z if TAUCIA is None else TAUCIA
            TAURAY = getattr(L, "TAURAY", None)
            TAURAY = z if TAURAY is None else TAURAY
            TAUDUST = getattr(L, "TAUDUST", None)
            TAUDUST = z if TAUDUST is None else TAUDUST
        return TAUCIA, TAUDUST, TAURAY, dTAUCON

    @staticmethod
    def _ansfm_total_opacity(TAUGAS, TAUCIA, TAUDUST, TAURAY):
        """LayerX.TAUTOT (NWAVE, NG, NLAY): the continuum terms joined to every g-ordinate in the order of :3989."""
        out = TAUGAS.copy()
        for cont in (TAUCIA, TAUDUST, TAURAY):
            out += cont[:, None, :]
        return out

    def _ansfm_units_and_surface(self):
        """xfac and EMISSIVITY exactly as calculate_thermal_emission_spectrum prepares them
        (ForwardModel_0.py:4157-4189).  REFLECTANCE is identically zero there (:4208-4213), so the
        solar-reflection term vanishes and SOLFLUX is not needed."""
        import scipy.interpolate
        S = self.SpectroscopyX
        xfac = None
        if int(self.MeasurementX.IFORM) == IFORM_FLUXRATIO:
            xfac = np.ones(S.NWAVE) * np.pi * 4. * np.pi * ((self.AtmosphereX.RADIUS) * 1.0e2) ** 2.
            self.StellarX.calc_solar_flux()
            f = scipy.interpolate.interp1d(self.StellarX.WAVE, self.StellarX.SOLFLUX)
            xfac = xfac / f(S.WAVE)
        if self.SurfaceX.TSURF > 0.0:
            f = scipy.interpolate.interp1d(self.SurfaceX.VEM, self.SurfaceX.EMISSIVITY)
            emissivity = f(S.WAVE)
        else:
            emissivity = None
        return xfac, emissivity

    def _ansfm_upload_table(self, eng):
        S = self.SpectroscopyX
        fp = _table_fingerprint(S)
        if getattr(eng, "_table_fp", None) != fp:
            if isinstance(S.K, KtaTableOnDevice):   # .kta / .lta files -> HBM without a host array (install_gpu_table_reader)
                up = eng.upload_lbltable_files if S.K.ext == ".lta" else eng.upload_ktable_files
                WAVE = up(S.K.paths, S.K.wavemin, S.K.wavemax)[0]
                if WAVE.shape != np.shape(S.WAVE) or not np.array_equal(WAVE, np.asarray(S.WAVE, dtype=np.float64)):
                    raise ValueError("the .kta files no longer give the wavenumber grid Spectroscopy.WAVE holds")
            elif int(S.ILBL) == ILBL_LBL_TABLES:      # K (NWAVE,NP,|NT|,NGAS); TEMP (NP,|NT|) when NT < 0
                eng.upload_lbltable(np.ascontiguousarray(S.K, dtype=np.float64), S.PRESS, S.TEMP, S.WAVE)
            else:
                eng.upload_ktable(np.ascontiguousarray(S.K, dtype=np.float64), S.PRESS, S.TEMP, S.WAVE, S.DELG)
            eng._table_fp = fp

    def _ansfm_layer_inputs(self):
        S, L, A = self.SpectroscopyX, self.LayerX, self.AtmosphereX
        f_gas = np.zeros((S.NGAS, L.NLAY))
        for i in range(S.NGAS):
            IGAS = A.locate_gas(S.ID[i], S.ISO[i])
            f_gas[i, :] = L.AMOUNT[:, IGAS] * SQ_CM_TO_SQ_METER             # :3861
        return f_gas

    # ---- single-scattering branch: host preparation of :4251-4336 for what has no g axis ---------------------------
    def _ansfm_cirsrad_singlescatt(self, eng, taucont, TAURAY, f_gas):
        """calculate_single_scattering_plane_parallel_spectrum (:4251-4336): scattering angle, phase functions of the aerosols
        and of Rayleigh scattering, their opacity-weighted mean per layer, solar flux, units, emissivity and BRDF on the host;
        OMEGA and the layer loop (calc_singlescatt_plane_spectrum :6509) on the device.  -> SPECOUT (NWAVE, NPATH)."""
        import scipy.interpolate
        S, L, P, Sc, Su, M = self.SpectroscopyX, self.LayerX, self.PathX, self.ScatterX, self.SurfaceX, self.MeasurementX
        WAVE = np.asarray(S.WAVE, dtype=np.float64)
        W, ND = WAVE.size, int(Sc.NDUST)
        sol, emi, azi = (np.asarray(a, dtype=np.float64) for a in (P.SOL_ANG, P.EMISS_ANG, P.AZI_ANG))
        rad = np.pi / 180.
        calpha = np.sin(sol * rad) * np.sin(emi * rad) * np.cos(azi * rad - np.pi) - np.cos(emi * rad) * np.cos(sol * rad)   # :4267
        alpha = np.arccos(calpha) / np.pi * 180.
        pf = np.zeros((W, len(sol), ND + 1))                                     # :4271-4274
        pf[:, :, 0:ND] = Sc.calc_phase(alpha, WAVE)
        pf[:, :, ND] = Sc.calc_phase_ray(alpha)
        TAUSCAT, TAUCLSCAT = L.TAUSCAT, L.TAUCLSCAT
        tsca = TAURAY + TAUSCAT
        # layer-mean phase function of every path: sum_c P_c(alpha) tau_c / (TAURAY + TAUSCAT) where that is > 0 (:4314-4322)
        num = np.einsum("wpc,wlc->pwl", pf[:, :, 0:ND], TAUCLSCAT) + pf[:, :, ND].T[:, :, None] * TAURAY[None, :, :]
        phase = np.where(num > 0, num / np.where(num > 0, tsca[None], 1.0), num)
        if self.StellarX.SOLEXIST is True:                                       # :4286-4290
            self.StellarX.calc_solar_flux()
            solar = np.interp(WAVE, self.StellarX.WAVE, self.StellarX.SOLFLUX)
        else:
            solar = np.zeros(W)
        xfac = np.ones(W)
        if int(M.IFORM) == IFORM_FLUXRATIO:                                      # :4293-4298 (the power spectrum on VCONV)
            xfac *= np.pi * 4. * np.pi * ((self.AtmosphereX.RADIUS) * 1.0e2) ** 2.
            xfac = xfac / scipy.interpolate.interp1d(self.StellarX.VCONV, self.StellarX.SOLSPEC)(WAVE)
        if Su.TSURF > 0.0:
            emissivity = scipy.interpolate.interp1d(Su.VEM, Su.EMISSIVITY)(WAVE)
        else:
            emissivity = np.zeros(W)
        if int(Su.LOWBC) != 0:                                                   # :4308-4311 (0 = THERMAL)
            BRDF = Su.calc_BRDF(WAVE, P.SOL_ANG, P.EMISS_ANG, P.AZI_ANG)
        else:
            BRDF = np.zeros((W, len(sol)))
        NPATH = len(sol)
        return eng.cirsrad_ck_singlescatt(
            int(M.ISPACE), np.asarray(L.PRESS, dtype=np.float64), np.asarray(L.TEMP, dtype=np.float64), f_gas, taucont, tsca,
            phase, np.asarray(P.NLAYIN, dtype=np.int32).reshape(NPATH), np.asarray(P.LAYINC, dtype=np.int32).reshape(-1, NPATH),
            np.asarray(P.SCALE, dtype=np.float64).reshape(-1, NPATH), np.asarray(P.EMTEMP, dtype=np.float64).reshape(-1, NPATH),
            float(Su.TSURF), emissivity, BRDF, solar, sol, emi, xfac=xfac)

    # ---- scattering branch: host preparation of scloud11wave (:5018-5165) for what has no g axis ----------------
    def _ansfm_cirsrad_scatter(self, eng, TAUCIA, TAUDUST, TAURAY, f_gas):
        """calculate_multiple_scattering_spectrum (:4343-4374) + scloud11wave (:5018-5165): the boundary vectors, phase
        functions and aerosol fractions are small host arrays prepared as the reference does; the gas opacities, TAUTOT,
        OMEGA and BB are formed on the device (ansfm_cirsrad_ck_scatter).  -> SPECOUT (NWAVE, NPATH)."""
        import scipy.interpolate
        S, L, P, Sc, Su, M = self.SpectroscopyX, self.LayerX, self.PathX, self.ScatterX, self.SurfaceX, self.MeasurementX
        WAVE = np.asarray(S.WAVE, dtype=np.float64)
        W, ISPACE = WAVE.size, int(M.ISPACE)
        solar = np.zeros(W)
        if self.StellarX.SOLEXIST:                                               # :4353-4357
            self.StellarX.calc_solar_flux()
            solar[:] = scipy.interpolate.interp1d(self.StellarX.WAVE, self.StellarX.SOLFLUX)(WAVE)
        # (the reference works xfac out here, :4359-4368, and never applies it to this branch's spectrum: nor is it here)
        NMU, NDUST = int(Sc.NMU), int(Sc.NDUST)
        RADGROUND = np.zeros((W, NMU))
        if Su.GASGIANT or (Su.TSURF <= 0.0):                                     # :5082-5089
            RADGROUND[:, :] = _planck(ISPACE, WAVE, L.TEMP[0])[:, None]
        else:
            emis = scipy.interpolate.interp1d(Su.VEM, Su.EMISSIVITY)(WAVE)
            RADGROUND[:, :] = (_planck(ISPACE, WAVE, Su.TSURF) * emis)[:, None]
        if (not Su.GASGIANT) and int(Su.LOWBC) != 0:                             # :5091-5094 (0 = THERMAL)
            BRDF = self.calc_brdf_matrix(WAVEC=WAVE, Surface=Su, Scatter=Sc)
        else:
            BRDF = np.zeros((W, NMU, NMU, int(Sc.NF) + 1))
        TAUSCAT, TAUCLSCAT = L.TAUSCAT, L.TAUCLSCAT
        FRAC = np.zeros((W, L.NLAY, NDUST))                                      # :5106-5114
        pos = TAUSCAT > 0.0
        FRAC[pos] = TAUCLSCAT[pos] / TAUSCAT[pos][:, None]
        FRAC = np.ascontiguousarray(np.transpose(FRAC, (0, 2, 1)))
        THETA = np.asarray(Sc.THETA, dtype=np.float64)
        PHASE = np.zeros((NDUST, W, 2, THETA.size))                              # :5125-5137
        if int(Sc.IMIE) == 0:                                                    # Henyey-Greenstein: f, g1, g2
            for i in range(NDUST):
                PHASE[i, :, 0, -1] = np.interp(WAVE, Sc.WAVE, Sc.F.T[i])
                PHASE[i, :, 0, -2] = np.interp(WAVE, Sc.WAVE, Sc.G1.T[i])
                PHASE[i, :, 0, -3] = np.interp(WAVE, Sc.WAVE, Sc.G2.T[i])
        else:
            PHASE[:, :, 0, :] = np.transpose(Sc.calc_phase(THETA, WAVE), (2, 0, 1))
        PHASE[:, :, 1, :] = np.cos(THETA * np.pi / 180)
        args = dict(ISPACE=ISPACE, lp=np.array(L.PRESS, dtype=np.float64), lt=np.array(L.TEMP, dtype=np.float64), f_gas=f_gas,
                    TAUCIA=np.asarray(TAUCIA, dtype=np.float64), TAUDUST=np.asarray(TAUDUST, dtype=np.float64),
                    TAURAY=np.asarray(TAURAY, dtype=np.float64), TAUSCAT=np.asarray(TAUSCAT, dtype=np.float64),
                    PHASE=np.ascontiguousarray(PHASE[:, :, :, ::-1]), FRAC=FRAC, RADGROUND=RADGROUND,
                    SOL_ANG=np.array(P.SOL_ANG, dtype=np.float64).reshape(-1), EMISS_ANG=np.array(P.EMISS_ANG, dtype=np.float64).reshape(-1),
                    AZI_ANG=np.array(P.AZI_ANG, dtype=np.float64).reshape(-1), solar=solar, LOWBC=int(Su.LOWBC), BRDF=BRDF,
                    MU=np.array(Sc.MU, dtype=np.float64), WTMU=np.array(Sc.WTMU, dtype=np.float64), NF=int(Sc.NF), NPHI=int(Sc.NPHI),
                    IRAY=int(Sc.IRAY), IMIE=int(Sc.IMIE))
        if eng is None:                  # the staged Jacobian route keeps the arguments and batches the call
            return args
        return self._ansfm_scatter_call(eng, args)

    @staticmethod
    def _ansfm_scatter_call(eng, a):
        return eng.cirsrad_ck_scatter(a["ISPACE"], a["lp"], a["lt"], a["f_gas"], a["TAUCIA"], a["TAUDUST"], a["TAURAY"], a["TAUSCAT"],
                                      a["PHASE"], a["FRAC"], a["RADGROUND"], a["SOL_ANG"], a["EMISS_ANG"], a["AZI_ANG"], a["solar"],
                                      a["LOWBC"], a["BRDF"], a["MU"], a["WTMU"], a["NF"], a["NPHI"], a["IRAY"], a["IMIE"])

    # ---- the seam ---------------------------------------------------------------------------------
    def CIRSrad(self, return_grad=False):
        if not self._ansfm_supported(return_grad):
            _delegate("CIRSrad case (IMOD / ILBL / emissions)")
            base = super()
            if hasattr(base, "CIRSrad"):
                return base.CIRSrad(return_grad)      # the reference's own implementation, in its process
            raise NotImplementedError("CIRSrad: only ILBL=K_TABLES, IMOD=THERMAL_EMISSION run on the GPU so far")
        eng = get_engine(self.ansfm_device)
        S, L, P = self.SpectroscopyX, self.LayerX, self.PathX
        self._ansfm_upload_table(eng)
        TAUCIA, TAUDUST, TAURAY, dTAUCON = self._ansfm_continuum(return_grad)
        taucont = TAUCIA + TAUDUST + TAURAY                                  # :3989 (g-independent part)
        f_gas = self._ansfm_layer_inputs()
        if self._ansfm_scatter_branch(int(np.unique(np.asarray(P.IMOD).astype(int))[0])):
            SPECOUT = self._ansfm_cirsrad_scatter(eng, TAUCIA, TAUDUST, TAURAY, f_gas)
            if self.ansfm_keep_side_products:
                L.TAUGAS = eng.get_taugas(L.NLAY, 0)                         # :3925
                L.TAUTOT = self._ansfm_total_opacity(L.TAUGAS, TAUCIA, TAUDUST, TAURAY)
            return SPECOUT
        imod0 = int(np.unique(np.asarray(P.IMOD).astype(int))[0])
        if self._ansfm_transmission_branch(imod0):
            import scipy.interpolate
            xf = None
            if int(self.MeasurementX.IFORM) == IFORM_ATMOSPHERIC_TRANSMISSION:   # :4119-4127: times the solar flux
                self.StellarX.calc_solar_flux()
                xf = scipy.interpolate.interp1d(self.StellarX.WAVE, self.StellarX.SOLFLUX)(S.WAVE)
            NPATH = np.asarray(P.LAYINC).shape[1]
            geom = (np.asarray(P.NLAYIN, dtype=np.int32).reshape(NPATH), np.asarray(P.LAYINC, dtype=np.int32).reshape(-1, NPATH),
                    np.asarray(P.SCALE, dtype=np.float64).reshape(-1, NPATH))
            lp, lt = np.asarray(L.PRESS, dtype=np.float64), np.asarray(L.TEMP, dtype=np.float64)
            if return_grad:                                                  # :4128-4131, then :4504-4508
                A = self.AtmosphereX
                NVMR = int(A.NVMR)
                NPAR = NVMR + 2 + int(self.ScatterX.NDUST)
                igas_map = np.array([A.locate_gas(S.ID[i], S.ISO[i]) for i in range(S.NGAS)], dtype=np.int32)
                SPECOUT, dSPECOUT = eng.cirsradg_ck_transmission(lp, lt, f_gas, taucont, dTAUCON, NVMR, NPAR, igas_map, *geom,
                                                                 xfac=xf)
            else:
                SPECOUT = eng.cirsrad_ck_transmission(lp, lt, f_gas, taucont, *geom, xfac=xf)
            if self.ansfm_keep_side_products:
                L.TAUGAS = eng.get_taugas(L.NLAY, 0)
                L.TAUTOT = self._ansfm_total_opacity(L.TAUGAS, TAUCIA, TAUDUST, TAURAY)
            if return_grad:
                return SPECOUT, dSPECOUT, np.zeros_like(SPECOUT)             # dTSURF: zeros through the g-quadrature
            return SPECOUT
        if (imod0 & IMOD_SINGLE_SCATTERING_PLANE_PARALLEL) and not (imod0 & IMOD_THERMAL_EMISSION):   # dispatch order :4487-4493
            SPECOUT = self._ansfm_cirsrad_singlescatt(eng, taucont, TAURAY, f_gas)
            if self.ansfm_keep_side_products:
                L.TAUGAS = eng.get_taugas(L.NLAY, 0)
                L.TAUTOT = self._ansfm_total_opacity(L.TAUGAS, TAUCIA, TAUDUST, TAURAY)
            return SPECOUT
        xfac, emissivity = self._ansfm_units_and_surface()
        NPATH = int(P.NPATH) if hasattr(P, "NPATH") else np.asarray(P.LAYINC).shape[1]
        LAYINC = np.asarray(P.LAYINC, dtype=np.int32).reshape(-1, NPATH)
        NLAYIN = np.asarray(P.NLAYIN, dtype=np.int32).reshape(NPATH)
        SCALE = np.asarray(P.SCALE, dtype=np.float64).reshape(-1, NPATH)
        EMTEMP = np.asarray(P.EMTEMP, dtype=np.float64).reshape(-1, NPATH)
        if return_grad:
            A = self.AtmosphereX
            NVMR = int(A.NVMR)
            NPAR = NVMR + 2 + int(self.ScatterX.NDUST)
            igas_map = np.array([A.locate_gas(S.ID[i], S.ISO[i]) for i in range(S.NGAS)], dtype=np.int32)
            xm = getattr(self, "_ansfm_xmap", None) if self.ansfm_select_gradient_gases else None
            selected = xm is not None and np.ndim(xm) == 3 and np.shape(xm)[1] == NPAR and hasattr(eng, "set_gradient_gases")
            if selected:
                used = np.any(np.asarray(xm) != 0.0, axis=(0, 2))
                eng.set_gradient_gases([i for i in range(S.NGAS) if used[igas_map[i]]], temperature=bool(used[NVMR]))
            try:
                SPECOUT, dSPECOUT, dTSURF = eng.cirsradg_ck_thermal(
                    int(self.MeasurementX.ISPACE), np.asarray(L.PRESS, dtype=np.float64),
                    np.asarray(L.TEMP, dtype=np.float64), f_gas, taucont, dTAUCON, NVMR, NPAR, igas_map, NLAYIN, LAYINC,
                    SCALE, EMTEMP, float(self.SurfaceX.TSURF), EMISSIVITY=emissivity, xfac=xfac)
            finally:
                if selected:
                    eng.set_gradient_gases(None)
        else:
            SPECOUT = eng.cirsrad_ck_thermal(
                int(self.MeasurementX.ISPACE), np.asarray(L.PRESS, dtype=np.float64),
                np.asarray(L.TEMP, dtype=np.float64), f_gas, taucont, NLAYIN, LAYINC, SCALE, EMTEMP,
                float(self.SurfaceX.TSURF), EMISSIVITY=emissivity,
                SOL_ANG=np.asarray(P.SOL_ANG, dtype=np.float64).reshape(NPATH),
                EMISS_ANG=np.asarray(P.EMISS_ANG, dtype=np.float64).reshape(NPATH), xfac=xfac)
        if self.ansfm_keep_side_products:
            L.TAUGAS = eng.get_taugas(L.NLAY, 0)                             # :3925
            L.TAUTOT = self._ansfm_total_opacity(L.TAUGAS, TAUCIA, TAUDUST, TAURAY)
        if return_grad:
            return SPECOUT, dSPECOUT, dTSURF          # (NWAVE,NPATH), (NWAVE,NPAR,NLAYINmax,NPATH), (NWAVE,NPATH)
        return SPECOUT                                                        # (NWAVE, NPATH)


def make_gpu_forward_model(reference_forward_model_cls, device=0):
    """Subclass of the reference's ForwardModel_0 with the GPU CIRSrad seam and `jacobian_nemesis` without the joblib
    fan-out (jacobian_dropin.JacobianGPU)."""
    from .jacobian_dropin import JacobianGPU
    return type("ForwardModel_0", (JacobianGPU, CIRSradGPU, reference_forward_model_cls),
                {"ansfm_device": device, "__doc__": reference_forward_model_cls.__doc__})


def install_gpu_forward_model(device=0):
    """Make the retrieval drivers build the GPU forward model: `coreretOE` imports the class at call time (`from archnemesis
    import ForwardModel_0`, OptimalEstimation_0.py:1255) and `Retrievals` / `NestedSampling_0` build `ans.ForwardModel_0(...)`
    (Retrievals.py:102, :181, :256), so the package attribute `archnemesis.ForwardModel_0` is what has to name the subclass
    (the submodule of the same name stays reachable through importlib / sys.modules).  The reference's class is kept as
    `archnemesis._ansfm_reference_ForwardModel_0`; calling this twice is harmless.  Returns the subclass."""
    import importlib
    pkg = importlib.import_module("archnemesis")
    ref = getattr(pkg, "_ansfm_reference_ForwardModel_0", None)
    if ref is None:
        ref = pkg.ForwardModel_0
        if not isinstance(ref, type):          # the attribute named the submodule (no star import of the class): take the class
            ref = importlib.import_module("archnemesis.ForwardModel_0").ForwardModel_0
        pkg._ansfm_reference_ForwardModel_0 = ref
    cls = make_gpu_forward_model(ref, device)
    pkg.ForwardModel_0 = cls
    for modname in ("archnemesis.OptimalEstimation_0", "archnemesis.Retrievals", "archnemesis.NestedSampling_0"):
        mod = sys.modules.get(modname)
        if mod is not None and isinstance(getattr(mod, "ForwardModel_0", None), type):
            mod.ForwardModel_0 = cls               # module-level `from archnemesis import ForwardModel_0` bindings
    return cls


def uninstall_gpu_forward_model():
    import importlib
    pkg = importlib.import_module("archnemesis")
    ref = getattr(pkg, "_ansfm_reference_ForwardModel_0", None)
    if ref is not None:
        pkg.ForwardModel_0 = ref
        for modname in ("archnemesis.OptimalEstimation_0", "archnemesis.Retrievals", "archnemesis.NestedSampling_0"):
            mod = sys.modules.get(modname)
            if mod is not None and isinstance(getattr(mod, "ForwardModel_0", None), type):
                mod.ForwardModel_0 = ref


def install_gpu_scattering_core(device=0):
    """Route the reference's multiple-scattering core through the GPU.

    ForwardModel_0.scloud11wave (ForwardModel_0.py:5018) prepares RADGROUND/BB/FRAC/OMEGA/PHASE_ARRAY on the host
    and imports `scloud11wave_core` from archnemesis.Multiple_Scattering_Core at call time (:5050); replacing that
    module attribute keeps all of the reference's host preparation and swaps only the core (K7).  More than 16 paths are run
    in groups of 16 by the engine; more than 32 streams go to the reference's own function."""
    import importlib
    msc = importlib.import_module("archnemesis.Multiple_Scattering_Core")
    eng = get_engine(device)
    ref_core = getattr(msc, "_ansfm_reference_core", None) or msc.scloud11wave_core

    def scloud11wave_core(phasarr, radg, sol_angs, emiss_angs, solar, aphis, lowbc, brdf_matrix, mu1, wt1, nf, vwaves, bnu,
                          taus, tauray, omegas_s, nphi, iray, imie, lfrac):
        try:
            return eng.scloud11wave_core(phasarr, radg, sol_angs, emiss_angs, solar, aphis, int(lowbc), brdf_matrix, mu1, wt1,
                                         nf, vwaves, bnu, taus, tauray, omegas_s, nphi, int(iray), int(imie), lfrac)
        except NotImplementedError:
            _delegate("scloud11wave_core with more than 32 streams")
            return ref_core(phasarr, radg, sol_angs, emiss_angs, solar, aphis, lowbc, brdf_matrix, mu1, wt1, nf, vwaves, bnu,
                            taus, tauray, omegas_s, nphi, iray, imie, lfrac)

    msc._ansfm_reference_core = ref_core
    msc.scloud11wave_core = scloud11wave_core
    return scloud11wave_core


def install_gpu_line_kernel(device=0):
    """Route LineData_0.add_line_set_monochromatic_absorption (LineData_0.py:280) through the GPU for the line shapes
    that are built (voigt / lorentz / gaussian); other `lineshape_fn` objects go to the reference's own kernel."""
    import importlib
    ld = importlib.import_module("archnemesis.LineData_0")
    ls = importlib.import_module("archnemesis.lineshape")
    eng = get_engine(device)
    ref_fn = getattr(ld, "_ansfm_reference_line_kernel", None) or ld.add_line_set_monochromatic_absorption
    ids = {id(ls.voigt): 0, id(ls.lorentz): 4, id(ls.gaussian): 12}

    def add_line_set_monochromatic_absorption(wn_grid, lineshape_fn, t_calc, t_ref, p_calc, p_ref, q_ratio,
                                              isotopic_abundance, isotopic_mass, mol_mix_frac, broadening_params, nu, sw,
                                              e_lower, stimulated_emission_at_t_ref, out, store=None, s_floor=0,
                                              wn_calc_window=25.0, wn_approx_window=75.0):
        lid = ids.get(id(lineshape_fn))
        ok = (lid is not None and isinstance(out, np.ndarray) and out.dtype == np.float64 and out.flags.c_contiguous
              and (store is None or (store.dtype == np.float64 and store.flags.c_contiguous)))
        if not ok:
            _delegate("line shape / buffer layout of add_line_set_monochromatic_absorption")
            return ref_fn(wn_grid, lineshape_fn, t_calc, t_ref, p_calc, p_ref, q_ratio, isotopic_abundance, isotopic_mass,
                          mol_mix_frac, broadening_params, nu, sw, e_lower, stimulated_emission_at_t_ref, out, store, s_floor,
                          wn_calc_window, wn_approx_window)
        eng.add_line_set_monochromatic_absorption(wn_grid, lid, t_calc, t_ref, p_calc, p_ref, q_ratio, isotopic_abundance,
                                                  isotopic_mass, mol_mix_frac, broadening_params, nu, sw, e_lower,
                                                  stimulated_emission_at_t_ref, out, store, s_floor, wn_calc_window,
                                                  wn_approx_window)
        return

    ld._ansfm_reference_line_kernel = ref_fn
    ld.add_line_set_monochromatic_absorption = add_line_set_monochromatic_absorption
    return add_line_set_monochromatic_absorption


def install_gpu_gradient_maps(device=0):
    """Route ForwardModel_0.map2pro / map2xvec (ForwardModel_0.py:5319, :5387) -- the layer -> profile -> state-vector
    gradient maps nemesisfmg applies right after CIRSrad(return_grad=True) (:704-711) -- through the GPU.  The arrays
    stay on the device between the three steps when they are passed on unmodified, as nemesisfmg does."""
    import importlib
    fm = importlib.import_module("archnemesis.ForwardModel_0")
    eng = get_engine(device)
    if not hasattr(fm, "_ansfm_reference_maps"):
        fm._ansfm_reference_maps = (fm.map2pro, fm.map2xvec)

    def map2pro(dSPECIN, NWAVE, NVMR, NDUST, NPRO, NPATH, NLAYIN, LAYINC, DTE, DAM, DCO, INCPAR=[-1]):
        return eng.map2pro(dSPECIN, NWAVE, NVMR, NDUST, NPRO, NPATH, NLAYIN, LAYINC, DTE, DAM, DCO, INCPAR=INCPAR)

    def map2xvec(dSPECIN, NWAVE, NVMR, NDUST, NPRO, NPATH, NX, xmap):
        return eng.map2xvec(dSPECIN, NWAVE, NVMR, NDUST, NPRO, NPATH, NX, xmap)

    fm.map2pro = map2pro
    fm.map2xvec = map2xvec
    return map2pro, map2xvec


def install_gpu_layering(device=0):
    """Route Layer_0.layer_average / layer_averageg (Layer_0.py:755, :1032; called by the Layer_0 methods of the same
    names, :509 / :552, from calc_path / calc_pathg) through the GPU."""
    import importlib
    l0 = importlib.import_module("archnemesis.Layer_0")
    eng = get_engine(device)
    if not hasattr(l0, "_ansfm_reference_layering"):
        l0._ansfm_reference_layering = (l0.layer_average, l0.layer_averageg)

    def _wrap(fn):
        def f(RADIUS, H, P, T, ID, VMR, DUST, PARAH2, BASEH, BASEP, LAYANG=0.0, LAYINT=0, LAYHT=0.0, NINT=101,
              DUST_UNITS=None, XMOLWT=None):
            r = list(fn(RADIUS, H, P, T, ID, VMR, DUST, PARAH2, BASEH, BASEP, LAYANG=LAYANG, LAYINT=int(LAYINT), LAYHT=LAYHT,
                        NINT=NINT, DUST_UNITS=DUST_UNITS, XMOLWT=XMOLWT))
            if np.ndim(VMR) == 1:                      # single-gas profiles come back 1-D in the reference
                r[4] = r[4][:, 0]; r[5] = r[5][:, 0]
            if DUST is not None and np.ndim(DUST) == 1:
                r[6] = r[6][:, 0]
            return tuple(r)
        return f

    l0.layer_average = _wrap(eng.layer_average)
    l0.layer_averageg = _wrap(eng.layer_averageg)
    return l0.layer_average, l0.layer_averageg


def install_gpu_convolution(device=0):
    """Route the ILS convolution kernels of Measurement_0 -- lblconv (:3335), lblconvg (:3799), lblconv_fil (:3549),
    lblconvg_fil (:3992), which the Measurement_0.lblconv / lblconvg methods call by module-global name with IGEOM = int
    (the way nemesisfm / nemesisfmg use them), and the *_ngeom variants (:3444, :3685, :3614, :3912; IGEOM = 'All') --
    through the GPU, and the filter-function branch of the k-table methods Measurement_0.conv / convg."""
    import importlib
    m0 = importlib.import_module("archnemesis.Measurement_0")
    eng = get_engine(device)
    if not hasattr(m0, "_ansfm_reference_conv"):
        m0._ansfm_reference_conv = (m0.lblconv, m0.lblconvg, m0.lblconv_fil, m0.lblconvg_fil, m0.lblconv_ngeom,
                                    m0.lblconvg_ngeom, m0.lblconv_fil_ngeom, m0.lblconvg_fil_ngeom)
    ref = m0._ansfm_reference_conv

    def _ascending(v):
        v = np.asarray(v)
        return v.ndim == 1 and np.all(v[1:] >= v[:-1])

    def lblconv(nwave, vwave, y, nconv, vconv, ishape, fwhm):
        if np.ndim(y) != 1 or not _ascending(vwave):
            _delegate("lblconv on an unsorted grid")
            return ref[0](nwave, vwave, y, nconv, vconv, ishape, fwhm)
        return eng.lblconv(nwave, vwave, y, nconv, vconv, int(ishape), fwhm)

    def lblconvg(nwave, vwave, y, dydx, nconv, vconv, ishape, fwhm):
        if np.ndim(y) != 1 or np.ndim(dydx) != 2 or not _ascending(vwave):
            _delegate("lblconvg on an unsorted grid")
            return ref[1](nwave, vwave, y, dydx, nconv, vconv, ishape, fwhm)
        return eng.lblconvg(nwave, vwave, y, dydx, nconv, vconv, int(ishape), fwhm)

    def lblconv_fil(nwave, vwave, y, nconv, vconv, nfil, vfil, afil):
        if np.ndim(y) != 1 or not _ascending(vwave):
            _delegate("lblconv_fil on an unsorted grid")
            return ref[2](nwave, vwave, y, nconv, vconv, nfil, vfil, afil)
        return eng.lblconv_fil(nwave, vwave, y, nconv, vconv, nfil, vfil, afil)

    def lblconvg_fil(nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil):
        if np.ndim(y) != 1 or np.ndim(dydx) != 2 or not _ascending(vwave):
            _delegate("lblconvg_fil on an unsorted grid")
            return ref[3](nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil)
        return eng.lblconvg_fil(nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil)

    def lblconv_ngeom(nwave, vwave, y, nconv, vconv, ishape, fwhm):
        if np.ndim(y) != 2 or not _ascending(vwave) or not fwhm > 0.0:     # (the reference returns nothing for y.ndim != 2)
            _delegate("lblconv_ngeom on an unsorted grid")
            return ref[4](nwave, vwave, y, nconv, vconv, ishape, fwhm)
        return eng.lblconv_ngeom(nwave, vwave, y, nconv, vconv, int(ishape), fwhm)

    def lblconvg_ngeom(nwave, vwave, y, dydx, nconv, vconv, ishape, fwhm):
        if np.ndim(y) != 2 or np.ndim(dydx) != 3 or not _ascending(vwave) or not fwhm > 0.0:
            _delegate("lblconvg_ngeom on an unsorted grid")
            return ref[5](nwave, vwave, y, dydx, nconv, vconv, ishape, fwhm)
        return eng.lblconvg_ngeom(nwave, vwave, y, dydx, nconv, vconv, int(ishape), fwhm)

    def lblconv_fil_ngeom(nwave, vwave, y, nconv, vconv, nfil, vfil, afil):
        if np.ndim(y) != 2 or not _ascending(vwave):
            _delegate("lblconv_fil_ngeom on an unsorted grid")
            return ref[6](nwave, vwave, y, nconv, vconv, nfil, vfil, afil)
        return eng.lblconv_fil_ngeom(nwave, vwave, y, nconv, vconv, nfil, vfil, afil)

    def lblconvg_fil_ngeom(nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil):
        if np.ndim(y) != 2 or np.ndim(dydx) != 3 or not _ascending(vwave):
            _delegate("lblconvg_fil_ngeom on an unsorted grid")
            return ref[7](nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil)
        return eng.lblconvg_fil_ngeom(nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil)

    # k-table runs: Measurement_0.conv (:2288) / convg (:2467).  Their FWHM < 0 branch (one filter per convolution point,
    # Python loops over NCONV x window with NX-vectors) goes to the GPU; FWHM == 0 (one scipy interp1d call) and the
    # cubic-spline FWHM > 0 branch of conv (convg raises there in the reference, :2611) are the reference's own host code.
    cls = m0.Measurement_0
    if not hasattr(cls, "_ansfm_reference_conv_methods"):
        cls._ansfm_reference_conv_methods = (cls.conv, cls.convg)
    rconv, rconvg = cls._ansfm_reference_conv_methods

    def _fil_case(self, Wave, IGEOM):
        return (not isinstance(IGEOM, str)) and self.FWHM < 0.0 and _ascending(Wave)

    def conv(self, Wave, ModSpec, IGEOM='All', FWHMEXIST=''):
        if not _fil_case(self, Wave, IGEOM):
            if self.FWHM != 0.0:
                _delegate("Measurement_0.conv outside the filter-function branch (FWHM > 0 spline, IGEOM='All')")
            return rconv(self, Wave, ModSpec, IGEOM=IGEOM, FWHMEXIST=FWHMEXIST)
        n = int(self.NCONV[IGEOM])
        return eng.conv_fil(Wave, np.asarray(ModSpec)[:len(Wave)], None, n, self.VCONV[:n, IGEOM], self.NFIL, self.VFIL, self.AFIL)

    def convg(self, Wave, ModSpec, ModGrad, IGEOM='All', FWHMEXIST=''):
        if not _fil_case(self, Wave, IGEOM):
            if self.FWHM != 0.0:
                _delegate("Measurement_0.convg outside the filter-function branch (FWHM > 0, IGEOM='All')")
            return rconvg(self, Wave, ModSpec, ModGrad, IGEOM=IGEOM, FWHMEXIST=FWHMEXIST)
        n = int(self.NCONV[IGEOM])
        return eng.conv_fil(Wave, np.asarray(ModSpec)[:len(Wave)], ModGrad, n, self.VCONV[:n, IGEOM], self.NFIL, self.VFIL,
                            self.AFIL)

    cls.conv, cls.convg = conv, convg
    # filter integrals (:4079-4300): np.trapz of filter x spectrum per convolution point
    if not hasattr(m0, "_ansfm_reference_intf"):
        m0._ansfm_reference_intf = (m0.integrate_filter, m0.integrate_filter_ngeom, m0.integrate_filterg, m0.integrate_filterg_ngeom)
    rint = m0._ansfm_reference_intf

    def _intf(k, nd, with_grad):
        def f(nwave, vwave, y, *rest):
            dydx = rest[0] if with_grad else None
            nconv, vconv, nfil, vfil, afil = rest[1:] if with_grad else rest
            if np.ndim(y) != nd or (with_grad and np.ndim(dydx) != nd + 1) or not _ascending(vwave):
                _delegate("integrate_filter* on an unsorted grid")
                return rint[k](nwave, vwave, y, *rest)
            return eng.integrate_filter(nwave, vwave, y, nconv, vconv, nfil, vfil, afil, dydx=dydx)
        return f

    m0.integrate_filter, m0.integrate_filter_ngeom = _intf(0, 1, False), _intf(1, 2, False)
    m0.integrate_filterg, m0.integrate_filterg_ngeom = _intf(2, 1, True), _intf(3, 2, True)
    m0.lblconv, m0.lblconvg, m0.lblconv_fil, m0.lblconvg_fil = lblconv, lblconvg, lblconv_fil, lblconvg_fil
    m0.lblconv_ngeom, m0.lblconvg_ngeom = lblconv_ngeom, lblconvg_ngeom
    m0.lblconv_fil_ngeom, m0.lblconvg_fil_ngeom = lblconv_fil_ngeom, lblconvg_fil_ngeom
    return lblconv, lblconvg, lblconv_fil, lblconvg_fil


def install_gpu_continuum(device=0):
    """Route ForwardModel_0.calc_tau_cia (ForwardModel_0.py:4516) -- per forward model ~2 scipy interp1d constructions per
    (layer, CIA pair) on the host --, calc_tau_rayleigh (:4869) and calc_tau_dust (:4790) through the GPU.  The wavenumber-only parametrisations co2cia / n2n2cia / n2h2cia
    (CIA_0.py:631-880, embedded data tables) are still evaluated by the reference's own functions and passed as vectors."""
    import importlib
    fm = importlib.import_module("archnemesis.ForwardModel_0")
    cm = importlib.import_module("archnemesis.CIA_0")
    eng = get_engine(device)
    cls = fm.ForwardModel_0
    ref = getattr(cls, "_ansfm_reference_calc_tau_cia", None) or cls.calc_tau_cia

    def calc_tau_cia(self, ISPACE=None, WAVEC=None, CIA=None, Atmosphere=None, Layer=None, MakePlot=False):
        if MakePlot:
            _delegate("calc_tau_cia(MakePlot=True)")
            return ref(self, ISPACE, WAVEC, CIA, Atmosphere, Layer, MakePlot)
        ISPACE = int(self.MeasurementX.ISPACE) if ISPACE is None else int(ISPACE)
        WAVEC = self.SpectroscopyX.WAVE if WAVEC is None else WAVEC
        CIA = self.CIAX if CIA is None else CIA
        A = self.AtmosphereX if Atmosphere is None else Atmosphere
        L = self.LayerX if Layer is None else Layer
        WAVEC = np.asarray(WAVEC, dtype=np.float64)
        WAVEN = WAVEC if ISPACE == 0 else np.sort(1.e4 / WAVEC)
        ID = np.asarray(A.ID); ISO = np.asarray(A.ISO)
        has = lambda gid: np.any((ID == gid))
        return eng.calc_tau_cia(ISPACE, WAVEC, CIA.WAVEN, CIA.TEMP, CIA.FRAC, int(CIA.NPARA), CIA.K_CIA,
                                [int(g) for g in CIA.IPAIRG1], [int(g) for g in CIA.IPAIRG2], [int(g) for g in CIA.INORMALT],
                                int(CIA.INORMAL), CIA.locate_INORMAL_pairs(), ID, ISO, L.PP, L.PRESS, L.TEMP, L.FRAC, L.TOTAM, L.DELH,
                                k_co2=cm.co2cia(WAVEN) if has(2) else None, k_n2n2=cm.n2n2cia(WAVEN) if has(22) else None,
                                k_n2h2=cm.n2h2cia(WAVEN) if (has(22) and has(39)) else None)

    cls._ansfm_reference_calc_tau_cia = ref
    cls.calc_tau_cia = calc_tau_cia

    # Rayleigh scattering (:4869) and aerosols (:4790): same pattern, the reference's methods keep their signatures
    ref_ray = getattr(cls, "_ansfm_reference_calc_tau_rayleigh", None) or cls.calc_tau_rayleigh
    ref_dust = getattr(cls, "_ansfm_reference_calc_tau_dust", None) or cls.calc_tau_dust

    def calc_tau_rayleigh(self, IRAY=None, ISPACE=None, WAVEC=None, ID=None, ISO=None, Layer=None, MakePlot=False):
        if MakePlot:
            _delegate("calc_tau_rayleigh(MakePlot=True)")
            return ref_ray(self, IRAY, ISPACE, WAVEC, ID, ISO, Layer, MakePlot)
        IRAY = int(self.ScatterX.IRAY if IRAY is None else IRAY)
        ISPACE = int(self.MeasurementX.ISPACE if ISPACE is None else ISPACE)
        WAVEC = self.SpectroscopyX.WAVE if WAVEC is None else WAVEC
        ID = self.AtmosphereX.ID if ID is None else ID
        ISO = self.AtmosphereX.ISO if ISO is None else ISO
        L = self.LayerX if Layer is None else Layer
        VMR = (np.asarray(L.PP).T / np.asarray(L.PRESS)).T if IRAY == 4 else None
        return eng.calc_tau_rayleigh(IRAY, ISPACE, WAVEC, L.TOTAM, ID, ISO, VMR)

    def calc_tau_dust(self, WAVEC=None, Scatter=None, Layer=None, MakePlot=False):
        WAVEC = self.SpectroscopyX.WAVE if WAVEC is None else WAVEC
        S = self.ScatterX if Scatter is None else Scatter
        L = self.LayerX if Layer is None else Layer
        WAVEC = np.asarray(WAVEC, dtype=np.float64)
        if self.Scatter.NDUST > 0:                                      # the reference's own range test (:4819-4823)
            if (WAVEC.min() < S.WAVE.min()) & (WAVEC.max() > S.WAVE.min()):
                raise ValueError('error calc_tau_dust :: Spectral range for calculation is outside of range in which the Aerosol properties are defined')
        for i in range(S.NDUST):                                        # side effect on Layer.CONT kept (:4833-4834)
            if i in self.AtmosphereX.DUST_RENORMALISATION.keys():
                L.CONT[:, i] = L.CONT[:, i] / L.CONT[:, i].sum() * 1e4 * self.AtmosphereX.DUST_RENORMALISATION[i]
        if S.NDUST == 0:
            z = np.zeros((len(WAVEC), L.NLAY, 0))
            return z, z.copy(), z.copy(), z.copy()
        return eng.calc_tau_dust(WAVEC, S.WAVE, np.asarray(S.KEXT)[:, :S.NDUST], np.asarray(S.KSCA)[:, :S.NDUST],
                                 np.asarray(L.CONT)[:, :S.NDUST])

    cls._ansfm_reference_calc_tau_rayleigh = ref_ray
    cls._ansfm_reference_calc_tau_dust = ref_dust
    cls.calc_tau_rayleigh = calc_tau_rayleigh
    cls.calc_tau_dust = calc_tau_dust
    return calc_tau_cia


def install_gpu_table_reader(device=0):
    """Spectroscopy_0.read_tables (Spectroscopy_0.py:1448) for binary k-tables (.kta) and LBL tables (.lta, ILBL = 2;
    read_lbltable :2626 loops over (wavenumber, pressure) in Python) without the host array: the header logic
    (read_header, the searchsorted cut of WAVE to [wavemin, wavemax], :1482-1494) is the reference's, but instead of
    unpacking every table with the Python loops of read_ktable (:2846-2850) into a float64 (NWAVE,NG,NP,NT,NGAS) array,
    Spectroscopy.K becomes a KtaTableOnDevice description and the GPU CIRSrad streams the files into HBM
    (ansfm_upload_ktable_files) -- once, since nemesisfm re-reads the tables for every forward model (:480-482) and the
    fingerprint of unchanged files matches.  Other table kinds go to the reference's read_tables."""
    import importlib
    sp = importlib.import_module("archnemesis.Spectroscopy_0")
    cls = sp.Spectroscopy_0
    ref = getattr(cls, "_ansfm_reference_read_tables", None) or cls.read_tables

    def read_tables(self, wavemin=0., wavemax=1.0e10, wavedelta=1.0):
        ext = {ILBL_K_TABLES: "kta", ILBL_LBL_TABLES: "lta"}.get(int(self.ILBL))
        binary = (ext is not None and self.LOCATION is not None and not getattr(self, "ONLINE", False)
                  and len(self.LOCATION) > 0 and all(str(p).endswith(ext) for p in self.LOCATION))
        if not binary:
            _delegate("read_tables for tables that are not binary .kta / .lta files")
            return ref(self, wavemin, wavemax, wavedelta)
        if self.WAVE is None:
            self.read_header()
        iwl = np.searchsorted(self.WAVE, wavemin, side='right') - 1      # :1486-1494
        if iwl < 0:
            iwl = 0
        iwh = np.searchsorted(self.WAVE, wavemax, side='left')
        if iwh >= self.NWAVE:
            iwh = self.NWAVE - 1
        wave1 = self.WAVE[iwl:iwh + 1]
        self.NWAVE = len(wave1)
        self.WAVE = wave1
        if ext == "lta":
            self.K = KtaTableOnDevice(self.LOCATION, self.WAVE.min(), self.WAVE.max(),       # NT < 0: |NT| temperatures per level
                                      (self.NWAVE, self.NP, abs(int(self.NT)), self.NGAS), sp.read_lbltable, ".lta", 8)
        else:
            self.K = KtaTableOnDevice(self.LOCATION, self.WAVE.min(), self.WAVE.max(),
                                      (self.NWAVE, self.NG, self.NP, self.NT, self.NGAS), sp.read_ktable)

    cls._ansfm_reference_read_tables = ref
    cls.read_tables = read_tables
    return read_tables


class Installed(list):
    """What install_all() returns: the names installed, and `summary()` -- every case that left the GPU path since then
    (DELEGATED: case -> count; each is also announced once as a RuntimeWarning) and every deliberate difference (NOTES)."""

    @staticmethod
    def summary():
        return summary()

    def __repr__(self):
        s = summary()
        return "Installed(%s; delegated to the reference so far: %s; notes: %s)" % (list.__repr__(self), s["delegated"] or "nothing",
                                                                                     s["notes"] or "none")


def install_all(device=0, oe_linalg=True, ktable_generator=True, forward_model=True):
    """Every replacement this package has for the imported reference, in one call; returns the
    names installed as an `Installed` list whose `summary()` reports what was handed to the reference's CPU code since.
    forward_model=True also makes `archnemesis.ForwardModel_0` name the GPU subclass (install_gpu_forward_model), so that
    `coreretOE` / `retrieval_nemesis` build it."""
    done = Installed()
    if forward_model:
        install_gpu_forward_model(device); done.append("install_gpu_forward_model")
    for f in (install_gpu_gradient_maps, install_gpu_scattering_core, install_gpu_line_kernel, install_gpu_layering,
              install_gpu_convolution, install_gpu_continuum, install_gpu_table_reader):
        f(device); done.append(f.__name__)
    if oe_linalg:
        from .oe_linalg import install_gpu_oe_linalg
        install_gpu_oe_linalg(device); done.append("install_gpu_oe_linalg")
    if ktable_generator:
        from .ktable_gen import install_gpu_ktable_generator
        install_gpu_ktable_generator(device); done.append("install_gpu_ktable_generator")
    return done
