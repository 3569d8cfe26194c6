"""Reference-built forward model -> one batched GPU evaluation of many state vectors (the "profile" route of
jacobian_dropin.JacobianGPU).

`batched_model_from_reference(fm)` looks at a ForwardModel_0 as `coreretOE` builds it (OptimalEstimation_0.py:1318) and,
when every variable is a continuous profile (Models/PreRTModels/model_0.py) of temperature or of a gas mixing ratio, or a
scaling factor of one (model_2.py, model_3.py),
returns an object that does for n state vectors at once what `nemesisfm` (ForwardModel_0.py:437-589) does for one:

    subprofretg (:2397-2560)   hydrostatic re-adjustment of the heights (Atmosphere_0.adjust_hydrostatH :1027 with
                               calc_grav :858), restated over a state axis; model 0 = the profile itself (exp where the
                               element is carried as a logarithm)                                   host, O(n NP)
    calc_path (:2948-3065)     layer_split per state (the grid follows the heights), ONE k_layer_average launch for all
                               states, AtmCalc_0's ray geometry per state                           layering.py + GPU
    calculate_layer_opacity    Rayleigh for all states in HBM; CIA and aerosol opacities through the engine's kernels
                               with the layers of all states laid end to end (layers are independent there)
    CIRSrad (:4376)            ONE batched call, layers equal to the unperturbed state's not recomputed
    conv (:2288)               FWHM = 0: linear interpolation onto VCONV on the device; otherwise Measurement.conv per state
    NGEOM / NAV                one pass per (geometry, averaging point), packed into NY like execute_fm (:2171-2174)

Anything else (other models, AMFORM = 1 rescaling, Telluric, emissions, scattering, line-by-line tables, FluxRatio units,
several locations) makes `batched_model_from_reference` return (None, reason) and the caller takes the staged route.
tests/test_jacobian_dropin.py holds this route to the staged one (same reference objects) and to the reference fixture."""
import numpy as np

from . import layering
from . import forward_model as _fm

R_GAS = 8.31446261815324               # Data/constants.py: R
G_NEWTON = 6.67199976e-11              # Data/constants.py: G
ISCAT_THERMAL_EMISSION = 0
IFORM_RADIANCE = 0
AMFORM_SCALE_VMR = 1                   # AtmosphericProfileFormatEnum.CALC_MOLECULAR_WEIGHT_SCALE_VMR_TO_ONE
PLANCK_AT_BIN_CENTRE = 8192            # PathCalcEnum.PLANCK_FUNCTION_AT_BIN_CENTRE


class GravityField:
    """Atmosphere_0.calc_grav (:858-930; Lindal et al. 1986) split into what depends on the latitude only (computed once,
    with the same scipy Legendre polynomials) and what depends on the height: g(H) for arrays of any shape."""

    def __init__(self, mass, rotation_days, flatten, Jcoeff, radius_km, latitude_deg):
        from scipy.special import legendre
        xgm = mass * G_NEWTON * 1.0e6
        omega = 2. * np.pi / (rotation_days * 24. * 3600.)
        ellip = 1.0 / (1.0 - flatten)
        xc = (Jcoeff[0] / 1.0e3, Jcoeff[1] / 1.0e6, Jcoeff[2] / 1.0e8)
        self.xradius = radius_km * 1.0e5
        lat = 2 * np.pi * latitude_deg / 360.
        latc = np.arctan(np.tan(lat) / ellip ** 2.)
        s, c = np.sin(latc), np.cos(latc)
        self.Rr = np.sqrt(c ** 2 + (ellip ** 2. * s ** 2.))
        pol = [float(np.ravel(legendre(i + 1)([s]))[0]) for i in range(6)]
        g = 1.0
        gt = 0.0
        for i in range(3):
            ix = i + 1
            g = g - ((2 * ix + 1) * self.Rr ** (2 * ix) * xc[ix - 1] * pol[2 * ix - 1])
            gt = gt - (4. * ix ** 2 * self.Rr ** (2 * ix) * xc[ix - 1] * (pol[2 * ix - 1 - 1] - s * pol[2 * ix - 1]) / c)
        self.a_r, self.b_r = g * xgm, omega ** 2. * c ** 2.            # gradial = a_r / r^2 - r b_r
        self.a_t, self.b_t = gt * xgm, omega ** 2 * c * s              # gtheta  = a_t / r^2 + r b_t

    @classmethod
    def of(cls, atm):
        atm.calc_grav()                 # fills PLANET_* from the planet table when IPLANET > 0 (:868-874)
        return cls(atm.PLANET_MASS, atm.PLANET_ROTATION, atm.PLANET_FLATTEN, atm.PLANET_J, atm.PLANET_RADIUS, float(atm.LATITUDE))

    def __call__(self, H):
        r = (self.xradius + np.asarray(H, float) * 1.0e2) / self.Rr
        gr = (self.a_r / r ** 2.) - (r * self.b_r)
        gt = (self.a_t / r ** 2) + (r * self.b_t)
        return np.sqrt(gr ** 2. + gt ** 2.) * 0.01


def adjust_hydrostat_heights(H, P, T, MOLWT, grav):
    """Atmosphere_0.adjust_hydrostatH (:1027-1090) for n profiles at once: H, T, MOLWT (n, NP), P (NP,) -> H (n, NP).
    Heights are rebuilt from the level nearest z = 0 by the hydrostatic equation with the mean scale height of adjacent
    levels, repeated while the depth of the atmosphere changes by more than 1 %; every profile stops on its own."""
    H = np.array(H, dtype=float)
    n, NP = H.shape
    T = np.broadcast_to(np.asarray(T, float), (n, NP)); MOLWT = np.broadcast_to(np.asarray(MOLWT, float), (n, NP))
    lnp = np.log(P[1:] / P[:-1])                                   # log(p[i] / p[i-1])
    lnm = np.log(P[:-1] / P[1:])                                   # log(p[i] / p[i+1])
    ialt = np.argmin(np.abs(H - 0.0), axis=1)
    live = np.ones(n, bool)
    while live.any():
        k = np.nonzero(live)[0]
        Hk = H[k]
        depth = Hk[:, -1] - Hk[:, 0]
        scale = R_GAS * T[k] / (MOLWT[k] * grav(Hk))
        h = Hk.copy()
        ia = ialt[k]
        rows = np.arange(k.size)
        pin = (ia > 0) & (ia < NP - 1)
        h[rows[pin], ia[pin]] = 0.0
        for i in range(1, NP):                                     # upwards from each profile's own reference level
            m = i > ia
            if m.any():
                sh = 0.5 * (scale[m, i - 1] + scale[m, i])
                h[m, i] = h[m, i - 1] - sh * lnp[i - 1]
        for i in range(NP - 2, -1, -1):                            # and downwards
            m = i < ia
            if m.any():
                sh = 0.5 * (scale[m, i + 1] + scale[m, i])
                h[m, i] = h[m, i + 1] - sh * lnm[i]
        xdepth = 100. * np.abs(((h[:, -1] - h[:, 0]) - depth) / depth)
        H[k] = h
        live[k[~(xdepth > 1)]] = False
    return H


def batched_model_from_reference(fm):
    """-> (ReferenceProfileBatch, None) or (None, why not)."""
    V, A, M, S, Sc = fm.Variables, fm.Atmosphere, fm.Measurement, fm.Spectroscopy, fm.Scatter
    if getattr(A, "NLOCATIONS", 1) > 1 or getattr(fm.Surface, "NLOCATIONS", 1) > 1:
        return None, "several locations"
    if getattr(fm, "Telluric", None) is not None or getattr(fm, "Emissions", None) is not None:
        return None, "Telluric / layer emissions"
    if int(S.ILBL) != _fm.ILBL_K_TABLES or S.NGAS <= 0:
        return None, "not a k-table run"
    if int(Sc.ISCAT) != ISCAT_THERMAL_EMISSION or int(M.IFORM) != IFORM_RADIANCE:
        return None, "scattering or units other than radiance"
    if int(getattr(V, "JPRE", -1)) != -1 or int(getattr(V, "JTAN", -1)) != -1:
        return None, "pressure / tangent-height retrieval"
    if int(A.AMFORM) == AMFORM_SCALE_VMR:
        return None, "AMFORM = 1 rescales the other gases with every state"
    blocks = []
    ix = 0
    for ivar, mdl in enumerate(V.models):
        vid = np.asarray(V.VARIDENT).reshape(-1, 3)[ivar]
        mid, nent = int(getattr(mdl, "id", -999)), int(mdl.n_state_vector_entries)
        # model 0: the profile itself, one element per level (model_0.py); models 2 / 3: ONE element that scales the profile
        # of the reference atmosphere, carried as it is / as its logarithm (model_2.py, model_3.py: `profile *= scf`)
        if mid == 0 and int(vid[2]) == 0 and nent == A.NP:
            how = "profile"
        elif mid in (2, 3) and int(vid[2]) == mid and nent == 1:
            how = "scale"
        else:
            return None, "a variable that is neither a continuous profile nor a scaling of one (model %s)" % vid[2]
        if mdl.state_vector_start != ix:
            return None, "state vector not laid out variable after variable"
        if vid[0] == 0:
            blocks.append(("T", None, how, ix, nent))
        elif vid[0] > 0:
            j = np.nonzero((np.asarray(A.ID) == vid[0]) & (np.asarray(A.ISO) == vid[1]))[0]
            if len(j) != 1:
                return None, "gas of a variable not found once in the atmosphere"
            blocks.append(("VMR", int(j[0]), how, ix, nent))
        else:
            return None, "aerosol / para-H2 / cloud-fraction profile"
        ix += nent
    if ix != V.NX:
        return None, "state vector longer than its models"
    try:
        return ReferenceProfileBatch(fm, blocks), None
    except NotImplementedError as e:
        return None, str(e)


class ReferenceProfileBatch:
    def __init__(self, fm, blocks):
        import copy
        self.fm, self.blocks = fm, blocks
        A, L = fm.Atmosphere, fm.Layer
        self.eng = _fm.get_engine(fm.ansfm_device)
        self.NP, self.NVMR = int(A.NP), int(A.NVMR)
        self.LX = np.asarray(fm.Variables.LX).astype(int)
        self.P = np.array(A.P, float); self.T0 = np.array(A.T, float); self.VMR0 = np.array(A.VMR, float)
        self.MOLWT = np.array(A.MOLWT, float)
        self.H0 = np.array(A.H, float)
        self.hydro = bool(fm.adjust_hydrostat)
        if self.hydro:
            a = copy.deepcopy(A)
            self.grav = GravityField.of(a)
            # subprofretg's first adjustment (:2440-2444) acts on the unperturbed copy: the same for every state
            self.H1 = adjust_hydrostat_heights(self.H0[None], self.P, self.T0[None], self.MOLWT[None], self.grav)[0]
        else:
            self.H1 = self.H0
        self.last_rows = (0, 0)
        self._lay = dict(RADIUS=float(L.RADIUS), NLAY=int(L.NLAY), LAYTYP=int(L.LAYTYP), LAYINT=int(L.LAYINT), NINT=int(L.NINT),
                         LAYHT=float(L.LAYHT), H_base=getattr(L, "H_base", None), P_base=getattr(L, "P_base", None))
        self._pending = None

    # ---- subprofretg for n states -----------------------------------------------------------------------------------
    def profiles(self, X):
        X = np.atleast_2d(np.asarray(X, float))
        n = X.shape[0]
        vals = np.where(self.LX[None, :] > 0, np.exp(X), X)         # ModelBase.get_parameter_values_from_state_vector
        T = np.repeat(self.T0[None], n, 0)
        VMR = np.repeat(self.VMR0[None], n, 0)
        for kind, j, how, ix, nent in self.blocks:        # in the order of the variables, like subprofretg's loop (:2500-2510)
            xb = vals[:, ix:ix + nent]
            if kind == "T":
                T = xb.copy() if how == "profile" else T * xb
            elif how == "profile":
                VMR[:, :, j] = xb
            else:
                VMR[:, :, j] = VMR[:, :, j] * xb
        H = np.repeat(self.H1[None], n, 0)
        if self.hydro:                                               # second adjustment, with the state's temperatures
            H = adjust_hydrostat_heights(H, self.P, T, self.MOLWT[None], self.grav)
        return H, T, VMR

    # ---- calc_path + continuum + CIRSrad for one (geometry, averaging point) -------------------------------------------
    def _geometry(self, IGEOM, IAV):
        M = self.fm.Measurement
        emi = float(M.EMISS_ANG[IGEOM, IAV])
        if emi >= 0.0:
            return dict(pointing=layering.NADIR, ANGLE=emi, EMISS_ANG=emi, LAYANG=0.0, LAYHT=self._lay["LAYHT"])
        return dict(pointing=layering.LIMB, ANGLE=90.0, EMISS_ANG=emi, LAYANG=90.0, LAYHT=float(M.TANHE[IGEOM, IAV]) * 1.0e3)

    def _spectra(self, H, T, VMR, IGEOM, IAV, WAVE):
        """torch (n, NWAVE) on the engine's device"""
        import torch
        fm, eng, la = self.fm, self.eng, self._lay
        A, Sc, Su = fm.Atmosphere, fm.Scatter, fm.Surface
        n = H.shape[0]
        ge = self._geometry(IGEOM, IAV)
        split = lambda h: layering.layer_split(la["RADIUS"], h, self.P, LAYANG=ge["LAYANG"], LAYHT=ge["LAYHT"], NLAY=la["NLAY"],
                                               LAYTYP=la["LAYTYP"], H_base=la["H_base"], P_base=la["P_base"])
        if self.hydro:
            BASEH = np.stack([split(H[i])[0] for i in range(n)])
        else:
            BASEH = np.repeat(split(H[0])[0][None], n, 0)
        DUST = None if A.DUST is None or np.size(A.DUST) == 0 else np.repeat(np.asarray(A.DUST, float)[None], n, 0)
        PARAH2 = None if getattr(A, "PARAH2", None) is None else np.repeat(np.asarray(A.PARAH2, float)[None], n, 0)
        out = eng.layer_average(la["RADIUS"], H, np.repeat(self.P[None], n, 0), T, A.ID, VMR, DUST, PARAH2, BASEH, None,
                                LAYANG=ge["LAYANG"], LAYINT=la["LAYINT"], LAYHT=ge["LAYHT"], NINT=la["NINT"],
                                DUST_UNITS=A.DUST_UNITS_FLAG, XMOLWT=self.MOLWT)
        names = ("HEIGHT", "PRESS", "TEMP", "TOTAM", "AMOUNT", "PP", "CONT", "FRAC", "DELH", "BASET", "LAYSF")
        lay = dict(zip(names, out))
        L = lay["PRESS"].shape[1]
        paths = [layering.calc_path(la["RADIUS"], BASEH[i], lay["DELH"][i], lay["TEMP"][i], float(H[i, -1]), pointing=ge["pointing"],
                                    BOTLAY=0, ANGLE=ge["ANGLE"], EMISS_ANG=ge["EMISS_ANG"], IPZEN=layering.IPZEN_BOTTOM,
                                    path_calc=PLANCK_AT_BIN_CENTRE | layering.THERMAL_EMISSION) for i in range(n if self.hydro else 1)]
        p0 = paths[0]
        if p0.NPATH != 1 or any(not np.array_equal(p.LAYINC, p0.LAYINC) for p in paths[1:]):
            raise NotImplementedError("the ray crosses different layers in different states")
        if self.hydro:
            SCALE = np.stack([p.SCALE for p in paths]); EMTEMP = np.stack([p.EMTEMP for p in paths])
        else:
            SCALE = np.repeat(p0.SCALE[None], n, 0)
            EMTEMP = layering.calc_path(la["RADIUS"], BASEH[0], lay["DELH"][0], lay["TEMP"], float(H[0, -1]), pointing=ge["pointing"],
                                        BOTLAY=0, ANGLE=ge["ANGLE"], EMISS_ANG=ge["EMISS_ANG"], IPZEN=layering.IPZEN_BOTTOM).EMTEMP
        S = fm.SpectroscopyX
        igas = np.array([A.locate_gas(S.ID[i], S.ISO[i]) for i in range(S.NGAS)], dtype=np.int64)
        amount = np.ascontiguousarray(np.transpose(lay["AMOUNT"][:, :, igas], (0, 2, 1))) * _fm.SQ_CM_TO_SQ_METER
        dev = getattr(eng, "torch_device", None) or torch.device("cuda", eng.device)
        td = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        W = WAVE.size
        ISPACE = int(fm.Measurement.ISPACE)
        cont = self._continuum(lay, n, L, WAVE, ISPACE, dev)
        emis = None
        if Su.TSURF > 0.0:
            import scipy.interpolate
            emis = td(scipy.interpolate.interp1d(Su.VEM, Su.EMISSIVITY)(WAVE))
        outt = torch.empty((n, W, 1), dtype=torch.float64, device=dev)
        args = (td(lay["PRESS"]), td(lay["TEMP"]), td(amount), cont, 1, p0.LAYINC.shape[0], td(p0.NLAYIN, torch.int32),
                td(p0.LAYINC, torch.int32), td(SCALE), td(EMTEMP), td(np.full(n, float(Su.TSURF))))
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        eng.cirsrad_ck_thermal_dev(ISPACE, n, L, *args, emis, None, None, None, None, None, outt)
        a, b = eng.last_layer_rows()
        self.last_rows = (self.last_rows[0] + a, self.last_rows[1] + b)
        eng.synchronize()
        return outt.reshape(n, W)

    def _continuum(self, lay, n, L, WAVE, ISPACE, dev):
        """TAUCIA + TAUDUST + TAURAY of every state, (n, NWAVE, NLAY) on the device (the sum order of :3989 per element)."""
        import importlib
        import torch
        fm, eng = self.fm, self.eng
        A, Sc, CIA = fm.Atmosphere, fm.Scatter, fm.CIA
        W = WAVE.size
        flat = lambda a: np.ascontiguousarray(a).reshape((n * L,) + a.shape[2:])
        parts = []
        if CIA is not None:
            cm = importlib.import_module("archnemesis.CIA_0")
            WAVEN = WAVE if ISPACE == 0 else np.sort(1.e4 / WAVE)
            ID = np.asarray(A.ID)
            has = lambda gid: np.any(ID == gid)
            tau = eng.calc_tau_cia(ISPACE, WAVE, CIA.WAVEN, CIA.TEMP, CIA.FRAC, int(CIA.NPARA), CIA.K_CIA, [int(g) for g in CIA.IPAIRG1],
                                   [int(g) for g in CIA.IPAIRG2], [int(g) for g in CIA.INORMALT], int(CIA.INORMAL),
                                   CIA.locate_INORMAL_pairs(), ID, np.asarray(A.ISO), flat(lay["PP"]), flat(lay["PRESS"]),
                                   flat(lay["TEMP"]), flat(lay["FRAC"]), flat(lay["TOTAM"]), flat(lay["DELH"]),
                                   k_co2=cm.co2cia(WAVEN) if has(2) else None, k_n2n2=cm.n2n2cia(WAVEN) if has(22) else None,
                                   k_n2h2=cm.n2h2cia(WAVEN) if (has(22) and has(39)) else None, with_grad=False)
            parts.append(np.asarray(tau).reshape(W, n, L))
        if int(Sc.NDUST) > 0:
            ND = int(Sc.NDUST)
            CONT = np.array(lay["CONT"][:, :, :ND], float)
            for i in range(ND):                                      # calc_tau_dust's renormalisation of Layer.CONT (:4833)
                if i in A.DUST_RENORMALISATION.keys():
                    CONT[:, :, i] = CONT[:, :, i] / CONT[:, :, i].sum(axis=1, keepdims=True) * 1e4 * A.DUST_RENORMALISATION[i]
            td1 = eng.calc_tau_dust(WAVE, Sc.WAVE, np.asarray(Sc.KEXT)[:, :ND], np.asarray(Sc.KSCA)[:, :ND], flat(CONT))[0]
            td1 = np.clip(np.nan_to_num(td1), 0, 1e20)               # :3966
            parts.append(np.sum(td1, 2).reshape(W, n, L))
        cont = None
        if parts:
            host = parts[0] if len(parts) == 1 else parts[0] + parts[1]           # TAUCIA + TAUDUST
            cont = torch.as_tensor(np.ascontiguousarray(np.transpose(host, (1, 0, 2))), dtype=torch.float64, device=dev)
        if int(Sc.IRAY) != 0:
            ray = torch.empty((n, W, L), dtype=torch.float64, device=dev)
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()
            eng.calc_tau_rayleigh_batch_dev(int(Sc.IRAY), ISPACE, lay["TOTAM"], ray, ID=A.ID, ISO=A.ISO,
                                            VMR=lay["PP"] / lay["PRESS"][:, :, None])
            eng.synchronize()
            cont = ray if cont is None else cont + ray                             # (TAUCIA + TAUDUST) + TAURAY
        return cont

    # ---- nemesisfm for n states ---------------------------------------------------------------------------------------
    def spectra_batch(self, X):
        """Measurement vectors of the states X (n, NX) are formed in two steps like the staged route: this one returns the
        spectra on each geometry's calculation grid, `measurement_vector` maps them to NY."""
        import torch
        from copy import deepcopy
        fm = self.fm
        M = fm.Measurement
        fm.check_gas_spec_atm()
        fm.check_wave_range_consistency()
        H, T, VMR = self.profiles(X)
        self.last_rows = (0, 0)
        per_geom = []
        for IGEOM in range(M.NGEOM):
            M.build_ils(IGEOM=IGEOM)
            wmin, wmax = M.calc_wave_range(apply_doppler=True, IGEOM=IGEOM)
            fm.SpectroscopyX = deepcopy(fm.Spectroscopy)
            fm.SpectroscopyX.read_tables(wavemin=wmin, wavemax=wmax)
            fm._ansfm_upload_table(self.eng)
            WAVE = np.asarray(fm.SpectroscopyX.WAVE, dtype=np.float64)
            spec = None
            for IAV in range(int(M.NAV[IGEOM])):
                s = self._spectra(H, T, VMR, IGEOM, IAV, WAVE) * float(M.WGEOM[IGEOM, IAV])      # :531
                spec = s if spec is None else spec + s
            per_geom.append((WAVE, spec))
        self._pending = per_geom
        return torch.cat([s for _, s in per_geom], dim=1)

    def measurement_vector(self, Y):
        """(n, sum of NWAVE) -> (n, NY): conv of every geometry (:556-581) packed like execute_fm (:2171-2174)."""
        import os
        import torch
        fm = self.fm
        M = fm.Measurement
        out, off = [], 0
        for IGEOM, (WAVE, _) in enumerate(self._pending):
            W = WAVE.size
            spec = Y[:, off:off + W]; off += W
            nc = int(M.NCONV[IGEOM])
            if float(M.FWHM) == 0.0:          # conv's channel-integrator branch (Measurement_0.py:2330-2336 / :2388): interp1d
                v = np.asarray(M.VCONV[0:nc, IGEOM], dtype=np.float64)
                if v.min() < WAVE[0] or v.max() > WAVE[-1]:
                    raise ValueError("A value in x_new is outside the interpolation range.")       # scipy's, bounds_error
                hi = np.clip(np.searchsorted(WAVE, v, side="left"), 1, W - 1)
                lo = hi - 1
                dx = torch.as_tensor(WAVE[hi] - WAVE[lo], dtype=Y.dtype, device=Y.device)
                off_x = torch.as_tensor(v - WAVE[lo], dtype=Y.dtype, device=Y.device)
                lo_t = torch.as_tensor(lo, device=Y.device); hi_t = torch.as_tensor(hi, device=Y.device)
                ylo, yhi = spec[:, lo_t], spec[:, hi_t]
                out.append((yhi - ylo) / dx[None, :] * off_x[None, :] + ylo)        # scipy's _call_linear: slope * (x - x_lo) + y_lo
            else:
                fw = fm.runname if os.path.exists(fm.runname + ".fwh") else ""
                host = spec.cpu().numpy()
                conv = np.stack([np.asarray(M.conv(WAVE, host[i], IGEOM=IGEOM, FWHMEXIST=fw))[0:nc] for i in range(host.shape[0])])
                out.append(torch.as_tensor(conv, dtype=Y.dtype, device=Y.device))
        return torch.cat(out, dim=1)
