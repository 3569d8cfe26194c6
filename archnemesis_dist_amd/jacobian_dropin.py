"""`ForwardModel_0.jacobian_nemesis` (ForwardModel_0.py:2184-2361) as a drop-in on one GPU per process.

The reference fans the nfm = NX_run + 1 forward models of the numerical part out to joblib / loky workers (:2322-2337);
every worker is a pickled copy of the whole ForwardModel_0, runs `nemesisfm()` per column (8 deep copies, `read_tables`,
`subprofretg`, `calc_path`, CIRSrad) and hands back a zero-padded (NY, nfm) matrix that the parent sums.  With CIRSrad
on a GPU that fan-out is the wrong shape: every worker would open its own context and upload its own k-table.  The
override keeps the method's signature, state-vector arithmetic (:2234-2242), NUM / FIX / ISCAT rules (:2251-2255,
:2291-2302) and quotient (:2348-2359), and replaces the fan-out by three in-process routes, tried in this order:

  "profile"  every variable is a continuous profile (Models/PreRTModels/model_0.py) or a scaling of one (model_2.py,
             model_3.py) of temperature or a gas, k-tables, thermal emission, radiance units: profile_dropin.py maps the
             reference's objects to ONE batched evaluation of all forward models -- hydrostatic re-adjustment, CIA / aerosol /
             Rayleigh continuum, several geometries and averaging points included (sharded over ranks when a process group is
             given).
  "staged"   any model of the reference's zoo, any continuum, several geometries / averaging points, hydrostatic
             re-adjustment, instrument convolution: the reference's OWN host code (select_Measurement, the deep copies,
             subprofretg, calc_path, the continuum routines) is run per state up to the point where nemesisfm would call
             CIRSrad (:490-516); what CIRSrad would read is kept, and the states of one (geometry, averaging point) go
             to the GPU as ONE batched call (layers bit-identical to the unperturbed state's are not recomputed,
             DESIGN.md 4.1d); the second half of nemesisfm (FOV weights, convolution, subspecret, :531-587) then runs
             per state.  read_tables runs once per geometry instead of once per forward model.
             With nemesisL = True the same is done for nemesisLfm (:1254-1368: all tangent paths of a state in one CIRSrad
             call, then the interpolation to the measurement's tangent heights and the convolution over all geometries), with
             nemesisSO = True for nemesisSOfm's plain branch (:909-978: calc_path_SO, CIRSrad's transmission branch).
             nemesisdisc = True: nemesisdiscfm (:1609-1716) is nemesisfm with process_IAV per averaging point: same route.
             nemesisC = True: nemesisCfm (:1526-1604; several viewing angles in one multiple-scattering call per state).
             nemesisPT = True: nemesisPTfm (:1838-1995; the transit depth from the transmission of the tangent paths).
  "loop"     everything else (AOTF orders in nemesisSOfm, line-by-line runtime, Telluric, a CIRSrad branch
             that has no batch axis): the reference's `execute_fm` (:2121) per column, in this process, one after the
             other -- never joblib workers on one GPU.  NCores > 1 is noted once (RuntimeWarning) and not used.

`ansfm_last_jacobian` records which route ran, the number of forward models and the layer rows computed / total."""
import warnings
from copy import deepcopy

import numpy as np

from . import forward_model as _fm
from .jacobian import chunk_range, perturbed_states

ISCAT_THERMAL_EMISSION = 0             # ScatteringCalculationModeEnum.THERMAL_EMISSION
IFORM_NORMALISED_RADIANCE = 5          # SpectraUnitEnum.Normalised_radiance
IFORM_INTEGRATED_RADIANCE = 6          # SpectraUnitEnum.Integrated_radiance


class JacobianGPU:
    """Mixin: jacobian_nemesis without the joblib fan-out.  Sits in front of CIRSradGPU and the reference class."""

    ansfm_jacobian_route = "auto"      # "auto" | "profile" | "staged" | "loop": force a route (tests, A/B timing)
    ansfm_jacobian_group = None        # (rank, world_size, process group): shard the forward models over ranks
    ansfm_last_jacobian = None

    # ---- the method the retrieval loop calls (OptimalEstimation_0.py:1333, :1467; Retrievals.py:182, :257) ----------
    def jacobian_nemesis(self, NCores=1, nemesisSO=False, nemesisL=False, nemesisC=False, nemesisdisc=False,
                         nemesisPT=False, analytical_gradient=True):
        V, M = self.Variables, self.Measurement
        flags = dict(nemesisSO=nemesisSO, nemesisL=nemesisL, nemesisC=nemesisC, nemesisdisc=nemesisdisc, nemesisPT=nemesisPT)
        V.calc_DSTEP()
        XN0 = np.array(V.XN, dtype=float)
        xnx = perturbed_states(XN0, V.DSTEP)                                       # :2234-2242
        if int(self.Scatter.ISCAT) != ISCAT_THERMAL_EMISSION or analytical_gradient is False:
            V.NUM[:] = 1                                                           # :2251-2255
        NY, NX = int(M.NY), int(V.NX)
        KK = np.zeros((NY, NX))
        YN = None
        info = dict(route=None, nfm=0, analytic_columns=0, rows=(0, 0))
        ian = np.where(np.asarray(V.NUM) == 0)[0]
        if len(ian) > 0:                                                           # analytic part: nemesisfmg (:2263-2285)
            SPECMOD, dSPECMOD = self.select_nemesis_fm(analytical_gradient=True, **flags)()
            YN = np.zeros(NY)
            ik = 0
            for ig in range(M.NGEOM):
                nc = int(M.NCONV[ig])
                YN[ik:ik + nc] = SPECMOD[0:nc, ig]
                KK[ik:ik + nc, :] = dSPECMOD[0:nc, ig, :]
                ik += nc
            info["analytic_columns"] = int(len(ian))
        inum = np.where((np.asarray(V.NUM) == 1) & (np.asarray(V.FIX) == 0))[0]     # :2291-2302
        lead = 0 if YN is not None else 1
        nfm = len(inum) + lead
        ixrun = np.zeros(nfm, dtype="int32")
        ixrun[lead:] = inum + 1
        info["nfm"] = int(nfm)
        if nfm > 0:
            try:
                YNtot = self._ansfm_forward_models(xnx, ixrun, flags, NCores, info)   # (NY, nfm)
            finally:
                V.XN = XN0                       # the reference mutates worker copies only (:2154)
            if YN is None:
                YN = YNtot[:, 0].copy()
            for i, ix in enumerate(inum):                                          # :2348-2359
                xn1 = XN0[ix] * 1.05
                if xn1 == 0.0:
                    xn1 = 0.05
                KK[:, ix] = (YNtot[:, i + lead] - YN) / (xn1 - XN0[ix])
        self.ansfm_last_jacobian = info
        return YN, KK

    # ---- route selection -------------------------------------------------------------------------------------------
    def _ansfm_forward_models(self, xnx, ixrun, flags, NCores, info):
        want = self.ansfm_jacobian_route
        plain = not any(flags.values())
        routes = []
        if plain and want in ("auto", "profile"):
            routes.append(("profile", self._ansfm_profile_route))
        if plain and want in ("auto", "staged"):
            routes.append(("staged", self._ansfm_staged_route))
        if bool(flags["nemesisdisc"]) and not any(v for k, v in flags.items() if k != "nemesisdisc") and want in ("auto", "staged"):
            routes.append(("staged", lambda xnx, ixrun, info: self._ansfm_staged_route(xnx, ixrun, info, disc=True)))
        for flag, kind in (("nemesisL", "L"), ("nemesisSO", "SO"), ("nemesisC", "C"), ("nemesisPT", "PT")):
            if bool(flags[flag]) and not any(v for k, v in flags.items() if k != flag) and want in ("auto", "staged"):
                routes.append(("staged", lambda xnx, ixrun, info, kind=kind: self._ansfm_staged_limb_route(xnx, ixrun, info, kind)))
        if want not in ("auto", "loop") and not routes:
            raise ValueError("jacobian_nemesis: route %r is not available for these flags" % (want,))
        for name, fn in routes:
            if want == name:                     # a forced route fails loudly
                Y = fn(xnx, ixrun, info)
            else:
                # "auto": a route that stumbles over something it did not foresee must not take the retrieval down -- the next
                # route is closer to the reference's own code, and the last one IS the reference's code, which raises
                # whatever the reference would have raised
                try:
                    Y = fn(xnx, ixrun, info)
                except Exception as exc:         # noqa: BLE001 -- recorded, announced once, and the next route takes over
                    info["error_" + name] = "%s: %s" % (type(exc).__name__, exc)
                    _fm._note("jacobian_nemesis: the %s route failed (%s: %s); falling back" % (name, type(exc).__name__, exc))
                    self.Variables.XN = xnx[:, 0].copy()
                    Y = None
            if Y is not None:
                info["route"] = name
                return Y
            if want == name:
                raise NotImplementedError("jacobian_nemesis: the %s route does not cover this configuration (%s)"
                                          % (name, info.get("why_not_" + name, "")))
        info["route"] = "loop"
        return self._ansfm_loop_route(xnx, ixrun, flags, NCores, info)

    # ---- "loop": execute_fm per column, in process -------------------------------------------------------------------
    def _ansfm_loop_route(self, xnx, ixrun, flags, NCores, info):
        if NCores is not None and int(NCores) > 1:
            _fm._note("jacobian_nemesis(NCores > 1): the forward models run one after the other in this process on the GPU, "
                      "no joblib workers")
        nfm = len(ixrun)
        Y = np.zeros((int(self.Measurement.NY), nfm))
        rank, world, group = self.ansfm_jacobian_group or (0, 1, None)
        s, e = chunk_range(nfm, world, rank)
        for ifm in range(s, e):
            Y = self.execute_fm((ifm, nfm, xnx, ixrun, flags["nemesisSO"], flags["nemesisL"], flags["nemesisC"],
                                 flags["nemesisdisc"], flags["nemesisPT"], Y, 1))
        if world > 1:
            Y = self._ansfm_gather(Y, s, e, nfm, rank, world, group)
        return Y

    def _ansfm_gather(self, Y, s, e, nfm, rank, world, group):
        """One all_gather of the (nfm_local, NY) blocks of a sharded run (RCCL when the group's backend is nccl)."""
        import torch
        from .jacobian import gather_columns
        dev = torch.device("cuda", self.ansfm_device) if (group is not None and torch.cuda.is_available()) else "cpu"
        block = torch.as_tensor(np.ascontiguousarray(Y[:, s:e].T), dtype=torch.float64, device=dev)
        return np.ascontiguousarray(gather_columns(block, nfm, rank, world, group=group).cpu().numpy().T)

    # ---- "staged": the reference's host code per state, one batched CIRSrad per (geometry, averaging point) ------------
    def _ansfm_staged_supported(self, info):
        A, Sf, M = self.Atmosphere, self.Surface, self.Measurement
        why = None
        if getattr(A, "NLOCATIONS", 1) > 1 or getattr(Sf, "NLOCATIONS", 1) > 1:
            why = "several locations"
        elif getattr(self, "Telluric", None) is not None:
            why = "Telluric transmission"
        elif getattr(self, "Emissions", None) is not None:
            why = "layer emissions"
        elif int(self.Spectroscopy.ILBL) not in (_fm.ILBL_K_TABLES, _fm.ILBL_LBL_TABLES) or self.Spectroscopy.NGAS <= 0:
            why = "spectral mode without tables"
        info["why_not_staged"] = why
        return why is None

    def _ansfm_stage_one(self, IGEOM, IAV):
        """nemesisfm between the table read and the CIRSrad call (:490-516) for the state in Variables.XN."""
        self.select_Measurement(IGEOM, IAV)
        for name in ("Atmosphere", "Scatter", "Stellar", "Surface", "Layer", "CIA", "Telluric"):
            setattr(self, name + "X", deepcopy(getattr(self, name)))
        MX, SX = self.MeasurementX, self.ScatterX
        if MX.EMISS_ANG[0, 0] >= 0.0:
            SX.SOL_ANG, SX.EMISS_ANG, SX.AZI_ANG = MX.SOL_ANG[0, 0], MX.EMISS_ANG[0, 0], MX.AZI_ANG[0, 0]
        else:
            SX.SOL_ANG, SX.EMISS_ANG = MX.TANHE[0, 0], MX.EMISS_ANG[0, 0]
        self.subprofretg()
        self.LayerX.DUST_UNITS_FLAG = self.AtmosphereX.DUST_UNITS_FLAG
        self.calc_path()

    def _ansfm_staged_route(self, xnx, ixrun, info, disc=False):
        """nemesisfm per state up to CIRSrad, one batched call per (geometry, averaging point), the rest per state.  disc = True
        (jacobian_nemesis(nemesisdisc=True)): nemesisdiscfm (:1609-1716) is the same forward model with the averaging points
        of a geometry run through process_IAV (:1998: the very stage nemesisfm has inline; joblib workers over the points in
        the reference, one batched call here), their weighted sum taken at once, and no filter-integral branch."""
        if not self._ansfm_staged_supported(info):
            return None
        M, V = self.Measurement, self.Variables
        nfm = len(ixrun)
        rank, world, group = self.ansfm_jacobian_group or (0, 1, None)
        s, e = chunk_range(nfm, world, rank)
        # the unperturbed state leads every batch although its spectrum may not be wanted (another rank's chunk, or YN came
        # from the analytic part): the engine shares layers with the FIRST model of a batch, and every perturbed state
        # is one step away from the unperturbed one, not from its neighbour
        owners = list(range(s, e))
        if not owners:
            return self._ansfm_gather(np.zeros((int(M.NY), nfm)), s, e, nfm, rank, world, group) if world > 1 else np.zeros((int(M.NY), nfm))
        cols = [int(ixrun[i]) for i in owners]
        if cols[0] != 0:
            cols, owners = [0] + cols, [None] + owners
        states = cols
        eng = _fm.get_engine(self.ansfm_device)
        self.check_gas_spec_atm()
        self.check_wave_range_consistency()
        SPECONV = np.zeros((len(states),) + tuple(M.MEAS.shape))
        rows_c = rows_t = 0
        for IGEOM in range(M.NGEOM):
            M.build_ils(IGEOM=IGEOM)                                                # :476-482, once per geometry
            wmin, wmax = M.calc_wave_range(apply_doppler=True, IGEOM=IGEOM)
            self.SpectroscopyX = deepcopy(self.Spectroscopy)
            if self.SpectroscopyX.NGAS > 0:
                self.SpectroscopyX.read_tables(wavemin=wmin, wavemax=wmax)
            W = int(self.SpectroscopyX.NWAVE)
            NAV = int(M.NAV[IGEOM])
            staged = [[None] * NAV for _ in states]
            for k, col in enumerate(states):
                V.XN = xnx[:, col]                                                  # execute_fm :2154
                for IAV in range(NAV):
                    self._ansfm_stage_one(IGEOM, IAV)
                    if int(self.PathX.NPATH) > 1:    # nemesisfm's several-paths branch (:521-527) fails on a shape mismatch
                        info["why_not_staged"] = "NPATH > 1"     # in the reference: let its own code say so ("loop")
                        return None
                    rec = self._ansfm_thermal_inputs()
                    if rec is None:
                        rec = self._ansfm_scatter_inputs()
                    if rec is None:                  # a CIRSrad branch without a batch axis: this state runs on its own
                        rec = dict(alone=self.CIRSrad())
                    staged[k][IAV] = rec
            SPEC = np.zeros((len(states), W))
            for IAV in range(NAV):
                recs = [staged[k][IAV] for k in range(len(states))]
                spectra, rc, rt = self._ansfm_run_batches(eng, recs, W)
                rows_c += rc; rows_t += rt
                for k, sp in enumerate(spectra):                                    # :531 (NAV >= 1 inside this loop)
                    SPEC[k] += M.WGEOM[IGEOM, IAV] * sp[:, 0]
            for k in range(len(states)):
                SPECONV[k, 0:int(M.NCONV[IGEOM]), IGEOM] = self._ansfm_convolve(SPEC[k], IGEOM, disc)
        Y = np.zeros((int(M.NY), nfm))
        for k, (col, ifm) in enumerate(zip(states, owners)):
            if ifm is None:
                continue
            V.XN = xnx[:, col]
            dS = np.zeros((int(M.NCONV.max()), int(M.NGEOM), int(V.NX)))
            SP, _ = self.subspecret(SPECONV[k], dS)                                 # :585-587
            ik = 0
            for ig in range(M.NGEOM):                                               # execute_fm :2171-2174
                nc = int(M.NCONV[ig])
                Y[ik:ik + nc, ifm] = SP[0:nc, ig]
                ik += nc
        info["rows"] = (int(rows_c), int(rows_t))
        if world > 1:
            Y = self._ansfm_gather(Y, s, e, nfm, rank, world, group)
        return Y

    # ---- "staged" for nemesisL: nemesisLfm's host code per state, ONE batched CIRSrad for all states ------------------------
    @staticmethod
    def _ansfm_limb_to_tangent_heights(SPECOUT, BASEH_TANHE, TANHE):
        """nemesisLfm :1322-1344: the spectra of the NPATH tangent paths (one per layer base, km) brought to the tangent
        heights of the measurement by linear weights between the two neighbouring paths.  The reference's arithmetic is kept as
        it stands: the nearest base is divided by 1e3 a second time before it is compared with the tangent height (:1328), the
        weights are (1 - fhl) and (1 - fhh), a lower neighbour of -1 wraps to the last path as a Python index does, and above
        the top path the lower neighbour's spectrum is taken."""
        NPATH = BASEH_TANHE.size
        out = np.zeros((SPECOUT.shape[0], len(TANHE)))
        for i in range(len(TANHE)):
            t = TANHE[i]
            near = int(np.argmin(np.abs(BASEH_TANHE - t)))
            lo, hi = (near, near + 1) if BASEH_TANHE[near] / 1.0e3 <= t else (near - 1, near)
            if hi > NPATH - 1:
                out[:, i] = SPECOUT[:, lo]
            else:
                span = BASEH_TANHE[hi] - BASEH_TANHE[lo]
                fhl, fhh = (t - BASEH_TANHE[lo]) / span, (BASEH_TANHE[hi] - t) / span
                out[:, i] = SPECOUT[:, lo] * (1. - fhl) + SPECOUT[:, hi] * (1. - fhh)
        return out

    @staticmethod
    def _ansfm_transit_depth(SPECOUT, base_km, r_star_km, r_planet_m, ngeom):
        """nemesisPTfm :1931-1950 (gradients = False): SPECOUT (NWAVE, NPATH) the transmission of the tangent paths, base_km
        their tangent heights.  Absorbing annuli (1 - T) 2 pi (h + R) integrated over height by trapezoids, path after path
        in the reference's order, plus the disc below the lowest tangent height, over the stellar disc, in per cent."""
        area_star = np.pi * ((r_star_km * 1.0e3) ** 2)
        area_disc = np.pi * ((r_planet_m + base_km[0] * 1.0e3) ** 2)
        out = np.zeros((SPECOUT.shape[0], ngeom))
        for i in range(len(base_km) - 1):
            lower = (1. - SPECOUT[:, i]) * 2. * np.pi * (base_km[i] * 1.0e3 + r_planet_m)
            upper = (1. - SPECOUT[:, i + 1]) * 2. * np.pi * (base_km[i + 1] * 1.0e3 + r_planet_m)
            out[:, 0] += 0.5 * (lower + upper) * ((base_km[i + 1] - base_km[i]) * 1.0e3)
        return (out + area_disc) / area_star * 100.

    def _ansfm_staged_limb_route(self, xnx, ixrun, info, kind="L"):
        """jacobian_nemesis(nemesisL=True): every forward model is nemesisLfm (:1254-1368) -- all tangent paths of a state in one
        CIRSrad call.  Its host code runs per state as the reference wrote it (deep copies, subprofretg with the hydrostatic
        re-adjustment off, calc_path_L), the CIRSrad calls of all states become one batched engine call (NPATH paths each), then
        the interpolation to the measurement's tangent heights, the convolution over all geometries and subspecret per state.
        kind = "SO" (nemesisSO=True): nemesisSOfm's plain branch (:909-978) has the same shape with calc_path_SO and CIRSrad's
        transmission branch; its AOTF branch (a forward model per diffraction order, :824-907) stays with the loop route.
        kind = "C" (nemesisC=True): nemesisCfm (:1526-1604) -- an instrument looking up or down at several viewing angles, one
        path per geometry in ONE multiple-scattering CIRSrad call per state: hydrostatic re-adjustment on, calc_path_C, then
        subspecret on the unconvolved spectra and convg / lblconv over all geometries.
        kind = "PT" (nemesisPT=True): nemesisPTfm (:1838-1995, gradients = False) -- a primary transit: calc_path_PT, the
        transmission of every tangent path in one CIRSrad call per state, then the absorbing area integrated over the tangent
        heights (trapezoids of (1 - T) 2 pi r), the planet's disc added, over the star's area, in per cent."""
        if not self._ansfm_staged_supported(info):
            return None
        if kind == "SO" and getattr(self.Measurement, "NORDERS_AOTF", None) is not None:
            info["why_not_staged"] = "AOTF diffraction orders"
            return None
        if kind == "PT" and (int(self.Measurement.IFORM) != _fm.IFORM_TRANSIT_DEPTH or int(self.Measurement.NGEOM) != 1):
            info["why_not_staged"] = "nemesisPTfm refuses this measurement"         # :1876-1880: let its own code say so
            return None
        M, V = self.Measurement, self.Variables
        nfm = len(ixrun)
        rank, world, group = self.ansfm_jacobian_group or (0, 1, None)
        s, e = chunk_range(nfm, world, rank)
        owners = list(range(s, e))
        if not owners:
            return self._ansfm_gather(np.zeros((int(M.NY), nfm)), s, e, nfm, rank, world, group) if world > 1 else np.zeros((int(M.NY), nfm))
        cols = [int(ixrun[i]) for i in owners]
        if cols[0] != 0:                           # the unperturbed state leads the batch (see _ansfm_staged_route)
            cols, owners = [0] + cols, [None] + owners
        eng = _fm.get_engine(self.ansfm_device)
        self.check_gas_spec_atm()                                                   # :1294-1295
        self.check_wave_range_consistency()
        M.build_ils(IGEOM=0)                                                        # :1298-1303, once: no state changes them
        wmin, wmax = M.calc_wave_range(apply_doppler=True, IGEOM=None)
        self.SpectroscopyX = deepcopy(self.Spectroscopy)
        if self.SpectroscopyX.NGAS > 0:
            self.SpectroscopyX.read_tables(wavemin=wmin, wavemax=wmax)
        W = int(self.SpectroscopyX.NWAVE)
        recs, kept = [], []
        for col in cols:
            V.XN = xnx[:, col]                                                      # execute_fm :2154
            self.Variables1 = deepcopy(self.Variables)                              # :1283-1291
            for name in ("Measurement", "Atmosphere", "Scatter", "Stellar", "Surface", "Layer", "CIA"):
                setattr(self, name + "X", deepcopy(getattr(self, name)))
            self.adjust_hydrostat = kind in ("C", "PT")                             # :1306 / :1576 / :1893
            self.subprofretg()
            base_km = None
            if kind == "C":
                MX, SX = self.MeasurementX, self.ScatterX                           # :1582-1587
                SX.SOL_ANG, SX.EMISS_ANG, SX.AZI_ANG = MX.SOL_ANG[0, 0], MX.EMISS_ANG[0, 0], MX.AZI_ANG[0, 0]
                self.calc_path_C()
            else:
                self.LayerX.DUST_UNITS_FLAG = self.AtmosphereX.DUST_UNITS_FLAG
                if kind == "SO":
                    self.calc_path_SO()
                elif kind == "PT":
                    self.calc_path_PT()
                else:
                    self.calc_path_L()
                P, L = self.PathX, self.LayerX
                base_km = np.array([L.BASEH[P.LAYINC[int(P.NLAYIN[i] / 2), i]] / 1.0e3 for i in range(int(P.NPATH))])    # :1314-1316
            rec = self._ansfm_transmission_inputs()
            if rec is None:
                rec = self._ansfm_thermal_inputs()
            if rec is None:
                rec = self._ansfm_scatter_inputs()
            if rec is None:
                rec = dict(alone=self.CIRSrad())
            recs.append(rec)
            if kind == "PT":                       # the radii the transit depth is formed with are the state's (:1931-1932)
                base_km = (base_km, float(self.StellarX.RADIUS), float(self.AtmosphereX.RADIUS))
            kept.append((self.MeasurementX, base_km))
        spectra, rows_c, rows_t = self._ansfm_run_batches(eng, recs, W)
        Y = np.zeros((int(M.NY), nfm))
        S = self.SpectroscopyX
        for k, (col, ifm) in enumerate(zip(cols, owners)):
            if ifm is None:
                continue
            MX, base_km = kept[k]
            self.MeasurementX = MX
            V.XN = xnx[:, col]
            if kind == "C":                                                         # :1592-1602
                SPECOUT = np.asarray(spectra[k]).reshape(W, -1)
                dS = np.zeros((W, int(MX.NGEOM), int(V.NX)))
                SPECOUT, dS = self.subspecret(SPECOUT, dS)
                if int(S.ILBL) == _fm.ILBL_K_TABLES:
                    SP, _ = MX.convg(S.WAVE, SPECOUT, dS, IGEOM='All')
                else:
                    SP = MX.lblconv(S.WAVE, SPECOUT, IGEOM='All')
                ik = 0
                for ig in range(M.NGEOM):                                           # execute_fm :2171-2174
                    nc = int(M.NCONV[ig])
                    Y[ik:ik + nc, ifm] = SP[0:nc, ig]
                    ik += nc
                continue
            if kind == "PT":
                SPECMOD = self._ansfm_transit_depth(np.asarray(spectra[k]).reshape(W, -1), *base_km, int(MX.NGEOM))
            else:
                SPECMOD = self._ansfm_limb_to_tangent_heights(np.asarray(spectra[k]).reshape(W, -1), base_km,
                                                              [MX.TANHE[i] for i in range(int(MX.NGEOM))])
            if kind == "L" and int(MX.IFORM) == IFORM_INTEGRATED_RADIANCE:          # :1348-1361 (nemesisSOfm has no filter branch)
                SPECONV = MX.integrate_filter(S.WAVE, SPECMOD, IGEOM='All')
            elif int(S.ILBL) == _fm.ILBL_K_TABLES:
                SPECONV = MX.conv(S.WAVE, SPECMOD, IGEOM='All')
            else:
                SPECONV = MX.lblconv(S.WAVE, SPECMOD, IGEOM='All')
            dS = np.zeros((int(MX.NCONV.max()), int(MX.NGEOM), int(V.NX)))
            SP, _ = self.subspecret(SPECONV, dS)                                    # :1366
            ik = 0
            for ig in range(M.NGEOM):                                               # execute_fm :2171-2174
                nc = int(M.NCONV[ig])
                Y[ik:ik + nc, ifm] = SP[0:nc, ig]
                ik += nc
        info["rows"] = (int(rows_c), int(rows_t))
        if world > 1:
            Y = self._ansfm_gather(Y, s, e, nfm, rank, world, group)
        return Y

    def _ansfm_convolve(self, SPEC, IGEOM, disc=False):
        """nemesisfm :556-581 for one geometry: filter integral, ILS convolution by table kind, normalisation.  disc: as
        nemesisdiscfm does it (:1692-1708: no filter-integral branch)."""
        import os
        M, S = self.Measurement, self.SpectroscopyX
        nc = int(M.NCONV[IGEOM])
        if not disc and int(M.IFORM) == IFORM_INTEGRATED_RADIANCE:
            return np.asarray(M.integrate_filter(S.WAVE, SPEC, IGEOM=IGEOM))[0:nc]
        if int(S.ILBL) == _fm.ILBL_K_TABLES:
            fw = self.runname if os.path.exists(self.runname + ".fwh") else ""
            out = M.conv(S.WAVE, SPEC, IGEOM=IGEOM, FWHMEXIST=fw)
        else:
            out = M.lblconv(S.WAVE, SPEC, IGEOM=IGEOM)
        out = np.array(out[0:nc], dtype=float)
        if int(M.IFORM) == IFORM_NORMALISED_RADIANCE:
            out /= np.interp(M.VNORM, M.VCONV[0:nc, IGEOM], out)
        return out

    def _ansfm_thermal_inputs(self):
        """What CIRSrad's thermal-emission branch hands to the engine (forward_model.CIRSradGPU.CIRSrad), kept instead
        of run; None when the staged state takes another branch of CIRSrad's dispatch."""
        if not self._ansfm_supported(False):
            return None
        P = self.PathX
        imod = int(np.unique(np.asarray(P.IMOD).astype(int))[0])
        if self._ansfm_transmission_branch(imod) or not (imod & _fm.IMOD_THERMAL_EMISSION):
            return None
        L = self.LayerX
        TAUCIA, TAUDUST, TAURAY, _ = self._ansfm_continuum(False)
        xfac, emissivity = self._ansfm_units_and_surface()
        NPATH = int(P.NPATH) if hasattr(P, "NPATH") else np.asarray(P.LAYINC).shape[1]
        return dict(
            ISPACE=int(self.MeasurementX.ISPACE), lp=np.array(L.PRESS, dtype=np.float64), lt=np.array(L.TEMP, dtype=np.float64),
            f_gas=self._ansfm_layer_inputs(), taucont=TAUCIA + TAUDUST + TAURAY,
            NLAYIN=np.asarray(P.NLAYIN, dtype=np.int32).reshape(NPATH), LAYINC=np.asarray(P.LAYINC, dtype=np.int32).reshape(-1, NPATH),
            SCALE=np.asarray(P.SCALE, dtype=np.float64).reshape(-1, NPATH), EMTEMP=np.asarray(P.EMTEMP, dtype=np.float64).reshape(-1, NPATH),
            TSURF=float(self.SurfaceX.TSURF), emissivity=emissivity, xfac=xfac,
            SOL_ANG=np.asarray(P.SOL_ANG, dtype=np.float64).reshape(NPATH), EMISS_ANG=np.asarray(P.EMISS_ANG, dtype=np.float64).reshape(NPATH))

    def _ansfm_transmission_inputs(self):
        """What CIRSrad's pure-transmission branch (:4478-4483, calculate_transmission_spectrum :4110) hands to the engine, kept
        instead of run; None when the staged state takes another branch."""
        if not self._ansfm_supported(False):
            return None
        P, L, S = self.PathX, self.LayerX, self.SpectroscopyX
        imod = int(np.unique(np.asarray(P.IMOD).astype(int))[0])
        if not self._ansfm_transmission_branch(imod):
            return None
        TAUCIA, TAUDUST, TAURAY, _ = self._ansfm_continuum(False)
        xf = None
        if int(self.MeasurementX.IFORM) == _fm.IFORM_ATMOSPHERIC_TRANSMISSION:     # :4119-4127: times the solar flux
            import scipy.interpolate
            self.StellarX.calc_solar_flux()
            xf = scipy.interpolate.interp1d(self.StellarX.WAVE, self.StellarX.SOLFLUX)(S.WAVE)
        NPATH = np.asarray(P.LAYINC).shape[1]
        return dict(transmission=True, lp=np.array(L.PRESS, dtype=np.float64), lt=np.array(L.TEMP, dtype=np.float64),
                    f_gas=self._ansfm_layer_inputs(), taucont=TAUCIA + TAUDUST + TAURAY,
                    NLAYIN=np.asarray(P.NLAYIN, dtype=np.int32).reshape(NPATH), LAYINC=np.asarray(P.LAYINC, dtype=np.int32).reshape(-1, NPATH),
                    SCALE=np.asarray(P.SCALE, dtype=np.float64).reshape(-1, NPATH), xfac=xf)

    @staticmethod
    def _ansfm_transmission_key(r):
        opt = lambda a: b"-" if a is None else np.ascontiguousarray(a, dtype=np.float64).tobytes()
        return ("transmission", r["lp"].shape, r["LAYINC"].shape, r["NLAYIN"].tobytes(), r["LAYINC"].tobytes(), opt(r["xfac"]))

    def _ansfm_scatter_inputs(self):
        """What CIRSrad's multiple-scattering branch hands to the engine (CIRSradGPU._ansfm_cirsrad_scatter), kept instead of
        run; None when the staged state is not on that branch."""
        if not self._ansfm_supported(False):
            return None
        imod = int(np.unique(np.asarray(self.PathX.IMOD).astype(int))[0])
        if not self._ansfm_scatter_branch(imod):
            return None
        TAUCIA, TAUDUST, TAURAY, _ = self._ansfm_continuum(False)
        rec = self._ansfm_cirsrad_scatter(None, TAUCIA, TAUDUST, TAURAY, self._ansfm_layer_inputs())
        rec["scatter"] = True
        return rec

    @staticmethod
    def _ansfm_scatter_key(r):
        """States that may share one batched scattering call: everything without a model axis in ansfm_cirsrad_ck_scatter_batch"""
        b = lambda a: np.ascontiguousarray(a, dtype=np.float64).tobytes()
        return ("scatter", r["ISPACE"], r["lp"].shape, r["PHASE"].shape, b(r["PHASE"]), b(r["SOL_ANG"]), b(r["EMISS_ANG"]),
                b(r["AZI_ANG"]), b(r["solar"]), r["LOWBC"], b(r["BRDF"]), b(r["MU"]), b(r["WTMU"]), r["NF"], r["NPHI"], r["IRAY"],
                r["IMIE"])

    @staticmethod
    def _ansfm_batch_key(r):
        """States that may share one batched call: same path structure and same per-wavenumber boundary vectors."""
        opt = lambda a: b"-" if a is None else np.ascontiguousarray(a, dtype=np.float64).tobytes()
        return (r["ISPACE"], r["lp"].shape, r["LAYINC"].shape, r["NLAYIN"].tobytes(), r["LAYINC"].tobytes(), opt(r["emissivity"]),
                opt(r["xfac"]), r["SOL_ANG"].tobytes(), r["EMISS_ANG"].tobytes())

    def _ansfm_run_batches(self, eng, recs, W):
        """recs: one staged record per state -> list of SPECOUT (NWAVE, NPATH) per state, rows computed, rows total."""
        out = [None] * len(recs)
        groups = {}
        for k, r in enumerate(recs):
            if "alone" in r:
                out[k] = np.asarray(r["alone"]).reshape(W, -1)
            elif r.get("scatter"):
                groups.setdefault(self._ansfm_scatter_key(r), []).append(k)
            elif r.get("transmission"):
                groups.setdefault(self._ansfm_transmission_key(r), []).append(k)
            else:
                groups.setdefault(self._ansfm_batch_key(r), []).append(k)
        self._ansfm_upload_table(eng)
        rc = rt = 0
        for key, ks in groups.items():
            r0 = recs[ks[0]]
            st = lambda name: np.stack([recs[k][name] for k in ks])
            if key[0] == "scatter":
                # the NX + 1 multiple-scattering forward models the reference runs when ISCAT != THERMAL_EMISSION (:2251-2252)
                spec = eng.cirsrad_ck_scatter_batch(r0["ISPACE"], st("lp"), st("lt"), st("f_gas"), st("TAUCIA"), st("TAUDUST"),
                                                    st("TAURAY"), st("TAUSCAT"), r0["PHASE"], st("FRAC"), st("RADGROUND"), r0["SOL_ANG"],
                                                    r0["EMISS_ANG"], r0["AZI_ANG"], r0["solar"], r0["LOWBC"], r0["BRDF"], r0["MU"],
                                                    r0["WTMU"], r0["NF"], r0["NPHI"], r0["IRAY"], r0["IMIE"])
                spec = np.asarray(spec).reshape(len(ks), W, -1)
                if hasattr(eng, "last_scatter_cache"):
                    a, b = eng.last_scatter_cache()           # layers of models 1.. taken from model 0's doublings / all of them
                    nlay = int(r0["lp"].shape[0])
                    rc += (b - a) + nlay; rt += b + nlay
                for j, k in enumerate(ks):
                    out[k] = spec[j]
                continue
            if key[0] == "transmission":
                spec = eng.cirsrad_ck_transmission(st("lp"), st("lt"), st("f_gas"), st("taucont"), r0["NLAYIN"], r0["LAYINC"],
                                                   st("SCALE"), xfac=r0["xfac"])
                spec = np.asarray(spec).reshape(len(ks), W, -1)
                if hasattr(eng, "last_layer_rows"):
                    a, b = eng.last_layer_rows()
                    rc += a; rt += b
                for j, k in enumerate(ks):
                    out[k] = spec[j]
                continue
            spec = eng.cirsrad_ck_thermal(r0["ISPACE"], st("lp"), st("lt"), st("f_gas"), st("taucont"), r0["NLAYIN"], r0["LAYINC"],
                                          st("SCALE"), st("EMTEMP"), np.array([recs[k]["TSURF"] for k in ks]),
                                          EMISSIVITY=r0["emissivity"], SOL_ANG=r0["SOL_ANG"], EMISS_ANG=r0["EMISS_ANG"], xfac=r0["xfac"])
            spec = np.asarray(spec).reshape(len(ks), W, -1)
            if hasattr(eng, "last_layer_rows"):
                a, b = eng.last_layer_rows()
                rc += a; rt += b
            for j, k in enumerate(ks):
                out[k] = spec[j]
        return out, rc, rt

    # ---- "profile": model-0 state vectors through the vectorised host path --------------------------------------------
    def _ansfm_profile_route(self, xnx, ixrun, info):
        from .profile_dropin import batched_model_from_reference
        model, why = batched_model_from_reference(self)
        info["why_not_profile"] = why
        if model is None:
            return None
        import torch
        rank, world, group = self.ansfm_jacobian_group or (0, 1, None)
        nfm = len(ixrun)
        s, e = chunk_range(nfm, world, rank)
        cols = [int(c) for c in ixrun[s:e]]
        lead = 0 if (not cols or cols[0] == 0) else 1          # the unperturbed state leads every batch (de-duplication)
        X = np.ascontiguousarray(xnx[:, [0] * lead + cols].T)
        Yd = model.spectra_batch(X)[lead:]                        # torch (nfm_local, NWAVE * NPATH) on the device
        info["rows"] = tuple(int(v) for v in model.last_rows)
        Yd = model.measurement_vector(Yd)                         # calculation grid -> NY (interp / ILS), still on the device
        if world > 1:
            from .jacobian import gather_columns
            Yd = gather_columns(Yd, nfm, rank, world, group=group)
        return np.ascontiguousarray(Yd.cpu().numpy().T) if isinstance(Yd, torch.Tensor) else np.ascontiguousarray(np.asarray(Yd).T)
