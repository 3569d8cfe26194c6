"""State vector <-> atmosphere <-> spectra for a batch of states: what `ForwardModel_0.nemesisfm` does between
`Variables.XN` and `SPECONV` (ForwardModel_0.py:437-589) for the configuration of BASELINE configs[2],
evaluated for n states at once so that a numerical Jacobian is ONE batched call per GPU instead of nfm calls of
nemesisfm (the reference fans them out to joblib workers, :2305-2337).

    state vector  --subprofretg / model 0-->  T(level), VMR(level, gas)          host, a few kB per state
                  --calc_path: layer_average (Curtis-Godson) + ray geometry-->   k_layer_average (one launch, all states)
                  --calculate_layer_opacity: Rayleigh continuum-->               k_tau_rayleigh, stays in HBM
                  --CIRSrad: calc_k + k_overlap + thermal emission-->            k_ck_overlap + k_thermal_rt, layers that
                                                                                  equal the first state's are not recomputed
Supported here: ILBL = K_TABLES, thermal emission, nadir / limb rays through AtmCalc_0's geometry, continuous-profile
variables (Models/PreRTModels/model_0.py:35: temperature, ln volume mixing ratio), no hydrostatic re-adjustment
(`adjust_hydrostat = False`), no instrument convolution (FWHM = 0 on the calculation grid).  Anything else keeps using
the per-column route of jacobian.jacobian_nemesis_sharded.
"""
import numpy as np

from . import layering

SQ_CM_TO_SQ_METER = 1.0e-4       # ForwardModel_0.py:66


class ContinuousProfileState:
    """`Variables_0` restricted to model 0 blocks: every element of the state vector is the value of one atmospheric
    profile at one level -- T itself for temperature, ln(value) for a volume mixing ratio
    (model_0.py from_apr_to_state_vector: x0 = ref for temperature, log(ref) otherwise; subprofretg un-logs it,
    ForwardModel_0.py:2397-2560).

    H (m), P (Pa), T (K): (NPRO,); VMR (NPRO, NVMR); blocks: sequence of "T" or ("VMR", column)."""

    def __init__(self, H, P, T, VMR, blocks):
        self.H = np.asarray(H, float); self.P = np.asarray(P, float); self.T = np.asarray(T, float)
        self.VMR = np.asarray(VMR, float)
        self.NPRO = self.H.size
        self.blocks = [("T", None) if b == "T" else ("VMR", int(b[1])) for b in blocks]
        self.NX = self.NPRO * len(self.blocks)
        xs = [self.T if kind == "T" else np.log(self.VMR[:, j]) for kind, j in self.blocks]
        self.XN = np.concatenate(xs) if xs else np.zeros(0)
        self.NUM = np.ones(self.NX, dtype=np.int32)      # jacobian_nemesis treats every element numerically here
        self.FIX = np.zeros(self.NX, dtype=np.int32)
        self.DSTEP = None

    def calc_DSTEP(self):
        """Variables_0.calc_DSTEP (:535): 5 % of each element."""
        self.DSTEP = 0.05 * self.XN
        return self.DSTEP

    def profiles(self, X):
        """X (n, NX) -> T (n, NPRO), VMR (n, NPRO, NVMR): the model-0 part of subprofretg for n states."""
        X = np.atleast_2d(np.asarray(X, float))
        n = X.shape[0]
        T = np.repeat(self.T[None], n, 0)
        VMR = np.repeat(self.VMR[None], n, 0)
        for b, (kind, j) in enumerate(self.blocks):
            xb = X[:, b * self.NPRO:(b + 1) * self.NPRO]
            if kind == "T":
                T = xb.copy()
            else:
                VMR[:, :, j] = np.exp(xb)
        return T, VMR


class BatchedCKThermalModel:
    """nemesisfm for n states at once (see the module docstring).

    eng: AnsfmEngine with the k-table uploaded (its wavenumber grid is the calculation grid).
    state: ContinuousProfileState.  ID / ISO: gas identifiers of the VMR columns; igas_map (S,): VMR column of every
    spectroscopic gas of the table (AtmosphereX.locate_gas, ForwardModel_0.py:3860).
    layering: dict(NLAY, LAYTYP, LAYINT, LAYHT, LAYANG, NINT) of Layer_0; geometry: dict(pointing, BOTLAY, ANGLE,
    EMISS_ANG, IPZEN) of AtmCalc_0.  IRAY as in calc_tau_rayleigh (0 = none); extra_continuum (W, L) is added to every
    state (e.g. a fixed aerosol opacity)."""

    def __init__(self, eng, state, RADIUS, ID, ISO, igas_map, layering_args=None, geometry=None, ISPACE=0, IRAY=0,
                 TSURF=-1.0, extra_continuum=None, DUST=None, PARAH2=None):
        self.eng, self.state = eng, state
        self.RADIUS = float(RADIUS)
        self.ID = np.asarray(ID); self.ISO = np.asarray(ISO)
        self.igas_map = np.asarray(igas_map, dtype=np.int64)
        la = dict(NLAY=20, LAYTYP=layering.EQUAL_LOG_PRESSURE, LAYINT=1, LAYHT=0.0, LAYANG=0.0, NINT=101)
        la.update(layering_args or {})
        self.lay = la
        ge = dict(pointing=layering.NADIR, BOTLAY=0, ANGLE=0.0, EMISS_ANG=0.0, IPZEN=layering.IPZEN_BOTTOM)
        ge.update(geometry or {})
        self.geo = ge
        self.ISPACE, self.IRAY, self.TSURF = int(ISPACE), int(IRAY), float(TSURF)
        self.extra = None if extra_continuum is None else np.asarray(extra_continuum, float)
        self.DUST, self.PARAH2 = DUST, PARAH2
        dims, _ = eng.ktable_info()
        self.W = int(dims[0])
        # the layer grid is a property of the pressure-height profile, which no supported variable changes
        self.BASEH, self.BASEP = layering.layer_split(self.RADIUS, state.H, state.P, LAYANG=la["LAYANG"], LAYHT=la["LAYHT"],
                                                      NLAY=la["NLAY"], LAYTYP=la["LAYTYP"])
        self.last_rows = (0, 0)

    # ---- what the sharded Jacobian asks of a model -----------------------------------------------------------------
    def ny(self):
        """length of a measurement vector of this model (its wavenumbers x paths)"""
        return self.W * (layering.calc_path(self.RADIUS, self.BASEH, np.ones_like(self.BASEH), np.ones_like(self.BASEH),
                                            float(self.state.H[-1]), **{k: self.geo[k] for k in ("pointing", "BOTLAY", "ANGLE",
                                                                                                  "EMISS_ANG", "IPZEN")}).NPATH)

    def torch_device(self):
        import torch
        return torch.device("cuda", self.eng.device)

    def ny_local_all(self, world_size):
        """wavenumber-sharded mode: NY of every rank when the table of `global_waves` wavenumbers is split with
        jacobian.chunk_range (this model holds one of the parts)"""
        from .jacobian import chunk_range
        Wg = int(getattr(self, "global_waves", None) or 0)
        if Wg <= 0:
            raise ValueError("wavenumber-sharded mode: set model.global_waves to the length of the whole spectral grid")
        P_ = self.ny() // self.W
        return [(e - s) * P_ for s, e in (chunk_range(Wg, world_size, r) for r in range(world_size))]

    # ---- host part: subprofretg + calc_path for n states -------------------------------------------------------
    def layers(self, X):
        """X (n, NX) -> dict of per-state layer arrays (Layer_0 attributes after calc_layering) and the ray path."""
        st, la = self.state, self.lay
        T, VMR = st.profiles(X)
        n = T.shape[0]
        H = np.repeat(st.H[None], n, 0); P = np.repeat(st.P[None], n, 0)
        out = self.eng.layer_average(self.RADIUS, H, P, T, self.ID, VMR, self.DUST, self.PARAH2, self.BASEH, self.BASEP,
                                     LAYANG=la["LAYANG"], LAYINT=la["LAYINT"], LAYHT=la["LAYHT"], NINT=la["NINT"])
        names = ("HEIGHT", "PRESS", "TEMP", "TOTAM", "AMOUNT", "PP", "CONT", "FRAC", "DELH", "BASET", "LAYSF")
        lay = dict(zip(names, out))
        ge = self.geo
        lay["path"] = layering.calc_path(self.RADIUS, self.BASEH, lay["DELH"][0], lay["TEMP"], float(st.H[-1]),
                                         pointing=ge["pointing"], BOTLAY=ge["BOTLAY"], ANGLE=ge["ANGLE"],
                                         EMISS_ANG=ge["EMISS_ANG"], IPZEN=ge["IPZEN"])
        # amount of every spectroscopic gas in molecule cm-2 (ForwardModel_0.py:3861)
        lay["amount"] = np.ascontiguousarray(np.transpose(lay["AMOUNT"][:, :, self.igas_map], (0, 2, 1))) * SQ_CM_TO_SQ_METER
        lay["VMRLAY"] = lay["PP"] / lay["PRESS"][:, :, None]          # Layer_0: PP / PRESS is the layer's mixing ratio
        return lay

    # ---- the same with the layers left in HBM ----------------------------------------------------------------------
    device_layers = True      # False: `layers` (host arrays) feeds spectra_batch, as in rounds 1-2; same numbers bit for bit
    fused_rayleigh = True     # False: the Rayleigh continuum of all n * NLAY layers as an array of its own (same bits)

    def layers_dev(self, X, dev, order=lambda: None, drain=lambda: None):
        """`layers` without the round trip over PCIe: the profiles of the n states are put together on the device (the
        elements of the state vector are the only thing uploaded: exp() of a log-VMR block is taken on the host, as
        `profiles` takes it), k_layer_average leaves the layers in HBM, the arrays CIRSrad and the Rayleigh kernel want are
        views / gathers of them.  The ray geometry depends on the layer grid only and stays on the host.  -> dict of device
        tensors (PRESS, TEMP, TOTAM, amount, VMRLAY, EMTEMP) and "path".  order / drain: called before / after the engine's
        kernel when torch and the engine do not share a stream."""
        import torch
        st, la, ge = self.state, self.lay, self.geo
        X = np.atleast_2d(np.asarray(X, float))
        n = X.shape[0]
        td = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        # what the cached device constants were made from: the state's arrays (by identity and a checksum -- they are not
        # meant to be edited in place), the layering, the geometry, the surface temperature
        key = (n, str(dev), id(st), float(st.H.sum()), float(st.P.sum()), float(st.T.sum()), float(st.VMR.sum()), self.TSURF,
               tuple(sorted((k, float(v)) for k, v in la.items())), tuple(sorted((k, float(v)) for k, v in ge.items())),
               float(self.BASEH.sum()), tuple(int(i) for i in self.igas_map))
        c = getattr(self, "_dev_consts", None)
        if c is None or c["key"] != key:
            rep = lambda a: td(a)[None].expand((n,) + np.shape(a)).contiguous()
            DELH0 = np.append(self.BASEH[1:] - self.BASEH[:-1], st.H[-1] - self.BASEH[-1])      # k_layer_average's DELH
            path = layering.calc_path(self.RADIUS, self.BASEH, DELH0, np.zeros(self.BASEH.size), float(st.H[-1]),
                                      pointing=ge["pointing"], BOTLAY=ge["BOTLAY"], ANGLE=ge["ANGLE"],
                                      EMISS_ANG=ge["EMISS_ANG"], IPZEN=ge["IPZEN"])
            LIMAX = path.LAYINC.shape[0]
            inside = np.arange(LIMAX)[:, None] < path.NLAYIN[None, :]
            c = dict(key=key, H=rep(st.H), P=rep(st.P), T0=rep(st.T), VMR0=td(st.VMR), BASEH=rep(self.BASEH), path=path,
                     LAYINC=td(path.LAYINC, torch.int64), inside=td(inside, torch.bool), igas=td(self.igas_map, torch.int64),
                     NLAYIN32=td(path.NLAYIN, torch.int32), LAYINC32=td(path.LAYINC, torch.int32),
                     SCALE=rep(path.SCALE), TSURF=td(np.full(n, self.TSURF)))
            self._dev_consts = c
        T, VMR = c["T0"], None
        for b, (kind, j) in enumerate(st.blocks):
            xb = X[:, b * st.NPRO:(b + 1) * st.NPRO]
            if kind == "T":
                T = td(xb)
            else:
                if VMR is None:
                    VMR = c["VMR0"][None].repeat(n, 1, 1)
                VMR[:, :, j] = td(np.exp(xb))
        if VMR is None:
            VMR = c["VMR0"][None].repeat(n, 1, 1)
        order()
        lay = self.eng.layer_average_dev(self.RADIUS, c["H"], c["P"], T, VMR, None, None, c["BASEH"], LAYANG=la["LAYANG"],
                                         LAYINT=la["LAYINT"], LAYHT=la["LAYHT"], NINT=la["NINT"])
        drain()
        out = dict(PRESS=lay["PRESS"], TEMP=lay["TEMP"], TOTAM=lay["TOTAM"], path=c["path"], consts=c)
        out["amount"] = (lay["AMOUNT"][:, :, c["igas"]].transpose(1, 2) * SQ_CM_TO_SQ_METER).contiguous()
        out["VMRLAY"] = lay["PP"] / lay["PRESS"][:, :, None]
        out["EMTEMP"] = torch.where(c["inside"][None], lay["TEMP"][:, c["LAYINC"]], torch.zeros((), dtype=torch.float64, device=dev))
        return out

    # ---- device part ------------------------------------------------------------------------------------------
    def spectra_batch(self, X, device=None):
        """Spectra of the n states X (n, NX): torch tensor (n, NY), NY = NWAVE * NPATH (path fastest like SPECOUT),
        resident on the engine's device."""
        import torch
        eng = self.eng
        dev = torch.device("cuda", eng.device) if device is None else device
        if self.device_layers and self.DUST is None and self.PARAH2 is None:
            return self._spectra_batch_dev(X, dev)
        lay = self.layers(X)
        n, L = lay["PRESS"].shape
        td = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        path = lay["path"]
        P_, LIMAX = path.NPATH, path.LAYINC.shape[0]
        # torch and the engine share these buffers: on one stream they are ordered by construction; otherwise (the engine on
        # its own stream, e.g. because torch's current stream is the default one, whose handle is null) what torch has queued
        # is waited for before the engine starts and the engine is drained before torch sees the result
        cur = torch.cuda.current_stream(dev)
        shared = cur.cuda_stream != 0 and eng.stream_ptr == cur.cuda_stream
        order = (lambda: None) if shared else cur.synchronize
        cont = None
        if self.IRAY != 0 or self.extra is not None:
            if self.IRAY != 0:       # the kernel assigns every element
                cont = torch.empty((n, self.W, L), dtype=torch.float64, device=dev)
                order()
                eng.calc_tau_rayleigh_batch_dev(self.IRAY, self.ISPACE, lay["TOTAM"], cont, ID=self.ID, ISO=self.ISO,
                                                VMR=lay["VMRLAY"])
                if not shared:
                    eng.synchronize()
            else:
                cont = torch.zeros((n, self.W, L), dtype=torch.float64, device=dev)
            if self.extra is not None:
                cont += td(self.extra)[None]
        out = torch.empty((n, self.W, P_), dtype=torch.float64, device=dev)
        scale = np.repeat(path.SCALE[None], n, 0)
        args = (td(lay["PRESS"]), td(lay["TEMP"]), td(lay["amount"]), cont, P_, LIMAX, td(path.NLAYIN, torch.int32),
                td(path.LAYINC, torch.int32), td(scale), td(path.EMTEMP), td(np.full(n, self.TSURF)))
        order()
        eng.cirsrad_ck_thermal_dev(self.ISPACE, n, L, *args, None, None, None, None, None, None, out)
        self.last_rows = eng.last_layer_rows()
        if not shared:
            eng.synchronize()
        return out.reshape(n, self.W * P_)

    def _spectra_batch_dev(self, X, dev):
        import torch
        eng = self.eng
        cur = torch.cuda.current_stream(dev)
        shared = cur.cuda_stream != 0 and eng.stream_ptr == cur.cuda_stream
        order = (lambda: None) if shared else cur.synchronize      # see spectra_batch
        drain = (lambda: None) if shared else eng.synchronize
        lay = self.layers_dev(X, dev, order, drain)
        c, path = lay["consts"], lay["path"]
        n, L = lay["PRESS"].shape
        P_, LIMAX = path.NPATH, path.LAYINC.shape[0]
        out = torch.empty((n, self.W, P_), dtype=torch.float64, device=dev)
        if self.IRAY != 0 and self.extra is None and self.fused_rayleigh:
            # the Rayleigh continuum is formed inside the call, once per distinct layer of the batch
            f4 = eng.rayleigh_f4(self.ID, self.ISO, lay["VMRLAY"]) if self.IRAY == 4 else None
            args = (lay["PRESS"].contiguous(), lay["TEMP"].contiguous(), lay["amount"], self.IRAY, lay["TOTAM"].contiguous(), f4, P_,
                    LIMAX, c["NLAYIN32"], c["LAYINC32"], c["SCALE"], lay["EMTEMP"].contiguous(), c["TSURF"])
            order()
            eng.cirsrad_ck_thermal_ray_dev(self.ISPACE, n, L, *args, None, None, None, None, None, None, out)
            self.last_rows = eng.last_layer_rows()
            drain()
            return out.reshape(n, self.W * P_)
        cont = None
        if self.IRAY != 0 or self.extra is not None:
            if self.IRAY != 0:
                cont = torch.empty((n, self.W, L), dtype=torch.float64, device=dev)
                totam, vmrlay = lay["TOTAM"].contiguous(), lay["VMRLAY"]
                order()
                eng.calc_tau_rayleigh_batch_dev(self.IRAY, self.ISPACE, totam, cont, ID=self.ID, ISO=self.ISO, VMR=vmrlay)
                drain()
            else:
                cont = torch.zeros((n, self.W, L), dtype=torch.float64, device=dev)
            if self.extra is not None:
                cont += torch.as_tensor(self.extra, dtype=torch.float64, device=dev)[None]
        args = (lay["PRESS"].contiguous(), lay["TEMP"].contiguous(), lay["amount"], cont, P_, LIMAX, c["NLAYIN32"], c["LAYINC32"],
                c["SCALE"], lay["EMTEMP"].contiguous(), c["TSURF"])
        order()
        eng.cirsrad_ck_thermal_dev(self.ISPACE, n, L, *args, None, None, None, None, None, None, out)
        self.last_rows = eng.last_layer_rows()
        drain()
        return out.reshape(n, self.W * P_)

    # ---- analytic route: nemesisfmg for the same configuration ----------------------------------------------------
    def jacobian_analytic(self, X0=None):
        """YN (NY,), KK (NY, NX) at the state X0 (default: the state's own XN) by analytic gradients -- what
        `nemesisfmg` does (ForwardModel_0.py:593-779): `layer_averageg` (layer properties + DTE / DAM / DCO),
        `CIRSrad(return_grad=True)` (k_ck_overlapg + k_thermal_rtg), `map2pro` (layers -> levels, :705) and `map2xvec`
        (levels -> state vector, :711; for model 0 the map is 1 for a temperature element and the mixing ratio itself for
        a ln(VMR) element, model_0.py), the two maps chained on the device.  Like the reference's, the temperature
        columns are first order in DTE at fixed amounts (a level temperature enters through the layer temperatures as
        `layer_averageg` linearises them, not through the number density) and, for TSURF <= 0, leave out the ground term's
        dependence on the bottom layer's temperature (calc_thermal_emission_spectrumg :6484-6494 differentiates it with
        respect to TSURF only) -- a finite difference through `layer_average` sees both.  The Rayleigh continuum enters the
        gradients the reference's way (calculate_layer_opacity :3952-3957: dTAURAY, the cross section per unit column, is
        added to EVERY gas's parameter; its dependence on the composition, IRAY = 4, is not differentiated)."""
        eng, st, la, ge = self.eng, self.state, self.lay, self.geo
        X0 = st.XN if X0 is None else np.asarray(X0, float)
        T, VMR = st.profiles(X0[None])
        T, VMR = T[0], VMR[0]
        out = eng.layer_averageg(self.RADIUS, st.H, st.P, T, self.ID, VMR, self.DUST, self.PARAH2, self.BASEH, self.BASEP,
                                 LAYANG=la["LAYANG"], LAYINT=la["LAYINT"], LAYHT=la["LAYHT"], NINT=la["NINT"])
        names = ("HEIGHT", "PRESS", "TEMP", "TOTAM", "AMOUNT", "PP", "CONT", "FRAC", "DELH", "BASET", "LAYSF", "DTE", "DAM",
                 "DCO", "DPH")
        lay = dict(zip(names, out))
        self.last_analytic_layers = lay              # layer properties and the DTE / DAM / DCO matrices of this call
        path = layering.calc_path(self.RADIUS, self.BASEH, lay["DELH"], lay["TEMP"], float(st.H[-1]), pointing=ge["pointing"],
                                  BOTLAY=ge["BOTLAY"], ANGLE=ge["ANGLE"], EMISS_ANG=ge["EMISS_ANG"], IPZEN=ge["IPZEN"])
        amount = np.ascontiguousarray(lay["AMOUNT"][:, self.igas_map].T) * SQ_CM_TO_SQ_METER
        NPRO, NVMR = VMR.shape
        NDUST = lay["CONT"].shape[1]
        NPAR = NVMR + 2 + NDUST
        P_ = path.NPATH
        cont, dcont = self.extra, None
        if self.IRAY != 0:
            tauray, dtauray = eng.calc_tau_rayleigh(self.IRAY, self.ISPACE, eng.WAVE, lay["TOTAM"], ID=self.ID, ISO=self.ISO,
                                                    VMR=lay["PP"] / lay["PRESS"][:, None])
            cont = tauray if cont is None else cont + tauray
            dcont = dtauray                          # (W, L): goes to every gas parameter inside the kernel
        lay["_cont"], lay["_dcont"] = cont, dcont
        # only the gases the state vector names need their amount gradients (the others' rows of xmap are zero)
        wanted = {j for kind, j in st.blocks if kind == "VMR"}
        eng.set_gradient_gases([i for i, col in enumerate(self.igas_map) if int(col) in wanted],
                               temperature=any(kind == "T" for kind, _ in st.blocks))
        try:
            spec = self._analytic_chain(eng, lay, amount, path, NVMR, NPAR, NDUST, NPRO, P_)
        finally:
            eng.set_gradient_gases(None)
        xmap = np.zeros((st.NX, NPAR, NPRO))
        lev = np.arange(NPRO)
        for b, (kind, j) in enumerate(st.blocks):
            if kind == "T":
                xmap[b * NPRO + lev, NVMR, lev] = 1.0
            else:
                xmap[b * NPRO + lev, j, lev] = VMR[:, j]           # d VMR / d ln VMR
        xv = eng.map2xvec(None, self.W, NVMR, NDUST, NPRO, P_, st.NX, xmap)          # (W, P, NX)
        return spec.reshape(self.W * P_), xv.reshape(self.W * P_, st.NX)

    def _analytic_chain(self, eng, lay, amount, path, NVMR, NPAR, NDUST, NPRO, P_):
        # the layer-level and the level-level gradients (80 MB each at C3) stay on the device: only KK comes back
        spec, _, _ = eng.cirsradg_ck_thermal(self.ISPACE, lay["PRESS"], lay["TEMP"], amount, lay["_cont"], None, NVMR, NPAR,
                                             self.igas_map.astype(np.int32), path.NLAYIN, path.LAYINC, path.SCALE,
                                             path.EMTEMP, self.TSURF, gradients_on_device=True, dtau_every_gas=lay["_dcont"])
        eng.map2pro(None, self.W, NVMR, NDUST, NPRO, P_, path.NLAYIN, path.LAYINC, lay["DTE"], lay["DAM"], lay["DCO"],
                    to_host=False)
        return spec


