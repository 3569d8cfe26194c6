"""AnsfmEngine: one context per GPU around libansfm.so (see include/ansfm.h).

Host-array methods take/return NumPy arrays in the reference's layouts (drop-in for the numba
seams); `*_dev` methods take torch CUDA(HIP) tensors already resident in HBM and run
asynchronously on the engine's stream.  torch is plumbing only (device memory, streams).
"""
import ctypes as C
import os
import numpy as np

from . import _lib

_f8 = np.float64


def _np(a, dtype=_f8):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _is_f32(a):
    return getattr(a, "dtype", None) == np.float32


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())  # torch tensor


def _fingerprint(a):
    """Identity of a result array handed back to the caller: address, shape and a few sampled values, to recognise
    "the array I returned" when it comes back as the next step's input.  Arrays that have a device twin are handed out
    read-only (flags.writeable = False), so "unmodified" does not rest on the samples: an in-place edit raises, an
    edited copy has another address."""
    flat = a.reshape(-1)
    idx = np.linspace(0, flat.size - 1, num=min(flat.size, 32)).astype(np.int64)
    return (a.ctypes.data, a.shape, a.strides, flat[idx].tobytes())


class AnsfmEngine:
    def __init__(self, device=0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._lib.ansfm_create(int(device), C.byref(self._ctx))
        if rc != 0:
            raise _lib.AnsfmError(f"ansfm_create(device={device}) failed ({_lib.ERR_NAMES.get(rc, rc)}): "
                                  "no usable HIP device; there is no CPU fallback")
        self.device = int(device)
        self.dims = None

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.ansfm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.ansfm_last_error(self._ctx)
            msg = msg.decode() if msg else ""
            exc = ValueError if rc in (1, 4) else (NotImplementedError if rc == 5 else _lib.AnsfmError)
            raise exc(f"{what}: {_lib.ERR_NAMES.get(rc, rc)}: {msg}")

    # ---- stream -------------------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr):
        """Run on the caller's HIP stream.  0 / None = the engine's own (non-blocking) stream: a null handle cannot name the
        legacy default stream through the C-ABI, so code that shares device buffers with another runtime on ITS default
        stream must order the two itself (see `stream_ptr`, `BatchedCKThermalModel.spectra_batch`)."""
        self._check(self._lib.ansfm_set_stream(self._ctx, C.c_void_p(hip_stream_ptr or 0)), "set_stream")
        self._stream_ptr = int(hip_stream_ptr or 0)

    def _order_after_torch(self, t):
        """A torch CUDA tensor handed to a device entry point: unless torch's current stream IS the engine's, wait for what
        torch has queued (its kernels may still be writing the tensor)."""
        try:
            import torch
            cur = torch.cuda.current_stream(t.device)
            if cur.cuda_stream == 0 or cur.cuda_stream != self.stream_ptr:
                cur.synchronize()
        except ImportError:
            pass

    @property
    def stream_ptr(self):
        """handle of the stream the engine was given, 0 while it runs on its own"""
        return getattr(self, "_stream_ptr", 0)

    def synchronize(self):
        self._check(self._lib.ansfm_synchronize(self._ctx), "synchronize")

    def set_gradient_gases(self, gases=None, temperature=True):
        """Spectroscopic gases (indices into the uploaded table) whose amount gradients `cirsradg_ck_*` computes; None = all
        (the reference's behaviour).  The others' parameters come back without their gas part.  temperature=False also
        leaves out the k-table part of the temperature gradient (a state vector without temperature elements).  Sticky."""
        if gases is not None:
            bad = [int(g) for g in gases if not 0 <= int(g) < 31]
            if bad:
                raise ValueError("set_gradient_gases: gas indices must lie in [0, 31) (bit 31 is the temperature slot): %s" % bad)
        mask = 0xFFFFFFFF if gases is None else sum(1 << int(g) for g in set(int(g) for g in gases))
        mask = (mask | 0x80000000) if temperature else (mask & 0x7FFFFFFF)
        self._check(self._lib.ansfm_set_gradient_gases(self._ctx, C.c_uint(mask & 0xFFFFFFFF)), "set_gradient_gases")

    def set_f32_semantics(self, grid_f32, delg_f32):
        """Reproduce NumPy's float32 arithmetic when Spectroscopy_0.PRESS/TEMP (grid) / DELG are
        float32 arrays, as they are after read_tables on .kta files (see include/ansfm.h)."""
        self.grid_f32, self.delg_f32 = bool(grid_f32), bool(delg_f32)
        self._check(self._lib.ansfm_set_f32_semantics(self._ctx, int(self.grid_f32), int(self.delg_f32)),
                    "set_f32_semantics")

    # ---- k-table ------------------------------------------------------------------------------
    def upload_ktable(self, K, PRESS, TEMP, WAVE, DELG):
        """K (W,G,NP,NT,S) float64: NumPy array (host) or torch CUDA tensor (device).
        float32 PRESS/TEMP/DELG arrays switch on the matching float32 semantics."""
        self.set_f32_semantics(_is_f32(PRESS) or _is_f32(TEMP), _is_f32(DELG))
        PRESS = _np(PRESS); TEMP = _np(TEMP); WAVE = _np(WAVE); DELG = _np(DELG)
        W, G, NP, NT, S = (int(x) for x in K.shape)
        assert PRESS.shape == (NP,) and TEMP.shape == (NT,) and WAVE.shape == (W,) and DELG.shape == (G,)
        if isinstance(K, np.ndarray):
            K = _np(K)
            rc = self._lib.ansfm_upload_ktable(self._ctx, W, G, NP, NT, S, _ptr(K), _ptr(PRESS), _ptr(TEMP),
                                               _ptr(WAVE), _ptr(DELG))
        else:
            assert K.is_contiguous() and K.dtype.is_floating_point and K.element_size() == 8
            self._order_after_torch(K)
            rc = self._lib.ansfm_upload_ktable_dev(self._ctx, W, G, NP, NT, S, _ptr(K), _ptr(PRESS), _ptr(TEMP),
                                                   _ptr(WAVE), _ptr(DELG))
        self._check(rc, "upload_ktable")
        self.dims = (W, G, NP, NT, S)
        self.WAVE, self.DELG = WAVE, DELG

    def upload_ktable_files(self, paths, wavemin=0.0, wavemax=1.0e10):
        """Spectroscopy_0.read_tables (:1448) for binary .kta tables, straight from the files into HBM (the float64 K
        array is never built on the host).  Returns WAVE, PRESS, TEMP, DELG of the uploaded table."""
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        rc = self._lib.ansfm_upload_ktable_files(self._ctx, len(paths), arr, float(wavemin), float(wavemax))
        self._check(rc, "upload_ktable_files")
        self.dims, _ = self.ktable_info()
        self.grid_f32 = True
        W, G, NP, NT, S = self.dims
        WAVE, PRESS, TEMP, DELG = np.empty(W), np.empty(NP), np.empty(NT), np.empty(G)
        self._check(self._lib.ansfm_ktable_grids(self._ctx, _ptr(WAVE), _ptr(PRESS), _ptr(TEMP), _ptr(DELG)), "ktable_grids")
        return WAVE, PRESS.astype(np.float32), TEMP.astype(np.float32), DELG.astype(np.float32)

    def upload_lbltable_files(self, paths, wavemin=0.0, wavemax=1.0e10):
        """Spectroscopy_0.read_tables (:1448) for binary .lta LBL tables (ILBL = 2), straight from the files into HBM
        (read_lbltable's Python loop over (wavenumber, pressure) never runs).  Returns WAVE, PRESS, TEMP."""
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        rc = self._lib.ansfm_upload_lbltable_files(self._ctx, len(paths), arr, float(wavemin), float(wavemax))
        self._check(rc, "upload_lbltable_files")
        self.dims, _ = self.ktable_info()
        self.grid_f32 = True
        W, G, NP, NT, S = self.dims
        per_level = _lib.read_lbltable_header(paths[-1])[4] < 0          # NT < 0 in the file: TEMP comes back (NP, |NT|)
        WAVE, PRESS, TEMP, DELG = np.empty(W), np.empty(NP), np.empty((NP, NT) if per_level else NT), np.empty(G)
        self._check(self._lib.ansfm_ktable_grids(self._ctx, _ptr(WAVE), _ptr(PRESS), _ptr(TEMP), _ptr(DELG)), "ktable_grids")
        self.WAVE, self.DELG = WAVE, np.array([1.0])
        return WAVE, PRESS.astype(np.float32), TEMP.astype(np.float32)

    def upload_lbltable(self, K, PRESS, TEMP, WAVE):
        """LBL table (ILBL = 2): K (W,NP,|NT|,S) float64 host array; TEMP (|NT|,) or (NP,|NT|) (NT < 0 form)."""
        self.set_f32_semantics(_is_f32(PRESS) or _is_f32(TEMP), False)
        K = _np(K); PRESS = _np(PRESS); TEMP = _np(TEMP); WAVE = _np(WAVE)
        W, NP, NT, S = (int(x) for x in K.shape)
        temp2d = int(TEMP.ndim == 2)
        assert PRESS.shape == (NP,) and WAVE.shape == (W,) and TEMP.shape == ((NP, NT) if temp2d else (NT,))
        self._check(self._lib.ansfm_upload_lbltable(self._ctx, W, NP, NT, S, _ptr(K), _ptr(PRESS), _ptr(TEMP), temp2d,
                                                    _ptr(WAVE)), "upload_lbltable")
        self.dims = (W, 1, NP, NT, S)
        self.WAVE, self.DELG = WAVE, np.array([1.0])

    def calc_klbl(self, press, temp, grad=False):
        press = _np(press); temp = _np(temp)
        W, _, NP, NT, S = self.dims
        L = press.shape[0]
        k = np.empty((W, L, S)); dk = np.empty((W, L, S)) if grad else None
        self._check(self._lib.ansfm_calc_klbl(self._ctx, L, _ptr(press), _ptr(temp), _ptr(k), _ptr(dk)), "calc_klbl")
        return (k, dk) if grad else k

    def ktable_info(self):
        dims = (C.c_int64 * 5)()
        mono = C.c_int()
        self._check(self._lib.ansfm_ktable_info(self._ctx, dims, C.byref(mono)), "ktable_info")
        return tuple(int(d) for d in dims), bool(mono.value)

    # ---- array-level seams --------------------------------------------------------------------
    def calc_k(self, press, temp, grad=False):
        press = _np(press); temp = _np(temp)
        W, G, NP, NT, S = self.dims
        L = press.shape[0]
        k = np.empty((W, G, L, S)); dk = np.empty((W, G, L, S)) if grad else None
        self._check(self._lib.ansfm_calc_k(self._ctx, L, _ptr(press), _ptr(temp), _ptr(k), _ptr(dk)), "calc_k")
        return (k, dk) if grad else k

    def k_overlap(self, del_g, k_w_g_l_gas, amount_layer):
        self.set_f32_semantics(getattr(self, "grid_f32", False), _is_f32(del_g))
        del_g = _np(del_g); k = _np(k_w_g_l_gas); am = _np(amount_layer)
        W, G, L, S = k.shape
        if am.shape != (S, L):
            raise ValueError("amount_layer must be (NGAS, NLAYER)")
        tau = np.empty((W, G, L))
        self._check(self._lib.ansfm_k_overlap(self._ctx, W, G, L, S, _ptr(del_g), _ptr(k), _ptr(am), _ptr(tau)),
                    "k_overlap")
        return tau

    def k_overlapg(self, del_g, k_w_g_l_gas, dkdT_w_g_l_gas, amount_layer):
        self.set_f32_semantics(getattr(self, "grid_f32", False), _is_f32(del_g))
        del_g = _np(del_g); k = _np(k_w_g_l_gas); dkdT = _np(dkdT_w_g_l_gas); am = _np(amount_layer)
        W, G, L, S = k.shape
        if am.shape != (S, L) or dkdT.shape != k.shape:
            raise ValueError("shapes: k, dkdT (NWAVE,NG,NLAYER,NGAS); amount (NGAS,NLAYER)")
        tau = np.empty((W, G, L)); dk = np.empty((W, G, L, S + 1))
        self._check(self._lib.ansfm_k_overlapg(self._ctx, W, G, L, S, _ptr(del_g), _ptr(k), _ptr(dkdT), _ptr(am),
                                               _ptr(tau), _ptr(dk)), "k_overlapg")
        return tau, dk

    def calc_thermal_emission_spectrum(self, ISPACE, WAVE, TAUTOT_PATH, EMITOT_PATH, TEMP, PRESS, TSURF,
                                       EMISSIVITY, SOLFLUX, REFLECTANCE, SOL_ANG, EMISS_ANG):
        WAVE = _np(WAVE); TAU = _np(TAUTOT_PATH); EMI = _np(EMITOT_PATH)
        W, G, Li = TAU.shape
        out = np.empty((W, G))
        rc = self._lib.ansfm_thermal_emission(self._ctx, int(ISPACE), W, G, Li, _ptr(WAVE), _ptr(TAU), _ptr(EMI),
                                              _ptr(_np(TEMP)), _ptr(_np(PRESS)), float(TSURF),
                                              _ptr(_np(EMISSIVITY)), _ptr(_np(SOLFLUX)), _ptr(_np(REFLECTANCE)),
                                              float(SOL_ANG), float(EMISS_ANG), _ptr(out))
        self._check(rc, "thermal_emission")
        return out

    # ---- fused CIRSrad (batch) ----------------------------------------------------------------
    def calc_thermal_emission_spectrumg(self, ISPACE, WAVE, TAUTOT_PATH, dTAUTOT_PATH, NVMR, TEMP, PRESS, TSURF, EMISSIVITY):
        """ForwardModel_0.calc_thermal_emission_spectrumg (:6380), same arguments -> SPECOUT (NWAVE, NG),
        dSPECOUT (NWAVE, NG, NPAR, NLAYIN), dTSURF (NWAVE, NG)."""
        TAU = _np(TAUTOT_PATH); dTAU = _np(dTAUTOT_PATH)
        W, G, NPAR, Li = dTAU.shape
        if TAU.shape != (W, G, Li):
            raise ValueError("TAUTOT_PATH must be (NWAVE, NG, NLAYIN) and dTAUTOT_PATH (NWAVE, NG, NPAR, NLAYIN)")
        spec = np.empty((W, G)); dspec = np.empty((W, G, NPAR, Li)); dts = np.empty((W, G))
        rc = self._lib.ansfm_thermal_emission_g(self._ctx, int(ISPACE), W, G, NPAR, Li, _ptr(_np(WAVE)), _ptr(TAU), _ptr(dTAU),
                                                int(NVMR), _ptr(_np(TEMP)), _ptr(_np(PRESS)), float(TSURF), _ptr(_np(EMISSIVITY)),
                                                _ptr(spec), _ptr(dspec), _ptr(dts))
        self._check(rc, "thermal_emission_g")
        return spec, dspec, dts

    def cirsrad_ck_thermal(self, ISPACE, lay_press_pa, lay_temp, amount, taucont, NLAYIN, LAYINC, SCALE, EMTEMP,
                           TSURF, EMISSIVITY=None, SOLFLUX=None, REFLECTANCE=None, SOL_ANG=None, EMISS_ANG=None,
                           xfac=None):
        """Host arrays; leading model axis optional.  Returns SPECOUT (n,W,P) (or (W,P))."""
        W, G, NP, NT, S = self.dims
        lay_press_pa = _np(lay_press_pa)
        single = lay_press_pa.ndim == 1
        lp = np.atleast_2d(lay_press_pa); n, L = lp.shape
        lt = _np(np.atleast_2d(_np(lay_temp)))
        am = _np(amount).reshape(n, S, L)
        tc = None if taucont is None else _np(taucont).reshape(n, W, L)
        LAYINC = _np(LAYINC, np.int32); NLAYIN = _np(np.atleast_1d(NLAYIN), np.int32)
        if LAYINC.ndim == 1:
            LAYINC = LAYINC[:, None]
        LIMAX, P = LAYINC.shape
        SC = _np(np.broadcast_to(_np(SCALE).reshape(-1, LIMAX, P), (n, LIMAX, P)))
        ET = _np(np.broadcast_to(_np(EMTEMP).reshape(-1, LIMAX, P), (n, LIMAX, P)))
        TS = _np(np.broadcast_to(np.atleast_1d(_np(TSURF)), (n,)))
        out = np.empty((n, W, P))
        rc = self._lib.ansfm_cirsrad_ck_thermal(
            self._ctx, int(ISPACE), n, L, _ptr(lp), _ptr(lt), _ptr(am), _ptr(tc), P, LIMAX, _ptr(NLAYIN),
            _ptr(LAYINC), _ptr(SC), _ptr(ET), _ptr(TS), _ptr(_np(EMISSIVITY)), _ptr(_np(SOLFLUX)),
            _ptr(_np(REFLECTANCE)), _ptr(None if SOL_ANG is None else _np(np.atleast_1d(SOL_ANG))),
            _ptr(None if EMISS_ANG is None else _np(np.atleast_1d(EMISS_ANG))), _ptr(_np(xfac)), _ptr(out))
        self._check(rc, "cirsrad_ck_thermal")
        return out[0] if single else out

    def calc_singlescatt_plane_spectrum(self, ISPACE, WAVE, TAUTOT_PATH, TEMP, OMEGA, PHASE, TSURF, EMISSIVITY, BRDF, SOLFLUX,
                                        SOL_ANG, EMISS_ANG):
        """ForwardModel_0.calc_singlescatt_plane_spectrum (:6509), same arguments -> SPECOUT (NWAVE, NG)."""
        TAU = _np(TAUTOT_PATH); W, G, Li = TAU.shape
        out = np.empty((W, G))
        rc = self._lib.ansfm_singlescatt_plane_spectrum(
            self._ctx, int(ISPACE), W, G, Li, _ptr(_np(WAVE)), _ptr(TAU), _ptr(_np(TEMP)), _ptr(_np(OMEGA).reshape(W, G, Li)),
            _ptr(_np(PHASE).reshape(W, Li)), float(TSURF), _ptr(_np(EMISSIVITY)), _ptr(_np(BRDF)), _ptr(_np(SOLFLUX)),
            float(SOL_ANG), float(EMISS_ANG), _ptr(out))
        self._check(rc, "singlescatt_plane_spectrum")
        return out

    def cirsrad_ck_singlescatt(self, ISPACE, lay_press_pa, lay_temp, amount, taucont, tausca, phase, NLAYIN, LAYINC, SCALE, EMTEMP,
                               TSURF, EMISSIVITY, BRDF, SOLFLUX, SOL_ANG, EMISS_ANG, xfac=None):
        """CIRSrad, single-scattering branch (:4251-4336) on the uploaded k-table: taucont / tausca (NWAVE, NLAY), phase
        (NPATH, NWAVE, NLAY), BRDF (NWAVE, NPATH) -> SPECOUT (NWAVE, NPATH)."""
        W, G, NP, NT, S = self.dims
        lp = _np(lay_press_pa); L = lp.shape[0]
        LAYINC = _np(LAYINC, np.int32); NLAYIN = _np(np.atleast_1d(NLAYIN), np.int32)
        if LAYINC.ndim == 1:
            LAYINC = LAYINC[:, None]
        LIMAX, P = LAYINC.shape
        out = np.empty((W, P))
        rc = self._lib.ansfm_cirsrad_ck_singlescatt(
            self._ctx, int(ISPACE), L, _ptr(lp), _ptr(_np(lay_temp)), _ptr(_np(amount).reshape(S, L)),
            _ptr(None if taucont is None else _np(taucont).reshape(W, L)), _ptr(_np(tausca).reshape(W, L)),
            _ptr(_np(phase).reshape(P, W, L)), P, LIMAX, _ptr(NLAYIN), _ptr(LAYINC), _ptr(_np(SCALE).reshape(LIMAX, P)),
            _ptr(_np(EMTEMP).reshape(LIMAX, P)), float(TSURF), _ptr(_np(EMISSIVITY)), _ptr(_np(BRDF).reshape(W, P)),
            _ptr(_np(SOLFLUX)), _ptr(_np(np.atleast_1d(SOL_ANG))), _ptr(_np(np.atleast_1d(EMISS_ANG))), _ptr(_np(xfac)), _ptr(out))
        self._check(rc, "cirsrad_ck_singlescatt")
        return out

    def cirsrad_ck_transmission(self, lay_press_pa, lay_temp, amount, taucont, NLAYIN, LAYINC, SCALE, xfac=None):
        """CIRSrad, pure-transmission branch (calculate_transmission_spectrum :4110): SPECOUT (n, W, P) (or (W, P)) =
        xfac * sum_g DELG exp(-sum over the path's layers of TAUTOT_LAYINC)."""
        W, G, NP, NT, S = self.dims
        lay_press_pa = _np(lay_press_pa)
        single = lay_press_pa.ndim == 1
        lp = np.atleast_2d(lay_press_pa); n, L = lp.shape
        lt = _np(np.atleast_2d(_np(lay_temp)))
        am = _np(amount).reshape(n, S, L)
        tc = None if taucont is None else _np(taucont).reshape(n, W, L)
        LAYINC = _np(LAYINC, np.int32); NLAYIN = _np(np.atleast_1d(NLAYIN), np.int32)
        if LAYINC.ndim == 1:
            LAYINC = LAYINC[:, None]
        LIMAX, P = LAYINC.shape
        SC = _np(np.broadcast_to(_np(SCALE).reshape(-1, LIMAX, P), (n, LIMAX, P)))
        out = np.empty((n, W, P))
        rc = self._lib.ansfm_cirsrad_ck_transmission(self._ctx, n, L, _ptr(lp), _ptr(lt), _ptr(am), _ptr(tc), P, LIMAX, _ptr(NLAYIN),
                                                     _ptr(LAYINC), _ptr(SC), _ptr(_np(xfac)), _ptr(out))
        self._check(rc, "cirsrad_ck_transmission")
        return out[0] if single else out

    def cirsradg_ck_transmission(self, lay_press_pa, lay_temp, amount, taucont, dtaucon, NVMR, NPAR, igas_map, NLAYIN,
                                 LAYINC, SCALE, xfac=None):
        """CIRSrad(return_grad=True), pure-transmission branch (:4110-4131, :4504-4507): SPECOUT (n, W, P) and
        dSPECOUT (n, W, NPAR, LIMAX, P) = -sum_g DELG xfac exp(-tau_path) dTAUTOT_LAYINC (leading axis dropped for a
        single model).  dTSURF of this branch is zero."""
        W, G, NP, NT, S = self.dims
        lay_press_pa = _np(lay_press_pa)
        single = lay_press_pa.ndim == 1
        lp = np.atleast_2d(lay_press_pa); n, L = lp.shape
        lt = _np(np.atleast_2d(_np(lay_temp)))
        am = _np(amount).reshape(n, S, L)
        tc = None if taucont is None else _np(taucont).reshape(n, W, L)
        dtc = None if dtaucon is None else _np(dtaucon).reshape(n, W, NPAR, L)
        LAYINC = _np(LAYINC, np.int32); NLAYIN = _np(np.atleast_1d(NLAYIN), np.int32)
        if LAYINC.ndim == 1:
            LAYINC = LAYINC[:, None]
        LIMAX, P = LAYINC.shape
        SC = _np(np.broadcast_to(_np(SCALE).reshape(-1, LIMAX, P), (n, LIMAX, P)))
        ig = _np(igas_map, np.int32)
        spec = np.empty((n, W, P)); dspec = np.empty((n, W, NPAR, LIMAX, P))
        rc = self._lib.ansfm_cirsradg_ck_transmission(
            self._ctx, n, L, _ptr(lp), _ptr(lt), _ptr(am), _ptr(tc), _ptr(dtc), int(NVMR), int(NPAR), _ptr(ig), P, LIMAX,
            _ptr(NLAYIN), _ptr(LAYINC), _ptr(SC), _ptr(_np(xfac)), _ptr(spec), _ptr(dspec))
        self._check(rc, "cirsradg_ck_transmission")
        self._chain_dspec = None
        if n == 1:
            dspec.flags.writeable = False
            self._chain_dspec = _fingerprint(dspec[0])
        return (spec[0], dspec[0]) if single else (spec, dspec)

    def cirsradg_ck_thermal(self, ISPACE, lay_press_pa, lay_temp, amount, taucont, dtaucon, NVMR, NPAR, igas_map,
                            NLAYIN, LAYINC, SCALE, EMTEMP, TSURF, EMISSIVITY=None, xfac=None, gradients_on_device=False,
                            dtau_every_gas=None):
        """CIRSrad(return_grad=True): returns SPECOUT (n,W,P), dSPECOUT (n,W,NPAR,LIMAX,P), dTSURF (n,W,P)
        (leading axis dropped for a single model).  gradients_on_device (single model): dSPECOUT is not copied to the host
        (None is returned in its place); `map2pro(None, ...)` takes it from the device.  dtau_every_gas (W, L), single
        model: added to dTAUCON of every gas parameter (the Rayleigh term, :3955-3957) without building the NVMR copies."""
        W, G, NP, NT, S = self.dims
        lay_press_pa = _np(lay_press_pa)
        single = lay_press_pa.ndim == 1
        lp = np.atleast_2d(lay_press_pa); n, L = lp.shape
        lt = _np(np.atleast_2d(_np(lay_temp)))
        am = _np(amount).reshape(n, S, L)
        tc = None if taucont is None else _np(taucont).reshape(n, W, L)
        dtc = None if dtaucon is None else _np(dtaucon).reshape(n, W, NPAR, L)
        LAYINC = _np(LAYINC, np.int32); NLAYIN = _np(np.atleast_1d(NLAYIN), np.int32)
        if LAYINC.ndim == 1:
            LAYINC = LAYINC[:, None]
        LIMAX, P = LAYINC.shape
        SC = _np(np.broadcast_to(_np(SCALE).reshape(-1, LIMAX, P), (n, LIMAX, P)))
        ET = _np(np.broadcast_to(_np(EMTEMP).reshape(-1, LIMAX, P), (n, LIMAX, P)))
        TS = _np(np.broadcast_to(np.atleast_1d(_np(TSURF)), (n,)))
        ig = _np(igas_map, np.int32)
        on_dev = bool(gradients_on_device) and n == 1
        if dtau_every_gas is not None:
            dg = _np(dtau_every_gas)
            if n != 1 or dg.shape != (W, L):
                raise ValueError("cirsradg_ck_thermal: dtau_every_gas must be (NWAVE, NLAY) for a single model")
            self._check(self._lib.ansfm_set_shared_gas_gradient(self._ctx, L, _ptr(dg)), "set_shared_gas_gradient")
        spec = np.empty((n, W, P)); dts = np.empty((n, W, P))
        dspec = None if on_dev else np.empty((n, W, NPAR, LIMAX, P))
        try:
            rc = self._lib.ansfm_cirsradg_ck_thermal(
                self._ctx, int(ISPACE), n, L, _ptr(lp), _ptr(lt), _ptr(am), _ptr(tc), _ptr(dtc), int(NVMR), int(NPAR),
                _ptr(ig), P, LIMAX, _ptr(NLAYIN), _ptr(LAYINC), _ptr(SC), _ptr(ET), _ptr(TS), _ptr(_np(EMISSIVITY)),
                _ptr(_np(xfac)), _ptr(spec), _ptr(dspec), _ptr(dts))
        finally:
            if dtau_every_gas is not None:           # consumed by a successful call; cancelled if the call failed before that
                self._lib.ansfm_set_shared_gas_gradient(self._ctx, 0, None)
        self._check(rc, "cirsradg_ck_thermal")
        self._chain_dspec = None
        if on_dev:
            self._chain_dspec = ("device", W, int(NPAR), LIMAX, P)
            return (spec[0], None, dts[0]) if single else (spec, None, dts)
        if n == 1:                       # device copy usable by map2pro: hand the host array out read-only
            dspec.flags.writeable = False
            self._chain_dspec = _fingerprint(dspec[0])
        return (spec[0], dspec[0], dts[0]) if single else (spec, dspec, dts)

    def cirsrad_ck_thermal_dev(self, ISPACE, n_models, L, lay_press_pa, lay_temp, amount, taucont, P, LIMAX,
                               NLAYIN, LAYINC, SCALE, EMTEMP, TSURF, EMISSIVITY, SOLFLUX, REFLECTANCE, SOL_ANG,
                               EMISS_ANG, xfac, SPECOUT):
        """All arguments torch device tensors (or None where optional); asynchronous."""
        rc = self._lib.ansfm_cirsrad_ck_thermal_dev(
            self._ctx, int(ISPACE), int(n_models), int(L), _ptr(lay_press_pa), _ptr(lay_temp), _ptr(amount),
            _ptr(taucont), int(P), int(LIMAX), _ptr(NLAYIN), _ptr(LAYINC), _ptr(SCALE), _ptr(EMTEMP), _ptr(TSURF),
            _ptr(EMISSIVITY), _ptr(SOLFLUX), _ptr(REFLECTANCE), _ptr(SOL_ANG), _ptr(EMISS_ANG), _ptr(xfac),
            _ptr(SPECOUT))
        self._check(rc, "cirsrad_ck_thermal_dev")

    def cirsrad_ck_thermal_ray_dev(self, ISPACE, n_models, L, lay_press_pa, lay_temp, amount, IRAY, TOTAM, f4, P, LIMAX,
                                   NLAYIN, LAYINC, SCALE, EMTEMP, TSURF, EMISSIVITY, SOLFLUX, REFLECTANCE, SOL_ANG,
                                   EMISS_ANG, xfac, SPECOUT, variant=None):
        """cirsrad_ck_thermal_dev for a batch whose continuum is Rayleigh scattering alone: TOTAM (n, L) and, IRAY 4,
        f4 (n, L, 4) (see rayleigh_f4) are device tensors; the continuum is formed per distinct layer inside the call."""
        mode = 12 if variant == "v" else int(IRAY)
        rc = self._lib.ansfm_cirsrad_ck_thermal_ray_dev(
            self._ctx, int(ISPACE), int(n_models), int(L), _ptr(lay_press_pa), _ptr(lay_temp), _ptr(amount), mode, _ptr(TOTAM),
            _ptr(f4), int(P), int(LIMAX), _ptr(NLAYIN), _ptr(LAYINC), _ptr(SCALE), _ptr(EMTEMP), _ptr(TSURF), _ptr(EMISSIVITY),
            _ptr(SOLFLUX), _ptr(REFLECTANCE), _ptr(SOL_ANG), _ptr(EMISS_ANG), _ptr(xfac), _ptr(SPECOUT))
        self._check(rc, "cirsrad_ck_thermal_ray_dev")

    @staticmethod
    def rayleigh_f4(ID, ISO, VMR):
        """calc_tau_rayleighls' composition (:5748-5767): VMR (..., NVMR) numpy array or torch tensor -> (..., 4) mixing
        ratios of H2, He, CH4, NH3, zero where the gas is absent; the last matching gas wins."""
        ID = np.asarray(ID); ISO = np.asarray(ISO)
        if isinstance(VMR, np.ndarray):
            f4 = np.zeros(VMR.shape[:-1] + (4,))
        else:
            import torch
            f4 = torch.zeros(tuple(VMR.shape[:-1]) + (4,), dtype=VMR.dtype, device=VMR.device)
        for j in range(ID.size):
            if ISO[j] in (0, 1):
                col = {39: 0, 40: 1, 6: 2, 11: 3}.get(int(ID[j]))
                if col is not None:
                    f4[..., col] = VMR[..., j]
        return f4

    def scloud11wave_core(self, phasarr, radg, sol_angs, emiss_angs, solar, aphis, lowbc, brdf_matrix, mu1, wt1, nf,
                          vwaves, bnu, taus, tauray, omegas_s, nphi, iray, imie, lfrac):
        """Multiple_Scattering_Core.scloud11wave_core (same arguments) -> rad (NPATH, NG, NWAVE)."""
        phasarr = _np(phasarr); taus = _np(taus)
        ncont, nwave, _, nth = phasarr.shape
        nmu = len(mu1); ngeom = len(emiss_angs)
        _, ng, nlay = taus.shape
        if ngeom > self.MS_PATHS_PER_CALL:      # the chain kernels keep one path per lane of a 16-lane row: groups of paths
            emi = np.asarray(emiss_angs, dtype=float)
            if not (np.all(emi < 90) or np.all(emi > 90)):
                raise ValueError("scloud11wave_core: INVALID: Emission angles are a mix of values above and below 90 degrees.")
            parts = [self.scloud11wave_core(phasarr, radg, np.asarray(sol_angs)[g], emi[g], solar, np.asarray(aphis)[g], lowbc,
                                            brdf_matrix, mu1, wt1, nf, vwaves, bnu, taus, tauray, omegas_s, nphi, iray, imie, lfrac)
                     for g in self._path_groups(ngeom)]
            return np.concatenate(parts, axis=0)
        rad = np.empty((ngeom, ng, nwave))
        rc = self._lib.ansfm_scloud11wave_core(
            self._ctx, ncont, nwave, nth, _ptr(phasarr), _ptr(_np(radg)), ngeom, _ptr(_np(sol_angs)),
            _ptr(_np(emiss_angs)), _ptr(_np(solar)), _ptr(_np(aphis)), int(lowbc), _ptr(_np(brdf_matrix)), nmu,
            _ptr(_np(mu1)), _ptr(_np(wt1)), int(nf), _ptr(_np(bnu)), ng, nlay, _ptr(taus), _ptr(_np(tauray)),
            _ptr(_np(omegas_s)), int(nphi), int(iray), int(imie), _ptr(_np(lfrac)), _ptr(rad))
        self._check(rc, "scloud11wave_core")
        return rad

    MS_PATHS_PER_CALL = 16

    def _path_groups(self, ngeom):
        n = self.MS_PATHS_PER_CALL
        return [slice(i, min(i + n, ngeom)) for i in range(0, ngeom, n)]

    def cirsrad_ck_scatter(self, ISPACE, lay_press_pa, lay_temp, amount, TAUCIA, TAUDUST, TAURAY, TAUSCAT, phasarr, lfrac,
                           radg, sol_angs, emiss_angs, aphis, solar, lowbc, brdf_matrix, mu1, wt1, nf, nphi, iray, imie,
                           xfac=None, return_spec_g=False):
        """CIRSrad, scattering branch (ForwardModel_0.py:4478-4501, :4343, scloud11wave :5018-5165) on the uploaded k-table:
        vertical gas opacities, TAUTOT, OMEGA and BB are formed in HBM and go straight into the doubling / adding kernels.
        TAUCIA / TAUDUST / TAURAY / TAUSCAT (NWAVE, NLAY) or None; phasarr (NDUST, NWAVE, 2, NTHETA) and lfrac (NWAVE, NDUST,
        NLAY) as scloud11wave_core takes them -> SPECOUT (NWAVE, NPATH) [, SPEC (NWAVE, NG, NPATH)]."""
        dims, _ = self.ktable_info()
        W, G, S = dims[0], dims[1], dims[4]
        lay_press_pa = _np(lay_press_pa); L = lay_press_pa.shape[0]
        amount = _np(amount)
        if amount.shape != (S, L):
            raise ValueError("amount must be (NGAS, NLAY)")
        wl = lambda a: None if a is None else _np(a).reshape(W, L)
        phasarr = None if phasarr is None else _np(phasarr)
        ncont = 0 if phasarr is None else phasarr.shape[0]
        nth = 0 if phasarr is None else phasarr.shape[3]
        sol = _np(np.atleast_1d(sol_angs)); emi = _np(np.atleast_1d(emiss_angs)); aph = _np(np.atleast_1d(aphis))
        P = sol.shape[0]
        if P > self.MS_PATHS_PER_CALL:          # more paths than one call takes: groups of paths, the chains of each re-run
            if not (np.all(emi < 90) or np.all(emi > 90)):
                raise ValueError("cirsrad_ck_scatter: INVALID: Emission angles are a mix of values above and below 90 degrees.")
            parts = [self.cirsrad_ck_scatter(ISPACE, lay_press_pa, lay_temp, amount, TAUCIA, TAUDUST, TAURAY, TAUSCAT, phasarr, lfrac,
                                             radg, sol[g], emi[g], aph[g], solar, lowbc, brdf_matrix, mu1, wt1, nf, nphi, iray, imie,
                                             xfac=xfac, return_spec_g=return_spec_g) for g in self._path_groups(P)]
            if return_spec_g:
                return np.concatenate([p_[0] for p_ in parts], axis=1), np.concatenate([p_[1] for p_ in parts], axis=2)
            return np.concatenate(parts, axis=1)
        mu1 = _np(mu1); nmu = mu1.shape[0]
        out = np.empty((W, P)); spec_g = np.empty((W, G, P)) if return_spec_g else None
        rc = self._lib.ansfm_cirsrad_ck_scatter(
            self._ctx, int(ISPACE), L, _ptr(lay_press_pa), _ptr(_np(lay_temp)), _ptr(amount), _ptr(wl(TAUCIA)), _ptr(wl(TAUDUST)),
            _ptr(wl(TAURAY)), _ptr(wl(TAUSCAT)), ncont, nth, _ptr(phasarr), _ptr(_np(lfrac)), _ptr(_np(radg)), P, _ptr(sol),
            _ptr(emi), _ptr(aph), _ptr(_np(solar)), int(lowbc), _ptr(_np(brdf_matrix)), nmu, _ptr(mu1), _ptr(_np(wt1)), int(nf),
            int(nphi), int(iray), int(imie), _ptr(_np(xfac)), _ptr(out), _ptr(spec_g))
        self._check(rc, "cirsrad_ck_scatter")
        return (out, spec_g) if return_spec_g else out

    def cirsrad_ck_scatter_batch(self, ISPACE, lay_press_pa, lay_temp, amount, TAUCIA, TAUDUST, TAURAY, TAUSCAT, phasarr, lfrac,
                                 radg, sol_angs, emiss_angs, aphis, solar, lowbc, brdf_matrix, mu1, wt1, nf, nphi, iray, imie,
                                 xfac=None):
        """The scattering branch of CIRSrad for the n forward models of a numerical Jacobian (ansfm_cirsrad_ck_scatter_batch):
        lay_press_pa / lay_temp (n, NLAY), amount (n, NGAS, NLAY), TAUCIA / TAUDUST / TAURAY / TAUSCAT (n, NWAVE, NLAY) or None,
        lfrac (n, NWAVE, NDUST, NLAY), radg (n, NWAVE, NMU); the rest as `cirsrad_ck_scatter` -> SPECOUT (n, NWAVE, NPATH).
        Layers whose inputs equal model 0's are taken from model 0's doubling results (`last_scatter_cache()`)."""
        dims, _ = self.ktable_info()
        W, G, S = dims[0], dims[1], dims[4]
        lp = _np(lay_press_pa); n, L = lp.shape
        am = _np(amount)
        if am.shape != (n, S, L):
            raise ValueError("amount must be (n_models, NGAS, NLAY)")
        nwl = lambda a: None if a is None else _np(a).reshape(n, W, L)
        phasarr = None if phasarr is None else _np(phasarr)
        ncont = 0 if phasarr is None else phasarr.shape[0]
        nth = 0 if phasarr is None else phasarr.shape[3]
        sol = _np(np.atleast_1d(sol_angs)); emi = _np(np.atleast_1d(emiss_angs)); aph = _np(np.atleast_1d(aphis))
        P = sol.shape[0]
        mu1 = _np(mu1); nmu = mu1.shape[0]
        lf = None if lfrac is None else _np(lfrac).reshape(n, W, ncont, L)
        rg = _np(radg).reshape(n, W, nmu)
        out = np.empty((n, W, P))
        rc = self._lib.ansfm_cirsrad_ck_scatter_batch(
            self._ctx, int(ISPACE), n, L, _ptr(lp), _ptr(_np(lay_temp).reshape(n, L)), _ptr(am), _ptr(nwl(TAUCIA)), _ptr(nwl(TAUDUST)),
            _ptr(nwl(TAURAY)), _ptr(nwl(TAUSCAT)), ncont, nth, _ptr(phasarr), _ptr(lf), _ptr(rg), P, _ptr(sol), _ptr(emi), _ptr(aph),
            _ptr(_np(solar)), int(lowbc), _ptr(_np(brdf_matrix)), nmu, _ptr(mu1), _ptr(_np(wt1)), int(nf), int(nphi), int(iray),
            int(imie), _ptr(_np(xfac)), _ptr(out))
        self._check(rc, "cirsrad_ck_scatter_batch")
        return out

    def last_scatter_cache(self):
        """(layers of models 1..n-1 taken from model 0's doubling results, all such layers) of the last batched scatter call"""
        a = C.c_int64(); b = C.c_int64()
        self._check(self._lib.ansfm_last_scatter_cache(self._ctx, C.byref(a), C.byref(b)), "last_scatter_cache")
        return int(a.value), int(b.value)

    def add_line_set_monochromatic_absorption(self, wn_grid, lineshape_id, t_calc, t_ref, p_calc, p_ref, q_ratio,
                                              isotopic_abundance, isotopic_mass, mol_mix_frac, broadening_params, nu, sw,
                                              e_lower, stimulated_emission_at_t_ref, out, store=None, s_floor=0.0,
                                              wn_calc_window=25.0, wn_approx_window=75.0):
        """LineData_0.add_line_set_monochromatic_absorption (:280).  Scalars t_calc/p_calc/q_ratio = the reference
        call (out (nw,), store (4,N)); 1-D arrays of length L = batched over (T,p) points (out (L,nw), store (L,4,N)).
        `out` (float64, C-contiguous) is added to in place and returned."""
        wn_grid = _np(wn_grid); mmf = _np(mol_mix_frac); bp = _np(broadening_params)
        t = _np(np.atleast_1d(t_calc)); p = _np(np.atleast_1d(p_calc)); q = _np(np.atleast_1d(q_ratio))
        L = t.shape[0]
        nu = _np(nu); N = nu.shape[0]; M = mmf.shape[0]
        if out.dtype != np.float64 or not out.flags.c_contiguous or out.size != L * wn_grid.shape[0]:
            raise ValueError("out must be a C-contiguous float64 array of shape (nw,) or (L,nw)")
        if store is not None and (store.dtype != np.float64 or not store.flags.c_contiguous or store.size != L * 4 * N):
            raise ValueError("store must be a C-contiguous float64 array of shape (4,N) or (L,4,N)")
        rc = self._lib.ansfm_add_line_set_monochromatic_absorption(
            self._ctx, wn_grid.shape[0], _ptr(wn_grid), int(lineshape_id), L, _ptr(t), float(t_ref), _ptr(p), float(p_ref),
            _ptr(q), float(isotopic_abundance), float(isotopic_mass), M, _ptr(mmf), N, _ptr(bp), _ptr(nu), _ptr(_np(sw)),
            _ptr(_np(e_lower)), _ptr(_np(stimulated_emission_at_t_ref)), _ptr(out), _ptr(store), float(s_floor),
            float(wn_calc_window), float(wn_approx_window))
        self._check(rc, "add_line_set_monochromatic_absorption")
        return out

    def layer_average(self, RADIUS, H, P, T, ID, VMR, DUST, PARAH2, BASEH, BASEP=None, LAYANG=0.0, LAYINT=0, LAYHT=0.0,
                      NINT=101, DUST_UNITS=None, XMOLWT=None):
        """Layer_0.layer_average (:755), same argument order (ID and BASEP are unused there too).  A leading
        state axis on H/P/T/VMR/DUST/PARAH2/BASEH batches the call.  Returns
        HEIGHT,PRESS,TEMP,TOTAM,AMOUNT,PP,CONT,FRAC,DELH,BASET,LAYSF."""
        return self._layer_average(False, RADIUS, H, P, T, VMR, DUST, PARAH2, BASEH, LAYANG, LAYINT, LAYHT, NINT,
                                   DUST_UNITS, XMOLWT)

    def layer_averageg(self, RADIUS, H, P, T, ID, VMR, DUST, PARAH2, BASEH, BASEP=None, LAYANG=0.0, LAYINT=0, LAYHT=0.0,
                       NINT=101, DUST_UNITS=None, XMOLWT=None):
        """Layer_0.layer_averageg (:1032): layer_average's outputs followed by DTE, DAM, DCO, DPH (NLAY, NPRO)."""
        return self._layer_average(True, RADIUS, H, P, T, VMR, DUST, PARAH2, BASEH, LAYANG, LAYINT, LAYHT, NINT,
                                   DUST_UNITS, XMOLWT)

    def _layer_average(self, grad, RADIUS, H, P, T, VMR, DUST, PARAH2, BASEH, LAYANG, LAYINT, LAYHT, NINT, DUST_UNITS, XMOLWT):
        H = _np(H); single = H.ndim == 1
        H2 = np.atleast_2d(H); n, NPRO = H2.shape
        P2 = _np(P).reshape(n, NPRO); T2 = _np(T).reshape(n, NPRO)
        V2 = _np(VMR).reshape(n, NPRO, -1); NV = V2.shape[2]
        D2 = None if DUST is None else _np(DUST).reshape(n, NPRO, -1)
        ND = 0 if D2 is None else D2.shape[2]
        PH = None if PARAH2 is None else _np(PARAH2).reshape(n, NPRO)
        XM = None if XMOLWT is None else _np(np.broadcast_to(_np(XMOLWT).reshape(-1, NPRO), (n, NPRO)))
        BH = _np(np.broadcast_to(_np(BASEH).reshape(-1, np.shape(BASEH)[-1]), (n, np.shape(BASEH)[-1]))); NL = BH.shape[1]
        mk = lambda *shape: np.empty(shape)
        HEIGHT, PRESS, TEMP, TOTAM, FRAC, DELH, BASET, LAYSF = (mk(n, NL) for _ in range(8))
        AMOUNT, PPo, CONT = mk(n, NL, NV), mk(n, NL, NV), mk(n, NL, ND)
        args = [self._ctx, n, float(RADIUS), NPRO, _ptr(H2), _ptr(P2), _ptr(T2), NV, _ptr(V2), ND, _ptr(D2), _ptr(PH), NL, _ptr(BH),
                float(LAYANG), int(LAYINT), float(LAYHT), int(NINT), _ptr(_np(DUST_UNITS, np.int32)), _ptr(XM), _ptr(HEIGHT),
                _ptr(PRESS), _ptr(TEMP), _ptr(TOTAM), _ptr(AMOUNT), _ptr(PPo), _ptr(CONT), _ptr(FRAC), _ptr(DELH), _ptr(BASET),
                _ptr(LAYSF)]
        out = (HEIGHT, PRESS, TEMP, TOTAM, AMOUNT, PPo, CONT, FRAC, DELH, BASET, LAYSF)
        if grad:
            M = tuple(mk(n, NL, NPRO) for _ in range(4))
            self._check(self._lib.ansfm_layer_averageg(*args, *(_ptr(m) for m in M)), "layer_averageg")
            out = out + M
        else:
            self._check(self._lib.ansfm_layer_average(*args), "layer_average")
        return tuple(a[0] for a in out) if single else out

    def layer_average_dev(self, RADIUS, H, P, T, VMR, DUST, PARAH2, BASEH, LAYANG=0.0, LAYINT=0, LAYHT=0.0, NINT=101,
                          DUST_UNITS=None, XMOLWT=None):
        """Layer_0.layer_average for n states whose profiles are torch device tensors -- H, P, T (n, NPRO), VMR (n, NPRO,
        NVMR), DUST (n, NPRO, NDUST) / PARAH2 / XMOLWT (n, NPRO) or None, BASEH (n, NLAY), all float64 and contiguous --
        with the layers left in HBM: dict of device tensors HEIGHT .. LAYSF (n, NLAY), AMOUNT, PP (n, NLAY, NVMR), CONT
        (n, NLAY, NDUST), views of one buffer.  Asynchronous on the engine's stream."""
        import torch
        n, NPRO = H.shape
        NV = VMR.shape[2]; NL = BASEH.shape[1]
        ND = 0 if DUST is None else DUST.shape[2]
        for a in (H, P, T, VMR, DUST, PARAH2, XMOLWT, BASEH):
            if a is not None and (a.dtype != torch.float64 or not a.is_contiguous() or not a.is_cuda or a.shape[0] != n):
                raise ValueError("layer_average_dev: contiguous float64 device tensors with the state axis first")
        nl = n * NL
        buf = torch.empty(nl * (8 + 2 * NV + ND), dtype=torch.float64, device=H.device)
        rc = self._lib.ansfm_layer_average_dev(self._ctx, n, float(RADIUS), NPRO, _ptr(H), _ptr(P), _ptr(T), NV, _ptr(VMR), ND,
                                               _ptr(DUST), _ptr(PARAH2), NL, _ptr(BASEH), float(LAYANG), int(LAYINT), float(LAYHT),
                                               int(NINT), _ptr(_np(DUST_UNITS, np.int32)), _ptr(XMOLWT), _ptr(buf))
        self._check(rc, "layer_average_dev")
        out = {name: buf[k * nl:(k + 1) * nl].view(n, NL)
               for k, name in enumerate(("HEIGHT", "PRESS", "TEMP", "TOTAM", "FRAC", "DELH", "BASET", "LAYSF"))}
        o = 8 * nl
        out["AMOUNT"] = buf[o:o + nl * NV].view(n, NL, NV); o += nl * NV
        out["PP"] = buf[o:o + nl * NV].view(n, NL, NV); o += nl * NV
        out["CONT"] = buf[o:o + nl * ND].view(n, NL, ND)
        return out

    def map2pro(self, dSPECIN, NWAVE, NVMR, NDUST, NPRO, NPATH, NLAYIN, LAYINC, DTE, DAM, DCO, INCPAR=(-1,), to_host=True):
        """ForwardModel_0.map2pro (:5319), same arguments -> dSPECOUT (NWAVE, NVMR+2+NDUST, NPRO, NPATH).
        When dSPECIN is the very array the last cirsradg_ck_thermal call returned, its device copy is used
        (no host->device transfer); the result stays on the device for a following map2xvec.  dSPECIN = None: the
        gradients a `cirsradg_ck_thermal(..., gradients_on_device=True)` call left on the device.  to_host = False: the
        result is not copied back either (None is returned; `map2xvec(None, ...)` continues from the device)."""
        dev_in = dSPECIN is None
        if dev_in:
            ch = getattr(self, "_chain_dspec", None)
            if not (isinstance(ch, tuple) and ch[0] == "device"):
                raise ValueError("map2pro: no device-resident cirsradg result (gradients_on_device=True) to continue from")
            W, NPAR, LIMAX, P = ch[1:]
        else:
            dSPECIN = _np(dSPECIN)
            W, NPAR, LIMAX, P = dSPECIN.shape
        if W != NWAVE or NPAR != NVMR + 2 + NDUST or P != NPATH:
            raise ValueError("map2pro: dSPECIN must be (NWAVE, NVMR+2+NDUST, NLAYIN, NPATH)")
        LAYINC = _np(LAYINC, np.int32)
        if LAYINC.ndim == 1:
            LAYINC = LAYINC[:, None]
        if LAYINC.shape != (LIMAX, P):
            raise ValueError("shapes (%d,%d) and %s not aligned: LAYINC rows must equal dSPECIN's layer axis"
                             % (W, LIMAX, LAYINC.shape))
        DTE = _np(DTE); DAM = _np(DAM); DCO = _np(DCO)
        NLAY = DTE.shape[0]
        inc = None if INCPAR[0] == -1 else _np(list(INCPAR), np.int32)
        if inc is not None and inc[0] > NVMR + NDUST:
            raise UnboundLocalError("local variable 'dSPECOUT1' referenced before assignment")   # as the reference
        out = np.empty((W, NPAR, NPRO, P)) if to_host else None
        chained = dev_in or (getattr(self, "_chain_dspec", None) is not None and self._chain_dspec == _fingerprint(dSPECIN))
        rc = self._lib.ansfm_map2pro(self._ctx, W, NPAR, LIMAX, P, int(NPRO), NLAY, int(NVMR), int(NDUST),
                                     None if chained else _ptr(dSPECIN), _ptr(LAYINC), _ptr(DTE), _ptr(DAM), _ptr(DCO),
                                     0 if inc is None else len(inc), _ptr(inc), _ptr(out))
        self._check(rc, "map2pro")
        if out is None:
            self._chain_map = ("device", W, NPAR, int(NPRO), P)
            return None
        out.flags.writeable = False      # its device twin feeds map2xvec
        self._chain_map = _fingerprint(out)
        return out

    def map2xvec(self, dSPECIN, NWAVE, NVMR, NDUST, NPRO, NPATH, NX, xmap):
        """ForwardModel_0.map2xvec (:5387), same arguments -> dSPECOUT (NWAVE, NPATH, NX).  dSPECIN = None: the result a
        `map2pro(..., to_host=False)` call left on the device."""
        if dSPECIN is None:
            ch = getattr(self, "_chain_map", None)
            if not (isinstance(ch, tuple) and ch[0] == "device"):
                raise ValueError("map2xvec: no device-resident map2pro result (to_host=False) to continue from")
            W, NPAR, NPROi, P = ch[1:]
        else:
            dSPECIN = _np(dSPECIN)
            W, NPAR, NPROi, P = dSPECIN.shape
        xmap = _np(xmap)
        if xmap.shape != (NX, NPAR, NPROi):
            raise ValueError("shape-mismatch for sum")     # np.tensordot's message
        out = np.empty((W, P, NX))
        chained = dSPECIN is None or (getattr(self, "_chain_map", None) is not None and self._chain_map == _fingerprint(dSPECIN))
        rc = self._lib.ansfm_map2xvec(self._ctx, W, NPAR, NPROi, P, int(NX), None if chained else _ptr(dSPECIN),
                                      _ptr(xmap), _ptr(out))
        self._check(rc, "map2xvec")
        return out

    # ---- continuum -----------------------------------------------------------------------------------------------
    def calc_tau_cia(self, ISPACE, WAVEC, CIA_WAVEN, CIA_TEMP, CIA_FRAC, NPARA, K_CIA, IPAIRG1, IPAIRG2, INORMALT, INORMAL,
                     INORMALD, ID, ISO, PP, PRESS, TEMP, FRAC, TOTAM, DELH, k_co2=None, k_n2n2=None, k_n2h2=None, with_grad=True):
        """ForwardModel_0.calc_tau_cia (:4516) with the reference's objects flattened into arrays (CIA.WAVEN/TEMP/FRAC/
        NPARA/K_CIA/IPAIRG1/IPAIRG2/INORMALT/INORMAL, CIA.locate_INORMAL_pairs(), Atmosphere.ID/ISO, Layer.PP/PRESS/TEMP/
        FRAC/TOTAM/DELH) and co2cia / n2n2cia / n2h2cia(WAVEN) as vectors.  -> TAUCIA (NWAVE,NLAY), dTAUCIA (NWAVE,NLAY,NVMR+2)."""
        WAVEC = _np(WAVEC); ID = np.asarray(ID); ISO = np.asarray(ISO)
        NVMR = ID.size
        if int(ISPACE) == 0:
            WAVEN, isort = WAVEC, None
        else:                                              # :4565-4568
            WAVEN = 1.e4 / WAVEC; isort = np.argsort(WAVEN); WAVEN = _np(WAVEN[isort])
        K = _np(K_CIA); NPAIR, NPE, NT, NWC = K.shape
        g1 = np.full(NPAIR, -1, np.int32); g2 = np.full(NPAIR, -1, np.int32)
        for ip in range(NPAIR):                            # :4676-4701
            a = np.where(ID == int(IPAIRG1[ip]))[0]; b = np.where(ID == int(IPAIRG2[ip]))[0]
            if len(a) > 1: a = np.where((ID == int(IPAIRG1[ip])) & (ISO == 1))[0]
            if len(b) > 1: b = np.where((ID == int(IPAIRG2[ip])) & (ISO == 1))[0]
            if len(a) == 1 and len(b) == 1 and not (INORMALD[ip] and int(INORMALT[ip]) != int(INORMAL)):
                g1[ip], g2[ip] = a[0], b[0]
        ico2 = ih2 = in2 = -1                              # :4545-4561
        for i in range(NVMR):
            if ID[i] == 39 and ISO[i] in (0, 1): ih2 = i
            if ID[i] == 22: in2 = i
            if ID[i] == 2 and ISO[i] in (0, 1): ico2 = i
        q = _np((_np(PP).T / _np(PRESS)).T)
        L = q.shape[0]
        xfac = _np((_np(TOTAM) * 1.0e-4) ** 2. / (_np(DELH) * 1.0e2))
        frac = _np(np.asarray(CIA_FRAC, float).reshape(-1))
        W = WAVEN.size
        tau = np.empty((W, L)); dtau = np.empty((W, L, NVMR + 2)) if with_grad else None
        rc = self._lib.ansfm_calc_tau_cia(self._ctx, W, _ptr(WAVEN), NWC, _ptr(_np(CIA_WAVEN)), NPAIR, NPE, NT, _ptr(K),
                                          _ptr(_np(CIA_TEMP)), frac.size, _ptr(frac), int(NPARA), _ptr(g1), _ptr(g2), L, NVMR,
                                          _ptr(_np(TEMP)), _ptr(_np(FRAC)), _ptr(q), _ptr(xfac), ico2, _ptr(_np(k_co2)), in2,
                                          _ptr(_np(k_n2n2)), ih2, _ptr(_np(k_n2h2)), _ptr(tau), _ptr(dtau))
        self._check(rc, "calc_tau_cia")
        if isort is not None:                              # :4741-4743
            tau = tau[isort, :]
            dtau = None if dtau is None else dtau[isort, :, :]
        return (tau, dtau) if with_grad else tau

    def calc_tau_rayleigh(self, IRAY, ISPACE, WAVEC, TOTAM, ID=None, ISO=None, VMR=None, variant=None):
        """ForwardModel_0.calc_tau_rayleigh (:4869): IRAY 0 -> zeros, 1 -> calc_tau_rayleighj, 2 -> calc_tau_rayleighv2,
        4 -> calc_tau_rayleighls (needs ID, ISO, VMR (NLAY, NVMR)); variant='v' -> calc_tau_rayleighv (:5598).
        -> TAURAY (NWAVE, NLAY), dTAURAY (NWAVE, NLAY)."""
        WAVEC = _np(WAVEC); TOTAM = _np(TOTAM)
        W, L = WAVEC.size, TOTAM.size
        IRAY = int(IRAY)
        if variant is None and IRAY == 0:
            return np.zeros((W, L)), np.zeros((W, L))
        mode = 12 if variant == "v" else IRAY
        if mode not in (1, 2, 4, 12):
            raise ValueError("error in CIRSrad :: IRAY = " + str(IRAY) + " type has not been implemented yet")
        f4 = None
        if mode == 4:
            ID = np.asarray(ID); ISO = np.asarray(ISO); VMR = _np(VMR)
            f4 = np.zeros((L, 4))
            for j in range(ID.size):                               # :5748-5767 (the last matching gas wins)
                if ISO[j] in (0, 1):
                    col = {39: 0, 40: 1, 6: 2, 11: 3}.get(int(ID[j]))
                    if col is not None:
                        f4[:, col] = VMR[:, j]
            f4 = _np(f4)
        tau = np.empty((W, L)); dtau = np.empty((W, L))
        rc = self._lib.ansfm_calc_tau_rayleigh(self._ctx, mode, int(ISPACE), W, _ptr(WAVEC), L, _ptr(TOTAM), _ptr(f4), _ptr(tau),
                                               _ptr(dtau))
        self._check(rc, "calc_tau_rayleigh")
        return tau, dtau

    def calc_tau_rayleigh_batch_dev(self, IRAY, ISPACE, TOTAM, out, ID=None, ISO=None, VMR=None, variant=None):
        """calc_tau_rayleigh (:4869) for the n states of a batch on the uploaded table's wavenumber grid, left in HBM:
        TOTAM (n, NLAY) host, VMR (n, NLAY, NVMR) host for IRAY 4, out = torch device tensor (n, NWAVE, NLAY) float64."""
        mode = 12 if variant == "v" else int(IRAY)
        if mode not in (1, 2, 4, 12):
            raise ValueError("error in CIRSrad :: IRAY = " + str(IRAY) + " type has not been implemented yet")
        if not isinstance(TOTAM, np.ndarray) and hasattr(TOTAM, "data_ptr"):      # torch device tensors (layer_average_dev)
            import torch
            n, L = TOTAM.shape
            f4 = self.rayleigh_f4(ID, ISO, VMR) if mode == 4 else None
            if tuple(out.shape)[0] != n or tuple(out.shape)[2] != L or not out.is_contiguous() or not TOTAM.is_contiguous():
                raise ValueError("out must be a contiguous (n, NWAVE, NLAY) float64 device tensor")
            rc = self._lib.ansfm_calc_tau_rayleigh_batch_dev_in(self._ctx, mode, int(ISPACE), n, L, _ptr(TOTAM), _ptr(f4), _ptr(out))
            self._check(rc, "calc_tau_rayleigh_batch_dev")
            self._keep = (TOTAM, f4)          # alive until the next call: the kernel may still be queued
            return out
        TOTAM = _np(TOTAM)
        n, L = TOTAM.shape
        f4 = None
        if mode == 4:
            ID = np.asarray(ID); ISO = np.asarray(ISO); VMR = _np(VMR).reshape(n, L, -1)
            f4 = np.zeros((n, L, 4))
            for j in range(ID.size):                               # :5748-5767 (the last matching gas wins)
                if ISO[j] in (0, 1):
                    col = {39: 0, 40: 1, 6: 2, 11: 3}.get(int(ID[j]))
                    if col is not None:
                        f4[:, :, col] = VMR[:, :, j]
            f4 = _np(f4)
        if tuple(out.shape)[0] != n or tuple(out.shape)[2] != L or not out.is_contiguous():
            raise ValueError("out must be a contiguous (n, NWAVE, NLAY) float64 device tensor")
        rc = self._lib.ansfm_calc_tau_rayleigh_batch_dev(self._ctx, mode, int(ISPACE), n, L, _ptr(TOTAM), _ptr(f4), _ptr(out))
        self._check(rc, "calc_tau_rayleigh_batch_dev")
        return out

    def calc_tau_dust(self, WAVEC, SWAVE, KEXT, KSCA, CONT):
        """ForwardModel_0.calc_tau_dust (:4790): KEXT / KSCA (NWAVE_scatter, NDUST) on SWAVE, CONT (NLAY, NDUST) ->
        TAUDUST, TAUCLSCAT, dTAUDUSTdq, dTAUCLSCATdq (NWAVE, NLAY, NDUST)."""
        WAVEC = _np(WAVEC); SWAVE = _np(SWAVE); KEXT = _np(KEXT); KSCA = _np(KSCA); CONT = _np(CONT)
        W, L, ND = WAVEC.size, CONT.shape[0], CONT.shape[1]
        if KEXT.shape != (SWAVE.size, ND) or KSCA.shape != KEXT.shape:
            raise ValueError("KEXT / KSCA must be (len(Scatter.WAVE), NDUST)")
        out = [np.empty((W, L, ND)) for _ in range(4)]
        if ND == 0:
            return tuple(out)
        rc = self._lib.ansfm_calc_tau_dust(self._ctx, W, _ptr(WAVEC), SWAVE.size, _ptr(SWAVE), ND, _ptr(KEXT), _ptr(KSCA), L,
                                           _ptr(CONT), *[_ptr(o) for o in out])
        self._check(rc, "calc_tau_dust")
        return tuple(out)

    def kdist_bins(self, wavecalc, kabs, vbinmin, vbinmax, g_ord, fil=None):
        """Numerical core of Spectroscopy_0.calc_ktable_chunk (:3620-3652): k-distribution of the line-by-line spectrum
        kabs on the uniform grid wavecalc inside each bin [vbinmin, vbinmax], read at the g-ordinates.  fil = (bin
        centres, nfil (nbin), dfil (NF, nbin) = VFIL - VCONV, afil (NF, nbin)) weights the points with the instrument
        function, None = equal weights.  -> (nbin, NG)."""
        wavecalc = _np(wavecalc); kabs = _np(kabs); vbinmin = _np(vbinmin); vbinmax = _np(vbinmax); g_ord = _np(g_ord)
        nbin, NG = vbinmin.size, g_ord.size
        out = np.empty((nbin, NG))
        if fil is None:
            args = (None, 0, None, None, None)
        else:
            wcen, nfil, dfil, afil = _np(fil[0]), _np(fil[1], np.int32), _np(fil[2]), _np(fil[3])
            if dfil.shape != afil.shape or dfil.shape[1] != nbin or nfil.shape != (nbin,) or wcen.shape != (nbin,):
                raise ValueError("fil = (centres (nbin), nfil (nbin), dfil (NF, nbin), afil (NF, nbin))")
            args = (_ptr(wcen), dfil.shape[0], _ptr(nfil), _ptr(dfil), _ptr(afil))
        rc = self._lib.ansfm_kdist_bins(self._ctx, wavecalc.size, _ptr(wavecalc), _ptr(kabs), nbin, _ptr(vbinmin), _ptr(vbinmax),
                                        *args, NG, _ptr(g_ord), _ptr(out))
        self._check(rc, "kdist_bins")
        return out

    # ---- instrument line shape ---------------------------------------------------------------------------------
    def lblconv(self, nwave, vwave, y, nconv, vconv, ishape, fwhm):
        """Measurement_0.lblconv (:3335), one geometry: y (nwave) -> yout (nconv)."""
        return self._lblconv(vwave, y, None, nconv, vconv, ishape, fwhm)[0]

    def lblconvg(self, nwave, vwave, y, dydx, nconv, vconv, ishape, fwhm):
        """Measurement_0.lblconvg (:3799), one geometry: -> yout (nconv), gradout (nconv, nx)."""
        return self._lblconv(vwave, y, dydx, nconv, vconv, ishape, fwhm)

    def lblconv_fil(self, nwave, vwave, y, nconv, vconv, nfil, vfil, afil):
        """Measurement_0.lblconv_fil (:3549)."""
        return self._lblconv(vwave, y, None, nconv, vconv, None, None, (nfil, vfil, afil))[0]

    def lblconvg_fil(self, nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil):
        """Measurement_0.lblconvg_fil (:3992)."""
        return self._lblconv(vwave, y, dydx, nconv, vconv, None, None, (nfil, vfil, afil))

    def lblconv_ngeom(self, nwave, vwave, y, nconv, vconv, ishape, fwhm):
        """Measurement_0.lblconv_ngeom (:3444): y (nwave, ngeom) -> yout (nconv, ngeom)."""
        return self._lblconv(vwave, y, None, nconv, vconv, ishape, fwhm, ngeom=True)[0]

    def lblconvg_ngeom(self, nwave, vwave, y, dydx, nconv, vconv, ishape, fwhm):
        """Measurement_0.lblconvg_ngeom (:3685): -> yout (nconv, ngeom), gradout (nconv, ngeom, nx)."""
        return self._lblconv(vwave, y, dydx, nconv, vconv, ishape, fwhm, ngeom=True)

    def lblconv_fil_ngeom(self, nwave, vwave, y, nconv, vconv, nfil, vfil, afil):
        """Measurement_0.lblconv_fil_ngeom (:3614)."""
        return self._lblconv(vwave, y, None, nconv, vconv, None, None, (nfil, vfil, afil), ngeom=True)[0]

    def lblconvg_fil_ngeom(self, nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil):
        """Measurement_0.lblconvg_fil_ngeom (:3912)."""
        return self._lblconv(vwave, y, dydx, nconv, vconv, None, None, (nfil, vfil, afil), ngeom=True)

    def conv_fil(self, vwave, y, dydx, nconv, vconv, nfil, vfil, afil):
        """FWHM < 0 branch of Measurement_0.conv (:2425-2461; dydx None) / convg (:2655-2691): filter average over the
        window bracketing each filter -> yout (nconv) [, gradout (nconv, nx)]."""
        r = self._lblconv(vwave, y, dydx, nconv, vconv, None, None, (nfil, vfil, afil), bracket=True)
        return r if dydx is not None else r[0]

    def integrate_filter(self, nwave, vwave, y, nconv, vconv, nfil, vfil, afil, dydx=None):
        """Measurement_0.integrate_filter (:4079) / integrate_filterg (:4188; dydx given) and, for y (nwave, ngeom) [dydx
        (nwave, ngeom, nx)], integrate_filter_ngeom (:4131) / integrate_filterg_ngeom (:4251)."""
        r = self._lblconv(vwave, y, dydx, nconv, vconv, None, None, (nfil, vfil, afil), ngeom=(np.ndim(y) == 2), integrate=True)
        return r if dydx is not None else r[0]

    def _lblconv(self, vwave, y, dydx, nconv, vconv, ishape, fwhm, fil=None, ngeom=False, bracket=False, integrate=False):
        vwave = _np(vwave); y = _np(y); vconv = _np(np.asarray(vconv)[:nconv])
        nd = 2 if ngeom else 1
        if y.ndim != nd or (dydx is not None and np.ndim(dydx) != nd + 1):
            raise ValueError("y (nwave) and dydx (nwave, nx), or the *_ngeom shapes y (nwave, ngeom), dydx (nwave, ngeom, nx)")
        dydx = None if dydx is None else _np(dydx)
        nx = 0 if dydx is None else dydx.shape[-1]
        ng = y.shape[1] if ngeom else 1
        if dydx is not None and ngeom and dydx.shape[1] != ng:
            raise ValueError("dydx (nwave, ngeom, nx) must have the NGEOM of y")
        yout = np.empty((nconv, ng)); gout = np.empty((nconv, ng, nx))
        if fil is not None:
            nfil = _np(np.asarray(fil[0])[:nconv], np.int32)
            vfil = _np(np.asarray(fil[1])[:, :nconv]); afil = _np(np.asarray(fil[2])[:, :nconv])
        if integrate:
            rc = self._lib.ansfm_integrate_filter(self._ctx, vwave.size, _ptr(vwave), ng, _ptr(y), nx, _ptr(dydx), int(nconv),
                                                  _ptr(vconv), vfil.shape[0], _ptr(nfil), _ptr(vfil), _ptr(afil), _ptr(yout),
                                                  _ptr(gout))
            self._check(rc, "integrate_filter")
            return (yout, gout) if ngeom else (yout[:, 0], gout[:, 0, :])
        if ngeom:
            if fil is None:
                rc = self._lib.ansfm_lblconv_ngeom(self._ctx, vwave.size, _ptr(vwave), ng, _ptr(y), nx, _ptr(dydx), int(nconv),
                                                   _ptr(vconv), int(ishape), float(fwhm), _ptr(yout), _ptr(gout))
            else:
                rc = self._lib.ansfm_lblconv_fil_ngeom(self._ctx, vwave.size, _ptr(vwave), ng, _ptr(y), nx, _ptr(dydx),
                                                       int(nconv), _ptr(vconv), vfil.shape[0], _ptr(nfil), _ptr(vfil),
                                                       _ptr(afil), _ptr(yout), _ptr(gout))
            self._check(rc, "lblconv_ngeom")
            return yout, gout
        if fil is None:
            rc = self._lib.ansfm_lblconv(self._ctx, vwave.size, _ptr(vwave), _ptr(y), nx, _ptr(dydx), int(nconv), _ptr(vconv),
                                         int(ishape), float(fwhm), _ptr(yout), _ptr(gout))
        else:
            fn = self._lib.ansfm_conv_fil if bracket else self._lib.ansfm_lblconv_fil
            rc = fn(self._ctx, vwave.size, _ptr(vwave), _ptr(y), nx, _ptr(dydx), int(nconv), _ptr(vconv),
                    vfil.shape[0], _ptr(nfil), _ptr(vfil), _ptr(afil), _ptr(yout), _ptr(gout))
        self._check(rc, "lblconv")
        return yout[:, 0], gout[:, 0, :]

    def get_taugas(self, L, model=0):
        W, G = self.dims[0], self.dims[1]
        out = np.empty((W, G, L))
        self._check(self._lib.ansfm_get_taugas(self._ctx, int(model), _ptr(out)), "get_taugas")
        return out

    def set_layer_dedup(self, enable=True):
        """Share the gas opacity of layers that are bit-identical to the first model's inside a batched call."""
        self._check(self._lib.ansfm_set_layer_dedup(self._ctx, int(bool(enable))), "set_layer_dedup")

    def set_merge_keys(self, bits=64):
        """Row-head keys of the forward k_overlap merge: 32 (float32 keys + exact tie branch) or 64 (double keys)."""
        self._check(self._lib.ansfm_set_merge_keys(self._ctx, int(bits)), "set_merge_keys")

    def merge_redo_count(self):
        """(wave, gas) merges the 32-bit-key kernel had to rerun in its exact mode since the last table upload."""
        n = C.c_int64(0)
        self._check(self._lib.ansfm_merge_redo_count(self._ctx, C.byref(n)), "merge_redo_count")
        return int(n.value)

    def last_rt_shared(self):
        """True when the last thermal-emission batch started its states' paths from state 0's records (ansfm_last_rt_shared)."""
        v = C.c_int(0)
        self._check(self._lib.ansfm_last_rt_shared(self._ctx, C.byref(v)), "last_rt_shared")
        return bool(v.value)

    def last_layer_rows(self):
        """(layer opacities computed, n_models * L) of the last cirsrad_ck_thermal call."""
        a = C.c_int(); b = C.c_int()
        self._check(self._lib.ansfm_last_layer_rows(self._ctx, C.byref(a), C.byref(b)), "last_layer_rows")
        return a.value, b.value

    def last_kernel_ms(self):
        a = C.c_double(); b = C.c_double(); na = C.c_int(); nb = C.c_int()
        self._check(self._lib.ansfm_last_kernel_ms(self._ctx, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)),
                    "last_kernel_ms")
        return {"overlap_ms": a.value, "overlap_launches": na.value, "rt_ms": b.value, "rt_launches": nb.value}
