"""ctypes loader for libansfm.so (the HIP/gfx950 engine behind include/ansfm.h).

The library is built in-tree (archnemesis_dist_amd/lib/libansfm.so) by `build()` /
`__graft_entry__.build()`.  There is NO CPU fallback: if the library is missing or no HIP device
is usable, this module raises -- product code never routes through the CPU oracle.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libansfm.so")
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

ANSFM_OK = 0
ERR_NAMES = {1: "ANSFM_ERR_INVALID", 2: "ANSFM_ERR_HIP", 3: "ANSFM_ERR_NOTABLE", 4: "ANSFM_ERR_UNSORTED",
             5: "ANSFM_ERR_UNSUPPORTED"}

# every symbol include/ansfm.h declares (tests check the library exports all of them)
EXPORTS = [
    "ansfm_abi_version", "ansfm_create", "ansfm_destroy", "ansfm_last_error", "ansfm_set_stream",
    "ansfm_synchronize", "ansfm_set_f32_semantics", "ansfm_upload_ktable", "ansfm_upload_ktable_dev", "ansfm_ktable_info",
    "ansfm_calc_k", "ansfm_k_overlap", "ansfm_thermal_emission", "ansfm_cirsrad_ck_thermal",
    "ansfm_cirsrad_ck_thermal_dev", "ansfm_get_taugas", "ansfm_last_kernel_ms",
    "ansfm_k_overlapg", "ansfm_cirsradg_ck_thermal", "ansfm_cirsradg_ck_thermal_dev", "ansfm_scloud11wave_core", "ansfm_upload_lbltable", "ansfm_calc_klbl", "ansfm_add_line_set_monochromatic_absorption", "ansfm_layer_average",
    "ansfm_map2pro", "ansfm_map2xvec", "ansfm_layer_averageg", "ansfm_lblconv", "ansfm_lblconv_fil", "ansfm_lblconv_ngeom", "ansfm_lblconv_fil_ngeom", "ansfm_conv_fil", "ansfm_integrate_filter", "ansfm_calc_tau_rayleigh", "ansfm_calc_tau_dust", "ansfm_set_layer_dedup", "ansfm_last_layer_rows",
    "ansfm_ktable_file_header", "ansfm_upload_ktable_files", "ansfm_ktable_grids", "ansfm_lbltable_file_header",
    "ansfm_upload_lbltable_files", "ansfm_kdist_bins", "ansfm_calc_tau_cia", "ansfm_set_merge_keys", "ansfm_merge_redo_count", "ansfm_calc_tau_rayleigh_batch_dev", "ansfm_cirsrad_ck_scatter", "ansfm_thermal_emission_g", "ansfm_cirsrad_ck_transmission", "ansfm_cirsradg_ck_transmission", "ansfm_set_gradient_gases", "ansfm_set_shared_gas_gradient", "ansfm_singlescatt_plane_spectrum",
    "ansfm_cirsrad_ck_singlescatt", "ansfm_cirsrad_ck_scatter_batch", "ansfm_last_scatter_cache",
    "ansfm_layer_average_dev", "ansfm_calc_tau_rayleigh_batch_dev_in", "ansfm_last_rt_shared",
    "ansfm_cirsrad_ck_thermal_ray_dev",
]

_lib = None


class AnsfmError(RuntimeError):
    pass


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into lib/libansfm.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, "ansfm_api.hip"), os.path.join(CSRC, "ansfm_kdist.hip"), os.path.join(CSRC, "ansfm_merge32.hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "ansfm.h")]
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(d) for d in deps)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    # one object per translation unit, compiled side by side, then one link
    from concurrent.futures import ThreadPoolExecutor
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility=hidden"]
    objs = [os.path.join(os.path.dirname(LIB_PATH), os.path.splitext(os.path.basename(src))[0] + ".o") for src in srcs]

    def compile_one(pair):
        src, obj = pair
        cmd = [hipcc_path()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(srcs)) as pool:
        list(pool.map(compile_one, zip(srcs, objs)))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so; libansfm.so is linked against the
    one under /opt/rocm.  Whichever is loaded first must serve both, or the second initialisation fails (seen as
    torch.cuda.is_available() == False after an engine was created).  If torch is installed and not yet imported, its
    copy is loaded first (by path, no `import torch`), so libansfm.so's DT_NEEDED entry resolves to it."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                                         # torch's runtime is already the process's runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                                       # fall back on the system runtime; torch must then be imported first


def load():
    """dlopen libansfm.so and declare prototypes.  Raises AnsfmError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AnsfmError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    lib.ansfm_abi_version.restype = ci
    lib.ansfm_create.argtypes = [ci, C.POINTER(vp)]
    lib.ansfm_destroy.argtypes = [vp]
    lib.ansfm_destroy.restype = None
    lib.ansfm_last_error.argtypes = [vp]
    lib.ansfm_last_error.restype = C.c_char_p
    lib.ansfm_set_stream.argtypes = [vp, vp]
    lib.ansfm_synchronize.argtypes = [vp]
    lib.ansfm_set_f32_semantics.argtypes = [vp, ci, ci]
    lib.ansfm_upload_ktable.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp]
    lib.ansfm_upload_ktable_dev.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp]
    lib.ansfm_ktable_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(ci)]
    lib.ansfm_calc_k.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.ansfm_k_overlap.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp]
    lib.ansfm_thermal_emission.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, cd, vp, vp, vp, cd, cd, vp]
    lib.ansfm_thermal_emission_g.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, ci, vp, vp, cd, vp, vp, vp, vp]
    cirs = [vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ansfm_cirsrad_ck_thermal.argtypes = cirs
    lib.ansfm_cirsrad_ck_thermal_dev.argtypes = cirs
    lib.ansfm_cirsrad_ck_thermal_ray_dev.argtypes = [vp, ci, ci, ci, vp, vp, vp, ci, vp, vp, ci, ci] + [vp] * 12
    lib.ansfm_cirsrad_ck_transmission.argtypes = [vp, ci, ci, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp]
    lib.ansfm_set_gradient_gases.argtypes = [vp, C.c_uint]
    lib.ansfm_set_shared_gas_gradient.argtypes = [vp, ci, vp]
    lib.ansfm_cirsradg_ck_transmission.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp]
    lib.ansfm_singlescatt_plane_spectrum.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, cd, vp, vp, vp, cd, cd, vp]
    lib.ansfm_cirsrad_ck_singlescatt.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp]
    lib.ansfm_get_taugas.argtypes = [vp, ci, vp]
    lib.ansfm_scloud11wave_core.argtypes = [vp, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, vp, ci, vp, ci, ci,
                                            vp, vp, vp, ci, ci, ci, vp, vp]
    lib.ansfm_cirsrad_ck_scatter.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, ci,
                                             vp, vp, ci, ci, ci, ci, vp, vp, vp]
    # (ctx, ISPACE, n_models, L, press, temp, amount, cia, dust, ray, scat, ncont, nth, phasarr, lfrac, radg, ngeom, sol, emi, aphi,
    #  solar, lowbc, brdf, nmu, mu1, wt1, nf, nphi, iray, imie, xfac, SPECOUT)
    lib.ansfm_cirsrad_ck_scatter_batch.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci,
                                                   vp, ci, vp, vp, ci, ci, ci, ci, vp, vp]
    lib.ansfm_last_scatter_cache.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.ansfm_upload_lbltable.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, ci, vp]
    lib.ansfm_calc_klbl.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.ansfm_add_line_set_monochromatic_absorption.argtypes = [vp, ci, vp, ci, ci, vp, cd, vp, cd, vp, cd, cd, ci, vp, ci, vp,
                                                                vp, vp, vp, vp, vp, vp, cd, cd, cd]
    lib.ansfm_layer_average.argtypes = [vp, ci, cd, ci, vp, vp, vp, ci, vp, ci, vp, vp, ci, vp, cd, ci, cd, ci, vp, vp] + [vp] * 11
    lib.ansfm_layer_average_dev.argtypes = [vp, ci, cd, ci, vp, vp, vp, ci, vp, ci, vp, vp, ci, vp, cd, ci, cd, ci, vp, vp, vp]
    lib.ansfm_layer_averageg.argtypes = [vp, ci, cd, ci, vp, vp, vp, ci, vp, ci, vp, vp, ci, vp, cd, ci, cd, ci, vp, vp] + [vp] * 15
    lib.ansfm_calc_tau_cia.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, vp, vp, ci, vp, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, vp, ci, vp,
                                       ci, vp, vp, vp]
    lib.ansfm_ktable_file_header.argtypes = [C.c_char_p, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ansfm_upload_ktable_files.argtypes = [vp, ci, C.POINTER(C.c_char_p), cd, cd]
    lib.ansfm_ktable_grids.argtypes = [vp, vp, vp, vp, vp]
    lib.ansfm_kdist_bins.argtypes = [vp, ci, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp]
    lib.ansfm_lbltable_file_header.argtypes = [C.c_char_p, vp, vp, vp, vp, vp, vp]
    lib.ansfm_upload_lbltable_files.argtypes = [vp, ci, C.POINTER(C.c_char_p), cd, cd]
    lib.ansfm_set_layer_dedup.argtypes = [vp, ci]
    lib.ansfm_set_merge_keys.argtypes = [vp, ci]
    lib.ansfm_merge_redo_count.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.ansfm_last_layer_rows.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    lib.ansfm_lblconv.argtypes = [vp, ci, vp, vp, ci, vp, ci, vp, ci, cd, vp, vp]
    lib.ansfm_lblconv_fil.argtypes = [vp, ci, vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp]
    lib.ansfm_calc_tau_rayleigh.argtypes = [vp, ci, ci, ci, vp, ci, vp, vp, vp, vp]
    lib.ansfm_calc_tau_rayleigh_batch_dev.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp]
    lib.ansfm_calc_tau_rayleigh_batch_dev_in.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp]
    lib.ansfm_last_rt_shared.argtypes = [vp, vp]
    lib.ansfm_calc_tau_dust.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp]
    lib.ansfm_integrate_filter.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp]
    lib.ansfm_conv_fil.argtypes = [vp, ci, vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp]
    lib.ansfm_lblconv_ngeom.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, cd, vp, vp]
    lib.ansfm_lblconv_fil_ngeom.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp]
    lib.ansfm_map2pro.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp]
    lib.ansfm_map2xvec.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp]
    lib.ansfm_k_overlapg.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    cirsg = [vp, ci, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ansfm_cirsradg_ck_thermal.argtypes = cirsg
    lib.ansfm_cirsradg_ck_thermal_dev.argtypes = cirsg
    lib.ansfm_last_kernel_ms.argtypes = [vp, C.POINTER(cd), C.POINTER(ci), C.POINTER(cd), C.POINTER(ci)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("ansfm_destroy", "ansfm_last_error"):
            fn.restype = ci
    _lib = lib
    return lib


def read_ktable_header(path):
    """Spectroscopy_0.read_ktahead (:2492) through the native reader (no GPU needed):
    nwave, wave, fwhm, npress, ntemp, ng, gasID, isoID, g_ord, del_g, presslevels, templevels."""
    import numpy as np
    lib = load()
    dims = (C.c_int64 * 4)(); ids = (C.c_int32 * 2)(); hdr = (C.c_double * 3)()
    if lib.ansfm_ktable_file_header(os.fsencode(path), dims, ids, hdr, None, None, None, None, None) != ANSFM_OK:
        raise ValueError("not a readable .kta table: %s" % path)
    nwave, ng, npress, ntemp = (int(d) for d in dims)
    wave = np.empty(nwave); g_ord = np.empty(ng, np.float32); del_g = np.empty(ng, np.float32)
    press = np.empty(npress, np.float32); temp = np.empty(ntemp, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.ansfm_ktable_file_header(os.fsencode(path), dims, ids, hdr, p(wave), p(g_ord), p(del_g), p(press), p(temp))
    return nwave, wave, float(hdr[2]), npress, ntemp, ng, int(ids[0]), int(ids[1]), g_ord, del_g, press, temp


def read_lbltable_header(path):
    """Spectroscopy_0.read_ltahead (:2451) through the native reader (no GPU needed):
    nwave, vmin, delv, npress, ntemp, gasID, isoID, presslevels, templevels  (+ the wavenumber grid as a 10th item)."""
    import numpy as np
    lib = load()
    dims = (C.c_int64 * 3)(); ids = (C.c_int32 * 2)(); hdr = (C.c_double * 2)()
    if lib.ansfm_lbltable_file_header(os.fsencode(path), dims, ids, hdr, None, None, None) != ANSFM_OK:
        raise ValueError("not a readable .lta table: %s" % path)
    nwave, npress, ntemp = (int(d) for d in dims)
    # NT < 0: one grid of -NT temperatures per pressure level (the reference's reader returns them as (npress, -NT), :2480-2483)
    tshape = (npress, -ntemp) if ntemp < 0 else (ntemp,)
    wave = np.empty(nwave); press = np.empty(npress, np.float32); temp = np.empty(tshape, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.ansfm_lbltable_file_header(os.fsencode(path), dims, ids, hdr, p(wave), p(press), p(temp))
    return nwave, float(hdr[0]), float(hdr[1]), npress, ntemp, int(ids[0]), int(ids[1]), press, temp, wave
