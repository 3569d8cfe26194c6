/*
 * ansfm.h -- C-ABI of libansfm.so: the MI355X (gfx950) forward-model engine for the
 * archNEMESIS radiative-transfer hot path.
 *
 * The reference (pure Python + numba) has no FFI; its seams are Python-level.
 * Every entry point below replaces one of those seams and cites it (paths relative to the
 * reference tree).  INTEGRATION.md shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *  - plain pointers + sizes, row-major (C order) float64 unless stated; int32 index arrays
 *  - enum-like ints carry the reference's IntEnum values (ISPACE: 0 wavenumber, 1 wavelength)
 *  - every function returns ANSFM_OK (0) or an error code; ansfm_last_error(ctx) gives the text
 *  - caller owns every array it passes; outputs are caller-allocated; the library never frees
 *    caller memory; device buffers (k-table, workspaces) are owned by the opaque ctx
 *  - one ctx per GPU; a ctx is not re-entrant, different ctx may be used from different threads
 *  - functions without suffix take HOST pointers (drop-in for the NumPy seams); `_dev` variants
 *    take DEVICE pointers (HBM-resident batches, what bench.py times) and run asynchronously on
 *    the ctx stream
 *  - there is NO CPU fallback: without a usable HIP device ansfm_create() fails
 */
#ifndef ANSFM_H
#define ANSFM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANSFM_ABI_VERSION 1

#define ANSFM_OK 0
#define ANSFM_ERR_INVALID 1     /* bad argument / shape                                  */
#define ANSFM_ERR_HIP 2         /* HIP runtime error (text in ansfm_last_error)          */
#define ANSFM_ERR_NOTABLE 3     /* k-table not uploaded                                  */
#define ANSFM_ERR_UNSORTED 4    /* k-distribution not non-decreasing in g*/
#define ANSFM_ERR_UNSUPPORTED 5 /* valid in the reference, not built yet                 */

#define ANSFM_MAX_NG 32

typedef struct ansfm_ctx ansfm_ctx;

int ansfm_abi_version(void);

/* Create / destroy the per-GPU context (device ordinal as in hipSetDevice). */
int ansfm_create(int device, ansfm_ctx **ctx);
void ansfm_destroy(ansfm_ctx *ctx);
const char *ansfm_last_error(const ansfm_ctx *ctx);

/* Use an external hipStream_t (e.g. torch's current stream) for all subsequent work; NULL
 * restores the ctx-owned stream. */
int ansfm_set_stream(ansfm_ctx *ctx, void *hip_stream);
int ansfm_synchronize(ansfm_ctx *ctx);

/* dtype semantics of the reference's NumPy arithmetic.  Spectroscopy_0.PRESS/TEMP ("grid") and DELG
 * are float32 arrays when the tables come from .kta files (read_ktahead, Spectroscopy_0.py:2544-2559);
 * NumPy (and numba) then evaluate np.log(PRESS[i]), phi-plo, thi-tlo, 1./(thi-tlo) (calc_k(g),
 * :2373-2389, :2236) and del_g[i]*del_g[j], np.cumsum(del_g) (k_overlap/rank, ForwardModel_0.py:6087,
 * :6142) in float32 -- a 4e-5 effect on tau.  Values are still passed as float64 (exactly the float32
 * values); these flags select the rounding.  Applies to every later call on this ctx. */
int ansfm_set_f32_semantics(ansfm_ctx *ctx, int grid_f32, int delg_f32);

/* ---- k-table ------------------------------------------------------------------------------
 * Replaces the state filled by Spectroscopy_0.read_tables (Spectroscopy_0.py:1448-1516):
 * K[W][G][NP][NT][S] (:213), PRESS[NP] (atm), TEMP[NT] (K), WAVE[W], DELG[G].
 * The table is re-laid out in HBM as ln k, [p][T][gas][g][wave] (wave fastest), float64.
 * `_dev`: K is a device pointer (same reference layout); the small vectors stay host pointers. */
int ansfm_upload_ktable(ansfm_ctx *ctx, int W, int G, int NP, int NT, int S, const double *K,
                        const double *PRESS, const double *TEMP, const double *WAVE,
                        const double *DELG);
int ansfm_upload_ktable_dev(ansfm_ctx *ctx, int W, int G, int NP, int NT, int S,
                            const double *K_dev, const double *PRESS, const double *TEMP,
                            const double *WAVE, const double *DELG);
/* dims = {W,G,NP,NT,S}; monotone = 1 when every k(g) column of the table is >=0 and
 * non-decreasing in g (precondition of the merge kernel's fast path). */
int ansfm_ktable_info(const ansfm_ctx *ctx, int64_t dims[5], int *monotone);

/* ---- LBL tables (ILBL = LINE_BY_LINE_TABLES) ------------------------------------------------------
 * State of Spectroscopy_0 after read_tables on .lta / HDF5 LBL tables: K[W][NP][|NT|][S] (NG = 1),
 * PRESS[NP] (atm), TEMP[|NT|] or -- temp2d != 0, the reference's NT < 0 -- TEMP[NP][|NT|] (one
 * temperature grid per pressure, Spectroscopy_0.py:1664-1669).  After this upload the fused
 * cirsrad entry points run the ILBL=2 branch of calculate_gaseous_line_opacity (ForwardModel_0.py
 * :3795-3817: calc_klbl(g), tau = sum_gas k * amount) instead of the k-distribution merge. */
int ansfm_upload_lbltable(ansfm_ctx *ctx, int W, int NP, int NT, int S, const double *K,
                          const double *PRESS, const double *TEMP, int temp2d, const double *WAVE);
/* Spectroscopy_0.calc_klbl (:1768) / calc_klblg (:1601, when dkdT_out != NULL; note its missing
 * it<0 clamp is reproduced): press[L] atm, temp[L] K -> k_out[W][L][S] (, dkdT_out[W][L][S]). */
int ansfm_calc_klbl(ansfm_ctx *ctx, int L, const double *press, const double *temp, double *k_out,
                    double *dkdT_out);

/* ---- native .kta reader ------------------------------------------------------------------
 * Spectroscopy_0.read_ktahead (Spectroscopy_0.py:2492) / read_ktable (:2733) / read_tables (:1448) for binary
 * k-tables: header irec0, nwave, vmin, delv, fwhm, npress, ntemp, ng, gasID, isoID (vmin / delv rounded to 7 decimals
 * like the reference), g_ord, del_g, two pad floats, P, T, the wavenumber list when delv <= 0, then k * 1e20 as
 * float32 [wave][press][temp][g] from record irec0.
 * ansfm_ktable_file_header needs no GPU: dims = {nwave, ng, npress, ntemp}, ids = {gasID, isoID},
 * hdr = {vmin, delv, fwhm}; the array outputs may be NULL.
 * ansfm_upload_ktable_files reads S tables that share one grid, cuts the wavenumbers to [wavemin, wavemax] with
 * read_tables' searchsorted rule, and streams each file's float32 block to the GPU where it is divided by 1e20 in
 * float32 (as NumPy does) and re-laid out -- the float64 K (NWAVE,NG,NP,NT,NGAS) array of the reference is never
 * formed on the host.  Sets the float32 semantics of ansfm_set_f32_semantics (the file's grids are float32 arrays in
 * the reference).  ansfm_ktable_grids returns the host copies of WAVE / PRESS / TEMP / DELG of the table in HBM. */
int ansfm_ktable_file_header(const char *path, int64_t dims[4], int32_t ids[2], double hdr[3], double *wave,
                             float *g_ord, float *del_g, float *press, float *temp);
int ansfm_upload_ktable_files(ansfm_ctx *ctx, int S, const char *const *paths, double wavemin, double wavemax);
int ansfm_ktable_grids(const ansfm_ctx *ctx, double *WAVE, double *PRESS, double *TEMP, double *DELG);
/* The same for binary LBL tables: Spectroscopy_0.read_ltahead (:2451) / read_lbltable (:2626) -- header irec0, nwave,
 * vmin, delv, npress, ntemp, gasID, isoID, P, T, then k * 1e20 as float32 [wave][press][temp]; the reference unpacks
 * them with a Python loop over (wavenumber, pressure), minutes for a 10^6-point table.  dims = {nwave, npress, ntemp},
 * hdr = {vmin, delv}.  Tables with one temperature grid per pressure level (ntemp < 0) are not streamed
 * (ANSFM_ERR_INVALID; read them with the reference and use ansfm_upload_lbltable).  After the upload the context is in
 * the LBL-table state of ansfm_upload_lbltable; ansfm_ktable_grids returns WAVE / PRESS / TEMP (DELG = {1}). */
int ansfm_lbltable_file_header(const char *path, int64_t dims[3], int32_t ids[2], double hdr[2], double *wave,
                               float *press, float *temp);
int ansfm_upload_lbltable_files(ansfm_ctx *ctx, int S, const char *const *paths, double wavemin, double wavemax);

/* The k-table GENERATOR's numerical core, Spectroscopy_0.calc_ktable_chunk (Spectroscopy_0.py:3620-3652): from a
 * line-by-line spectrum kabs[ncalc] on the uniform grid wavecalc[ncalc], for each of nbin bins [vbinmin, vbinmax] the
 * points inside are sorted by k, weighted by the instrument function (np.interp of afil over dfil = VFIL - VCONV at the
 * distance from the bin centre wcen; nfil == NULL: weight 1), g = cumsum(w dv) / sum(w dv), and k is read at the
 * g-ordinates: kout[nbin][NG] = np.interp(g_ord, g_sorted, k_sorted).  dfil / afil are [nfilmax][nbin] with nfil[bin]
 * valid rows.  A bin without points is ANSFM_ERR_INVALID (np.interp raises there).  The sort is one segmented radix
 * sort over all bins (rocPRIM). */
int ansfm_kdist_bins(ansfm_ctx *ctx, int ncalc, const double *wavecalc, const double *kabs, int nbin,
                     const double *vbinmin, const double *vbinmax, const double *wcen, int nfilmax,
                     const int32_t *nfil, const double *dfil, const double *afil, int NG, const double *g_ord,
                     double *kout);

/* ---- array-level seams (host pointers), one per numba/NumPy kernel of the reference ------- */

/* Spectroscopy_0.calc_k (Spectroscopy_0.py:2298) / calc_kg (:2147), WAVECALC=None.
 * press[L] in atm, temp[L] in K -> k_out[W][G][L][S]; dkdT_out may be NULL (calc_k). */
int ansfm_calc_k(ansfm_ctx *ctx, int L, const double *press, const double *temp, double *k_out,
                 double *dkdT_out);

/* ForwardModel_0.k_overlap (ForwardModel_0.py:6029): del_g[G], k[W][G][L][S], amount[S][L]
 * -> tau[W][G][L].  Does not need an uploaded table. */
int ansfm_k_overlap(ansfm_ctx *ctx, int W, int G, int L, int S, const double *del_g,
                    const double *k, const double *amount, double *tau);

/* ForwardModel_0.calc_thermal_emission_spectrum (ForwardModel_0.py:6287).
 * TAUTOT_PATH[W][G][Li], EMITOT_PATH[W][Li] or NULL, TEMP[Li], PRESS[Li], EMISSIVITY/SOLFLUX/
 * REFLECTANCE[W] -> SPECOUT[W][G]. */
int ansfm_thermal_emission(ansfm_ctx *ctx, int ISPACE, int W, int G, int NLAYIN,
                           const double *WAVE, const double *TAUTOT_PATH,
                           const double *EMITOT_PATH, const double *TEMP, const double *PRESS,
                           double TSURF, const double *EMISSIVITY, const double *SOLFLUX,
                           const double *REFLECTANCE, double SOL_ANG, double EMISS_ANG,
                           double *SPECOUT);

/* ForwardModel_0.calc_thermal_emission_spectrumg (ForwardModel_0.py:6380-6504), array level: TAUTOT_PATH[W][G][Li],
 * dTAUTOT_PATH[W][G][NPAR][Li], NVMR = index of the temperature parameter, TEMP / PRESS[Li], EMISSIVITY[W] (needed
 * when TSURF > 0) -> SPECOUT[W][G], dSPECOUT[W][G][NPAR][Li], dTSURF[W][G].  The reference's O(NPAR Li^2) recursion is
 * evaluated as one backward sweep (the fused CIRSrad entry below does the same, with the g-quadrature folded in). */
int ansfm_thermal_emission_g(ansfm_ctx *ctx, int ISPACE, int W, int G, int NPAR, int NLAYIN, const double *WAVE,
                             const double *TAUTOT_PATH, const double *dTAUTOT_PATH, int NVMR, const double *TEMP,
                             const double *PRESS, double TSURF, const double *EMISSIVITY, double *SPECOUT,
                             double *dSPECOUT, double *dTSURF);

/* ForwardModel_0.calc_singlescatt_plane_spectrum (ForwardModel_0.py:6509-6600), array level: plane-parallel thermal
 * emission + singly scattered sunlight (ssfac * omega * phase * SOLFLUX / 4 pi per layer) + lower boundary (always added) +
 * sunlight reflected by the ground (BRDF).  TAUTOT_PATH / OMEGA[W][G][Li], PHASE[W][Li], TEMP[Li], EMISSIVITY / BRDF /
 * SOLFLUX[W] -> SPECOUT[W][G]. */
int ansfm_singlescatt_plane_spectrum(ansfm_ctx *ctx, int ISPACE, int W, int G, int NLAYIN, const double *WAVE,
                                     const double *TAUTOT_PATH, const double *TEMP, const double *OMEGA, const double *PHASE,
                                     double TSURF, const double *EMISSIVITY, const double *BRDF, const double *SOLFLUX,
                                     double SOL_ANG, double EMISS_ANG, double *SPECOUT);

/* CIRSrad, single-scattering branch (IMOD & SINGLE_SCATTERING_PLANE_PARALLEL; ForwardModel_0.py:4493 ->
 * calculate_single_scattering_plane_parallel_spectrum :4251-4336) fused with the opacity assembly: the albedo
 * OMEGA = (TAURAY + TAUSCAT) / TAUTOT of every (wavenumber, g, layer) is formed in the RT kernel from the vertical opacities.
 * Host pointers: taucont[W][L] = TAUCIA + TAUDUST + TAURAY, tausca[W][L] = TAURAY + TAUSCAT, phase[P][W][L] = the layer-mean
 * phase function at each path's scattering angle (:4318-4322), BRDF[W][P], SOLFLUX[W], angles [P] -> SPECOUT[W][P]. */
int ansfm_cirsrad_ck_singlescatt(ansfm_ctx *ctx, int ISPACE, int L, const double *lay_press_pa, const double *lay_temp,
                                 const double *amount, const double *taucont, const double *tausca, const double *phase, int P,
                                 int LIMAX, const int32_t *NLAYIN, const int32_t *LAYINC, const double *SCALE,
                                 const double *EMTEMP, double TSURF, const double *EMISSIVITY, const double *BRDF,
                                 const double *SOLFLUX, const double *SOL_ANG, const double *EMISS_ANG, const double *xfac,
                                 double *SPECOUT);

/* ---- fused seam: CIRSrad, ILBL=K_TABLES, IMOD=THERMAL_EMISSION ------------------------------
 * ForwardModel_0.CIRSrad (ForwardModel_0.py:4376-4511) =
 *   calculate_gaseous_line_opacity (:3850-3877: calc_k -> k_overlap)
 *   -> calculate_layer_opacity (:3989 TAUTOT = TAUGAS+TAUCIA+TAUDUST+TAURAY; :4006 LAYINC*SCALE)
 *   -> calculate_thermal_emission_spectrum (:4216-4244) -> g-quadrature (:4504)
 * for a BATCH of n_models atmospheres sharing the uploaded k-table and the path structure
 * (the fan-out of jacobian_nemesis, ForwardModel_0.py:2305-2337).
 *
 *   lay_press_pa[n][L]   LayerX.PRESS (Pa; divided by 101325 internally, :3855)
 *   lay_temp[n][L]       LayerX.TEMP
 *   amount[n][S][L]      f_gas = LayerX.AMOUNT[:,IGAS]*1e-4 (cm-2, :3861)
 *   taucont[n][W][L]     TAUCIA+TAUDUST+TAURAY (vertical), or NULL (=0)
 *   NLAYIN[P], LAYINC[LIMAX][P] (int32), SCALE[n][LIMAX][P], EMTEMP[n][LIMAX][P]   PathX (:4006,:4218)
 *   TSURF[n]; EMISSIVITY/SOLFLUX/REFLECTANCE/xfac [W] or NULL (0,0,0,1); SOL_ANG/EMISS_ANG [P]
 *   SPECOUT[n][W][P]
 * n_models strides are contiguous.  Host-pointer version copies in/out; `_dev` takes device
 * pointers for every array (int32 arrays too) and is asynchronous on the ctx stream. */
int ansfm_cirsrad_ck_thermal(ansfm_ctx *ctx, int ISPACE, int n_models, int L,
                             const double *lay_press_pa, const double *lay_temp,
                             const double *amount, const double *taucont, int P, int LIMAX,
                             const int32_t *NLAYIN, const int32_t *LAYINC, const double *SCALE,
                             const double *EMTEMP, const double *TSURF, const double *EMISSIVITY,
                             const double *SOLFLUX, const double *REFLECTANCE,
                             const double *SOL_ANG, const double *EMISS_ANG, const double *xfac,
                             double *SPECOUT);
int ansfm_cirsrad_ck_thermal_dev(ansfm_ctx *ctx, int ISPACE, int n_models, int L,
                                 const double *lay_press_pa, const double *lay_temp,
                                 const double *amount, const double *taucont, int P, int LIMAX,
                                 const int32_t *NLAYIN, const int32_t *LAYINC,
                                 const double *SCALE, const double *EMTEMP, const double *TSURF,
                                 const double *EMISSIVITY, const double *SOLFLUX,
                                 const double *REFLECTANCE, const double *SOL_ANG,
                                 const double *EMISS_ANG, const double *xfac, double *SPECOUT);

/* The same for a batch whose only continuum is Rayleigh scattering (calc_tau_rayleigh, ForwardModel_0.py:4869; ray_mode =
 * IRAY 1, 2, 4, or 12 for calc_tau_rayleighv): TAURAY = k(wavenumber[, composition]) * TOTAM is formed inside the call from
 * the DEVICE arrays TOTAM[n][L] (m-2) and, ray_mode 4, f4[n][L][4] (mixing ratios of H2, He, CH4, NH3) -- once per DISTINCT
 * layer of the batch (TOTAM and f4 are part of a layer's identity next to pressure, temperature and amounts) and straight in
 * the layout the RT kernel reads.  The n * NWAVE * L array ansfm_calc_tau_rayleigh_batch_dev would fill (1.6 GB for the 201
 * states of BASELINE configs[2]) is neither written, transposed nor compared.  Same bits as the two-call route. */
int ansfm_cirsrad_ck_thermal_ray_dev(ansfm_ctx *ctx, int ISPACE, int n_models, int L, const double *lay_press_pa,
                                     const double *lay_temp, const double *amount, int ray_mode, const double *TOTAM,
                                     const double *f4, int P, int LIMAX, const int32_t *NLAYIN, const int32_t *LAYINC,
                                     const double *SCALE, const double *EMTEMP, const double *TSURF, const double *EMISSIVITY,
                                     const double *SOLFLUX, const double *REFLECTANCE, const double *SOL_ANG,
                                     const double *EMISS_ANG, const double *xfac, double *SPECOUT);

/* CIRSrad, pure-transmission branch (IMOD without ABSORBTION / THERMAL_EMISSION / scattering flags; ForwardModel_0.py:
 * 4478-4481 -> calculate_transmission_spectrum :4110-4131): SPECOUT[n][W][P] = xfac[W] * sum_g DELG[g] exp(-sum_layers
 * TAUTOT_LAYINC) -- the same opacity assembly as the thermal branch, the RT kernel's epilogue switched.  xfac = the
 * solar flux when IFORM = Atmospheric_transmission, else NULL.  (The reference's absorption branch,
 * calculate_absorption_spectrum :4133, is declared without `self` and cannot be called; it has no counterpart.) */
int ansfm_cirsrad_ck_transmission(ansfm_ctx *ctx, int n_models, int L, const double *lay_press_pa, const double *lay_temp,
                                  const double *amount, const double *taucont, int P, int LIMAX, const int32_t *NLAYIN,
                                  const int32_t *LAYINC, const double *SCALE, const double *xfac, double *SPECOUT);

/* The same branch with return_grad (calculate_transmission_spectrum :4128-4131 followed by CIRSrad's g-quadrature and
 * nan_to_num :4507): dSPECOUT[n][W][NPAR][LIMAX][P] = - sum_g DELG[g] xfac exp(-tau_path(g)) dTAUTOT_LAYINC[g], with
 * dTAUTOT_LAYINC assembled as in ansfm_cirsradg_ck_thermal (gas slots x 1e-4 through igas_map, temperature slot at NVMR,
 * dtaucon, x SCALE).  dTSURF of this branch is identically zero and not returned. */
int ansfm_cirsradg_ck_transmission(ansfm_ctx *ctx, int n_models, int L, const double *lay_press_pa,
                                   const double *lay_temp, const double *amount, const double *taucont,
                                   const double *dtaucon, int NVMR, int NPAR, const int32_t *igas_map, int P, int LIMAX,
                                   const int32_t *NLAYIN, const int32_t *LAYINC, const double *SCALE, const double *xfac,
                                   double *SPECOUT, double *dSPECOUT);

/* ---- analytic-gradient seams ---------------------------------------------------------------
 * ForwardModel_0.k_overlapg (ForwardModel_0.py:5842): + dkdT[W][G][L][S] -> tau[W][G][L],
 * dk[W][G][L][S+1] (slots 0..S-1 = d tau/d amount_gas, slot S = d tau/dT). */
int ansfm_k_overlapg(ansfm_ctx *ctx, int W, int G, int L, int S, const double *del_g,
                     const double *k, const double *dkdT, const double *amount, double *tau,
                     double *dk);

/* A continuum gradient that is the same for every gas parameter: dTAU_WL[W][L] (host) is added to dTAUCON[:, v, :] of all
 * v < NVMR in the NEXT ansfm_cirsradg_ck_thermal call of one model (then forgotten) -- what calculate_layer_opacity does with
 * dTAURAY (ForwardModel_0.py:3955-3957), without NVMR copies of the array crossing PCIe inside `dtaucon`.  NULL cancels. */
int ansfm_set_shared_gas_gradient(ansfm_ctx *ctx, int L, const double *dTAU_WL);

/* Which spectroscopic gases' amount gradients the k-table gradient path (ansfm_cirsradg_ck_*) computes: bit s of `mask` for
 * gas s of the uploaded table (default: all).  The reference always carries every gas through rankg (ForwardModel_0.py:
 * 5842-6026) and lets map2xvec drop what the state vector does not name; a caller that knows its state vector saves the replay
 * passes of the other gases (C2: 49 -> 21 passes for one gas).  The parameters of a gas that is switched off come back as the
 * continuum part only (dTAUCON), i.e. zero without one.  Bit 31: the temperature slot of the merge (d tau / dT through the
 * k-tables, two passes per merge); without it the temperature parameter keeps its continuum and Planck-function parts only.
 * Sticky until changed. */
int ansfm_set_gradient_gases(ansfm_ctx *ctx, unsigned int mask);

/* CIRSrad(return_grad=True), ILBL=K_TABLES, IMOD=THERMAL_EMISSION (ForwardModel_0.py:4376-4511
 * with :3853-3872 calc_kg/k_overlapg/dTAUGAS, :3993 dTAUTOT, :4012 LAYINC*SCALE, :4233
 * calc_thermal_emission_spectrumg, :4244-4247 xfac, :4504-4508 g-quadrature + nan_to_num).
 *   dtaucon[n][W][NPAR][L]  dTAUCON of calculate_layer_opacity (:3916-3981) or NULL
 *   NVMR, NPAR = NVMR+2+NDUST; igas_map[S] (HOST int32) = AtmosphereX.locate_gas(ID[i],ISO[i])
 *   SPECOUT[n][W][P], dSPECOUT[n][W][NPAR][LIMAX][P], dTSURF[n][W][P]
 * The reference's O(NPAR*Li^2) recursion is evaluated as one backward sweep.  * dSPECOUT may be NULL for n_models = 1: the gradients then stay on the device (no 8 W NPAR LIMAX P byte copy) for
 * ansfm_map2pro(dSPECIN = NULL); likewise ansfm_map2pro(dSPECOUT = NULL) keeps its result for ansfm_map2xvec(dSPECIN = NULL). */
int ansfm_cirsradg_ck_thermal(ansfm_ctx *ctx, int ISPACE, int n_models, int L,
                              const double *lay_press_pa, const double *lay_temp,
                              const double *amount, const double *taucont, const double *dtaucon,
                              int NVMR, int NPAR, const int32_t *igas_map, int P, int LIMAX,
                              const int32_t *NLAYIN, const int32_t *LAYINC, const double *SCALE,
                              const double *EMTEMP, const double *TSURF, const double *EMISSIVITY,
                              const double *xfac, double *SPECOUT, double *dSPECOUT, double *dTSURF);
int ansfm_cirsradg_ck_thermal_dev(ansfm_ctx *ctx, int ISPACE, int n_models, int L,
                                  const double *lay_press_pa, const double *lay_temp,
                                  const double *amount, const double *taucont,
                                  const double *dtaucon, int NVMR, int NPAR,
                                  const int32_t *igas_map_host, int P, int LIMAX,
                                  const int32_t *NLAYIN, const int32_t *LAYINC,
                                  const double *SCALE, const double *EMTEMP, const double *TSURF,
                                  const double *EMISSIVITY, const double *xfac, double *SPECOUT,
                                  double *dSPECOUT, double *dTSURF);

/* ---- multiple scattering ---------------------------------------------------------------------
 * Multiple_Scattering_Core.scloud11wave_core (Multiple_Scattering_Core.py:651-960): plane-parallel
 * matrix-operator doubling/adding with azimuth Fourier expansion.  Arguments, layouts and meaning
 * are the reference's (host pointers):
 *   phasarr[ncont][nwave][2][nth] (HG: f,g1,g2 in the first 3 slots of [..][0][:] when imie == 0),
 *   radg[nwave][nmu], sol_angs/emiss_angs/aphis[ngeom] (deg), solar[nwave], lowbc (0 thermal, >0 surface),
 *   brdf_matrix[nwave][nmu][nmu][nf+1], mu1/wt1[nmu], bnu[nwave][nlay], taus[nwave][ng][nlay],
 *   tauray[nwave][nlay], omegas_s[nwave][ng][nlay], lfrac[nwave][ncont][nlay]  ->  rad[ngeom][ng][nwave].
 * Mixed emission angles above/below 90 deg -> ANSFM_ERR_INVALID (the reference raises ValueError :776);
 * look-up geometry (all > 90) -> ANSFM_ERR_UNSUPPORTED for now. */
int ansfm_scloud11wave_core(ansfm_ctx *ctx, int ncont, int nwave, int nth, const double *phasarr,
                            const double *radg, int ngeom, const double *sol_angs,
                            const double *emiss_angs, const double *solar, const double *aphis,
                            int lowbc, const double *brdf_matrix, int nmu, const double *mu1,
                            const double *wt1, int nf, const double *bnu, int ng, int nlay,
                            const double *taus, const double *tauray, const double *omegas_s,
                            int nphi, int iray, int imie, const double *lfrac, double *rad);

/* CIRSrad, scattering branch (ILBL = K_TABLES, IMOD & MULTIPLE_SCATTERING; ForwardModel_0.py:4478-4501 ->
 * calculate_multiple_scattering_spectrum :4343 -> scloud11wave :5018-5165) with everything that has a g axis on the
 * device: calc_k + k_overlap give the VERTICAL gas opacities (not LAYINC-scaled: scloud11wave reads LayerX.TAUTOT),
 * TAUTOT = TAUGAS + TAUCIA + TAUDUST + TAURAY (:3989), OMEGA = (TAURAY + TAUSCAT) / TAUTOT and BB = planck(TEMP) (:5099-5119)
 * are formed in HBM and feed the doubling / adding kernels directly; the g-quadrature with DELG (:4504) ends the call.
 * Host pointers, reference layouts: taucia / taudust (summed over populations) / tauray / tauscat [W][L] (NULL = zeros),
 * lfrac[W][ncont][L] = TAUCLSCAT / TAUSCAT, the remaining scloud11wave_core arguments as above (W = the table's wavenumber
 * grid, ng = its g-ordinates), xfac[W] or NULL -> SPECOUT[W][ngeom]; SPEC_G[W][G][ngeom] (what scloud11wave returns, before
 * the quadrature) when not NULL.  ansfm_get_taugas returns the TAUGAS side product afterwards. */
int ansfm_cirsrad_ck_scatter(ansfm_ctx *ctx, int ISPACE, int L, const double *lay_press_pa, const double *lay_temp,
                             const double *amount, const double *taucia, const double *taudust, const double *tauray,
                             const double *tauscat, int ncont, int nth, const double *phasarr, const double *lfrac,
                             const double *radg, int ngeom, const double *sol_angs, const double *emiss_angs,
                             const double *aphis, const double *solar, int lowbc, const double *brdf_matrix, int nmu,
                             const double *mu1, const double *wt1, int nf, int nphi, int iray, int imie, const double *xfac,
                             double *SPECOUT, double *SPEC_G);

/* The same branch for the n_models forward models of a numerical Jacobian: jacobian_nemesis forces the numerical route
 * whenever ISCAT != THERMAL_EMISSION (ForwardModel_0.py:2251-2252), i.e. NX + 1 multiple-scattering forward models that
 * differ from the first in the few layers one state-vector element touches.  Arrays that depend on the state carry a
 * leading model axis: lay_press_pa / lay_temp [n][L], amount [n][S][L], taucia / taudust / tauray / tauscat [n][W][L] (NULL =
 * zeros), lfrac [n][W][ncont][L], radg [n][W][nmu]; phasarr, solar, brdf_matrix, xfac and the geometry are shared
 * -> SPECOUT [n][W][ngeom].
 * 16 streams: the doubled (R, T, J) of a layer (calc_rtj_matrix, Multiple_Scattering_Core.py:566-650) depend on that
 * layer's inputs only and are ~92 % of a chain's work; model 0's pass keeps them per (wavenumber, g, Fourier order,
 * layer) in HBM (slabs of the spectral axis sized to the free memory), every other model re-runs the adding sweep
 * (addp :481-533) over them and recomputes only the layers whose inputs differ from model 0's in any bit -- the same
 * numbers as n_models separate calls, bit for bit.  ansfm_set_layer_dedup(ctx, 0) (or another stream count) runs the
 * models one after the other through ansfm_cirsrad_ck_scatter.  ansfm_last_scatter_cache: (model, layer) pairs taken
 * from the cache / all pairs of models 1..n-1 in the last call; ansfm_last_layer_rows: gas opacity rows computed. */
int ansfm_cirsrad_ck_scatter_batch(ansfm_ctx *ctx, int ISPACE, int n_models, int L, const double *lay_press_pa,
                                   const double *lay_temp, const double *amount, const double *taucia, const double *taudust,
                                   const double *tauray, const double *tauscat, int ncont, int nth, const double *phasarr,
                                   const double *lfrac, const double *radg, int ngeom, const double *sol_angs,
                                   const double *emiss_angs, const double *aphis, const double *solar, int lowbc,
                                   const double *brdf_matrix, int nmu, const double *mu1, const double *wt1, int nf, int nphi,
                                   int iray, int imie, const double *xfac, double *SPECOUT);
int ansfm_last_scatter_cache(const ansfm_ctx *ctx, int64_t *layers_from_cache, int64_t *layers_total);

/* ---- runtime line-by-line (ILBL = LINE_BY_LINE_RUNTIME) -------------------------------------------
 * LineData_0.add_line_set_monochromatic_absorption (LineData_0.py:280-357), batched over L (T,p) points
 * (L = 1 is the reference's signature).  lineshape_id = SpectroscopicLineProfileEnum value: 0 VOIGT
 * (scipy.special.voigt_profile), 4 LORENTZ, 12 DOPPLER; the rest -> ANSFM_ERR_UNSUPPORTED (the reference's
 * enum map raises NotImplementedError for them too).
 *   wn_grid[nw] ascending; t_calc/p_calc/q_ratio[L]; mol_mix_frac[M]; broadening_params[3M][N]
 *   (gamma, n, delta per broadener); nu/sw/e_lower/stim_ref[N];
 *   out[L][nw] is ADDED to (like the reference); store[L][4][N] (strength, alpha_d, gamma_l, shift) or NULL. */
int ansfm_add_line_set_monochromatic_absorption(
    ansfm_ctx *ctx, int nw, const double *wn_grid, int lineshape_id, int L, const double *t_calc, double t_ref,
    const double *p_calc, double p_ref, const double *q_ratio, double isotopic_abundance,
    double isotopic_mass, int M, const double *mol_mix_frac, int N, const double *broadening_params,
    const double *nu, const double *sw, const double *e_lower, const double *stim_ref, double *out,
    double *store, double s_floor, double wn_calc_window, double wn_approx_window);

/* ---- layering ---------------------------------------------------------------------------------------
 * Layer_0.layer_average (Layer_0.py:755-1030), batched over n_models atmospheric states (the states of a
 * numerical Jacobian): LAYINT 0 = MID_PATH, 1 = ABSORBER_WEIGHTED_AVERAGE (Curtis-Godson, :949-1010).
 *   H,P,T[n][NPRO]; VMR[n][NPRO][NVMR]; DUST[n][NPRO][NDUST] or NULL; PARAH2[n][NPRO] or NULL;
 *   BASEH[n][NLAY] (layer_split output); DUST_UNITS[NDUST] (int32) / XMOLWT[n][NPRO] (kg/mol) or NULL
 *   -> HEIGHT,PRESS,TEMP,TOTAM,FRAC,DELH,BASET,LAYSF [n][NLAY]; AMOUNT,PP [n][NLAY][NVMR]; CONT [n][NLAY][NDUST].
 * 2 <= NINT <= 256 (even NINT: scipy.integrate.simpson's last-interval correction, as the reference gets). */
int ansfm_layer_average(ansfm_ctx *ctx, int n_models, double RADIUS, int NPRO, const double *H,
                        const double *P, const double *T, int NVMR, const double *VMR, int NDUST,
                        const double *DUST, const double *PARAH2, int NLAY, const double *BASEH,
                        double LAYANG, int LAYINT, double LAYHT, int NINT, const int32_t *DUST_UNITS,
                        const double *XMOLWT, double *HEIGHT, double *PRESS, double *TEMP, double *TOTAM,
                        double *AMOUNT, double *PP, double *CONT, double *FRAC, double *DELH,
                        double *BASET, double *LAYSF);

/* The same with every array argument resident on the DEVICE (DUST_UNITS stays a host array) and the results left there:
 * out_dev = HEIGHT,PRESS,TEMP,TOTAM,FRAC,DELH,BASET,LAYSF [n][NLAY] one after the other, then AMOUNT [n][NLAY][NVMR],
 * PP [n][NLAY][NVMR], CONT [n][NLAY][NDUST] -- n * NLAY * (8 + 2 NVMR + NDUST) doubles.  Asynchronous on the context's
 * stream: the states of a numerical Jacobian go from profiles to layers to CIRSrad without crossing PCIe.  With more than
 * one state, a layer whose sub-points read only levels at which a state holds state 0's numbers takes state 0's result
 * (bit-identical; also in ansfm_layer_average; ANSFM_LAYER_SHARE=0 integrates every layer). */
int ansfm_layer_average_dev(ansfm_ctx *ctx, int n_models, double RADIUS, int NPRO, const double *H,
                            const double *P, const double *T, int NVMR, const double *VMR, int NDUST,
                            const double *DUST, const double *PARAH2, int NLAY, const double *BASEH,
                            double LAYANG, int LAYINT, double LAYHT, int NINT, const int32_t *DUST_UNITS,
                            const double *XMOLWT, double *out_dev);

/* Layer_0.layer_averageg (Layer_0.py:1032-1398): the same plus DTE, DAM, DCO, DPH [n][NLAY][NPRO], the matrices
 * relating layer temperature / gas amounts / dust amounts / para-H2 fraction to the profile levels (consumed by
 * map2pro).  T, PARAH2, VMR and DUST go through the reference's own `interpg` bracket (:716-751) here, so the layer
 * values differ from ansfm_layer_average's in the last bits exactly as the reference's two functions do.
 * ANSFM_ERR_INVALID for an even NINT (ValueError :1188) and for MID_PATH with DUST_UNITS == -1 (the reference
 * fails there, :1255-1257). */
int ansfm_layer_averageg(ansfm_ctx *ctx, int n_models, double RADIUS, int NPRO, const double *H,
                         const double *P, const double *T, int NVMR, const double *VMR, int NDUST,
                         const double *DUST, const double *PARAH2, int NLAY, const double *BASEH,
                         double LAYANG, int LAYINT, double LAYHT, int NINT, const int32_t *DUST_UNITS,
                         const double *XMOLWT, double *HEIGHT, double *PRESS, double *TEMP, double *TOTAM,
                         double *AMOUNT, double *PP, double *CONT, double *FRAC, double *DELH,
                         double *BASET, double *LAYSF, double *DTE, double *DAM, double *DCO, double *DPH);

/* ---- gradient maps ------------------------------------------------------------------------------------
 * ForwardModel_0.map2pro (ForwardModel_0.py:5319-5383): layer gradients -> profile-level gradients,
 *   dSPECOUT[W][NPAR][NPRO][P] = sum_j dSPECIN[W][NPAR][LIMAX][P] * M[LAYINC[j][p]][NPRO], M = DAM for gas
 *   parameters (index < NVMR), DTE for temperature (== NVMR), DCO for dust (NVMR < index <= NVMR+NDUST);
 *   NPAR = NVMR+2+NDUST.  INCPAR[n_incpar] = the parameters to map (n_incpar = 0: all); unlisted slots stay 0.
 *   The para-H2 slot (NVMR+NDUST+1) receives the PREVIOUS listed parameter's result, as in the reference
 *   (:5373-5377 assign the stale dSPECOUT1); listing it first is the reference's UnboundLocalError ->
 *   ANSFM_ERR_INVALID.   LAYINC[LIMAX][P] int32, DTE/DAM/DCO[NLAY][NPRO].
 * ForwardModel_0.map2xvec (:5387-5424): dSPECOUT[W][P][NX] = sum_{par,pro} dSPECIN[W][NPAR][NPRO][P] * xmap[NX][NPAR][NPRO].
 * Device chaining: dSPECIN == NULL takes the device-resident result of the previous step of this ctx
 * (map2pro: the dSPECOUT of the last single-model ansfm_cirsradg_ck_thermal; map2xvec: the last map2pro) and
 * fails with ANSFM_ERR_INVALID if its dimensions differ.  dSPECOUT == NULL (map2pro only) keeps the result
 * on the device for the following map2xvec. */
int ansfm_map2pro(ansfm_ctx *ctx, int W, int NPAR, int LIMAX, int P, int NPRO, int NLAY, int NVMR, int NDUST,
                  const double *dSPECIN, const int32_t *LAYINC, const double *DTE, const double *DAM,
                  const double *DCO, int n_incpar, const int32_t *INCPAR, double *dSPECOUT);
int ansfm_map2xvec(ansfm_ctx *ctx, int W, int NPAR, int NPRO, int P, int NX, const double *dSPECIN,
                   const double *xmap, double *dSPECOUT);

/* ---- instrument line shape (the step after the path) ---------------------------------------
 * Measurement_0.lblconv (Measurement_0.py:3335; nx = 0) / lblconvg (:3799; nx > 0): convolution of the monochromatic
 * spectrum y[nwave] (and gradients dydx[nwave][nx]) with the ILS given by ISHAPE (0 square, 1 triangular, 2 gaussian,
 * 3 Hamming, 4 Hanning) and FWHM > 0 at the convolution wavenumbers vconv[nconv] -> yout[nconv], gradout[nconv][nx].
 * Reference behaviour kept: only weights > 0 count; the Hamming window of lblconv is the single point
 * vcen - 1.1 FWHM (:3391-3393) while lblconvg uses vcen -+ FWHM (:3866-3868); Hanning assigns no weight, so the
 * result is 0/0 = NaN like the reference's.  vwave must be ascending (ANSFM_ERR_UNSORTED otherwise).
 * lblconv_fil (:3549) / lblconvg_fil (:3992): the ILS is tabulated per convolution point, nfil[nconv] points in
 * column j of vfil / afil [nfilmax][nconv], weights by linear interpolation (np.interp). */
int ansfm_lblconv(ansfm_ctx *ctx, int nwave, const double *vwave, const double *y, int nx, const double *dydx,
                  int nconv, const double *vconv, int ishape, double fwhm, double *yout, double *gradout);
int ansfm_lblconv_fil(ansfm_ctx *ctx, int nwave, const double *vwave, const double *y, int nx,
                      const double *dydx, int nconv, const double *vconv, int nfilmax, const int32_t *nfil,
                      const double *vfil, const double *afil, double *yout, double *gradout);
/* lblconv_ngeom (:3444) / lblconvg_ngeom (:3685) / lblconv_fil_ngeom (:3614) / lblconvg_fil_ngeom (:3912): NGEOM spectra
 * on one grid, y[nwave][ngeom], dydx[nwave][ngeom][nx] (nx = 0: none) -> yout[nconv][ngeom], gradout[nconv][ngeom][nx].
 * Same arithmetic per column; the Hamming window of both ISHAPE variants is the single point vcen - FWHM (:3501-3503,
 * :3753-3755).  (The reference's lblconv_ngeom is an un-jitted Python loop.) */
int ansfm_lblconv_ngeom(ansfm_ctx *ctx, int nwave, const double *vwave, int ngeom, const double *y, int nx,
                        const double *dydx, int nconv, const double *vconv, int ishape, double fwhm, double *yout,
                        double *gradout);
int ansfm_lblconv_fil_ngeom(ansfm_ctx *ctx, int nwave, const double *vwave, int ngeom, const double *y, int nx,
                            const double *dydx, int nconv, const double *vconv, int nfilmax, const int32_t *nfil,
                            const double *vfil, const double *afil, double *yout, double *gradout);

/* integrate_filter (:4079) / integrate_filterg (:4188) and their *_ngeom variants (:4131, :4251): np.trapz of
 * (filter x spectrum) over the calculation points inside each filter, no normalisation.  y[nwave][ngeom],
 * dydx[nwave][ngeom][nx] (ngeom = 1 for the single-geometry functions) -> yout[nconv][ngeom], gradout[nconv][ngeom][nx]. */
int ansfm_integrate_filter(ansfm_ctx *ctx, int nwave, const double *vwave, int ngeom, const double *y, int nx,
                           const double *dydx, int nconv, const double *vconv, int nfilmax, const int32_t *nfil,
                           const double *vfil, const double *afil, double *yout, double *gradout);

/* Measurement_0.conv (:2288) / convg (:2467), k-table runs, FWHM < 0 branch (:2425-2461, :2655-2691): the filter
 * average of lblconv_fil, but over the window from the last calculation point BELOW vfil[0][j] to the first ABOVE
 * vfil[nfil[j]-1][j] (both must exist: the reference raises IndexError otherwise -> ANSFM_ERR_INVALID), np.interp
 * clamping to the edge values outside the filter. */
int ansfm_conv_fil(ansfm_ctx *ctx, int nwave, const double *vwave, const double *y, int nx, const double *dydx,
                   int nconv, const double *vconv, int nfilmax, const int32_t *nfil, const double *vfil,
                   const double *afil, double *yout, double *gradout);

/* ---- continuum ---------------------------------------------------------------------------
 * ForwardModel_0.calc_tau_cia (ForwardModel_0.py:4516-4760), wavenumber space (the caller converts and re-orders a
 * wavelength grid as the reference does, :4565-4568 / :4741-4743).  WAVEN[W] ascending; the CIA table
 * K_CIA[NPAIR][NPE][NT][NWC] on cia_waven / cia_temp / cia_frac[nfrac] (NPE = size of the para axis of K_CIA, NPARA =
 * CIA.NPARA, 0 for tables without ortho/para dependence); igas1 / igas2[NPAIR] = index of each partner among the NVMR
 * atmospheric gases, -1 when the pair is not to be used (gas missing or ambiguous :4676-4684, or the pair's INORMALT
 * differs from CIA.INORMAL :4696-4701); per layer temperature, para-H2 fraction, mixing ratios q[L][NVMR] = PP/PRESS
 * and xfac[L] = (TOTAM*1e-4)^2 / (DELH*1e2); ico2 / in2 / ih2 = index of CO2 / N2 / H2 (or -1) with the reference's
 * co2cia / n2n2cia / n2h2cia(WAVEN) vectors (wavenumber-only parametrisations; NULL when the index is -1).
 * -> TAUCIA[W][L], dTAUCIA[W][L][NVMR+2] (NULL to skip).  Reference quirks kept (temp1 overwritten by the upper
 * para-fraction clamp :4623; temperature gradient in slot NVMR-2 :4695). */
int ansfm_calc_tau_cia(ansfm_ctx *ctx, int W, const double *WAVEN, int NWC, const double *cia_waven, int NPAIR,
                       int NPE, int NT, const double *K_CIA, const double *cia_temp, int nfrac,
                       const double *cia_frac, int NPARA, const int32_t *igas1, const int32_t *igas2, int L,
                       int NVMR, const double *lay_temp, const double *lay_frac, const double *q,
                       const double *xfac, int ico2, const double *k_co2, int in2, const double *k_n2n2, int ih2,
                       const double *k_n2h2, double *TAUCIA, double *dTAUCIA);

/* ForwardModel_0.calc_tau_rayleigh (ForwardModel_0.py:4869): mode = IRAY -- 1 calc_tau_rayleighj (:5525, gas giants),
 * 2 calc_tau_rayleighv2 (:5647, CO2), 4 calc_tau_rayleighls (:5712, Jovian air; f4[L][4] = mixing ratios of H2, He, CH4,
 * NH3 per layer, 0 where absent) -- or 12 for calc_tau_rayleighv (:5598, not selected by any IRAY).  WAVEC[W] in the
 * units of ISPACE, TOTAM[L] in m-2 -> TAURAY[W][L], dTAURAY[W][L] (= dTAURAY/dTOTAM). */
int ansfm_calc_tau_rayleigh(ansfm_ctx *ctx, int mode, int ISPACE, int W, const double *WAVEC, int L,
                            const double *TOTAM, const double *f4, double *TAURAY, double *dTAURAY);

/* calc_tau_rayleigh for the n states of a Jacobian batch, on the uploaded table's wavenumber grid, left in HBM:
 * TOTAM[n][L] and (mode 4) f4[n][L][4] are host arrays (a few kB per state), TAURAY_dev[n][W][L] is a DEVICE pointer in
 * the layout ansfm_cirsrad_ck_thermal_dev takes as taucont (8 MB per state at C2 that never cross PCIe). */
int ansfm_calc_tau_rayleigh_batch_dev(ansfm_ctx *ctx, int mode, int ISPACE, int n_models, int L, const double *TOTAM,
                                      const double *f4, double *TAURAY_dev);
/* ... and with TOTAM / f4 resident on the device too (what ansfm_layer_average_dev left there); asynchronous. */
int ansfm_calc_tau_rayleigh_batch_dev_in(ansfm_ctx *ctx, int mode, int ISPACE, int n_models, int L,
                                         const double *TOTAM_dev, const double *f4_dev, double *TAURAY_dev);

/* ForwardModel_0.calc_tau_dust (ForwardModel_0.py:4790): KEXT / KSCA[NWS][NDUST] tabulated on SWAVE[NWS] (Scatter.WAVE,
 * strictly ascending) interpolated to WAVEC[W] like scipy interp1d(kind='cubic') (not-a-knot spline; linear when
 * NWS == 2; NWS == 3 is refused as scipy refuses it), out-of-range values replaced by the linear interpolant
 * (:4849-4859); CONT[L][NDUST] in particles m-2 (after any DUST_RENORMALISATION, which the caller applies).
 * -> TAUDUST, TAUCLSCAT, dTAUDUSTdq, dTAUCLSCATdq [W][L][NDUST].  A WAVEC outside SWAVE is ANSFM_ERR_INVALID (the
 * reference's interp1d raises ValueError). */
int ansfm_calc_tau_dust(ansfm_ctx *ctx, int W, const double *WAVEC, int NWS, const double *SWAVE, int NDUST,
                        const double *KEXT, const double *KSCA, int L, const double *CONT, double *TAUDUST,
                        double *TAUCLSCAT, double *dTAUDUSTdq, double *dTAUCLSCATdq);

/* Layer de-duplication inside a batch (n_models > 1) of the cirsrad_ck_thermal entry points.  The states of a
 * numerical Jacobian (ForwardModel_0.jacobian_nemesis :2234-2242) differ from the unperturbed one in two or three
 * layers; every layer (m, l) whose pressure, temperature and S amounts equal those of layer l of model 0 to the last
 * bit shares its gas opacity with it instead of being merged again.  Results are bit-identical with and without;
 * on by default; costs one stream synchronisation per batched call.  ansfm_last_layer_rows reports how many layer
 * opacities the last call computed out of n_models * L. */
int ansfm_set_layer_dedup(ansfm_ctx *ctx, int enable);
int ansfm_last_layer_rows(const ansfm_ctx *ctx, int *rows_computed, int *rows_total);

/* *shared = 1 when the last thermal-emission batch started the paths of its states from the records state 0 left behind
 * (a de-duplicated batch of at least 4 states: every state shares the top of each path with state 0 up to the first layer
 * whose opacity row, continuum, SCALE or EMTEMP differs; bit-identical; ANSFM_RT_PREFIX=0 switches it off). */
int ansfm_last_rt_shared(const ansfm_ctx *ctx, int *shared);

/* Row-head list of the forward random-overlap merge (k_overlap / rank, ForwardModel_0.py:6029-6173): bits = 64 (default)
 * orders the heads on double keys (k_ck_overlap: values closer than 2^-41 count as ties); bits = 32 on float32 keys
 * (k_ck_overlap32: heads whose float32 values coincide are settled by their exact double sums, every merge carries a
 * check that the sums were consumed in non-decreasing order and is rerun exactly when it was not).  Both produce the
 * reference's merged order.  Measured on MI355X: the 32-bit kernel issues fewer instructions but
 * exposes more LDS latency and is the slower of the two at C2, hence opt-in.  Input that is unsorted or negative runs
 * on the generic 64-bit path either way. */
int ansfm_set_merge_keys(ansfm_ctx *ctx, int bits);
/* Statistics of the 32-bit-key kernel: (wave, gas) merges whose fast pass did not consume the sums in non-decreasing
 * order and were rerun with every step popping the exact minimum; counted since the last table upload. */
int ansfm_merge_redo_count(ansfm_ctx *ctx, int64_t *count);

/* Vertical gas opacity of the last cirsrad call's first model, TAUGAS[W][G][L]
 * (what CIRSrad leaves in LayerX.TAUGAS, ForwardModel_0.py:3925) -- host pointer out. */
int ansfm_get_taugas(ansfm_ctx *ctx, int model, double *TAUGAS);

/* ---- measurement helpers (bench.py / profiling) -------------------------------------------
 * Time of the dominant kernel (ck_overlap) measured with hipEvents on the ctx stream around
 * the launches of the last cirsrad call: total milliseconds and number of launches. */
int ansfm_last_kernel_ms(const ansfm_ctx *ctx, double *overlap_ms, int *overlap_launches,
                         double *rt_ms, int *rt_launches);

#ifdef __cplusplus
}
#endif
#endif /* ANSFM_H */
