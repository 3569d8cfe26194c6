// ansfm_ms_kernels.hip.h -- doubling/adding multiple-scattering core on gfx950.
//
// Restates Multiple_Scattering_Core.scloud11wave_core (Multiple_Scattering_Core.py:651-960) and its
// callees phasint2 :141, hansen :200, add :275, double1 :321, idown :366, addp :481, angle_quadrature :535,
// calc_rtj_matrix :566, for both viewing geometries (look-down: surface first, layers bottom to top; look-up: layers
// top to bottom, the surface kept apart and combined through idown when lowbc > 0).
//
// Decomposition (every (wavenumber, g, Fourier order) chain is independent):
//   k_ms_phase   one block per (wave, scatterer): azimuth-integrated phase matrices P++ / P+- for
//                ic = 0..nf.  They do not depend on g; the reference recomputes them inside its g
//                loop (:780-815) -- hoisted here.
//   k_ms_hansen_seq  Hansen renormalisation factors, sequential in the reference's (g, wave) order
//                (its fc array is carried from one iteration to the next, see the kernel).
//   k_ms_chain   one wavefront per (wave, g, ic): per layer doubling (double1/add) and adding
//                (addp) of the (R,T,J) operators, nmu x nmu float64 matrices in LDS; then the
//                2x2 (mu0,mu) samples of R u0+ + T u- + J for every path -> drad[wave][g][ic][path].
//   k_ms_chain16 the same chain for nmu == 16 with the stream x stream products on the matrix cores
//                (v_mfma_f64_16x16x4_f64), operands chained in the accumulator layout.
//   k_ms_fourier one thread per (wave, g, path): the Fourier sum with the reference's early-out
//                (:949-958) -> rad[path][g][wave].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ansfm {

constexpr int kMsMaxMu = 32;
constexpr int kMsMaxPath = 16;

struct MsParams {
    // reference-layout inputs (device pointers)
    const double *phasarr;   // [ncont][nwave][2][nth]
    const double *radg;      // [nwave][nmu]
    const double *solar;     // [nwave]
    const double *brdf;      // [nwave][nmu][nmu][nf+1]
    const double *bnu;       // [nwave][nlay]
    const double *taus;      // [nwave][ng][nlay]
    const double *tauray;    // [nwave][nlay]
    const double *omegas;    // [nwave][ng][nlay]
    const double *lfrac;     // [nwave][ncont][nlay]
    // workspaces / outputs
    double *ppl, *pmi;       // [nwave][nf+1][ncomp][nmu*nmu]   raw azimuth integrals
    double *fc;              // [ng][nwave][ncomp][nmu*nmu]     Hansen factors in the reference's loop order
    double *drad;            // [nwave][ng][nf+1][ngeom]
    double *rad;             // [ngeom][ng][nwave]
    int ncont, ncomp, nwave, nth, ngeom, lowbc, nmu, nf, ng, nlay, nphi, iray, imie;
    double mu[kMsMaxMu], wtmu[kMsMaxMu];          // already reversed (:725-726)
    double sol_ang[kMsMaxPath], emiss_ang[kMsMaxPath], aphi[kMsMaxPath];
    double xfac;
    int hansen_comp0, phase_comp0;
    int lookup;              // all emission angles > 90: layers top to bottom, surface brought in with idown (:366-420)
    int ig0, ng_launch;      // k_ms_hansen_seq / k_ms_chain16: the g-ordinates [ig0, ig0 + ng_launch) of this launch
    int phase_tab;           // k_ms_phase: the cos(ic phi) table fits in LDS (else the cosines are evaluated in place)
    int phase_lds;           // k_ms_chain16: the phase matrices of the components in use fit in LDS beside the operators
    // Batched Jacobian of the scattering branch (ansfm_cirsrad_ck_scatter_batch; k_ms_chain16<.., CACHE>).  A launch covers
    // the wavenumbers [w0, w0 + wcount) (0 / nwave outside the batch path); taus / omegas / bnu of such a launch are laid out
    // [model of the launch][wcount][..] relative to w0, every other per-wavenumber array keeps the whole axis.
    int w0, wcount;
    int m0, n_launch;        // CACHE = 2: models [m0, m0 + n_launch) of the batch, one block per (model, wavenumber, g)
    double *cache;           // [wcount][ng][nf+1][nlay][528]: doubled (r, t, j) of every scattering layer of model 0
    int *cache_orders;       // [wcount][ng]: Fourier orders model 0 worked through (those are in the cache)
    const unsigned char *same;   // [n_models][nlay]: the layer's inputs are bit-identical to model 0's
    double *pcache;          // [wcount][ng][nf+1][npre][528]: model 0's STACK (rc, tc, jc) after every kMsPrefixStep-th layer of the sweep
    const int *lstart;       // [n_models]: sweep index (multiple of kMsPrefixStep) a model's adding sweep may start from
    const int *model_ids;    // CACHE = 2: the models in launch order (sorted by lstart: the blocks of one launch then walk the same
                             // layers at about the same time and find model 0's cache lines in L2); model of block ml = ids[m0 + ml]
    int npre;
    size_t st_wl, st_wcl, st_wm, st_rad;   // strides between models: tauray [W][L], lfrac [W][ncont][L], radg [W][nmu], rad
    size_t st_drad;                        // ... and drad (k_ms_chain_lane<N, CACHE> leaves the orders to k_ms_fourier)
    // 7 .. 15 streams on the 16-stream kernels: nmu = 16, nmu_real = the quadrature's size (0 = nmu).  mu / wtmu beyond it are
    // 1 / 0, the phase matrices, the surface operator and the boundary radiance zero there: every operator is block diagonal with
    // the quadrature's block in front and a block that couples to nothing behind it
    int nmu_real;
};

// radg [rows][nr] -> [rows][16]: the chain kernels read it back to front (radg[:, ::-1], :765), so the quadrature's values go to
// the END of a padded row.  brdf [W][nr][nr][nf1] -> [W][16][16][nf1], zero outside the block.
__global__ void k_ms_pad_radg(size_t rows, int nr, const double *__restrict__ src, double *__restrict__ dst)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * 16) return;
    const int k = (int)(idx % 16);
    const size_t r = idx / 16;
    dst[idx] = (k >= 16 - nr) ? src[r * nr + (k - (16 - nr))] : 0.0;
}
__global__ void k_ms_pad_brdf(size_t W, int nr, int nf1, const double *__restrict__ src, double *__restrict__ dst)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= W * 256 * nf1) return;
    const int ic = (int)(idx % nf1), j = (int)((idx / nf1) % 16), i = (int)((idx / ((size_t)nf1 * 16)) % 16);
    const size_t w = idx / ((size_t)nf1 * 256);
    dst[idx] = (i < nr && j < nr) ? src[((w * nr + i) * nr + j) * nf1 + ic] : 0.0;
}
constexpr int kMsCacheEntry = 528;   // doubles per cached layer: r (256) and t (256) in the MFMA accumulator layout, j (16)
constexpr int kMsPrefixStep = 4;     // the stack below is kept after sweep layers 3, 7, 11, ...
constexpr int kMsHansenDepth = 8;    // steps the Hansen walk fetches its matrices ahead (k_ms_hansen_seq)

__device__ __forceinline__ double ms_interp(double x, const double *xp, const double *fp, int n)
{   // np.interp
    if (x <= xp[0]) return fp[0];
    if (x >= xp[n - 1]) return fp[n - 1];
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid; else hi = mid; }
    double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
    return slope * (x - xp[lo]) + fp[lo];
}

// ---- phase matrices -----------------------------------------------------------------------------
// raw azimuth integrals (phasint2 :141-197), one block per (wave, scatterer), all ic.
// A thread owns matrix elements (i, j) and walks the azimuth grid ONCE for all Fourier orders: the phase function at
// (i, j, phi) does not depend on the order (the first version evaluated it per (order, element): nf + 1 times), and
// cos(ic phi) depends on neither i nor j -- a table in LDS, filled by the same expression.  Same sums in the same order.
constexpr int kMsPhaseOrders = 9;     // Fourier orders accumulated per pass over the azimuth grid (nf = 8: one pass)
__global__ __launch_bounds__(256) void k_ms_phase(MsParams p)
{
    extern __shared__ double ctab[];            // [nf + 2][nphi + 1]: cos(ic phi_k); the last row is cos(phi_k)
    const int widx = blockIdx.x, comp = blockIdx.y + p.phase_comp0;  // comp == ncont -> Rayleigh
    const int n = p.nmu, nn = n * n, tid = threadIdx.x;
    const double pi = 3.141592653589793;
    const double dphi = 2.0 * pi / p.nphi;
    const int jc = comp < p.ncont ? comp : (p.ncont > 0 ? p.ncont - 1 : 0);
    const double *pfunc = p.phasarr + (((size_t)jc * p.nwave + widx) * 2 + 0) * p.nth;
    const double *xmu = p.phasarr + (((size_t)jc * p.nwave + widx) * 2 + 1) * p.nth;
    const int iscat = (comp == p.ncont) ? 0 : (p.imie == 0 ? 2 : 4);
    const int nk = p.nphi + 1;
    const bool tab = p.phase_tab != 0;
    if (tab) {
        for (int t = tid; t < (p.nf + 2) * nk; t += blockDim.x) {
            const int ic = t / nk, k = t % nk;
            const double phi = k * dphi;
            ctab[t] = (ic <= p.nf) ? cos(ic * phi) : cos(phi);
        }
        __syncthreads();
    }
    const double *cphi_row = ctab + (size_t)(p.nf + 1) * nk;
    const int nr = p.nmu_real ? p.nmu_real : n;
    for (int e = tid; e < nn; e += blockDim.x) {
        const int i = e / n, j = e % n;
        if (i >= nr || j >= nr) {               // beyond the quadrature (a smaller one padded to 16 streams): nothing scatters there
            for (int ic = 0; ic <= p.nf; ++ic) {
                p.ppl[(((size_t)widx * (p.nf + 1) + ic) * p.ncomp + comp) * nn + e] = 0.0;
                p.pmi[(((size_t)widx * (p.nf + 1) + ic) * p.ncomp + comp) * nn + e] = 0.0;
            }
            continue;
        }
        const double sthi = sqrt(1.0 - p.mu[i] * p.mu[i]), sthj = sqrt(1.0 - p.mu[j] * p.mu[j]);
        const double ss = sthi * sthj, mmu = p.mu[i] * p.mu[j];
        for (int ic0 = 0; ic0 <= p.nf; ic0 += kMsPhaseOrders) {
            double spl[kMsPhaseOrders], smi[kMsPhaseOrders];
#pragma unroll
            for (int o = 0; o < kMsPhaseOrders; ++o) { spl[o] = 0.0; smi[o] = 0.0; }
            for (int k = 0; k <= p.nphi; ++k) {
                const double phi = k * dphi;
                const double cphi = tab ? cphi_row[k] : cos(phi);
                const double cpl = ss * cphi + mmu, cmi = ss * cphi - mmu;
                double pl, pm;
                if (iscat == 0) {
                    pl = 0.75 * (1.0 + cpl * cpl) / (4 * pi);
                    pm = 0.75 * (1.0 + cmi * cmi) / (4 * pi);
                } else if (iscat == 2) {
                    const double f1 = pfunc[0], f2 = 1.0 - f1;
                    const double hg11 = 1.0 - pfunc[1] * pfunc[1], hg12 = 2.0 - hg11;
                    const double hg21 = 1.0 - pfunc[2] * pfunc[2], hg22 = 2.0 - hg21;
                    double s1 = sqrt(hg12 - 2.0 * pfunc[1] * cpl), s2 = sqrt(hg22 - 2.0 * pfunc[2] * cpl);
                    pl = f1 * hg11 / (s1 * s1 * s1) + f2 * hg21 / (s2 * s2 * s2);
                    s1 = sqrt(hg12 - 2.0 * pfunc[1] * cmi); s2 = sqrt(hg22 - 2.0 * pfunc[2] * cmi);
                    pm = f1 * hg11 / (s1 * s1 * s1) + f2 * hg21 / (s2 * s2 * s2);
                    pl /= 4 * pi; pm /= 4 * pi;
                } else {
                    pl = ms_interp(cpl, xmu, pfunc, p.nth);
                    pm = ms_interp(cmi, xmu, pfunc, p.nth);
                }
                const double w = (k == 0 || k == p.nphi) ? 0.5 * dphi : dphi;
                const double w0 = w / (2.0 * pi), w1 = w / pi;
#pragma unroll
                for (int o = 0; o < kMsPhaseOrders; ++o) {
                    const int ic = ic0 + o;
                    if (ic <= p.nf) {
                        const double wphi = (ic == 0) ? w0 : w1;
                        const double cic = tab ? ctab[(size_t)ic * nk + k] : cos(ic * phi);
                        spl[o] += wphi * (pl * cic);
                        smi[o] += wphi * (pm * cic);
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < kMsPhaseOrders; ++o) {
                const int ic = ic0 + o;
                if (ic <= p.nf) {
                    p.ppl[(((size_t)widx * (p.nf + 1) + ic) * p.ncomp + comp) * nn + e] = spl[o];
                    p.pmi[(((size_t)widx * (p.nf + 1) + ic) * p.ncomp + comp) * nn + e] = smi[o];
                }
            }
        }
    }
}

// Cross-lane sums without the LDS crossbar (__shfl_xor = ds_bpermute: an LDS round trip per stage, six dependent stages for a
// norm, in front of every product that waits for the result).  Within a 16-lane row: DPP row rotations (VALU); across the
// four rows: gfx950's v_permlane16_swap / v_permlane32_swap (vdst's odd rows / upper half <-> src's even rows / lower half;
// with both operands the same register the two results are the partner values).  tools/calib/lane_reduce_check.hip.
template <int CTRL> __device__ __forceinline__ double ms_dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double ms_xor16_sum(double s)      // s[l] + s[l ^ 16]
{
    const long long b = __double_as_longlong(s);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __longlong_as_double(((long long)rh[0] << 32) | (unsigned long long)rl[0]) +
           __longlong_as_double(((long long)rh[1] << 32) | (unsigned long long)rl[1]);
}
__device__ __forceinline__ double ms_xor32_sum(double s)      // s[l] + s[l ^ 32]
{
    const long long b = __double_as_longlong(s);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __longlong_as_double(((long long)rh[0] << 32) | (unsigned long long)rl[0]) +
           __longlong_as_double(((long long)rh[1] << 32) | (unsigned long long)rl[1]);
}
// maximum over the 64 lanes, the same way (every lane gets it)
__device__ __forceinline__ double ms_wave_max(double x)
{
    x = fmax(x, ms_dpp_f64<0x128>(x));
    x = fmax(x, ms_dpp_f64<0x124>(x));
    x = fmax(x, ms_dpp_f64<0x122>(x));
    x = fmax(x, ms_dpp_f64<0x121>(x));
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = fmax(__longlong_as_double(((long long)rh[0] << 32) | (unsigned long long)rl[0]),
             __longlong_as_double(((long long)rh[1] << 32) | (unsigned long long)rl[1]));
    const long long b2 = __double_as_longlong(x);
    const unsigned lo2 = (unsigned)b2, hi2 = (unsigned)(b2 >> 32);
    const auto sl = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);
    const auto sh = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
    return fmax(__longlong_as_double(((long long)sh[0] << 32) | (unsigned long long)sl[0]),
                __longlong_as_double(((long long)sh[1] << 32) | (unsigned long long)sl[1]));
}

// Hansen renormalisation (hansen :200-233).  The reference keeps ONE fc array per scatterer for the whole
// call and hansen() updates it in place, so the factor found for (g, wave) is the starting point of the
// next (g, wave) in loop order (:780-815); the normalisation has no unique solution, so the result depends
// on that history (a 1e-4 effect on the radiance).  Reproduced: one wavefront per scatterer walks the
// (g outer, wave inner) sequence and stores fc[g][wave][comp][nmu*nmu].
// One wavefront per block: LDS instructions of a wave execute in order, so a compiler fence orders the lanes' LDS accesses.
// __syncthreads() also waits for every outstanding GLOBAL access (vmcnt(0)) -- here that is the prefetch of the matrices two
// steps ahead and the store of the factors, i.e. a full memory round trip in every step of a walk that is nothing but latency.
#define MS_WAVE_SYNC() __atomic_signal_fence(__ATOMIC_SEQ_CST)
template <int NMU>   // NMU = 16: compile-time size (unrolled sums, all LDS reads in flight); 0: any nmu <= kMsMaxMu
__global__ __launch_bounds__(64) void k_ms_hansen_seq(MsParams p)
{
    __shared__ double ppl_s[kMsMaxMu * kMsMaxMu], pmi_s[kMsMaxMu * kMsMaxMu], fc[kMsMaxMu * kMsMaxMu];
    __shared__ double rsum[kMsMaxMu], tsum[kMsMaxMu], xs[kMsMaxMu];
    // two waves on the whole chip walk while thousands of chain waves compute: the walk gates the chains of the next
    // g-ordinate, so its waves issue first wherever they share a SIMD
    __builtin_amdgcn_s_setprio(3);
    const int comp = blockIdx.x + p.hansen_comp0;
    const int n = NMU ? NMU : p.nmu, nn = n * n, tid = threadIdx.x;
    const double x1 = 2.0 * 3.141592653589793;
    const int nr = p.nmu_real ? p.nmu_real : n;              // the quadrature's size (a smaller one padded to 16 streams)
    constexpr int NE = NMU ? (NMU * NMU + 63) / 64 : (kMsMaxMu * kMsMaxMu + 63) / 64;   // matrix elements per lane
    // The walk is a chain of short steps (a converged start needs one pass over a 16 x 16 matrix) and every step needs two
    // matrices from HBM / L2: a microsecond away.  They are fetched kMsHansenDepth steps ahead into a ring of registers (the
    // loop is unrolled over the ring so that every slot is a fixed set of registers and the wait before a slot is used counts
    // the younger loads and stores still in flight instead of draining them); fetched ONE step ahead, as until round 3, the
    // step took the memory latency: 2.0-2.5 us, 20-25 ms per g-ordinate at 1e4 wavenumbers, which was the wall time of a C4
    // forward model (the chain kernels of g-ordinate ig wait for the factors of ig).
    constexpr int D = NMU ? kMsHansenDepth : 2;
    constexpr bool FULL = NMU != 0 && (NMU * NMU) % 64 == 0;     // every lane holds NE elements: no bounds tests on the accesses
    // a compile-time size that does not fill the lanes (5 x 5: the reference's default quadrature): the idle lanes repeat the
    // last element's access -- the same value to the same address -- so that the number of loads and stores stays fixed
    constexpr bool CLAMP = NMU != 0 && !FULL;
    double ring[D][2 * NE];
    auto fetch = [&](int d, int widx) {
        const double *gppl = p.ppl + (((size_t)widx * (p.nf + 1) + 0) * p.ncomp + comp) * nn;
        const double *gpmi = p.pmi + (((size_t)widx * (p.nf + 1) + 0) * p.ncomp + comp) * nn;
#pragma unroll
        for (int r = 0; r < NE; ++r) {
            const int e = CLAMP ? min(tid + 64 * r, nn - 1) : tid + 64 * r;
            if (FULL || CLAMP || e < nn) { ring[d][r] = gppl[e]; ring[d][NE + r] = gpmi[e]; }
        }
    };
    auto stage = [&](int d) {
#pragma unroll
        for (int r = 0; r < NE; ++r) {
            const int e = CLAMP ? min(tid + 64 * r, nn - 1) : tid + 64 * r;
            if (FULL || CLAMP || e < nn) { ppl_s[e] = ring[d][r]; pmi_s[e] = ring[d][NE + r]; }
        }
    };
    // the walk of this launch: g-ordinates [ig0, ig0 + ng_launch), continuing from the factors the previous launch left
    // for its last (g, wave) -- a launch per g-ordinate lets the chains of g run beside the walk of g + 1
    if (p.ig0 == 0) { for (int e = tid; e < nn; e += 64) fc[e] = 1.0; }
    else {
        const double *pfc = p.fc + (((size_t)(p.ig0 - 1) * p.nwave + (p.nwave - 1)) * p.ncomp + comp) * nn;
        for (int e = tid; e < nn; e += 64) fc[e] = pfc[e];
    }
    const long total = (long)p.ng_launch * p.nwave;
    int wf = 0;                                                  // wavenumber of the next fetch (wraps around: see below)
#pragma unroll
    for (int d = 0; d < D; ++d) { fetch(d, wf); if (++wf == p.nwave) wf = 0; }
    int ig = p.ig0, widx = 0;
    for (long iter0 = 0; iter0 < total; iter0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const long iter = iter0 + d;
        if (iter >= total) break;
        const double *ppl = ppl_s, *pmi = pmi_s;
        stage(d);                                               // the previous step's readers are behind this wave's fence
        // always issued (beyond the end of the walk it re-reads a matrix that is there): a fetch under a condition leaves the
        // number of loads in flight unknown to the compiler, which then waits for all of them
        fetch(d, wf);
        if (++wf == p.nwave) wf = 0;
        MS_WAVE_SYNC();
        double rs = 0.0;
        if (tid < nr) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < n; ++i) s += pmi[i * n + tid] * p.wtmu[i];
            rs = s * x1;
            rsum[tid] = rs;
        }
        for (int niter = 0; niter < 10000; ++niter) {
            double dev = 0.0;
            if (tid < nr) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < n; ++i) s += ppl[i * n + tid] * p.wtmu[i] * fc[i * n + tid];
                const double ts = s * x1;
                tsum[tid] = ts;
                dev = fabs(rs + ts - 1.0);
            }
            dev = ms_wave_max(dev);                  // VALU lane exchanges (six dependent ds_bpermute stages before)
            MS_WAVE_SYNC();
            if (dev < 1e-14) break;
            if (tid < nr) xs[tid] = (1.0 - rsum[tid]) / tsum[tid];   // one division per column instead of two per element
            else if (tid < n) xs[tid] = 1.0;                         // columns beyond the quadrature keep their factor of 1
            MS_WAVE_SYNC();
            for (int e = tid; e < nn; e += 64) {
                const int i = e / n, j = e % n;
                if (i <= j) {
                    const double xj = xs[j], xi = xs[i];
                    const double v = 0.5 * (fc[i * n + j] * xj + fc[j * n + i] * xi);
                    fc[i * n + j] = v;
                    fc[j * n + i] = v;
                }
            }
            MS_WAVE_SYNC();
        }
        double *ofc = p.fc + (((size_t)ig * p.nwave + widx) * p.ncomp + comp) * nn;
#pragma unroll
        for (int r = 0; r < NE; ++r) {           // a fixed number of stores per step (the waits before the ring's slots count them)
            const int e = CLAMP ? min(tid + 64 * r, nn - 1) : tid + 64 * r;
            if (FULL || CLAMP || e < nn) ofc[e] = fc[e];
        }
        if (++widx == p.nwave) { widx = 0; ++ig; }
        MS_WAVE_SYNC();
    }
    }
}

// ---- small dense helpers on LDS matrices (one wavefront = one block), any nmu <= kMsMaxMu = 32 -------------------
typedef double ms_v4f64 __attribute__((ext_vector_type(4)));

// (the blocks of k_ms_chain are ONE wavefront: a compiler fence orders its lanes' LDS accesses -- MS_WAVE_SYNC, as in the
//  Hansen walk; __syncthreads() would also wait for every global access in flight at each of the ~30 points of a layer)
__device__ __forceinline__ void ms_mm(int n, int ld, const double *A, const double *B, double *C, int lane)
{   // C = A B   (C distinct from A and B)
    for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += A[i * ld + k] * B[k * ld + j];
        C[i * ld + j] = s;
    }
    MS_WAVE_SYNC();
}
__device__ __forceinline__ void ms_mv(int n, int ld, const double *A, const double *x, double *y, int lane)
{   // y = A x   (y distinct from x)
    if (lane < n) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += A[lane * ld + k] * x[k];
        y[lane] = s;
    }
    MS_WAVE_SYNC();
}
__device__ __forceinline__ double ms_frob(int n, int ld, const double *r, int lane)
{
    double s = 0.0;
    for (int e = lane; e < n * n; e += 64) { const double v = r[(e / n) * ld + (e % n)]; s += v * v; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return sqrt(s);
}
// Ainv = inverse(A) by Gauss-Jordan with partial pivoting (first largest |.|, like LAPACK's idamax); A is
// destroyed.  col = scratch[n].  The pivot search is a wavefront arg-max over the n candidate rows.
__device__ __forceinline__ void ms_inv(int n, int ld, double *A, double *Ainv, double *col, int lane)
{
    for (int e = lane; e < n * n; e += 64) { const int i = e / n, j = e % n; Ainv[i * ld + j] = (i == j) ? 1.0 : 0.0; }
    MS_WAVE_SYNC();
    for (int c = 0; c < n; ++c) {
        double best = (lane >= c && lane < n) ? fabs(A[lane * ld + c]) : -1.0;
        int piv = lane;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {     // n <= 32: lanes 0..31 hold every candidate
            const double ob = __shfl_xor(best, off, 64);
            const int op = __shfl_xor(piv, off, 64);
            if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
        }
        piv = __builtin_amdgcn_readfirstlane(piv);
        if (lane < n) {
            double ac = A[c * ld + lane], wc = Ainv[c * ld + lane];
            if (piv != c) {
                const double ap = A[piv * ld + lane], wp = Ainv[piv * ld + lane];
                A[piv * ld + lane] = ac; Ainv[piv * ld + lane] = wc;
                ac = ap; wc = wp;
            }
            // pivot element after the swap, read by every lane from the (not yet scaled) row
            const double d = 1.0 / __shfl(ac, c, 64);
            A[c * ld + lane] = ac * d;
            Ainv[c * ld + lane] = wc * d;
        }
        MS_WAVE_SYNC();
        if (lane < n) col[lane] = A[lane * ld + c];
        MS_WAVE_SYNC();
        for (int e = lane; e < n * n; e += 64) {
            const int r = e / n, j = e % n;
            if (r != c) {
                const double f = col[r];
                A[r * ld + j] -= f * A[c * ld + j];
                Ainv[r * ld + j] -= f * Ainv[c * ld + j];
            }
        }
        MS_WAVE_SYNC();
    }
}

// ---- (R,T,J) chain of one (wave, g, ic) -------------------------------------------------------------------
// N: the stream count at compile time (5 = the reference's default quadrature, Scatter_0.py:59; 8), 0 = any nmu <= kMsMaxMu at
// run time.  With N known the element -> (row, column) divisions become multiplications and the k-loops of the small products
// unroll (C4 size at 5 streams: 0.26 -> 0.21 s).  127 registers = four waves per SIMD; capped at 85 / 64 registers (six / eight
// waves, 0 / 55 spilled) the kernel is no faster (0.21 / 0.24 s): it is not occupancy that binds it.
// CACHE (the batched numerical Jacobian, ansfm_cirsrad_ck_scatter_batch; see k_ms_chain16): 1 = model 0's pass over the slab
// [w0, w0 + wcount) stores the doubled (r1, t1, j1) of every scattering layer, [wavenumber of the slab][g][order][layer]
// [2 n^2 + n]; 2 = the models [m0, m0 + n_launch) of a launch (grid x n_launch) take the layers flagged `same` from there and
// run the adding sweep only.  Same numbers either way.
template <int N, int CACHE = 0>
__global__ __launch_bounds__(64) void k_ms_chain(MsParams p)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    const int n = N ? N : p.nmu, nn = n * n;
    const int ld = n;
    const int msz = n * ld;
    const unsigned per_model = (unsigned)p.wcount * (unsigned)p.ng_launch * (unsigned)(p.nf + 1);
    const int ml = (CACHE == 2) ? (int)(blockIdx.x / per_model) : 0;           // position in the launch
    const unsigned rest = (CACHE == 2) ? blockIdx.x % per_model : blockIdx.x;
    const int mg = (CACHE == 2) ? p.model_ids[p.m0 + ml] : p.m0;               // model (0 outside the batch path)
    const int ic = rest % (p.nf + 1);
    const int ig = p.ig0 + (int)((rest / (p.nf + 1)) % p.ng_launch);      // the g-ordinates [ig0, ig0 + ng_launch) of this launch
    const int wl = rest / ((p.nf + 1) * p.ng_launch);                      // wavenumber within the slab (w0 = 0 outside the batch path)
    const int widx = p.w0 + wl;
    const double pi = 3.141592653589793;
    // taus / omegas / bnu: [model of the launch][wavenumber of the slab]; the other per-wavenumber arrays keep the whole axis
    const size_t wrow = (size_t)ml * p.wcount + wl;
    const double *taus_w = p.taus + (wrow * p.ng + ig) * p.nlay, *omegas_w = p.omegas + (wrow * p.ng + ig) * p.nlay;
    const double *bnu_w = p.bnu + wrow * p.nlay, *tauray_w = p.tauray + (size_t)mg * p.st_wl + (size_t)widx * p.nlay;
    const double *lfrac_m = p.lfrac + (size_t)mg * p.st_wcl;
    const double *radg_m = p.radg + (size_t)mg * p.st_wm;
    const size_t centry = (size_t)(2 * nn + n);
    // LDS carve-up
    double *rc = sm, *tc = rc + msz, *r1 = tc + msz, *t1 = r1 + msz, *pp = t1 + msz, *pm = pp + msz;
    double *m0 = pm + msz, *m1 = m0 + msz, *m2 = m1 + msz, *m3 = m2 + msz, *m4 = m3 + msz, *m5 = m4 + msz;
    double *jc = m5 + msz, *j1 = jc + kMsMaxMu, *v0 = j1 + kMsMaxMu, *v1 = v0 + kMsMaxMu, *col = v1 + kMsMaxMu;
    double *radg = col + kMsMaxMu;
    (void)m5;
#define MS_FOR_IJ for (int e = lane, i = e / n, j = e % n; e < nn; e += 64, i = e / n, j = e % n)
#define MS_AT(M, i, j) M[(i) * ld + (j)]

    if (lane < n) radg[lane] = radg_m[(size_t)widx * n + (n - 1 - lane)];   // radg[:, ::-1] :765
    MS_WAVE_SYNC();
    bool defined = false;
    const bool lookup = p.lookup != 0;
    if (p.lowbc > 0 && !lookup) {  // surface operator first :824-836
        MS_FOR_IJ {
            MS_AT(rc, i, j) = (2. * (p.brdf[(((size_t)widx * n + i) * n + j) * (p.nf + 1) + ic] * pi) * p.mu[j] * p.wtmu[j]) * p.xfac;
            MS_AT(tc, i, j) = 0.0;
        }
        if (lane < n) jc[lane] = radg[lane];
        defined = true;
        MS_WAVE_SYNC();
    }
    const double *PPL = p.ppl + (((size_t)widx * (p.nf + 1) + ic) * p.ncomp) * nn;
    const double *PMI = p.pmi + (((size_t)widx * (p.nf + 1) + ic) * p.ncomp) * nn;
    const double *FC = p.fc + (((size_t)ig * p.nwave + widx) * p.ncomp) * nn;   // ppl *= fc (:232)

    for (int l = 0; l < p.nlay; ++l) {
        const int k = lookup ? p.nlay - 1 - l : l;  // look-down: bottom layer first (:842-845)
        const double taut = taus_w[k];
        const double bc = bnu_w[k];
        double omega = omegas_w[k];
        if (omega < 0) omega = 0.0;
        if (omega > 1) omega = 1.0;
        double tauscat = taut * omega;
        const double taur = tauray_w[k];
        tauscat = tauscat - taur;
        if (tauscat < 0) tauscat = 0.0;
        // ---- calc_rtj_matrix :566-647 -> (r1, t1, j1), iscl ------------------------------------------------
        int iscl = 0;
        omega = (tauscat + taur) / taut;
        if (taut == 0) {
            MS_FOR_IJ { MS_AT(r1, i, j) = 0.0; MS_AT(t1, i, j) = (i == j) ? 1.0 : 0.0; }
            if (lane < n) j1[lane] = 0.0;
            MS_WAVE_SYNC();
        } else if (omega == 0) {
            MS_FOR_IJ { MS_AT(r1, i, j) = 0.0; MS_AT(t1, i, j) = 0.0; }
            MS_WAVE_SYNC();
            if (lane < n) {
                const double tex = -(1. / p.mu[lane]) * taut;
                const double tt = (tex > -200.0) ? exp(tex) : 0.0;
                MS_AT(t1, lane, lane) = tt;
                j1[lane] = bc * (1.0 - tt);
            }
            MS_WAVE_SYNC();
        } else if (CACHE == 2 && p.same[(size_t)mg * p.nlay + k]) {
            // the layer of model 0, as its own pass left it (block-uniform branch)
            iscl = 1;
            const double *ce = p.cache + ((((size_t)wl * p.ng + ig) * (p.nf + 1) + ic) * p.nlay + k) * centry;
            MS_FOR_IJ { MS_AT(r1, i, j) = ce[e]; MS_AT(t1, i, j) = ce[nn + e]; }
            if (lane < n) j1[lane] = ce[2 * nn + lane];
            MS_WAVE_SYNC();
        } else {
            iscl = 1;
            const double fr = taur / (tauscat + taur), fs = tauscat / (tauscat + taur);
            MS_FOR_IJ {
                double a = (p.iray > 0) ? fr * (PPL[(size_t)p.ncont * nn + e] * FC[(size_t)p.ncont * nn + e]) : 0.0;
                double b = (p.iray > 0) ? fr * PMI[(size_t)p.ncont * nn + e] : 0.0;
                for (int c = 0; c < p.ncont; ++c) {
                    const double f = lfrac_m[((size_t)widx * p.ncont + c) * p.nlay + k];
                    a += fs * (PPL[(size_t)c * nn + e] * FC[(size_t)c * nn + e]) * f;
                    b += fs * PMI[(size_t)c * nn + e] * f;
                }
                MS_AT(pp, i, j) = a;
                MS_AT(pm, i, j) = b;
            }
            MS_WAVE_SYNC();
            // ---- double1 :321-362 --------------------------------------------------------------------------
            double con = omega * pi;
            con *= (ic == 0) ? 2.0 : 1.0;
            const int nd = (int)(log2(taut) + 12);   // python int(): truncation toward zero
            const double tau0 = taut * ((nd >= 1) ? 1.0 / exp2((double)nd) : 1.0);
            // Gamma++ = M^-1 (E - con P++ C) ;  Gamma+- = M^-1 con P+- C   (C, M^-1 diagonal)
            MS_FOR_IJ {
                const double gpp = (1. / p.mu[i]) * (((i == j) ? 1.0 : 0.0) - (MS_AT(pp, i, j) * p.wtmu[j]) * con);
                const double gpm = (1. / p.mu[i]) * ((MS_AT(pm, i, j) * p.wtmu[j]) * con);
                MS_AT(t1, i, j) = ((i == j) ? 1.0 : 0.0) - tau0 * gpp;
                MS_AT(r1, i, j) = tau0 * gpm;
            }
            if (lane < n) j1[lane] = (ic == 0) ? (1.0 - omega) * bc * tau0 * (1. / p.mu[lane]) : 0.0;
            MS_WAVE_SYNC();
            for (int it = 0; it < nd; ++it) {   // add :275-297
                ms_mm(n, ld, r1, r1, m0, lane);               // bcom
                if (ms_frob(n, ld, r1, lane) > 0.1) {
                    MS_FOR_IJ MS_AT(m1, i, j) = ((i == j) ? 1.0 : 0.0) - MS_AT(m0, i, j);
                    MS_WAVE_SYNC();
                    ms_inv(n, ld, m1, m2, col, lane);             // acom = inv(e - bcom)
                } else {
                    MS_FOR_IJ MS_AT(m2, i, j) = ((i == j) ? 1.0 : 0.0) + MS_AT(m0, i, j);
                    MS_WAVE_SYNC();
                }
                ms_mm(n, ld, t1, m2, m3, lane);               // ccom = t1 acom
                ms_mm(n, ld, m3, r1, m0, lane);               // rans = ccom r1
                ms_mm(n, ld, m0, t1, m1, lane);               // acom = rans t1
                ms_mm(n, ld, m3, t1, m4, lane);               // tans = ccom t1
                if (ic == 0) {
                    ms_mv(n, ld, r1, j1, v0, lane);               // jcom = r1 j1 + j1
                    if (lane < n) v0[lane] = v0[lane] + j1[lane];
                    MS_WAVE_SYNC();
                    ms_mv(n, ld, m3, v0, v1, lane);               // jans = ccom jcom + j1
                    if (lane < n) j1[lane] = v1[lane] + j1[lane];
                }
                MS_FOR_IJ { MS_AT(r1, i, j) = MS_AT(r1, i, j) + MS_AT(m1, i, j); MS_AT(t1, i, j) = MS_AT(m4, i, j); }
                MS_WAVE_SYNC();
            }
            if constexpr (CACHE == 1) {
                double *ce = p.cache + ((((size_t)wl * p.ng + ig) * (p.nf + 1) + ic) * p.nlay + k) * centry;
                MS_FOR_IJ { ce[e] = MS_AT(r1, i, j); ce[nn + e] = MS_AT(t1, i, j); }
                if (lane < n) ce[2 * nn + lane] = j1[lane];
            }
        }
        // ---- combine with the stack below :868-875 ------------------------------------------------------------
        if (l == 0 && !defined) {
            MS_FOR_IJ { MS_AT(rc, i, j) = MS_AT(r1, i, j); MS_AT(tc, i, j) = MS_AT(t1, i, j); }
            if (lane < n) jc[lane] = j1[lane];
            MS_WAVE_SYNC();
        } else if (iscl == 1) {   // addp, scattering layer :486-511 (rsub,tsub,jsub) = (rc,tc,jc)
            ms_mm(n, ld, rc, r1, m0, lane);                   // rsq = rsub r1
            if (ms_frob(n, ld, m0, lane) > 0.01) {
                MS_FOR_IJ MS_AT(m1, i, j) = ((i == j) ? 1.0 : 0.0) - MS_AT(m0, i, j);
                MS_WAVE_SYNC();
                ms_inv(n, ld, m1, m2, col, lane);
            } else {
                MS_FOR_IJ MS_AT(m2, i, j) = ((i == j) ? 1.0 : 0.0) + MS_AT(m0, i, j);
                MS_WAVE_SYNC();
            }
            ms_mm(n, ld, t1, m2, m3, lane);                   // ccom = t1 acom
            ms_mm(n, ld, m3, rc, m0, lane);                   // rans = ccom rsub
            ms_mm(n, ld, m0, t1, m1, lane);                   // bcom = rans t1
            ms_mm(n, ld, m3, tc, m4, lane);                   // tans = ccom tsub
            ms_mv(n, ld, rc, j1, v0, lane);                       // jcom = rsub j1 + jsub
            if (lane < n) v0[lane] += jc[lane];
            MS_WAVE_SYNC();
            ms_mv(n, ld, m3, v0, v1, lane);                       // jans = ccom jcom + j1
            if (lane < n) jc[lane] = v1[lane] + j1[lane];
            MS_FOR_IJ { MS_AT(rc, i, j) = MS_AT(r1, i, j) + MS_AT(m1, i, j); MS_AT(tc, i, j) = MS_AT(m4, i, j); }
            MS_WAVE_SYNC();
        } else {                  // addp, non-scattering layer :513-530
            ms_mv(n, ld, rc, j1, v0, lane);
            if (lane < n) v0[lane] += jc[lane];
            MS_WAVE_SYNC();
            MS_FOR_IJ {
                const double ta = MS_AT(t1, i, i), tb = MS_AT(t1, j, j);
                MS_AT(m0, i, j) = MS_AT(tc, i, j) * ta;
                MS_AT(m1, i, j) = MS_AT(rc, i, j) * ta * tb;
            }
            if (lane < n) v1[lane] = j1[lane] + MS_AT(t1, lane, lane) * v0[lane];
            MS_WAVE_SYNC();
            MS_FOR_IJ { MS_AT(tc, i, j) = MS_AT(m0, i, j); MS_AT(rc, i, j) = MS_AT(m1, i, j); }
            if (lane < n) jc[lane] = v1[lane];
            MS_WAVE_SYNC();
        }
    }
    if (ic != 0 && lane < n) jc[lane] = 0.0;   // :881-882
    MS_WAVE_SYNC();
    if (lookup && p.lowbc > 0) {
        // idown (:366-420) with rb = rs, tb = 0, jb = radg (js is set for every ic, :822):
        //   upl = (E - rc rs)^-1 (tc u0+ + (rc radg + jc));   m3 = the inverse, v0 = rc radg + jc
        MS_FOR_IJ MS_AT(m0, i, j) = (2. * (p.brdf[(((size_t)widx * n + i) * n + j) * (p.nf + 1) + ic] * pi) * p.mu[j] * p.wtmu[j]) * p.xfac;
        MS_WAVE_SYNC();
        ms_mm(n, ld, rc, m0, m1, lane);
        MS_FOR_IJ MS_AT(m2, i, j) = ((i == j) ? 1.0 : 0.0) - MS_AT(m1, i, j);
        MS_WAVE_SYNC();
        ms_inv(n, ld, m2, m3, col, lane);
        ms_mv(n, ld, rc, radg, v0, lane);
        if (lane < n) v0[lane] = v0[lane] + jc[lane];
        MS_WAVE_SYNC();
    }
    // ---- per path: the four (mu0, mu) samples and the bilinear interpolation :886-945 ---------------------------
    if (lane < p.ngeom) {
        const int ipath = lane;
        const double sol_ang = p.sol_ang[ipath];
        const double emiss_ang = lookup ? 180. - p.emiss_ang[ipath] : p.emiss_ang[ipath];   // new_emi :900-903
        double zmu0, solar1;
        if (sol_ang > 90.0) { zmu0 = cos((180 - sol_ang) * pi / 180.0); solar1 = p.solar[widx] * 0.0; }
        else { zmu0 = cos(sol_ang * pi / 180.0); solar1 = p.solar[widx]; }
        const double zmu = cos(emiss_ang * pi / 180.0);
        int isol = 0, iemm = 0;
        for (int j = 0; j < n - 1; ++j) if (zmu0 <= p.mu[j] && zmu0 > p.mu[j + 1]) isol = j;
        if (zmu0 <= p.mu[n - 1]) isol = n - 2;
        for (int j = 0; j < n - 1; ++j) if (zmu <= p.mu[j] && zmu > p.mu[j + 1]) iemm = j;
        if (zmu <= p.mu[n - 1]) iemm = n - 2;
        const double u = (p.mu[isol] - zmu0) / (p.mu[isol] - p.mu[isol + 1]);
        const double t = (p.mu[iemm] - zmu) / (p.mu[iemm] - p.mu[iemm + 1]);
        double yx[4];
        int ico = 0;
        for (int imu0 = isol; imu0 < isol + 2; ++imu0) {
            const double s0 = solar1 / (2.0 * pi * p.wtmu[imu0]);
            for (int imu = iemm; imu < iemm + 2; ++imu) {
                if (!lookup) {
                    double bcom = 0.0;   // (T utmi)[imu], utmi = radg for ic == 0 else 0
                    if (ic == 0) for (int kk = 0; kk < n; ++kk) bcom += MS_AT(tc, imu, kk) * radg[kk];
                    yx[ico++] = (MS_AT(rc, imu, imu0) * s0 + bcom) + jc[imu];
                } else if (p.lowbc == 0) {   // bottom of the atmosphere: T u0+ + R u- + J  (:929-933)
                    double bcom = 0.0;
                    if (ic == 0) for (int kk = 0; kk < n; ++kk) bcom += MS_AT(rc, imu, kk) * radg[kk];
                    yx[ico++] = (MS_AT(tc, imu, imu0) * s0 + bcom) + jc[imu];
                } else {
                    double upl = 0.0;
                    for (int kk = 0; kk < n; ++kk) upl += MS_AT(m3, imu, kk) * (MS_AT(tc, kk, imu0) * s0 + v0[kk]);
                    yx[ico++] = upl;
                }
            }
        }
        double drad = ((1 - t) * (1 - u) * yx[0] + t * (1 - u) * yx[1] + t * u * yx[3] + (1 - t) * u * yx[2]) *
                      cos(ic * p.aphi[ipath] * pi / 180.0);
        if (ic > 0) drad *= 2;
        p.drad[(size_t)mg * p.st_drad + (((size_t)widx * p.ng + ig) * (p.nf + 1) + ic) * p.ngeom + ipath] = drad;
    }
#undef MS_FOR_IJ
#undef MS_AT
}

// ---- nmu == 16: the same chain on the matrix cores ----------------------------------------------------------
// Every 16x16 float64 matrix lives in LDS (leading dimension 17) and, while it is being worked on, in the
// MFMA accumulator layout ("D layout": lane (c = l&15, q = l>>4) holds rows q, q+4, q+8, q+12 of column c).
// v_mfma_f64_16x16x4_f64 takes B operands in exactly that layout, so products chain in registers; only a
// LEFT operand has to be read back from LDS transposed-wise (a[kb] = M[c][q+4kb]).  Mat-vec products reuse
// the a-operand registers (4 FMAs + a reduction over q).  One wavefront per block: LDS is in-order per
// wave, so a compiler fence replaces the barriers.
#define MS16_FENCE() __atomic_signal_fence(__ATOMIC_SEQ_CST)
struct Ms16 {
    int c, q;
    __device__ __forceinline__ void load_a(const double *M, double a[4]) const
    {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) a[kb] = M[c * 17 + q + 4 * kb];
    }
    __device__ __forceinline__ ms_v4f64 load_d(const double *M) const
    {
        ms_v4f64 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = M[(q + 4 * r) * 17 + c];
        return v;
    }
    __device__ __forceinline__ void store_d(double *M, ms_v4f64 v) const
    {
#pragma unroll
        for (int r = 0; r < 4; ++r) M[(q + 4 * r) * 17 + c] = v[r];
    }
    __device__ __forceinline__ static ms_v4f64 mm(const double a[4], ms_v4f64 b)
    {
        ms_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb], b[kb], acc, 0, 0, 0);
        return acc;
    }
    // (M x)[c] for the matrix whose a-operands are given; x in LDS.  Same value in the four lanes of column c.
    __device__ __forceinline__ double mv(const double a[4], const double *x) const
    {
        double s = 0.0;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) s += a[kb] * x[q + 4 * kb];
        return ms_xor32_sum(ms_xor16_sum(s));
    }
    // |v|_F^2, and the smallest doubles whose (correctly rounded) square root exceeds 0.1 / 0.01: `frob(v) > 0.1` is
    // `sumsq(v) >= kFrobSq01` without the square root in the chain
    static constexpr double kFrobSq01 = 0x1.47ae147ae147dp-7, kFrobSq001 = 0x1.a36e2eb1c432fp-14;
    __device__ __forceinline__ static double sumsq(ms_v4f64 v)
    {
        double s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        s += ms_dpp_f64<0x128>(s);
        s += ms_dpp_f64<0x124>(s);
        s += ms_dpp_f64<0x122>(s);
        s += ms_dpp_f64<0x121>(s);
        return ms_xor32_sum(ms_xor16_sum(s));
    }
    __device__ __forceinline__ static double frob(ms_v4f64 v)
    {
        double s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        s += ms_dpp_f64<0x128>(s);      // row_ror:8, 4, 2, 1: every lane ends with its row's sum
        s += ms_dpp_f64<0x124>(s);
        s += ms_dpp_f64<0x122>(s);
        s += ms_dpp_f64<0x121>(s);
        return sqrt(ms_xor32_sum(ms_xor16_sum(s)));
    }
    __device__ __forceinline__ ms_v4f64 eye_plus(ms_v4f64 v, double sgn) const
    {   // E + sgn * v
        ms_v4f64 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = ((q + 4 * r == c) ? 1.0 : 0.0) + sgn * v[r];
        return o;
    }
};

// inverse of the 16x16 matrix in LDS A (ld 17) -> Ainv; Gauss-Jordan, partial pivoting (first largest |.|).
// A is destroyed.  Lane (c0,q) updates rows q+4r of column c0 in both matrices.  Out of line (it is the rare fallback of
// the series inverse; inlined three times it cost the chain kernel 11 registers), pointers in the LDS address space.
typedef __attribute__((address_space(3))) double ms_lds_double;
__device__ __attribute__((noinline)) void ms_inv16_lds(ms_lds_double *A, ms_lds_double *Ainv, int lane)
{
    const int c0 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) Ainv[(q + 4 * r) * 17 + c0] = (q + 4 * r == c0) ? 1.0 : 0.0;
    MS16_FENCE();
    for (int c = 0; c < 16; ++c) {
        double best = (c0 >= c) ? fabs(A[c0 * 17 + c]) : -1.0;    // every 16-lane row group searches alike
        int piv = c0;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int op = __shfl_xor(piv, off, 64);
            if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
        }
        piv = __builtin_amdgcn_readfirstlane(piv);
        // rows c and piv of both matrices: swap, then scale the pivot row
        const double ac = A[c * 17 + c0], wc = Ainv[c * 17 + c0];
        const double ap = A[piv * 17 + c0], wp = Ainv[piv * 17 + c0];
        const double d = 1.0 / A[piv * 17 + c];
        const double pa = ap * d, pw = wp * d;
        MS16_FENCE();
        if (q == 0) {
            A[piv * 17 + c0] = ac; Ainv[piv * 17 + c0] = wc;      // no-op content when piv == c, overwritten next
            A[c * 17 + c0] = pa; Ainv[c * 17 + c0] = pw;
        }
        MS16_FENCE();
        double f[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) f[r] = A[(q + 4 * r) * 17 + c];
        MS16_FENCE();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q + 4 * r;
            if (row != c) {
                A[row * 17 + c0] -= f[r] * pa;
                Ainv[row * 17 + c0] -= f[r] * pw;
            }
        }
        MS16_FENCE();
    }
}
__device__ __forceinline__ void ms_inv16(double *A, double *Ainv, int lane)
{
    ms_inv16_lds((ms_lds_double *)A, (ms_lds_double *)Ainv, lane);
}

// (E - B)^-1 for a 16x16 B given in D layout, on the matrix cores: the product form of the Neumann series,
//     (E - B)^-1 = (E + B)(E + B^2)(E + B^4)...,
// two MFMA products per factor; it stops when the next power is below 1e-17 in Frobenius norm (B = R R' of physical
// reflection operators has spectral radius < 1: 6 factors at |B| = 0.5, 9 at 0.9).  Returns false when it has not
// converged after 12 squarings (the caller then falls back on Gauss-Jordan).  mA / mB: LDS scratch matrices.
__device__ __forceinline__ bool ms_inv16_series(const Ms16 &L, ms_v4f64 B, double *mA, double *mB, ms_v4f64 &X)
{
    X = L.eye_plus(B, 1.0);
    ms_v4f64 Pw = B;
    double fP = Ms16::frob(B);
    for (int it = 0; it < 12; ++it) {
        double aP[4], aX[4];
        if (fP * fP < 1e-17) return true;                      // |P^2|_F <= |P|_F^2: the next power would be dropped anyway
        L.store_d(mA, Pw);
        L.store_d(mB, X);
        MS16_FENCE();
        L.load_a(mA, aP);
        const ms_v4f64 P2 = Ms16::mm(aP, Pw);                  // B^(2^(it+1))
        fP = Ms16::frob(P2);
        if (!(fP >= 1e-17)) return true;                       // also leaves on NaN
        L.load_a(mB, aX);
        const ms_v4f64 XP = Ms16::mm(aX, P2);
#pragma unroll
        for (int r = 0; r < 4; ++r) X[r] = X[r] + XP[r];       // X (E + P2)
        Pw = P2;
        MS16_FENCE();
    }
    return false;
}

// libm calls of the chain kernel, kept out of line: inlined, their polynomial constants and temporaries count towards the
// kernel's register allocation (log2 / exp2: 14 registers, exp: 14, cos: more) and it is the allocation, not the pressure inside
// the product loops, that decides how many waves share a SIMD.
__device__ __attribute__((noinline)) double ms_log2_ni(double x) { return log2(x); }
__device__ __attribute__((noinline)) double ms_exp2_ni(double x) { return exp2(x); }
__device__ __attribute__((noinline)) double ms_exp_ni(double x) { return exp(x); }
__device__ __attribute__((noinline)) double ms_cos_ni(double x) { return cos(x); }

// Register allocation and LDS decide how many waves share a SIMD, and a wave of this kernel spends half of its time waiting
// to issue (dependent products on a shared MFMA pipe).  History at C4 (1e4 wavenumbers): 280 registers after the Fourier-order
// loop moved into the block = ONE wave per SIMD, 1.05 s; capped at 256 = two, 0.73 s; libm calls and the Gauss-Jordan fallback
// out of line (they count towards the allocation: log2 / exp2 14 registers, exp 14, cos more), the inverse's scratch aliased
// onto r1 / t1 (4 instead of 6 LDS matrices) and a cap of 168 = three waves per SIMD, 0.59-0.60 s.  PHASE_LDS = true keeps the
// phase matrices of the Fourier order in LDS (17.5 KB: nine blocks per CU, no spills); PHASE_LDS = false reads them from HBM / L2
// in every layer (9.3 KB: twelve blocks, 65 registers spilled in the layer set-up) and is 2-4 % faster: the default.
// CACHE (the batched numerical Jacobian of the scattering configuration, ForwardModel_0.py:2251-2252: NX + 1 forward models
// that differ from the first in a few layers): the doubled (r, t, j) of a layer (calc_rtj_matrix :566-650) depend on that
// layer's inputs only and are 92 % of a chain's work (about twelve doublings of five products and an inverse each, against
// one adding step).  CACHE = 1: model 0's pass stores them per (wavenumber, g, Fourier order, layer) -- in the accumulator
// layout, 64 lanes x 8 doubles, coalesced -- and the number of orders it worked through.  CACHE = 2: one block per (model,
// wavenumber, g) re-runs the adding sweep (addp :481-533) and takes every layer whose inputs are bit-identical to model 0's
// from that cache; a changed layer, or an order beyond the cached ones, is computed as usual.  Same numbers either way.
template <bool PHASE_LDS, int CACHE = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_ms_chain16(MsParams p)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    constexpr int n = 16, nn = 256, ld = 17, msz = 16 * 17;
    const Ms16 L{lane & 15, lane >> 4};
    const int c = L.c, q = L.q;
    int ig, widx, ml = 0;
    if constexpr (CACHE == 2) {
        // blocks of one (wavenumber, g) pair -- they read the same cache lines -- sit 8 apart in the launch order: consecutive
        // workgroups go round the 8 XCDs, so the models of a pair share ONE XCD's L2
        const long grp = (long)(blockIdx.x >> 3), slot = (long)(blockIdx.x & 7);
        const long pair = (grp / p.n_launch) * 8 + slot;
        ml = (int)(grp % p.n_launch);
        if (pair >= (long)p.wcount * p.ng) return;
        ig = (int)(pair % p.ng);
        widx = p.w0 + (int)(pair / p.ng);
    } else {
        ig = p.ig0 + (int)(blockIdx.x % p.ng_launch);
        widx = p.w0 + (int)(blockIdx.x / p.ng_launch);
    }
    int mg = p.m0 + ml;
    if constexpr (CACHE == 2) mg = p.model_ids[p.m0 + ml];
    const int wl = widx - p.w0;
    int ncached = 0;
    if constexpr (CACHE == 2) ncached = p.cache_orders[(size_t)wl * p.ng + ig];
    int ndone = 0;
    const double pi = 3.141592653589793;
    // rc / tc: the stack below; r1 / t1: the layer's operators as LEFT operands (their right-operand form stays in registers).
    // The same two matrices are the scratch of the inverse and of the chained products (mA / mB): every step reads its left
    // operands out of r1 / t1 first and stores the new r1 / t1 last.
    double *rc = sm, *tc = rc + msz, *r1 = tc + msz, *t1 = r1 + msz, *mA = r1, *mB = t1;
    double *jc = t1 + msz, *j1 = jc + 16, *v0 = j1 + 16, *mus = v0 + 16, *wts = mus + 16;
    // radg[:, ::-1] (:765) sits in j1 once the layer loop is done
    double *radg = j1;
    // phase matrices of this Fourier order (P++ times the Hansen factor, P+-) by component, lane-private [comp][2][4][64]:
    // they do not depend on the layer (PHASE_LDS only)
    double *phl = wts + 16;
#define MS_AT(M, i, j) M[(i) * ld + (j)]

    // quadrature points / weights for per-lane indexing: a kernel-argument array indexed per lane would be copied to scratch
    // memory (and a second copy of this loop further down costs 64 registers: the scalar loads are hoisted and kept)
    for (int kk = 0; kk < n; ++kk)
        if (lane == kk) { mus[kk] = p.mu[kk]; wts[kk] = p.wtmu[kk]; }
    MS16_FENCE();
    const bool lookup = p.lookup != 0;
    const double mu_c = mus[c], wt_c = wts[c];
    const double rmu_c = 1. / mu_c;
    double rmu_i[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rmu_i[r] = 1. / mus[q + 4 * r];
    // Fourier sum of every path with the reference's early-out (:903-958), kept by lane `ipath`: the orders are worked
    // through one after the other by this block, and it stops as soon as every path has converged -- the reference builds the
    // operators of all NF + 1 orders first (:790) and never reads the ones beyond the break.
    double frad = 0.0;
    bool fconv1 = false, fdone = (lane >= p.ngeom);
    for (int ic = 0; ic <= p.nf; ++ic) {
    bool defined = false;
    if (p.lowbc > 0 && !lookup) {  // surface operator first :824-836
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = q + 4 * r, j = c;
            MS_AT(rc, i, j) = (2. * (p.brdf[(((size_t)widx * n + i) * n + j) * (p.nf + 1) + ic] * pi) * mu_c * wt_c) * p.xfac;
            MS_AT(tc, i, j) = 0.0;
        }
        if (lane < n) jc[lane] = p.radg[(size_t)mg * p.st_wm + (size_t)widx * n + (n - 1 - lane)];
        defined = true;
    }
    MS16_FENCE();
    const double *PPL = p.ppl + (((size_t)widx * (p.nf + 1) + ic) * p.ncomp) * nn;
    const double *PMI = p.pmi + (((size_t)widx * (p.nf + 1) + ic) * p.ncomp) * nn;
    const double *FC = p.fc + (((size_t)ig * p.nwave + widx) * p.ncomp) * nn;   // ppl *= fc (:232)
    const int ncu = p.ncont + (p.iray > 0 ? 1 : 0);      // components in use: the aerosols, then Rayleigh
    if constexpr (PHASE_LDS) {
        for (int cc = 0; cc < ncu; ++cc)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = (q + 4 * r) * 16 + c;
                phl[((cc * 2 + 0) * 4 + r) * 64 + lane] = PPL[(size_t)cc * nn + e] * FC[(size_t)cc * nn + e];
                phl[((cc * 2 + 1) * 4 + r) * 64 + lane] = PMI[(size_t)cc * nn + e];
            }
        MS16_FENCE();
    }

    // the layer's scalars are fetched one layer ahead
    const size_t wrow = (size_t)ml * p.wcount + wl;       // taus / omegas / bnu: [model of the launch][wavenumber of the slab]
    const double *taus_w = p.taus + (wrow * p.ng + ig) * p.nlay, *omegas_w = p.omegas + (wrow * p.ng + ig) * p.nlay;
    const double *bnu_w = p.bnu + wrow * p.nlay, *tauray_w = p.tauray + (size_t)mg * p.st_wl + (size_t)widx * p.nlay;
    const double *lfrac_m = p.lfrac + (size_t)mg * p.st_wcl;
    // CACHE = 2: every layer of the sweep below lstart is model 0's, and so is the stack they add up to (the lower boundary
    // included: the host leaves lstart at 0 when the boundary radiance differs) -- take it from model 0's pass and start there
    int lbeg = 0;
    if constexpr (CACHE == 2) {
        if (ic < ncached) lbeg = p.lstart[mg];
        if (lbeg > 0) {
            const double *pe = p.pcache + ((((size_t)wl * p.ng + ig) * (p.nf + 1) + ic) * p.npre + (lbeg / kMsPrefixStep - 1)) * kMsCacheEntry;
            ms_v4f64 sR, sT;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sR[r] = pe[r * 64 + lane]; sT[r] = pe[(4 + r) * 64 + lane]; }
            const double sj = pe[512 + c];
            MS16_FENCE();
            L.store_d(rc, sR); L.store_d(tc, sT);
            if (q == 0) jc[c] = sj;
            MS16_FENCE();
            defined = true;
        }
    }
    const int kfirst = lookup ? p.nlay - 1 - lbeg : lbeg;
    double n_taut = taus_w[kfirst], n_bc = bnu_w[kfirst], n_omega = omegas_w[kfirst], n_taur = tauray_w[kfirst];
    for (int l = lbeg; l < p.nlay; ++l) {
        const int k = lookup ? p.nlay - 1 - l : l;  // look-down: bottom layer first (:842-845)
        const double taut = n_taut, bc = n_bc;
        double omega = n_omega;
        const double taur_l = n_taur;
        if (l + 1 < p.nlay) {
            const int k1 = lookup ? k - 1 : k + 1;
            n_taut = taus_w[k1]; n_bc = bnu_w[k1]; n_omega = omegas_w[k1]; n_taur = tauray_w[k1];
        }
        if (omega < 0) omega = 0.0;
        if (omega > 1) omega = 1.0;
        double tauscat = taut * omega;
        const double taur = taur_l;
        tauscat = tauscat - taur;
        if (tauscat < 0) tauscat = 0.0;
        // ---- calc_rtj_matrix :566-647 -> (r1, t1, j1), iscl ------------------------------------------------
        int iscl = 0;
        omega = (tauscat + taur) / taut;
        if (taut == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { MS_AT(r1, q + 4 * r, c) = 0.0; MS_AT(t1, q + 4 * r, c) = (q + 4 * r == c) ? 1.0 : 0.0; }
            if (lane < n) j1[lane] = 0.0;
            MS16_FENCE();
        } else if (omega == 0) {
            const double tex = -rmu_c * taut;
            const double tt = (tex > -200.0) ? ms_exp_ni(tex) : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { MS_AT(r1, q + 4 * r, c) = 0.0; MS_AT(t1, q + 4 * r, c) = (q + 4 * r == c) ? tt : 0.0; }
            if (lane < n) j1[lane] = bc * (1.0 - tt);
            MS16_FENCE();
        } else if (CACHE == 2 && ic < ncached && p.same[(size_t)mg * p.nlay + k]) {
            // the layer of model 0, as its own pass left it (wave-uniform branch)
            iscl = 1;
            const double *ce = p.cache + ((((size_t)wl * p.ng + ig) * (p.nf + 1) + ic) * p.nlay + k) * kMsCacheEntry;
            ms_v4f64 bR, bT;
#pragma unroll
            for (int r = 0; r < 4; ++r) { bR[r] = ce[r * 64 + lane]; bT[r] = ce[(4 + r) * 64 + lane]; }
            const double jv = ce[512 + c];
            L.store_d(r1, bR); L.store_d(t1, bT);
            if (q == 0) j1[c] = jv;
            MS16_FENCE();
        } else {
            iscl = 1;
            const double fr = taur / (tauscat + taur), fs = tauscat / (tauscat + taur);
            // ---- double1 :321-362: starting (r,t,j) of the 2^-nd sub-layer, built straight in D layout -----
            double con = omega * pi;
            con *= (ic == 0) ? 2.0 : 1.0;
            const int nd = (int)(ms_log2_ni(taut) + 12);   // python int(): truncation toward zero
            const double tau0 = taut * ((nd >= 1) ? 1.0 / ms_exp2_ni((double)nd) : 1.0);
            ms_v4f64 bR, bT;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = q + 4 * r, j = c, e = i * 16 + j;
                double a, b;
                if constexpr (PHASE_LDS) {
                    a = (p.iray > 0) ? fr * phl[((p.ncont * 2 + 0) * 4 + r) * 64 + lane] : 0.0;
                    b = (p.iray > 0) ? fr * phl[((p.ncont * 2 + 1) * 4 + r) * 64 + lane] : 0.0;
                    for (int cc = 0; cc < p.ncont; ++cc) {
                        const double f = lfrac_m[((size_t)widx * p.ncont + cc) * p.nlay + k];
                        a += fs * phl[((cc * 2 + 0) * 4 + r) * 64 + lane] * f;
                        b += fs * phl[((cc * 2 + 1) * 4 + r) * 64 + lane] * f;
                    }
                } else {
                    // one uniform base per array and component, 32-bit byte offsets: no 64-bit address pair per element
                    const unsigned eo = (unsigned)e * 8u;
                    auto at = [&](const double *base, int cc) {
                        return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base + (size_t)cc * nn) + eo);
                    };
                    a = (p.iray > 0) ? fr * (at(PPL, p.ncont) * at(FC, p.ncont)) : 0.0;
                    b = (p.iray > 0) ? fr * at(PMI, p.ncont) : 0.0;
                    for (int cc = 0; cc < p.ncont; ++cc) {
                        const double f = lfrac_m[((size_t)widx * p.ncont + cc) * p.nlay + k];
                        a += fs * (at(PPL, cc) * at(FC, cc)) * f;
                        b += fs * at(PMI, cc) * f;
                    }
                }
                // Gamma++ = M^-1 (E - con P++ C) ;  Gamma+- = M^-1 con P+- C   (C, M^-1 diagonal)
                const double gpp = rmu_i[r] * (((i == j) ? 1.0 : 0.0) - (a * wt_c) * con);
                const double gpm = rmu_i[r] * ((b * wt_c) * con);
                bT[r] = ((i == j) ? 1.0 : 0.0) - tau0 * gpp;
                bR[r] = tau0 * gpm;
            }
            double jv = (ic == 0) ? (1.0 - omega) * bc * tau0 * rmu_c : 0.0;    // j1[c], same in the 4 lanes of c
            L.store_d(r1, bR); L.store_d(t1, bT);
            if (q == 0) j1[c] = jv;
            MS16_FENCE();
            for (int it = 0; it < nd; ++it) {   // add :275-297
                double aR[4], aT[4], aC[4];
                L.load_a(r1, aR);
                L.load_a(t1, aT);                                             // before r1 / t1 serve as scratch
                const ms_v4f64 bcom = Ms16::mm(aR, bR);                       // r1 r1
                // rans t1 = (ccom r1) t1 is formed as ccom (r1 t1): r1 t1 needs nothing of the inverse and keeps the matrix
                // cores busy while it is worked out, and ccom is the only intermediate that has to change layout (one LDS
                // round trip instead of two).  Same products, other association: the last bits differ from the reference's order.
                const ms_v4f64 r1t1 = Ms16::mm(aR, bT);                       // r1 t1
                ms_v4f64 acom;
                if (Ms16::sumsq(bR) >= Ms16::kFrobSq01) {                     // |r1|_F > 0.1: inv(e - bcom)
                    if (!ms_inv16_series(L, bcom, mA, mB, acom)) {
                        L.store_d(mA, L.eye_plus(bcom, -1.0));
                        MS16_FENCE();
                        ms_inv16(mA, mB, lane);
                        acom = L.load_d(mB);
                    }
                } else
                    acom = L.eye_plus(bcom, 1.0);
                const ms_v4f64 ccom = Ms16::mm(aT, acom);                     // t1 acom
                L.store_d(mB, ccom);
                double jcom = 0.0;
                if (ic == 0) {
                    jcom = L.mv(aR, j1) + jv;                                 // r1 j1 + j1
                    if (q == 0) v0[c] = jcom;
                }
                MS16_FENCE();
                L.load_a(mB, aC);
                const ms_v4f64 tans = Ms16::mm(aC, bT);                       // ccom t1
                const ms_v4f64 acc = Ms16::mm(aC, r1t1);                      // ccom (r1 t1) = rans t1
                if (ic == 0) jv = L.mv(aC, v0) + jv;                          // ccom jcom + j1
#pragma unroll
                for (int r = 0; r < 4; ++r) bR[r] = bR[r] + acc[r];
                bT = tans;
                MS16_FENCE();
                L.store_d(r1, bR); L.store_d(t1, bT);
                if (q == 0) j1[c] = jv;
                MS16_FENCE();
            }
            if constexpr (CACHE == 1) {
                double *ce = p.cache + ((((size_t)wl * p.ng + ig) * (p.nf + 1) + ic) * p.nlay + k) * kMsCacheEntry;
#pragma unroll
                for (int r = 0; r < 4; ++r) { ce[r * 64 + lane] = bR[r]; ce[(4 + r) * 64 + lane] = bT[r]; }
                if (q == 0) ce[512 + c] = jv;
            }
        }
        // ---- combine with the stack below :868-875 ------------------------------------------------------------
        if (l == 0 && !defined) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { MS_AT(rc, q + 4 * r, c) = MS_AT(r1, q + 4 * r, c); MS_AT(tc, q + 4 * r, c) = MS_AT(t1, q + 4 * r, c); }
            if (lane < n) jc[lane] = j1[lane];
            MS16_FENCE();
        } else if (iscl == 1) {   // addp, scattering layer :486-511 (rsub,tsub,jsub) = (rc,tc,jc)
            double aRc[4], aT[4], aC[4];
            const ms_v4f64 bR1 = L.load_d(r1), bT1 = L.load_d(t1), bTc = L.load_d(tc);
            const double j1v = j1[c];
            L.load_a(rc, aRc);
            L.load_a(t1, aT);                                                 // before r1 / t1 serve as scratch
            const ms_v4f64 rsq = Ms16::mm(aRc, bR1);                          // rsub r1
            const ms_v4f64 rst1 = Ms16::mm(aRc, bT1);                         // rsub t1: rans t1 = ccom (rsub t1), as in the doubling
            ms_v4f64 acom;
            if (Ms16::sumsq(rsq) >= Ms16::kFrobSq001) {                       // |rsq|_F > 0.01
                if (!ms_inv16_series(L, rsq, mA, mB, acom)) {
                    L.store_d(mA, L.eye_plus(rsq, -1.0));
                    MS16_FENCE();
                    ms_inv16(mA, mB, lane);
                    acom = L.load_d(mB);
                }
            } else
                acom = L.eye_plus(rsq, 1.0);
            const ms_v4f64 ccom = Ms16::mm(aT, acom);                         // t1 acom
            L.store_d(mB, ccom);
            const double jcom = L.mv(aRc, j1) + jc[c];                        // rsub j1 + jsub
            if (q == 0) v0[c] = jcom;
            MS16_FENCE();
            L.load_a(mB, aC);
            const ms_v4f64 tans = Ms16::mm(aC, bTc);                          // ccom tsub
            const ms_v4f64 bcomm = Ms16::mm(aC, rst1);                        // ccom (rsub t1) = rans t1
            const double jans = L.mv(aC, v0) + j1v;                           // ccom jcom + j1
            ms_v4f64 rnew;
#pragma unroll
            for (int r = 0; r < 4; ++r) rnew[r] = bR1[r] + bcomm[r];
            MS16_FENCE();
            L.store_d(rc, rnew); L.store_d(tc, tans);
            if (q == 0) jc[c] = jans;
            MS16_FENCE();
        } else {                  // addp, non-scattering layer :513-530
            double aRc[4];
            L.load_a(rc, aRc);
            const double jcom = L.mv(aRc, j1) + jc[c];
            const double tcc = MS_AT(t1, c, c);
            const double jn = j1[c] + tcc * jcom;
            ms_v4f64 tn, rn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = q + 4 * r;
                const double ta = MS_AT(t1, i, i);
                tn[r] = MS_AT(tc, i, c) * ta;
                rn[r] = MS_AT(rc, i, c) * ta * tcc;
            }
            MS16_FENCE();
            L.store_d(tc, tn); L.store_d(rc, rn);
            if (q == 0) jc[c] = jn;
            MS16_FENCE();
        }
        if constexpr (CACHE == 1) {
            if ((l % kMsPrefixStep) == kMsPrefixStep - 1 && l / kMsPrefixStep < p.npre) {
                double *pe = p.pcache + ((((size_t)wl * p.ng + ig) * (p.nf + 1) + ic) * p.npre + l / kMsPrefixStep) * kMsCacheEntry;
                const ms_v4f64 sR = L.load_d(rc), sT = L.load_d(tc);
#pragma unroll
                for (int r = 0; r < 4; ++r) { pe[r * 64 + lane] = sR[r]; pe[(4 + r) * 64 + lane] = sT[r]; }
                if (q == 0) pe[512 + c] = jc[c];
            }
        }
    }
    if (ic != 0 && lane < n) jc[lane] = 0.0;   // :881-882
    if (lane < n) radg[lane] = p.radg[(size_t)mg * p.st_wm + (size_t)widx * n + (n - 1 - lane)];   // j1 is free until the next order's first layer
    __syncthreads();
    if (lookup && p.lowbc > 0) {
        // idown (:366-420) with rb = rs, tb = 0, jb = radg (js is set for every ic, :822):
        //   upl = (E - rc rs)^-1 (tc u0+ + (rc radg + jc));   mB = the inverse, v0 = rc radg + jc
        ms_v4f64 rs;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = q + 4 * r, j = c;
            rs[r] = (2. * (p.brdf[(((size_t)widx * n + i) * n + j) * (p.nf + 1) + ic] * pi) * mu_c * wt_c) * p.xfac;
        }
        double aRc[4];
        L.load_a(rc, aRc);
        L.store_d(mA, L.eye_plus(Ms16::mm(aRc, rs), -1.0));
        MS16_FENCE();
        ms_inv16(mA, mB, lane);
        const double wv = L.mv(aRc, radg) + jc[c];
        if (q == 0) v0[c] = wv;
        __syncthreads();
    }
    // ---- per path: the four (mu0, mu) samples and the bilinear interpolation :886-945 ---------------------------
    if (lane < p.ngeom) {
        const int ipath = lane;
        const double sol_ang = p.sol_ang[ipath];
        const double emiss_ang = lookup ? 180. - p.emiss_ang[ipath] : p.emiss_ang[ipath];   // new_emi :900-903
        double zmu0, solar1;
        if (sol_ang > 90.0) { zmu0 = ms_cos_ni((180 - sol_ang) * pi / 180.0); solar1 = p.solar[widx] * 0.0; }
        else { zmu0 = ms_cos_ni(sol_ang * pi / 180.0); solar1 = p.solar[widx]; }
        const double zmu = ms_cos_ni(emiss_ang * pi / 180.0);
        int isol = 0, iemm = 0;
        const int nq = p.nmu_real ? p.nmu_real : n;     // the quadrature's points (a smaller quadrature padded to 16 streams)
        for (int j = 0; j < nq - 1; ++j) if (zmu0 <= mus[j] && zmu0 > mus[j + 1]) isol = j;
        if (zmu0 <= mus[nq - 1]) isol = nq - 2;
        for (int j = 0; j < nq - 1; ++j) if (zmu <= mus[j] && zmu > mus[j + 1]) iemm = j;
        if (zmu <= mus[nq - 1]) iemm = nq - 2;
        const double u = (mus[isol] - zmu0) / (mus[isol] - mus[isol + 1]);
        const double t = (mus[iemm] - zmu) / (mus[iemm] - mus[iemm + 1]);
        double yx[4];
        int ico = 0;
        for (int imu0 = isol; imu0 < isol + 2; ++imu0) {
            const double s0 = solar1 / (2.0 * pi * wts[imu0]);
            for (int imu = iemm; imu < iemm + 2; ++imu) {
                if (!lookup) {
                    double bcom = 0.0;   // (T utmi)[imu], utmi = radg for ic == 0 else 0
                    if (ic == 0) for (int kk = 0; kk < n; ++kk) bcom += MS_AT(tc, imu, kk) * radg[kk];
                    yx[ico++] = (MS_AT(rc, imu, imu0) * s0 + bcom) + jc[imu];
                } else if (p.lowbc == 0) {   // bottom of the atmosphere: T u0+ + R u- + J  (:929-933)
                    double bcom = 0.0;
                    if (ic == 0) for (int kk = 0; kk < n; ++kk) bcom += MS_AT(rc, imu, kk) * radg[kk];
                    yx[ico++] = (MS_AT(tc, imu, imu0) * s0 + bcom) + jc[imu];
                } else {
                    double upl = 0.0;
                    for (int kk = 0; kk < n; ++kk) upl += MS_AT(mB, imu, kk) * (MS_AT(tc, kk, imu0) * s0 + v0[kk]);
                    yx[ico++] = upl;
                }
            }
        }
        double drad = ((1 - t) * (1 - u) * yx[0] + t * (1 - u) * yx[1] + t * u * yx[3] + (1 - t) * u * yx[2]) *
                      ms_cos_ni(ic * p.aphi[ipath] * pi / 180.0);
        if (ic > 0) drad *= 2;
        if (!fdone) {                                   // :945-958
            frad += drad;
            const double conv = fabs(drad / frad);
            if (conv < 1e-5 && fconv1) fdone = true;
            fconv1 = (conv < 1e-5);
        }
    }
    ++ndone;
    if (__builtin_amdgcn_ballot_w64(!fdone) == 0) break;
    __syncthreads();                                    // the LDS matrices are rebuilt by the next order
    }
    if constexpr (CACHE == 1) { if (lane == 0) p.cache_orders[(size_t)wl * p.ng + ig] = ndone; }
    if (lane < p.ngeom) p.rad[(size_t)mg * p.st_rad + ((size_t)lane * p.ng + ig) * p.nwave + widx] = frad;
#undef MS_AT
}

// ---- Fourier sum with the reference's convergence early-out :903-958 -----------------------------------------
__global__ void k_ms_fourier(MsParams p)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)p.nwave * p.ng * p.ngeom;
    if (idx >= total) return;
    const int ipath = (int)(idx % p.ngeom);
    const int ig = (int)((idx / p.ngeom) % p.ng);
    const int widx = (int)(idx / ((size_t)p.ngeom * p.ng));
    const double *d = p.drad + (((size_t)widx * p.ng + ig) * (p.nf + 1)) * p.ngeom + ipath;
    double rad = 0.0;
    bool conv1 = false;
    for (int ic = 0; ic <= p.nf; ++ic) {
        const double drad = d[(size_t)ic * p.ngeom];
        rad += drad;
        const double conv = fabs(drad / rad);
        if (conv < 1e-5 && conv1) break;
        conv1 = (conv < 1e-5);
    }
    p.rad[((size_t)ipath * p.ng + ig) * p.nwave + widx] = rad;
}


// ---- scattering branch of CIRSrad on the device -----------------------------------------------------------------------
// calculate_layer_opacity (ForwardModel_0.py:3989) and the host preparation of scloud11wave (:5099-5119) for the arrays that
// have a g axis: TAUTOT = TAUGAS + TAUCIA + TAUDUST + TAURAY, OMEGA = (TAURAY + TAUSCAT) / TAUTOT where TAUTOT > 0, and
// BB = planck(TEMP) -- from the merge kernel's [L][G][Wpad] gas opacities straight into the layouts the chain kernels
// read, so that TAUGAS / TAUTOT / OMEGA (NWAVE x NG x NLAY each) never leave HBM.  One thread per (wavenumber, layer).
struct MsOpticsParams {
    const double *taugas;       // [L][G][Wpad]
    const double *taucia;       // [W][L] or null
    const double *taudust;      // [W][L] aerosol extinction summed over the populations, or null
    const double *tauray;       // [W][L] or null
    const double *tauscat;      // [W][L] or null
    const double *wave;         // [W]
    const double *lay_temp;     // [L]
    double *taus, *omegas;      // [W][G][L]
    double *bnu;                // [W][L]
    int W, Wpad, G, L, ispace;
};

__global__ __launch_bounds__(128) void k_ms_optics(MsOpticsParams p)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
    if (w >= p.W) return;
    const size_t wl = (size_t)w * p.L + l;
    const double cia = p.taucia ? p.taucia[wl] : 0.0, dust = p.taudust ? p.taudust[wl] : 0.0;
    const double ray = p.tauray ? p.tauray[wl] : 0.0, sca = p.tauscat ? p.tauscat[wl] : 0.0;
    for (int g = 0; g < p.G; ++g) {
        const double tt = ((p.taugas[((size_t)l * p.G + g) * p.Wpad + w] + cia) + dust) + ray;      // the sum order of :3989
        const size_t o = ((size_t)w * p.G + g) * p.L + l;
        p.taus[o] = tt;
        p.omegas[o] = (tt > 0.0) ? (ray + sca) / tt : 0.0;
    }
    const double c1 = 1.1911e-12, c2 = 1.439;                     // ForwardModel_0.py:6214-6215
    const double wv = p.wave[w];
    double y, a;
    if (p.ispace == 0) { y = wv; a = c1 * (y * y * y); }
    else { y = 1.0e4 / wv; a = c1 * (y * y * y * y * y) / 1.0e4; }
    p.bnu[wl] = a / (exp(c2 * y / p.lay_temp[l]) - 1.0);
}

// The same for the models [m0, m0 + nm) of a batch on the wavenumbers [w0, w0 + wcount): the gas opacity of (model, layer)
// is row slot[model][layer] of the merge kernel's output (rows shared with model 0 where the layer inputs are identical),
// the continuum arrays carry a model axis, the outputs are laid out [model of the launch][wavenumber of the slab][..].
struct MsOpticsBatchParams {
    const double *taugas;       // [rows][G][Wpad]
    const int32_t *slot;        // [n][L]
    const double *taucia, *taudust, *tauray, *tauscat;   // [n][W][L] or null
    const double *wave;         // [W]
    const double *lay_temp;     // [n][L]
    double *taus, *omegas;      // [nm][wcount][G][L]
    double *bnu;                // [nm][wcount][L]
    const int *model_ids;       // launch position -> model (null: position m0 + ml is the model)
    int W, Wpad, G, L, ispace, w0, wcount, m0, nm;
};
__global__ __launch_bounds__(128) void k_ms_optics_batch(MsOpticsBatchParams p)
{
    const int wl = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y, ml = blockIdx.z;
    if (wl >= p.wcount) return;
    const int w = p.w0 + wl, m = p.model_ids ? p.model_ids[p.m0 + ml] : p.m0 + ml;
    const size_t in = ((size_t)m * p.W + w) * p.L + l;
    const double cia = p.taucia ? p.taucia[in] : 0.0, dust = p.taudust ? p.taudust[in] : 0.0;
    const double ray = p.tauray ? p.tauray[in] : 0.0, sca = p.tauscat ? p.tauscat[in] : 0.0;
    const size_t row = (size_t)p.slot[(size_t)m * p.L + l];
    const size_t wrow = (size_t)ml * p.wcount + wl;
    for (int g = 0; g < p.G; ++g) {
        const double tt = ((p.taugas[(row * p.G + g) * p.Wpad + w] + cia) + dust) + ray;             // the sum order of :3989
        const size_t o = (wrow * p.G + g) * p.L + l;
        p.taus[o] = tt;
        p.omegas[o] = (tt > 0.0) ? (ray + sca) / tt : 0.0;
    }
    const double c1 = 1.1911e-12, c2 = 1.439;
    const double wv = p.wave[w];
    double y, a;
    if (p.ispace == 0) { y = wv; a = c1 * (y * y * y); }
    else { y = 1.0e4 / wv; a = c1 * (y * y * y * y * y) / 1.0e4; }
    p.bnu[wrow * p.L + l] = a / (exp(c2 * y / p.lay_temp[(size_t)m * p.L + l]) - 1.0);
}

// same[m][l] = 1 when every input of layer l of model m is bit-identical to model 0's: the gas opacity row is shared (which
// covers pressure, temperature and the gas amounts) ...
__global__ void k_ms_same_init(int n_models, int L, const int32_t *__restrict__ slot, unsigned char *__restrict__ same)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_models * L) same[i] = (slot[i] == i % L) ? 1 : 0;
}
// ... and so are its columns of the (wavenumber, layer) arrays: arr [n][W][X][L] (X = 1 for the continuum opacities, ncont
// for the aerosol fractions).  One thread per (model, wavenumber, x); a differing element clears the layer's flag.
__global__ void k_ms_same_cols(int n_models, int W, int X, int L, const double *__restrict__ arr, unsigned char *__restrict__ same)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)W * X;
    if (i >= (size_t)(n_models - 1) * per) return;
    const size_t m = 1 + i / per, r = i % per;
    const double *a = arr + (m * per + r) * L, *b = arr + r * L;
    for (int l = 0; l < L; ++l)
        if (__double_as_longlong(a[l]) != __double_as_longlong(b[l])) same[m * L + l] = 0;
}

// CIRSrad's g-quadrature of the scattering branch (:4504): SPECOUT[w][path] = xfac[w] * sum_g rad[path][g][w] * DELG[g];
// the radiances before the quadrature go out as well when wanted (spec_g[w][g][path], what scloud11wave returns).
__global__ void k_ms_gquad(const double *__restrict__ rad, const double *__restrict__ delg, const double *__restrict__ xfac,
                           double *__restrict__ specout, double *__restrict__ spec_g, int W, int G, int P)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)W * P) return;
    const int ip = (int)(idx % P), w = (int)(idx / P);
    const double xf = xfac ? xfac[w] : 1.0;
    double acc = 0.0;
    for (int g = 0; g < G; ++g) {
        const double r = rad[((size_t)ip * G + g) * W + w] * xf;
        if (spec_g) spec_g[((size_t)w * G + g) * P + ip] = r;
        acc += r * delg[g];
    }
    specout[idx] = acc;
}

}  // namespace ansfm
