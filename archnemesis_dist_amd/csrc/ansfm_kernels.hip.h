// ansfm_kernels.hip.h -- gfx950 kernels of the correlated-k thermal-emission hot path.
//
// Data layout in HBM (all float64, "wave fastest" so that a wavefront = 64 consecutive
// wavenumbers reads/writes 512 contiguous bytes):
//   lnK    [NP][NT][S][G][Wpad]   ln k for k>0; k<=0 stored NaN-boxed (see encode_lnk)
//   tau    [n][L][G][Wpad]        vertical gas opacity per model/layer/g
//   cont   [n][L][Wpad]           continuum opacity (TAUCIA+TAUDUST+TAURAY), transposed on upload
// Wpad = W rounded up to 64; pad lanes carry k=0 and are never written back to the caller.
//
// Reference seams restated here (paths relative to the reference tree):
//   Spectroscopy_0.calc_k/calc_kg            Spectroscopy_0.py:2298-2437 / :2147-2295
//   ForwardModel_0.k_overlap / rank           ForwardModel_0.py:6029-6173
//   ForwardModel_0.calculate_layer_opacity    ForwardModel_0.py:3989, :4006
//   ForwardModel_0.calc_thermal_emission_spectrum / planck   ForwardModel_0.py:6287-6377 / :6183
//   ForwardModel_0.CIRSrad g-quadrature       ForwardModel_0.py:4504
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ansfm_merge_common.hip.h"

namespace ansfm {

__global__ void k_layer_prep(int n_layers_total, const double *__restrict__ lay_press_pa,
                             const double *__restrict__ lay_temp, int NP,
                             const double *__restrict__ PRESS, int NT,
                             const double *__restrict__ TEMP, double press_div, int grid_f32,
                             LayerInterp *__restrict__ out)
{
    // grid_f32: Spectroscopy_0.PRESS/TEMP are float32 arrays (tables read from .kta): NumPy then takes
    // np.log(PRESS[i]), phi-plo, thi-tlo and 1./(thi-tlo) in float32 (see include/ansfm.h, ansfm_set_f32_semantics)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_layers_total) return;
    double press1 = lay_press_pa[i] / press_div;  // LayerX.PRESS/ATM_TO_PASCAL  ForwardModel_0.py:3855
    double temp1 = lay_temp[i];
    int ip = 0;
    double best = fabs(PRESS[0] - press1);
    for (int k = 1; k < NP; ++k) {
        double d = fabs(PRESS[k] - press1);
        if (d < best) { best = d; ip = k; }
    }
    int ipl, iph;
    bool pclamp = false;
    if (PRESS[ip] >= press1) {
        iph = ip;
        if (ip == 0) { press1 = PRESS[0]; ipl = 0; iph = 1; pclamp = true; }
        else ipl = ip - 1;
    } else {
        ipl = ip;
        if (ip == NP - 1) { press1 = PRESS[NP - 1]; iph = NP - 1; ipl = NP - 2; pclamp = true; }
        else iph = ip + 1;
    }
    int it = 0;
    best = fabs(TEMP[0] - temp1);
    for (int k = 1; k < NT; ++k) {
        double d = fabs(TEMP[k] - temp1);
        if (d < best) { best = d; it = k; }
    }
    int itl, ith;
    bool tclamp = false;
    if (TEMP[it] >= temp1) {
        ith = it;
        if (it == 0) { temp1 = TEMP[0]; itl = 0; ith = 1; tclamp = true; }
        else itl = it - 1;
    } else {
        itl = it;
        if (it == NT - 1) { temp1 = TEMP[NT - 1]; ith = NT - 1; itl = NT - 2; tclamp = true; }
        else ith = it + 1;
    }
    double lpress = log(press1), plo = log(PRESS[ipl]), phi = log(PRESS[iph]);
    double tlo = TEMP[itl], thi = TEMP[ith];
    double pden = phi - plo, tden = thi - tlo, dudt = 1. / tden;
    if (grid_f32) {
        plo = (double)(float)plo;
        phi = (double)(float)phi;
        if (pclamp) lpress = (double)(float)lpress;
        pden = (double)((float)phi - (float)plo);
        tden = (double)((float)thi - (float)tlo);
        dudt = (double)(1.0f / (float)tden);
    }
    LayerInterp r;
    r.ipl = ipl; r.iph = iph; r.itl = itl; r.ith = ith;
    r.v = (lpress - plo) / pden;
    r.u = (temp1 - tlo) / tden;
    if (grid_f32 && pclamp) r.v = (double)(((float)lpress - (float)plo) / (float)pden);   // all-float32 expression
    if (grid_f32 && tclamp) r.u = (double)(((float)temp1 - (float)tlo) / (float)tden);
    r.dudt = dudt;
    out[i] = r;
}

// ------------------------------------------------------------------------------------------------
// Table upload: K[W][G][NP][NT][S] (reference layout) -> lnK[NP][NT][S][G][Wpad].
// For every g this is a transpose between w and q = (p, t, s): 64 x 64 tiles through LDS, reads coalesced along q,
// writes coalesced along w.  k_table_check flags k < 0 / NaN or k decreasing in g (flag[0] |= 1), reading the source
// coalesced as well (one thread per (w, q), g sequential).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_table_relayout(const double *__restrict__ K, double *__restrict__ lnK, int W,
                                                        int Wpad, int G, int Q)
{
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
    const int w0 = blockIdx.x * 64, q0 = blockIdx.y * 64, g = blockIdx.z;
    for (int r = ty; r < 64; r += 4) {
        const int w = w0 + r, q = q0 + tx;
        tile[r][tx] = (w < W && q < Q) ? K[((size_t)w * G + g) * Q + q] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int q = q0 + r;
        if (q < Q) lnK[((size_t)q * G + g) * Wpad + w0 + tx] = encode_lnk(tile[tx][r]);      // pad lanes: k = 0
    }
}

__global__ void k_table_check(const double *__restrict__ K, int W, int G, int Q, int *flag)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)W * Q) return;
    const int q = (int)(idx % Q);
    const size_t w = idx / Q;
    const double *src = K + w * (size_t)G * Q + q;
    bool bad = false;
    double prev = 0.0;
    for (int g = 0; g < G; ++g) {
        const double k = src[(size_t)g * Q];
        bad |= !(k >= 0.0) || (g > 0 && k < prev);
        prev = k;
    }
    if (bad) atomicOr(flag, 1);
}

// Array-level seam calc_k / calc_kg: writes the reference layout k[W][G][L][S] directly.
__global__ void k_calc_k_seam(const double *__restrict__ lnK, int W, int Wpad, int G, int NT, int S,
                              int L, const LayerInterp *__restrict__ li, double *__restrict__ k_out,
                              double *__restrict__ dk_out)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)L * S * G * W;
    if (idx >= total) return;
    int w = (int)(idx % W);
    size_t r = idx / W;
    int g = (int)(r % G); r /= G;
    int s = (int)(r % S);
    int l = (int)(r / S);
    LayerInterp q = li[l];
    size_t strideT = (size_t)S * G * Wpad;
    size_t off = ((size_t)s * G + g) * Wpad + w;
    double l1 = lnK[((size_t)q.ipl * NT + q.itl) * strideT + off];
    double l2 = lnK[((size_t)q.ipl * NT + q.ith) * strideT + off];
    double h1 = lnK[((size_t)q.iph * NT + q.itl) * strideT + off];
    double h2 = lnK[((size_t)q.iph * NT + q.ith) * strideT + off];
    double kk, dk;
    interp_kg(l1, l2, h1, h2, q.v, q.u, q.dudt, kk, dk);
    size_t o = (((size_t)w * G + g) * L + l) * S + s;
    k_out[o] = kk;
    if (dk_out) dk_out[o] = dk;
}

// ------------------------------------------------------------------------------------------------
// K1+K2 fused: (P,T) interpolation + random-overlap merge.   "ck_overlap"
//
// One LANE per (wavenumber, layer) cell; a wavefront = 64 consecutive wavenumbers of ONE layer, so
// corner indices, u, v and gas amounts are wave-uniform (SGPRs) and every table access is one
// 512-byte coalesced row.  The reference sorts the G*G sums tau_i + k_j*amount (argsort) and
// walks the sorted list once to re-bin it (rank, ForwardModel_0.py:6117-6173).  Both inputs are
// already sorted in g, so the sorted sequence is produced by a G-way streaming merge of the rows
// (a_i + b_0..b_{G-1}) -- the row heads held as a sorted list in registers, see merge_step -- and rank's walk
// consumes it on the fly: nothing of size G*G is ever stored.  The per-lane arrays a[G], b[G+1] live in LDS as
// [index][lane], so a lane-dependent index never causes a bank conflict (bank depends on the lane only).
//
// LDS per wave: (2G+1)*64*8 bytes (+ shared del_g / g_ord tables)  -> 21.1 KiB at G=20, 7 waves per CU.
// ------------------------------------------------------------------------------------------------
// LDS byte offsets of the tables that open the merge kernels' dynamic LDS block (the kernels have no static LDS, so the
// block starts at address 0 -- checked once per launch): reads become `ds_read vaddr = index << k, offset:const`.
constexpr unsigned kLdsDG = 0, kLdsGORD = kMaxG * 8, kLdsDGF = (2 * kMaxG + 2) * 8, kLdsA = kLdsDGF + kMaxG * 4;

// One popped element of the merge with everything the rank walk and the row's next key need, fetched from LDS
// as soon as the winner key is known (software pipelining: the walk of element t and the rest of the insertion
// pass run while the operands of element t+1 are in flight).
// Bin records, per block [bin][2][lane] pairs of doubles: pair 0 = (kacc, sum1), pair 1 = (gd, code of the element that
// closed the bin: row | column << 5), each pair one 16-byte store per lane, the lanes of a pair contiguous (1 KiB rows).
// The stores sit in the merge loop's crossing branch, which runs in about every second step, and every store
// traffic there is not free (doubling five 8-byte row stores: +29 % on the forward kernel) -- so the closing element's
// value and weight are not stored, the resolve loop recomputes them from LDS (same operations), and what is stored goes
// out as two wide stores.  Same-box comparisons: gradient kernel 25.7 -> 23.7 ms against six 8-byte rows; forward kernel
// within +-1 % of five 8-byte rows and of three pairs (it is not store-bound at this level).  The gradient kernel's resolve rewrites the
// pairs as (frac, 1/weight-sum) and (weight, code) for its replay passes.
typedef double dbl2 __attribute__((ext_vector_type(2)));
constexpr unsigned kRecRow = 64u * 16u, kRecBin = 2u * kRecRow;

struct MergeElem {
    double ai, bc, bn, w;
    int ci, np;
};

// Weight of element (i, j) = del_g[i] * del_g[j].  DELG float32 (W32): NumPy forms the float32 product, which is one
// v_mul_f32 of the float32 copies kept behind the double tables (DG, GORD) in LDS.
__device__ __forceinline__ const float *delg_f32_table(const double *DG) { return reinterpret_cast<const float *>(DG + 2 * kMaxG + 2); }
template <bool W32>
__device__ __forceinline__ double pair_weight(const double *DG, int i, int j)
{
    if constexpr (W32) {
        const float *DGF = delg_f32_table(DG);
        return (double)(DGF[i] * DGF[j]);
    } else
        return DG[i] * DG[j];
}

// SORTED = false (generic path): the rows / columns were sorted per lane beforehand; PA / PB give the original
// g-ordinate of each sorted position, which is the one whose weight applies.
template <bool W32, bool SORTED = true>
__device__ __forceinline__ void merge_fetch(double key, int lane, const double *A, const double *B,
                                            const double *DG, MergeElem &e,
                                            const unsigned char *PA = nullptr, const unsigned char *PB = nullptr)
{
    const unsigned kb = (unsigned)__double_as_longlong(key);
    const int ci = kb & 31, cp = (kb >> 5) & 63;
    e.ci = ci;
    e.np = cp + 1;
    e.ai = A[ci * kWave + lane];
    const unsigned ab = lds_addr(B + lane) + ((unsigned)cp << 9);
    e.bc = lds_ld(ab);
    e.bn = lds_ld(ab + 512);                    // B[G] = sentinel column: an exhausted row re-enters as "huge"
    if constexpr (SORTED) e.w = pair_weight<W32>(DG, ci, cp);
    else e.w = pair_weight<W32>(DG, PA[ci * kWave + lane], PB[cp * kWave + lane]);
}

// List keys: the element value a_i + b_j with the low 11 mantissa bits replaced by (col << 5) | row  (col <= 32,
// row <= 31).  Keys compare like the values except among values closer than 2^-41 relative (treated as ties, which
// rank() orders arbitrarily anyway); the exact value is recomputed from a_i + b_j when the element is consumed, so the
// sums are the reference's.
__device__ __forceinline__ double pack_key11(double v, int row, int col)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    b = (b & ~0x7FFULL) | (unsigned long long)((col << 5) | row);
    return __longlong_as_double((long long)b);
}

// rank() walk state of one lane (ForwardModel_0.py:6155-6170).  Bin boundaries are recorded and resolved
// after the loop: frac needs a division, and the next bin's (1-frac) share is added there too -- the same
// sums in a different association.
struct WalkState {
    double gd, kacc, sum1, gnext;
    unsigned roff;      // byte offset of this lane's first pair in the record of the bin being filled: ig * kRecBin + lane * 16
    unsigned gaddr;     // LDS byte address of GORD[ig + 1]
};
__device__ __forceinline__ WalkState walk_begin(const double *GORD, int lane)
{
    WalkState ws;
    ws.gd = 0.0; ws.kacc = 0.0; ws.sum1 = 0.0;
    ws.gaddr = lds_addr(GORD + 1);
    ws.gnext = lds_ld(ws.gaddr);
    ws.roff = (unsigned)lane * 16u;
    return ws;
}
// number of bins closed so far
__device__ __forceinline__ int walk_bins(const WalkState &ws, const double *GORD) { return (int)((ws.gaddr - lds_addr(GORD + 1)) >> 3); }

template <bool REC_CODE>
__device__ __forceinline__ bool merge_walk(const MergeElem &e, WalkState &ws, double *rec, const double *GORD,
                                           int lane)
{
    const double cv = e.ai + e.bc;
    const double w = e.w;
    const double gdn = ws.gd + w;
    double kn = ws.kacc + cv * w, sn = ws.sum1 + w;
    // ordered >= : GORD[G+1] is NaN, so nothing crosses after the last bin whatever gdn is (garbage weights of a call
    // that is going to be rerun on the generic path, NaN / inf input) -- the record index stays <= G
    const bool cross = (gdn >= ws.gnext);
    if (cross) {                                // this element straddles the bin boundary
        // The branch runs in about every second step (some lane of the 64 crosses), so it is kept to four stores and two
        // adds: the record slot is a running 32-bit byte offset onto the wave-uniform base (no 64-bit index arithmetic).
        gst<dbl2>(rec, ws.roff, dbl2{ws.kacc, ws.sum1});
        gst<dbl2>(rec, ws.roff + kRecRow, dbl2{ws.gd, __longlong_as_double((long long)(e.ci | ((e.np - 1) << 5)))});
        kn = 0.0; sn = 0.0;
        ws.roff += kRecBin;
        ws.gaddr += 8u;
        ws.gnext = lds_ld(ws.gaddr);            // GORD[G+1] = NaN: nothing crosses after the last bin
    }
    ws.kacc = kn; ws.sum1 = sn;
    ws.gd = gdn;
    return cross;
}

// Division-free form of the walk (forward kernel, NODIV).  rank()'s boundary element contributes frac * cont * w to the bin
// it closes and (1 - frac) * cont * w to the next one, frac = (g_ord[ig+1] - gdist_prev) / w: that is
// (g_ord[ig+1] - gdist_prev) * cont and (gdist - g_ord[ig+1]) * cont -- no division -- and the bin's weight sum (carry
// (1-frac) w of the previous boundary element, the weights inside, frac w) telescopes to g_ord[ig+1] - g_ord[ig].  A closed
// bin is then ONE 8-byte store of its un-normalised sum (rec = [bin][lane] doubles) instead of a 32-byte record, and the
// resolve pass is a division by the bin width when the merged spectrum is read back.  Differs from the recorded form
// in the last bits only (the reference itself forms frac from a difference of cumulative sums, good to ~1e-12).
// Precondition (checked at launch): the first element of the merged order does not close a bin -- rank()'s python
// gdist[-1] wrap, which only the recorded form reproduces.
__device__ __forceinline__ bool merge_walk_nodiv(const MergeElem &e, WalkState &ws, double *rec)
{
    const double cv = e.ai + e.bc;
    const double w = e.w;
    const double gdn = ws.gd + w;
    double kn = fma(cv, w, ws.kacc);
    const bool cross = (gdn >= ws.gnext);       // ordered: GORD[G+1] is NaN
    if (cross) {
        gst<double>(rec, ws.roff, fma(ws.gnext - ws.gd, cv, ws.kacc));
        kn = (gdn - ws.gnext) * cv;
        ws.roff += kWave * 8u;
        ws.gaddr += 8u;
        ws.gnext = lds_ld(ws.gaddr);
    }
    ws.kacc = kn;
    ws.gd = gdn;
    return cross;
}

// The heads of the G rows are kept as a SORTED LIST IN REGISTERS (R[0] = the current winner): popping is free and the
// row's next key is inserted by one pass of v_max_f64 + v_min_f64 pairs over statically indexed
// registers -- no tree in LDS, no lane-dependent addressing, and the next winner is known after the FIRST
// compare-exchange, so its operands' LDS reads are hidden behind the rest of the pass and the walk.  NR = list length
// (compile time, >= G; unused entries hold "huge" keys).
// Returns the consumed element's (row, column) and whether it closed a bin, as 16 bits: the gradient kernel
// records them and replays the sorted order for the gradient rows.
template <int NR, bool W32, bool REC_CODE = false, bool SORTED = true, bool NODIV = false>
__device__ __forceinline__ unsigned merge_step(double (&R)[NR], MergeElem &e, MergeElem &en, WalkState &ws,
                                               int lane, const double *A, const double *B,
                                               const double *DG, const double *GORD, double *rec,
                                               const unsigned char *PA = nullptr, const unsigned char *PB = nullptr)
{
    // 1. the popped row's next element x enters the list s_1 <= s_2 <= ... (s_0 was popped):
    //        t_0 = min(x, s_1),   t_k = min(max(x, s_k), s_{k+1}),   t_{NR-1} = max(x, s_{NR-1})
    //    -- every output independent of the others (no carry chain), in place in ascending k.
    const double x = pack_key11(e.ai + e.bn, e.ci, e.np);
    asm("v_min_f64 %0, %1, %2" : "=v"(R[0]) : "v"(x), "v"(R[1]));
    // 2. fetch the operands of the new winner (LDS reads in flight during the rest of the pass and the walk)
    merge_fetch<W32, SORTED>(R[0], lane, A, B, DG, en, PA, PB);
    // 3. finish the insertion: every max first, then every min -- no result is consumed by a neighbouring instruction
    //    (6.40 -> 6.32 ms against blocks of 6, same box)
    constexpr int kBlk = NR;
#pragma unroll
    for (int k0 = 1; k0 < NR - 1; k0 += kBlk) {
        double mk[kBlk];
#pragma unroll
        for (int j = 0; j < kBlk; ++j)
            if (k0 + j < NR - 1) asm("v_max_f64 %0, %1, %2" : "=v"(mk[j]) : "v"(x), "v"(R[k0 + j]));
#pragma unroll
        for (int j = 0; j < kBlk; ++j)
            if (k0 + j < NR - 1) asm("v_min_f64 %0, %1, %2" : "=v"(R[k0 + j]) : "v"(mk[j]), "v"(R[k0 + j + 1]));
    }
    asm("v_max_f64 %0, %1, %2" : "=v"(R[NR - 1]) : "v"(x), "v"(R[NR - 1]));
    // 4. rank walk on the element just consumed
    bool cross;
    if constexpr (NODIV) cross = merge_walk_nodiv(e, ws, rec);
    else cross = merge_walk<REC_CODE>(e, ws, rec, GORD, lane);
    // step code of the gradient replay, 12 bits: row (0-4), column (5-9), "the element closed a bin" (10); kCodesPerWord of
    // them to a 64-bit word of the stream
    return (unsigned)(e.ci | ((e.np - 1) << 5) | (cross ? 0x400 : 0));
}

// R[i] = head of row i = a_i + b_0: ascending in i when a is.  A loaded gas is (fast path: by precondition; generic: sorted
// first); a MERGED spectrum is non-decreasing only up to the rounding of its bin averages, and two neighbours that
// rounding has swapped can fall on either side of a key boundary (seen with k(g) flat to 1e-9: an unsorted list loses an
// entry in the insertion network and a sentinel is consumed).  So the keys are checked, and when some lane's are not
// ascending the heads are put in order one by one (the merge itself only needs every ROW ascending, i.e. b sorted).
template <int NR>
__device__ __forceinline__ void merge_init(double (&R)[NR], int G, int lane, const double *A, double b0, double huge)
{
#pragma unroll
    for (int i = 0; i < NR; ++i) R[i] = (i < G) ? pack_key11(A[(i < G ? i : 0) * kWave + lane] + b0, i, 0) : huge;
    bool bad = false;
#pragma unroll
    for (int i = 0; i + 1 < NR; ++i) bad |= (R[i + 1] < R[i]);
    if (__builtin_amdgcn_ballot_w64(bad) != 0) {
        double T[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) { T[i] = R[i]; R[i] = huge; }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (i < G) {
                const double x = T[i];
#pragma unroll
                for (int k = NR - 1; k >= 1; --k) R[k] = fmin(fmax(x, R[k - 1]), R[k]);
                R[0] = fmin(x, R[0]);
            }
        }
    }
}

// Per-lane insertion sort of one LDS column (values ascending, stable) carrying the original index of every
// position in P.  Only the generic path (k not sorted in g) uses it.
__device__ __forceinline__ void sort_column(double *X, unsigned char *P, int G, int lane)
{
    for (int g = 0; g < G; ++g) P[g * kWave + lane] = (unsigned char)g;
    for (int i = 1; i < G; ++i) {
        const double key = X[i * kWave + lane];
        const unsigned char pk = P[i * kWave + lane];
        int j = i - 1;
        while (j >= 0 && X[j * kWave + lane] > key) {
            X[(j + 1) * kWave + lane] = X[j * kWave + lane];
            P[(j + 1) * kWave + lane] = P[j * kWave + lane];
            --j;
        }
        X[(j + 1) * kWave + lane] = key;
        P[(j + 1) * kWave + lane] = pk;
    }
}

// SORTED = false: generic path for k-distributions that are not non-decreasing in g (the reference sorts the G*G
// products itself, :6150).  Each gas's (k, weight) pairs are sorted per lane first -- the multiset of
// (product, weight) is unchanged, so rank()'s walk sees the same sequence up to the order of exact ties -- and the
// skip rules keep looking at the LAST g-ordinate in the original order (:6075-6102).  A spectrum that passes through
// unmerged comes out in its original order.
template <int NR, bool FROM_K, bool W32, bool SORTED = true, bool NODIV = false>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_ck_overlap(OverlapParams p)
{
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int G = p.G;
    // tables first: their LDS addresses are compile-time constants (dynamic LDS starts at 0), so a table read is
    // `ds_read vaddr = index << k, offset:const` with no base add
    double *DG = smem;                           // [kMaxG] doubles, then GORD [kMaxG + 2], then the float32 copy of DG
    double *GORD = DG + kMaxG;
    double *A = reinterpret_cast<double *>(reinterpret_cast<char *>(GORD + kMaxG + 2) + kMaxG * sizeof(float));
    double *B = A + G * kWave;                   // G+1 rows
    unsigned char *PA = reinterpret_cast<unsigned char *>(B + (G + 1) * kWave);   // SORTED = false only
    unsigned char *PB = PA + G * kWave;
    if (lane < G) {
        DG[lane] = p.del_g[lane];
        const_cast<float *>(delg_f32_table(DG))[lane] = (float)p.del_g[lane];
    }
    if (lane < G + 2) GORD[lane] = p.g_ord[lane];
    const double HUGE_KEY = __longlong_as_double(0x7FE0000000000000LL);   // finite, above any optical depth
    B[G * kWave + lane] = HUGE_KEY;
    __syncthreads();
    double wsum = 0.0;
    for (int g = 0; g < G; ++g) wsum += DG[g];
    const double wtot = wsum * wsum;  // stands in for gdist[-1] (python wrap at iloop==0)

    // per-block scratch: closed-bin records, see kRecBin
    double *rec = p.scratch + (size_t)blockIdx.x * 6 * G * kWave;
    TileQueue tq;
    tq.init();
    for (;;) {
        int vt = 0, m = 0, l = 0;
        if (!tq.next(p, lane, vt, m, l)) break;
        const int nu = vt * kWave + lane;
        LayerInterp q;
        if constexpr (!FROM_K) q = p.li[(size_t)m * p.L + l];
        bool unsorted = false;

        load_gas<FROM_K>(p, q, m, l, 0, nu, A, lane, unsorted);
        double alast = A[(G - 1) * kWave + lane];       // last g-ordinate in the ORIGINAL order
        if constexpr (!SORTED) sort_column(A, PA, G, lane);
        for (int s = 1; s < p.S; ++s) {
            load_gas<FROM_K>(p, q, m, l, s, nu, B, lane, unsorted);
            if constexpr (SORTED)                       // the call is rerun on the generic path: no point in merging
                if (__builtin_amdgcn_ballot_w64(unsorted) != 0) break;
            const double blast = B[(G - 1) * kWave + lane];
            if constexpr (!SORTED) sort_column(B, PB, G, lane);
            if constexpr (SORTED) alast = A[(G - 1) * kWave + lane];
            // skip rules, cutoff = 0  (ForwardModel_0.py:6073-6102)
            bool takeB, keepA;
            if (s == 1) { takeB = (alast <= 0.0); keepA = !takeB && (blast <= 0.0); }
            else { keepA = (blast <= 0.0); takeB = !keepA && (alast <= 0.0); }
            const bool do_merge = !(takeB | keepA);
            if (takeB) {
                for (int g = 0; g < G; ++g) A[g * kWave + lane] = B[g * kWave + lane];
                if constexpr (!SORTED) {
                    for (int g = 0; g < G; ++g) PA[g * kWave + lane] = PB[g * kWave + lane];
                    alast = blast;
                }
            }
            if (do_merge) {
                // ---- sorted list of the G row heads (row i = a_i + b_j, j ascending) -------------
                double R[NR];
                merge_init<NR>(R, G, lane, A, B[lane], HUGE_KEY);
                MergeElem e0, e1;
                merge_fetch<W32, SORTED>(R[0], lane, A, B, DG, e0, PA, PB);
                WalkState ws = walk_begin(GORD, lane);
                if constexpr (NODIV) ws.roff = (unsigned)lane * 8u;
                const int nloop = G * G;
                int it = 0;
                for (; it + 1 < nloop; it += 2) {   // ping-pong: no register rotation
                    merge_step<NR, W32, false, SORTED, NODIV>(R, e0, e1, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    merge_step<NR, W32, false, SORTED, NODIV>(R, e1, e0, ws, lane, A, B, DG, GORD, rec, PA, PB);
                }
                if (it < nloop) merge_step<NR, W32, false, SORTED, NODIV>(R, e0, e1, ws, lane, A, B, DG, GORD, rec, PA, PB);
                if constexpr (NODIV) {
                    // ---- normalise: closed bins by their width; the open one as rank()'s trailing `if ig == ng-1` (:6171) ----
                    const int ig = walk_bins(ws, GORD);
                    for (int g0 = 0; g0 < G; g0 += kLoadBatch) {
                        double r[kLoadBatch];
#pragma unroll
                        for (int k = 0; k < kLoadBatch; ++k)
                            r[k] = gld<double>(rec, (unsigned)((g0 + k < G) ? g0 + k : G - 1) * (kWave * 8u) + (unsigned)lane * 8u);
#pragma unroll
                        for (int k = 0; k < kLoadBatch; ++k) {
                            const int b = g0 + k;
                            if (b < G) {
                                double outv = 0.0;
                                if (b < ig) outv = fast_div(r[k], GORD[b + 1] - GORD[b]);
                                else if (b == ig) outv = (b == G - 1) ? fast_div(ws.kacc, ws.gd - GORD[G - 1]) : ws.kacc;
                                A[b * kWave + lane] = outv;
                            }
                        }
                    }
                } else {
                // ---- resolve the bins --------------------------------------------------------------------
                // The closing element of every bin is re-formed from LDS (a[row] + b[col], its weight) exactly as the
                // walk formed it, so `a` must stay intact until the last bin is done: the outputs go to row 0 of the bin
                // records (full-wave coalesced stores) and come back into A afterwards.
                double ck = 0.0, cs = 0.0;   // (1-frac) share carried into the next bin
                const int ig = walk_bins(ws, GORD);
                constexpr int kRB = 5;       // records of kRB bins are fetched together (one round trip)
                for (int b0 = 0; b0 < G; b0 += kRB) {
                    double rka[kRB], rs1[kRB], rgd[kRB];
                    unsigned rcd[kRB];
#pragma unroll
                    for (int k = 0; k < kRB; ++k) {
                        const int bi = (b0 + k < G) ? b0 + k : G - 1;
                        const unsigned ro = (unsigned)bi * kRecBin + (unsigned)lane * 16u;
                        const dbl2 v0 = gld<dbl2>(rec, ro), v1 = gld<dbl2>(rec, ro + kRecRow);
                        rka[k] = v0.x; rs1[k] = v0.y; rgd[k] = v1.x; rcd[k] = (unsigned)__double_as_longlong(v1.y);
                    }
#pragma unroll
                    for (int k = 0; k < kRB; ++k) {
                        const int b = b0 + k;
                        if (b < G) {
                            double outv = 0.0;
                            if (b < ig) {
                                const int crow = rcd[k] & 31, ccol = (rcd[k] >> 5) & 63;
                                const double cv = A[crow * kWave + lane] + B[ccol * kWave + lane];
                                double w;
                                if constexpr (SORTED) w = pair_weight<W32>(DG, crow, ccol);
                                else w = pair_weight<W32>(DG, PA[crow * kWave + lane], PB[ccol * kWave + lane]);
                                const double ka = rka[k], s1 = rs1[k], cw = cv * w;
                                // a crossing at the very first element (nothing accumulated yet) sees python's gdist[-1]
                                const double gd0 = rgd[k];
                                const double gprev = (b == 0 && s1 == 0.0) ? wtot : gd0;
                                const double gdn = gd0 + w;                 // the same add the walk made
                                const double frac = fast_div(GORD[b + 1] - gprev, gdn - gprev);     // <= 1 ulp, as every division of the resolve
                                const double kb = (ck + ka) + frac * cw;
                                const double sb = (cs + s1) + frac * w;
                                outv = fast_div(kb, sb);
                                ck = (1.0 - frac) * cw;
                                cs = (1.0 - frac) * w;
                            } else if (b == ig) {
                                // trailing `if ig == ng-1` (:6171); an unfinished earlier bin stays un-normalised
                                const double kb = ck + ws.kacc, sb = cs + ws.sum1;
                                outv = (b == G - 1) ? fast_div(kb, sb) : kb;
                            }
                            gst<double>(rec, (unsigned)b * kRecBin + (unsigned)lane * 16u, outv);
                        }
                    }
                }
                for (int g0 = 0; g0 < G; g0 += kLoadBatch) {          // merged spectrum: records' row 0 -> A
                    double r[kLoadBatch];
#pragma unroll
                    for (int k = 0; k < kLoadBatch; ++k)
                        r[k] = gld<double>(rec, (unsigned)((g0 + k < G) ? g0 + k : G - 1) * kRecBin + (unsigned)lane * 16u);
#pragma unroll
                    for (int k = 0; k < kLoadBatch; ++k)
                        if (g0 + k < G) A[(g0 + k) * kWave + lane] = r[k];
                }
                }
                if constexpr (!SORTED) {   // the merged spectrum is ascending with the plain del_g weights
                    for (int g = 0; g < G; ++g) PA[g * kWave + lane] = (unsigned char)g;
                    alast = A[(G - 1) * kWave + lane];
                }
            }
        }
        double *out = p.tau + (((size_t)m * p.L + l) * G) * p.Wpad + nu;
        if constexpr (SORTED) {
            if (unsorted) atomicOr(p.err_flag, 1);
            for (int g = 0; g < G; ++g) out[(size_t)g * p.Wpad] = A[g * kWave + lane];
        } else {
            for (int g = 0; g < G; ++g) out[(size_t)PA[g * kWave + lane] * p.Wpad] = A[g * kWave + lane];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1g+K2g fused: calc_kg + k_overlapg/rankg (ForwardModel_0.py:5842-6026).   "ck_overlapg"
//
// rankg accumulates, per output bin, weight * gradient-row of every element in sorted order
// (:6002-6025).  The gradient row of element (i, j) is (:5918-5921, :5946-5949)
//      slot pp <= igas : D_old[pp][i]          (previous stage's dk_g_param, by ROW)
//      slot igas+1     : k_new[j]              (by COLUMN)
//      slot igas+2     : D_old[igas+1][i] + dkdT_new[j]*amount
// so every slot is a weighted gather of a G-vector along the sorted order.  The merge runs ONCE (the
// forward kernel's loop) and records the order as 16 bits per step; the slots are then produced by
// replaying that order with up to three G-vectors staged in the LDS the merge no longer needs
// (gathers out of LDS [index][lane] are conflict-free; out of global memory they touch ~40 cache
// lines per wave-load).  Bin boundaries are deferred exactly like the forward walk: the replay stores the
// raw partial sums, a uniform 20-iteration pass applies frac / (1-frac) carry / normalisation.
// Slot bookkeeping of the skip branches (:5897-5937) is reproduced, including the stale slots they
// leave behind.
// ------------------------------------------------------------------------------------------------
struct OverlapGParams {
    OverlapParams o;          // o.scratch: [grid][G+1][6][64] bin records
    const double *dkin;       // FROM_K: dkdT[S][L][G][Wpad]
    double *dk;               // out [n][L][S+1][G][Wpad]
    double *gscratch;         // [grid][2 + 2*(S+1) + 1][G][64]   KRB, DTB, Dbuf0, Dbuf1, Asave
    unsigned long long *perm; // [grid][ceil(G*G/4)][64]: four 16-bit step codes per word
    unsigned gas_mask;        // bit s: the slot of gas s (d tau / d amount_s) is wanted.  A state vector names one or two gases:
                              // every other gas's slot would cost a replay pass per later merge for nothing (its rows of dk are
                              // written as zeros).  Bit 31: the temperature slot (two passes per merge).
};

// from_k (array-level k_overlapg seam) is a run-time flag here: the load phase is a few per cent of the kernel and one
// template parameter less halves the number of instantiations of the largest kernel of the library.
__device__ __forceinline__ void load_gas_g(const OverlapGParams &pg, const LayerInterp &q, int m, int l, int s,
                                           int nu, double *DST, double *KR, double *DT, int lane, bool &unsorted)
{
    const bool FROM_K = pg.o.kin != nullptr;
    const OverlapParams &p = pg.o;
    const int G = p.G;
    const double amt = p.amount[((size_t)m * p.S + s) * p.L + l];
    double prev = -__builtin_inf();
    if (FROM_K) {
        const size_t base = (((size_t)s * p.L + l) * G) * p.Wpad + nu;
        for (int g0 = 0; g0 < G; g0 += kLoadBatch) {
            double r1[kLoadBatch], r2[kLoadBatch];
#pragma unroll
            for (int k = 0; k < kLoadBatch; ++k) {
                const int gi = (g0 + k < G) ? g0 + k : G - 1;
                r1[k] = p.kin[base + (size_t)gi * p.Wpad];
                r2[k] = pg.dkin[base + (size_t)gi * p.Wpad];
            }
#pragma unroll
            for (int k = 0; k < kLoadBatch; ++k)
                if (g0 + k < G) {
                    const int g = g0 + k;
                    const double kk = r1[k] * amt;
                    DST[g * kWave + lane] = kk;
                    KR[g * kWave + lane] = r1[k];
                    DT[g * kWave + lane] = r2[k] * amt;
                    unsorted |= (kk < prev);
                    prev = kk;
                }
        }
    } else {
        const size_t strideT = (size_t)p.S * G * p.Wpad;
        const size_t off = (size_t)s * G * p.Wpad + nu;
        const double *c1 = p.lnK + ((size_t)q.ipl * p.NT + q.itl) * strideT + off;
        const double *c2 = p.lnK + ((size_t)q.ipl * p.NT + q.ith) * strideT + off;
        const double *c3 = p.lnK + ((size_t)q.iph * p.NT + q.itl) * strideT + off;
        const double *c4 = p.lnK + ((size_t)q.iph * p.NT + q.ith) * strideT + off;
        for (int g0 = 0; g0 < G; g0 += kLoadBatch) {
            double r1[kLoadBatch], r2[kLoadBatch], r3[kLoadBatch], r4[kLoadBatch];
#pragma unroll
            for (int k = 0; k < kLoadBatch; ++k) {
                const int gi = (g0 + k < G) ? g0 + k : G - 1;
                const size_t go = (size_t)gi * p.Wpad;
                r1[k] = __builtin_nontemporal_load(c1 + go);
                r2[k] = __builtin_nontemporal_load(c2 + go);
                r3[k] = __builtin_nontemporal_load(c3 + go);
                r4[k] = __builtin_nontemporal_load(c4 + go);
            }
#pragma unroll
            for (int k = 0; k < kLoadBatch; ++k)
                if (g0 + k < G) {
                    const int g = g0 + k;
                    double kraw, dkr;
                    interp_kg(r1[k], r2[k], r3[k], r4[k], q.v, q.u, q.dudt, kraw, dkr);
                    const double kk = kraw * amt;
                    DST[g * kWave + lane] = kk;
                    KR[g * kWave + lane] = kraw;
                    DT[g * kWave + lane] = dkr * amt;
                    unsorted |= (kk < prev);
                    prev = kk;
                }
        }
    }
}

// global [G][64] -> LDS [G][64], loads batched: one memory round trip per kStageBatch rows (every replay pass starts
// with one of these and the wave has at most one sibling to hide it behind)
constexpr int kStageBatch = 20;
__device__ __forceinline__ void stage_slice(double *dst_lds, const double *__restrict__ src, int G, int lane)
{
    for (int g0 = 0; g0 < G; g0 += kStageBatch) {
        double r[kStageBatch];
#pragma unroll
        for (int k = 0; k < kStageBatch; ++k)
            r[k] = gld<double>(src, (unsigned)(((g0 + k < G) ? g0 + k : G - 1) * kWave + lane) * 8u);
#pragma unroll
        for (int k = 0; k < kStageBatch; ++k)
            if (g0 + k < G) dst_lds[(g0 + k) * kWave + lane] = r[k];
    }
}

// generic path: LDS position g holds the value of the ORIGINAL g-ordinate P[g] (the column was sorted per lane)
__device__ __forceinline__ void stage_slice_perm(double *dst_lds, const double *__restrict__ src, const unsigned char *P,
                                                 int G, int lane)
{
    for (int g = 0; g < G; ++g) dst_lds[g * kWave + lane] = src[(size_t)P[g * kWave + lane] * kWave + lane];
}

constexpr int kCodesPerWord = 5;       // 12-bit step codes in a 64-bit word of the replay stream (60 bits used)
// Replay of the recorded order for one gathered vector: SL[row] (slots of the earlier gases, row part of the
// temperature slot) or, COL, SL[col] (the new gas's slot, column part of the temperature slot).
// Store-free and branch-free: the running sum is written every step to the LDS row of
// the lane's current bin, so each row ends up holding the sum before the element that closed the bin; global
// stores inside this loop would sit in front of the code-word loads in the (in-order) vmcnt queue.
// OUTL has G+1 rows (row G collects what follows the last bin).  Returns the sum after the last boundary.
template <bool COL, bool W32, bool SORTED = true>
__device__ __forceinline__ double grad_replay(int nloop, int lane, const unsigned long long *__restrict__ perm,
                                              const double *SL, double *OUTL, const double *DG,
                                              const unsigned char *PA = nullptr, const unsigned char *PB = nullptr)
{
    double acc = 0.0;
    unsigned bo = lds_addr(OUTL + lane);            // LDS byte address of the lane's slot in the row of its current bin
    const unsigned lane8 = (unsigned)lane * 8u;     // SL is the A region (offset kLdsA)
    // the steps of one code word: all LDS operands first (one LDS round trip per word), then the dependent part.
    // Field k of the word = bits [12k, 12k + 12): row, column, closed-a-bin.  The row / column are taken out already
    // multiplied by 4 (byte offsets into the float32 weight table; << 7 more = the row of an [index][lane] array).
    auto group = [&](unsigned long long word, int nst) {
        double g[kCodesPerWord], wr[kCodesPerWord];
        const unsigned wlo = (unsigned)word, whi = (unsigned)(word >> 32);
        const unsigned wmid = __builtin_amdgcn_alignbit(whi, wlo, 24);      // bits 24..55: the field across the two halves
        unsigned crs[kCodesPerWord];
#pragma unroll
        for (int k = 0; k < kCodesPerWord; ++k) {
            const unsigned src = (k < 2) ? wlo : (k == 2 ? wmid : whi);
            constexpr int offs[kCodesPerWord] = {0, 12, 0, 4, 16};
            const int off = offs[k];
            crs[k] = src & (0x400u << off);
            const unsigned r4 = (off >= 2 ? (src >> (off - 2)) : (src << (2 - off))) & 0x7Cu;
            const unsigned c4 = (src >> (off + 3)) & 0x7Cu;
            if constexpr (SORTED) {
                if constexpr (W32) wr[k] = (double)(lds_ldf(kLdsDGF + r4) * lds_ldf(kLdsDGF + c4));
                else wr[k] = lds_ld(kLdsDG + 2 * r4) * lds_ld(kLdsDG + 2 * c4);
            } else
                wr[k] = pair_weight<W32>(DG, PA[(r4 >> 2) * kWave + lane], PB[(c4 >> 2) * kWave + lane]);
            g[k] = lds_ld(kLdsA + (((COL ? c4 : r4) << 7) + lane8));
        }
#pragma unroll
        for (int k = 0; k < kCodesPerWord; ++k) {
            if (k < nst) {
                const bool cross = crs[k] != 0;
                lds_st(bo, acc);
                const double an = acc + g[k] * wr[k];
                acc = cross ? 0.0 : an;
                bo += cross ? kWave * 8u : 0u;
            }
        }
    };
    const int nfull = nloop / kCodesPerWord, ngrp = (nloop + kCodesPerWord - 1) / kCodesPerWord;
    // Code words are fetched kPF words (4*kPF steps) ahead into kPF statically named registers: no register
    // rotation (a move of the newest word would wait for its load) and no predicated loads (clamped index).
    constexpr int kPF = 4;
    unsigned long long q[kPF];
    const unsigned lane8p = (unsigned)lane * 8u;
#pragma unroll
    for (int k = 0; k < kPF; ++k) q[k] = gld<unsigned long long>(perm, (unsigned)(k < ngrp ? k : ngrp - 1) * (kWave * 8u) + lane8p);
    int gidx = 0;
    for (; gidx + kPF <= nfull; gidx += kPF) {
#pragma unroll
        for (int j = 0; j < kPF; ++j) {
            const unsigned long long word = q[j];
            const int nxt = gidx + j + kPF;
            q[j] = gld<unsigned long long>(perm, (unsigned)(nxt < ngrp ? nxt : ngrp - 1) * (kWave * 8u) + lane8p);
            group(word, kCodesPerWord);
        }
    }
    // remaining full words and the partial last one (their loads are already in flight / clamped duplicates)
#pragma unroll
    for (int j = 0; j < kPF; ++j) {
        if (gidx + j < ngrp) {
            const int left = nloop - kCodesPerWord * (gidx + j);
            group(q[j], left < kCodesPerWord ? left : kCodesPerWord);
        }
    }
    return acc;
}

// deferred bin boundaries of one replayed vector.  The record of bin b is ONE 16-byte pair per lane, (frac, 1 / weight-sum), with
// the closing element's (row, column) in the 11 lowest mantissa bits of frac; its weight is re-formed from the tables.  (Four
// values in two pairs until the end of round 2: the records are read by every replay pass -- 49 per cell -- and with the step
// codes and the slot vectors they cycle through L2 at 5 TB/s, profiles/r02_grad_traffic.json: the kernel is bound by that
// stream.)  ACCUM: the column part of the temperature slot is added to the row part already in OUT.
template <bool COL, bool ACCUM, bool W32, bool SORTED>
__device__ __forceinline__ void grad_resolve(int G, int lane, int ig, const double *__restrict__ rec,
                                             const double *SL, const double *OUTL, double tail,
                                             double *__restrict__ OUT, const double *DG, const unsigned char *PA,
                                             const unsigned char *PB)
{
    double carry = 0.0;
    constexpr int kRB = 10;     // two memory round trips per pass for G = 20
    for (int b0 = 0; b0 < G; b0 += kRB) {
        double rfr[kRB], rri[kRB], rold[kRB];
#pragma unroll
        for (int k = 0; k < kRB; ++k) {
            const int bi = (b0 + k < G) ? b0 + k : G - 1;
            const dbl2 v0 = gld<dbl2>(rec, (unsigned)bi * kRecBin + (unsigned)lane * 16u);
            rfr[k] = v0.x; rri[k] = v0.y;
            if constexpr (ACCUM) rold[k] = gld<double>(OUT, (unsigned)(bi * kWave + lane) * 8u);
        }
#pragma unroll
        for (int k = 0; k < kRB; ++k) {
            const int b = b0 + k;
            if (b < G) {
                double v = 0.0;
                if (b < ig) {
                    const long long fb = __double_as_longlong(rfr[k]);
                    const unsigned code = (unsigned)fb & 0x7FFu;
                    const int crow = code & 31, ccol = (code >> 5) & 63;
                    rfr[k] = __longlong_as_double(fb & ~0x7FFLL);
                    double wk;
                    if constexpr (SORTED) wk = pair_weight<W32>(DG, crow, ccol);
                    else wk = pair_weight<W32>(DG, PA[crow * kWave + lane], PB[ccol * kWave + lane]);
                    const double g = SL[(COL ? ccol : crow) * kWave + lane];
                    const double gw = g * wk;
                    v = ((carry + OUTL[b * kWave + lane]) + rfr[k] * gw) * rri[k];
                    carry = (1.0 - rfr[k]) * gw;
                } else if (b == ig)
                    v = (carry + tail) * rri[k];
                if constexpr (ACCUM) v += rold[k];
                gst<double>(OUT, (unsigned)(b * kWave + lane) * 8u, v);
            }
        }
    }
}

// SORTED = false: generic path (k not non-decreasing in g), as in k_ck_overlap: A and B are sorted per lane, PA / PB give
// the original g-ordinate of each sorted position.  The gradient rows in the global scratch stay in ORIGINAL order while
// a spectrum is unmerged (rows and columns are staged through PA / PB for the replay) and are in bin order -- the
// identity -- after a merge.
template <int NR, bool W32, bool SORTED = true>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_ck_overlapg(OverlapGParams pg)
{
    const OverlapParams &p = pg.o;
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int G = p.G;
    const int NP1 = p.S + 1;
    // tables first: their LDS addresses are compile-time constants (dynamic LDS starts at 0), so a table read is
    // `ds_read vaddr = index << k, offset:const` with no base add
    double *DG = smem;                           // [kMaxG] doubles, then GORD [kMaxG + 2], then the float32 copy of DG
    double *GORD = DG + kMaxG;
    double *A = reinterpret_cast<double *>(reinterpret_cast<char *>(GORD + kMaxG + 2) + kMaxG * sizeof(float));
    double *B = A + G * kWave;                   // G+1 rows
    unsigned char *PA = reinterpret_cast<unsigned char *>(B + (G + 1) * kWave);   // SORTED = false only
    unsigned char *PB = PA + G * kWave;
    if (lds_addr(smem) != 0) {                   // the replay addresses the tables and A by literal LDS offsets
        if (lane == 0) atomicOr(p.err_flag, 2);
        return;
    }
    if (lane < G) {
        DG[lane] = p.del_g[lane];
        const_cast<float *>(delg_f32_table(DG))[lane] = (float)p.del_g[lane];
    }
    if (lane < G + 2) GORD[lane] = p.g_ord[lane];
    const double HUGE_KEY = __longlong_as_double(0x7FE0000000000000LL);
    __syncthreads();
    double wsum = 0.0;
    for (int g = 0; g < G; ++g) wsum += DG[g];
    const double wtot = wsum * wsum;

    double *rec = p.scratch + (size_t)blockIdx.x * 6 * (G + 1) * kWave;
    const size_t GW = (size_t)G * kWave;
    double *gs = pg.gscratch + (size_t)blockIdx.x * (3 + 2 * (size_t)NP1) * GW;
    double *KRB = gs, *DTB = gs + GW;
    // the two gradient-row buffers as offsets onto `gs`: indexing an array of pointers would lose the global address
    // space (flat loads / stores, which also tie up the LDS counter)
    const size_t dboff[2] = {2 * GW, (2 + (size_t)NP1) * GW};
    double *const Dbuf0 = gs + dboff[0];
    double *ASAVE = gs + (2 + 2 * (size_t)NP1) * GW;
    const int nloop = G * G;
    unsigned long long *perm = pg.perm + (size_t)blockIdx.x * ((nloop + kCodesPerWord - 1) / kCodesPerWord) * kWave;

    TileQueue tq;
    tq.init();
    for (;;) {
        int vt = 0, m = 0, l = 0;
        if (!tq.next(p, lane, vt, m, l)) break;
        const int nu = vt * kWave + lane;
        LayerInterp q;
        if (p.kin == nullptr) q = p.li[(size_t)m * p.L + l];
        bool unsorted = false;
        int cur = 0;
        // gas 0: a = k0*amount0 ; D[0] = k0 (d/d amount0), D[1] = dkdT0*amount0 (d/dT), rest 0
        load_gas_g(pg, q, m, l, 0, nu, A, Dbuf0, Dbuf0 + GW, lane, unsorted);
        double alast = A[(G - 1) * kWave + lane];       // last g-ordinate in the ORIGINAL order
        if constexpr (!SORTED) sort_column(A, PA, G, lane);
        for (int pp = 2; pp < NP1; ++pp)
            for (int g = 0; g < G; ++g) Dbuf0[(size_t)pp * GW + g * kWave + lane] = 0.0;

        for (int s = 1; s < p.S; ++s) {
            const int igas = s - 1;
            const int n = igas + 3;  // rankg's `n`
            load_gas_g(pg, q, m, l, s, nu, B, KRB, DTB, lane, unsorted);
            if constexpr (SORTED)                       // the call is rerun on the generic path: no point in merging
                if (__builtin_amdgcn_ballot_w64(unsorted) != 0) break;
            double *Dold = gs + (cur ? dboff[1] : dboff[0]), *Dnew = gs + (cur ? dboff[0] : dboff[1]);
            const double blast = B[(G - 1) * kWave + lane];
            if constexpr (!SORTED) sort_column(B, PB, G, lane);
            if constexpr (SORTED) alast = A[(G - 1) * kWave + lane];
            bool takeB, keepA;
            if (s == 1) { takeB = (alast <= 0.0); keepA = !takeB && (blast <= 0.0); }
            else { keepA = (blast <= 0.0); takeB = !keepA && (alast <= 0.0); }
            const bool do_merge = !(takeB | keepA);
            if (!do_merge) {
                // skip branches :5897-5907, :5930-5937 (slots beyond the ones written keep their old content)
                for (int pp = 0; pp < NP1; ++pp)
                    for (int g = 0; g < G; ++g) {
                        const size_t o = (size_t)pp * GW + g * kWave + lane;
                        double v = Dold[o];
                        if (keepA) {
                            if (pp == igas + 2) v = Dold[(size_t)(igas + 1) * GW + g * kWave + lane];
                            else if (pp == igas + 1) v = 0.0;
                        } else {  // takeB
                            if (pp == igas + 1) v = KRB[g * kWave + lane];
                            else if (pp == igas + 2) v = DTB[g * kWave + lane];
                            else if (s == 1 && pp == 0) v = 0.0;
                        }
                        Dnew[o] = v;
                    }
                if (takeB) {
                    for (int g = 0; g < G; ++g) A[g * kWave + lane] = B[g * kWave + lane];
                    if constexpr (!SORTED) {
                        for (int g = 0; g < G; ++g) PA[g * kWave + lane] = PB[g * kWave + lane];
                        alast = blast;
                    }
                }
            } else {
                // ---- the forward merge, recording the order -------------------------------------------------
                B[G * kWave + lane] = HUGE_KEY;
                double R[NR];
                merge_init<NR>(R, G, lane, A, B[lane], HUGE_KEY);
                MergeElem e0, e1;
                merge_fetch<W32, SORTED>(R[0], lane, A, B, DG, e0, PA, PB);
                WalkState ws = walk_begin(GORD, lane);
                unsigned long long *pw = perm + lane;
                int it = 0;
                auto put5 = [&](unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned c4) {
                    *reinterpret_cast<uint2 *>(pw) = make_uint2(c0 | (c1 << 12) | (c2 << 24), (c2 >> 8) | (c3 << 4) | (c4 << 16));
                    pw += kWave;
                };
                // five 12-bit codes per word; the two element registers swap roles every step, so ten steps are written out
                for (; it + 9 < nloop; it += 10) {
                    const unsigned c0 = merge_step<NR, W32, true, SORTED>(R, e0, e1, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    const unsigned c1 = merge_step<NR, W32, true, SORTED>(R, e1, e0, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    const unsigned c2 = merge_step<NR, W32, true, SORTED>(R, e0, e1, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    const unsigned c3 = merge_step<NR, W32, true, SORTED>(R, e1, e0, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    const unsigned c4 = merge_step<NR, W32, true, SORTED>(R, e0, e1, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    put5(c0, c1, c2, c3, c4);
                    const unsigned c5 = merge_step<NR, W32, true, SORTED>(R, e1, e0, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    const unsigned c6 = merge_step<NR, W32, true, SORTED>(R, e0, e1, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    const unsigned c7 = merge_step<NR, W32, true, SORTED>(R, e1, e0, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    const unsigned c8 = merge_step<NR, W32, true, SORTED>(R, e0, e1, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    const unsigned c9 = merge_step<NR, W32, true, SORTED>(R, e1, e0, ws, lane, A, B, DG, GORD, rec, PA, PB);
                    put5(c5, c6, c7, c8, c9);
                }
                if (it < nloop) {   // G*G not a multiple of 10: the remaining steps, one at a time
                    unsigned long long word = 0;
                    int k = 0;
                    for (int par = 0; it < nloop; ++it, par ^= 1) {
                        const unsigned long long c = par ? merge_step<NR, W32, true, SORTED>(R, e1, e0, ws, lane, A, B, DG, GORD, rec, PA, PB)
                                                         : merge_step<NR, W32, true, SORTED>(R, e0, e1, ws, lane, A, B, DG, GORD, rec, PA, PB);
                        word |= c << (12 * k);
                        if (++k == kCodesPerWord) { *pw = word; pw += kWave; word = 0; k = 0; }
                    }
                    if (k) *pw = word;
                }
                // ---- resolve the bins: merged k -> ASAVE, (frac, 1/sum, weight) -> rec rows 0-2 -------------------
                double ck = 0.0, cs = 0.0;
                const int ig = walk_bins(ws, GORD);
                constexpr int kRB = 5;
                for (int b0 = 0; b0 < G; b0 += kRB) {
                    double rka[kRB], rs1[kRB], rgd[kRB];
                    unsigned rcd[kRB];
#pragma unroll
                    for (int k = 0; k < kRB; ++k) {
                        const int bi = (b0 + k < G) ? b0 + k : G - 1;
                        const unsigned ro = (unsigned)bi * kRecBin + (unsigned)lane * 16u;
                        const dbl2 v0 = gld<dbl2>(rec, ro), v1 = gld<dbl2>(rec, ro + kRecRow);
                        rka[k] = v0.x; rs1[k] = v0.y; rgd[k] = v1.x; rcd[k] = (unsigned)__double_as_longlong(v1.y);
                    }
#pragma unroll
                    for (int k = 0; k < kRB; ++k) {
                        const int b = b0 + k;
                        if (b < G) {
                            double outv = 0.0, fr = 0.0, rinv = 1.0, w = 0.0;
                            if (b < ig) {
                                // the closing element, re-formed from LDS as the walk formed it
                                const int crow = rcd[k] & 31, ccol = (rcd[k] >> 5) & 63;
                                const double cv = A[crow * kWave + lane] + B[ccol * kWave + lane];
                                if constexpr (SORTED) w = pair_weight<W32>(DG, crow, ccol);
                                else w = pair_weight<W32>(DG, PA[crow * kWave + lane], PB[ccol * kWave + lane]);
                                const double ka = rka[k], s1 = rs1[k], cw = cv * w, gd0 = rgd[k];
                                const double gprev = (b == 0 && s1 == 0.0) ? wtot : gd0;
                                const double gdn = gd0 + w;
                                fr = fast_div(GORD[b + 1] - gprev, gdn - gprev);
                                const double kb = (ck + ka) + fr * cw;
                                const double sb = (cs + s1) + fr * w;
                                rinv = fast_div(1.0, sb);
                                outv = fast_div(kb, sb);
                                ck = (1.0 - fr) * cw;
                                cs = (1.0 - fr) * w;
                            } else if (b == ig) {
                                const double kb = ck + ws.kacc, sb = cs + ws.sum1;
                                if (b == G - 1) { outv = fast_div(kb, sb); rinv = fast_div(1.0, sb); } else outv = kb;
                            }
                            const unsigned ro = (unsigned)b * kRecBin + (unsigned)lane * 16u;
                            ASAVE[b * kWave + lane] = outv;
                            // what the replay passes read: (frac | row, column of the closing element; 1 / weight-sum)
                            const double frc = __longlong_as_double((__double_as_longlong(fr) & ~0x7FFLL) | (long long)(rcd[k] & 0x7FFu));
                            gst<dbl2>(rec, ro, dbl2{frc, rinv});
                        }
                    }
                }
                // ---- replay, one gathered vector per pass: the vector in A, bin sums in B (G+1 rows) ---------
                auto stage_row = [&](const double *src) {
                    if constexpr (SORTED) stage_slice(A, src, G, lane); else stage_slice_perm(A, src, PA, G, lane);
                };
                auto stage_col = [&](const double *src) {
                    if constexpr (SORTED) stage_slice(A, src, G, lane); else stage_slice_perm(A, src, PB, G, lane);
                };
                if (pg.gas_mask >> 31) {   // temperature slot: D_old[igas+1][row] + dkdT_new[col]*amount, as a row pass plus a column pass
                    double *DT = Dnew + (size_t)(igas + 2) * GW;
                    stage_row(Dold + (size_t)(igas + 1) * GW);
                    double tail = grad_replay<false, W32, SORTED>(nloop, lane, perm, A, B, DG, PA, PB);
                    grad_resolve<false, false, W32, SORTED>(G, lane, ig, rec, A, B, tail, DT, DG, PA, PB);
                    stage_col(DTB);
                    tail = grad_replay<true, W32, SORTED>(nloop, lane, perm, A, B, DG, PA, PB);
                    grad_resolve<true, true, W32, SORTED>(G, lane, ig, rec, A, B, tail, DT, DG, PA, PB);
                }
                if ((pg.gas_mask >> (igas + 1)) & 1u) {   // the new gas's slot: k_new[col]
                    stage_col(KRB);
                    const double tail = grad_replay<true, W32, SORTED>(nloop, lane, perm, A, B, DG, PA, PB);
                    grad_resolve<true, false, W32, SORTED>(G, lane, ig, rec, A, B, tail, Dnew + (size_t)(igas + 1) * GW, DG, PA, PB);
                }
                for (int pp = 0; pp <= igas; ++pp) {   // earlier gases: D_old[pp][row]
                    if (!((pg.gas_mask >> pp) & 1u)) continue;
                    stage_row(Dold + (size_t)pp * GW);
                    const double tail = grad_replay<false, W32, SORTED>(nloop, lane, perm, A, B, DG, PA, PB);
                    grad_resolve<false, false, W32, SORTED>(G, lane, ig, rec, A, B, tail, Dnew + (size_t)pp * GW, DG, PA, PB);
                }
                for (int pp = n; pp < NP1; ++pp)
                    for (int g = 0; g < G; ++g) Dnew[(size_t)pp * GW + g * kWave + lane] = 0.0;
                stage_slice(A, ASAVE, G, lane);
                if constexpr (!SORTED) {   // the merged spectrum is ascending with the plain del_g weights
                    for (int g = 0; g < G; ++g) PA[g * kWave + lane] = (unsigned char)g;
                    alast = A[(G - 1) * kWave + lane];
                }
            }
            cur ^= 1;
        }
        double *out = p.tau + (((size_t)m * p.L + l) * G) * p.Wpad + nu;
        if constexpr (SORTED) {
            if (unsorted) atomicOr(p.err_flag, 1);
            for (int g = 0; g < G; ++g) out[(size_t)g * p.Wpad] = A[g * kWave + lane];
        } else {
            for (int g = 0; g < G; ++g) out[(size_t)PA[g * kWave + lane] * p.Wpad] = A[g * kWave + lane];
        }
        double *dout = pg.dk + (((size_t)m * p.L + l) * NP1) * G * p.Wpad + nu;
        const double *Dc = gs + (cur ? dboff[1] : dboff[0]);
        for (int pp = 0; pp < NP1; ++pp) {
            const bool wanted = ((pg.gas_mask >> (pp == NP1 - 1 ? 31 : pp)) & 1u) != 0;
            for (int g = 0; g < G; ++g)
                dout[((size_t)pp * G + g) * p.Wpad] = wanted ? Dc[(size_t)pp * GW + g * kWave + lane] : 0.0;
        }
    }
}

// internal dk[L][NP1][G][Wpad] -> reference dk[W][G][L][NP1]   (array-level k_overlapg seam)
__global__ void k_dk_to_ref(const double *__restrict__ src, double *__restrict__ dst, int W, int Wpad, int G,
                            int L, int NP1)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)W * G * L * NP1;
    if (idx >= total) return;
    int pp = (int)(idx % NP1);
    size_t r = idx / NP1;
    int l = (int)(r % L); r /= L;
    int g = (int)(r % G);
    int w = (int)(r / G);
    dst[idx] = src[(((size_t)l * NP1 + pp) * G + g) * Wpad + w];
}

// ------------------------------------------------------------------------------------------------
// .kta file block -> lnK.  The file stores k * 1e20 as float32 in the order [wave][press][temp][g] (Spectroscopy_0
// .read_ktable :2829-2850); the reader divides the float32 array by the Python float 1e20, which NumPy does in float32.
// One gas per launch: kf = the selected wavenumbers' block, as read.  Pad lanes (w >= W) get k = 0.
// ------------------------------------------------------------------------------------------------
__global__ void k_kta_relayout(const float *__restrict__ kf, double *__restrict__ lnK, int W, int Wpad, int G, int NP,
                               int NT, int S, int s, int *flag)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)NP * NT * G * Wpad;
    if (idx >= total) return;
    const int w = (int)(idx % Wpad);
    size_t r = idx / Wpad;
    const int g = (int)(r % G); r /= G;
    const int t = (int)(r % NT);
    const int p = (int)(r / NT);
    double k = 0.0;
    if (w < W) {
        const size_t src = (((size_t)w * NP + p) * NT + t) * G + g;
        const float q = kf[src] / 1.0e20f;
        k = (double)q;
        bool bad = !(k >= 0.0);
        if (g > 0 && q < kf[src - 1] / 1.0e20f) bad = true;
        if (bad) atomicOr(flag, 1);
    }
    lnK[((((size_t)p * NT + t) * S + s) * G + g) * Wpad + w] = encode_lnk(k);
}

// ------------------------------------------------------------------------------------------------
// Layer de-duplication inside a batch of atmospheric states.  The states of a numerical Jacobian differ from
// the unperturbed one at a single profile level, i.e. in two or three layers; every other layer has bit-identical
// (pressure, temperature, amounts) and therefore bit-identical gas opacities.  k_dedup_mark compares each layer
// (m, l) of models m >= 1 with layer l of model 0 and hands out rows of the opacity buffer: row l for a copy,
// a fresh row (atomic counter) otherwise.  k_dedup_gather packs the inputs of the rows that have to be computed
// so that the merge kernel sees them as the layers of one pseudo-model; k_thermal_rt follows tau_slot.
// Nothing is approximated: a layer is shared only when all of its S+2 inputs are equal to the last bit.
// ------------------------------------------------------------------------------------------------
// x1 [n][L], x4 [n][L][4] (or nullptr): further per-layer inputs that are part of a row's identity (the column and the
// composition the Rayleigh continuum of a row is formed from, ansfm_cirsrad_ck_thermal_ray_dev).
__global__ void k_dedup_mark(int n_models, int L, int S, const double *__restrict__ press,
                             const double *__restrict__ temp, const double *__restrict__ amount,
                             int32_t *__restrict__ slot, int32_t *__restrict__ work, int *__restrict__ counter,
                             const double *__restrict__ x1 = nullptr, const double *__restrict__ x4 = nullptr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_models * L) return;
    const int m = i / L, l = i % L;
    if (m == 0) { slot[i] = l; work[l] = i; return; }
    auto bits = [](double x) { return __double_as_longlong(x); };
    bool same = bits(press[i]) == bits(press[l]) && bits(temp[i]) == bits(temp[l]);
    if (same && x1) same = bits(x1[i]) == bits(x1[l]);
    if (same && x4)
        for (int c = 0; c < 4 && same; ++c) same = bits(x4[(size_t)i * 4 + c]) == bits(x4[(size_t)l * 4 + c]);
    for (int s = 0; s < S && same; ++s)
        same = bits(amount[((size_t)m * S + s) * L + l]) == bits(amount[(size_t)s * L + l]);
    if (same) { slot[i] = l; return; }
    const int w = L + atomicAdd(counter, 1);
    slot[i] = w;
    work[w] = i;                                  // (m, l) flattened
}

__global__ void k_dedup_gather(int nwork, int L, int S, const int32_t *__restrict__ work, const double *__restrict__ press,
                               const double *__restrict__ temp, const double *__restrict__ amount,
                               double *__restrict__ press_w, double *__restrict__ temp_w, double *__restrict__ amount_w)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwork) return;
    const int i = work[w], m = i / L, l = i % L;
    press_w[w] = press[i];
    temp_w[w] = temp[i];
    for (int s = 0; s < S; ++s) amount_w[(size_t)s * nwork + w] = amount[((size_t)m * S + s) * L + l];
}

// [rows][G][Wpad] addressed through slot[L] -> reference TAUGAS[W][G][L]
__global__ void k_taugas_from_slots(const double *__restrict__ src, const int32_t *__restrict__ slot, double *__restrict__ dst,
                                    int W, int Wpad, int L, int G)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)W * G * L) return;
    const int l = (int)(idx % L), g = (int)((idx / L) % G), w = (int)(idx / ((size_t)L * G));
    dst[idx] = src[((size_t)slot[l] * G + g) * Wpad + w];
}

// ------------------------------------------------------------------------------------------------
// K11: LBL-table mode (ILBL = LINE_BY_LINE_TABLES): Spectroscopy_0.calc_klbl :1768-1919 /
// calc_klblg :1601-1765 and the gas sum of calculate_gaseous_line_opacity (:3795-3817).
// The table is stored like the k-table with G = 1: lnK[NP][NTa][S][1][Wpad].
// ------------------------------------------------------------------------------------------------
struct LblInterp {
    int ip, a1, b1, a2, b2;   // corner temperature indices (a = it with python wrap, b = it+1)
    double v, u1, u2, omu1, omu2, du1, du2;
};

__device__ __forceinline__ int searchsorted_left_dev(const double *a, int n, double x)
{
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

// One thread per (model, layer).  TEMP is [NTa] or, when temp2d (the reference's NT < 0), [NP][NTa].
// with_grad selects calc_klblg's bracket (no it<0 clamp: python [-1] wrap, :1672-1675).
__global__ void k_layer_prep_lbl(int n_layers_total, const double *__restrict__ lay_press,
                                 const double *__restrict__ lay_temp, int NP, const double *__restrict__ PRESS,
                                 int NTa, const double *__restrict__ TEMP, int temp2d, double press_div,
                                 int grid_f32, int with_grad, LblInterp *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_layers_total) return;
    auto lg = [&](double x) { double r = log(x); return grid_f32 ? (double)(float)r : r; };
    double pmin = __builtin_inf(), pmax = -__builtin_inf();
    for (int k = 0; k < NP; ++k) { double l = lg(PRESS[k]); pmin = fmin(pmin, l); pmax = fmax(pmax, l); }
    double p_l = log(lay_press[i] / press_div);
    bool pcl = false, tcl = false;
    if (p_l < pmin) { p_l = pmin; pcl = true; }
    if (p_l > pmax) { p_l = pmax; pcl = true; }
    const int nt_all = temp2d ? NP * NTa : NTa;
    double tmin = __builtin_inf(), tmax = -__builtin_inf();
    for (int k = 0; k < nt_all; ++k) { tmin = fmin(tmin, TEMP[k]); tmax = fmax(tmax, TEMP[k]); }
    double t_l = lay_temp[i];
    if (t_l < tmin) { t_l = tmin; tcl = true; }
    if (t_l > tmax) { t_l = tmax; tcl = true; }
    // searchsorted(log PRESS, p_l) - 1 on the (ascending) log grid
    int lo = 0, hi = NP;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (lg(PRESS[mid]) < p_l) lo = mid + 1; else hi = mid; }
    int ip = lo - 1;
    if (ip < 0) ip = 0;
    if (ip >= NP - 1) ip = NP - 2;
    const double l0 = lg(PRESS[ip]), l1 = lg(PRESS[ip + 1]);
    const double pden = grid_f32 ? (double)((float)l1 - (float)l0) : l1 - l0;
    LblInterp r;
    r.ip = ip;
    r.v = (grid_f32 && pcl) ? (double)(((float)p_l - (float)l0) / (float)pden) : (p_l - l0) / pden;
    for (int side = 0; side < 2; ++side) {
        const double *T = temp2d ? TEMP + (size_t)(ip + side) * NTa : TEMP;
        int it = searchsorted_left_dev(T, NTa, t_l) - 1;
        if (!with_grad && it < 0) it = 0;
        if (it >= NTa - 1) it = NTa - 2;
        const int itw = it < 0 ? it + NTa : it, itn = it + 1;
        const double den = grid_f32 ? (double)((float)T[itn] - (float)T[itw]) : T[itn] - T[itw];
        double u = (t_l - T[itw]) / den, omu;
        if (grid_f32 && tcl) {
            const float uf = ((float)t_l - (float)T[itw]) / (float)den;
            u = (double)uf;
            omu = (double)(1.0f - uf);
        } else
            omu = 1.0 - u;
        const double du = grid_f32 ? (double)(1.0f / (float)den) : 1. / den;
        if (side) { r.a2 = itw; r.b2 = itn; r.u2 = u; r.omu2 = omu; r.du2 = du; }
        else { r.a1 = itw; r.b1 = itn; r.u1 = u; r.omu1 = omu; r.du1 = du; }
    }
    out[i] = r;
}

__device__ __forceinline__ void interp_klbl(double l1, double l2, double h1, double h2, const LblInterp &q,
                                            double &kk, double &dk)
{   // l1 = (ip,it1) l2 = (ip,it1+1) h1 = (ip+1,it2) h2 = (ip+1,it2+1)      :1898-1917 / :1727-1762
    const bool b1 = lnk_is_boxed(l1), b2 = lnk_is_boxed(l2), b3 = lnk_is_boxed(h1), b4 = lnk_is_boxed(h2);
    kk = 0.0; dk = 0.0;
    const double omv = 1.0 - q.v;
    if (!(b1 | b2 | b3 | b4)) {
        kk = exp(omv * q.omu1 * l1 + q.v * q.omu2 * h1 + q.v * q.u2 * h2 + omv * q.u1 * l2);
        dk = kk * (-l1 * omv * q.du1 - h1 * q.v * q.du2 + h2 * q.v * q.du2 + l2 * omv * q.du1);
    } else if (b1 & b2 & b3 & b4) {
        const double klo1 = lnk_unbox(l1), klo2 = lnk_unbox(l2), khi1 = lnk_unbox(h1), khi2 = lnk_unbox(h2);
        kk = omv * q.omu1 * klo1 + q.v * q.omu2 * khi1 + q.v * q.u2 * khi2 + omv * q.u1 * klo2;
        dk = -klo1 * omv * q.du1 - khi1 * q.v * q.du2 + khi2 * q.v * q.du2 + klo2 * omv * q.du1;
    }
}

// array-level seam: k[W][L][S] (+dkdT)
__global__ void k_calc_klbl_seam(const double *__restrict__ lnK, int W, int Wpad, int NTa, int S, int L,
                                 const LblInterp *__restrict__ li, double *__restrict__ k_out,
                                 double *__restrict__ dk_out)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)L * S * W;
    if (idx >= total) return;
    const int w = (int)(idx % W);
    const int s = (int)((idx / W) % S);
    const int l = (int)(idx / ((size_t)W * S));
    const LblInterp q = li[l];
    const size_t strideT = (size_t)S * Wpad, off = (size_t)s * Wpad + w;
    const double l1 = lnK[((size_t)q.ip * NTa + q.a1) * strideT + off];
    const double l2 = lnK[((size_t)q.ip * NTa + q.b1) * strideT + off];
    const double h1 = lnK[((size_t)(q.ip + 1) * NTa + q.a2) * strideT + off];
    const double h2 = lnK[((size_t)(q.ip + 1) * NTa + q.b2) * strideT + off];
    double kk, dk;
    interp_klbl(l1, l2, h1, h2, q, kk, dk);
    const size_t o = ((size_t)w * L + l) * S + s;
    k_out[o] = kk;
    if (dk_out) dk_out[o] = dk;
}

// fused: tau[n][L][1][Wpad] = sum_s k_s * amount_s ; dk[n][L][S+1][1][Wpad]: slot s = k_s, slot S = sum dkdT_s*amount_s
__global__ void k_lbl_tau(const double *__restrict__ lnK, int Wpad, int NTa, int S, int L, int n_models,
                          const LblInterp *__restrict__ li, const double *__restrict__ amount,
                          double *__restrict__ tau, double *__restrict__ dk)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n_models * L * Wpad;
    if (idx >= total) return;
    const int w = (int)(idx % Wpad);
    const int l = (int)((idx / Wpad) % L);
    const int m = (int)(idx / ((size_t)Wpad * L));
    const LblInterp q = li[(size_t)m * L + l];
    const size_t strideT = (size_t)S * Wpad;
    double t = 0.0, dT = 0.0;
    for (int s = 0; s < S; ++s) {
        const size_t off = (size_t)s * Wpad + w;
        const double l1 = lnK[((size_t)q.ip * NTa + q.a1) * strideT + off];
        const double l2 = lnK[((size_t)q.ip * NTa + q.b1) * strideT + off];
        const double h1 = lnK[((size_t)(q.ip + 1) * NTa + q.a2) * strideT + off];
        const double h2 = lnK[((size_t)(q.ip + 1) * NTa + q.b2) * strideT + off];
        double kk, dkk;
        interp_klbl(l1, l2, h1, h2, q, kk, dkk);
        const double am = amount[((size_t)m * S + s) * L + l];
        t += kk * am;                                   // TAUGAS[:,0,:,i] = k*VLOSDENS ; np.sum(TAUGAS,3)  :3810,:3817
        if (dk) {
            dk[(((size_t)m * L + l) * (S + 1) + s) * Wpad + w] = kk;     // :3813
            dT += dkk * am;                                              // :3814
        }
    }
    tau[((size_t)m * L + l) * Wpad + w] = t;
    if (dk) dk[(((size_t)m * L + l) * (S + 1) + S) * Wpad + w] = dT;
}

// ------------------------------------------------------------------------------------------------
// K3+K4+K5+K6 fused: total opacity, LAYINC gather * SCALE, layer loop with Planck emission,
// ground / solar terms, unit factor and g-quadrature.   "thermal_rt"
// Block = 64 wavenumbers x GY g-groups; thread (lane, gy) integrates g = gy, gy+GY, ...
// ------------------------------------------------------------------------------------------------
constexpr int kGY = 8;       // g-groups per block of the forward RT kernel (157 wavenumber tiles at C2: more waves per tile)
constexpr int kGPer = kMaxG / kGY;  // 4

struct RtParams {
    const double *tau;      // [n][L][G][Wpad], or [unique layers][G][Wpad] addressed through tau_slot
    const int32_t *tau_slot;// [n][L] row of tau holding layer (m, l), or nullptr (identity)
    const double *cont;     // [n][L][Wpad] or nullptr; cont_by_row: [rows][Wpad] addressed like tau
    const double *emi;      // [Li][Wpad] or nullptr  (array-level seam only)
    const double *wave;     // [W]
    const double *delg;     // [G]
    const int32_t *nlayin;  // [P]
    const int32_t *layinc;  // [LIMAX][P]
    const double *scale;    // [n][LIMAX][P]
    const double *emtemp;   // [n][LIMAX][P]
    const double *lay_press;// [n][L]  (Pa)
    const double *tsurf;    // [n]
    const double *emissivity, *solflux, *reflectance, *xfac;  // [W] or nullptr
    const double *sol_ang, *emiss_ang;                        // [P] or nullptr
    double *out;            // per_g ? [n][W][G] : [n][W][P]
    int W, Wpad, G, L, P, LIMAX, ispace, per_g;
    int mode;               // 0 thermal emission; 1 transmission exp(-sum tau) of the path (calculate_transmission_spectrum :4110);
                            // 2 single scattering, plane parallel (calc_singlescatt_plane_spectrum :6509-6600)
    // mode 2: single-scattering albedo of every layer, either given per g (array-level seam) or formed from the vertical
    // opacities as (TAURAY + TAUSCAT) / TAUTOT where TAUTOT > 0 (:4276-4283); layer-mean phase function per path; BRDF
    const double *omega;    // [Li][G][Wpad] along the path, or nullptr
    const double *sca;      // [L][Wpad] TAURAY + TAUSCAT of the layers, or nullptr
    const double *phase;    // [P][L][Wpad] (by layer; the array-level seam passes L = Li, identity LAYINC)
    const double *brdf;     // [W][P] or nullptr
    // The states of a numerical Jacobian share the top of every path with state 0 (k_thermal_rt<.., PREFIX>): state 0's
    // launch (PREFIX 1, m0 = 0) leaves (taud, trold, spec) after every layer of the path in `prefix`
    // [P][LIMAX][3][G][Wpad]; the launch of the states m0 .. (PREFIX 2) starts state m's path ip at layer jstart[m][ip] --
    // the first one whose opacity row, continuum, SCALE or EMTEMP is not state 0's -- from that record.  Same bits.
    double *prefix;
    const int32_t *jstart;  // [n][P]
    int m0;
    int cont_by_row;
};

// same[m][lay] = the opacity row of (m, lay) is state 0's row and (cont != nullptr) so is its continuum, bit for bit.
// grid (L, n - 1), block 256
__global__ void k_rt_same(int L, int Wpad, const int32_t *__restrict__ slot, const double *__restrict__ cont, unsigned char *same)
{
    const int lay = blockIdx.x, m = blockIdx.y + 1;
    int differs = slot[(size_t)m * L + lay] != slot[lay];
    if (!differs && cont) {
        const long long *a = reinterpret_cast<const long long *>(cont + ((size_t)m * L + lay) * Wpad);
        const long long *b = reinterpret_cast<const long long *>(cont + (size_t)lay * Wpad);
        for (int i = threadIdx.x; i < Wpad; i += blockDim.x) differs |= (a[i] != b[i]);
    }
    differs = __syncthreads_or(differs);
    if (threadIdx.x == 0) same[(size_t)m * L + lay] = differs ? 0 : 1;
}

// jstart[m][ip] = number of leading layers of path ip that state m shares with state 0 (one wave per (m, ip); m = 0: 0)
__global__ __launch_bounds__(64) void k_rt_jstart(int n, int L, int P, int LIMAX, const int32_t *__restrict__ nlayin,
                                                  const int32_t *__restrict__ layinc, const double *__restrict__ scale,
                                                  const double *__restrict__ emtemp, const unsigned char *__restrict__ same,
                                                  int32_t *jstart)
{
    const int idx = blockIdx.x, lane = threadIdx.x;
    const int m = idx / P, ip = idx % P;
    int first = 0;
    if (m > 0) {
        const int nl = nlayin[ip];
        const size_t pm = (size_t)m * LIMAX * P + ip, p0 = ip;
        first = nl;
        for (int j0 = 0; j0 < nl; j0 += 64) {
            const int j = j0 + lane;
            bool bad = false;
            if (j < nl) {
                const int lay = layinc[(size_t)j * P + ip];
                bad = !same[(size_t)m * L + lay] ||
                      __double_as_longlong(scale[pm + (size_t)j * P]) != __double_as_longlong(scale[p0 + (size_t)j * P]) ||
                      __double_as_longlong(emtemp[pm + (size_t)j * P]) != __double_as_longlong(emtemp[p0 + (size_t)j * P]);
            }
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(bad);
            if (hit != 0) { first = j0 + __builtin_ctzll(hit); break; }
        }
    }
    if (lane == 0) jstart[idx] = first;
}

__device__ __forceinline__ double planck_bb(double a, double c2y, double T)
{
    return a / (exp(c2y / T) - 1.0);  // ForwardModel_0.py:6223-6225
}

// BATCH: the build for many models per launch (a Jacobian's states).  One block is eight waves, two per SIMD; at the
// kernel's natural 142 registers a second block does not fit on the CU, and a batch has the blocks to fill it: capped at 128
// (14 spilled) the 201 states of a C3 Jacobian take 9.2 instead of 11.4 ms.  A single model has 157 blocks for 256 CUs and
// only pays for the spills (0.093 -> 0.107 ms): it keeps the uncapped build.
template <bool BATCH, int PREFIX = 0>
__global__ __launch_bounds__(kWave *kGY) __attribute__((amdgpu_waves_per_eu(BATCH ? 4 : 1, BATCH ? 4 : 8))) void k_thermal_rt(RtParams p)
{
    __shared__ double red[kGY][kWave];
    const int lane = threadIdx.x, gy = threadIdx.y;
    // grid = (models, paths, wavenumber tiles): the models of a batch that share opacity rows (de-duplicated Jacobian
    // states) run next to each other on a wavenumber tile, so the rows are re-read out of L2 instead of HBM
    const int nu = blockIdx.z * kWave + lane;
    const int nuc = nu < p.W ? nu : p.W - 1;
    const int ip = blockIdx.y, m = blockIdx.x + (PREFIX != 0 ? p.m0 : 0);
    const int nl = p.nlayin[ip];
    const int G = p.G;
    const double c1 = 1.1911e-12, c2 = 1.439;  // ForwardModel_0.py:6214-6215
    const double wv = p.wave[nuc];
    double y, a;
    if (p.ispace == 0) { y = wv; a = c1 * (y * y * y); }
    else { y = 1.0e4 / wv; a = c1 * (y * y * y * y * y) / 1.0e4; }
    const double c2y = c2 * y;

    double taud[kGPer], trold[kGPer], spec[kGPer];
#pragma unroll
    for (int k = 0; k < kGPer; ++k) { taud[k] = 0.0; trold[k] = 1.0; spec[k] = 0.0; }

    const size_t pathbase = (size_t)m * p.LIMAX * p.P + ip;
    // Per-layer metadata of the path (opacity row, SCALE, Planck function at EMTEMP) once into LDS: the layer loop then
    // has no dependent index -> row -> data chain, and the opacity loads of layer j+1 are issued before layer j is
    // integrated (the loop is a serial recurrence in the optical depth; without this it runs at memory latency).
    extern __shared__ double rt_meta[];                 // [3][LIMAX]: row index (as double), scale, B(nu-independent part: T)
    double *m_row = rt_meta, *m_sc = rt_meta + p.LIMAX, *m_T = rt_meta + 2 * p.LIMAX;
    for (int j = lane + gy * kWave; j < nl; j += kWave * kGY) {
        const int lay = p.layinc[(size_t)j * p.P + ip];
        const size_t ri = p.tau_slot ? (size_t)p.tau_slot[(size_t)m * p.L + lay] : (size_t)m * p.L + lay;
        m_row[j] = (double)ri;                          // < 2^53, exact
        m_sc[j] = p.scale[pathbase + (size_t)j * p.P];
        m_T[j] = p.emtemp[pathbase + (size_t)j * p.P];
        rt_meta[3 * p.LIMAX + j] = (double)lay;
    }
    __syncthreads();
    const double *m_lay = rt_meta + 3 * p.LIMAX;
    auto fetch = [&](int j, double tv[kGPer], double &tc, double &em) {
        const size_t ri = (size_t)m_row[j];
        const int lay = (int)m_lay[j];
        const double *trow = p.tau + (ri * G) * p.Wpad + nu;
        tc = p.cont ? p.cont[(p.cont_by_row ? ri : (size_t)m * p.L + lay) * p.Wpad + nu] : 0.0;
        em = p.emi ? p.emi[(size_t)j * p.Wpad + nu] : 0.0;
#pragma unroll
        for (int k = 0; k < kGPer; ++k) {
            const int g = gy + k * kGY;
            tv[k] = (g < G) ? trow[(size_t)g * p.Wpad] : 0.0;
        }
    };
    // mode 2 (single scattering): ssfac = mu0 / (mu0 + mu) and the solar flux over 4 pi, wave-uniform per path (:6557-6559)
    const double PI_ = 3.141592653589793;
    double ssfac = 0.0, mu0s = 0.0, sflux = 0.0;
    if (p.mode == 2) {
        const double mu = cos(p.emiss_ang[ip] / 180. * PI_);
        mu0s = cos(p.sol_ang[ip] / 180. * PI_);
        ssfac = mu0s / (mu0s + mu);
        sflux = p.solflux ? p.solflux[nuc] : 0.0;
    }
    auto scatter_term = [&](int j, int k, double tvk, double tc, double dtr) -> double {
        // (trold - tr) * ssfac * omega * phase * SOLFLUX / (4 pi), in the reference's order of operations (:6577)
        const int lay = (int)m_lay[j];
        const int g = gy + k * kGY;
        double om;
        if (p.omega) om = p.omega[((size_t)j * G + g) * p.Wpad + nu];
        else {
            const double tt = tvk + tc;                                   // vertical TAUTOT of the layer (:3989)
            om = (tt > 0.0) ? p.sca[(size_t)lay * p.Wpad + nu] / tt : 0.0;
        }
        const double ph = p.phase[((size_t)ip * p.L + lay) * p.Wpad + nu];
        return dtr * ssfac * om * ph * sflux / (4. * PI_);
    };
    double tvA[kGPer], tvB[kGPer], tcA = 0.0, tcB = 0.0, emA = 0.0, emB = 0.0;
    auto integrate = [&](int j, const double tv[kGPer], double tc, double em) {
        const double sc = m_sc[j];
        const double bb = planck_bb(a, c2y, m_T[j]);
#pragma unroll
        for (int k = 0; k < kGPer; ++k) {
            const int g = gy + k * kGY;
            if (g < G) {
                const double t = (tv[k] + tc) * sc;  // :3989, :4006
                taud[k] += t;
                const double tr = exp(-taud[k]);
                if (p.mode == 2) spec[k] += scatter_term(j, k, tv[k], tc, trold[k] - tr);   // before the thermal term (:6577-6581)
                spec[k] += (trold[k] - tr) * bb;  // :6345-6348
                if (p.emi) spec[k] += em * tr;
                trold[k] = tr;
            }
        }
    };
    // the record after layer jd of the path: [ip][jd][3][G][Wpad]
    auto record = [&](int jd) -> double * { return p.prefix + (((size_t)ip * p.LIMAX + jd) * 3) * (size_t)G * p.Wpad + nu; };
    auto leave = [&](int jd) {
        if constexpr (PREFIX == 1) {
            double *r = record(jd);
#pragma unroll
            for (int k = 0; k < kGPer; ++k) {
                const int g = gy + k * kGY;
                if (g < G) {
                    r[(size_t)g * p.Wpad] = taud[k]; r[((size_t)G + g) * p.Wpad] = trold[k]; r[((size_t)2 * G + g) * p.Wpad] = spec[k];
                }
            }
        }
    };
    int j = 0;
    if constexpr (PREFIX == 2) {
        j = p.jstart[(size_t)m * p.P + ip];              // block-uniform
        if (j > 0) {
            const double *r = record(j - 1);
#pragma unroll
            for (int k = 0; k < kGPer; ++k) {
                const int g = gy + k * kGY;
                if (g < G) {
                    taud[k] = r[(size_t)g * p.Wpad]; trold[k] = r[((size_t)G + g) * p.Wpad]; spec[k] = r[((size_t)2 * G + g) * p.Wpad];
                }
            }
        }
    }
    if (j < nl) fetch(j, tvA, tcA, emA);
    for (; j + 1 < nl; j += 2) {                         // ping-pong buffers: no register rotation
        fetch(j + 1, tvB, tcB, emB);
        integrate(j, tvA, tcA, emA);
        leave(j);
        if (j + 2 < nl) fetch(j + 2, tvA, tcA, emA);
        integrate(j + 1, tvB, tcB, emB);
        leave(j + 1);
    }
    if (j < nl) { integrate(j, tvA, tcA, emA); leave(j); }
    // surface / bottom-of-atmosphere term  (:6354-6365)
    int i1 = (int)(nl / 2.0) - 1;
    if (i1 < 0) i1 += nl;
    const double *lp = p.lay_press + (size_t)m * p.L;
    const double p1 = lp[p.layinc[(size_t)i1 * p.P + ip]];
    const double p2 = lp[p.layinc[(size_t)(nl - 1) * p.P + ip]];
    double radground = 0.0;
    const bool ground = p2 > p1;
    if (ground) {
        const double ts = p.tsurf[m];
        if (ts <= 0.0) radground = planck_bb(a, c2y, p.emtemp[pathbase + (size_t)(nl - 1) * p.P]);
        else radground = planck_bb(a, c2y, ts) * (p.emissivity ? p.emissivity[nuc] : 0.0);
    }
    const double sola = p.sol_ang ? p.sol_ang[ip] : 180.0;
    const double emia = p.emiss_ang ? p.emiss_ang[ip] : 180.0;
    const bool solar_on = (emia < 90.) && (sola < 90.);
    double solterm = 0.0, muratio = 0.0;
    if (solar_on) {
        const double PI = 3.141592653589793;
        const double mu = cos(emia / 180. * PI), mu0 = cos(sola / 180. * PI);
        muratio = mu / mu0;
        solterm = (p.solflux ? p.solflux[nuc] : 0.0) * (p.reflectance ? p.reflectance[nuc] : 0.0);
    }
    const double xf = p.xfac ? p.xfac[nuc] : 1.0;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < kGPer; ++k) {
        const int g = gy + k * kGY;
        if (g < G) {
            double s = spec[k];
            if (p.mode == 1) s = exp(-taud[k]);                               // :4116, xfac = solar flux when IFORM = 4 (:4119-4127)
            else if (p.mode == 2) {                                           // :6585-6596: lower boundary whatever the geometry
                const double ts = p.tsurf[m];
                double rg;
                if (ts <= 0.0) rg = planck_bb(a, c2y, p.emtemp[pathbase + (size_t)(nl - 1) * p.P]);
                else rg = planck_bb(a, c2y, ts) * (p.emissivity ? p.emissivity[nuc] : 0.0);
                s += trold[k] * rg;
                s += trold[k] * sflux * mu0s * (p.brdf ? p.brdf[(size_t)nuc * p.P + ip] : 0.0);
            } else {
                if (ground) s += trold[k] * radground;
                if (solar_on) s += trold[k] * exp(-taud[k] * muratio) * solterm;  // :6368-6373
            }
            s = s * xf;                                                       // :4244
            if (p.per_g) {
                if (nu < p.W) p.out[((size_t)m * p.W + nu) * G + g] = s;
            } else {
                acc += s * p.delg[g];                                         // :4504
            }
        }
    }
    if (!p.per_g) {
        red[gy][lane] = acc;
        __syncthreads();
        if (gy == 0 && nu < p.W) {
            double t = red[0][lane];
#pragma unroll
            for (int k = 1; k < kGY; ++k) t += red[k][lane];
            p.out[((size_t)m * p.W + nu) * p.P + ip] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3g+K4g+K6: thermal emission with analytic gradients.   "thermal_rtg"
// calc_thermal_emission_spectrumg (ForwardModel_0.py:6380-6504) carries dtr/dq for every
// (parameter, layer) through the layer loop: O(NPAR*Li^2) per (nu,g).  The recursion is linear in
// dTAU, so   dspec/dq[k,m] = c_m * dTAU[k,m]  (+ (trold_m - tr_m) dB/dT_m for k == NVMR)   with
//     c_m = tr_m B_m - R_m ,   R_m = sum_{j>m} (trold_j - tr_j) B_j + trold_N * radground ,
// one backward sweep (O(Li)); the g-quadrature (:4507) is folded in:
//     out[k,m] = xfac * ( SCALE_m * ( fac_k * sum_g dg c_g dk[slot_k][g] + dcont_k * sum_g dg c_g )
//                         + [k==NVMR] dBdT_m * sum_g dg (trold_m - tr_m)_g )
// so neither dTAUTOT_LAYINC (W,G,NPAR,Li,P) nor dSPECOUT (W,G,NPAR,Li) is materialised.
// Block = 64 wavenumbers x 4 g-groups; pass 1 stores trold_j per (g) to a workspace.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxPar = 256;   // parameters of dSPECOUT (NVMR + 2 + NDUST): sizes the slot table in the kernel arguments only
struct RtGParams {
    RtParams r;              // r.out = SPECOUT [n][W][P]
    const double *dk;        // [n][L][NP1][G][Wpad]
    const double *dcont;     // [n][NPAR][L][Wpad] or nullptr   (dTAUCON)
    const double *dcont_gas; // [L][Wpad] or nullptr: one array added to the dTAUCON of EVERY gas parameter (kpar < NVMR) -- the
                             // Rayleigh term of calculate_layer_opacity (:3955-3957) without NVMR copies of it
    double *trold_ws;        // [n][P][LIMAX+1][G][Wpad]
    double *dspec;           // [n][P][NPAR][LIMAX][Wpad]   (internal layout)
    double *dtsurf;          // [n][W][P]
    int NPAR, NVMR, NP1;
    unsigned gas_mask;                    // as OverlapGParams::gas_mask: the slots of the other gases are zero and not read
    signed char slot_of_param[kMaxPar];   // -1 none, 0..S-1 gas slot (x1e-4), S = temperature slot
};

__device__ __forceinline__ void planckg_dev(int ispace, double y, double T, double &bb, double &dBdT)
{
    const double c1 = 1.1911e-12, c2 = 1.439;
    double a, ap;
    if (ispace == 0) { a = c1 * (y * y * y); ap = c1 * c2 * (y * y * y * y) / (T * T); }
    else { a = c1 * (y * y * y * y * y) / 1.0e4; ap = c1 * c2 * (y * y * y * y * y * y) / 1.0e4 / (T * T); }
    const double e = exp(c2 * y / T);
    const double b = e - 1.0;
    bb = a / b;
    dBdT = e * ap / (b * b);   // ForwardModel_0.py:6274-6281
}

// GY = g-groups (waves) per wavenumber tile: the kernel streams (S+1) gradient rows per layer and C2 has only 157 tiles,
// so the launch picks the largest GY whose reduction buffer fits in LDS (16 up to S = 12).
template <int GY>
__global__ __launch_bounds__(kWave *GY) void k_thermal_rtg(RtGParams q)
{
    constexpr int kGPerG = kMaxG / GY;
    const RtParams &p = q.r;
    extern __shared__ double red[];  // [NP1+2][GY][kWave]
    const int lane = threadIdx.x, gy = threadIdx.y;
    // grid = (models, paths, wavenumber tiles): the models of a batch that share opacity rows (de-duplicated Jacobian
    // states) run next to each other on a wavenumber tile, so the rows are re-read out of L2 instead of HBM
    const int nu = blockIdx.z * kWave + lane;
    const int nuc = nu < p.W ? nu : p.W - 1;
    const int ip = blockIdx.y, m = blockIdx.x;
    const int nl = p.nlayin[ip];
    const bool transmission = p.mode == 1;   // calculate_transmission_spectrum with return_grad (:4110-4131)
    const int G = p.G, NP1 = q.NP1, NR = NP1 + 2;
    const double wv = p.wave[nuc];
    const double y = (p.ispace == 0) ? wv : 1.0e4 / wv;
    const size_t pathbase = (size_t)m * p.LIMAX * p.P + ip;
    const size_t GWp = (size_t)G * p.Wpad;
    double *tws = q.trold_ws + (((size_t)m * p.P + ip) * (p.LIMAX + 1)) * GWp + nu;

    double trold[kGPerG], spec[kGPerG];
#pragma unroll
    for (int k = 0; k < kGPerG; ++k) { trold[k] = 1.0; spec[k] = 0.0; }
    // ---- pass 1: forward, product form tr = trold*exp(-tau_j) (:6446-6452) -----------------------------
    for (int j = 0; j < nl; ++j) {
        const int lay = p.layinc[(size_t)j * p.P + ip];
        const double sc = p.scale[pathbase + (size_t)j * p.P];
        const double T = p.emtemp[pathbase + (size_t)j * p.P];
        const double tc = p.cont ? p.cont[((size_t)m * p.L + lay) * p.Wpad + nu] : 0.0;
        double bb = 0.0, dB = 0.0;
        if (!transmission) planckg_dev(p.ispace, y, T, bb, dB);
        const double *trow = p.tau + (((size_t)m * p.L + lay) * G) * p.Wpad + nu;
#pragma unroll
        for (int k = 0; k < kGPerG; ++k) {
            const int g = gy + k * GY;
            if (g < G) {
                tws[(size_t)j * GWp + (size_t)g * p.Wpad] = trold[k];
                const double t = (trow[(size_t)g * p.Wpad] + tc) * sc;
                const double tr = trold[k] * exp(-t);
                spec[k] += (trold[k] - tr) * bb;
                trold[k] = tr;
            }
        }
    }
    int i1 = (int)(nl / 2.0) - 1;
    if (i1 < 0) i1 += nl;
    const double *lp = p.lay_press + (size_t)m * p.L;
    const bool ground = !transmission && lp[p.layinc[(size_t)(nl - 1) * p.P + ip]] > lp[p.layinc[(size_t)i1 * p.P + ip]];
    double radground = 0.0, dradgrounddT = 0.0;
    if (ground) {
        const double ts = p.tsurf[m];
        if (ts <= 0.0) planckg_dev(p.ispace, y, p.emtemp[pathbase + (size_t)(nl - 1) * p.P], radground, dradgrounddT);
        else {
            planckg_dev(p.ispace, y, ts, radground, dradgrounddT);
            const double em = p.emissivity ? p.emissivity[nuc] : 0.0;
            radground *= em;
            dradgrounddT *= em;
        }
    }
    const double xf = p.xfac ? p.xfac[nuc] : 1.0;
    double R[kGPerG];
    {
        double accs = 0.0, acct = 0.0;
#pragma unroll
        for (int k = 0; k < kGPerG; ++k) {
            const int g = gy + k * GY;
            R[k] = 0.0;
            if (g < G) {
                double sgl = transmission ? trold[k] : spec[k];     // mode 1: exp(-tau of the path) (:4110)
                if (ground) sgl += trold[k] * radground;
                accs += (sgl * xf) * p.delg[g];
                acct += ((ground ? trold[k] * dradgrounddT : 0.0) * xf) * p.delg[g];
                R[k] = ground ? trold[k] * radground : 0.0;
            }
        }
        red[(0 * GY + gy) * kWave + lane] = accs;
        red[(1 * GY + gy) * kWave + lane] = acct;
        __syncthreads();
        if (gy == 0 && nu < p.W) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int k = 0; k < GY; ++k) { a += red[(0 * GY + k) * kWave + lane]; b += red[(1 * GY + k) * kWave + lane]; }
            p.out[((size_t)m * p.W + nu) * p.P + ip] = a;
            q.dtsurf[((size_t)m * p.W + nu) * p.P + ip] = b;
        }
        __syncthreads();
    }
    // ---- pass 2: backward sweep ------------------------------------------------------------------------
    double trnext[kGPerG];  // tr_m = trold_{m+1}
    double trfin[kGPerG];   // transmission of the whole path
#pragma unroll
    for (int k = 0; k < kGPerG; ++k) trnext[k] = trfin[k] = trold[k];
    double *dsp = q.dspec + (((size_t)m * p.P + ip) * q.NPAR) * (size_t)p.LIMAX * p.Wpad + nu;
    for (int mm = nl - 1; mm >= 0; --mm) {
        const int lay = p.layinc[(size_t)mm * p.P + ip];
        const double sc = p.scale[pathbase + (size_t)mm * p.P];
        const double T = p.emtemp[pathbase + (size_t)mm * p.P];
        double bb = 0.0, dB = 0.0;
        if (!transmission) planckg_dev(p.ispace, y, T, bb, dB);
        const double *dkl = q.dk + (((size_t)m * p.L + lay) * NP1) * GWp + nu;
        double X = 0.0, Z = 0.0;
        double *rb = red;
        (void)NR;
        double cg[kGPerG];
#pragma unroll
        for (int k = 0; k < kGPerG; ++k) {
            const int g = gy + k * GY;
            cg[k] = 0.0;
            if (g < G) {
                const double to = tws[(size_t)mm * GWp + (size_t)g * p.Wpad];
                const double tr = trnext[k];
                const double c = transmission ? -trfin[k] : tr * bb - R[k];   // mode 1: d exp(-tau) / d tau_m (:4129)
                const double dgk = p.delg[g];
                cg[k] = c * dgk;
                X += cg[k];
                Z += (to - tr) * dgk;
                R[k] += (to - tr) * bb;
                trnext[k] = to;
            }
        }
        __syncthreads();   // the previous layer's partial sums have been consumed by every thread
        for (int sidx = 0; sidx < NP1; ++sidx) {
            if (!((q.gas_mask >> (sidx == NP1 - 1 ? 31 : sidx)) & 1u)) continue;     // slot_of_param points away from it
            double ysum = 0.0;
#pragma unroll
            for (int k = 0; k < kGPerG; ++k) {
                const int g = gy + k * GY;
                if (g < G) ysum += cg[k] * dkl[((size_t)sidx * G + g) * p.Wpad];
            }
            rb[((2 + sidx) * GY + gy) * kWave + lane] = ysum;
        }
        rb[(0 * GY + gy) * kWave + lane] = X;
        rb[(1 * GY + gy) * kWave + lane] = Z;
        __syncthreads();
        double Xs = 0.0, Zs = 0.0;
#pragma unroll
        for (int k = 0; k < GY; ++k) { Xs += rb[(0 * GY + k) * kWave + lane]; Zs += rb[(1 * GY + k) * kWave + lane]; }
        for (int kpar = gy; kpar < q.NPAR; kpar += GY) {
            const int slot = q.slot_of_param[kpar];
            double v = 0.0;
            if (slot >= 0) {
                double ys = 0.0;
#pragma unroll
                for (int k = 0; k < GY; ++k) ys += rb[((2 + slot) * GY + k) * kWave + lane];
                v = ys * ((slot == NP1 - 1) ? 1.0 : 1.0e-4);      // :3870 / :3872
            }
            if (q.dcont) v += q.dcont[(((size_t)m * q.NPAR + kpar) * p.L + lay) * p.Wpad + nu] * Xs;
            if (q.dcont_gas && kpar < q.NVMR) v += q.dcont_gas[(size_t)lay * p.Wpad + nu] * Xs;
            v *= sc;                                               // :4012
            if (kpar == q.NVMR && !transmission) v += Zs * dB;     // :6467-6468
            v *= xf;                                               // :4247
            if (v != v) v = 0.0;                                   // nan_to_num :4507
            dsp[((size_t)kpar * p.LIMAX + mm) * p.Wpad] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Array-level seam of calc_thermal_emission_spectrumg (ForwardModel_0.py:6380-6504) on the reference's layouts: one
// thread per (wavenumber, g).  The reference carries d tr / dq for every (parameter, layer) pair through the layer loop,
// O(NPAR Li^2); the recursion is linear in dTAU, so (as in k_thermal_rtg)
//     dspec[k][m] = dTAU[k][m] * (tr_m B_m - R_m) + [k == NVMR] (trold_m - tr_m) dB/dT_m ,
//     R_m = sum_{j > m} (trold_j - tr_j) B_j + tr_N * radground
// -- a forward pass that parks trold_j in the output's parameter-0 row and one backward sweep.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_thermal_emission_g_seam(int ispace, int W, int G, int NPAR, int Li, int NVMR,
                                                                  const double *__restrict__ wave,
                                                                  const double *__restrict__ tau,      // [W][G][Li]
                                                                  const double *__restrict__ dtau,     // [W][G][NPAR][Li]
                                                                  const double *__restrict__ temp,     // [Li]
                                                                  const double *__restrict__ press,    // [Li]
                                                                  double tsurf, const double *__restrict__ emissivity,
                                                                  double *__restrict__ spec,           // [W][G]
                                                                  double *__restrict__ dspec,          // [W][G][NPAR][Li]
                                                                  double *__restrict__ dtsurf)         // [W][G]
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)W * G) return;
    const int w = (int)(idx / G);
    const double wv = wave[w];
    const double y = (ispace == 0) ? wv : 1.0e4 / wv;
    const double *t = tau + idx * Li;
    const double *dt = dtau + idx * (size_t)NPAR * Li;
    double *ds = dspec + idx * (size_t)NPAR * Li;
    double trold = 1.0, sp = 0.0;
    for (int j = 0; j < Li; ++j) {                       // :6446-6452, product form of the transmission
        double bb, dB;
        planckg_dev(ispace, y, temp[j], bb, dB);
        const double tr = trold * exp(-t[j]);
        sp += (trold - tr) * bb;
        ds[j] = trold;                                   // parked: read back (then overwritten) by the sweep
        trold = tr;
    }
    int i1 = (int)(Li / 2.0) - 1;                        // python index int(NLAYIN/2)-1, -1 wraps to the last layer
    if (i1 < 0) i1 += Li;
    double radground = 0.0, dradground = 0.0, R = 0.0;
    if (press[Li - 1] > press[i1]) {                     // not a limb path: the lower boundary contributes (:6479-6496)
        if (tsurf <= 0.0) planckg_dev(ispace, y, temp[Li - 1], radground, dradground);
        else {
            planckg_dev(ispace, y, tsurf, radground, dradground);
            radground *= emissivity[w];
            dradground *= emissivity[w];
        }
        sp += trold * radground;
        R = trold * radground;
    }
    spec[idx] = sp;
    dtsurf[idx] = (press[Li - 1] > press[i1]) ? trold * dradground : 0.0;
    double trn = trold;                                  // tr_m = trold_{m+1}
    for (int m = Li - 1; m >= 0; --m) {
        double bb, dB;
        planckg_dev(ispace, y, temp[m], bb, dB);
        const double to = ds[m];
        const double c = trn * bb - R;
        for (int k = 0; k < NPAR; ++k) {
            double v = dt[(size_t)k * Li + m] * c;
            if (k == NVMR) v += (to - trn) * dB;         // :6467-6468
            ds[(size_t)k * Li + m] = v;
        }
        R += (to - trn) * bb;
        trn = to;
    }
}

// internal dspec[P][NPAR][LIMAX][Wpad] -> reference dSPECOUT[W][NPAR][LIMAX][P]
__global__ void k_dspec_to_ref(const double *__restrict__ src, double *__restrict__ dst, int W, int Wpad,
                               int NPAR, int LIMAX, int P, const int32_t *__restrict__ nlayin)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)W * NPAR * LIMAX * P;
    if (idx >= total) return;
    int ip = (int)(idx % P);
    size_t r = idx / P;
    int j = (int)(r % LIMAX); r /= LIMAX;
    int k = (int)(r % NPAR);
    int w = (int)(r / NPAR);
    dst[idx] = (j < nlayin[ip]) ? src[(((size_t)ip * NPAR + k) * LIMAX + j) * Wpad + w] : 0.0;
}

// ------------------------------------------------------------------------------------------------
// layout helpers (host-pointer seams): src[W][X1][X2] -> dst[(x1,x2 or x2,x1)][Wpad]
// ------------------------------------------------------------------------------------------------
// blockIdx.z = model of a batch (src_stride / dst_stride elements apart; 0 for a single array).
// Through a 32 x 32 LDS tile: block (32, 8); grid (Wpad / 32, ceil(X1 X2 / 32), batch).  Reads run along x (the source's
// fastest axis), writes along w.
__global__ __launch_bounds__(256) void k_transpose_w_last(const double *__restrict__ src, double *__restrict__ dst, int W, int Wpad,
                                                          int X1, int X2, int swap12, double padval, size_t src_stride,
                                                          size_t dst_stride)
{
    __shared__ double tile[32][33];
    const int X = X1 * X2;
    src += (size_t)blockIdx.z * src_stride;
    dst += (size_t)blockIdx.z * dst_stride;
    const int w0 = blockIdx.x * 32, x0 = blockIdx.y * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int w = w0 + ty + 8 * k, x = x0 + tx;
        tile[ty + 8 * k][tx] = (w < W && x < X) ? src[(size_t)w * X + x] : padval;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + ty + 8 * k;           // source column = x1 * X2 + x2
        if (x < X) {
            const int row = swap12 ? (x % X2) * X1 + x / X2 : x;
            dst[(size_t)row * Wpad + w0 + tx] = tile[tx][ty + 8 * k];
        }
    }
}
// src[X1][X2][Wpad] -> dst[W][X1][X2]  (swap12: dst[W][X2][X1])
__global__ void k_w_to_first(const double *__restrict__ src, double *__restrict__ dst, int W, int Wpad,
                             int X1, int X2, int swap12)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)W * X1 * X2;
    if (idx >= total) return;
    int w = (int)(idx / ((size_t)X1 * X2));
    int r = (int)(idx % ((size_t)X1 * X2));
    int x1, x2;
    if (swap12) { x1 = r % X1; x2 = r / X1; }
    else { x2 = r % X2; x1 = r / X2; }
    dst[idx] = src[((size_t)x1 * X2 + x2) * Wpad + w];
}
// k[W][G][L][S] (reference layout) -> kin[S][L][G][Wpad]
__global__ void k_kin_permute(const double *__restrict__ src, double *__restrict__ dst, int W, int Wpad,
                              int G, int L, int S)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)S * L * G * Wpad;
    if (idx >= total) return;
    int w = (int)(idx % Wpad);
    size_t r = idx / Wpad;
    int g = (int)(r % G); r /= G;
    int l = (int)(r % L);
    int s = (int)(r / L);
    dst[idx] = (w < W) ? src[(((size_t)w * G + g) * L + l) * S + s] : 0.0;
}

}  // namespace ansfm
