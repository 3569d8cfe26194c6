// ansfm_merge32.hip.h -- k_ck_overlap32: calc_k + k_overlap / rank (Spectroscopy_0.py:2298, ForwardModel_0.py:6029-6173)
// with the row heads of the G-way merge ordered on 32-BIT KEYS.
//
// Same decomposition as k_ck_overlap (ansfm_kernels.hip.h): one lane per (wavenumber, layer) cell, a[G] / b[G+1] per lane in
// LDS as [index][lane], the sorted sequence of the G*G sums a_i + b_j produced by a G-way streaming merge whose row heads
// are a sorted list in registers, rank()'s walk consuming it on the fly.  What is different:
//
//  * list keys are 32 bits: the float32 rounding of the head's value (v_cvt_f32_f64 is monotone, values are >= 0, so
//    the bit patterns order like unsigned integers) with the low 5 mantissa bits replaced by the row.  One insertion is
//    NR full-rate instructions  t_0 = min(x, s_1), t_k = med3(x, s_k, s_k+1), t_NR-1 = max(x, s_NR-1)  (v_med3_u32)
//    instead of 2(NR-1) half-rate v_min_f64 / v_max_f64;
//  * the column a row has reached is no longer part of the key: one byte per (row, lane) in LDS, four rows to a dword
//    so that the bank depends on the lane only.  The byte of the list's SECOND entry is fetched one step ahead (its row
//    is known then), so the chain from "winner known" to "next key ready" still holds one LDS round trip, not two;
//  * two keys whose float32 value bits coincide (values within 2^-18) are ordered by their exact double sums: when the
//    two smallest keys of any lane tie, a rarely taken branch compares the exact sums of all heads that share the
//    winner's value bits and rotates the smallest to the front.  The merged order is therefore the exact order (exact
//    ties in an arbitrary order, like argsort);
//  * rank()'s boundary element needs no division: frac * cont_weight = (g_ord[ig+1] - gdist_prev) * cont, the bin's weight
//    sum is g_ord[ig+1] - g_ord[ig] (the carry (1-frac) w, the weights inside and frac w telescope), and the share carried
//    into the next bin is (gdist - g_ord[ig+1]) * cont.  A closed bin is ONE 8-byte store (its un-normalised sum); the
//    record / resolve pass of k_ck_overlap and its 32 bytes per bin are gone.
//
// Preconditions (checked; otherwise the call runs on k_ck_overlap): every k(g) non-decreasing AND non-negative, G >= 2,
// del_g[0]^2 < del_g[0].
#pragma once
#include "ansfm_merge_common.hip.h"

namespace ansfm {

// LDS of one block (= one wave), dynamic LDS starting at address 0 (checked once per launch like k_ck_overlap):
//   [0, 256 nq)   column counters, nq = ceil(G/4): byte of row t at ((t >> 2) << 8) | (lane << 2) | (t & 3), i.e. four rows
//                 to a dword and the bank a function of the lane only.  The region starts at 0 and is < 2048 bytes, so the
//                 address is ((t * 65) & 0x703) | (lane << 2): two instructions
//   a[G][64], b[G+1][64] doubles
//   tables: g_ord[G + 2], bin widths[G] doubles, then the weights: (float)del_g[G] (W32) or del_g[G] doubles
// 22 688 bytes at G = 20: seven blocks per CU (the LDS allocation granule of gfx950 is 1280 bytes: 18 granules).
__host__ __device__ constexpr unsigned m32_col_bytes(int G) { return (unsigned)((G + 3) / 4) * kWave * 4u; }
__host__ __device__ constexpr unsigned m32_tab_bytes(int G, bool w32)
{
    return (unsigned)(G + 2) * 8u + (unsigned)G * 8u + (unsigned)G * (w32 ? 4u : 8u);
}
__host__ __device__ constexpr unsigned m32_lds_bytes(int G, bool w32)
{
    return m32_col_bytes(G) + (unsigned)(2 * G + 1) * kWave * 8u + ((m32_tab_bytes(G, w32) + 15u) & ~15u);
}

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) unsigned int lds_u32;
__device__ __forceinline__ unsigned lds_ld8(unsigned a) { return *(const lds_u8 *)(size_t)a; }
__device__ __forceinline__ void lds_st8(unsigned a, unsigned v) { *(lds_u8 *)(size_t)a = (unsigned char)v; }
__device__ __forceinline__ void lds_st32(unsigned a, unsigned v) { *(lds_u32 *)(size_t)a = v; }

struct M32Lane {            // per-lane LDS addresses
    unsigned a, b;          // &a[0][lane], &b[0][lane]
    unsigned lane4;         // lane << 2
    unsigned wtab;          // weight table: (float)del_g (W32) or del_g
};
// byte of row `tag` (< 32) in the column counters
__device__ __forceinline__ unsigned m32_col_addr(const M32Lane &L, unsigned tag)
{
    return ((tag * 65u) & 0x703u) | L.lane4;
}

template <bool W32> struct M32Elem;
template <> struct M32Elem<true> {
    double ai, bc, bn;
    float wr, wc;
    unsigned tag, c, caddr;
    __device__ __forceinline__ double weight() const { return (double)(wr * wc); }   // NumPy's float32 product
};
template <> struct M32Elem<false> {
    double ai, bc, bn;
    double wr, wc;
    unsigned tag, c, caddr;
    __device__ __forceinline__ double weight() const { return wr * wc; }
};

// operands of the head of row `tag` at column c (caddr = the address of the row's column counter)
template <bool W32>
__device__ __forceinline__ void m32_fetch(M32Elem<W32> &e, const M32Lane &L, unsigned tag, unsigned c, unsigned caddr)
{
    e.tag = tag;
    e.c = c;
    e.caddr = caddr;
    e.ai = lds_ld(L.a + (tag << 9));
    const unsigned ab = L.b + (c << 9);
    e.bc = lds_ld(ab);
    e.bn = lds_ld(ab + 512u);                   // b[G] = "huge": an exhausted row re-enters the list at its end
    if constexpr (W32) {
        e.wr = lds_ldf(L.wtab + (tag << 2));
        e.wc = lds_ldf(L.wtab + (c << 2));
    } else {
        e.wr = lds_ld(L.wtab + (tag << 3));
        e.wc = lds_ld(L.wtab + (c << 3));
    }
}

__device__ __forceinline__ unsigned m32_key(double v, unsigned tag)
{
    return (__float_as_uint((float)v) & ~31u) | tag;
}

// rank() walk, division-free (see the header).  gaddr = LDS address of g_ord[ig + 1]; roff = byte offset of this lane's
// slot in the row of bin ig.
struct M32Walk {
    double gd, kacc, gnext;
    unsigned roff, gaddr;
};

template <bool W32>
__device__ __forceinline__ double m32_walk(const M32Elem<W32> &e, M32Walk &ws, double *rec)
{
    const double cv = e.ai + e.bc;
    const double w = e.weight();
    const double gdn = ws.gd + w;
    double kn = fma(cv, w, ws.kacc);
    // ordered >= : g_ord[G+1] is NaN, nothing crosses after the last bin whatever the weights are
    if (gdn >= ws.gnext) {
        gst<double>(rec, ws.roff, fma(ws.gnext - ws.gd, cv, ws.kacc));
        kn = (gdn - ws.gnext) * cv;
        ws.roff += kWave * 8u;
        ws.gaddr += 8u;
        ws.gnext = lds_ld(ws.gaddr);
    }
    ws.kacc = kn;
    ws.gd = gdn;
    return cv;
}

// Heads whose float32 value bits equal the winner's: order them by their exact sums.  K[0..] is sorted; the group is a
// run from position 0.  The exact minimum moves to the front, the others keep their order.
template <int NR, bool W32>
__device__ __forceinline__ void m32_tie_fix(unsigned (&K)[NR], M32Elem<W32> &en, unsigned &c1, unsigned &c1addr,
                                            const M32Lane &L)
{
    const unsigned k0 = K[0];
    // a list whose smallest key is a sentinel / inf / NaN has nothing left to order
    const bool tied = ((k0 ^ K[1]) < 32u) && (k0 < 0x7F800000u);
    if (__builtin_amdgcn_ballot_w64(tied) == 0) return;
    auto exact = [&](unsigned key) -> double {
        const unsigned t = key & 31u;
        const unsigned c = lds_ld8(m32_col_addr(L, t));
        return lds_ld(L.a + (t << 9)) + lds_ld(L.b + (c << 9));
    };
    double best = 0.0;
    if (tied) best = exact(k0);
    unsigned pick = k0;
    int kstar = 0;
#pragma unroll
    for (int k = 1; k < NR; ++k) {
        const bool tk = tied && ((K[k] ^ k0) < 32u);
        if (__builtin_amdgcn_ballot_w64(tk) == 0) break;
        if (tk) {
            const double v = exact(K[k]);
            if (v < best) { best = v; pick = K[k]; kstar = k; }
        }
    }
    if (__builtin_amdgcn_ballot_w64(kstar != 0) != 0) {
#pragma unroll
        for (int j = NR - 1; j >= 1; --j) K[j] = (j <= kstar) ? K[j - 1] : K[j];
        K[0] = pick;
        const unsigned t0 = K[0] & 31u, a0 = m32_col_addr(L, t0);
        m32_fetch<W32>(en, L, t0, lds_ld8(a0), a0);
        c1addr = m32_col_addr(L, K[1] & 31u);
        c1 = lds_ld8(c1addr);
    }
}

// Fast pass: the two smallest keys tie in their value bits and sit in DIFFERENT columns -- compare their exact sums and swap
// them if the second is smaller.  (Same column: a_row decides, i.e. the key order is the exact order.)
template <int NR, bool W32>
__device__ __forceinline__ void m32_pair_fix(unsigned (&K)[NR], M32Elem<W32> &en, unsigned &c1, unsigned &c1addr,
                                             const M32Lane &L, bool fix)
{
    const double v0 = en.ai + en.bc;
    double v1 = v0;
    const unsigned t1 = K[1] & 31u;
    if (fix) v1 = lds_ld(L.a + (t1 << 9)) + lds_ld(L.b + (c1 << 9));
    const bool swap = fix && (v1 < v0);
    if (__builtin_amdgcn_ballot_w64(swap) != 0) {
        const unsigned k0 = K[0], k1 = K[1];
        const unsigned oc = en.c, ocaddr = en.caddr;
        if (swap) {
            K[0] = k1;
            K[1] = k0;
            m32_fetch<W32>(en, L, t1, c1, c1addr);
            c1 = oc;
            c1addr = ocaddr;
        }
    }
}

// One step: consume element e (operands fetched a step ago), insert its row's next element, fetch the new winner -> en.
// c1 / c1addr = column counter (and its address) of the list's second entry, fetched a step ahead.
// exact = false (fast pass): only the front PAIR is ordered exactly, and only when its columns differ; the consumed
// values are checked to be non-decreasing (descent) -- a sequence that is, is a sorted order of the G*G sums, whatever
// produced it.  exact = true (the rerun of a merge whose fast pass was not): every step pops the exact minimum.
template <int NR, bool W32>
__device__ __forceinline__ void m32_step(unsigned (&K)[NR], const M32Elem<W32> &e, M32Elem<W32> &en, unsigned &c1,
                                         unsigned &c1addr, M32Walk &ws, const M32Lane &L, double *rec, double &vprev,
                                         bool &descent, bool exact)
{
    const unsigned xk = m32_key(e.ai + e.bn, e.tag);
    const unsigned cn = e.c + 1u;
    lds_st8(e.caddr, cn);
    unsigned k0;
    asm("v_min_u32 %0, %1, %2" : "=v"(k0) : "v"(xk), "v"(K[1]));
    // the winner is either x (same row, next column) or the old second entry (column fetched a step ago)
    const bool isx = (k0 == xk);
    m32_fetch<W32>(en, L, k0 & 31u, isx ? cn : c1, isx ? e.caddr : c1addr);
    K[0] = k0;
    asm("v_med3_u32 %0, %1, %0, %2" : "+v"(K[1]) : "v"(xk), "v"(K[2]));
    c1addr = m32_col_addr(L, K[1] & 31u);
    c1 = lds_ld8(c1addr);                       // in order behind the store above: sees cn if x is now second
    __builtin_amdgcn_sched_barrier(0);          // everything above is issued before the rest of the pass and the walk
#pragma unroll
    for (int k = 2; k < NR - 1; ++k) asm("v_med3_u32 %0, %1, %0, %2" : "+v"(K[k]) : "v"(xk), "v"(K[k + 1]));
    asm("v_max_u32 %0, %1, %0" : "+v"(K[NR - 1]) : "v"(xk));
    const double cv = m32_walk<W32>(e, ws, rec);
    descent |= (cv < vprev);
    vprev = cv;
    const bool tie = ((K[0] ^ K[1]) < 32u);
    if (!exact) {
        const bool fix = tie && (en.c != c1);
        if (__builtin_amdgcn_ballot_w64(fix) != 0) m32_pair_fix<NR, W32>(K, en, c1, c1addr, L, fix);
    } else if (__builtin_amdgcn_ballot_w64(tie) != 0)
        m32_tie_fix<NR, W32>(K, en, c1, c1addr, L);
}

// Insert key x into the ascending list K, dropping its last entry (list set-up: row heads in any order).
template <int NR>
__device__ __forceinline__ void m32_insert(unsigned (&K)[NR], unsigned x)
{
#pragma unroll
    for (int k = NR - 1; k >= 1; --k) asm("v_med3_u32 %0, %1, %2, %0" : "+v"(K[k]) : "v"(x), "v"(K[k - 1]));
    asm("v_min_u32 %0, %1, %0" : "+v"(K[0]) : "v"(x));
}

template <int NR, bool FROM_K, bool W32>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_ck_overlap32(OverlapParams p)
{
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int G = p.G;
    char *base = reinterpret_cast<char *>(smem);
    double *A = reinterpret_cast<double *>(base + m32_col_bytes(G));
    double *B = A + G * kWave;                   // G + 1 rows
    char *tab = reinterpret_cast<char *>(B + (G + 1) * kWave);
    double *GORD = reinterpret_cast<double *>(tab);
    double *WID = GORD + (G + 2);
    double *DG = WID + G;                        // del_g (!W32)
    float *DGF = reinterpret_cast<float *>(WID + G);   // (float)del_g (W32): the same bytes, one of the two is used
    if (lds_addr(smem) != 0u) {                  // the column-counter addressing assumes it
        if (lane == 0) atomicOr(p.err_flag, 2);
        return;
    }
    if (lane < G) {
        if constexpr (W32) DGF[lane] = (float)p.del_g[lane];
        else DG[lane] = p.del_g[lane];
        WID[lane] = p.g_ord[lane + 1] - p.g_ord[lane];
    }
    if (lane < G + 2) GORD[lane] = p.g_ord[lane];
    const double HUGE_KEY = __longlong_as_double(0x7FE0000000000000LL);   // finite, float32 key = +inf
    B[G * kWave + lane] = HUGE_KEY;
    __syncthreads();
    M32Lane L;
    L.a = lds_addr(A + lane);
    L.b = lds_addr(B + lane);
    L.lane4 = (unsigned)lane * 4u;
    L.wtab = W32 ? lds_addr(reinterpret_cast<double *>(DGF)) : lds_addr(DG);
    const unsigned gord0 = lds_addr(GORD);
    const int nq = (G + 3) / 4;

    double *rec = p.scratch + (size_t)blockIdx.x * G * kWave;     // per block: un-normalised bin sums [G][64]
    TileQueue tq;
    tq.init();
    for (;;) {
        int vt = 0, m = 0, l = 0;
        if (!tq.next(p, lane, vt, m, l)) break;
        const int nu = vt * kWave + lane;
        LayerInterp q;
        if constexpr (!FROM_K) q = p.li[(size_t)m * p.L + l];
        bool unsorted = false;

        load_gas<FROM_K, true>(p, q, m, l, 0, nu, A, lane, unsorted);
        for (int s = 1; s < p.S; ++s) {
            load_gas<FROM_K, true>(p, q, m, l, s, nu, B, lane, unsorted);
            if (__builtin_amdgcn_ballot_w64(unsorted) != 0) break;     // the call is rerun on k_ck_overlap's generic path
            const double blast = B[(G - 1) * kWave + lane];
            const double alast = A[(G - 1) * kWave + lane];
            // skip rules, cutoff = 0  (ForwardModel_0.py:6073-6102)
            bool takeB, keepA;
            if (s == 1) { takeB = (alast <= 0.0); keepA = !takeB && (blast <= 0.0); }
            else { keepA = (blast <= 0.0); takeB = !keepA && (alast <= 0.0); }
            const bool do_merge = !(takeB | keepA);
            if (takeB) {
                for (int g = 0; g < G; ++g) A[g * kWave + lane] = B[g * kWave + lane];
            }
            if (do_merge) {
                unsigned K[NR];
                const double b0 = B[lane];
                M32Walk ws;
                bool exact = false;
                for (;;) {
#pragma unroll
                    for (int i = 0; i < NR; ++i)
                        K[i] = (i < G) ? m32_key(A[(i < G ? i : 0) * kWave + lane] + b0, (unsigned)i) : (0xFFFFFFE0u | (unsigned)i);
                    // a merged spectrum is non-decreasing only up to the rounding of its bin averages: when some lane's keys
                    // are not ascending, the heads are put in order one by one (the merge needs the columns ascending, no more)
                    bool bad = false;
#pragma unroll
                    for (int i = 0; i + 1 < NR; ++i) bad |= (K[i + 1] < K[i]);
                    if (__builtin_amdgcn_ballot_w64(bad) != 0) {
                        unsigned T[NR];
#pragma unroll
                        for (int i = 0; i < NR; ++i) { T[i] = K[i]; K[i] = 0xFFFFFFE0u | (unsigned)i; }
#pragma unroll
                        for (int i = 0; i < NR; ++i)
                            if (i < G) m32_insert<NR>(K, T[i]);
                    }
                    for (int qd = 0; qd < nq; ++qd) lds_st32(L.lane4 + (unsigned)qd * 256u, 0u);
                    M32Elem<W32> e0, e1;
                    m32_fetch<W32>(e0, L, K[0] & 31u, 0u, m32_col_addr(L, K[0] & 31u));
                    unsigned c1 = 0u, c1addr = m32_col_addr(L, K[1] & 31u);
                    if (exact) m32_tie_fix<NR, W32>(K, e0, c1, c1addr, L);
                    ws.gd = 0.0; ws.kacc = 0.0;
                    ws.gaddr = gord0 + 8u;
                    ws.gnext = lds_ld(ws.gaddr);
                    ws.roff = (unsigned)lane * 8u;
                    double vprev = 0.0;
                    bool descent = false;
                    const int nloop = G * G;
                    int it = 0;
                    for (; it + 1 < nloop; it += 2) {   // ping-pong: no register rotation
                        m32_step<NR, W32>(K, e0, e1, c1, c1addr, ws, L, rec, vprev, descent, exact);
                        m32_step<NR, W32>(K, e1, e0, c1, c1addr, ws, L, rec, vprev, descent, exact);
                    }
                    if (it < nloop) m32_step<NR, W32>(K, e0, e1, c1, c1addr, ws, L, rec, vprev, descent, exact);
                    // the fast pass consumed the sums in non-decreasing order in every lane: it was a sorted order
                    if (exact || __builtin_amdgcn_ballot_w64(descent) == 0) break;
                    exact = true;
                    if (lane == 0) atomicAdd(p.err_flag + 12, 1);      // statistics: merges rerun (ansfm_merge_redo_count)
                }
                // ---- normalise: closed bins by their width; the open one as rank()'s trailing `if ig == ng-1` (:6171) ----
                const int ig = (int)((ws.gaddr - (gord0 + 8u)) >> 3);
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
                            if (b < ig) outv = fast_div(r[k], WID[b]);
                            else if (b == ig) outv = (b == G - 1) ? fast_div(ws.kacc, ws.gd - GORD[G - 1]) : ws.kacc;
                            A[b * kWave + lane] = outv;
                        }
                    }
                }
            }
        }
        double *out = p.tau + (((size_t)m * p.L + l) * G) * p.Wpad + nu;
        if (unsorted) atomicOr(p.err_flag, 1);
        for (int g = 0; g < G; ++g) out[(size_t)g * p.Wpad] = A[g * kWave + lane];
    }
}

}  // namespace ansfm
