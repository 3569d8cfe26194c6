// ansfm_merge_common.hip.h -- device-side pieces shared by the merge kernels' translation units (ansfm_api.hip,
// ansfm_merge32.hip): the ln-k table encoding and (P,T) interpolation of calc_k (Spectroscopy_0.py:2298-2437), the
// parameter block, the table reads of one gas, LDS / global access helpers and the per-XCD tile queues.  Templates
// and inline device functions only -- no kernels -- so that it can be included from more than one translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ansfm {

constexpr int kWave = 64;
constexpr int kMaxG = 32;

// ------------------------------------------------------------------------------------------------
// ln-k table encoding.  k > 0  -> ln k (finite double)
//                       k <= 0 -> quiet NaN whose 51 payload bits are the top 51 bits of k
// (sign, exponent, 39 mantissa bits: exact for tables that were float32 on disk).  The
// good/bad/mixed corner logic of calc_k (Spectroscopy_0.py:2391-2403) needs the sign and, for the
// all-non-positive "bad" branch, the raw value.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double encode_lnk(double k)
{
    if (k > 0.0) return log(k);
    unsigned long long b = (unsigned long long)__double_as_longlong(k);
    unsigned long long box = 0x7FF8000000000000ULL | (b >> 13);
    return __longlong_as_double((long long)box);
}
__device__ __forceinline__ bool lnk_is_boxed(double x) { return x != x; }
__device__ __forceinline__ double lnk_unbox(double x)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return __longlong_as_double((long long)((b & 0x0007FFFFFFFFFFFFULL) << 13));
}

// Per (model, layer) interpolation constants: the nearest-then-bracket corner choice of calc_k
// (Spectroscopy_0.py:2336-2389) is wave-uniform, so it is computed once per layer.
struct LayerInterp {
    int ipl, iph, itl, ith;
    double v, u, dudt;
};


// k for one (corner set, u, v): Spectroscopy_0.py:2391-2403 (+ dk/dT :2241-2247 when wanted)
__device__ __forceinline__ double interp_k(double l1, double l2, double h1, double h2, double v,
                                           double u)
{
    // l1 = (ip_low,it_low)  l2 = (ip_low,it_high)  h1 = (ip_high,it_low)  h2 = (ip_high,it_high)
    bool b1 = lnk_is_boxed(l1), b2 = lnk_is_boxed(l2), b3 = lnk_is_boxed(h1), b4 = lnk_is_boxed(h2);
    double kk = 0.0;
    if (!(b1 | b2 | b3 | b4)) {
        double x = (1.0 - v) * (1.0 - u) * l1 + v * (1.0 - u) * h1 + v * u * h2 + (1.0 - v) * u * l2;
        kk = exp(x);
    } else if (b1 & b2 & b3 & b4) {
        double klo1 = lnk_unbox(l1), klo2 = lnk_unbox(l2), khi1 = lnk_unbox(h1), khi2 = lnk_unbox(h2);
        kk = (1.0 - v) * (1.0 - u) * klo1 + v * (1.0 - u) * khi1 + v * u * khi2 + (1.0 - v) * u * klo2;
    }
    return kk;
}
__device__ __forceinline__ void interp_kg(double l1, double l2, double h1, double h2, double v,
                                          double u, double dudt, double &kk, double &dk)
{
    bool b1 = lnk_is_boxed(l1), b2 = lnk_is_boxed(l2), b3 = lnk_is_boxed(h1), b4 = lnk_is_boxed(h2);
    kk = 0.0; dk = 0.0;
    if (!(b1 | b2 | b3 | b4)) {
        double x = (1.0 - v) * (1.0 - u) * l1 + v * (1.0 - u) * h1 + v * u * h2 + (1.0 - v) * u * l2;
        kk = exp(x);
        double dxdt = (-l1 * (1.0 - v) - h1 * v + h2 * v + l2 * (1.0 - v)) * dudt;
        dk = kk * dxdt;
    } else if (b1 & b2 & b3 & b4) {
        double klo1 = lnk_unbox(l1), klo2 = lnk_unbox(l2), khi1 = lnk_unbox(h1), khi2 = lnk_unbox(h2);
        kk = (1.0 - v) * (1.0 - u) * klo1 + v * (1.0 - u) * khi1 + v * u * khi2 + (1.0 - v) * u * klo2;
        dk = (-klo1 * (1.0 - v) - khi1 * v + khi2 * v + klo2 * (1.0 - v)) * dudt;
    }
}


struct OverlapParams {
    const double *lnK;        // [NP][NT][S][G][Wpad]            (FROM_K: unused)
    const double *kin;        // FROM_K: k[S][L][G][Wpad] (array-level k_overlap seam)
    const LayerInterp *li;    // [n][L]
    const double *amount;     // [n][S][L]
    const double *del_g;      // [G]
    double *tau;              // [n][L][G][Wpad]
    double *scratch;          // [gridDim.x][2][G][64]
    int *err_flag;            // bit0: unsorted input k-distribution
    unsigned int *tile_counter;  // [8] dynamic tile queues, one per XCD (zeroed before every launch)
    int W, Wpad, G, NT, S, L, n_models;
    int delg_f32;             // DELG is a float32 array: del_g[i]*del_g[j] is a float32 product
    double g_ord[kMaxG + 2];  // [0, cumsum(del_g)] (float32 cumsum when delg_f32), g_ord[G]=1, NaN
};


// Table reads of one gas for one (64-wavenumber, layer) tile.  The loads of kLoadBatch g-ordinates (4 corner
// rows each) are all issued before the first value is used: one memory round trip per batch instead of one per
// g-ordinate (a wave has at most one sibling on its SIMD to hide it behind).  Indices are clamped, not
// predicated, so the batch stays branch-free.
constexpr int kLoadBatch = 10;

// NONNEG (the 32-bit-key merge kernel, ansfm_merge32.hip.h): a negative value counts as "unsorted" too (that kernel
// orders float32 bit patterns as unsigned integers) and -0.0 is stored as +0.0.
template <bool FROM_K, bool NONNEG = false>
__device__ __forceinline__ void load_gas(const OverlapParams &p, const LayerInterp &q, int m, int l,
                                         int s, int nu, double *DST, int lane, bool &unsorted)
{
    const int G = p.G;
    const double amt = p.amount[((size_t)m * p.S + s) * p.L + l];
    double prev = NONNEG ? 0.0 : -__builtin_inf();
    if constexpr (FROM_K) {
        const double *src = p.kin + (((size_t)s * p.L + l) * G) * p.Wpad + nu;
        for (int g0 = 0; g0 < G; g0 += kLoadBatch) {
            double r[kLoadBatch];
#pragma unroll
            for (int k = 0; k < kLoadBatch; ++k) {
                const int gi = (g0 + k < G) ? g0 + k : G - 1;
                r[k] = src[(size_t)gi * p.Wpad];
            }
#pragma unroll
            for (int k = 0; k < kLoadBatch; ++k)
                if (g0 + k < G) {
                    double kk = r[k] * amt;
                    if constexpr (NONNEG) kk += 0.0;
                    DST[(g0 + k) * kWave + lane] = kk;
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
                // streamed once per tile: non-temporal so the table does not push the merge scratch out of L2
                r1[k] = __builtin_nontemporal_load(c1 + go);
                r2[k] = __builtin_nontemporal_load(c2 + go);
                r3[k] = __builtin_nontemporal_load(c3 + go);
                r4[k] = __builtin_nontemporal_load(c4 + go);
            }
#pragma unroll
            for (int k = 0; k < kLoadBatch; ++k)
                if (g0 + k < G) {
                    double kk = interp_k(r1[k], r2[k], r3[k], r4[k], q.v, q.u) * amt;
                    if constexpr (NONNEG) kk += 0.0;
                    DST[(g0 + k) * kWave + lane] = kk;
                    unsorted |= (kk < prev);
                    prev = kk;
                }
        }
    }
}

__device__ __forceinline__ double fast_div(double n, double d)
{   // n/d with v_rcp_f64 + 2 Newton steps + residual correction (<= ~1 ulp; frac of rank()).  One Newton step gives the
    // same quotients (tools/calib/div_check.hip) but k_ck_overlap measured 1.2 % SLOWER with it (5.88 -> 5.95 ms, same box,
    // twice): the second step's two instructions fill issue slots the resolve otherwise leaves empty
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double q = n * r;
    return fma(fma(-d, q, n), r, q);
}


typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ unsigned lds_addr(const double *p) { return (unsigned)(size_t)(const lds_double *)p; }
__device__ __forceinline__ double lds_ld(unsigned a) { return *(const lds_double *)(size_t)a; }
__device__ __forceinline__ void lds_st(unsigned a, double v) { *(lds_double *)(size_t)a = v; }
// Global access as wave-uniform base + 32-bit byte offset: the compiler emits `global_load/store v_off, s[base]` and the
// per-access address arithmetic stays 32-bit (64-bit pointer adds are multi-pass VALU instructions).
template <class T> __device__ __forceinline__ T gld(const void *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <class T> __device__ __forceinline__ void gst(void *base, unsigned byte_off, T v)
{
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ float lds_ldf(unsigned a) { return *(const lds_float *)(size_t)a; }

// Dynamic tile queues.  With 5 resident waves per CU one SIMD hosts two waves that run slower than the solo
// ones; static striding would make the launch wait for them.  One relaxed atomic per tile (~2800*7 merge steps
// of work) -- every wave exits when the counters pass the tile counts.
// Eight queues, queue q = wavenumber tiles vt with vt % 8 == q (layer fastest): the layers of one wavenumber
// tile share k-table corner rows, so they are kept on one XCD's L2.  A wave starts on the queue of the XCD it
// runs on (HW_REG_XCC_ID; affinity only, any placement is correct) and steals from the others when its own
// is empty.
struct TileQueue {
    int myq, qoff;
    __device__ __forceinline__ void init()
    {
        myq = (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7);
        qoff = 0;
    }
    __device__ __forceinline__ bool next(const OverlapParams &p, int lane, int &vt, int &m, int &l)
    {
        const int NVT = p.Wpad / kWave;
        while (qoff < 8) {
            const int qq = (myq + qoff) & 7;
            const int nvt_q = (NVT - qq + 7) / 8;                     // tiles vt = qq, qq+8, ...
            const long nq = (long)p.n_models * nvt_q * p.L;
            unsigned int tq = 0;
            if (lane == 0 && nq > 0) tq = atomicAdd(p.tile_counter + qq, 1u);
            const long t = (long)__builtin_amdgcn_readfirstlane(tq);
            if (nq > 0 && t < nq) {
                l = (int)(t % p.L);
                const long r = t / p.L;
                vt = qq + 8 * (int)(r % nvt_q);
                m = (int)(r / nvt_q);
                return true;
            }
            ++qoff;
        }
        return false;
    }
};


}  // namespace ansfm
