// ansfm_layer_kernels.hip.h -- Layer_0.layer_average on gfx950, batched over atmospheric states.
//
// Restates archnemesis/Layer_0.py:755-1030 (MID_PATH and the Curtis-Godson ABSORBER_WEIGHTED_AVERAGE branch
// :949-1010): slant-path sub-division of every layer into NINT points, linear interpolation (with
// extrapolation) of the profile in height, n = p/(k_B T), Simpson integrals of n, h n, p n, T n, f n, vmr n,
// vmr p n and dust (scipy.integrate.simpson incl. its even-N end correction), then the scaling back to vertical
// columns (:1013-1023).
// One workgroup per (state, layer): phase 1 = one thread per sub-point (geometry, bracket, p, T, n), phase 2 =
// one thread per integrated quantity (scipy's unequal-spacing Simpson sum, sequential like np.sum's order to
// rounding).  A numerical Jacobian re-derives the layers for every perturbed state (jacobian_nemesis ->
// nemesisfm -> calc_path, ForwardModel_0.py:516); this kernel does all of them in one launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ansfm {

constexpr int kLayMaxNint = 256;
constexpr int kLayPairTerms = 2048;   // Simpson terms (quantity x pair of intervals) a workgroup forms side by side in LDS

struct LayerAvgParams {
    const double *H, *P, *T;        // [n][NPRO]
    const double *VMR;              // [n][NPRO][NVMR]
    const double *DUST;             // [n][NPRO][NDUST] or nullptr
    const double *PARAH2;           // [n][NPRO] or nullptr (zeros)
    const double *XMOLWT;           // [n][NPRO] kg/mol or nullptr
    const double *BASEH;            // [n][NLAY]
    const int32_t *dust_units;      // [NDUST] or nullptr
    double *HEIGHT, *PRESS, *TEMP, *TOTAM, *FRAC, *DELH, *BASET, *LAYSF;   // [n][NLAY]
    double *AMOUNT, *PP;            // [n][NLAY][NVMR]
    double *CONT;                   // [n][NLAY][NDUST]
    double RADIUS, LAYANG, LAYHT;
    int n_models, NPRO, NVMR, NDUST, NLAY, LAYINT, NINT;
    // layer_averageg (Layer_0.py:1032-1398): T / PARAH2 / VMR / DUST through the reference's `interpg` and the
    // matrices DTE, DAM, DCO, DPH [n][NLAY][NPRO] (zeroed by the caller)
    int with_grad, any_dust_units;
    double *DTE, *DAM, *DCO, *DPH;
    // A numerical Jacobian's states differ from state 0 at one profile level or two: m0 = first state of this launch; a layer
    // whose sub-points read only levels at which the state holds state 0's very numbers takes state 0's results instead of
    // integrating again (k_layer_share, after state 0's own launch, without gradients) -- same bits either way
    int m0;
    const unsigned char *share;   // [n][NLAY] or nullptr: 1 = k_layer_share copied state 0's layer, nothing to do
};

__device__ __forceinline__ int lay_bracket(const double *x, int n, double xn);

// One thread per (state m >= 1, layer): the sub-points of a layer lie between its two ends along the ray, and their heights --
// the same expressions as k_layer_average's phase 1, every operation monotone in s -- between the heights of those ends: the
// layer reads the levels bracket(h(S0)) - 1 .. bracket(h(S1)), those of its base (BASET), and the top level (SMAX).  If the
// state holds state 0's numbers there (bit for bit: H, P, T, para-H2, every gas, the dust, the molecular weight; the layer's
// bases), its results are state 0's: copy them and flag the layer.
__global__ void k_layer_share(LayerAvgParams p, unsigned char *flag)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int NL = p.NLAY, NPRO = p.NPRO, V = p.NVMR, D = p.NDUST;
    if (idx >= (p.n_models - 1) * NL) return;
    const int m = 1 + idx / NL, I = idx % NL;
    const double PI = 3.141592653589793;
    const double *H = p.H + (size_t)m * NPRO, *BASEH = p.BASEH + (size_t)m * NL;
    auto ne = [](double a, double b) { return __double_as_longlong(a) != __double_as_longlong(b); };
    bool differs = ne(H[NPRO - 1], p.H[NPRO - 1]) || ne(BASEH[I], p.BASEH[I]) || (I < NL - 1 && ne(BASEH[I + 1], p.BASEH[I + 1]));
    if (!differs) {
        const double sn = sin(p.LAYANG * PI / 180), cs = cos(p.LAYANG * PI / 180);
        const double z0 = p.RADIUS + p.LAYHT, zmax = p.RADIUS + H[NPRO - 1];
        const double SMAX = sqrt(zmax * zmax - (z0 * sn) * (z0 * sn)) - z0 * cs;
        auto bases = [&](int i) { const double r = p.RADIUS + BASEH[i]; return sqrt(r * r - (z0 * sn) * (z0 * sn)) - z0 * cs; };
        const double S0 = bases(I), S1 = (I < NL - 1) ? bases(I + 1) : SMAX;
        double sa = S0, sb = S1;
        if (p.LAYINT == 0) sa = sb = (I < NL - 1) ? (bases(I + 1) + S0) / 2 : (SMAX + S0) / 2;
        const double ha = sqrt(sa * sa + z0 * z0 + 2 * sa * z0 * cs) - p.RADIUS, hb = sqrt(sb * sb + z0 * z0 + 2 * sb * z0 * cs) - p.RADIUS;
        const int ia = lay_bracket(H, NPRO, ha), ib = lay_bracket(H, NPRO, hb), ic = lay_bracket(H, NPRO, BASEH[I]);
        const int lo = min(min(ia, ib), ic) - 1, hi = max(max(ia, ib), ic);
        const bool xm = p.dust_units && p.XMOLWT;
        for (int lev = lo; lev <= hi && !differs; ++lev) {
            const size_t a = (size_t)m * NPRO + lev;
            differs = ne(p.H[a], p.H[lev]) || ne(p.P[a], p.P[lev]) || ne(p.T[a], p.T[lev]);
            if (p.PARAH2) differs = differs || ne(p.PARAH2[a], p.PARAH2[lev]);
            if (xm) differs = differs || ne(p.XMOLWT[a], p.XMOLWT[lev]);
            for (int j = 0; j < V && !differs; ++j) differs = ne(p.VMR[a * V + j], p.VMR[(size_t)lev * V + j]);
            for (int j = 0; j < D && !differs; ++j) differs = ne(p.DUST[a * D + j], p.DUST[(size_t)lev * D + j]);
        }
    }
    const size_t o = (size_t)m * NL + I, b = I;
    flag[o] = differs ? 0 : 1;
    if (differs) return;
    p.HEIGHT[o] = p.HEIGHT[b]; p.PRESS[o] = p.PRESS[b]; p.TEMP[o] = p.TEMP[b]; p.TOTAM[o] = p.TOTAM[b];
    p.FRAC[o] = p.FRAC[b]; p.DELH[o] = p.DELH[b]; p.BASET[o] = p.BASET[b]; p.LAYSF[o] = p.LAYSF[b];
    for (int J = 0; J < V; ++J) { p.AMOUNT[o * V + J] = p.AMOUNT[b * V + J]; p.PP[o * V + J] = p.PP[b * V + J]; }
    for (int J = 0; J < D; ++J) p.CONT[o * D + J] = p.CONT[b * D + J];
}

// Layer_0.interpg (:716-751): j = clip(#{x <= X}, 1, n-1)
__device__ __forceinline__ int lay_bracket_g(const double *x, int n, double xn)
{
    int lo = 0, hi = n;                       // first index with x > xn
    while (lo < hi) { int mid = (lo + hi) >> 1; if (x[mid] <= xn) lo = mid + 1; else hi = mid; }
    return lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
}
__device__ __forceinline__ double lay_interp_g(const double *y, int stride, int j, double F)
{
    return (1.0 - F) * y[(size_t)(j - 1) * stride] + F * y[(size_t)j * stride];
}

__device__ __forceinline__ int lay_bracket(const double *x, int n, double xn)
{   // scipy interp1d linear: idx = clip(searchsorted(x, xn, 'left'), 1, n-1)
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (x[mid] < xn) lo = mid + 1; else hi = mid; }
    return lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
}
__device__ __forceinline__ double lay_interp(const double *x, const double *y, int stride, int idx, double xn)
{
    const double ylo = y[(size_t)(idx - 1) * stride], yhi = y[(size_t)idx * stride];
    const double slope = (yhi - ylo) / (x[idx] - x[idx - 1]);
    return slope * (xn - x[idx - 1]) + ylo;
}

__global__ __launch_bounds__(256) void k_layer_average(LayerAvgParams p)
{
    __shared__ double S[kLayMaxNint], hh[kLayMaxNint], pp[kLayMaxNint], duds[kLayMaxNint], mw[kLayMaxNint];
    __shared__ int idx[kLayMaxNint], jg[kLayMaxNint];
    __shared__ double FF[kLayMaxNint];
    __shared__ double res[160];
    __shared__ double cA[kLayMaxNint / 2], cB0[kLayMaxNint / 2], cB1[kLayMaxNint / 2], cB2[kLayMaxNint / 2];
    __shared__ double IN[kLayPairTerms];
    const bool GR = p.with_grad != 0;
    const double k_B = 1.38065e-23, AVOGAD = 6.02214076e23, PI = 3.141592653589793;
    const int I = blockIdx.x, m = blockIdx.y + p.m0, tid = threadIdx.x;
    const int NPRO = p.NPRO, V = p.NVMR, D = p.NDUST, NL = p.NLAY;
    const double *H = p.H + (size_t)m * NPRO, *P = p.P + (size_t)m * NPRO, *T = p.T + (size_t)m * NPRO;
    const double *VMR = p.VMR + (size_t)m * NPRO * V;
    const double *DUST = p.DUST ? p.DUST + (size_t)m * NPRO * D : nullptr;
    const double *PH2 = p.PARAH2 ? p.PARAH2 + (size_t)m * NPRO : nullptr;
    const double *XM = (p.dust_units && p.XMOLWT) ? p.XMOLWT + (size_t)m * NPRO : nullptr;
    const double *BASEH = p.BASEH + (size_t)m * NL;
    const double sn = sin(p.LAYANG * PI / 180), cs = cos(p.LAYANG * PI / 180);
    const double z0 = p.RADIUS + p.LAYHT, zmax = p.RADIUS + H[NPRO - 1];
    const double SMAX = sqrt(zmax * zmax - (z0 * sn) * (z0 * sn)) - z0 * cs;
    auto bases = [&](int i) { const double r = p.RADIUS + BASEH[i]; return sqrt(r * r - (z0 * sn) * (z0 * sn)) - z0 * cs; };
    const double S0 = bases(I), S1 = (I < NL - 1) ? bases(I + 1) : SMAX;
    const double DELS = S1 - S0;
    const double DELH = (I < NL - 1) ? BASEH[I + 1] - BASEH[I] : H[NPRO - 1] - BASEH[NL - 1];
    const double LAYSF = DELS / DELH;
    const int npts = (p.LAYINT == 0) ? 1 : p.NINT;
    if (p.share != nullptr && p.share[(size_t)m * NL + I]) return;     // k_layer_share took state 0's results for this layer
    // ---- phase 1: sub-points -------------------------------------------------------------------------
    for (int k = tid; k < npts; k += blockDim.x) {
        double s;
        if (p.LAYINT == 0) s = (I < NL - 1) ? (bases(I + 1) + S0) / 2 : (SMAX + S0) / 2;      // :899-901
        else { const double step = (S1 - S0) / (p.NINT - 1); s = (k == p.NINT - 1) ? S1 : k * step + S0; }   // np.linspace
        const double h = sqrt(s * s + z0 * z0 + 2 * s * z0 * cs) - p.RADIUS;
        const int ix = lay_bracket(H, NPRO, h);
        const double pk = lay_interp(H, P, 1, ix, h);
        double tk = lay_interp(H, T, 1, ix, h);
        if (GR) {
            const int j = lay_bracket_g(H, NPRO, h);
            const double F = (h - H[j - 1]) / (H[j] - H[j - 1]);
            jg[k] = j; FF[k] = F;
            tk = lay_interp_g(T, 1, j, F);
        }
        S[k] = s; hh[k] = h; idx[k] = ix; pp[k] = pk;
        duds[k] = pk / (k_B * tk);
        mw[k] = XM ? lay_interp(H, XM, 1, ix, h) * 1000. : 0.0;       // XMOLWT *= 1000 :877
    }
    __syncthreads();
    // ---- phase 2: one thread per integrated quantity ----------------------------------------------------
    const int NQ = 5 + 2 * V + D;
    auto yval_q = [&](int q, int k) -> double {
            const double h = hh[k];
            const int ix = idx[k];
            if (q == 0) return duds[k];
            if (q == 1) return h * duds[k];
            if (q == 2) return pp[k] * duds[k];
            if (GR) {
                const int j = jg[k];
                const double F = FF[k];
                if (q == 3) return lay_interp_g(T, 1, j, F) * duds[k];
                if (q == 4) return (PH2 ? lay_interp_g(PH2, 1, j, F) : 0.0) * duds[k];
                if (q < 5 + V) return lay_interp_g(VMR + (q - 5), V, j, F) * duds[k];
                if (q < 5 + 2 * V) return (lay_interp_g(VMR + (q - 5 - V), V, j, F) * pp[k]) * duds[k];
                const int J = q - 5 - 2 * V;
                const double dd = (p.LAYINT == 0) ? lay_interp(H, DUST + J, D, ix, h)      // MID_PATH: interp1d (:1231)
                                                  : lay_interp_g(DUST + J, D, j, F);
                return (p.dust_units && p.dust_units[J] == -1) ? dd * duds[k] * mw[k] / AVOGAD : dd;
            }
            if (q == 3) return lay_interp(H, T, 1, ix, h) * duds[k];
            if (q == 4) return (PH2 ? lay_interp(H, PH2, 1, ix, h) : 0.0) * duds[k];
            if (q < 5 + V) return lay_interp(H, VMR + (q - 5), V, ix, h) * duds[k];
            if (q < 5 + 2 * V) return (lay_interp(H, VMR + (q - 5 - V), V, ix, h) * pp[k]) * duds[k];
            const int J = q - 5 - 2 * V;
            const double dd = lay_interp(H, DUST + J, D, ix, h);
            return (p.dust_units && p.dust_units[J] == -1) ? dd * duds[k] * mw[k] / AVOGAD : dd;
    };
    // scipy _basic_simpson, unequal-spacing form: the weights of a pair of intervals once per workgroup, the bracket of every
    // (quantity, pair) side by side (the profile reads are a microsecond each from a cold cache: one thread walking the 50
    // pairs of a quantity took 0.2 ms per layer), then one thread per quantity adds its terms in order
    const int nodd_ = (npts & 1) ? npts : npts - 1;
    const int npairs = (p.LAYINT != 0 && npts > 2) ? (nodd_ - 1) / 2 : 0;
    const bool side_by_side = npairs > 0 && NQ * npairs <= kLayPairTerms;
    if (side_by_side) {
        for (int e = tid; e < npairs; e += blockDim.x) {
            const int i = 2 * e;
            const double h0 = S[i + 1] - S[i], h1 = S[i + 2] - S[i + 1];
            const double hsum = h0 + h1, hprod = h0 * h1;
            const double h0divh1 = (h1 != 0) ? h0 / h1 : 0.0;
            const double inv = (h0divh1 != 0) ? 1.0 / h0divh1 : 0.0;
            const double hq = (hprod != 0) ? hsum / hprod : 0.0;
            cA[e] = hsum / 6.0; cB0[e] = 2.0 - inv; cB1[e] = hsum * hq; cB2[e] = 2.0 - h0divh1;
        }
        __syncthreads();
        for (int e = tid; e < NQ * npairs; e += blockDim.x) {
            const int q = e / npairs, ip = e - q * npairs, i = 2 * ip;
            IN[e] = yval_q(q, i) * cB0[ip] + yval_q(q, i + 1) * cB1[ip] + yval_q(q, i + 2) * cB2[ip];
        }
        __syncthreads();
    }
    for (int q = tid; q < NQ; q += blockDim.x) {
        auto yval = [&](int k) -> double { return yval_q(q, k); };
        double r = 0.0;
        if (p.LAYINT == 0) {
            r = yval(0);   // point value; combined below
        } else if (npts == 2) {
            r = 0.5 * (S[1] - S[0]) * (yval(1) + yval(0));     // scipy simpson with two points
        } else {
            const int nodd = (npts & 1) ? npts : npts - 1;
            if (side_by_side)
                for (int ip = 0; ip < npairs; ++ip) r += cA[ip] * IN[q * npairs + ip];
            else
            for (int i = 0; i + 2 < nodd; i += 2) {   // (more terms than the LDS array holds: one thread per quantity all the way)
                const double h0 = S[i + 1] - S[i], h1 = S[i + 2] - S[i + 1];
                const double hsum = h0 + h1, hprod = h0 * h1;
                const double h0divh1 = (h1 != 0) ? h0 / h1 : 0.0;
                const double inv = (h0divh1 != 0) ? 1.0 / h0divh1 : 0.0;
                const double hq = (hprod != 0) ? hsum / hprod : 0.0;
                r += hsum / 6.0 * (yval(i) * (2.0 - inv) + yval(i + 1) * (hsum * hq) + yval(i + 2) * (2.0 - h0divh1));
            }
            if (!(npts & 1)) {   // even number of points: simpson()'s correction for the last interval
                const double h0 = S[npts - 2] - S[npts - 3], h1 = S[npts - 1] - S[npts - 2];
                double num = 2 * (h1 * h1) + 3 * h0 * h1, den = 6 * (h1 + h0);
                const double alpha = (den != 0) ? num / den : 0.0;
                num = h1 * h1 + 3.0 * h0 * h1; den = 6 * h0;
                const double beta = (den != 0) ? num / den : 0.0;
                num = 1 * pow(h1, 3.0); den = 6 * h0 * (h0 + h1);
                const double eta = (den != 0) ? num / den : 0.0;
                r += alpha * yval(npts - 1) + beta * yval(npts - 2) - eta * yval(npts - 3);
            }
        }
        res[q] = r;
    }
    __syncthreads();
    // ---- combine -----------------------------------------------------------------------------------------
    const size_t o = (size_t)m * NL + I;
    if (p.LAYINT == 0) {
        const double DUDS = res[0], TOT = DUDS * DELS;
        const double hmid = hh[0], PR = pp[0];
        const int ix = idx[0];
        if (tid == 0) {
            p.HEIGHT[o] = hmid; p.PRESS[o] = PR;
            p.TEMP[o] = GR ? lay_interp_g(T, 1, jg[0], FF[0]) : lay_interp(H, T, 1, ix, hmid);
            p.FRAC[o] = PH2 ? (GR ? lay_interp_g(PH2, 1, jg[0], FF[0]) : lay_interp(H, PH2, 1, ix, hmid)) : 0.0;
            p.TOTAM[o] = TOT / LAYSF;
            if (GR) {   // :1209-1221, :1252-1254, then / LAYSF (:1388-1390)
                const int j = jg[0];
                const double F = FF[0];
                double *dte = p.DTE + o * NPRO, *dam = p.DAM + o * NPRO, *dco = p.DCO + o * NPRO, *dph = p.DPH + o * NPRO;
                dte[j - 1] += (1.0 - F); dte[j] += F;
                dph[j - 1] += (1.0 - F); dph[j] += F;
                dam[j - 1] = ((1.0 - F) * TOT) / LAYSF; dam[j] = (F * TOT) / LAYSF;
                if (D > 0) { dco[j - 1] = (1.0 - F) / LAYSF; dco[j] = F / LAYSF; }
            }
            p.DELH[o] = DELH; p.LAYSF[o] = LAYSF; p.BASET[o] = lay_interp(H, T, 1, lay_bracket(H, NPRO, BASEH[I]), BASEH[I]);
        }
        for (int J = tid; J < V; J += blockDim.x) {
            const double a = GR ? lay_interp_g(VMR + J, V, jg[0], FF[0]) : lay_interp(H, VMR + J, V, ix, hmid);
            p.PP[o * V + J] = a * PR;
            p.AMOUNT[o * V + J] = (a * TOT) * pow(LAYSF, -1.0);
        }
        for (int J = tid; J < D; J += blockDim.x) {
            const double dd = lay_interp(H, DUST + J, D, ix, hmid);
            const double c = (p.dust_units && p.dust_units[J] == -1) ? dd * TOT * mw[0] / AVOGAD : dd * DELS;
            p.CONT[o * D + J] = c * pow(LAYSF, -1.0);
        }
    } else {
        const double TOT = res[0];
        if (tid == 0) {
            p.TOTAM[o] = TOT / LAYSF;
            p.HEIGHT[o] = res[1] / TOT; p.PRESS[o] = res[2] / TOT; p.TEMP[o] = res[3] / TOT; p.FRAC[o] = res[4] / TOT;
            p.DELH[o] = DELH; p.LAYSF[o] = LAYSF; p.BASET[o] = lay_interp(H, T, 1, lay_bracket(H, NPRO, BASEH[I]), BASEH[I]);
        }
        for (int J = tid; J < V; J += blockDim.x) {
            p.AMOUNT[o * V + J] = res[5 + J] * pow(LAYSF, -1.0);
            p.PP[o * V + J] = res[5 + V + J] / TOT;
        }
        for (int J = tid; J < D; J += blockDim.x) p.CONT[o * D + J] = res[5 + 2 * V + J] * pow(LAYSF, -1.0);
        if (GR && tid < 4) {
            // :1280-1284, :1303-1305, :1336-1343 -- sequential over the sub-points like the reference, one thread per
            // matrix (0 DTE, 1 DPH, 2 DAM, 3 DCO), then the scaling of :1346-1350 and the / LAYSF of :1388-1390
            if (tid == 3 && D == 0) return;
            double *row = (tid == 0 ? p.DTE : tid == 1 ? p.DPH : tid == 2 ? p.DAM : p.DCO) + o * NPRO;
            for (int k = 0; k < npts; ++k) {
                const double w = (k == 0 || k == npts - 1) ? 1.0 : ((k & 1) ? 4.0 : 2.0);
                const int j = jg[k];
                const double F = FF[k];
                double lo, hi;
                if (tid <= 1) { lo = (1. - F) * w * duds[k]; hi = F * w * duds[k]; }
                else if (tid == 2) { lo = (1. - F) * duds[k] * w; hi = F * duds[k] * w; }
                else if (!p.any_dust_units) { lo = (1. - F) * w; hi = F * w; }
                else { lo = (1. - F) * w * duds[k] * mw[k] / AVOGAD; hi = F * w * duds[k] * mw[k] / AVOGAD; }
                row[j - 1] += lo; row[j] += hi;
            }
            for (int k = 0; k < NPRO; ++k) {
                double v = row[k] * DELS / (p.NINT - 1.) / 3.;
                if (tid <= 1) v = v / TOT; else v = v / LAYSF;
                row[k] = v;
            }
        }
    }
}

}  // namespace ansfm
