// ansfm_ms_kernels.hip.h -- doubling/adding multiple-scattering core on gfx950.
//
// Restates Multiple_Scattering_Core.scloud11wave_core (Multiple_Scattering_Core.py:651-960) and its
// callees phasint2 :141, hansen :200, add :275, double1 :321, addp :481, angle_quadrature :535,
// calc_rtj_matrix :566, for the look-down geometry.
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
//   k_ms_fourier one thread per (wave, g, path): the Fourier sum with the reference's early-out
//                (:949-958) -> rad[path][g][wave].
// First correct version: VALU matmuls out of LDS.  The stream x stream products are the place for
// v_mfma_f64_16x16x4 when nmu = 16.
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
};

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
// raw azimuth integrals (phasint2 :141-197), one block per (wave, scatterer), all ic
__global__ __launch_bounds__(256) void k_ms_phase(MsParams p)
{
    const int widx = blockIdx.x, comp = blockIdx.y + p.phase_comp0;  // comp == ncont -> Rayleigh
    const int n = p.nmu, nn = n * n, tid = threadIdx.x;
    const double pi = 3.141592653589793;
    const double dphi = 2.0 * pi / p.nphi;
    const int jc = comp < p.ncont ? comp : (p.ncont > 0 ? p.ncont - 1 : 0);
    const double *pfunc = p.phasarr + (((size_t)jc * p.nwave + widx) * 2 + 0) * p.nth;
    const double *xmu = p.phasarr + (((size_t)jc * p.nwave + widx) * 2 + 1) * p.nth;
    const int iscat = (comp == p.ncont) ? 0 : (p.imie == 0 ? 2 : 4);
    for (int work = tid; work < nn * (p.nf + 1); work += blockDim.x) {
        const int ic = work / nn, e = work % nn;
        const int i = e / n, j = e % n;
        const double sthi = sqrt(1.0 - p.mu[i] * p.mu[i]), sthj = sqrt(1.0 - p.mu[j] * p.mu[j]);
        const double ss = sthi * sthj, mmu = p.mu[i] * p.mu[j];
        double spl = 0.0, smi = 0.0;
        for (int k = 0; k <= p.nphi; ++k) {
            const double phi = k * dphi;
            const double cphi = cos(phi);
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
            double wphi = (k == 0 || k == p.nphi) ? 0.5 * dphi : dphi;
            if (ic == 0) wphi /= (2.0 * pi); else wphi /= pi;
            const double cic = cos(ic * phi);
            spl += wphi * (pl * cic);
            smi += wphi * (pm * cic);
        }
        p.ppl[(((size_t)widx * (p.nf + 1) + ic) * p.ncomp + comp) * nn + e] = spl;
        p.pmi[(((size_t)widx * (p.nf + 1) + ic) * p.ncomp + comp) * nn + e] = smi;
    }
}

// Hansen renormalisation (hansen :200-233).  The reference keeps ONE fc array per scatterer for the whole
// call and hansen() updates it in place, so the factor found for (g, wave) is the starting point of the
// next (g, wave) in loop order (:780-815); the normalisation has no unique solution, so the result depends
// on that history (a 1e-4 effect on the radiance).  Reproduced: one wavefront per scatterer walks the
// (g outer, wave inner) sequence and stores fc[g][wave][comp][nmu*nmu].
__global__ __launch_bounds__(64) void k_ms_hansen_seq(MsParams p)
{
    __shared__ double ppl[kMsMaxMu * kMsMaxMu], pmi[kMsMaxMu * kMsMaxMu], fc[kMsMaxMu * kMsMaxMu];
    __shared__ double rsum[kMsMaxMu], tsum[kMsMaxMu];
    __shared__ double test_s;
    const int comp = blockIdx.x + p.hansen_comp0;
    const int n = p.nmu, nn = n * n, tid = threadIdx.x;
    const double x1 = 2.0 * 3.141592653589793;
    for (int e = tid; e < nn; e += 64) fc[e] = 1.0;
    __syncthreads();
    for (int ig = 0; ig < p.ng; ++ig)
        for (int widx = 0; widx < p.nwave; ++widx) {
            const double *gppl = p.ppl + (((size_t)widx * (p.nf + 1) + 0) * p.ncomp + comp) * nn;
            const double *gpmi = p.pmi + (((size_t)widx * (p.nf + 1) + 0) * p.ncomp + comp) * nn;
            for (int e = tid; e < nn; e += 64) { ppl[e] = gppl[e]; pmi[e] = gpmi[e]; }
            __syncthreads();
            if (tid < n) {
                double s = 0.0;
                for (int i = 0; i < n; ++i) s += pmi[i * n + tid] * p.wtmu[i];
                rsum[tid] = s * x1;
            }
            __syncthreads();
            for (int niter = 0; niter < 10000; ++niter) {
                if (tid < n) {
                    double s = 0.0;
                    for (int i = 0; i < n; ++i) s += ppl[i * n + tid] * p.wtmu[i] * fc[i * n + tid];
                    tsum[tid] = s * x1;
                }
                __syncthreads();
                if (tid == 0) {
                    double t = 0.0;
                    for (int j = 0; j < n; ++j) { double v = fabs(rsum[j] + tsum[j] - 1.0); if (v > t) t = v; }
                    test_s = t;
                }
                __syncthreads();
                if (test_s < 1e-14) break;
                for (int e = tid; e < nn; e += 64) {
                    const int i = e / n, j = e % n;
                    if (i <= j) {
                        const double xj = (1.0 - rsum[j]) / tsum[j], xi = (1.0 - rsum[i]) / tsum[i];
                        const double v = 0.5 * (fc[i * n + j] * xj + fc[j * n + i] * xi);
                        fc[i * n + j] = v;
                        fc[j * n + i] = v;
                    }
                }
                __syncthreads();
            }
            double *ofc = p.fc + (((size_t)ig * p.nwave + widx) * p.ncomp + comp) * nn;
            for (int e = tid; e < nn; e += 64) ofc[e] = fc[e];
            __syncthreads();
        }
}

// ---- small dense helpers on LDS matrices (one wavefront = one block) ----------------------------------
__device__ __forceinline__ void ms_mm(int n, const double *A, const double *B, double *C, int lane)
{   // C = A B   (C distinct from A and B)
    for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += A[i * n + k] * B[k * n + j];
        C[e] = s;
    }
    __syncthreads();
}
__device__ __forceinline__ void ms_mv(int n, const double *A, const double *x, double *y, int lane)
{   // y = A x   (y distinct from x)
    if (lane < n) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += A[lane * n + k] * x[k];
        y[lane] = s;
    }
    __syncthreads();
}
__device__ __forceinline__ double ms_frob(int n, const double *r, int lane)
{
    double s = 0.0;
    for (int e = lane; e < n * n; e += 64) s += r[e] * r[e];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return sqrt(s);
}
// Ainv = inverse(A) by Gauss-Jordan with partial pivoting; A is destroyed.  col = scratch[n], piv_s = scratch int
__device__ __forceinline__ void ms_inv(int n, double *A, double *Ainv, double *col, int *piv_s, int lane)
{
    for (int e = lane; e < n * n; e += 64) Ainv[e] = ((e / n) == (e % n)) ? 1.0 : 0.0;
    __syncthreads();
    for (int c = 0; c < n; ++c) {
        if (lane == 0) {
            int piv = c;
            double best = fabs(A[c * n + c]);
            for (int r = c + 1; r < n; ++r) { double v = fabs(A[r * n + c]); if (v > best) { best = v; piv = r; } }
            *piv_s = piv;
        }
        __syncthreads();
        const int piv = *piv_s;
        if (piv != c && lane < n) {
            double t = A[c * n + lane]; A[c * n + lane] = A[piv * n + lane]; A[piv * n + lane] = t;
            t = Ainv[c * n + lane]; Ainv[c * n + lane] = Ainv[piv * n + lane]; Ainv[piv * n + lane] = t;
        }
        __syncthreads();
        const double d = 1.0 / A[c * n + c];
        __syncthreads();
        if (lane < n) { A[c * n + lane] *= d; Ainv[c * n + lane] *= d; col[lane] = A[lane * n + c]; }
        __syncthreads();
        for (int e = lane; e < n * n; e += 64) {
            const int r = e / n, j = e % n;
            if (r != c) {
                const double f = col[r];
                A[e] -= f * A[c * n + j];
                Ainv[e] -= f * Ainv[c * n + j];
            }
        }
        __syncthreads();
    }
}

// ---- (R,T,J) chain of one (wave, g, ic) -------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ms_chain(MsParams p)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    const int n = p.nmu, nn = n * n;
    const int ic = blockIdx.x % (p.nf + 1);
    const int ig = (blockIdx.x / (p.nf + 1)) % p.ng;
    const int widx = blockIdx.x / ((p.nf + 1) * p.ng);
    const double pi = 3.141592653589793;
    // LDS carve-up
    double *rc = sm, *tc = rc + nn, *r1 = tc + nn, *t1 = r1 + nn, *pp = t1 + nn, *pm = pp + nn;
    double *m0 = pm + nn, *m1 = m0 + nn, *m2 = m1 + nn, *m3 = m2 + nn, *m4 = m3 + nn, *m5 = m4 + nn;
    double *jc = m5 + nn, *j1 = jc + kMsMaxMu, *v0 = j1 + kMsMaxMu, *v1 = v0 + kMsMaxMu, *col = v1 + kMsMaxMu;
    double *radg = col + kMsMaxMu;
    int *piv_s = reinterpret_cast<int *>(radg + kMsMaxMu);

    if (lane < n) radg[lane] = p.radg[(size_t)widx * n + (n - 1 - lane)];   // radg[:, ::-1] :765
    __syncthreads();
    bool defined = false;
    if (p.lowbc > 0) {  // surface operator first :824-836
        for (int e = lane; e < nn; e += 64) {
            const int i = e / n, j = e % n;
            rc[e] = (2. * (p.brdf[(((size_t)widx * n + i) * n + j) * (p.nf + 1) + ic] * pi) * p.mu[j] * p.wtmu[j]) * p.xfac;
            tc[e] = 0.0;
        }
        if (lane < n) jc[lane] = radg[lane];
        defined = true;
        __syncthreads();
    }
    const double *PPL = p.ppl + (((size_t)widx * (p.nf + 1) + ic) * p.ncomp) * nn;
    const double *PMI = p.pmi + (((size_t)widx * (p.nf + 1) + ic) * p.ncomp) * nn;
    const double *FC = p.fc + (((size_t)ig * p.nwave + widx) * p.ncomp) * nn;   // ppl *= fc (:232)

    for (int l = 0; l < p.nlay; ++l) {
        const int k = l;  // look-down: bottom layer first (:842-845)
        const double taut = p.taus[((size_t)widx * p.ng + ig) * p.nlay + k];
        const double bc = p.bnu[(size_t)widx * p.nlay + k];
        double omega = p.omegas[((size_t)widx * p.ng + ig) * p.nlay + k];
        if (omega < 0) omega = 0.0;
        if (omega > 1) omega = 1.0;
        double tauscat = taut * omega;
        const double taur = p.tauray[(size_t)widx * p.nlay + k];
        tauscat = tauscat - taur;
        if (tauscat < 0) tauscat = 0.0;
        // ---- calc_rtj_matrix :566-647 -> (r1, t1, j1), iscl ------------------------------------------------
        int iscl = 0;
        omega = (tauscat + taur) / taut;
        if (taut == 0) {
            for (int e = lane; e < nn; e += 64) { r1[e] = 0.0; t1[e] = ((e / n) == (e % n)) ? 1.0 : 0.0; }
            if (lane < n) j1[lane] = 0.0;
            __syncthreads();
        } else if (omega == 0) {
            for (int e = lane; e < nn; e += 64) { r1[e] = 0.0; t1[e] = 0.0; }
            __syncthreads();
            if (lane < n) {
                const double tex = -(1. / p.mu[lane]) * taut;
                const double tt = (tex > -200.0) ? exp(tex) : 0.0;
                t1[lane * n + lane] = tt;
                j1[lane] = bc * (1.0 - tt);
            }
            __syncthreads();
        } else {
            iscl = 1;
            const double fr = taur / (tauscat + taur), fs = tauscat / (tauscat + taur);
            for (int e = lane; e < nn; e += 64) {
                double a = (p.iray > 0) ? fr * (PPL[(size_t)p.ncont * nn + e] * FC[(size_t)p.ncont * nn + e]) : 0.0;
                double b = (p.iray > 0) ? fr * PMI[(size_t)p.ncont * nn + e] : 0.0;
                for (int c = 0; c < p.ncont; ++c) {
                    const double f = p.lfrac[((size_t)widx * p.ncont + c) * p.nlay + k];
                    a += fs * (PPL[(size_t)c * nn + e] * FC[(size_t)c * nn + e]) * f;
                    b += fs * PMI[(size_t)c * nn + e] * f;
                }
                pp[e] = a;
                pm[e] = b;
            }
            __syncthreads();
            // ---- double1 :321-362 --------------------------------------------------------------------------
            double con = omega * pi;
            con *= (ic == 0) ? 2.0 : 1.0;
            const int nd = (int)(log2(taut) + 12);   // python int(): truncation toward zero
            const double tau0 = taut * ((nd >= 1) ? 1.0 / exp2((double)nd) : 1.0);
            // Gamma++ = M^-1 (E - con P++ C) ;  Gamma+- = M^-1 con P+- C   (C, M^-1 diagonal)
            for (int e = lane; e < nn; e += 64) {
                const int i = e / n, j = e % n;
                const double gpp = (1. / p.mu[i]) * (((i == j) ? 1.0 : 0.0) - (pp[e] * p.wtmu[j]) * con);
                const double gpm = (1. / p.mu[i]) * ((pm[e] * p.wtmu[j]) * con);
                t1[e] = ((i == j) ? 1.0 : 0.0) - tau0 * gpp;
                r1[e] = tau0 * gpm;
            }
            if (lane < n) j1[lane] = (ic == 0) ? (1.0 - omega) * bc * tau0 * (1. / p.mu[lane]) : 0.0;
            __syncthreads();
            for (int it = 0; it < nd; ++it) {   // add :275-297
                ms_mm(n, r1, r1, m0, lane);                       // bcom
                if (ms_frob(n, r1, lane) > 0.1) {
                    for (int e = lane; e < nn; e += 64) m1[e] = (((e / n) == (e % n)) ? 1.0 : 0.0) - m0[e];
                    __syncthreads();
                    ms_inv(n, m1, m2, col, piv_s, lane);          // acom = inv(e - bcom)
                } else {
                    for (int e = lane; e < nn; e += 64) m2[e] = (((e / n) == (e % n)) ? 1.0 : 0.0) + m0[e];
                    __syncthreads();
                }
                ms_mm(n, t1, m2, m3, lane);                       // ccom = t1 acom
                ms_mm(n, m3, r1, m0, lane);                       // rans = ccom r1
                ms_mm(n, m0, t1, m1, lane);                       // acom = rans t1
                ms_mm(n, m3, t1, m4, lane);                       // tans = ccom t1
                if (ic == 0) {
                    ms_mv(n, r1, j1, v0, lane);                   // jcom = r1 j1 + j1
                    if (lane < n) v0[lane] = v0[lane] + j1[lane];
                    __syncthreads();
                    ms_mv(n, m3, v0, v1, lane);                   // jans = ccom jcom + j1
                    if (lane < n) j1[lane] = v1[lane] + j1[lane];
                }
                for (int e = lane; e < nn; e += 64) { r1[e] = r1[e] + m1[e]; t1[e] = m4[e]; }
                __syncthreads();
            }
        }
        // ---- combine with the stack below :868-875 ------------------------------------------------------------
        if (l == 0 && !defined) {
            for (int e = lane; e < nn; e += 64) { rc[e] = r1[e]; tc[e] = t1[e]; }
            if (lane < n) jc[lane] = j1[lane];
            __syncthreads();
        } else if (iscl == 1) {   // addp, scattering layer :486-511 (rsub,tsub,jsub) = (rc,tc,jc)
            ms_mm(n, rc, r1, m0, lane);                           // rsq = rsub r1
            if (ms_frob(n, m0, lane) > 0.01) {
                for (int e = lane; e < nn; e += 64) m1[e] = (((e / n) == (e % n)) ? 1.0 : 0.0) - m0[e];
                __syncthreads();
                ms_inv(n, m1, m2, col, piv_s, lane);
            } else {
                for (int e = lane; e < nn; e += 64) m2[e] = (((e / n) == (e % n)) ? 1.0 : 0.0) + m0[e];
                __syncthreads();
            }
            ms_mm(n, t1, m2, m3, lane);                           // ccom = t1 acom
            ms_mm(n, m3, rc, m0, lane);                           // rans = ccom rsub
            ms_mm(n, m0, t1, m1, lane);                           // bcom = rans t1
            ms_mm(n, m3, tc, m4, lane);                           // tans = ccom tsub
            ms_mv(n, rc, j1, v0, lane);                           // jcom = rsub j1 + jsub
            if (lane < n) v0[lane] += jc[lane];
            __syncthreads();
            ms_mv(n, m3, v0, v1, lane);                           // jans = ccom jcom + j1
            if (lane < n) jc[lane] = v1[lane] + j1[lane];
            for (int e = lane; e < nn; e += 64) { rc[e] = r1[e] + m1[e]; tc[e] = m4[e]; }
            __syncthreads();
        } else {                  // addp, non-scattering layer :513-530
            ms_mv(n, rc, j1, v0, lane);
            if (lane < n) v0[lane] += jc[lane];
            __syncthreads();
            for (int e = lane; e < nn; e += 64) {
                const int i = e / n, j = e % n;
                const double ta = t1[i * n + i], tb = t1[j * n + j];
                m0[e] = tc[e] * ta;
                m1[e] = rc[e] * ta * tb;
            }
            if (lane < n) v1[lane] = j1[lane] + t1[lane * n + lane] * v0[lane];
            __syncthreads();
            for (int e = lane; e < nn; e += 64) { tc[e] = m0[e]; rc[e] = m1[e]; }
            if (lane < n) jc[lane] = v1[lane];
            __syncthreads();
        }
    }
    if (ic != 0 && lane < n) jc[lane] = 0.0;   // :881-882
    __syncthreads();
    // ---- per path: the four (mu0, mu) samples and the bilinear interpolation :886-945 ---------------------------
    if (lane < p.ngeom) {
        const int ipath = lane;
        const double sol_ang = p.sol_ang[ipath], emiss_ang = p.emiss_ang[ipath];
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
                double bcom = 0.0;   // (T utmi)[imu], utmi = radg for ic == 0 else 0
                if (ic == 0) for (int kk = 0; kk < n; ++kk) bcom += tc[imu * n + kk] * radg[kk];
                yx[ico++] = (rc[imu * n + imu0] * s0 + bcom) + jc[imu];
            }
        }
        double drad = ((1 - t) * (1 - u) * yx[0] + t * (1 - u) * yx[1] + t * u * yx[3] + (1 - t) * u * yx[2]) *
                      cos(ic * p.aphi[ipath] * pi / 180.0);
        if (ic > 0) drad *= 2;
        p.drad[(((size_t)widx * p.ng + ig) * (p.nf + 1) + ic) * p.ngeom + ipath] = drad;
    }
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

}  // namespace ansfm
