// ansfm_cont_kernels.hip.h -- continuum opacities of the layers.
//
// ForwardModel_0.calc_tau_cia (:4516-4760): collision-induced absorption.  Per layer the CIA table K_CIA[pair][para]
// [T][wavenumber] is interpolated bilinearly in (temperature, para-H2 fraction) and linearly in wavenumber
// (scipy interp1d), multiplied by the two partners' mixing ratios and by TOTAM^2/DELH; the gradients with respect to
// the mixing ratios and temperature go with it.  The host resolves what does not depend on wavenumber (brackets and
// weights per layer, which atmospheric gases a pair refers to, the ortho/para selection) -- see ansfm_calc_tau_cia;
// the (wavenumber x layer) work is here, one thread per (wavenumber, layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ansfm {

struct CiaLayer {       // what calc_tau_cia derives per layer before touching the spectral axis (:4588-4666)
    int itl, ithi, ipl, iphi;
    double fhl_t, fhh_t, dfhldT, fhl_f, fhh_f, xfac;
};

struct CiaParams {
    const double *waven;        // [W] ascending
    const double *cia_waven;    // [NWC]
    const double *K;            // [NPAIR][NPE][NT][NWC]
    const CiaLayer *lay;        // [L]
    const int32_t *g1, *g2;     // [NPAIR] atmospheric gas of each partner, -1: pair not used
    const double *q;            // [L][NVMR]
    const double *k_co2, *k_n2n2, *k_n2h2;   // [W] or nullptr
    double *tau;                // [W][L]
    double *dtau;               // [W][L][NVMR+2] (zeroed) or nullptr
    int W, NWC, NPAIR, NPE, NT, L, NVMR, covers, ico2, in2, ih2;
};

__global__ __launch_bounds__(128) void k_tau_cia(CiaParams p)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
    if (w >= p.W) return;
    const CiaLayer c = p.lay[l];
    const double x = p.waven[w];
    const double *q = p.q + (size_t)l * p.NVMR;
    double *d = p.dtau ? p.dtau + ((size_t)w * p.L + l) * (p.NVMR + 2) : nullptr;
    double sum1 = 0.0;
    if (p.covers) {
        int lo = 0, hi = p.NWC;          // scipy interp1d: idx = clip(searchsorted(x, xn, 'left'), 1, n-1)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.cia_waven[mid] < x) lo = mid + 1; else hi = mid; }
        const int iw = lo < 1 ? 1 : (lo > p.NWC - 1 ? p.NWC - 1 : lo);
        const double xlo = p.cia_waven[iw - 1], dx = p.cia_waven[iw] - xlo;
        const size_t sT = (size_t)p.NWC, sP = (size_t)p.NT * p.NWC, sPair = (size_t)p.NPE * p.NT * p.NWC;
        for (int ip = 0; ip < p.NPAIR; ++ip) {
            const int g1 = p.g1[ip], g2 = p.g2[ip];
            if (g1 < 0 || g2 < 0) continue;
            const double *Kp = p.K + ip * sPair;
            double kt[2], dk[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = iw - 1 + e;
                const double a = Kp[c.ipl * sP + c.itl * sT + i], b = Kp[c.ipl * sP + c.ithi * sT + i];
                const double cc = Kp[c.iphi * sP + c.itl * sT + i], dd = Kp[c.iphi * sP + c.ithi * sT + i];
                const double ktlo = a * c.fhh_t + b * c.fhl_t, kthi = cc * c.fhh_t + dd * c.fhl_t;
                kt[e] = ktlo * c.fhh_f + kthi * c.fhl_f;
                dk[e] = (kthi - ktlo) * c.dfhldT;
            }
            const double k_cia = ((kt[1] - kt[0]) / dx) * (x - xlo) + kt[0];
            const double dkdT = ((dk[1] - dk[0]) / dx) * (x - xlo) + dk[0];
            sum1 = sum1 + k_cia * q[g1] * q[g2];
            if (d) {
                d[g1] = d[g1] + q[g2] * k_cia;
                d[g2] = d[g2] + q[g1] * k_cia;
                d[p.NVMR - 2] = d[p.NVMR - 2] + dkdT * q[g1] * q[g2];      // slot NVMR-2, as the reference (:4695)
            }
        }
    }
    if (p.ico2 >= 0) {
        const double k = p.k_co2[w];
        sum1 = sum1 + k * q[p.ico2] * q[p.ico2];
        if (d) d[p.ico2] = d[p.ico2] + 2. * q[p.ico2] * k;
    }
    if (p.in2 >= 0) {
        const double k = p.k_n2n2[w];
        sum1 = sum1 + k * q[p.in2] * q[p.in2];
        if (d) d[p.in2] = d[p.in2] + 2. * q[p.in2] * k;
    }
    if (p.in2 >= 0 && p.ih2 >= 0) {
        const double k = p.k_n2h2[w];
        sum1 = sum1 + k * q[p.in2] * q[p.ih2];
        if (d) {
            d[p.ih2] = d[p.ih2] + q[p.in2] * k;
            d[p.in2] = d[p.in2] + q[p.ih2] * k;
        }
    }
    p.tau[(size_t)w * p.L + l] = sum1 * c.xfac;
    if (d)
        for (int s = 0; s < p.NVMR + 2; ++s) d[s] = d[s] * c.xfac;
}

// ------------------------------------------------------------------------------------------------------------------
// Rayleigh scattering opacity of the layers: ForwardModel_0.calc_tau_rayleighj (:5525, IRAY 1, gas giants, Allen 1976),
// calc_tau_rayleighv2 (:5647, IRAY 2, CO2, Ityaksov et al. 2008), calc_tau_rayleighls (:5712, IRAY 4, Jovian air after
// Sromovsky: H2 / He / CH4 / NH3 weighted by the layer's composition) and the older calc_tau_rayleighv (:5598, not
// selected by any IRAY).  tau[w][l] = k(w[, l]) * TOTAM[l], dtau[w][l] = k.  One thread per (wavenumber, layer); the
// operation order of the reference's expressions is kept.
// ------------------------------------------------------------------------------------------------------------------
struct RayParams {
    const double *wavec;        // [W] wavenumber (ISPACE 0) or wavelength in micron (ISPACE 1)
    const double *totam;        // [L]
    const double *f4;           // mode 4: [L][4] mixing ratios of H2, He, CH4, NH3 (0 where the gas is absent)
    double *tau, *dtau;         // [W][L]  (batch: [n][W][Lm], the L = n * Lm layers of n states; dtau may be null)
    int W, L, mode, ispace;     // mode = IRAY (1, 2, 4); 12 = calc_tau_rayleighv
    int Lm;                     // layers per state of a batch, 0 = one state
};

// k(wavenumber[, composition of the layer]) of the four parametrisations: TAURAY = k * TOTAM
__device__ __forceinline__ double rayleigh_k(int mode, int ispace, double v, const double *f4row)
{
#pragma clang fp contract(off)      // n*n - 1 with n = 1 + 4e-4 cancels: a fused multiply-add would differ from NumPy by 1e-13
    const double PI = 3.141592653589793;
    double k = 0.0;
    if (mode == 1) {                                           // :5543-5577
        const double AH2 = 13.58E-5, BH2 = 7.52E-3, AHe = 3.48E-5, BHe = 2.30E-3, fH2 = 0.864;
        const double kb = 1.37971e-23, P0 = 1.01325e5, T0 = 273.15;
        const double LAMBDA = (ispace == 0) ? 1. / v * 1.0e-2 : v * 1.0e-6;
        double x = 1.0 / (LAMBDA * 1.0e6);
        const double nH2 = AH2 * (1.0 + BH2 * x * x);
        const double nHe = AHe * (1.0 + BHe * x * x);
        const double nAir = fH2 * nH2 + (1 - fH2) * nHe;
        const double temp = 32 * (PI * PI * PI) * (nAir * nAir);
        const double N0 = P0 / (kb * T0);
        x = N0 * LAMBDA * LAMBDA;
        const double faniso = (6.0 + 3.0 * 0.0) / (6.0 - 7.0 * 0.0);
        k = temp * faniso / (3. * (x * x));
    } else if (mode == 12) {                                   // :5620-5632
        const double LAMBDA = (ispace == 0) ? 1. / v * 1.0e4 : v;
        const double l2 = LAMBDA * LAMBDA;
        k = 8.8e-28 / (l2 * l2) * 1.0e-4;
    } else if (mode == 2) {                                    // :5670-5695
        const double LAMBDA = (ispace == 0) ? 1. / v * 1.0e4 : v;
        const double dens = 2.5475605e+19;
        const double lam = LAMBDA * 1.0e-4;
        const double f_king = 1.14 + (25.3e-12) / (lam * lam);
        const double nu2 = 1. / lam / lam;
        const double term1 = 5799.3 / (16.618e9 - nu2) + 120.05 / (7.9609e9 - nu2) + 5.3334 / (5.6306e9 - nu2) +
                             4.3244 / (4.6020e9 - nu2) + 1.218e-5 / (5.84745e6 - nu2);
        const double n = 1.0 + 1.1427e3 * term1;
        const double r = (n * n - 1) / (n * n + 2.0);
        const double factor1 = r * r;
        const double l2 = lam * lam;
        k = (24. * (PI * PI * PI) / (l2 * l2) / (dens * dens)) * factor1 * f_king;
        k = k * 1.0e-4;
    } else {                                                     // mode 4, :5745-5824
        const double *f = f4row;
        const double fh2 = f[0], fhe = f[1], fch4 = f[2], fnh3 = f[3];
        double fheh2 = 0.0, fch4h2 = 0.0;
        if (fh2 > 0.0) { fheh2 = fhe / fh2; fch4h2 = fch4 / fh2; }
        double comp[4];
        comp[0] = (1.0 - fnh3) / (1.0 + fheh2 + fch4h2);
        comp[1] = fheh2 * comp[0];
        comp[2] = fch4h2 * comp[0];
        comp[3] = fnh3;
        const double losch = 2.687e19 * 1.0E+12;                 // loschpm3 as the reference forms it (:5786)
        const double wl = (ispace == 0) ? 1. / v * 1.0e4 : v;
        const double A[4] = {13.58e-5, 3.48e-5, 37.0e-5, 37.0e-5};
        const double B[4] = {7.52e-3, 2.3e-3, 12.0e-3, 12.0e-3};
        const double D[4] = {0.0221, 0.025, .0922, .0922};
        double xc1 = 0.0, sumwt = 0.0;
        const double wl2 = wl * wl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double nr = 1.0 + A[j] * (1.0 + B[j] / wl2);
            const double t = nr * nr - 1.0;
            xc1 = xc1 + (t * t) * comp[j] * (6.0 + 3.0 * D[j]) / (6.0 - 7.0 * D[j]);
            sumwt = sumwt + comp[j];
        }
        const double fact = 8.0 * (PI * PI * PI) / (3.0 * (wl2 * wl2) * (losch * losch));
        k = fact * xc1 * 1.0E-8 / sumwt * 1.0e-4;
    }
    return k;
}

__global__ __launch_bounds__(128) void k_tau_rayleigh(RayParams p)
{
#pragma clang fp contract(off)
    // one thread per output element in storage order (layer fastest): the [W][L] / [n][W][Lm] arrays are written in whole
    // cache lines (a thread per wavenumber and a block row per layer wrote 8 bytes every L * 8: 3.1 ms for the 201 states
    // of a C3 Jacobian)
    const size_t flat = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Lm = p.Lm ? p.Lm : p.L;
    if (flat >= (size_t)p.W * p.L) return;
    const int lm = (int)(flat % Lm), w = (int)((flat / Lm) % p.W), l = (int)(flat / ((size_t)Lm * p.W)) * Lm + lm;
    const double k = rayleigh_k(p.mode, p.ispace, p.wavec[w], p.mode == 4 ? p.f4 + (size_t)l * 4 : nullptr);
    p.tau[flat] = k * p.totam[l];
    if (p.dtau) p.dtau[flat] = k;
}

// The same for the distinct layers of a batch, in the layout the thermal RT reads ([row][Wpad], zero beyond W): row r is
// layer work[r] (flattened (state, layer); nullptr: r itself) of totam [n * L] / f4 [n * L][4].  One thread per (row, wavenumber).
__global__ __launch_bounds__(256) void k_tau_rayleigh_rows(int rows, int W, int Wpad, int mode, int ispace, const double *__restrict__ wavec,
                                                           const int32_t *__restrict__ work, const double *__restrict__ totam,
                                                           const double *__restrict__ f4, double *__restrict__ out)
{
#pragma clang fp contract(off)
    const size_t flat = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (flat >= (size_t)rows * Wpad) return;
    const int w = (int)(flat % Wpad), r = (int)(flat / Wpad);
    if (w >= W) { out[flat] = 0.0; return; }
    const int l = work ? work[r] : r;
    const double k = rayleigh_k(mode, ispace, wavec[w], mode == 4 ? f4 + (size_t)l * 4 : nullptr);
    out[flat] = k * totam[l];
}

// ------------------------------------------------------------------------------------------------------------------
// Aerosol opacity of the layers: ForwardModel_0.calc_tau_dust (:4790-4867).  Per aerosol population the extinction and
// scattering cross sections tabulated on Scatter.WAVE are interpolated to the calculation grid with scipy's
// interp1d(kind='cubic') = the not-a-knot cubic spline (piecewise coefficients from the host, ansfm_calc_tau_dust), or
// linearly when only two points are tabulated; where the spline leaves the physical range (ksca < 0 < kext, kext < 0 <
// ksca, kext < ksca -- all three tested on the spline values, :4849-4851) the linear interpolant replaces it
// (:4853-4859).  tau = k * 1e-4 * CONT[layer][population], dtau/dq = k * 1e-4.  One thread per (wavenumber, population).
// ------------------------------------------------------------------------------------------------------------------
struct DustParams {
    const double *wavec;        // [W]
    const double *swave;        // [NWS] ascending
    const double *kext, *ksca;  // [NWS][NDUST] tabulated values
    const double *cext, *csca;  // [NDUST][NWS-1][3] spline coefficients b, c, d per interval (cubic only)
    const double *cont;         // [L][NDUST]
    double *taudust, *tauclscat, *dtaudust, *dtauclscat;   // [W][L][NDUST]
    int W, NWS, NDUST, L, cubic;
};

__global__ __launch_bounds__(128) void k_tau_dust(DustParams p)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (w >= p.W) return;
    const double x = p.wavec[w];
    const int n = p.NWS;
    // spline interval: x in [x_a, x_a+1] (the last one closed); interp1d's: searchsorted(left) clipped to [1, n-1], minus 1
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.swave[mid] < x) lo = mid + 1; else hi = mid; }
    int idx = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    const int a = idx - 1;
    const double xa = p.swave[a], xb = p.swave[a + 1];
    const double ea = p.kext[(size_t)a * p.NDUST + i], eb = p.kext[(size_t)(a + 1) * p.NDUST + i];
    const double sa = p.ksca[(size_t)a * p.NDUST + i], sb = p.ksca[(size_t)(a + 1) * p.NDUST + i];
    const double lin_e = ((eb - ea) / (xb - xa)) * (x - xa) + ea;
    const double lin_s = ((sb - sa) / (xb - xa)) * (x - xa) + sa;
    double kext = lin_e, ksca = lin_s;
    if (p.cubic) {
        const double t = x - xa;
        const double *ce = p.cext + ((size_t)i * (n - 1) + a) * 3;
        const double *cs = p.csca + ((size_t)i * (n - 1) + a) * 3;
        const double se = ea + t * (ce[0] + t * (ce[1] + t * ce[2]));
        const double ss = sa + t * (cs[0] + t * (cs[1] + t * cs[2]));
        const bool inv_s = (ss < 0) && (se > 0), inv_e = (se < 0) && (ss > 0), inv_b = (se < ss);
        kext = (inv_e || inv_b) ? lin_e : se;
        ksca = (inv_s || inv_b) ? lin_s : ss;
    }
    const double de = kext * 1.0e-4, ds = ksca * 1.0e-4;
    for (int j = 0; j < p.L; ++j) {
        const double c = p.cont[(size_t)j * p.NDUST + i];
        const size_t o = ((size_t)w * p.L + j) * p.NDUST + i;
        p.taudust[o] = de * c;
        p.tauclscat[o] = ds * c;
        p.dtaudust[o] = de;
        p.dtauclscat[o] = ds;
    }
}

}  // namespace ansfm
