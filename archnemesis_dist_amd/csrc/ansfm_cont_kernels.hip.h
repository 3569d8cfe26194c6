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

}  // namespace ansfm
