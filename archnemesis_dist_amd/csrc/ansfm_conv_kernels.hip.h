// ansfm_conv_kernels.hip.h -- instrument-line-shape convolution of a monochromatic spectrum and its gradients.
//
// Measurement_0.lblconv (:3335), lblconvg (:3799), lblconv_fil (:3549), lblconvg_fil (:3992) and their *_ngeom
// variants (:3444, :3685, :3614, :3912; ny spectra and ny*nx gradient columns on one grid): for every convolution
// wavenumber a weighted mean over the calculation points inside the ILS window,
//     yout[j] = sum_i f1_i y_i / sum_i f1_i  (only f1_i > 0 counts),   gradout[j,x] likewise with dydx[i,x],
// f1 from ISHAPE (square, triangular, gaussian, Hamming; Hanning assigns no weight in the reference -> 0/0) or from a
// tabulated filter interpolated like np.interp.  It is a banded (nconv x window)·(window x (nx+1)) product; one
// block per (convolution point, 128 columns): the weights of 128 window points are computed once per block into
// LDS, each thread then accumulates its column in index order -- the reference's summation order.
// vwave must be ascending (the window is found by bisection instead of np.where).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ansfm {

struct ConvParams {
    const double *vwave, *y, *dydx;      // [nwave], [nwave][ny], [nwave][nx] or nullptr   (ngeom: ny = NGEOM, nx = NGEOM*NX)
    const double *vconv;                 // [nconv]
    const int32_t *nfil;                 // filter mode: [nconv]
    const double *vfil, *afil;           // filter mode: [nfilmax][nconv]
    double *yout, *gradout;              // [nconv][ny], [nconv][nx]
    int nwave, nx, ny, nconv, ishape, hamming_rule, filter;   // filter 2: Measurement_0.conv / convg window (bracketing points)
                                                               // filter 3: integrate_filter* -- np.trapz of filter x spectrum, no normalisation
    double fwhm;
};

__device__ __forceinline__ double conv_shape(int ishape, double v, double vcen, double fwhm, double sig)
{
    const double PI = 3.141592653589793;
    if (ishape == 0) return 1.0;
    if (ishape == 1) return 1.0 - fabs(v - vcen) / fwhm;
    if (ishape == 2) { const double t = (v - vcen) / sig; return exp(-(t * t)); }
    if (ishape == 3) {                                                   // :3412-3420
        const double a = 0.907 / fwhm, k = v - vcen;
        if (k == 0.0) return a * 1.08;
        const double num = a * (1.08 - ((0.64 * (a * a)) * (k * k))) * sin(((2 * PI) * a) * k);
        const double den = (1 - (4 * (a * a)) * (k * k)) * (((2 * PI) * a) * k);
        return num / den;
    }
    return 0.0;                                                          // Hanning: `else: pass`
}

__global__ __launch_bounds__(128) void k_ils_conv(ConvParams p)
{
    __shared__ double fw[128];
    const int j = blockIdx.x, tid = threadIdx.x;
    const int c = blockIdx.y * 128 + tid;                 // column: < nx gradient, then the ny spectra
    const int ncol = p.nx + p.ny;
    const double vcen = p.vconv[j];
    double v1, v2, sig = 0.0;
    const double *xp = nullptr, *yp = nullptr;
    int nf = 0;
    if (p.filter) {
        nf = p.nfil[j];
        xp = p.vfil + j; yp = p.afil + j;                 // column j, stride nconv
        v1 = xp[0]; v2 = xp[(size_t)(nf - 1) * p.nconv];
    } else if (p.ishape == 0) { v1 = vcen - 0.5 * p.fwhm; v2 = v1 + p.fwhm; }
    else if (p.ishape == 1) { v1 = vcen - p.fwhm; v2 = vcen + p.fwhm; }
    else if (p.ishape == 2) { sig = 0.5 * p.fwhm / sqrt(log(2.0)); v1 = vcen - 3. * sig; v2 = vcen + 3. * sig; }
    else if (p.ishape == 3) {
        if (p.hamming_rule == 1) { v1 = vcen - p.fwhm; v2 = vcen + p.fwhm; }          // lblconvg :3866-3868
        else if (p.hamming_rule == 2) { v1 = vcen - p.fwhm; v2 = vcen - p.fwhm; }     // *_ngeom  :3501-3503, :3753-3755
        else { v1 = vcen - 1.1 * p.fwhm; v2 = vcen - 1.1 * p.fwhm; }                  // lblconv  :3391-3393
    } else { v1 = vcen - 3. * p.fwhm; v2 = vcen + 3. * p.fwhm; }
    // window [i0, i1): vwave >= v1 and vwave <= v2
    int lo = 0, hi = p.nwave;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.vwave[mid] < v1) lo = mid + 1; else hi = mid; }
    const int i0 = lo;
    hi = p.nwave;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.vwave[mid] <= v2) lo = mid + 1; else hi = mid; }
    int i1 = lo;
    int i0x = i0;
    if (p.filter == 2) { i0x = i0 - 1; i1 = i1 + 1; }     // conv/convg (:2437-2442): from the last point below v1 to the first above v2 (host checked both exist)
    double acc = 0.0, nor = 0.0;
    for (int base = i0x; base < i1; base += 128) {
        const int i = base + tid;
        double f = 0.0;
        if (i < i1) {
            const double v = p.vwave[i];
            if (p.filter) {                                // np.interp(v, xp, yp), v inside [xp[0], xp[nf-1]]
                int a = 0, b = nf - 1;
                while (b - a > 1) { const int mid = (a + b) >> 1; if (xp[(size_t)mid * p.nconv] <= v) a = mid; else b = mid; }
                const double x0 = xp[(size_t)a * p.nconv], x1 = xp[(size_t)(a + 1) * p.nconv];
                const double y0 = yp[(size_t)a * p.nconv], y1 = yp[(size_t)(a + 1) * p.nconv];
                f = (v >= v2) ? yp[(size_t)(nf - 1) * p.nconv] : (v <= v1) ? yp[0] : ((y1 - y0) / (x1 - x0)) * (v - x0) + y0;
                if (p.filter == 3) {       // trapezoid node weight: f_i (dv_left + dv_right) / 2 inside the window (:4124)
                    const double dl = (i > i0x) ? v - p.vwave[i - 1] : 0.0;
                    const double dr = (i + 1 < i1) ? p.vwave[i + 1] - v : 0.0;
                    f = f * (0.5 * (dl + dr));
                }
            } else
                f = conv_shape(p.ishape, v, vcen, p.fwhm, sig);
        }
        __syncthreads();
        fw[tid] = f;
        __syncthreads();
        if (c < ncol) {
            const int n = min(128, i1 - base);
            for (int k = 0; k < n; ++k) {
                const double fk = fw[k];
                if (fk > 0.0 || p.filter == 3) {           // :3433 (the integrals take every point)
                    const double val = (c < p.nx) ? p.dydx[(size_t)(base + k) * p.nx + c]
                                                  : p.y[(size_t)(base + k) * p.ny + (c - p.nx)];
                    acc = acc + fk * val;
                    nor = nor + fk;
                }
            }
        }
    }
    const double res = (p.filter == 3) ? acc : acc / nor;
    if (c < p.nx) p.gradout[(size_t)j * p.nx + c] = res;
    else if (c < ncol) p.yout[(size_t)j * p.ny + (c - p.nx)] = res;
}

}  // namespace ansfm
