// ansfm_lbl_kernels.hip.h -- runtime line-by-line absorption on gfx950.
//
// Restates LineData_0.add_line_set_monochromatic_absorption (LineData_0.py:280-357) = line_strength :206,
// doppler_width :144, lorentz_width :159, line_shift :189, add_line_set_monochromatic_spectrum :229-277, with the
// line shapes lineshape/voigt (scipy.special.voigt_profile), lorentz and gaussian.
//
// The reference loops lines (outer) x grid points (inner) and accumulates into out[j]: a scatter with a
// race on j if parallelised over lines.  Here one thread owns one (layer, grid point) and gathers the lines
// whose +-wn_approx_window contains it, in ascending line order (the order the reference adds them in);
// a block of 256 consecutive grid points walks one shared line range, so line parameters are broadcast loads.
// Faddeeva function: |z| >= 8 -> Laplace continued fraction (2..12 terms by |z|), else trapezoid/midpoint rule (h = 1/2,
// node weights exp(-t^2) tabulated, two nodes per division) with pole correction -- <= 1.2e-13 relative against
// scipy.special.wofz for y in [1e-12, 1e2].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ansfm {

// exp(-t^2) at the 28 quadrature nodes t = k/2 (k = -14..13) and at the mid-point set t = k/2 + 1/4
__device__ const double kLblE0[28] = {5.242885663363464e-22, 4.4777324417183015e-19, 2.319522830243569e-16, 7.287724095819692e-14, 1.3887943864964021e-11, 1.6052280551856116e-09, 1.1253517471925912e-07, 4.785117392129009e-06, 0.00012340980408667956, 0.0019304541362277093, 0.01831563888873418, 0.10539922456186433, 0.36787944117144233, 0.7788007830714049, 1.0, 0.7788007830714049, 0.36787944117144233, 0.10539922456186433, 0.01831563888873418, 0.0019304541362277093, 0.00012340980408667956, 4.785117392129009e-06, 1.1253517471925912e-07, 1.6052280551856116e-09, 1.3887943864964021e-11, 7.287724095819692e-14, 2.319522830243569e-16, 4.4777324417183015e-19};
__device__ const double kLblE1[28] = {1.6310139226701858e-20, 1.0848552640429378e-17, 4.37661850287085e-15, 1.0709232382508077e-12, 1.5893910094516368e-10, 1.4307241918567688e-08, 7.811489408304491e-07, 2.586810022265412e-05, 0.0005195746821548384, 0.006329715427485747, 0.04677062238395898, 0.2096113871510978, 0.569782824730923, 0.9394130628134758, 0.9394130628134758, 0.569782824730923, 0.2096113871510978, 0.04677062238395898, 0.006329715427485747, 0.0005195746821548384, 2.586810022265412e-05, 7.811489408304491e-07, 1.4307241918567688e-08, 1.5893910094516368e-10, 1.0709232382508077e-12, 4.37661850287085e-15, 1.0848552640429378e-17, 1.6310139226701858e-20};

constexpr int kLblRows = 8;   // constants per (layer, line) in the store, see LblParams::store

// n / d by v_rcp_f64 (|1 - d r| <= 4.7e-8 measured) + ONE Newton step (-> ~2e-15) + the residual correction of the
// quotient (-> ~1e-30 before the final rounding): equal to the IEEE quotient on all 2^26 random pairs of
// tools/calib/div_check.hip, as the two-step form is
__device__ __forceinline__ double lbl_div(double n, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

// |z| < 8: exponentially convergent trapezoid / midpoint rule with pole correction (x >= 0).
// A function of its own, NOT inlined, and its node loop only unrolled twice: few evaluations come here (points within 8
// widths of a line centre), but inlined into the four-point body of k_lbl_accumulate it (a) raised the kernel to 159
// VGPRs = 3 waves per SIMD and (b) let the compiler move e^{2 pi y / h}, which depends on the line only, ahead of the window
// tests: ~35 instructions per (line, thread) that almost no point used (4.49e11 -> 3.62e11 VALU instructions on C5).
__device__ __attribute__((noinline)) double lbl_rew_near(double x, double y)
{
    const double PI = 3.141592653589793;
    const double h = 0.5;
    const double fr = x / h - floor(x / h);
    const bool use_mid = (fr < 0.25) || (fr > 0.75);
    const double shift = use_mid ? 0.25 : 0.0;
    const double yy = y * y;
    double s = 0.0;
#pragma unroll 2
    for (int k = 0; k < 28; k += 2) {          // two nodes per division: e1/d1 + e2/d2 = (e1 d2 + e2 d1)/(d1 d2)
        const double t1 = (k - 14) * h + shift, t2 = (k - 13) * h + shift;
        const double e1 = use_mid ? kLblE1[k] : kLblE0[k], e2 = use_mid ? kLblE1[k + 1] : kLblE0[k + 1];
        const double d1 = (x - t1) * (x - t1) + yy, d2 = (x - t2) * (x - t2) + yy;
        s += lbl_div(e1 * d2 + e2 * d1, d1 * d2);
    }
    s *= y;
    s *= h / PI;
    if (y < PI / h) {
        const double er = exp(-(x * x - y * y)), ang = -2.0 * x * y;
        double sa, ca, s2, c2;
        sincos(ang, &sa, &ca);
        const double nr = 2.0 * er * ca, ni = 2.0 * er * sa;
        const double em = exp(2.0 * PI * y / h);
        sincos(-2.0 * PI * x / h, &s2, &c2);
        const double sg = use_mid ? -1.0 : 1.0;
        const double dr = 1.0 - sg * em * c2, di = -sg * em * s2;
        s += (nr * dr + ni * di) / (dr * dr + di * di);
    }
    return s;
}

__device__ __forceinline__ double lbl_rew(double x, double y)
{
    const double PI = 3.141592653589793;
    x = fabs(x);
    const double r2 = x * x + y * y;
    if (r2 >= 64.0) {
        // Laplace continued fraction; the number of terms needed for ~1e-14 falls quickly with |z| (measured against
        // scipy.special.wofz: 12 terms at |z| = 8, 8 at 12, 6 at 20, 4 at 40, 3 at 100, 2 beyond 1000).  Far line
        // wings -- most evaluations inside the +-wn_calc_window -- take 2-3 terms; one reciprocal per term.
        const int n = r2 >= 1e6 ? 2 : r2 >= 1e4 ? 3 : r2 >= 1600.0 ? 4 : r2 >= 400.0 ? 6 : r2 >= 144.0 ? 8 : 12;
        double rr = 0.0, ri = 0.0;
        for (int k = n; k > 0; --k) {
            const double dr = x - rr, di = y - ri, inv = lbl_div(k * 0.5, dr * dr + di * di);
            rr = dr * inv;
            ri = -(di * inv);
        }
        const double dr = x - rr, di = y - ri;
        return lbl_div(di, 1.7724538509055159 * (dr * dr + di * di));
    }
    return lbl_rew_near(x, y);
}

__device__ __forceinline__ double lbl_voigt_profile(double x, double sigma, double gamma)
{   // scipy.special.voigt_profile
    const double PI = 3.141592653589793;
    if (sigma == 0.0) {
        if (gamma == 0.0) return (x == 0.0) ? __builtin_inf() : 0.0;
        return gamma / PI / (x * x + gamma * gamma);
    }
    if (gamma == 0.0) return 1.0 / sqrt(2.0 * PI) / sigma * exp(-(x / sigma) * (x / sigma) / 2.0);
    const double isq2 = 0.70710678118654752440;
    return lbl_rew(x / sigma * isq2, gamma / sigma * isq2) / sigma / sqrt(2.0 * PI);
}

// ids = SpectroscopicLineProfileEnum: 0 VOIGT, 4 LORENTZ, 12 DOPPLER
__device__ __forceinline__ double lbl_lineshape(int id, double dwn, double alpha_d, double gamma_l)
{
    const double PI = 3.141592653589793;
    if (id == 4) return gamma_l / (PI * (gamma_l * gamma_l + dwn * dwn));
    if (id == 12) return sqrt(log(2.0) / PI) / alpha_d * exp(-(dwn * dwn * log(2.0)) / (alpha_d * alpha_d));
    return lbl_voigt_profile(dwn, alpha_d / sqrt(2.0 * log(2.0)), gamma_l);
}

struct LblParams {
    const double *wn_grid;  // [nw] ascending
    const double *nu, *sw, *e_lower, *stim_ref;  // [N], nu ascending
    const double *bparams;  // [3M][N]
    const double *mmf;      // [M]
    const double *t_calc, *p_calc, *q_ratio;  // [L]
    double *store;          // [L][N][kLblRows]: strength, shifted centre nu + shift, wing numerator (iso * strength *
                            // line_approx_const * cmax^2), the Voigt constants of the line 1/(sigma sqrt 2), y = gamma/(sigma
                            // sqrt 2), 1/(sigma sqrt(2 pi)) (the first = 0: the line takes the general lineshape function),
                            // alpha_d, gamma_l -- everything k_lbl_accumulate reads about a line, in 64 bytes
    double *shift;          // [L][N] pressure shift (only the caller's `store` wants it)
    double *out;            // [L][nw]  (added to)
    int nw, N, M, L, lineshape_id;
    double t_ref, p_ref, iso_abundance, iso_mass, s_floor, wn_calc_window, wn_approx_window, max_shift;
};

// per (layer, line): strength, alpha_d, gamma_l, shift exactly as the reference fills them (:306-341) + the wing term's
// numerator (:261, :270)
__global__ void k_lbl_line_params(LblParams p)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)p.L * p.N) return;
    const int i = (int)(idx % p.N), l = (int)(idx / p.N);
    const double c_light_cgs = 2.99792458E10, h_planck_cgs = 6.62607015E-27, k_boltzmann_cgs = 1.380649E-16,
                 N_avogadro = 6.02214129E+23;
    const double c2_cgs = c_light_cgs * h_planck_cgs / k_boltzmann_cgs;
    const double t_calc = p.t_calc[l], p_calc = p.p_calc[l];
    const double boltz = c2_cgs * (t_calc - p.t_ref) / (t_calc * p.t_ref);
    const double dconst = (1.0 / c_light_cgs) * sqrt(2 * log(2.0) * N_avogadro * k_boltzmann_cgs);
    const double t_ratio = p.t_ref / t_calc, p_ratio = p_calc / p.p_ref;
    const double nu = p.nu[i];
    const double strength = p.sw[i] * ((1 - exp(-c2_cgs * nu / t_calc)) / p.stim_ref[i]) * exp(boltz * p.e_lower[i]) * p.q_ratio[l];
    const double alpha_d = dconst * nu * sqrt(t_calc / p.iso_mass);
    double g = 0, sh = 0;
    for (int j = 0; j < p.M; ++j) {
        g += (pow(t_ratio, p.bparams[(size_t)(3 * j + 1) * p.N + i])) * p.bparams[(size_t)(3 * j) * p.N + i] * p.mmf[j] * p_ratio;
        sh += (p_ratio * p.bparams[(size_t)(3 * j + 2) * p.N + i]) * p.mmf[j];
    }
    double *st = p.store + ((size_t)l * p.N + i) * kLblRows;      // one line's constants are contiguous (64 bytes)
    st[0] = strength;
    st[1] = nu + sh;                                              // :264
    st[6] = alpha_d;
    st[7] = g;
    p.shift[(size_t)l * p.N + i] = sh;
    // the whole numerator of the wing term (:270), same association as the reference's expression
    st[2] = p.iso_abundance * strength * lbl_lineshape(p.lineshape_id, p.wn_calc_window, alpha_d, g) *
                              (p.wn_calc_window * p.wn_calc_window);
    // Voigt: scipy's voigt_profile(x, sigma, gamma) = Re w((x + i gamma) / (sigma sqrt 2)) / (sigma sqrt(2 pi)) -- the
    // three per-line factors once per (layer, line) instead of three divisions per grid point
    double xs = 0.0, yv = 0.0, nrm = 0.0;
    if (p.lineshape_id == 0) {
        const double sigma = alpha_d / sqrt(2.0 * log(2.0));
        if (sigma > 0.0 && g > 0.0) {
            const double isq2 = 0.70710678118654752440;
            xs = isq2 / sigma;
            yv = g / sigma * isq2;
            nrm = 1.0 / sigma / sqrt(2.0 * 3.141592653589793);
        }
    }
    st[3] = xs;
    st[4] = yv;
    st[5] = nrm;
}

// kLblPts grid points per thread (256 apart, so a wave's loads and stores stay coalesced): the per-line work that does not
// depend on the grid point -- scalar loads of the line's constants, the strength test, the loop itself -- is paid once
// per kLblPts evaluations instead of once per evaluation.
constexpr int kLblPts = 4;

// Seven waves per SIMD (72 VGPRs): the scalar loads of each line's constants are a round trip per line that only other
// waves can cover.  What is spilled (128 bytes) are polynomial coefficients of exp() in the Doppler / general-lineshape
// branches, not the Voigt wing path.  C5 wall time on one box: near-centre branch inlined, 3 waves 0.847 s, 4 waves
// 0.746, 5 waves 0.729, 6 waves 0.762 (spills in the hot path); near-centre branch out of line, 5 waves 0.675, 6 waves
// 0.646, 7 waves 0.635, 8 waves 0.634.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_lbl_accumulate(LblParams p)
{
    const int l = blockIdx.y;
    const int j0 = blockIdx.x * (256 * kLblPts);
    const int j1 = min(j0 + 256 * kLblPts - 1, p.nw - 1);
    double wn[kLblPts], acc[kLblPts];
#pragma unroll
    for (int k = 0; k < kLblPts; ++k) {
        const int j = j0 + k * 256 + (int)threadIdx.x;
        wn[k] = p.wn_grid[j < p.nw ? j : p.nw - 1];
        acc[k] = (j < p.nw) ? p.out[(size_t)l * p.nw + j] : 0.0;
    }
    // line range shared by the block: every line that can reach any of its grid points
    const double lo_wn = p.wn_grid[j0] - p.wn_approx_window - p.max_shift;
    const double hi_wn = p.wn_grid[j1] + p.wn_approx_window + p.max_shift;
    int a = 0, b = p.N;
    while (a < b) { int mid = (a + b) >> 1; if (p.nu[mid] < lo_wn) a = mid + 1; else b = mid; }
    const int ilo = a;
    b = p.N;
    while (a < b) { int mid = (a + b) >> 1; if (p.nu[mid] <= hi_wn) a = mid + 1; else b = mid; }
    const int ihi = a;
    const double *stl = p.store + (size_t)l * p.N * kLblRows;
    const double cmin = -1 * p.wn_calc_window, cmax = p.wn_calc_window;
    const double amin = -1 * p.wn_approx_window, amax = p.wn_approx_window;
    for (int i = ilo; i < ihi; ++i) {
        const double *st = stl + (size_t)i * kLblRows;                          // 64 contiguous bytes: one scalar load
        const double strength = st[0];
        if (strength < p.s_floor) continue;                                     // :258
        const double centre = st[1];
        const double wing = st[2];
        const double xs = st[3], yv = st[4], nrm = st[5];
        const double amp = p.iso_abundance * strength;
#pragma unroll
        for (int k = 0; k < kLblPts; ++k) {
            const double wn_delta = wn[k] - centre;
            if (wn_delta >= amax || wn_delta < amin) continue;                  // :266-269
            if (cmin <= wn_delta && wn_delta < cmax) {
                const double shape = (xs != 0.0) ? lbl_rew(wn_delta * xs, yv) * nrm
                                                 : lbl_lineshape(p.lineshape_id, wn_delta, st[6], st[7]);
                acc[k] += amp * shape;
            } else
                acc[k] += lbl_div(wing, wn_delta * wn_delta);
        }
    }
#pragma unroll
    for (int k = 0; k < kLblPts; ++k) {
        const int j = j0 + k * 256 + (int)threadIdx.x;
        if (j < p.nw) p.out[(size_t)l * p.nw + j] = acc[k];
    }
}

}  // namespace ansfm
