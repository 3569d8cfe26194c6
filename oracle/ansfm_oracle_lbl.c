/*
 * ansfm_oracle_lbl.c -- CPU restatement of the runtime line-by-line kernels.  TEST INFRASTRUCTURE ONLY.
 * Follows archnemesis/LineData_0.py: line_strength :206, doppler_width :144, lorentz_width :159,
 * line_shift :189, add_line_set_monochromatic_spectrum :229-277, add_line_set_monochromatic_absorption :280,
 * and lineshape/{voigt_impl/voigt_scipy.py:8, lorentz.py:8, gaussian.py:8}.
 *
 * Third-party arithmetic: the default Voigt is scipy.special.voigt_profile (scipy un-pinned in the
 * reference's setup.py; 1.15.3 in the build container) = Re[wofz((x+i*gamma)/(sigma*sqrt2))]/(sigma*sqrt(2pi)).
 * wofz is restated here as: |z| >= 8 -> 12-term Laplace continued fraction; otherwise the exponentially
 * convergent trapezoid/midpoint rule with pole correction (Matta & Reichel 1971; Al Azah & Chandler-Wilde
 * 2021), h = 1/2, nodes chosen to stay >= h/4 away from Re z.  Measured against scipy.special.wofz on
 * y in [1e-12, 30], x in [0, 9]: relative error of Re w <= 3.4e-14 (oracle/gen_golden_lbl.py lattice).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#define ORC_API __attribute__((visibility("default")))
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static double rew(double x, double y)
{   /* Re w(x + i y), y >= 0 */
    x = fabs(x);
    if (x * x + y * y >= 64.0) {
        /* w = (i/sqrt(pi)) / (z - (1/2)/(z - 1/(z - (3/2)/(z - ...)))) */
        double rr = 0.0, ri = 0.0;
        for (int k = 12; k > 0; --k) {
            double dr = x - rr, di = y - ri, den = dr * dr + di * di;
            rr = (k * 0.5) * dr / den;
            ri = -(k * 0.5) * di / den;
        }
        double dr = x - rr, di = y - ri, den = dr * dr + di * di;
        /* (i/sqrt(pi)) / (dr + i di) = (i (dr - i di))/(sqrt(pi) den) -> real part = di/(sqrt(pi) den) */
        return di / (sqrt(M_PI) * den);
    }
    const double h = 0.5;
    double fr = x / h - floor(x / h);
    int use_mid = (fr < 0.25) || (fr > 0.75);
    double shift = use_mid ? 0.5 : 0.0;
    double s = 0.0;
    for (int k = -14; k <= 13; ++k) {
        double t = (k + shift) * h;
        s += exp(-t * t) * y / ((x - t) * (x - t) + y * y);
    }
    s *= h / M_PI;
    if (y < M_PI / h) {
        /* 2 exp(-z^2) / (1 -/+ exp(-2 pi i z / h)) */
        double er = exp(-(x * x - y * y)), ang = -2.0 * x * y;          /* exp(-z^2) = er (cos ang + i sin ang) */
        double nr = 2.0 * er * cos(ang), ni = 2.0 * er * sin(ang);
        double em = exp(2.0 * M_PI * y / h), a2 = -2.0 * M_PI * x / h;   /* exp(-2 pi i z/h) = em (cos a2 + i sin a2) */
        double sg = use_mid ? -1.0 : 1.0;
        double dr = 1.0 - sg * em * cos(a2), di = -sg * em * sin(a2);
        s += (nr * dr + ni * di) / (dr * dr + di * di);
    }
    return s;
}

ORC_API double orc_rew(double x, double y) { return rew(x, y); }

/* scipy.special.voigt_profile(x, sigma, gamma) */
ORC_API double orc_voigt_profile(double x, double sigma, double gamma)
{
    if (sigma == 0.0) {
        if (gamma == 0.0) return (x == 0.0) ? INFINITY : 0.0;
        return gamma / M_PI / (x * x + gamma * gamma);
    }
    if (gamma == 0.0) return 1.0 / sqrt(2.0 * M_PI) / sigma * exp(-(x / sigma) * (x / sigma) / 2.0);
    const double isq2 = 0.70710678118654752440;
    double zr = x / sigma * isq2, zi = gamma / sigma * isq2;
    return rew(zr, zi) / sigma / sqrt(2.0 * M_PI);
}

/* lineshape ids = SpectroscopicLineProfileEnum: 0 VOIGT, 4 LORENTZ, 12 DOPPLER (gaussian) */
static double lineshape(int id, double dwn, double alpha_d, double gamma_l)
{
    if (id == 4) return gamma_l / (M_PI * (gamma_l * gamma_l + dwn * dwn));
    if (id == 12) return sqrt(log(2.0) / M_PI) / alpha_d * exp(-(dwn * dwn * log(2.0)) / (alpha_d * alpha_d));
    return orc_voigt_profile(dwn, alpha_d / sqrt(2.0 * log(2.0)), gamma_l);
}
ORC_API double orc_lineshape(int id, double dwn, double alpha_d, double gamma_l) { return lineshape(id, dwn, alpha_d, gamma_l); }

static const double c_light_cgs = 2.99792458E10, h_planck_cgs = 6.62607015E-27, k_boltzmann_cgs = 1.380649E-16,
                    N_avogadro = 6.02214129E+23;

/* add_line_set_monochromatic_absorption :280-357 (store[4][N] filled like the reference; out is ADDED to) */
ORC_API void orc_add_line_set_monochromatic_absorption(
    int nw, const double *wn_grid, int lineshape_id, double t_calc, double t_ref, double p_calc, double p_ref,
    double q_ratio, double isotopic_abundance, double isotopic_mass, int M, const double *mol_mix_frac,
    int N, const double *broadening_params /*[3M][N]*/, const double *nu, const double *sw, const double *e_lower,
    const double *stim_ref, double *out, double *store /*[4][N] or NULL*/, double s_floor, double wn_calc_window,
    double wn_approx_window)
{
    const double c2_cgs = c_light_cgs * h_planck_cgs / k_boltzmann_cgs;
    double *st = store ? store : (double *)malloc(sizeof(double) * 4 * N);
    double *strength = st, *alpha_d = st + N, *gamma_l = st + 2 * N, *shift = st + 3 * N;
    const double boltz = c2_cgs * (t_calc - t_ref) / (t_calc * t_ref);
    const double dconst = (1.0 / c_light_cgs) * sqrt(2 * log(2.0) * N_avogadro * k_boltzmann_cgs);
    const double t_ratio = t_ref / t_calc, p_ratio = p_calc / p_ref;
    for (int i = 0; i < N; ++i) {
        strength[i] = sw[i] * ((1 - exp(-c2_cgs * nu[i] / t_calc)) / stim_ref[i]) * exp(boltz * e_lower[i]) * q_ratio;
        alpha_d[i] = dconst * nu[i] * sqrt(t_calc / isotopic_mass);
        double g = 0, sh = 0;
        for (int j = 0; j < M; ++j) {
            g += (pow(t_ratio, broadening_params[(size_t)(3 * j + 1) * N + i])) * broadening_params[(size_t)(3 * j) * N + i] *
                 mol_mix_frac[j] * p_ratio;
            sh += (p_ratio * broadening_params[(size_t)(3 * j + 2) * N + i]) * mol_mix_frac[j];
        }
        gamma_l[i] = g;
        shift[i] = sh;
    }
    /* add_line_set_monochromatic_spectrum :229-277 */
    const double cmin = -1 * wn_calc_window, cmax = wn_calc_window, amin = -1 * wn_approx_window, amax = wn_approx_window;
    for (int i = 0; i < N; ++i) {
        if (strength[i] < s_floor) continue;
        double line_approx_const = lineshape(lineshape_id, cmax, alpha_d[i], gamma_l[i]);
        for (int j = 0; j < nw; ++j) {
            double wn_delta = wn_grid[j] - (nu[i] + shift[i]);
            if (wn_delta >= amax) break;
            if (wn_delta < amin) continue;
            if (cmin <= wn_delta && wn_delta < cmax)
                out[j] += isotopic_abundance * strength[i] * lineshape(lineshape_id, wn_delta, alpha_d[i], gamma_l[i]);
            else
                out[j] += isotopic_abundance * strength[i] * line_approx_const * pow(cmax, 2.) / (wn_delta * wn_delta);
        }
    }
    if (!store) free(st);
}
