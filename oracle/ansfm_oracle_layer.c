/*
 * ansfm_oracle_layer.c -- CPU restatement of Layer_0.layer_average (archnemesis/Layer_0.py:755-1030),
 * both LAYINT schemes (MID_PATH = 0, ABSORBER_WEIGHTED_AVERAGE = 1: the Curtis-Godson branch :949-1010).
 * TEST INFRASTRUCTURE ONLY.
 *
 * Third-party arithmetic on the path (scipy un-pinned by the reference; 1.15.3 in the build container):
 *   scipy.interpolate.interp1d(kind='linear', fill_value='extrapolate')  (Layer_0.interp :645)
 *       -> idx = clip(searchsorted(x, xnew), 1, n-1); slope*(xnew - x_lo) + y_lo
 *   scipy.integrate.simpson(y, x=S) (:969-1010)
 *       -> scipy/integrate/_quadrature.py::_basic_simpson with the unequal-spacing weights; for an even number of
 *          points the last-interval correction of simpson() (alpha, beta, eta)
 *   numpy.linspace (step*arange + start, last point forced to stop)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#define ORC_API __attribute__((visibility("default")))
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static double interp_lin(const double *x, const double *y, int n, double xn)
{
    int lo = 0, hi = n;                       /* searchsorted side='left' */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (x[mid] < xn) lo = mid + 1; else hi = mid; }
    int idx = lo;
    if (idx < 1) idx = 1;
    if (idx > n - 1) idx = n - 1;
    double slope = (y[idx] - y[idx - 1]) / (x[idx] - x[idx - 1]);
    return slope * (xn - x[idx - 1]) + y[idx - 1];
}

static double simpson_x(const double *y, const double *x, int n)
{   /* scipy.integrate.simpson(y, x=x): n odd -> _basic_simpson; n even -> _basic_simpson on the first n-1 points plus the
     * last-interval correction (alpha, beta, eta; scipy 1.15 _quadrature.py, "Cartwright"); n == 2 -> trapezoid */
    if (n == 2) return 0.5 * (x[1] - x[0]) * (y[1] + y[0]);
    const int nodd = (n % 2 == 0) ? n - 1 : n;
    double res = 0.0;
    for (int i = 0; i + 2 < nodd; i += 2) {
        double h0 = x[i + 1] - x[i], h1 = x[i + 2] - x[i + 1];
        double hsum = h0 + h1, hprod = h0 * h1;
        double h0divh1 = (h1 != 0) ? h0 / h1 : 0.0;
        double inv = (h0divh1 != 0) ? 1.0 / h0divh1 : 0.0;
        double hq = (hprod != 0) ? hsum / hprod : 0.0;
        res += hsum / 6.0 * (y[i] * (2.0 - inv) + y[i + 1] * (hsum * hq) + y[i + 2] * (2.0 - h0divh1));
    }
    if (n % 2 == 0) {
        const double h0 = x[n - 2] - x[n - 3], h1 = x[n - 1] - x[n - 2];
        double num = 2 * (h1 * h1) + 3 * h0 * h1, den = 6 * (h1 + h0);
        const double alpha = (den != 0) ? num / den : 0.0;
        num = h1 * h1 + 3.0 * h0 * h1; den = 6 * h0;
        const double beta = (den != 0) ? num / den : 0.0;
        num = 1 * pow(h1, 3); den = 6 * h0 * (h0 + h1);
        const double eta = (den != 0) ? num / den : 0.0;
        res += alpha * y[n - 1] + beta * y[n - 2] - eta * y[n - 3];
    }
    return res;
}

/* Returns 0 (5: NINT < 2). */
ORC_API int orc_layer_average(
    double RADIUS, int NPRO, const double *H, const double *P, const double *T, int NVMR, const double *VMR /*[NPRO][NVMR]*/,
    int NDUST, const double *DUST /*[NPRO][NDUST] or NULL*/, const double *PARAH2 /*[NPRO] or NULL*/, int NLAY,
    const double *BASEH, double LAYANG, int LAYINT, double LAYHT, int NINT, const int *DUST_UNITS /*[NDUST] or NULL*/,
    const double *XMOLWT_kg /*[NPRO] or NULL*/, double *HEIGHT, double *PRESS, double *TEMP, double *TOTAM,
    double *AMOUNT /*[NLAY][NVMR]*/, double *PP /*[NLAY][NVMR]*/, double *CONT /*[NLAY][NDUST]*/, double *FRAC, double *DELH,
    double *BASET, double *LAYSF)
{
    const double k_B = 1.38065e-23, AVOGAD = 6.02214076e23;      /* Layer_0.py:828, :36 */
    if (LAYINT == 1 && NINT < 2) return 5;
    const double sn = sin(LAYANG * M_PI / 180), cs = cos(LAYANG * M_PI / 180);
    const double z0 = RADIUS + LAYHT, zmax = RADIUS + H[NPRO - 1];
    const double SMAX = sqrt(zmax * zmax - (z0 * sn) * (z0 * sn)) - z0 * cs;
    double *BASES = (double *)malloc(sizeof(double) * NLAY), *DELS = (double *)malloc(sizeof(double) * NLAY);
    double *zero = (double *)calloc(NPRO, sizeof(double)), *molwt_g = (double *)calloc(NPRO, sizeof(double));
    if (DUST_UNITS && XMOLWT_kg) for (int i = 0; i < NPRO; ++i) molwt_g[i] = XMOLWT_kg[i] * 1000.;
    const double *parah2 = PARAH2 ? PARAH2 : zero;
    for (int i = 0; i < NLAY; ++i) BASES[i] = sqrt((RADIUS + BASEH[i]) * (RADIUS + BASEH[i]) - (z0 * sn) * (z0 * sn)) - z0 * cs;
    for (int i = 0; i < NLAY; ++i) {
        DELH[i] = (i < NLAY - 1) ? BASEH[i + 1] - BASEH[i] : H[NPRO - 1] - BASEH[NLAY - 1];
        DELS[i] = (i < NLAY - 1) ? BASES[i + 1] - BASES[i] : SMAX - BASES[NLAY - 1];
        LAYSF[i] = DELS[i] / DELH[i];
        BASET[i] = interp_lin(H, T, NPRO, BASEH[i]);
    }
    double *col = (double *)malloc(sizeof(double) * NPRO);
    if (LAYINT == 0) {
        for (int I = 0; I < NLAY; ++I) {
            double S = (I < NLAY - 1) ? (BASES[I + 1] + BASES[I]) / 2 : (SMAX + BASES[NLAY - 1]) / 2;
            double hh = sqrt(S * S + z0 * z0 + 2 * S * z0 * cs) - RADIUS;
            HEIGHT[I] = hh;
            PRESS[I] = interp_lin(H, P, NPRO, hh);
            TEMP[I] = interp_lin(H, T, NPRO, hh);
            FRAC[I] = interp_lin(H, parah2, NPRO, hh);
            double MOLWT = interp_lin(H, molwt_g, NPRO, hh);
            double DUDS = PRESS[I] / (k_B * TEMP[I]);
            TOTAM[I] = DUDS * DELS[I];
            for (int J = 0; J < NVMR; ++J) {
                for (int k = 0; k < NPRO; ++k) col[k] = VMR[(size_t)k * NVMR + J];
                double a = interp_lin(H, col, NPRO, hh);
                PP[(size_t)I * NVMR + J] = a * PRESS[I];
                AMOUNT[(size_t)I * NVMR + J] = a * TOTAM[I];
            }
            for (int J = 0; J < NDUST; ++J) {
                for (int k = 0; k < NPRO; ++k) col[k] = DUST[(size_t)k * NDUST + J];
                double DD = interp_lin(H, col, NPRO, hh);
                if (DUST_UNITS && DUST_UNITS[J] == -1) CONT[(size_t)I * NDUST + J] = DD * TOTAM[I] * MOLWT / AVOGAD;
                else CONT[(size_t)I * NDUST + J] = DD * DELS[I];
            }
        }
    } else {
        double *S = (double *)malloc(sizeof(double) * NINT), *h = (double *)malloc(sizeof(double) * NINT);
        double *p = (double *)malloc(sizeof(double) * NINT), *tt = (double *)malloc(sizeof(double) * NINT);
        double *duds = (double *)malloc(sizeof(double) * NINT), *f = (double *)malloc(sizeof(double) * NINT);
        double *mw = (double *)malloc(sizeof(double) * NINT), *a = (double *)malloc(sizeof(double) * NINT);
        for (int I = 0; I < NLAY; ++I) {
            double S0 = BASES[I], S1 = (I < NLAY - 1) ? BASES[I + 1] : SMAX;
            double step = (S1 - S0) / (NINT - 1);
            for (int k = 0; k < NINT; ++k) S[k] = k * step + S0;     /* np.linspace */
            S[NINT - 1] = S1;
            for (int k = 0; k < NINT; ++k) {
                h[k] = sqrt(S[k] * S[k] + z0 * z0 + 2 * S[k] * z0 * cs) - RADIUS;
                p[k] = interp_lin(H, P, NPRO, h[k]);
                tt[k] = interp_lin(H, T, NPRO, h[k]);
                mw[k] = interp_lin(H, molwt_g, NPRO, h[k]);
                duds[k] = p[k] / (k_B * tt[k]);
            }
            TOTAM[I] = simpson_x(duds, S, NINT);
            for (int k = 0; k < NINT; ++k) f[k] = h[k] * duds[k];
            HEIGHT[I] = simpson_x(f, S, NINT) / TOTAM[I];
            for (int k = 0; k < NINT; ++k) f[k] = p[k] * duds[k];
            PRESS[I] = simpson_x(f, S, NINT) / TOTAM[I];
            for (int k = 0; k < NINT; ++k) f[k] = tt[k] * duds[k];
            TEMP[I] = simpson_x(f, S, NINT) / TOTAM[I];
            for (int k = 0; k < NINT; ++k) f[k] = interp_lin(H, parah2, NPRO, h[k]) * duds[k];
            FRAC[I] = simpson_x(f, S, NINT) / TOTAM[I];
            for (int J = 0; J < NVMR; ++J) {
                for (int k = 0; k < NPRO; ++k) col[k] = VMR[(size_t)k * NVMR + J];
                for (int k = 0; k < NINT; ++k) { a[k] = interp_lin(H, col, NPRO, h[k]); f[k] = a[k] * duds[k]; }
                AMOUNT[(size_t)I * NVMR + J] = simpson_x(f, S, NINT);
                for (int k = 0; k < NINT; ++k) f[k] = (a[k] * p[k]) * duds[k];
                PP[(size_t)I * NVMR + J] = simpson_x(f, S, NINT) / TOTAM[I];
            }
            for (int J = 0; J < NDUST; ++J) {
                for (int k = 0; k < NPRO; ++k) col[k] = DUST[(size_t)k * NDUST + J];
                for (int k = 0; k < NINT; ++k) {
                    double dd = interp_lin(H, col, NPRO, h[k]);
                    f[k] = (DUST_UNITS && DUST_UNITS[J] == -1) ? dd * duds[k] * mw[k] / AVOGAD : dd;
                }
                CONT[(size_t)I * NDUST + J] = simpson_x(f, S, NINT);
            }
        }
        free(S); free(h); free(p); free(tt); free(duds); free(f); free(mw); free(a);
    }
    for (int I = 0; I < NLAY; ++I) {   /* scale back to vertical layers :1013-1023 */
        TOTAM[I] = TOTAM[I] / LAYSF[I];
        double inv = pow(LAYSF[I], -1);
        for (int J = 0; J < NVMR; ++J) AMOUNT[(size_t)I * NVMR + J] *= inv;
        for (int J = 0; J < NDUST; ++J) CONT[(size_t)I * NDUST + J] *= inv;
    }
    free(BASES); free(DELS); free(zero); free(molwt_g); free(col);
    return 0;
}


/* ------------------------------------------------------------------------------------------------------------
 * Layer_0.layer_averageg (Layer_0.py:1032-1398): layer_average plus the matrices DTE, DAM, DCO, DPH [NLAY][NPRO]
 * relating layer temperature / gas amount / dust amount / para-H2 fraction to the profile levels.  Differences from
 * layer_average that show in the numbers: T, PARAH2, VMR and DUST go through the reference's own `interpg` (:716-751:
 * j = clip(#{X_data <= X}, 1, n-1), F = (X - x[j-1])/(x[j]-x[j-1]), (1-F) y[j-1] + F y[j]) while P and XMOLWT keep
 * scipy's interp1d; in the MID_PATH branch CONT is recomputed with interp1d (:1228-1250).
 * Returns 0, 5 for an even NINT (the reference raises ValueError, :1189), 6 for MID_PATH with DUST_UNITS == -1 (the
 * reference's mis-indented else branch :1255-1257 assigns an array to a matrix element and raises ValueError). */
static int interpg_j(const double *x, int n, double X)
{
    int j = 0;
    while (j < n && x[j] <= X) ++j;
    if (j > n - 1) j = n - 1;
    if (j == 0) j = 1;
    return j;
}

ORC_API int orc_layer_averageg(
    double RADIUS, int NPRO, const double *H, const double *P, const double *T, int NVMR, const double *VMR, int NDUST,
    const double *DUST, const double *PARAH2, int NLAY, const double *BASEH, double LAYANG, int LAYINT, double LAYHT,
    int NINT, const int *DUST_UNITS, const double *XMOLWT_kg, double *HEIGHT, double *PRESS, double *TEMP, double *TOTAM,
    double *AMOUNT, double *PP, double *CONT, double *FRAC, double *DELH, double *BASET, double *LAYSF,
    double *DTE, double *DAM, double *DCO, double *DPH /* each [NLAY][NPRO], zeroed here */)
{
    const double k_B = 1.38065e-23, AVOGAD = 6.02214076e23;
    if ((NINT % 2) == 0) return 5;                                   /* :1188 (checked for both schemes) */
    int any_units = 0;
    if (DUST_UNITS) for (int J = 0; J < NDUST; ++J) if (DUST_UNITS[J] == -1) any_units = 1;
    if (LAYINT == 0 && any_units && NDUST > 0) return 6;
    const double sn = sin(LAYANG * M_PI / 180), cs = cos(LAYANG * M_PI / 180);
    const double z0 = RADIUS + LAYHT, zmax = RADIUS + H[NPRO - 1];
    const double SMAX = sqrt(zmax * zmax - (z0 * sn) * (z0 * sn)) - z0 * cs;
    double *BASES = (double *)malloc(sizeof(double) * NLAY), *DELS = (double *)malloc(sizeof(double) * NLAY);
    double *zero = (double *)calloc(NPRO, sizeof(double)), *molwt_g = (double *)calloc(NPRO, sizeof(double));
    if (DUST_UNITS && XMOLWT_kg) for (int i = 0; i < NPRO; ++i) molwt_g[i] = XMOLWT_kg[i] * 1000.;
    const double *parah2 = PARAH2 ? PARAH2 : zero;
    const size_t nm = (size_t)NLAY * NPRO;
    memset(DTE, 0, nm * sizeof(double)); memset(DAM, 0, nm * sizeof(double));
    memset(DCO, 0, nm * sizeof(double)); memset(DPH, 0, nm * sizeof(double));
    for (int i = 0; i < NLAY; ++i) BASES[i] = sqrt((RADIUS + BASEH[i]) * (RADIUS + BASEH[i]) - (z0 * sn) * (z0 * sn)) - z0 * cs;
    for (int i = 0; i < NLAY; ++i) {
        DELH[i] = (i < NLAY - 1) ? BASEH[i + 1] - BASEH[i] : H[NPRO - 1] - BASEH[NLAY - 1];
        DELS[i] = (i < NLAY - 1) ? BASES[i + 1] - BASES[i] : SMAX - BASES[NLAY - 1];
        LAYSF[i] = DELS[i] / DELH[i];
        BASET[i] = interp_lin(H, T, NPRO, BASEH[i]);
    }
    double *col = (double *)malloc(sizeof(double) * NPRO);
#define IG(y, stride, j, F) ((1.0 - (F)) * (y)[(size_t)((j) - 1) * (stride)] + (F) * (y)[(size_t)(j) * (stride)])
    if (LAYINT == 0) {
        for (int I = 0; I < NLAY; ++I) {
            double S = (I < NLAY - 1) ? (BASES[I + 1] + BASES[I]) / 2 : (SMAX + BASES[NLAY - 1]) / 2;
            double hh = sqrt(S * S + z0 * z0 + 2 * S * z0 * cs) - RADIUS;
            HEIGHT[I] = hh;
            PRESS[I] = interp_lin(H, P, NPRO, hh);
            const int j = interpg_j(H, NPRO, hh);
            const double F = (hh - H[j - 1]) / (H[j] - H[j - 1]);
            TEMP[I] = IG(T, 1, j, F);
            DTE[(size_t)I * NPRO + j - 1] += (1.0 - F); DTE[(size_t)I * NPRO + j] += F;
            FRAC[I] = IG(parah2, 1, j, F);
            DPH[(size_t)I * NPRO + j - 1] += (1.0 - F); DPH[(size_t)I * NPRO + j] += F;
            double DUDS = PRESS[I] / (k_B * TEMP[I]);
            TOTAM[I] = DUDS * DELS[I];
            for (int J = 0; J < NVMR; ++J) {
                double a = IG(VMR + J, NVMR, j, F);
                PP[(size_t)I * NVMR + J] = a * PRESS[I];
                AMOUNT[(size_t)I * NVMR + J] = a * TOTAM[I];
            }
            DAM[(size_t)I * NPRO + j - 1] += (1.0 - F) * TOTAM[I]; DAM[(size_t)I * NPRO + j] += F * TOTAM[I];
            for (int J = 0; J < NDUST; ++J) {
                for (int k = 0; k < NPRO; ++k) col[k] = DUST[(size_t)k * NDUST + J];
                CONT[(size_t)I * NDUST + J] = interp_lin(H, col, NPRO, hh) * DELS[I];    /* second block, interp1d :1231-1240 */
            }
            if (NDUST > 0) { DCO[(size_t)I * NPRO + j - 1] += (1.0 - F); DCO[(size_t)I * NPRO + j] += F; }
        }
    } else {
        double *S = (double *)malloc(sizeof(double) * NINT), *h = (double *)malloc(sizeof(double) * NINT);
        double *p = (double *)malloc(sizeof(double) * NINT), *tt = (double *)malloc(sizeof(double) * NINT);
        double *duds = (double *)malloc(sizeof(double) * NINT), *f = (double *)malloc(sizeof(double) * NINT);
        double *mw = (double *)malloc(sizeof(double) * NINT), *a = (double *)malloc(sizeof(double) * NINT);
        double *FF = (double *)malloc(sizeof(double) * NINT), *w = (double *)malloc(sizeof(double) * NINT);
        int *JJ = (int *)malloc(sizeof(int) * NINT);
        for (int k = 0; k < NINT; ++k) w[k] = (k == 0 || k == NINT - 1) ? 1.0 : ((k & 1) ? 4.0 : 2.0);   /* :1192-1197 */
        for (int I = 0; I < NLAY; ++I) {
            double S0 = BASES[I], S1 = (I < NLAY - 1) ? BASES[I + 1] : SMAX;
            double step = (S1 - S0) / (NINT - 1);
            for (int k = 0; k < NINT; ++k) S[k] = k * step + S0;
            S[NINT - 1] = S1;
            double *dte = DTE + (size_t)I * NPRO, *dam = DAM + (size_t)I * NPRO, *dco = DCO + (size_t)I * NPRO, *dph = DPH + (size_t)I * NPRO;
            for (int k = 0; k < NINT; ++k) {
                h[k] = sqrt(S[k] * S[k] + z0 * z0 + 2 * S[k] * z0 * cs) - RADIUS;
                p[k] = interp_lin(H, P, NPRO, h[k]);
                mw[k] = interp_lin(H, molwt_g, NPRO, h[k]);
                JJ[k] = interpg_j(H, NPRO, h[k]);
                FF[k] = (h[k] - H[JJ[k] - 1]) / (H[JJ[k]] - H[JJ[k] - 1]);
                tt[k] = IG(T, 1, JJ[k], FF[k]);
                duds[k] = p[k] / (k_B * tt[k]);
            }
            TOTAM[I] = simpson_x(duds, S, NINT);
            for (int k = 0; k < NINT; ++k) f[k] = h[k] * duds[k];
            HEIGHT[I] = simpson_x(f, S, NINT) / TOTAM[I];
            for (int k = 0; k < NINT; ++k) f[k] = p[k] * duds[k];
            PRESS[I] = simpson_x(f, S, NINT) / TOTAM[I];
            for (int k = 0; k < NINT; ++k) f[k] = tt[k] * duds[k];
            TEMP[I] = simpson_x(f, S, NINT) / TOTAM[I];
            for (int k = 0; k < NINT; ++k) f[k] = IG(parah2, 1, JJ[k], FF[k]) * duds[k];
            FRAC[I] = simpson_x(f, S, NINT) / TOTAM[I];
            for (int k = 0; k < NINT; ++k) {                                     /* :1280-1284 */
                dte[JJ[k] - 1] += (1. - FF[k]) * w[k] * duds[k]; dte[JJ[k]] += FF[k] * w[k] * duds[k];
                dph[JJ[k] - 1] += (1. - FF[k]) * w[k] * duds[k]; dph[JJ[k]] += FF[k] * w[k] * duds[k];
            }
            for (int J = 0; J < NVMR; ++J) {
                for (int k = 0; k < NINT; ++k) { a[k] = IG(VMR + J, NVMR, JJ[k], FF[k]); f[k] = a[k] * duds[k]; }
                AMOUNT[(size_t)I * NVMR + J] = simpson_x(f, S, NINT);
                for (int k = 0; k < NINT; ++k) f[k] = (a[k] * p[k]) * duds[k];
                PP[(size_t)I * NVMR + J] = simpson_x(f, S, NINT) / TOTAM[I];
            }
            for (int k = 0; k < NINT; ++k) {                                     /* :1303-1305 */
                dam[JJ[k] - 1] += (1. - FF[k]) * duds[k] * w[k]; dam[JJ[k]] += FF[k] * duds[k] * w[k];
            }
            for (int J = 0; J < NDUST; ++J) {
                for (int k = 0; k < NINT; ++k) {
                    double dd = IG(DUST + J, NDUST, JJ[k], FF[k]);
                    f[k] = (DUST_UNITS && DUST_UNITS[J] == -1) ? dd * duds[k] * mw[k] / AVOGAD : dd;
                }
                CONT[(size_t)I * NDUST + J] = simpson_x(f, S, NINT);
            }
            if (NDUST > 0) {
                for (int k = 0; k < NINT; ++k) {                                 /* :1336-1343 */
                    if (!any_units) { dco[JJ[k] - 1] += (1. - FF[k]) * w[k]; dco[JJ[k]] += FF[k] * w[k]; }
                    else {
                        dco[JJ[k] - 1] += (1. - FF[k]) * w[k] * duds[k] * mw[k] / AVOGAD;
                        dco[JJ[k]] += FF[k] * w[k] * duds[k] * mw[k] / AVOGAD;
                    }
                }
            }
        }
        for (int I = 0; I < NLAY; ++I)                                           /* :1346-1350 */
            for (int k = 0; k < NPRO; ++k) {
                const size_t o = (size_t)I * NPRO + k;
                DTE[o] = DTE[o] * DELS[I] / (NINT - 1.) / 3. / TOTAM[I];
                DPH[o] = DPH[o] * DELS[I] / (NINT - 1.) / 3. / TOTAM[I];
                DAM[o] = DAM[o] * DELS[I] / (NINT - 1.) / 3.;
                DCO[o] = DCO[o] * DELS[I] / (NINT - 1.) / 3.;
            }
        free(S); free(h); free(p); free(tt); free(duds); free(f); free(mw); free(a); free(FF); free(w); free(JJ);
    }
#undef IG
    for (int I = 0; I < NLAY; ++I) {
        TOTAM[I] = TOTAM[I] / LAYSF[I];
        double inv = pow(LAYSF[I], -1);
        for (int J = 0; J < NVMR; ++J) AMOUNT[(size_t)I * NVMR + J] *= inv;
        for (int J = 0; J < NDUST; ++J) CONT[(size_t)I * NDUST + J] *= inv;
        for (int k = 0; k < NPRO; ++k) { DAM[(size_t)I * NPRO + k] /= LAYSF[I]; DCO[(size_t)I * NPRO + k] /= LAYSF[I]; }
    }
    free(BASES); free(DELS); free(zero); free(molwt_g); free(col);
    return 0;
}
