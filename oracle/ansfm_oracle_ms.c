/*
 * ansfm_oracle_ms.c -- CPU restatement of the doubling/adding multiple-scattering core.
 * TEST INFRASTRUCTURE ONLY (see ansfm_oracle.c).  Follows
 *   archnemesis/Multiple_Scattering_Core.py: phasint2 :141, hansen :200, calc_pmat6 :236,
 *   add :275, double1 :321, addp :481, angle_quadrature :535, calc_rtj_matrix :566,
 *   scloud11wave_core :651-960 (look-down geometry; look-up raises UNSUPPORTED)
 * in IEEE double (the reference's numba build uses fastmath; the un-jitted Python that pins this file
 * does not).  np.linalg.inv is restated as Gauss-Jordan with partial pivoting (LAPACK differs at
 * rounding level).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#define ORC_API __attribute__((visibility("default")))
#define MAXMU 32
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef double mat[MAXMU * MAXMU];

static void mm_mul(int n, const double *A, const double *B, double *C)
{ /* matmul :247 -- i,j,k order, += into zeros */
    double T[MAXMU * MAXMU];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[i * n + k] * B[k * n + j];
            T[i * n + j] = s;
        }
    memcpy(C, T, sizeof(double) * n * n);
}
static void mv_mul(int n, const double *A, const double *x, double *y)
{
    double T[MAXMU];
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += A[i * n + k] * x[k];
        T[i] = s;
    }
    memcpy(y, T, sizeof(double) * n);
}
static double frob(int n, const double *r)
{
    double s = 0.0;
    for (int i = 0; i < n * n; ++i) s += r[i] * r[i];
    return sqrt(s);
}
static void mat_inv(int n, const double *A, double *Ainv)
{
    double a[MAXMU * MAXMU], b[MAXMU * MAXMU];
    memcpy(a, A, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) b[i * n + j] = (i == j);
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(a[c * n + c]);
        for (int r = c + 1; r < n; ++r) if (fabs(a[r * n + c]) > best) { best = fabs(a[r * n + c]); piv = r; }
        if (piv != c)
            for (int j = 0; j < n; ++j) {
                double t = a[c * n + j]; a[c * n + j] = a[piv * n + j]; a[piv * n + j] = t;
                t = b[c * n + j]; b[c * n + j] = b[piv * n + j]; b[piv * n + j] = t;
            }
        double d = 1.0 / a[c * n + c];
        for (int j = 0; j < n; ++j) { a[c * n + j] *= d; b[c * n + j] *= d; }
        for (int r = 0; r < n; ++r) if (r != c) {
            double f = a[r * n + c];
            if (f != 0.0) for (int j = 0; j < n; ++j) { a[r * n + j] -= f * a[c * n + j]; b[r * n + j] -= f * b[c * n + j]; }
        }
    }
    memcpy(Ainv, b, sizeof(double) * n * n);
}

static double np_interp(double x, const double *xp, const double *fp, int n)
{
    if (x <= xp[0]) return fp[0];
    if (x >= xp[n - 1]) return fp[n - 1];
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid; else hi = mid; }
    double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
    return slope * (x - xp[lo]) + fp[lo];
}

/* phasint2 :141-197.  iscat 0 Rayleigh, 2 Henyey-Greenstein (pfunc[0..2] = f,g1,g2), else tabulated. */
static void phasint2(int nphi, int ic, int nmu, const double *mu, int iscat, const double *pfunc,
                     const double *xmu, int nth, double *pplpl, double *pplmi)
{
    const double pi = M_PI, dphi = 2.0 * pi / nphi;
    for (int i = 0; i < nmu; ++i)
        for (int j = 0; j < nmu; ++j) {
            double sthi = sqrt(1.0 - mu[i] * mu[i]), sthj = sqrt(1.0 - mu[j] * mu[j]);
            double ss = sthi * sthj, mmu = mu[i] * mu[j];
            double spl = 0.0, smi = 0.0;
            for (int k = 0; k <= nphi; ++k) {
                double phi = k * dphi;
                double cpl = ss * cos(phi) + mmu, cmi = ss * cos(phi) - mmu;
                double pl, pm;
                if (iscat == 0) {
                    pl = 0.75 * (1.0 + cpl * cpl) / (4 * pi);
                    pm = 0.75 * (1.0 + cmi * cmi) / (4 * pi);
                } else if (iscat == 2) {
                    double f1 = pfunc[0], f2 = 1.0 - f1;
                    double hg11 = 1.0 - pfunc[1] * pfunc[1], hg12 = 2.0 - hg11;
                    double hg21 = 1.0 - pfunc[2] * pfunc[2], hg22 = 2.0 - hg21;
                    pl = f1 * hg11 / pow(sqrt(hg12 - 2.0 * pfunc[1] * cpl), 3) + f2 * hg21 / pow(sqrt(hg22 - 2.0 * pfunc[2] * cpl), 3);
                    pm = f1 * hg11 / pow(sqrt(hg12 - 2.0 * pfunc[1] * cmi), 3) + f2 * hg21 / pow(sqrt(hg22 - 2.0 * pfunc[2] * cmi), 3);
                    pl /= 4 * pi; pm /= 4 * pi;
                } else {
                    pl = np_interp(cpl, xmu, pfunc, nth);
                    pm = np_interp(cmi, xmu, pfunc, nth);
                }
                double wphi = (k == 0 || k == nphi) ? 0.5 * dphi : dphi;
                if (ic == 0) wphi /= (2.0 * pi); else wphi /= pi;
                double cic = cos(ic * phi);
                spl += wphi * (pl * cic);
                smi += wphi * (pm * cic);
            }
            pplpl[i * nmu + j] = spl;
            pplmi[i * nmu + j] = smi;
        }
}

/* hansen :200-233 (fc updated in place, persists across calls) */
static void hansen(int ic, double *ppl, const double *pmi, const double *wtmu, int nmu, double *fc)
{
    const double x1 = 2.0 * M_PI;
    if (ic == 0) {
        double rsum[MAXMU], tsum[MAXMU];
        for (int j = 0; j < nmu; ++j) {
            double s = 0.0;
            for (int i = 0; i < nmu; ++i) s += pmi[i * nmu + j] * wtmu[i];
            rsum[j] = s * x1;
        }
        for (int niter = 0; niter < 10000; ++niter) {
            double test = 0.0;
            for (int j = 0; j < nmu; ++j) {
                double s = 0.0;
                for (int i = 0; i < nmu; ++i) s += ppl[i * nmu + j] * wtmu[i] * fc[i * nmu + j];
                tsum[j] = s * x1;
                double t = fabs(rsum[j] + tsum[j] - 1.0);
                if (t > test) test = t;
            }
            if (test < 1e-14) break;
            for (int j = 0; j < nmu; ++j) {
                double xj = (1.0 - rsum[j]) / tsum[j];
                for (int i = 0; i <= j; ++i) {
                    double xi = (1.0 - rsum[i]) / tsum[i];
                    fc[i * nmu + j] = 0.5 * (fc[i * nmu + j] * xj + fc[j * nmu + i] * xi);
                    fc[j * nmu + i] = fc[i * nmu + j];
                }
            }
        }
    }
    for (int i = 0; i < nmu * nmu; ++i) ppl[i] *= fc[i];
}

/* add :275-297 */
static void add_layer(int n, double *r1, double *t1, double *j1, const double *e, int ic)
{
    mat bcom, acom, ccom, rans, tans;
    double jcom[MAXMU], jans[MAXMU];
    mm_mul(n, r1, r1, bcom);
    if (frob(n, r1) > 0.1) {
        mat d;
        for (int i = 0; i < n * n; ++i) d[i] = e[i] - bcom[i];
        mat_inv(n, d, acom);
    } else
        for (int i = 0; i < n * n; ++i) acom[i] = e[i] + bcom[i];
    mm_mul(n, t1, acom, ccom);
    mm_mul(n, ccom, r1, rans);
    mm_mul(n, rans, t1, acom);
    for (int i = 0; i < n * n; ++i) rans[i] = r1[i] + acom[i];
    mm_mul(n, ccom, t1, tans);
    if (ic == 0) {
        mv_mul(n, r1, j1, jcom);
        for (int i = 0; i < n; ++i) jcom[i] = jcom[i] + j1[i];
        mv_mul(n, ccom, jcom, jans);
        for (int i = 0; i < n; ++i) jans[i] = jans[i] + j1[i];
    } else
        memcpy(jans, j1, sizeof(double) * n);
    memcpy(r1, rans, sizeof(double) * n * n);
    memcpy(t1, tans, sizeof(double) * n * n);
    memcpy(j1, jans, sizeof(double) * n);
}

/* double1 :321-362 */
static void double1(int ic, int n, const double *cc, const double *pplpl, const double *pplmi, double omega,
                    double taut, double bc, const double *mminv, const double *e, double *r1, double *t1, double *j1)
{
    const int ipow0 = 12;
    double con = omega * M_PI;
    double del01 = (ic == 0) ? 1.0 : 0.0;
    con *= (1.0 + del01);
    mat acom, bcom, gplpl, gplmi;
    mm_mul(n, pplpl, cc, acom);
    for (int i = 0; i < n * n; ++i) { acom[i] *= con; bcom[i] = e[i] - acom[i]; }
    mm_mul(n, mminv, bcom, gplpl);
    mm_mul(n, pplmi, cc, acom);
    for (int i = 0; i < n * n; ++i) acom[i] *= con;
    mm_mul(n, mminv, acom, gplmi);
    int nn = (int)(log2(taut) + ipow0);                 /* python int(): truncation toward zero */
    double xfac = (nn >= 1) ? 1.0 / pow(2.0, nn) : 1.0;
    double tau0 = taut * xfac;
    for (int i = 0; i < n * n; ++i) t1[i] = e[i] - tau0 * gplpl[i];
    mat eg;
    mm_mul(n, e, gplmi, eg);
    for (int i = 0; i < n * n; ++i) r1[i] = tau0 * eg[i];
    for (int i = 0; i < n; ++i) j1[i] = (ic == 0) ? (1.0 - omega) * bc * tau0 * mminv[i * n + i] : 0.0;
    if (nn < 1) return;
    for (int k = 0; k < nn; ++k) add_layer(n, r1, t1, j1, e, ic);
}

/* calc_rtj_matrix :566-647 */
static int calc_rtj(int ic, int n, const double *bbp, double tautot, double tauscat, double tauray,
                    const double *frac, int ncont, const double *ppln, const double *pmin, const double *pplr,
                    const double *pmir, const double *cc, const double *mminv, const double *e, double *rl, double *tl,
                    double *jl)
{
    const double bb = *bbp;
    double omega = (tauscat + tauray) / tautot;
    memset(rl, 0, sizeof(double) * n * n); memset(tl, 0, sizeof(double) * n * n); memset(jl, 0, sizeof(double) * n);
    if (tautot == 0) {
        for (int i = 0; i < n; ++i) tl[i * n + i] = 1.0;
        return 0;
    } else if (omega == 0) {
        for (int i = 0; i < n; ++i) {
            double tex = -mminv[i * n + i] * tautot;
            tl[i * n + i] = (tex > -200.0) ? exp(tex) : 0.0;
            jl[i] = bb * (1.0 - tl[i * n + i]);
        }
        return 0;
    }
    mat pplpl, pplmi;
    double fr = tauray / (tauscat + tauray), fs = tauscat / (tauscat + tauray);
    for (int i = 0; i < n * n; ++i) { pplpl[i] = fr * pplr[i]; pplmi[i] = fr * pmir[i]; }
    for (int j1 = 0; j1 < ncont; ++j1)
        for (int i = 0; i < n * n; ++i) {
            pplpl[i] += fs * ppln[j1 * n * n + i] * frac[j1];
            pplmi[i] += fs * pmin[j1 * n * n + i] * frac[j1];
        }
    double1(ic, n, cc, pplpl, pplmi, omega, tautot, bb, mminv, e, rl, tl, jl);
    return 1;
}

/* addp :481-532: new layer (r1,t1,j1,iscat1) on top of the combination (rsub,tsub,jsub) -> in place */
static void addp(int n, const double *r1, const double *t1, const double *j1, int iscat1, const double *e,
                 double *rsub, double *tsub, double *jsub)
{
    mat rans, tans;
    double jans[MAXMU], jcom[MAXMU];
    if (iscat1 == 1) {
        mat rsq, acom, ccom, bcom;
        mm_mul(n, rsub, r1, rsq);
        if (frob(n, rsq) > 0.01) {
            mat d;
            for (int i = 0; i < n * n; ++i) d[i] = e[i] - rsq[i];
            mat_inv(n, d, acom);
        } else
            for (int i = 0; i < n * n; ++i) acom[i] = e[i] + rsq[i];
        mm_mul(n, t1, acom, ccom);
        mm_mul(n, ccom, rsub, rans);
        mm_mul(n, rans, t1, bcom);
        for (int i = 0; i < n * n; ++i) rans[i] = r1[i] + bcom[i];
        mm_mul(n, ccom, tsub, tans);
        mv_mul(n, rsub, j1, jcom);
        for (int i = 0; i < n; ++i) jcom[i] += jsub[i];
        mv_mul(n, ccom, jcom, jans);
        for (int i = 0; i < n; ++i) jans[i] += j1[i];
    } else {
        mv_mul(n, rsub, j1, jcom);
        for (int i = 0; i < n; ++i) jcom[i] += jsub[i];
        for (int i = 0; i < n; ++i) {
            double ta = t1[i * n + i];
            for (int j = 0; j < n; ++j) {
                double tb = t1[j * n + j];
                tans[i * n + j] = tsub[i * n + j] * ta;
                rans[i * n + j] = rsub[i * n + j] * ta * tb;
            }
            jans[i] = j1[i] + ta * jcom[i];
        }
    }
    memcpy(rsub, rans, sizeof(double) * n * n);
    memcpy(tsub, tans, sizeof(double) * n * n);
    memcpy(jsub, jans, sizeof(double) * n);
}

/* scloud11wave_core :651-960.  Array layouts are the reference's.  Returns 0, or 1 (mixed emission angles,
 * the reference raises ValueError :776).  Both geometries: look-down (all emission angles < 90: layers bottom
 * to top, surface first) and look-up (all > 90: layers top to bottom, surface kept apart and brought in with
 * idown :366-420 when lowbc > 0). */
ORC_API int orc_scloud11wave_core(
    int ncont, int nwave, int nth, const double *phasarr /*[ncont][nwave][2][nth]*/, const double *radg_in /*[nwave][nmu]*/,
    int ngeom, const double *sol_angs, const double *emiss_angs, const double *solar /*[nwave]*/, const double *aphis,
    int lowbc, const double *brdf /*[nwave][nmu][nmu][nf+1]*/, int nmu, const double *mu1, const double *wt1, int nf,
    const double *bnu /*[nwave][nlay]*/, int ng, int nlay, const double *taus /*[nwave][ng][nlay]*/,
    const double *tauray /*[nwave][nlay]*/, const double *omegas_s /*[nwave][ng][nlay]*/, int nphi, int iray, int imie,
    const double *lfrac /*[nwave][ncont][nlay]*/, double *rad /*[ngeom][ng][nwave]*/)
{
    if (nmu > MAXMU) return 1;
    int nless = 0, nmore = 0;
    for (int i = 0; i < ngeom; ++i) { if (emiss_angs[i] < 90) ++nless; if (emiss_angs[i] > 90) ++nmore; }
    if (nless != ngeom && nmore != ngeom) return 1;
    const int lookdown = (nless == ngeom);
    const int n = nmu, NF1 = nf + 1;
    double xfac = 0.0;
    for (int i = 0; i < n; ++i) xfac += mu1[i] * wt1[i];
    xfac = 0.5 / xfac;
    double mu[MAXMU], wtmu[MAXMU];
    for (int i = 0; i < n; ++i) { mu[i] = mu1[n - 1 - i]; wtmu[i] = wt1[n - 1 - i]; }
    mat e, mminv, cc;
    memset(e, 0, sizeof e); memset(mminv, 0, sizeof mminv); memset(cc, 0, sizeof cc);
    for (int i = 0; i < n; ++i) { e[i * n + i] = 1.0; mminv[i * n + i] = 1. / mu[i]; cc[i * n + i] = wtmu[i]; }
    double *fc = (double *)malloc(sizeof(double) * (ncont + 1) * n * n);
    for (int i = 0; i < (ncont + 1) * n * n; ++i) fc[i] = 1.0;
    double *ppln = (double *)malloc(sizeof(double) * (ncont > 0 ? ncont : 1) * n * n);
    double *pmin = (double *)malloc(sizeof(double) * (ncont > 0 ? ncont : 1) * n * n);
    double *rcomb = (double *)malloc(sizeof(double) * NF1 * n * n);
    double *tcomb = (double *)malloc(sizeof(double) * NF1 * n * n);
    double *jcomb = (double *)malloc(sizeof(double) * NF1 * n);
    double *rsurf = (double *)calloc((size_t)NF1 * n * n, sizeof(double));      /* rs[:,:,ic]; ts = 0; js = radg (every ic) */
    memset(rad, 0, sizeof(double) * ngeom * ng * nwave);
    for (int ig = 0; ig < ng; ++ig)
        for (int widx = 0; widx < nwave; ++widx) {
            double radg[MAXMU];
            for (int j = 0; j < n; ++j) radg[j] = radg_in[(size_t)widx * n + (n - 1 - j)];
            for (int ic = 0; ic <= nf; ++ic) {
                mat pplr, pmir;
                memset(pplr, 0, sizeof pplr); memset(pmir, 0, sizeof pmir);
                memset(ppln, 0, sizeof(double) * (ncont > 0 ? ncont : 1) * n * n);
                memset(pmin, 0, sizeof(double) * (ncont > 0 ? ncont : 1) * n * n);
                const double *pfunc = NULL, *xmu = NULL;
                for (int j1 = 0; j1 < ncont; ++j1) {
                    pfunc = phasarr + (((size_t)j1 * nwave + widx) * 2 + 0) * nth;
                    xmu = phasarr + (((size_t)j1 * nwave + widx) * 2 + 1) * nth;
                    int iscat = (imie == 0) ? 2 : 4;
                    phasint2(nphi, ic, n, mu, iscat, pfunc, xmu, nth, ppln + (size_t)j1 * n * n, pmin + (size_t)j1 * n * n);
                    hansen(ic, ppln + (size_t)j1 * n * n, pmin + (size_t)j1 * n * n, wtmu, n, fc + (size_t)j1 * n * n);
                }
                if (iray > 0) {
                    phasint2(nphi, ic, n, mu, 0, pfunc, xmu, nth, pplr, pmir);
                    hansen(ic, pplr, pmir, wtmu, n, fc + (size_t)ncont * n * n);
                }
                double *rc = rcomb + (size_t)ic * n * n, *tc = tcomb + (size_t)ic * n * n, *jc = jcomb + (size_t)ic * n;
                int surface_defined = 0;
                if (lowbc > 0) {   /* :822-836: look-down puts the surface first in the stack */
                    double *rs = rsurf + (size_t)ic * n * n;
                    for (int i = 0; i < n; ++i)
                        for (int j = 0; j < n; ++j)
                            rs[i * n + j] = (2. * (brdf[(((size_t)widx * n + i) * n + j) * NF1 + ic] * M_PI) * mu[j] * wtmu[j]) * xfac;
                    if (lookdown) {
                        for (int i = 0; i < n; ++i) jc[i] = radg[i];
                        memcpy(rc, rs, sizeof(double) * n * n);
                        memset(tc, 0, sizeof(double) * n * n);
                        surface_defined = 1;
                    }
                }
                for (int l = 0; l < nlay; ++l) {
                    int k = lookdown ? l : nlay - 1 - l;                       /* :841-844 */
                    double taut = taus[((size_t)widx * ng + ig) * nlay + k];
                    double bc = bnu[(size_t)widx * nlay + k];
                    double omega = omegas_s[((size_t)widx * ng + ig) * nlay + k];
                    if (omega < 0) omega = 0.0;
                    if (omega > 1) omega = 1.0;
                    double tauscat = taut * omega;
                    double taur = tauray[(size_t)widx * nlay + k];
                    tauscat = tauscat - taur;
                    if (tauscat < 0) tauscat = 0.0;
                    double frac[64];
                    for (int j1 = 0; j1 < ncont; ++j1) frac[j1] = lfrac[((size_t)widx * ncont + j1) * nlay + k];
                    mat rl, tl; double jl[MAXMU];
                    int iscl = calc_rtj(ic, n, &bc, taut, tauscat, taur, frac, ncont, ppln, pmin, pplr, pmir, cc, mminv, e, rl, tl, jl);
                    if (l == 0 && !surface_defined) {
                        memcpy(rc, rl, sizeof(double) * n * n); memcpy(tc, tl, sizeof(double) * n * n);
                        memcpy(jc, jl, sizeof(double) * n);
                    } else
                        addp(n, rl, tl, jl, iscl, e, rc, tc, jc);
                }
                if (ic != 0) memset(jc, 0, sizeof(double) * n);
            }
            for (int ipath = 0; ipath < ngeom; ++ipath) {
                int conv1 = 0;
                const double defconv = 1e-5;
                double sol_ang = sol_angs[ipath], emiss_ang = emiss_angs[ipath], aphi = aphis[ipath];
                /* angle_quadrature :535-563 */
                double zmu0, solar1;
                if (sol_ang > 90.0) { zmu0 = cos((180 - sol_ang) * M_PI / 180.0); solar1 = solar[widx] * 0.0; }
                else { zmu0 = cos(sol_ang * M_PI / 180.0); solar1 = solar[widx]; }
                if (!lookdown) emiss_ang = 180. - emiss_ang;                   /* new_emi :900-903 */
                double zmu = cos(emiss_ang * M_PI / 180.0);
                int isol = 0, iemm = 0;
                for (int j = 0; j < n - 1; ++j) if (zmu0 <= mu[j] && zmu0 > mu[j + 1]) isol = j;
                if (zmu0 <= mu[n - 1]) isol = n - 2;
                for (int j = 0; j < n - 1; ++j) if (zmu <= mu[j] && zmu > mu[j + 1]) iemm = j;
                if (zmu <= mu[n - 1]) iemm = n - 2;
                double u = (mu[isol] - zmu0) / (mu[isol] - mu[isol + 1]);
                double t = (mu[iemm] - zmu) / (mu[iemm] - mu[iemm + 1]);
                double *radp = rad + ((size_t)ipath * ng + ig) * nwave + widx;
                for (int ic = 0; ic <= nf; ++ic) {
                    const double *rc = rcomb + (size_t)ic * n * n, *tc = tcomb + (size_t)ic * n * n, *jc = jcomb + (size_t)ic * n;
                    double u0pl[MAXMU], utmi[MAXMU], yx[4];
                    for (int j = 0; j < n; ++j) { u0pl[j] = 0.0; utmi[j] = (ic == 0) ? radg[j] : 0.0; }
                    int ico = 0;
                    for (int imu0 = isol; imu0 < isol + 2; ++imu0) {
                        u0pl[imu0] = solar1 / (2.0 * M_PI * wtmu[imu0]);
                        double acom[MAXMU], bcom[MAXMU], upl[MAXMU];
                        if (lookdown) {
                            mv_mul(n, rc, u0pl, acom);
                            mv_mul(n, tc, utmi, bcom);
                            for (int i = 0; i < n; ++i) upl[i] = (acom[i] + bcom[i]) + jc[i];
                        } else if (lowbc == 0) {                               /* :929-933 */
                            mv_mul(n, tc, u0pl, acom);
                            mv_mul(n, rc, utmi, bcom);
                            for (int i = 0; i < n; ++i) upl[i] = (acom[i] + bcom[i]) + jc[i];
                        } else {                                               /* idown :366-420, rb = rs, tb = 0, jb = radg */
                            const double *rs = rsurf + (size_t)ic * n * n;
                            mat ac, em, bc2, zero;
                            double xcom[MAXMU], ycom[MAXMU];
                            mm_mul(n, rc, rs, ac);
                            for (int i = 0; i < n * n; ++i) em[i] = e[i] - ac[i];
                            mat_inv(n, em, bc2);
                            mv_mul(n, tc, u0pl, xcom);
                            memset(zero, 0, sizeof zero);
                            mm_mul(n, rc, zero, ac);                            /* R10*T21 with T21 = 0 */
                            mv_mul(n, ac, utmi, ycom);
                            for (int i = 0; i < n; ++i) xcom[i] += ycom[i];
                            mv_mul(n, rc, radg, ycom);                          /* R10*J21-, js = radg for every ic (:822) */
                            for (int i = 0; i < n; ++i) xcom[i] += ycom[i] + jc[i];
                            mv_mul(n, bc2, xcom, upl);
                        }
                        for (int imu = iemm; imu < iemm + 2; ++imu) {
                            yx[ico++] = upl[imu];
                            u0pl[imu0] = 0.0;
                        }
                    }
                    double drad = ((1 - t) * (1 - u) * yx[0] + t * (1 - u) * yx[1] + t * u * yx[3] + (1 - t) * u * yx[2]) *
                                  cos(ic * aphi * M_PI / 180.0);
                    if (ic > 0) drad *= 2;
                    *radp += drad;
                    double conv = fabs(drad / *radp);
                    if (conv < defconv && conv1) break;
                    conv1 = (conv < defconv);
                }
            }
        }
    free(fc); free(ppln); free(pmin); free(rcomb); free(tcomb); free(jcomb); free(rsurf);
    return 0;
}
