/*
 * ansfm_oracle.c -- CPU restatement of the archNEMESIS correlated-k thermal-emission hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle (and the "port" CPU baseline timed
 * by bench.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  The product path (archnemesis_dist_amd/csrc/ HIP sources behind include/ansfm.h) never calls it.
 *
 * Every function follows the reference loop nest it cites (paths relative to the reference
 * tree), in the same arithmetic order, in IEEE double.  Pinned against the reference itself
 * (imported un-jitted in the build container) by oracle/gen_golden.py -> tests/golden/ npz files and
 * tests/test_oracle_golden.py.
 *
 * Array layouts are the reference's (row-major / C order):
 *   K      [W][G][NP][NT][S]      Spectroscopy_0.K                 Spectroscopy_0.py:213
 *   k_gas  [W][G][L][S]           calc_k / calc_kg output          Spectroscopy_0.py:2329
 *   amount [S][L]                 f_gas                            ForwardModel_0.py:3857-3861
 *   tau    [W][G][L]              k_overlap output                 ForwardModel_0.py:6055
 *   dk     [W][G][L][S+1]         k_overlapg output                ForwardModel_0.py:5869
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                     */
/* ------------------------------------------------------------------------------------------ */

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* dtype semantics.  When Spectroscopy_0.PRESS/TEMP (grid) or DELG are float32 arrays -- they are
 * when the tables come from .kta files (read_ktahead, Spectroscopy_0.py:2544-2559) -- NumPy (and
 * numba) evaluate np.log(PRESS[i]), phi-plo, thi-tlo, 1./(thi-tlo), del_g[i]*del_g[j] and
 * np.cumsum(del_g) in float32.  The product/cumsum effect on tau is 4e-5, so it is restated.
 * float32 log is taken as the correctly rounded value; NumPy's SIMD float32 log differs from it by
 * 1 ulp for ~6 % of arguments (a 2e-7 effect on k the reference itself does not pin). */
static int g_grid_f32 = 0, g_delg_f32 = 0;
ORC_API void orc_set_f32_semantics(int grid_f32, int delg_f32)
{
    g_grid_f32 = grid_f32;
    g_delg_f32 = delg_f32;
}
static inline double logx(double x) { return g_grid_f32 ? (double)(float)log(x) : log(x); }
static inline double wprod(double a, double b) { return g_delg_f32 ? (double)(float)(a * b) : a * b; }

/* argsort ascending by value; ties broken by original index (deterministic).  numpy's default
 * argsort (introsort) is not stable, so tie order may differ from the reference; tied keys have
 * equal `cont`, so only rounding-level differences can result (ForwardModel_0.py:6147). */
static void argsort_f64(const double *v, int n, int *idx, int *tmp)
{
    /* bottom-up merge sort on indices */
    for (int i = 0; i < n; ++i) idx[i] = i;
    /* insertion sort runs of 8 */
    for (int s = 0; s < n; s += 8) {
        int e = s + 8 < n ? s + 8 : n;
        for (int i = s + 1; i < e; ++i) {
            int x = idx[i];
            double vx = v[x];
            int j = i - 1;
            while (j >= s && (v[idx[j]] > vx)) { idx[j + 1] = idx[j]; --j; }
            idx[j + 1] = x;
        }
    }
    int *src = idx, *dst = tmp;
    for (int width = 8; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            int mid = lo + width < n ? lo + width : n;
            int hi = lo + 2 * width < n ? lo + 2 * width : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (v[src[j]] < v[src[i]]) dst[k++] = src[j++];
                else dst[k++] = src[i++];
            }
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        int *t = src; src = dst; dst = t;
    }
    if (src != idx) memcpy(idx, src, (size_t)n * sizeof(int));
}

/* ------------------------------------------------------------------------------------------ */
/* K1: (P,T) interpolation of the k-table      Spectroscopy_0.calc_k  :2298-2437               */
/*                                             Spectroscopy_0.calc_kg :2147-2295               */
/* ------------------------------------------------------------------------------------------ */

/* Nearest-then-bracket choice of the four table corners for one layer.
 * Follows Spectroscopy_0.py:2336-2371 (calc_k) == :2181-2218 (calc_kg): argmin(|grid-x|) takes the
 * FIRST minimum; below/above the grid the value is clamped to the edge. */
ORC_API void orc_bracket(int NP, const double *PRESS, int NT, const double *TEMP, double press1,
                         double temp1, int *ipl, int *iph, int *itl, int *ith, double *v, double *u,
                         double *dudt)
{
    int ip = 0;
    double best = fabs(PRESS[0] - press1);
    for (int i = 1; i < NP; ++i) {
        double d = fabs(PRESS[i] - press1);
        if (d < best) { best = d; ip = i; }
    }
    int ip_low = 0, ip_high = 0, pclamp = 0;
    if (PRESS[ip] >= press1) {
        ip_high = ip;
        if (ip == 0) { press1 = PRESS[0]; ip_low = 0; ip_high = 1; pclamp = 1; }
        else ip_low = ip - 1;
    } else {
        ip_low = ip;
        if (ip == NP - 1) { press1 = PRESS[NP - 1]; ip_high = NP - 1; ip_low = NP - 2; pclamp = 1; }
        else ip_high = ip + 1;
    }
    int it = 0;
    best = fabs(TEMP[0] - temp1);
    for (int i = 1; i < NT; ++i) {
        double d = fabs(TEMP[i] - temp1);
        if (d < best) { best = d; it = i; }
    }
    int it_low = 0, it_high = 0, tclamp = 0;
    if (TEMP[it] >= temp1) {
        it_high = it;
        if (it == 0) { temp1 = TEMP[0]; it_low = 0; it_high = 1; tclamp = 1; }
        else it_low = it - 1;
    } else {
        it_low = it;
        if (it == NT - 1) { temp1 = TEMP[NT - 1]; it_high = NT - 1; it_low = NT - 2; tclamp = 1; }
        else it_high = it + 1;
    }
    /* a clamped press1 is PRESS[0|NP-1] itself, i.e. a float32 scalar whose log is float32 too */
    double lpress = pclamp ? logx(press1) : log(press1);
    double plo = logx(PRESS[ip_low]);
    double phi = logx(PRESS[ip_high]);
    double tlo = TEMP[it_low];
    double thi = TEMP[it_high];
    double pden = g_grid_f32 ? (double)((float)phi - (float)plo) : phi - plo;
    double tden = g_grid_f32 ? (double)((float)thi - (float)tlo) : thi - tlo;
    *v = (lpress - plo) / pden;
    *u = (temp1 - tlo) / tden;
    if (g_grid_f32 && pclamp) *v = (double)(((float)lpress - (float)plo) / (float)pden);   /* all-float32 expression */
    if (g_grid_f32 && tclamp) *u = (double)(((float)temp1 - (float)tlo) / (float)tden);
    *dudt = g_grid_f32 ? (double)(1.0f / (float)tden) : 1. / tden; /* python float / float32 -> float32 */
    *ipl = ip_low; *iph = ip_high; *itl = it_low; *ith = it_high;
}

/* k_gas[W][G][L][S] (+ dkdT when non-NULL).  Spectroscopy_0.py:2373-2403 / :2220-2247.
 * "good" = all four corners > 0 -> exp of bilinear-in-log; "bad" = all four <= 0 -> linear;
 * mixed sign -> stays 0. */
ORC_API void orc_calc_k(int W, int G, int NP, int NT, int S, const double *K, const double *PRESS,
                        const double *TEMP, int L, const double *press, const double *temp,
                        double *k_out, double *dkdT_out)
{
    for (int l = 0; l < L; ++l) {
        int ipl, iph, itl, ith;
        double v, u, dudt;
        orc_bracket(NP, PRESS, NT, TEMP, press[l], temp[l], &ipl, &iph, &itl, &ith, &v, &u, &dudt);
#pragma omp parallel for schedule(static)
        for (int w = 0; w < W; ++w) {
            for (int g = 0; g < G; ++g) {
                const double *base = K + (((size_t)w * G + g) * NP) * NT * S;
                for (int s = 0; s < S; ++s) {
                    double klo1 = base[((size_t)ipl * NT + itl) * S + s];
                    double klo2 = base[((size_t)ipl * NT + ith) * S + s];
                    double khi2 = base[((size_t)iph * NT + ith) * S + s];
                    double khi1 = base[((size_t)iph * NT + itl) * S + s];
                    double kk = 0.0, dk = 0.0;
                    if (klo1 > 0.0 && klo2 > 0.0 && khi1 > 0.0 && khi2 > 0.0) {
                        double l1 = log(klo1), l2 = log(klo2), h1 = log(khi1), h2 = log(khi2);
                        kk = (1.0 - v) * (1.0 - u) * l1 + v * (1.0 - u) * h1 + v * u * h2 +
                             (1.0 - v) * u * l2;
                        kk = exp(kk);
                        double dxdt = (-l1 * (1.0 - v) - h1 * v + h2 * v + l2 * (1.0 - v)) * dudt;
                        dk = kk * dxdt;
                    } else if (klo1 <= 0.0 && klo2 <= 0.0 && khi1 <= 0.0 && khi2 <= 0.0) {
                        kk = (1.0 - v) * (1.0 - u) * klo1 + v * (1.0 - u) * khi1 + v * u * khi2 +
                             (1.0 - v) * u * klo2;
                        dk = (-klo1 * (1.0 - v) - khi1 * v + khi2 * v + klo2 * (1.0 - v)) * dudt;
                    }
                    size_t o = (((size_t)w * G + g) * L + l) * S + s;
                    k_out[o] = kk;
                    if (dkdT_out) dkdT_out[o] = dk;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* K2: random-overlap merge            ForwardModel_0.rank :6117-6173, rankg :5959-6026        */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int ng, nloop, nparam;
    double *g_ord;       /* ng+2 (one guard slot) */
    int *ico, *tmp;      /* nloop */
    double *cont_s, *w_s, *gdist; /* nloop */
    double *grad_s;      /* nloop*nparam */
} rank_ws;

static void rank_ws_alloc(rank_ws *ws, int ng, int nparam)
{
    ws->ng = ng; ws->nloop = ng * ng; ws->nparam = nparam;
    ws->g_ord = (double *)malloc(sizeof(double) * (ng + 2));
    ws->ico = (int *)malloc(sizeof(int) * ws->nloop);
    ws->tmp = (int *)malloc(sizeof(int) * ws->nloop);
    ws->cont_s = (double *)malloc(sizeof(double) * ws->nloop);
    ws->w_s = (double *)malloc(sizeof(double) * ws->nloop);
    ws->gdist = (double *)malloc(sizeof(double) * ws->nloop);
    ws->grad_s = nparam ? (double *)malloc(sizeof(double) * ws->nloop * nparam) : NULL;
}
static void rank_ws_free(rank_ws *ws)
{
    free(ws->g_ord); free(ws->ico); free(ws->tmp); free(ws->cont_s); free(ws->w_s);
    free(ws->gdist); free(ws->grad_s);
}

/* rank / rankg.  grad may be NULL (rank).  n = number of gradient columns in use (rankg's `n`).
 * k_g[ng]; dkdq[ng][nparam] fully rewritten (zeros beyond column n), like the fresh arrays the
 * reference returns. */
static void rank_core(rank_ws *ws, const double *weight, const double *cont, const double *del_g,
                      const double *grad, int n, double *k_g, double *dkdq)
{
    const int ng = ws->ng, nloop = ws->nloop, nparam = ws->nparam;
    double *g_ord = ws->g_ord;
    /* :6141-6143  g_ord = [0, cumsum(del_g)], g_ord[ng] = 1 */
    g_ord[0] = 0.0;
    double acc = 0.0;
    if (g_delg_f32) {
        float accf = 0.0f;
        for (int i = 0; i < ng; ++i) { accf += (float)del_g[i]; g_ord[i + 1] = (double)accf; }
    } else {
        for (int i = 0; i < ng; ++i) { acc += del_g[i]; g_ord[i + 1] = acc; }
    }
    g_ord[ng] = 1.0;
    g_ord[ng + 1] = INFINITY; /* guard: the reference reads past the end here (numba: no check) */

    argsort_f64(cont, nloop, ws->ico, ws->tmp);
    acc = 0.0;
    for (int i = 0; i < nloop; ++i) {
        int s = ws->ico[i];
        ws->cont_s[i] = cont[s];
        ws->w_s[i] = weight[s];
        acc += weight[s];          /* np.cumsum: sequential */
        ws->gdist[i] = acc;
        if (grad) memcpy(ws->grad_s + (size_t)i * nparam, grad + (size_t)s * nparam,
                         sizeof(double) * nparam);
    }
    for (int i = 0; i < ng; ++i) k_g[i] = 0.0;
    if (dkdq) memset(dkdq, 0, sizeof(double) * ng * nparam);

    int ig = 0;
    double sum1 = 0.0;
    for (int iloop = 0; iloop < nloop; ++iloop) {
        double w = ws->w_s[iloop];
        double cw = ws->cont_s[iloop] * w; /* cont_weight */
        if (ig < ng && ws->gdist[iloop] < g_ord[ig + 1]) {
            k_g[ig] = k_g[ig] + cw;
            if (grad) for (int p = 0; p < n; ++p)
                dkdq[ig * nparam + p] += ws->grad_s[(size_t)iloop * nparam + p] * w;
            sum1 = sum1 + w;
        } else {
            if (ig >= ng) break; /* reference behaviour undefined here (out-of-range g_ord) */
            double gprev = ws->gdist[iloop == 0 ? nloop - 1 : iloop - 1]; /* python [-1] wrap */
            double frac = (g_ord[ig + 1] - gprev) / (ws->gdist[iloop] - gprev);
            k_g[ig] = k_g[ig] + frac * cw;
            if (grad) for (int p = 0; p < n; ++p)
                dkdq[ig * nparam + p] += frac * (ws->grad_s[(size_t)iloop * nparam + p] * w);
            sum1 = sum1 + frac * w;
            k_g[ig] = k_g[ig] / sum1;
            if (grad) for (int p = 0; p < n; ++p) dkdq[ig * nparam + p] = dkdq[ig * nparam + p] / sum1;
            ig = ig + 1;
            if (ig < ng) {
                sum1 = (1.0 - frac) * w;
                k_g[ig] = (1.0 - frac) * cw;
                if (grad) for (int p = 0; p < n; ++p)
                    dkdq[ig * nparam + p] = (1.0 - frac) * (ws->grad_s[(size_t)iloop * nparam + p] * w);
            }
        }
    }
    if (ig == ng - 1) {
        k_g[ig] = k_g[ig] / sum1;
        if (grad) for (int p = 0; p < n; ++p) dkdq[ig * nparam + p] = dkdq[ig * nparam + p] / sum1;
    }
}

/* Array-level seam identical to rank(weight, cont, del_g) -> k_g  (unit-test entry). */
ORC_API void orc_rank(int ng, const double *weight, const double *cont, const double *del_g,
                      double *k_g)
{
    rank_ws ws;
    rank_ws_alloc(&ws, ng, 0);
    rank_core(&ws, weight, cont, del_g, NULL, 0, k_g, NULL);
    rank_ws_free(&ws);
}

/* One (wave, layer) cell of k_overlap / k_overlapg.  k_cell[G][S] strided access via ks (stride
 * between g) -- the reference's k_g_gas = k[iwave,:,ilayer,:].  amount[S].
 * tau_g[G]; dk_g[G][S+1] when with_grad.   ForwardModel_0.py:6060-6113 / :5878-5955 */
static void overlap_cell(rank_ws *ws, int G, int S, const double *del_g, const double *k_cell,
                         size_t kstride, const double *dkdT_cell, const double *amount,
                         double *tau_g, double *dk_g, double *rw, double *rt, double *rg,
                         double *tmp_tau, double *tmp_dk)
{
    const int NP1 = S + 1;
    const int with_grad = dkdT_cell != NULL;
    const double cutoff = 0;
#define KG(g, s) k_cell[(size_t)(g) * kstride + (s)]
#define DT(g, s) dkdT_cell[(size_t)(g) * kstride + (s)]
    for (int g = 0; g < G; ++g) tau_g[g] = 0.0;
    if (with_grad) {
        memset(dk_g, 0, sizeof(double) * G * NP1);
        memset(rg, 0, sizeof(double) * G * G * NP1); /* random_grad zeros once per cell :5886 */
    }
    for (int igas = 0; igas < S - 1; ++igas) {
        if (igas == 0) {
            if (KG(G - 1, igas) * amount[igas] <= cutoff) {
                for (int g = 0; g < G; ++g) tau_g[g] = KG(g, igas + 1) * amount[igas + 1];
                if (with_grad) for (int g = 0; g < G; ++g) {
                    dk_g[g * NP1 + igas + 1] = KG(g, igas + 1);
                    dk_g[g * NP1 + igas + 2] = DT(g, igas + 1) * amount[igas + 1];
                }
            } else if (KG(G - 1, igas + 1) * amount[igas + 1] <= cutoff) {
                for (int g = 0; g < G; ++g) tau_g[g] = KG(g, igas) * amount[igas];
                if (with_grad) for (int g = 0; g < G; ++g) {
                    dk_g[g * NP1 + igas] = KG(g, igas);
                    dk_g[g * NP1 + igas + 2] = DT(g, igas) * amount[igas];
                }
            } else {
                int iloop = 0;
                for (int ig = 0; ig < G; ++ig)
                    for (int jg = 0; jg < G; ++jg) {
                        rw[iloop] = wprod(del_g[ig], del_g[jg]);
                        rt[iloop] = KG(ig, igas) * amount[igas] + KG(jg, igas + 1) * amount[igas + 1];
                        if (with_grad) {
                            rg[iloop * NP1 + igas] = KG(ig, igas);
                            rg[iloop * NP1 + igas + 1] = KG(jg, igas + 1);
                            rg[iloop * NP1 + igas + 2] =
                                DT(ig, igas) * amount[igas] + DT(jg, igas + 1) * amount[igas + 1];
                        }
                        ++iloop;
                    }
                rank_core(ws, rw, rt, del_g, with_grad ? rg : NULL, igas + 3, tmp_tau, tmp_dk);
                memcpy(tau_g, tmp_tau, sizeof(double) * G);
                if (with_grad) memcpy(dk_g, tmp_dk, sizeof(double) * G * NP1);
            }
        } else {
            if (KG(G - 1, igas + 1) * amount[igas + 1] <= cutoff) {
                if (with_grad) for (int g = 0; g < G; ++g) {
                    dk_g[g * NP1 + igas + 2] = dk_g[g * NP1 + igas + 1];
                    dk_g[g * NP1 + igas + 1] *= 0;
                }
            } else if (tau_g[G - 1] <= cutoff) {
                for (int g = 0; g < G; ++g) tau_g[g] = KG(g, igas + 1) * amount[igas + 1];
                if (with_grad) for (int g = 0; g < G; ++g) {
                    dk_g[g * NP1 + igas + 1] = KG(g, igas + 1);
                    dk_g[g * NP1 + igas + 2] = DT(g, igas + 1) * amount[igas + 1];
                }
            } else {
                int iloop = 0;
                for (int ig = 0; ig < G; ++ig)
                    for (int jg = 0; jg < G; ++jg) {
                        rw[iloop] = wprod(del_g[ig], del_g[jg]);
                        rt[iloop] = tau_g[ig] + KG(jg, igas + 1) * amount[igas + 1];
                        if (with_grad) {
                            for (int p = 0; p < igas + 1; ++p) rg[iloop * NP1 + p] = dk_g[ig * NP1 + p];
                            rg[iloop * NP1 + igas + 1] = KG(jg, igas + 1);
                            rg[iloop * NP1 + igas + 2] =
                                dk_g[ig * NP1 + igas + 1] + DT(jg, igas + 1) * amount[igas + 1];
                        }
                        ++iloop;
                    }
                rank_core(ws, rw, rt, del_g, with_grad ? rg : NULL, igas + 3, tmp_tau, tmp_dk);
                memcpy(tau_g, tmp_tau, sizeof(double) * G);
                if (with_grad) memcpy(dk_g, tmp_dk, sizeof(double) * G * NP1);
            }
        }
    }
#undef KG
#undef DT
}

/* k_overlap (dkdT == NULL, dk_out == NULL) / k_overlapg.
 * k[W][G][L][S], amount[S][L] -> tau[W][G][L], dk[W][G][L][S+1]. */
ORC_API void orc_k_overlap(int W, int G, int L, int S, const double *del_g, const double *k,
                           const double *dkdT, const double *amount_layer, double *tau_out,
                           double *dk_out)
{
    const int with_grad = (dkdT != NULL && dk_out != NULL);
    const int NP1 = S + 1;
    if (S == 1) { /* :6056-6058 / :5871-5876 */
        for (size_t w = 0; w < (size_t)W; ++w)
            for (int g = 0; g < G; ++g)
                for (int l = 0; l < L; ++l) {
                    size_t i = (w * G + g) * L + l;
                    tau_out[i] = k[i] * amount_layer[l];
                    if (with_grad) {
                        dk_out[i * 2 + 0] = k[i];
                        dk_out[i * 2 + 1] = dkdT[i] * amount_layer[l];
                    }
                }
        return;
    }
#pragma omp parallel
    {
        rank_ws ws;
        rank_ws_alloc(&ws, G, with_grad ? NP1 : 0);
        double *rw = (double *)malloc(sizeof(double) * G * G);
        double *rt = (double *)malloc(sizeof(double) * G * G);
        double *rg = with_grad ? (double *)malloc(sizeof(double) * G * G * NP1) : NULL;
        double *tau_g = (double *)malloc(sizeof(double) * G);
        double *tmp_tau = (double *)malloc(sizeof(double) * G);
        double *dk_g = with_grad ? (double *)malloc(sizeof(double) * G * NP1) : NULL;
        double *tmp_dk = with_grad ? (double *)malloc(sizeof(double) * G * NP1) : NULL;
        double *amount = (double *)malloc(sizeof(double) * S);
#pragma omp for schedule(dynamic, 4)
        for (int w = 0; w < W; ++w) {
            for (int l = 0; l < L; ++l) {
                for (int s = 0; s < S; ++s) amount[s] = amount_layer[(size_t)s * L + l];
                const double *kc = k + ((size_t)w * G * L + l) * S;
                const double *dc = with_grad ? dkdT + ((size_t)w * G * L + l) * S : NULL;
                overlap_cell(&ws, G, S, del_g, kc, (size_t)L * S, dc, amount, tau_g, dk_g, rw, rt,
                             rg, tmp_tau, tmp_dk);
                for (int g = 0; g < G; ++g) {
                    tau_out[((size_t)w * G + g) * L + l] = tau_g[g];
                    if (with_grad)
                        memcpy(dk_out + (((size_t)w * G + g) * L + l) * NP1, dk_g + g * NP1,
                               sizeof(double) * NP1);
                }
            }
        }
        free(rw); free(rt); free(rg); free(tau_g); free(tmp_tau); free(dk_g); free(tmp_dk);
        free(amount);
        rank_ws_free(&ws);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* K5: Planck function                 ForwardModel_0.planck :6183-6227, planckg :6230-6283    */
/* ------------------------------------------------------------------------------------------ */
static inline double planck_fn(int ispace, double wave, double temp)
{
    const double c1 = 1.1911e-12, c2 = 1.439;
    double y, a;
    if (ispace == 0) { y = wave; a = c1 * pow(y, 3.); }
    else { y = 1.0e4 / wave; a = c1 * pow(y, 5.) / 1.0e4; }
    double tmp = c2 * y / temp;
    double b = exp(tmp) - 1;
    return a / b;
}
static inline void planckg_fn(int ispace, double wave, double temp, double *bb, double *dBdT)
{
    const double c1 = 1.1911e-12, c2 = 1.439;
    double y, a, ap;
    if (ispace == 0) {
        y = wave; a = c1 * pow(y, 3.); ap = c1 * c2 * pow(y, 4.) / pow(temp, 2.);
    } else {
        y = 1.0e4 / wave; a = c1 * pow(y, 5.) / 1.0e4;
        ap = c1 * c2 * pow(y, 6.) / 1.0e4 / pow(temp, 2.);
    }
    double tmp = c2 * y / temp;
    double b = exp(tmp) - 1;
    *bb = a / b;
    double bp = pow(exp(tmp) - 1., 2.);
    double tp = exp(tmp) * ap;
    *dBdT = tp / bp;
}
ORC_API void orc_planck(int ispace, int n, const double *wave, const double *temp, double *bb,
                        double *dBdT)
{
    for (int i = 0; i < n; ++i) {
        if (dBdT) planckg_fn(ispace, wave[i], temp[i], &bb[i], &dBdT[i]);
        else bb[i] = planck_fn(ispace, wave[i], temp[i]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* K4: layer-by-layer thermal emission  calc_thermal_emission_spectrum  :6287-6377             */
/* ------------------------------------------------------------------------------------------ */
ORC_API void orc_thermal_emission(int ISPACE, int W, int G, int NLAYIN, const double *WAVE,
                                  const double *TAUTOT_PATH /*[W][G][Li]*/,
                                  const double *EMITOT_PATH /*[W][Li] or NULL*/,
                                  const double *TEMP, const double *PRESS, double TSURF,
                                  const double *EMISSIVITY, const double *SOLFLUX,
                                  const double *REFLECTANCE, double SOL_ANG, double EMISS_ANG,
                                  double *SPECOUT /*[W][G]*/)
{
    int i1 = (int)(NLAYIN / 2.0) - 1;
    if (i1 < 0) i1 += NLAYIN; /* python negative index wrap */
    const double p1 = PRESS[i1];
    const double p2 = PRESS[NLAYIN - 1];
#pragma omp parallel for schedule(static)
    for (int iwave = 0; iwave < W; ++iwave) {
        for (int ig = 0; ig < G; ++ig) {
            double taud = 0., trold = 1., specg = 0.;
            const double *tau = TAUTOT_PATH + ((size_t)iwave * G + ig) * NLAYIN;
            for (int j = 0; j < NLAYIN; ++j) {
                taud += tau[j];
                double tr = exp(-taud);
                double bb = planck_fn(ISPACE, WAVE[iwave], TEMP[j]);
                specg += (trold - tr) * bb;
                if (EMITOT_PATH) specg += EMITOT_PATH[(size_t)iwave * NLAYIN + j] * tr;
                trold = tr;
            }
            if (p2 > p1) {
                double radground;
                if (TSURF <= 0.0) radground = planck_fn(ISPACE, WAVE[iwave], TEMP[NLAYIN - 1]);
                else radground = planck_fn(ISPACE, WAVE[iwave], TSURF) * EMISSIVITY[iwave];
                specg += trold * radground;
            }
            if (EMISS_ANG < 90. && SOL_ANG < 90.) {
                double refl = REFLECTANCE[iwave], solar = SOLFLUX[iwave];
                double mu = cos(EMISS_ANG / 180. * M_PI);
                double mu0 = cos(SOL_ANG / 180. * M_PI);
                specg += trold * exp(-taud * mu / mu0) * solar * refl;
            }
            SPECOUT[(size_t)iwave * G + ig] = specg;
        }
    }
}

/* calc_thermal_emission_spectrumg  :6380-6504 (literal O(NPAR*Li^2) recursion). */
ORC_API void orc_thermal_emissiong(int ISPACE, int W, int G, int NPAR, int NLAYIN,
                                   const double *WAVE, const double *TAUTOT_PATH /*[W][G][Li]*/,
                                   const double *dTAUTOT_PATH /*[W][G][NPAR][Li]*/, int NVMR,
                                   const double *TEMP, const double *PRESS, double TSURF,
                                   const double *EMISSIVITY, double *SPECOUT /*[W][G]*/,
                                   double *dSPECOUT /*[W][G][NPAR][Li]*/, double *dTSURF /*[W][G]*/)
{
    int i1 = (int)(NLAYIN / 2.0) - 1;
    if (i1 < 0) i1 += NLAYIN;
    const double p1 = PRESS[i1];
    const double p2 = PRESS[NLAYIN - 1];
#pragma omp parallel
    {
        double *dtolddq = (double *)malloc(sizeof(double) * NPAR * NLAYIN);
        double *dtrdq = (double *)malloc(sizeof(double) * NPAR * NLAYIN);
        double *dspecg = (double *)malloc(sizeof(double) * NPAR * NLAYIN);
#pragma omp for schedule(static)
        for (int iwave = 0; iwave < W; ++iwave) {
            for (int ig = 0; ig < G; ++ig) {
                double taud = 0., trold = 1., specg = 0., tlayer;
                memset(dtolddq, 0, sizeof(double) * NPAR * NLAYIN);
                memset(dtrdq, 0, sizeof(double) * NPAR * NLAYIN);
                memset(dspecg, 0, sizeof(double) * NPAR * NLAYIN);
                const double *tau = TAUTOT_PATH + ((size_t)iwave * G + ig) * NLAYIN;
                const double *dtau = dTAUTOT_PATH + ((size_t)iwave * G + ig) * NPAR * NLAYIN;
                for (int j = 0; j < NLAYIN; ++j) {
                    taud += tau[j];
                    tlayer = exp(-tau[j]);
                    double tr = trold * tlayer;
                    double bb, dBdT;
                    planckg_fn(ISPACE, WAVE[iwave], TEMP[j], &bb, &dBdT);
                    specg += (trold - tr) * bb;
                    for (int k = 0; k < NPAR; ++k) {
                        int j1 = 0;
                        while (j1 < j) {
                            dtrdq[k * NLAYIN + j1] = dtolddq[k * NLAYIN + j1] * tlayer;
                            dspecg[k * NLAYIN + j1] +=
                                (dtolddq[k * NLAYIN + j1] - dtrdq[k * NLAYIN + j1]) * bb;
                            j1 += 1;
                        }
                        double tmp = dtau[k * NLAYIN + j1];
                        dtrdq[k * NLAYIN + j1] = -tmp * tlayer * trold;
                        dspecg[k * NLAYIN + j1] +=
                            (dtolddq[k * NLAYIN + j1] - dtrdq[k * NLAYIN + j1]) * bb;
                        if (k == NVMR) dspecg[k * NLAYIN + j] += (trold - tr) * dBdT;
                    }
                    trold = tr;
                    for (int k = 0; k < NPAR; ++k)
                        for (int j1 = 0; j1 <= j; ++j1)
                            dtolddq[k * NLAYIN + j1] = dtrdq[k * NLAYIN + j1];
                }
                double tempgtsurf = 0.;
                if (p2 > p1) {
                    double radground, dradgrounddT;
                    if (TSURF <= 0.0)
                        planckg_fn(ISPACE, WAVE[iwave], TEMP[NLAYIN - 1], &radground, &dradgrounddT);
                    else {
                        double bbsurf, dbsurfdT;
                        planckg_fn(ISPACE, WAVE[iwave], TSURF, &bbsurf, &dbsurfdT);
                        radground = bbsurf * EMISSIVITY[iwave];
                        dradgrounddT = dbsurfdT * EMISSIVITY[iwave];
                    }
                    specg += trold * radground;
                    tempgtsurf = trold * dradgrounddT;
                    for (int j = 0; j < NLAYIN; ++j)
                        for (int k = 0; k < NPAR; ++k)
                            dspecg[k * NLAYIN + j] += radground * dtolddq[k * NLAYIN + j];
                }
                SPECOUT[(size_t)iwave * G + ig] = specg;
                memcpy(dSPECOUT + ((size_t)iwave * G + ig) * NPAR * NLAYIN, dspecg,
                       sizeof(double) * NPAR * NLAYIN);
                dTSURF[(size_t)iwave * G + ig] = tempgtsurf;
            }
        }
        free(dtolddq); free(dtrdq); free(dspecg);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Fused seam: CIRSrad for ILBL=K_TABLES, IMOD=THERMAL_EMISSION, no layer emissions.           */
/*   calculate_gaseous_line_opacity :3850-3877 -> calculate_layer_opacity :3989,:4006          */
/*   -> calculate_thermal_emission_spectrum :4216-4244 -> g-quadrature :4504                   */
/* Inputs mirror what CIRSrad reads from SpectroscopyX / LayerX / PathX / SurfaceX.            */
/* ------------------------------------------------------------------------------------------ */
ORC_API void orc_cirsrad_ck_thermal(
    int ISPACE, int W, int G, int NP, int NT, int S, const double *K, const double *TPRESS,
    const double *TTEMP, const double *WAVE, const double *DELG, int L,
    const double *lay_press_atm /*[L] LayerX.PRESS/101325*/, const double *lay_temp /*[L]*/,
    const double *lay_press_pa /*[L] LayerX.PRESS*/, const double *amount /*[S][L] cm-2*/,
    const double *TAUCONT /*[W][L]  TAUCIA+TAUDUST+TAURAY, may be NULL*/, int NPATH, int LIMAX,
    const int *NLAYIN /*[P]*/, const int *LAYINC /*[LIMAX][P]*/, const double *SCALE /*[LIMAX][P]*/,
    const double *EMTEMP /*[LIMAX][P]*/, double TSURF, const double *EMISSIVITY /*[W]*/,
    const double *SOLFLUX, const double *REFLECTANCE, const double *SOL_ANG /*[P]*/,
    const double *EMISS_ANG /*[P]*/, const double *xfac /*[W] or NULL*/,
    double *SPECOUT /*[W][P]*/, double *TAUGAS_out /*[W][G][L] or NULL*/)
{
    size_t nk = (size_t)W * G * L * S;
    double *k_gas = (double *)malloc(sizeof(double) * nk);
    double *tau = (double *)malloc(sizeof(double) * (size_t)W * G * L);
    orc_calc_k(W, G, NP, NT, S, K, TPRESS, TTEMP, L, lay_press_atm, lay_temp, k_gas, NULL);
    orc_k_overlap(W, G, L, S, DELG, k_gas, NULL, amount, tau, NULL);
    free(k_gas);
    if (TAUGAS_out) memcpy(TAUGAS_out, tau, sizeof(double) * (size_t)W * G * L);
    /* TAUTOT = TAUGAS + TAUCIA + TAUDUST + TAURAY  :3989 (the three continuum terms are summed
       left to right by the caller into TAUCONT in the same order) */
    if (TAUCONT)
        for (size_t w = 0; w < (size_t)W; ++w)
            for (int g = 0; g < G; ++g)
                for (int l = 0; l < L; ++l) tau[(w * G + g) * L + l] += TAUCONT[w * L + l];
    double *zeros = (double *)calloc((size_t)W, sizeof(double));
    for (int ip = 0; ip < NPATH; ++ip) {
        int nl = NLAYIN[ip];
        double *tpath = (double *)malloc(sizeof(double) * (size_t)W * G * nl);
        double *emtemp = (double *)malloc(sizeof(double) * nl);
        double *empress = (double *)malloc(sizeof(double) * nl);
        double *spec = (double *)malloc(sizeof(double) * (size_t)W * G);
        for (int j = 0; j < nl; ++j) {
            int lay = LAYINC[(size_t)j * NPATH + ip];
            emtemp[j] = EMTEMP[(size_t)j * NPATH + ip];
            empress[j] = lay_press_pa[lay];
        }
        for (size_t w = 0; w < (size_t)W; ++w)
            for (int g = 0; g < G; ++g)
                for (int j = 0; j < nl; ++j) {
                    int lay = LAYINC[(size_t)j * NPATH + ip];
                    tpath[(w * G + g) * nl + j] =
                        tau[(w * G + g) * L + lay] * SCALE[(size_t)j * NPATH + ip]; /* :4006 */
                }
        orc_thermal_emission(ISPACE, W, G, nl, WAVE, tpath, NULL, emtemp, empress, TSURF,
                             EMISSIVITY ? EMISSIVITY : zeros, SOLFLUX ? SOLFLUX : zeros,
                             REFLECTANCE ? REFLECTANCE : zeros, SOL_ANG ? SOL_ANG[ip] : 180.,
                             EMISS_ANG ? EMISS_ANG[ip] : 180., spec);
        for (size_t w = 0; w < (size_t)W; ++w) {
            double acc = 0.0; /* tensordot over g :4504 */
            for (int g = 0; g < G; ++g) {
                double s = spec[w * G + g];
                if (xfac) s = s * xfac[w]; /* :4244 */
                acc += s * DELG[g];
            }
            SPECOUT[w * NPATH + ip] = acc;
        }
        free(tpath); free(emtemp); free(empress); free(spec);
    }
    free(zeros);
    free(tau);
}

/* ------------------------------------------------------------------------------------------ */
/* Fused gradient seam: CIRSrad(return_grad=True), ILBL=K_TABLES, IMOD=THERMAL_EMISSION.       */
/*   calc_kg :3853 -> k_overlapg :3865 -> dTAUGAS[:,:,IGAS,:] = dk*1e-4, dTAUGAS[:,:,NVMR,:]   */
/*   = dk[...,NGAS] :3868-3872 -> dTAUTOT = dTAUGAS + dTAUCON :3993 -> LAYINC*SCALE :4012      */
/*   -> calc_thermal_emission_spectrumg :4233 -> *xfac :4244-4247 -> tensordot(DELG),          */
/*   nan_to_num :4504-4508.                                                                    */
/* igas_map[S]: AtmosphereX.locate_gas(ID[i],ISO[i]); dTAUCON[W][NPAR][L] or NULL.             */
/* Outputs: SPECOUT[W][P], dSPECOUT[W][NPAR][LIMAX][P], dTSURF[W][P].                          */
/* ------------------------------------------------------------------------------------------ */
ORC_API void orc_cirsradg_ck_thermal(
    int ISPACE, int W, int G, int NP, int NT, int S, const double *K, const double *TPRESS,
    const double *TTEMP, const double *WAVE, const double *DELG, int L,
    const double *lay_press_atm, const double *lay_temp, const double *lay_press_pa,
    const double *amount /*[S][L] cm-2*/, const double *TAUCONT /*[W][L] or NULL*/,
    const double *dTAUCON /*[W][NPAR][L] or NULL*/, int NVMR, int NPAR, const int *igas_map,
    int NPATH, int LIMAX, const int *NLAYIN, const int *LAYINC, const double *SCALE,
    const double *EMTEMP, double TSURF, const double *EMISSIVITY, const double *xfac,
    double *SPECOUT, double *dSPECOUT, double *dTSURF)
{
    const int NP1 = S + 1;
    size_t nk = (size_t)W * G * L * S;
    double *k_gas = (double *)malloc(sizeof(double) * nk);
    double *dkdT = (double *)malloc(sizeof(double) * nk);
    double *tau = (double *)malloc(sizeof(double) * (size_t)W * G * L);
    double *dk = (double *)malloc(sizeof(double) * (size_t)W * G * L * NP1);
    orc_calc_k(W, G, NP, NT, S, K, TPRESS, TTEMP, L, lay_press_atm, lay_temp, k_gas, dkdT);
    orc_k_overlap(W, G, L, S, DELG, k_gas, dkdT, amount, tau, dk);
    free(k_gas); free(dkdT);
    /* dTAUTOT[W][G][NPAR][L] */
    double *dtautot = (double *)calloc((size_t)W * G * NPAR * L, sizeof(double));
    for (size_t w = 0; w < (size_t)W; ++w)
        for (int g = 0; g < G; ++g) {
            double *dst = dtautot + (w * G + g) * NPAR * L;
            for (int i = 0; i < S; ++i)      /* assignment, later gases overwrite (:3870) */
                for (int l = 0; l < L; ++l)
                    dst[(size_t)igas_map[i] * L + l] = dk[((w * G + g) * L + l) * NP1 + i] * 1.0e-4;
            for (int l = 0; l < L; ++l) dst[(size_t)NVMR * L + l] = dk[((w * G + g) * L + l) * NP1 + S];
            if (dTAUCON)
                for (int kpar = 0; kpar < NPAR; ++kpar)
                    for (int l = 0; l < L; ++l) dst[(size_t)kpar * L + l] += dTAUCON[(w * NPAR + kpar) * L + l];
        }
    free(dk);
    if (TAUCONT)
        for (size_t w = 0; w < (size_t)W; ++w)
            for (int g = 0; g < G; ++g)
                for (int l = 0; l < L; ++l) tau[(w * G + g) * L + l] += TAUCONT[w * L + l];
    double *zeros = (double *)calloc((size_t)W, sizeof(double));
    memset(dSPECOUT, 0, sizeof(double) * (size_t)W * NPAR * LIMAX * NPATH);
    for (int ip = 0; ip < NPATH; ++ip) {
        int nl = NLAYIN[ip];
        double *tpath = (double *)malloc(sizeof(double) * (size_t)W * G * nl);
        double *dtpath = (double *)malloc(sizeof(double) * (size_t)W * G * NPAR * nl);
        double *emtemp = (double *)malloc(sizeof(double) * nl);
        double *empress = (double *)malloc(sizeof(double) * nl);
        double *spec = (double *)malloc(sizeof(double) * (size_t)W * G);
        double *dspec = (double *)malloc(sizeof(double) * (size_t)W * G * NPAR * nl);
        double *dts = (double *)malloc(sizeof(double) * (size_t)W * G);
        for (int j = 0; j < nl; ++j) {
            int lay = LAYINC[(size_t)j * NPATH + ip];
            emtemp[j] = EMTEMP[(size_t)j * NPATH + ip];
            empress[j] = lay_press_pa[lay];
        }
        for (size_t w = 0; w < (size_t)W; ++w)
            for (int g = 0; g < G; ++g)
                for (int j = 0; j < nl; ++j) {
                    int lay = LAYINC[(size_t)j * NPATH + ip];
                    double sc = SCALE[(size_t)j * NPATH + ip];
                    tpath[(w * G + g) * nl + j] = tau[(w * G + g) * L + lay] * sc;
                    for (int kpar = 0; kpar < NPAR; ++kpar)
                        dtpath[((w * G + g) * NPAR + kpar) * nl + j] =
                            dtautot[((w * G + g) * NPAR + kpar) * L + lay] * sc;
                }
        orc_thermal_emissiong(ISPACE, W, G, NPAR, nl, WAVE, tpath, dtpath, NVMR, emtemp, empress, TSURF,
                              EMISSIVITY ? EMISSIVITY : zeros, spec, dspec, dts);
        for (size_t w = 0; w < (size_t)W; ++w) {
            double xf = xfac ? xfac[w] : 1.0;
            double acc = 0.0, accs = 0.0;
            for (int g = 0; g < G; ++g) {
                acc += (spec[w * G + g] * xf) * DELG[g];
                accs += (dts[w * G + g] * xf) * DELG[g];
            }
            SPECOUT[w * NPATH + ip] = acc;
            dTSURF[w * NPATH + ip] = accs;
            for (int kpar = 0; kpar < NPAR; ++kpar)
                for (int j = 0; j < nl; ++j) {
                    double a = 0.0;
                    for (int g = 0; g < G; ++g)
                        a += (dspec[((w * G + g) * NPAR + kpar) * nl + j] * xf) * DELG[g];
                    if (a != a) a = 0.0; /* nan_to_num :4507 */
                    dSPECOUT[((w * NPAR + kpar) * LIMAX + j) * NPATH + ip] = a;
                }
        }
        free(tpath); free(dtpath); free(emtemp); free(empress); free(spec); free(dspec); free(dts);
    }
    free(zeros); free(dtautot); free(tau);
}

/* ------------------------------------------------------------------------------------------ */
/* K11: LBL-table (P,T) interpolation   Spectroscopy_0.calc_klbl :1768-1919, calc_klblg :1601  */
/* K[W][NP][NTa][S]; PRESS[NP] (atm); TEMP[NTa] or, when temp2d (NT<0 in the reference),       */
/* TEMP[NP][NTa] (one temperature grid per pressure).  press[L] (atm), temp[L].                */
/* with_grad selects calc_klblg, which lacks the it<0 clamp (:1672-1675): a layer exactly at    */
/* the lowest table temperature then indexes TEMP[-1] (python wrap) -- reproduced.             */
/* ------------------------------------------------------------------------------------------ */
static int searchsorted_left(const double *a, int n, double x)
{
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
ORC_API void orc_lbl_bracket(int NP, const double *PRESS, int NTa, const double *TEMP, int temp2d,
                             double press1, double temp1, int with_grad, int *ip_o, int *it1_o, int *it2_o,
                             double *v_o, double *u1_o, double *u2_o, double *du1_o, double *du2_o,
                             double *omu1_o, double *omu2_o)
{
    double lp[256];
    double pmin = INFINITY, pmax = -INFINITY;
    for (int i = 0; i < NP; ++i) { lp[i] = logx(PRESS[i]); if (lp[i] < pmin) pmin = lp[i]; if (lp[i] > pmax) pmax = lp[i]; }
    double p_l = log(press1);
    int pcl = 0;   /* clamped p_l = np.min/np.max(log PRESS): float32 scalar when PRESS is float32 */
    if (p_l < pmin) { p_l = pmin; pcl = 1; }
    if (p_l > pmax) { p_l = pmax; pcl = 1; }
    int nt_all = temp2d ? NP * NTa : NTa;
    double tmin = INFINITY, tmax = -INFINITY;
    for (int i = 0; i < nt_all; ++i) { if (TEMP[i] < tmin) tmin = TEMP[i]; if (TEMP[i] > tmax) tmax = TEMP[i]; }
    double t_l = temp1;
    int tcl = 0;   /* a clamped t_l is np.min/np.max(TEMP): a float32 scalar when TEMP is float32, and NumPy
                      then evaluates u and (1.0-u) in float32 */
    if (t_l < tmin) { t_l = tmin; tcl = 1; }
    if (t_l > tmax) { t_l = tmax; tcl = 1; }
    int ip = searchsorted_left(lp, NP, p_l) - 1;
    if (ip < 0) ip = 0;
    if (ip >= NP - 1) ip = NP - 2;
    double pden = g_grid_f32 ? (double)((float)lp[ip + 1] - (float)lp[ip]) : lp[ip + 1] - lp[ip];
    if (g_grid_f32 && pcl) *v_o = (double)(((float)p_l - (float)lp[ip]) / (float)pden);
    else *v_o = (p_l - lp[ip]) / pden;
    const double *Tn = temp2d ? TEMP + (size_t)ip * NTa : TEMP;
    const double *Tn2 = temp2d ? TEMP + (size_t)(ip + 1) * NTa : TEMP;
    for (int side = 0; side < 2; ++side) {
        const double *T = side ? Tn2 : Tn;
        int it = searchsorted_left(T, NTa, t_l) - 1;
        if (!with_grad && it < 0) it = 0;
        if (it >= NTa - 1) it = NTa - 2;
        int itw = it < 0 ? it + NTa : it;          /* python negative index */
        int itn = it + 1;                           /* it+1 == 0 when it == -1 */
        double den = g_grid_f32 ? (double)((float)T[itn] - (float)T[itw]) : T[itn] - T[itw];
        double u = (t_l - T[itw]) / den, omu;
        if (g_grid_f32 && tcl) {
            float uf = ((float)t_l - (float)T[itw]) / (float)den;
            u = (double)uf;
            omu = (double)(1.0f - uf);
        } else
            omu = 1.0 - u;
        double du = g_grid_f32 ? (double)(1.0f / (float)den) : 1. / den;
        if (side) { *it2_o = it; *u2_o = u; *du2_o = du; *omu2_o = omu; }
        else { *it1_o = it; *u1_o = u; *du1_o = du; *omu1_o = omu; }
    }
    *ip_o = ip;
}

ORC_API void orc_calc_klbl(int W, int NP, int NTa, int S, const double *K, const double *PRESS, const double *TEMP,
                           int temp2d, int L, const double *press, const double *temp, double *k_out /*[W][L][S]*/,
                           double *dkdT_out /*or NULL*/)
{
    const int with_grad = dkdT_out != NULL;
    for (int l = 0; l < L; ++l) {
        int ip, it1, it2;
        double v, u1, u2, du1, du2, omu1, omu2;
        orc_lbl_bracket(NP, PRESS, NTa, TEMP, temp2d, press[l], temp[l], with_grad, &ip, &it1, &it2, &v, &u1, &u2, &du1, &du2,
                        &omu1, &omu2);
        int a1 = it1 < 0 ? it1 + NTa : it1, b1 = it1 + 1;
        int a2 = it2 < 0 ? it2 + NTa : it2, b2 = it2 + 1;
#pragma omp parallel for schedule(static)
        for (int w = 0; w < W; ++w)
            for (int s = 0; s < S; ++s) {
                const double *base = K + (size_t)w * NP * NTa * S;
                double klo1 = base[((size_t)ip * NTa + a1) * S + s];
                double klo2 = base[((size_t)ip * NTa + b1) * S + s];
                double khi1 = base[((size_t)(ip + 1) * NTa + a2) * S + s];
                double khi2 = base[((size_t)(ip + 1) * NTa + b2) * S + s];
                double kk = 0.0, dk = 0.0;
                if (klo1 > 0.0 && klo2 > 0.0 && khi1 > 0.0 && khi2 > 0.0) {
                    double l1 = log(klo1), l2 = log(klo2), h1 = log(khi1), h2 = log(khi2);
                    kk = exp((1.0 - v) * omu1 * l1 + v * omu2 * h1 + v * u2 * h2 + (1.0 - v) * u1 * l2);
                    double dxdt = -l1 * (1.0 - v) * du1 - h1 * v * du2 + h2 * v * du2 + l2 * (1.0 - v) * du1;
                    dk = kk * dxdt;
                } else if (klo1 <= 0.0 && klo2 <= 0.0 && khi1 <= 0.0 && khi2 <= 0.0) {
                    kk = (1.0 - v) * omu1 * klo1 + v * omu2 * khi1 + v * u2 * khi2 + (1.0 - v) * u1 * klo2;
                    dk = -klo1 * (1.0 - v) * du1 - khi1 * v * du2 + khi2 * v * du2 + klo2 * (1.0 - v) * du1;
                }
                size_t o = ((size_t)w * L + l) * S + s;
                k_out[o] = kk;
                if (with_grad) dkdT_out[o] = dk;
            }
    }
}
