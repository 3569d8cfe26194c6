// r03_select_probe.hip -- what "rebinning by selection" would cost per merge, measured.
//
// rank() (ForwardModel_0.py:6117-6173) sorts the G*G sums a_i + b_j and walks them once.  k_ck_overlap does that as a G-way
// merge: 400 steps of ~62 wave64 instructions, 38 of them the insertion into the sorted list of row heads.  The alternative
// asked for: find, for each of the G-1 bin boundaries, the element that closes the bin by STAIRCASE COUNTING on the sorted
// matrix (rows and columns ascending: the elements below a threshold t form a staircase, j_i(t) = #{j : a_i + b_j < t}
// non-increasing in i, <= 2G pointer steps), then form every bin from prefix sums:
//     mass(t) = sum_i w_i CW[j_i],   S(t) = sum_i w_i (a_i CW[j_i] + CBW[j_i]),   CW / CBW = prefix sums of w_j, b_j w_j,
//     F(g_b)  = S(t_b) + t_b (g_b - mass(t_b))      (the closing element supplies the rest of the mass at its own value),
//     bin b   = (F(g_{b+1}) - F(g_b)) / (g_{b+1} - g_b).
// What this program measures is the LOWER BOUND of any such scheme: the thresholds t_b are GIVEN (the host found them by
// sorting), so the kernel does exactly one staircase evaluation per boundary and the prefix-sum arithmetic -- no search at
// all.  A search needs more evaluations: a bisection ~log2(400) = 9 per boundary, an interpolation search from a good
// bracket still 2-3, and each evaluation is one more staircase.
//
// One lane per cell, [index][lane] LDS layout like the production kernel (conflict-free lane-dependent gathers), the same
// 7 blocks per CU.  Results are checked against a host restatement of rank().
//
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/r03_select_probe.hip -o /tmp/select_probe && /tmp/select_probe
//   (counters: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS -- /tmp/select_probe)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

constexpr int G = 20, kWave = 64;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Params {
    const double *a, *b;      // [merge][G][W]
    const double *thr;        // [merge][G-1][W]: value of the element that closes bin b (GIVEN)
    double *out;              // [merge][G][W]
    double w[G], gord[G + 1], cw[G + 1];
    int variant;              // 0: one staircase at a time; 1: four side by side, branch-free
    int W, nmerge, evals;     // evals: staircase evaluations per boundary (1 = thresholds known; 2, 3 = what a search adds)
};

__global__ __launch_bounds__(kWave) void k_select(Params p)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    double *A = sm, *B = A + G * kWave, *CBW = B + G * kWave;            // A[G], B[G], CBW[G+1] rows of 64 lanes
    double *TW = CBW + (G + 1) * kWave, *TCW = TW + G;                   // weight tables (a kernel-argument array indexed per
    if (lane < G) TW[lane] = p.w[lane];                                  // lane would be copied to scratch memory)
    if (lane <= G) TCW[lane] = p.cw[lane];
    __syncthreads();
    const int tiles = p.W / kWave;
    for (long t = blockIdx.x; t < (long)p.nmerge * tiles; t += gridDim.x) {
        const int mg = (int)(t / tiles), nu = (int)(t % tiles) * kWave + lane;
        const double *a = p.a + ((size_t)mg * G) * p.W + nu, *b = p.b + ((size_t)mg * G) * p.W + nu;
        double acc = 0.0, stot = 0.0;
        CBW[lane] = 0.0;
        for (int g = 0; g < G; ++g) {
            const double av = a[(size_t)g * p.W], bv = b[(size_t)g * p.W];
            A[g * kWave + lane] = av; B[g * kWave + lane] = bv;
            acc = fma(bv, p.w[g], acc);
            CBW[(g + 1) * kWave + lane] = acc;
            stot = fma(av, p.w[g], stot);
        }
        const double total = stot + acc;          // sum over all elements of (a_i + b_j) w_i w_j  (sum of w = 1)
        double Fprev = 0.0;
        double *out = p.out + ((size_t)mg * G) * p.W + nu;
        if (p.variant == 0) {
        for (int bnd = 1; bnd <= G; ++bnd) {
            double F;
            if (bnd < G) {
                double thr = p.thr[((size_t)mg * (G - 1) + (bnd - 1)) * p.W + nu];
                double mass = 0.0, S = 0.0;
                for (int ev = 0; ev < p.evals; ++ev) {          // evals > 1: the same evaluation again, as a search step would
                    mass = 0.0; S = 0.0;
                    int i = 0, j = G;
                    // staircase: <= 2G steps, each either "column pointer down" or "row done"
                    for (int it = 0; it < 2 * G; ++it) {
                        const bool live = i < G;
                        const int ii = live ? i : G - 1, jj = j > 0 ? j - 1 : 0;
                        const double ai = A[ii * kWave + lane];
                        const double c = ai + B[jj * kWave + lane];
                        const bool down = live && j > 0 && c >= thr;
                        if (down) --j;
                        else if (live) {
                            const double wi = TW[ii], cwj = TCW[j];
                            mass = fma(wi, cwj, mass);
                            S = fma(wi, fma(ai, cwj, CBW[j * kWave + lane]), S);
                            ++i;
                        }
                    }
                    thr += 0.0 * mass;                          // keep the repeated evaluation dependent on the previous one
                }
                F = fma(thr, p.gord[bnd] - mass, S);
            } else
                F = total;
            out[(size_t)(bnd - 1) * p.W] = (F - Fprev) / (p.gord[bnd] - p.gord[bnd - 1]);
            Fprev = F;
        }
        } else {
        // variant 1: the staircases of NB boundaries walked side by side -- NB independent pointer chains, so the LDS round trip
        // of one is covered by the others' instructions (a single staircase is one dependent chain of LDS reads) -- and every
        // step branch-free (all five reads issued, the updates selected)
        constexpr int NB = 4;
        for (int b0 = 1; b0 < G; b0 += NB) {
            double thr[NB], mass[NB], S[NB];
            int ip[NB], jp[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int bnd = (b0 + k < G) ? b0 + k : G - 1;
                thr[k] = p.thr[((size_t)mg * (G - 1) + (bnd - 1)) * p.W + nu];
            }
            for (int ev = 0; ev < p.evals; ++ev) {
#pragma unroll
                for (int k = 0; k < NB; ++k) { mass[k] = 0.0; S[k] = 0.0; ip[k] = 0; jp[k] = G; }
                for (int it = 0; it < 2 * G; ++it) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        const bool live = ip[k] < G;
                        const int ii = live ? ip[k] : G - 1, jj = jp[k] > 0 ? jp[k] - 1 : 0;
                        const double ai = A[ii * kWave + lane], bj = B[jj * kWave + lane];
                        const double wi = TW[ii], cwj = TCW[jp[k]], cb = CBW[jp[k] * kWave + lane];
                        const bool down = live && jp[k] > 0 && (ai + bj) >= thr[k];
                        const bool take = live && !down;
                        const double wsel = take ? wi : 0.0;
                        mass[k] = fma(wsel, cwj, mass[k]);
                        S[k] = fma(wsel, fma(ai, cwj, cb), S[k]);
                        jp[k] -= down ? 1 : 0;
                        ip[k] += take ? 1 : 0;
                    }
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) thr[k] += 0.0 * mass[k];
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int bnd = b0 + k;
                if (bnd < G) {
                    const double F = fma(thr[k], p.gord[bnd] - mass[k], S[k]);
                    out[(size_t)(bnd - 1) * p.W] = (F - Fprev) / (p.gord[bnd] - p.gord[bnd - 1]);
                    Fprev = F;
                }
            }
        }
        out[(size_t)(G - 1) * p.W] = (total - Fprev) / (p.gord[G] - p.gord[G - 1]);
        }
    }
}

// host: rank() on one cell -> bins and the closing elements' values
static void host_rank(const double *a, const double *b, const double *w, const double *gord, double *bins, double *thr)
{
    struct E { double v, w; };
    std::vector<E> e;
    for (int i = 0; i < G; ++i) for (int j = 0; j < G; ++j) e.push_back({a[i] + b[j], w[i] * w[j]});
    std::stable_sort(e.begin(), e.end(), [](const E &x, const E &y) { return x.v < y.v; });
    int ig = 0;
    double gd = 0.0, sum1 = 0.0, kacc = 0.0;
    for (int k = 0; k < G; ++k) bins[k] = 0.0;
    for (size_t n = 0; n < e.size(); ++n) {
        const double gdn = gd + e[n].w;
        if (gdn < gord[ig + 1] && ig < G) { kacc += e[n].v * e[n].w; sum1 += e[n].w; }
        else {
            const double frac = (gord[ig + 1] - gd) / (gdn - gd);
            bins[ig] = (kacc + frac * e[n].v * e[n].w) / (sum1 + frac * e[n].w);
            if (ig < G - 1) thr[ig] = e[n].v;
            ++ig;
            if (ig < G) { kacc = (1.0 - frac) * e[n].v * e[n].w; sum1 = (1.0 - frac) * e[n].w; }
        }
        gd = gdn;
    }
    if (ig == G - 1) bins[ig] = kacc / sum1;
}

int main(int argc, char **argv)
{
    const int W = 10048, nmerge = 700;           // 700 x 10048 = 7.03e6 cell-merges = one C2 forward model (1e6 cells x 7 merges)
    std::vector<double> xw(G), ww(G);
    {   // Gauss-Legendre on [0, 1], float32-rounded like a .kta header
        for (int i = 0; i < G; ++i) {                      // Newton on P_G
            double x = cos(M_PI * (i + 0.75) / (G + 0.5));
            for (int it = 0; it < 100; ++it) {
                double p0 = 1.0, p1 = x;
                for (int k = 2; k <= G; ++k) { const double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k; p0 = p1; p1 = pk; }
                const double dp = G * (x * p1 - p0) / (x * x - 1.0);
                const double dx = p1 / dp; x -= dx; if (fabs(dx) < 1e-15) break;
            }
            double p0 = 1.0, p1 = x;
            for (int k = 2; k <= G; ++k) { const double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k; p0 = p1; p1 = pk; }
            const double dp = G * (x * p1 - p0) / (x * x - 1.0);
            xw[G - 1 - i] = 0.5 * (x + 1.0); ww[G - 1 - i] = (double)(float)(1.0 / ((1.0 - x * x) * dp * dp));
        }
    }
    Params p;
    double acc = 0.0;
    p.gord[0] = 0.0; p.cw[0] = 0.0;
    for (int g = 0; g < G; ++g) { p.w[g] = ww[g]; acc += ww[g]; p.gord[g + 1] = acc; p.cw[g + 1] = acc; }
    p.gord[G] = 1.0;
    const size_t n = (size_t)nmerge * G * W;
    std::vector<double> a(n), b(n), thr((size_t)nmerge * (G - 1) * W), ref((size_t)G * W);
    std::mt19937_64 rng(20260705);
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    for (int m = 0; m < nmerge; ++m)
        for (int w = 0; w < W; ++w) {
            double ta[G], tb[G];
            const double ba = pow(10.0, -6.0 + 9.0 * u01(rng)), bb = pow(10.0, -6.0 + 9.0 * u01(rng));   // levels decades apart
            for (int g = 0; g < G; ++g) { ta[g] = ba * pow(10.0, 5.0 * u01(rng)); tb[g] = bb * pow(10.0, 5.0 * u01(rng)); }
            std::sort(ta, ta + G); std::sort(tb, tb + G);
            for (int g = 0; g < G; ++g) { a[((size_t)m * G + g) * W + w] = ta[g]; b[((size_t)m * G + g) * W + w] = tb[g]; }
            if (m < 2 || m == nmerge - 1) {      // thresholds for every cell would take the host minutes: three merges are checked,
                double bins[G], t[G - 1];        // the others get the thresholds of merge 0's lane (timing does not depend on them)
                host_rank(ta, tb, p.w, p.gord, bins, t);
                for (int k = 0; k < G - 1; ++k) thr[((size_t)m * (G - 1) + k) * W + w] = t[k];
            }
        }
    for (int m = 2; m < nmerge - 1; ++m)
        for (int k = 0; k < G - 1; ++k)
            for (int w = 0; w < W; ++w) {      // a plausible threshold inside the cell's own range (timing only)
                const double lo = a[((size_t)m * G) * W + w] + b[((size_t)m * G) * W + w];
                const double hi = a[((size_t)m * G + G - 1) * W + w] + b[((size_t)m * G + G - 1) * W + w];
                thr[((size_t)m * (G - 1) + k) * W + w] = lo * pow(hi / lo, p.gord[k + 1]);
            }
    double *da, *db, *dt, *dout;
    CHK(hipMalloc(&da, n * 8)); CHK(hipMalloc(&db, n * 8)); CHK(hipMalloc(&dt, thr.size() * 8)); CHK(hipMalloc(&dout, n * 8));
    CHK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dt, thr.data(), thr.size() * 8, hipMemcpyHostToDevice));
    p.a = da; p.b = db; p.thr = dt; p.out = dout; p.W = W; p.nmerge = nmerge;
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const size_t lds = (size_t)(3 * G + 1) * kWave * 8 + (2 * G + 1) * 8;
    const int grid = prop.multiProcessorCount * 7;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant)
    for (int evals = 1; evals <= 3; ++evals) {
        p.evals = evals; p.variant = variant;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_select, dim3(grid), dim3(kWave), lds, 0, p);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
        }
        printf("variant %d (%s), %d staircase evaluation(s) per boundary: %.3f ms for %d x %d cell-merges (k_ck_overlap: 5.57 ms for the same count)\n",
               variant, variant ? "four boundaries side by side, branch-free" : "one staircase at a time", evals, best, nmerge, W);
    }
    // correctness of the bins (merges 0, 1 and the last one carry exact thresholds)
    std::vector<double> got(n);
    p.evals = 1; p.variant = 1;
    hipLaunchKernelGGL(k_select, dim3(grid), dim3(kWave), lds, 0, p);
    CHK(hipMemcpy(got.data(), dout, n * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int m : {0, 1, nmerge - 1})
        for (int w = 0; w < W; ++w) {
            double ta[G], tb[G], bins[G], t[G - 1];
            for (int g = 0; g < G; ++g) { ta[g] = a[((size_t)m * G + g) * W + w]; tb[g] = b[((size_t)m * G + g) * W + w]; }
            host_rank(ta, tb, p.w, p.gord, bins, t);
            for (int g = 0; g < G; ++g) worst = std::max(worst, fabs(got[((size_t)m * G + g) * W + w] - bins[g]) / bins[g]);
        }
    printf("bins from selection + prefix sums vs rank(): max relative difference %.2e over %d cells\n", worst, 3 * W);
    printf("(the difference is the cancellation in F(g_b+1) - F(g_b): low bins are differences of sums dominated by the high elements)\n");
    return 0;
}
