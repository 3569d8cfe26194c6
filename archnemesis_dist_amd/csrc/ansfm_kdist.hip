// ansfm_kdist.hip -- k-distribution of a line-by-line spectrum inside spectral bins: the numerical core of the k-table
// generator Spectroscopy_0.calc_ktable_chunk (Spectroscopy_0.py:3620-3652).
//
// For every bin the reference selects the line-by-line points inside it, sorts the absorption coefficients
// (np.argsort), weights every point with the instrument function at its distance from the bin centre (np.interp of
// AFIL over VFIL - VCONV, or 1), forms the cumulative distribution g = cumsum(w dv) / sum(w dv) and reads k at the
// g-ordinates (np.interp(G_ORD, g_sorted, k_sorted)).  Here: one gather kernel builds the (k, w dv) pairs of all bins
// (bins may overlap), ONE segmented radix sort (hipCUB / rocPRIM -- plain library work) orders every bin by k, one
// block per bin then scans the weights and interpolates at the g-ordinates.
//
// Own translation unit: the rocPRIM templates are kept out of the main library source.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include "ansfm_kdist.hip.h"

namespace ansfm {

__global__ __launch_bounds__(256) void k_kdist_gather(KdistParams p)
{
    const int b = blockIdx.x;
    const int64_t o = p.off[b];
    const int n = (int)(p.off[b + 1] - o);
    const int i0 = p.i0[b];
    const int nf = p.nfil ? p.nfil[b] : 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double w = 1.0;
        if (p.nfil) {                                  // np.interp(delta, xp, fp): clamped outside [xp[0], xp[nf-1]]
            const double d = p.wavecalc[i0 + j] - p.wcen[b];
            const double *xp = p.dfil + b, *fp = p.afil + b;
            const size_t st = (size_t)p.nbin;
            if (d <= xp[0]) w = fp[0];
            else if (d >= xp[(size_t)(nf - 1) * st]) w = fp[(size_t)(nf - 1) * st];
            else {
                int a = 0, c = nf - 1;
                while (c - a > 1) { const int m = (a + c) >> 1; if (xp[(size_t)m * st] <= d) a = m; else c = m; }
                const double x0 = xp[(size_t)a * st], x1 = xp[(size_t)(a + 1) * st];
                const double y0 = fp[(size_t)a * st], y1 = fp[(size_t)(a + 1) * st];
                w = ((y1 - y0) / (x1 - x0)) * (d - x0) + y0;
            }
        }
        p.keys[o + j] = p.kabs[i0 + j];
        p.vals[o + j] = w * p.dv;
    }
}

// after the sort: keys = k_sorted, vals = (w dv) in that order.  vals becomes g_sorted in place.
__global__ __launch_bounds__(256) void k_kdist_quantiles(KdistParams p)
{
    __shared__ double part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t o = p.off[b];
    const int n = (int)(p.off[b + 1] - o);
    double *g = p.vals + o;
    const double *ks = p.keys + o;
    const int per = (n + 255) / 256;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    double s = 0.0;
    for (int i = lo; i < hi; ++i) s += g[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {                                    // exclusive scan of the 256 chunk sums, in index order
        double acc = 0.0;
        for (int t = 0; t < 256; ++t) { const double v = part[t]; part[t] = acc; acc += v; }
    }
    __syncthreads();
    double acc = part[tid];
    for (int i = lo; i < hi; ++i) { acc += g[i]; g[i] = acc; }
    __syncthreads();
    const double total = g[n - 1];
    for (int i = lo; i < hi; ++i) g[i] = g[i] / total;
    __syncthreads();
    for (int q = tid; q < p.NG; q += 256) {            // np.interp(G_ORD, g_sorted, k_sorted)
        const double x = p.g_ord[q];
        double v;
        if (x <= g[0]) v = ks[0];
        else if (x >= g[n - 1]) v = ks[n - 1];
        else {
            int a = 0, c = n - 1;
            while (c - a > 1) { const int m = (a + c) >> 1; if (g[m] <= x) a = m; else c = m; }
            const double slope = (ks[a + 1] - ks[a]) / (g[a + 1] - g[a]);
            v = slope * (x - g[a]) + ks[a];
        }
        p.kout[(size_t)b * p.NG + q] = v;
    }
}

}  // namespace ansfm

// Host driver (device pointers in, device result out); returns a hipError_t as int, 0 on success.
extern "C" __attribute__((visibility("hidden"))) int ansfm_kdist_run(void *stream_v, ansfm::KdistParams p, int64_t total)
{
    hipStream_t stream = (hipStream_t)stream_v;
    double *keys_out = nullptr, *vals_out = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    hipError_t e;
#define KD(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)
    KD(hipMalloc(&keys_out, (size_t)total * sizeof(double)));
    KD(hipMalloc(&vals_out, (size_t)total * sizeof(double)));
    hipLaunchKernelGGL(ansfm::k_kdist_gather, dim3((unsigned)p.nbin), dim3(256), 0, stream, p);
    KD(hipGetLastError());
    KD(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, temp_bytes, (const double *)p.keys, keys_out, (const double *)p.vals,
                                                   vals_out, (int)total, p.nbin, p.off, p.off + 1, 0, 64, stream));
    KD(hipMalloc(&temp, temp_bytes ? temp_bytes : 8));
    KD(hipcub::DeviceSegmentedRadixSort::SortPairs(temp, temp_bytes, (const double *)p.keys, keys_out, (const double *)p.vals,
                                                   vals_out, (int)total, p.nbin, p.off, p.off + 1, 0, 64, stream));
    {
        ansfm::KdistParams q = p;
        q.keys = keys_out; q.vals = vals_out;
        hipLaunchKernelGGL(ansfm::k_kdist_quantiles, dim3((unsigned)p.nbin), dim3(256), 0, stream, q);
        KD(hipGetLastError());
    }
    KD(hipStreamSynchronize(stream));
done:
#undef KD
    if (keys_out) (void)hipFree(keys_out);
    if (vals_out) (void)hipFree(vals_out);
    if (temp) (void)hipFree(temp);
    return (int)e;
}
