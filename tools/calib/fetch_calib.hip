// Calibration of rocprofv3's FETCH_SIZE for 8-byte-per-lane coalesced loads on gfx950 (the k-table reads of
// k_ck_overlap are of this kind).  Reads a 1 GiB buffer once with (a) plain and (b) non-temporal loads.
// build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip ; run under
//        rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool NT>
__global__ void k_read8(const double *__restrict__ src, size_t n, double *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (; i < n; i += stride) s += NT ? __builtin_nontemporal_load(src + i) : src[i];
    if (s == 123.456) out[0] = s;
}
int main()
{
    const size_t n = (size_t)1 << 27;   // 1 GiB of doubles
    double *d, *o;
    hipMalloc(&d, n * 8); hipMalloc(&o, 8);
    hipMemset(d, 0, n * 8);
    hipDeviceSynchronize();
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_read8<false>, dim3(256 * 8), dim3(256), 0, 0, d, n, o);
        hipLaunchKernelGGL(k_read8<true>, dim3(256 * 8), dim3(256), 0, 0, d, n, o);
    }
    hipDeviceSynchronize();
    printf("bytes per launch: %zu\n", n * 8);
    return 0;
}
