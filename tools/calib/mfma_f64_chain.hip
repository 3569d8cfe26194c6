// What the fp64 matrix cores of gfx950 deliver for the product pattern of the doubling / adding chain (k_ms_chain16): one wave =
// one 16 x 16 problem, a product = four dependent v_mfma_f64_16x16x4 (k-blocks accumulate), products chained through an LDS
// round trip (D layout -> A layout of the next left operand).  W waves per SIMD (one-wave blocks), variants:
//   0  independent products (fresh accumulator per product, operands in registers): the pipe's ceiling at W waves
//   1  every product's left operand is the previous result taken through LDS (store_d, fence, load_a): the chain's pattern
//   2  as 1 plus the Frobenius norm of the result (DPP / permlane reduction + sqrt) after every product
//   hipcc --offload-arch=gfx950 -O3 -o tools/calib/mfma_f64_chain tools/calib/mfma_f64_chain.hip ; ./mfma_f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4 __attribute__((ext_vector_type(4)));
#define FENCE() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double frob(v4 v)
{
    double s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    s += dpp_f64<0x128>(s); s += dpp_f64<0x124>(s); s += dpp_f64<0x122>(s); s += dpp_f64<0x121>(s);
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    return sqrt(s);
}

template <int VAR, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k(double *out, int iters, double seed)
{
    __shared__ double M[16 * 17];
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    v4 b = {seed + lane * 1e-3, seed * 0.5, seed * 0.25 + c * 1e-4, seed * 0.125};
    double a[4] = {seed * 1e-2 + q, seed * 2e-2, seed * 3e-2, seed * 4e-2};
    v4 sum = {0, 0, 0, 0};
    double nrm = 0.0;
    for (int it = 0; it < iters; ++it) {
        v4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb], b[kb], acc, 0, 0, 0);
        if (VAR >= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(q + 4 * r) * 17 + c] = acc[r] * 1e-3;
            FENCE();
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) a[kb] = M[c * 17 + q + 4 * kb];
            FENCE();
        }
        if (VAR >= 2) nrm += frob(acc);
        sum += acc;
    }
    out[(size_t)blockIdx.x * 64 + lane] = sum[0] + sum[1] + sum[2] + sum[3] + nrm;
}

template <int VAR, int WAVES> static void run(double *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<VAR, WAVES>), dim3(blocks), dim3(64), 0, 0, d, 16, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<VAR, WAVES>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * iters * 4 * 2048.0;
    printf("variant %d  %d waves/SIMD  %6d blocks  %8.3f ms  %6.1f TFLOP/s = %4.1f %% of 78.6\n", VAR, WAVES, blocks, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100);
}

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    double *d;
    hipMalloc(&d, (size_t)cus * 4 * 8 * 64 * sizeof(double));
    const int iters = 200000;
    run<0, 1>(d, cus * 4 * 1, iters); run<0, 2>(d, cus * 4 * 2, iters); run<0, 3>(d, cus * 4 * 3, iters); run<0, 4>(d, cus * 4 * 4, iters);
    run<1, 1>(d, cus * 4 * 1, iters); run<1, 2>(d, cus * 4 * 2, iters); run<1, 3>(d, cus * 4 * 3, iters); run<1, 4>(d, cus * 4 * 4, iters);
    run<2, 1>(d, cus * 4 * 1, iters); run<2, 2>(d, cus * 4 * 2, iters); run<2, 3>(d, cus * 4 * 3, iters); run<2, 4>(d, cus * 4 * 4, iters);
    hipFree(d);
    return 0;
}
