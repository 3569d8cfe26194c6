// How close are the v_rcp_f64-based quotients to the IEEE division?  hipcc --offload-arch=gfx950 -O3 div_check.hip -o div_check
// Counts, over 2^26 random (n, d) pairs, the quotients that differ from n / d (correctly rounded) for one and for two
// Newton steps on the hardware reciprocal before the residual correction, and the worst relative error of each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ inline double div_nr(double n, double d, int steps)
{
    double r = __builtin_amdgcn_rcp(d);
    for (int s = 0; s < steps; ++s) r = fma(fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

__device__ inline uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void k_check(unsigned long long *bad, double *worst, double *rcp_err)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long b1 = 0, b2 = 0;
    double w1 = 0, w2 = 0, wr = 0;
    for (int t = 0; t < 64; ++t) {
        const uint64_t a = mix(i * 64 + t), b = mix(a);
        // mantissas uniform, exponents within +-40
        const double n = ldexp(1.0 + (double)(a >> 12) * 0x1p-52, (int)(a & 63) - 32);
        const double d = ldexp(1.0 + (double)(b >> 12) * 0x1p-52, (int)(b & 63) - 32);
        const double q = n / d, q1 = div_nr(n, d, 1), q2 = div_nr(n, d, 2);
        b1 += q1 != q; b2 += q2 != q;
        w1 = fmax(w1, fabs(q1 - q) / q); w2 = fmax(w2, fabs(q2 - q) / q);
        wr = fmax(wr, fabs(fma(-d, __builtin_amdgcn_rcp(d), 1.0)));
    }
    atomicAdd(&bad[0], b1); atomicAdd(&bad[1], b2);
    // non-negative doubles order like their bit patterns
    atomicMax((unsigned long long *)&worst[0], (unsigned long long)__double_as_longlong(w1));
    atomicMax((unsigned long long *)&worst[1], (unsigned long long)__double_as_longlong(w2));
    atomicMax((unsigned long long *)rcp_err, (unsigned long long)__double_as_longlong(wr));
}

int main()
{
    unsigned long long *bad; double *worst, *rerr;
    (void)hipMalloc(&bad, 16); (void)hipMalloc(&worst, 16); (void)hipMalloc(&rerr, 8);
    (void)hipMemset(bad, 0, 16); (void)hipMemset(worst, 0, 16); (void)hipMemset(rerr, 0, 8);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, bad, worst, rerr);
    unsigned long long hb[2]; double hw[2], hr;
    (void)hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hw, worst, 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&hr, rerr, 8, hipMemcpyDeviceToHost);
    printf("pairs %llu\nrcp worst |1 - d*rcp(d)| = %.3e\none Newton step : %llu differ from n/d, worst rel err %.3e\ntwo Newton steps: %llu differ from n/d, worst rel err %.3e\n",
           4096ull * 256 * 64, hr, hb[0], hw[0], hb[1], hw[1]);
    return 0;
}
