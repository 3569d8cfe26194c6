#include <hip/hip_runtime.h>
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double xor16_sum(double s)
{
    const long long b = __double_as_longlong(s);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double x = __longlong_as_double(((long long)rh[0] << 32) | (unsigned long long)rl[0]);
    const double y = __longlong_as_double(((long long)rh[1] << 32) | (unsigned long long)rl[1]);
    return x + y;
}
__device__ __forceinline__ double xor32_sum(double s)
{
    const long long b = __double_as_longlong(s);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double x = __longlong_as_double(((long long)rh[0] << 32) | (unsigned long long)rl[0]);
    const double y = __longlong_as_double(((long long)rh[1] << 32) | (unsigned long long)rl[1]);
    return x + y;
}
__global__ void k(const double *in, double *out)
{
    double s = in[threadIdx.x];
    double a = xor32_sum(xor16_sum(s));                 // column sums over the four 16-lane rows
    double r = s;
    r += dpp_f64<0x128>(r); r += dpp_f64<0x124>(r); r += dpp_f64<0x122>(r); r += dpp_f64<0x121>(r);
    r = xor32_sum(xor16_sum(r));
    out[threadIdx.x] = a; out[64 + threadIdx.x] = r;
}
int main()
{
    double h[64], o[128], *d, *e;
    for (int i = 0; i < 64; ++i) h[i] = 1.0 + i * 0.37;
    hipMalloc(&d, sizeof h); hipMalloc(&e, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
    double tot = 0; for (int i = 0; i < 64; ++i) tot += h[i];
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int c = i & 15; const double cs = h[c] + h[c + 16] + h[c + 32] + h[c + 48];
        if (fabs(o[i] - cs) > 1e-12) ++bad;
        if (fabs(o[64 + i] - tot) > 1e-10) ++bad;
    }
    printf("bad %d tot %.15g got %.15g colsum0 %.15g got %.15g\n", bad, tot, o[64], h[0]+h[16]+h[32]+h[48], o[0]);
    return bad != 0;
}
