// lds_granule.hip -- how many 64-thread blocks fit on a CU as a function of their dynamic LDS (the allocation granule).
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/calib/lds_granule.hip -o /tmp/lds_granule && /tmp/lds_granule
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ double smem[];
__global__ __launch_bounds__(64) void k_probe(double *out) { smem[threadIdx.x] = 1.0; __syncthreads(); out[threadIdx.x] = smem[63 - threadIdx.x]; }
int main()
{
    int prev = -1;
    for (size_t lds = 16384; lds <= 40960; lds += 64) {
        int nb = 0;
        hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_probe, 64, lds) != hipSuccess) { printf("query failed at %zu\n", lds); return 1; }
        if (nb != prev) { printf("lds >= %zu bytes: %d blocks per CU\n", lds, nb); prev = nb; }
    }
    return 0;
}
