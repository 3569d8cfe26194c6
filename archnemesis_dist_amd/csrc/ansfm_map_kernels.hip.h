// ansfm_map_kernels.hip.h -- layer -> profile -> state-vector gradient maps on gfx950.
//
// ForwardModel_0.map2pro (:5319-5383) and map2xvec (:5387-5424) are small dense contractions
// (np.tensordot) applied to the analytic-gradient spectrum right after CIRSrad:
//   map2pro : dSPECOUT[w,par,pro,p] = sum_j dSPECIN[w,par,j,p] * M_par[LAYINC[j,p], pro]   (M = DAM | DTE | DCO)
//   map2xvec: dSPECOUT[w,p,x]       = sum_{par,pro} dSPECIN[w,par,pro,p] * xmap[x,par,pro]
// Both are one strided, batched float64 GEMM on the matrix cores (v_mfma_f64_16x16x4_f64): the operands are
// addressed in the reference's own array layouts through element strides, so nothing is transposed or
// re-packed in HBM.  Block = 4 wavefronts = a 64x64 tile of C (each wave 32x32 = 2x2 MFMA tiles), K walked in
// chunks of 16 staged through LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ansfm {

struct GemmBatch { long long a_off, b_off, c_off; };   // element offsets of one batch entry

struct GemmParams {
    const double *A, *B;
    double *C;
    const GemmBatch *batch;
    int M, N, K;
    long long a_sm, a_sk, b_sk, b_sn, c_sm, c_sn;      // element strides
};

typedef double map_v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_gemm_f64(GemmParams g)
{
    __shared__ double As[64][17];
    __shared__ double Bs[16][65];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int wr = wv >> 1, wc = wv & 1;
    const GemmBatch bt = g.batch[blockIdx.z];
    const double *A = g.A + bt.a_off, *B = g.B + bt.b_off;
    double *C = g.C + bt.c_off;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    map_v4f64 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = map_v4f64{0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = t + 256 * r;
            {   // A tile: k fastest
                const int i = idx >> 4, kk = idx & 15;
                const int m = m0 + i, k = k0 + kk;
                As[i][kk] = (m < g.M && k < g.K) ? A[(long long)m * g.a_sm + (long long)k * g.a_sk] : 0.0;
            }
            {   // B tile: n fastest
                const int kk = idx >> 6, j = idx & 63;
                const int k = k0 + kk, n = n0 + j;
                Bs[kk][j] = (k < g.K && n < g.N) ? B[(long long)k * g.b_sk + (long long)n * g.b_sn] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double a0 = As[wr * 32 + c][kb * 4 + q], a1 = As[wr * 32 + 16 + c][kb * 4 + q];
            const double b0 = Bs[kb * 4 + q][wc * 32 + c], b1 = Bs[kb * 4 + q][wc * 32 + 16 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + i * 16 + q + 4 * r, n = n0 + wc * 32 + j * 16 + c;
                if (m < g.M && n < g.N) C[(long long)m * g.c_sm + (long long)n * g.c_sn] = acc[i][j][r];
            }
}

}  // namespace ansfm
