// Small fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32) for the embedding FC and the AAM cosine
// GEMM and their gradients (reference scripts/model.py:357,485; nn.Linear / F.linear and autograd).
//   C[m][n] = alpha * sum_k A(m,k) * B(k,n) [+ bias[n]] [+ C[m][n] if accumulate]
// with A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn] (any transposition is a stride choice).
// One block = one 32x32 output tile; its four waves split each staged K chunk four ways and fold through
// LDS in a fixed order, so small-M problems (M = batch = 256) still fill the chip and stay deterministic.
#include "spk_common.h"

#define GK 64

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                       float* __restrict__ Cm, const float* __restrict__ bias, int M, int N,
                                                       int K, long long sam, long long sak, long long sbk, long long sbn,
                                                       long long ldc, float alpha, int accumulate) {
    __shared__ float As[GK][33];
    __shared__ float Bs[GK][33];
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += GK) {
        __syncthreads();
        // stage A[32 x GK] and B[GK x 32]; thread order follows whichever index is contiguous in memory
        for (int i = tid; i < 32 * GK; i += 256) {
            int mm, kk;
            if (sak == 1) { kk = i % GK; mm = i / GK; } else { mm = i % 32; kk = i / 32; }
            const int m = m0 + mm, k = k0 + kk;
            As[kk][mm] = (m < M && k < K) ? A[m * sam + k * sak] : 0.f;
        }
        for (int i = tid; i < 32 * GK; i += 256) {
            int nn, kk;
            if (sbk == 1) { kk = i % GK; nn = i / GK; } else { nn = i % 32; kk = i / 32; }
            const int n = n0 + nn, k = k0 + kk;
            Bs[kk][nn] = (n < N && k < K) ? Bm[k * sbk + n * sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GK / 8; ++s) {
            const int kk = (wave * (GK / 8) + s) * 2 + h;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk][r], Bs[kk][r], acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = acc[e] + red[0][e][lane];
            v += red[1][e][lane];
            v += red[2][e][lane];
            const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * h, n = n0 + r;
            if (m < M && n < N) {
                v *= alpha;
                if (bias) v += bias[n];
                float* dst = Cm + m * ldc + n;
                *dst = accumulate ? *dst + v : v;
            }
        }
    }
}

extern "C" int spk_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long long sam,
                            long long sak, long long sbk, long long sbn, long long ldc, float alpha, int accumulate,
                            void* stream) {
    SPK_REQUIRE(A && B && C, "spk_gemm_f32: null pointer");
    SPK_REQUIRE(M > 0 && N > 0 && K > 0, "spk_gemm_f32: empty problem %dx%dx%d", M, N, K);
    SPK_REQUIRE(ldc >= N, "spk_gemm_f32: ldc=%lld < N=%d", ldc, N);
    dim3 grid(spk_ceil_div(N, 32), spk_ceil_div(M, 32));
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, bias, M, N, K, sam, sak, sbk, sbn, ldc,
                       alpha, accumulate);
    SPK_LAUNCH_CHECK("spk_gemm_f32");
    return 0;
}

// db[n] = sum_m dY[m][n]  (bias gradient of nn.Linear)
__global__ void colsum_kernel(const float* __restrict__ dy, float* __restrict__ db, int M, int N, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dy[(size_t)m * N + n];
    db[n] = accumulate ? db[n] + s : s;
}

extern "C" int spk_colsum(const float* dy, float* db, int M, int N, int accumulate, void* stream) {
    SPK_REQUIRE(dy && db && M > 0 && N > 0, "spk_colsum: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3(spk_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, dy, db, M, N, accumulate);
    SPK_LAUNCH_CHECK("spk_colsum");
    return 0;
}
