// Small fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32) for the embedding FC and the AAM cosine
// GEMM and their gradients (reference scripts/model.py:357,485; nn.Linear / F.linear and autograd).
//   C[m][n] = alpha * sum_k A(m,k) * B(k,n) [+ bias[n]] [+ C[m][n] if accumulate]
// with A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn] (any transposition is a stride choice).
// (this is the "embedding FC / AAM cosine GEMM" the north star wants on the matrix cores: fp32 in, fp32 out)
#include "spk_common.h"

// ---- tiled split-K kernel ------------------------------------------------------------------------------------------
// Block = 64 x 64 output tile, four waves = 2 x 2 sub-tiles of 32 x 32 (one MFMA accumulator each).  The K range is cut
// over blockIdx.z so that even the 256 x 256 x 5120 embedding FC puts >= 256 blocks on the chip; every slice writes its
// partial tile to a workspace slab and gemm_splitk_reduce_kernel folds the slabs in a fixed order (deterministic; also
// applies alpha / bias / accumulate).  Staging: 16-byte global loads along whichever axis is contiguous in memory
// (k, or the row / column axis - then transposed on the way into LDS), scalar loads for tails and unaligned rows.
// LDS tiles are [row][k] with k contiguous and a 36-float pitch: lane (r, h) reads k = 8g + 4h .. 8g + 4h + 3 with one
// ds_read_b128 per operand and feeds four v_mfma_f32_32x32x2_f32 (the two lane halves supply the two k of each MFMA;
// any pairing is a valid sum as long as A and B agree).
#define GBM 64
#define GBN 64
#define GBK 32
#define GLP 36   // LDS row pitch in floats (16-byte aligned rows, bank-staggered)

// stage a [64 rows][32 k] tile: element (row, k) = src[row * s_row + k * s_k] for row < rows, k < ks, else 0
static __device__ __forceinline__ void gemm_stage(float (*dst)[GLP], const float* __restrict__ src, long long s_row,
                                                  long long s_k, int rows, int ks, int tid) {
    const bool al = (((size_t)src) & 15) == 0;
    if (s_k == 1 && al && (s_row & 3) == 0 && ks == GBK) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {            // 64 rows x 8 float4
            const int row = p * 32 + (tid >> 3), kq = (tid & 7) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < rows) v = *(const f32x4*)(src + row * s_row + kq);
            *(f32x4*)&dst[row][kq] = v;
        }
    } else if (s_row == 1 && al && (s_k & 3) == 0 && rows == GBM) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {            // 32 k x 16 float4 along the rows
            const int k = p * 16 + (tid >> 4), rq = (tid & 15) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < ks) v = *(const f32x4*)(src + k * s_k + rq);
            dst[rq][k] = v[0];
            dst[rq + 1][k] = v[1];
            dst[rq + 2][k] = v[2];
            dst[rq + 3][k] = v[3];
        }
    } else if (s_row == 1) {                     // rows contiguous: consecutive lanes read consecutive rows
        for (int i = tid; i < GBM * GBK; i += 256) {
            const int row = i & 63, k = i >> 6;
            dst[row][k] = (row < rows && k < ks) ? src[row + k * s_k] : 0.f;
        }
    } else {                                      // k contiguous (or a general stride): consecutive lanes read consecutive k
        for (int i = tid; i < GBM * GBK; i += 256) {
            const int k = i & 31, row = i >> 5;
            dst[row][k] = (row < rows && k < ks) ? src[row * s_row + k * s_k] : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                       float* __restrict__ out, int M, int N, int K, long long sam,
                                                       long long sak, long long sbk, long long sbn, long long ldo,
                                                       long long slab, int kper) {
    __shared__ __attribute__((aligned(16))) float As[GBM][GLP];
    __shared__ __attribute__((aligned(16))) float Bs[GBN][GLP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const int kbeg = blockIdx.z * kper, kend = min(K, kbeg + kper);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        __syncthreads();
        gemm_stage(As, A + m0 * sam + k0 * sak, sam, sak, min(GBM, M - m0), min(GBK, kend - k0), tid);
        gemm_stage(Bs, Bm + n0 * sbn + k0 * sbk, sbn, sbk, min(GBN, N - n0), min(GBK, kend - k0), tid);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < GBK / 8; ++g) {
            const f32x4 av = *(const f32x4*)&As[wm * 32 + r][g * 8 + h * 4];
            const f32x4 bv = *(const f32x4*)&Bs[wn * 32 + r][g * 8 + h * 4];
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
        }
    }
    float* dst = out + (size_t)blockIdx.z * slab;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, n = n0 + wn * 32 + r;
        if (m < M && n < N) dst[m * ldo + n] = acc[e];
    }
}

// C[m][n] = alpha * sum_z part[z][m][n] (+ bias[n]) (+ C[m][n]); slabs folded in index order
__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ Cm,
                                          const float* __restrict__ bias, int M, int N, int nsplit, long long ldc, float alpha,
                                          int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * N) return;
    const int m = i / N, n = i - m * N;
    double acc = 0.0;                      // the slices are few: folding them in fp64 costs nothing and removes one rounding chain
    for (int z = 0; z < nsplit; ++z) acc += (double)part[(size_t)z * M * N + i];
    float v = (float)acc * alpha;
    if (bias) v += bias[n];
    float* dst = Cm + m * ldc + n;
    *dst = accumulate ? *dst + v : v;
}

// number of K slices the launcher will use for a problem (the caller sizes the workspace with it)
extern "C" int spk_gemm_splitk(int M, int N, int K) {
    const int tiles = spk_ceil_div(M, GBM) * spk_ceil_div(N, GBN);
    const int chunks = spk_ceil_div(K, GBK);
    int want = spk_ceil_div(512, tiles);            // ~2 blocks per CU
    if (want > chunks / 2) want = chunks / 2;       // at least two K chunks per slice
    if (want < 1) want = 1;
    if (want > 32) want = 32;
    return want;
}

extern "C" size_t spk_gemm_workspace(int M, int N, int K) {
    return (size_t)spk_gemm_splitk(M, N, K) * M * N * sizeof(float);
}

// workspace: spk_gemm_workspace(M, N, K) bytes (may be NULL when spk_gemm_splitk(M, N, K) == 1 and alpha == 1, no bias,
// no accumulate - then the tile is written straight to C)
extern "C" int spk_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long long sam,
                            long long sak, long long sbk, long long sbn, long long ldc, float alpha, int accumulate,
                            float* workspace, void* stream) {
    SPK_REQUIRE(A && B && C, "spk_gemm_f32: null pointer");
    SPK_REQUIRE(M > 0 && N > 0 && K > 0, "spk_gemm_f32: empty problem %dx%dx%d", M, N, K);
    SPK_REQUIRE(ldc >= N, "spk_gemm_f32: ldc=%lld < N=%d", ldc, N);
    const int nsplit = spk_gemm_splitk(M, N, K);
    const bool direct = nsplit == 1 && alpha == 1.f && !bias && !accumulate;
    SPK_REQUIRE(direct || workspace, "spk_gemm_f32: this problem needs a workspace of spk_gemm_workspace() bytes");
    const int kper = spk_ceil_div(spk_ceil_div(K, nsplit), GBK) * GBK;
    dim3 grid(spk_ceil_div(N, GBN), spk_ceil_div(M, GBM), nsplit);
    hipStream_t st = (hipStream_t)stream;
    if (direct) {
        hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, st, A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, 0LL, kper);
        SPK_LAUNCH_CHECK("spk_gemm_f32");
        return 0;
    }
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, st, A, B, workspace, M, N, K, sam, sak, sbk, sbn, (long long)N,
                       (long long)M * N, kper);
    SPK_LAUNCH_CHECK("spk_gemm_f32");
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(spk_ceil_div(M * N, 256)), dim3(256), 0, st, workspace, C, bias, M, N, nsplit,
                       ldc, alpha, accumulate);
    SPK_LAUNCH_CHECK("spk_gemm_f32(reduce)");
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
