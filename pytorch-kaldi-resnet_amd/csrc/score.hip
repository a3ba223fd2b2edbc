// Cosine scoring back end on the device: the step after the embedding path.
//   spk_center_normalize - v = emb - mean;  out = v / max(||v||, eps)          (reference scripts/cosine_score.py:52-65:
//                          mean subtraction + F.cosine_similarity, eps 1e-8; scripts/compute_topk_mean_std.py:13-17:
//                          F.normalize, eps 1e-12)
//   spk_trial_cosine     - out[t] = <en[ia[t]], te[ib[t]]> for rows already normalised (one wave per trial)
//   spk_topk_mean_std    - per row of a score matrix: mean and unbiased std of its k largest entries
//                          (scripts/compute_topk_mean_std.py:18-21: scores.topk(300) + torch.std_mean)
// All three are HBM-bound streams over [N][D] tables (D = 256 for this model); the cohort score matrix itself is
// one spk_gemm_f32 call.
#include "spk_common.h"

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void center_normalize_kernel(const float* __restrict__ emb, const float* __restrict__ mean,
                                                               float* __restrict__ out, int N, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* src = emb + (size_t)row * D;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = src[d] - (mean ? mean[d] : 0.f);
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float inv = 1.f / fmaxf(sqrtf(ss), eps);
    float* dst = out + (size_t)row * D;
    for (int d = lane; d < D; d += 64) dst[d] = (src[d] - (mean ? mean[d] : 0.f)) * inv;
}

extern "C" int spk_center_normalize(const float* emb, const float* mean, float* out, int N, int D, float eps, void* stream) {
    SPK_REQUIRE(emb && out && N > 0 && D > 0, "spk_center_normalize: bad arguments");
    hipLaunchKernelGGL(center_normalize_kernel, dim3(spk_ceil_div(N, 4)), dim3(256), 0, (hipStream_t)stream, emb, mean, out, N,
                       D, eps);
    SPK_LAUNCH_CHECK("spk_center_normalize");
    return 0;
}

__global__ __launch_bounds__(256) void trial_cosine_kernel(const float* __restrict__ en, const float* __restrict__ te,
                                                           const int* __restrict__ ia, const int* __restrict__ ib,
                                                           float* __restrict__ out, int T, int D) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float* a = en + (size_t)ia[t] * D;
    const float* b = te + (size_t)ib[t] * D;
    float s = 0.f;
    if ((D & 3) == 0) {
        for (int d = lane * 4; d < D; d += 256) {
            const f32x4 x = *(const f32x4*)(a + d), y = *(const f32x4*)(b + d);
            s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
        }
    } else {
        for (int d = lane; d < D; d += 64) s += a[d] * b[d];
    }
    s = wave_sum(s);
    if (lane == 0) out[t] = s;
}

extern "C" int spk_trial_cosine(const float* en, const float* te, const int* ia, const int* ib, float* out, int T, int D,
                                int n_en, int n_te, void* stream) {
    SPK_REQUIRE(en && te && ia && ib && out && T > 0 && D > 0 && n_en > 0 && n_te > 0, "spk_trial_cosine: bad arguments");
    (void)n_en; (void)n_te;   // index ranges are validated by the caller that built the index arrays (host side)
    hipLaunchKernelGGL(trial_cosine_kernel, dim3(spk_ceil_div(T, 4)), dim3(256), 0, (hipStream_t)stream, en, te, ia, ib, out, T,
                       D);
    SPK_LAUNCH_CHECK("spk_trial_cosine");
    return 0;
}

// One block per row: the row is sorted (descending, bitonic network in LDS, padded with -inf to a power of two) and
// the first k entries are averaged.  M <= 16384 (64 KiB of LDS).
__global__ __launch_bounds__(256) void topk_mean_std_kernel(const float* __restrict__ scores, float* __restrict__ mean_out,
                                                            float* __restrict__ std_out, int M, int P, int k, long long ld) {
    extern __shared__ float row[];
    __shared__ float part[4];
    const int tid = threadIdx.x;
    const float* src = scores + (size_t)blockIdx.x * ld;
    for (int i = tid; i < P; i += 256) row[i] = i < M ? src[i] : -INFINITY;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (P >> 1); i += 256) {
                const int lo = 2 * i - (i & (stride - 1));     // index with the `stride` bit cleared
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const float x = row[lo], y = row[hi];
                if ((x < y) == desc) {
                    row[lo] = y;
                    row[hi] = x;
                }
            }
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = tid; i < k; i += 256) s += row[i];
    s = wave_sum(s);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    const float mu = (part[0] + part[1] + part[2] + part[3]) / (float)k;
    __syncthreads();
    float q = 0.f;
    for (int i = tid; i < k; i += 256) {
        const float d = row[i] - mu;
        q += d * d;
    }
    q = wave_sum(q);
    if ((tid & 63) == 0) part[tid >> 6] = q;
    __syncthreads();
    if (tid == 0) {
        mean_out[blockIdx.x] = mu;
        std_out[blockIdx.x] = sqrtf((part[0] + part[1] + part[2] + part[3]) / (float)(k - 1));
    }
}

extern "C" int spk_topk_mean_std(const float* scores, float* mean_out, float* std_out, int N, int M, int k, long long ld,
                                 void* stream) {
    SPK_REQUIRE(scores && mean_out && std_out && N > 0, "spk_topk_mean_std: bad arguments");
    SPK_REQUIRE(k >= 2 && k <= M, "spk_topk_mean_std: k=%d must lie in [2, M=%d]", k, M);
    SPK_REQUIRE(M <= 16384 && ld >= M, "spk_topk_mean_std: M=%d exceeds 16384 columns (or ld=%lld < M)", M, ld);
    int P = 2;
    while (P < M) P <<= 1;
    hipLaunchKernelGGL(topk_mean_std_kernel, dim3(N), dim3(256), (size_t)P * sizeof(float), (hipStream_t)stream, scores,
                       mean_out, std_out, M, P, k, ld);
    SPK_LAUNCH_CHECK("spk_topk_mean_std");
    return 0;
}
