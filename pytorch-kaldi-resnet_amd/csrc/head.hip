// Classification heads: row L2-normalisation, AAM margin (reference AAMLayer, scripts/model.py:459-501),
// softmax cross-entropy (nn.CrossEntropyLoss, scripts/train_resnet.py:201,317) and top-k rank for
// accuracy (scripts/accuracy.py:4-16).  The cosine / linear products themselves run in gemm.hip.
#include "spk_common.h"

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// y = x / max(||x||, eps), one wave per row  (F.normalize, scripts/model.py:485)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ invn, int R, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* p = x + (size_t)row * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s = fmaf(p[i], p[i], s);
    s = wave_sum(s);
    const float inv = 1.f / fmaxf(sqrtf(s), eps);
    for (int i = lane; i < D; i += 64) y[(size_t)row * D + i] = p[i] * inv;
    if (lane == 0) invn[row] = inv;
}

// dx = (dy - y*(y.dy)) * inv   (inv = 1/||x||);  when the norm was clamped (inv == 1/eps) dx = dy*inv
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ invn,
                                                         const float* __restrict__ dy, float* __restrict__ dx, int R, int D,
                                                         float eps, int accumulate) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* yp = y + (size_t)row * D;
    const float* gp = dy + (size_t)row * D;
    float dot = 0.f;
    for (int i = lane; i < D; i += 64) dot = fmaf(yp[i], gp[i], dot);
    dot = wave_sum(dot);
    const float inv = invn[row];
    if (inv >= 1.f / eps) dot = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float v = (gp[i] - yp[i] * dot) * inv;
        float* d = dx + (size_t)row * D + i;
        *d = accumulate ? *d + v : v;
    }
}

extern "C" int spk_l2norm_fwd(const float* x, float* y, float* inv_norm, int R, int D, float eps, void* stream) {
    SPK_REQUIRE(x && y && inv_norm && R > 0 && D > 0, "spk_l2norm_fwd: bad arguments");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(spk_ceil_div(R, 4)), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, R, D, eps);
    SPK_LAUNCH_CHECK("spk_l2norm_fwd");
    return 0;
}
extern "C" int spk_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, float* dx, int R, int D, float eps,
                              int accumulate, void* stream) {
    SPK_REQUIRE(y && inv_norm && dy && dx && R > 0 && D > 0, "spk_l2norm_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(spk_ceil_div(R, 4)), dim3(256), 0, (hipStream_t)stream, y, inv_norm, dy, dx, R, D,
                       eps, accumulate);
    SPK_LAUNCH_CHECK("spk_l2norm_bwd");
    return 0;
}

// logits = s * (j == label ? phi(cos) : cos)   (scripts/model.py:487-499, easy_margin = False)
__global__ void aam_margin_fwd_kernel(const float* __restrict__ cosv, const long long* __restrict__ label,
                                      float* __restrict__ logits, int Bn, int S, float cos_m, float sin_m, float th, float mm,
                                      float s) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Bn * S) return;
    const int b = (int)(idx / S), j = (int)(idx - (long long)b * S);
    const float c = cosv[idx];
    float v = c;
    if ((long long)j == label[b]) {
        const float sine = sqrtf(fminf(fmaxf(1.f - c * c, 0.f), 1.f));
        const float phi = c * cos_m - sine * sin_m;
        v = (c - th) > 0.f ? phi : c - mm;
    }
    logits[idx] = v * s;
}

// dcos = s * dlogits * (j == label ? dphi/dcos : 1), following torch autograd of the expression above:
// d sqrt(u)/du = 1/(2 sqrt(u)); clamp passes the gradient for 0 <= u <= 1; where() routes it to the taken branch.
__global__ void aam_margin_bwd_kernel(const float* __restrict__ cosv, const long long* __restrict__ label,
                                      const float* __restrict__ dlogits, float* __restrict__ dcos, int Bn, int S, float cos_m,
                                      float sin_m, float th, float s) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Bn * S) return;
    const int b = (int)(idx / S), j = (int)(idx - (long long)b * S);
    float g = dlogits[idx] * s;
    if ((long long)j == label[b]) {
        const float c = cosv[idx];
        if ((c - th) > 0.f) {
            const float u = 1.f - c * c;
            const float uc = fminf(fmaxf(u, 0.f), 1.f);
            const float sine = sqrtf(uc);
            float dphi = cos_m;
            if (u >= 0.f && u <= 1.f) dphi += (-sin_m) * (1.f / (2.f * sine)) * (-2.f * c);
            g *= dphi;
        }
    }
    dcos[idx] = g;
}

extern "C" int spk_aam_margin_fwd(const float* cosv, const long long* label, float* logits, int B, int S, float m, float s,
                                  void* stream) {
    SPK_REQUIRE(cosv && label && logits && B > 0 && S > 0, "spk_aam_margin_fwd: bad arguments");
    const double pi = 3.14159265358979323846;
    const float cos_m = (float)cos((double)m), sin_m = (float)sin((double)m);
    const float th = (float)cos(pi - (double)m), mm = (float)(sin(pi - (double)m) * (double)m);
    const long long total = (long long)B * S;
    hipLaunchKernelGGL(aam_margin_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cosv,
                       label, logits, B, S, cos_m, sin_m, th, mm, s);
    SPK_LAUNCH_CHECK("spk_aam_margin_fwd");
    return 0;
}
extern "C" int spk_aam_margin_bwd(const float* cosv, const long long* label, const float* dlogits, float* dcos, int B, int S,
                                  float m, float s, void* stream) {
    SPK_REQUIRE(cosv && label && dlogits && dcos && B > 0 && S > 0, "spk_aam_margin_bwd: bad arguments");
    const double pi = 3.14159265358979323846;
    const float cos_m = (float)cos((double)m), sin_m = (float)sin((double)m);
    const float th = (float)cos(pi - (double)m);
    const long long total = (long long)B * S;
    hipLaunchKernelGGL(aam_margin_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cosv,
                       label, dlogits, dcos, B, S, cos_m, sin_m, th, s);
    SPK_LAUNCH_CHECK("spk_aam_margin_bwd");
    return 0;
}

// One block per row: loss_row = logsumexp(row) - row[label]; dlogits = (softmax - onehot) * gscale;
// rank = #{j : row[j] > row[label]} (top-k accuracy: correct@k <=> rank < k).
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                         float* __restrict__ loss_row, float* __restrict__ dlogits,
                                                         int* __restrict__ rank, int S, float gscale) {
    __shared__ float sred[4];
    __shared__ int ired[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (size_t)b * S;
    const int lab = (int)label[b];
    const float tgt = row[lab];
    float mx = -INFINITY;
    int cnt = 0;
    for (int j = tid; j < S; j += 256) {
        const float v = row[j];
        mx = fmaxf(mx, v);
        cnt += (v > tgt) ? 1 : 0;
    }
    mx = wave_max(mx);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) {
        sred[wave] = mx;
        ired[wave] = cnt;
    }
    __syncthreads();
    mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    cnt = ired[0] + ired[1] + ired[2] + ired[3];
    __syncthreads();
    float se = 0.f;
    for (int j = tid; j < S; j += 256) se += expf(row[j] - mx);
    se = wave_sum(se);
    if (lane == 0) sred[wave] = se;
    __syncthreads();
    se = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    const float lse = mx + logf(se);
    if (tid == 0) {
        loss_row[b] = lse - tgt;
        if (rank) rank[b] = cnt;
    }
    if (dlogits) {
        float* d = dlogits + (size_t)b * S;
        for (int j = tid; j < S; j += 256) {
            float p = expf(row[j] - lse);
            if (j == lab) p -= 1.f;
            d[j] = p * gscale;
        }
    }
}

extern "C" int spk_softmax_ce(const float* logits, const long long* label, float* loss_row, float* dlogits, int* rank, int B,
                              int S, float grad_scale, void* stream) {
    SPK_REQUIRE(logits && label && loss_row && B > 0 && S > 0, "spk_softmax_ce: bad arguments");
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, label, loss_row, dlogits, rank, S,
                       grad_scale);
    SPK_LAUNCH_CHECK("spk_softmax_ce");
    return 0;
}

// mean of a small vector in fixed order (loss = mean_b loss_row[b]); single block
__global__ void mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)n);
}
extern "C" int spk_mean(const float* v, float* out, int n, void* stream) {
    SPK_REQUIRE(v && out && n > 0, "spk_mean: bad arguments");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v, out, n);
    SPK_LAUNCH_CHECK("spk_mean");
    return 0;
}

// elementwise relu forward / backward on [n] (BatchNorm1d + ReLU heads reuse bn.hip for the BN part)
__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
extern "C" int spk_relu_bwd(const float* y, const float* dy, float* dx, long long n, void* stream) {
    SPK_REQUIRE(y && dy && dx && n > 0, "spk_relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n);
    SPK_LAUNCH_CHECK("spk_relu_bwd");
    return 0;
}
