// Statistics pooling over time (reference StatsPooling, scripts/model.py:435-457) on the NHWC trunk
// output x[B][H][W][C] (H = freq/8, W = frames/8).  One thread per (b,h,c), c fastest, so every
// step over w is a coalesced row of C floats; two passes (mean, then centred second moment).
//   'mean+std' (mode 1): out[b][c*2H + h] = unbiased var over W,  out[b][c*2H + H + h] = sqrt(mean over W)
//                        -- the reference binds torch.var_mean's (var, mean) to (mean, var) at :450, so the
//                        layer really emits cat([var, sqrt(mean)]); reproduced exactly.
//   'mean'     (mode 0): out[b][c*H + h] = mean over W    (AdaptiveAvgPool2d((None,1)), :439)
// The output index order is nn.Flatten(1,-1) of the reference's [B,C,2H] / [B,C,H,1] tensors (:352).
#include "spk_common.h"

__global__ __launch_bounds__(256) void stats_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B,
                                                             int H, int W, int C, int mode) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long bh = idx / C;
    const int h = (int)(bh % H);
    const int b = (int)(bh / H);
    const float* p = x + (size_t)bh * W * C + c;
    float s = 0.f;
    for (int w = 0; w < W; ++w) s += p[(size_t)w * C];
    const float mean = s / (float)W;
    if (mode == 0) {
        out[(size_t)b * C * H + (size_t)c * H + h] = mean;
        return;
    }
    float m2 = 0.f;
    for (int w = 0; w < W; ++w) {
        const float d = p[(size_t)w * C] - mean;
        m2 = fmaf(d, d, m2);
    }
    float* o = out + (size_t)b * C * 2 * H + (size_t)c * 2 * H;
    o[h] = m2 / (float)(W - 1);
    o[H + h] = sqrtf(mean);
}

// dx = gvar * 2(x - mean)/(W-1) + gsqrt / (2 sqrt(mean)) / W      (IEEE semantics kept: mean == 0 gives inf/nan
// exactly like torch's sqrt backward)
__global__ __launch_bounds__(256) void stats_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                             float* __restrict__ dx, int B, int H, int W, int C, int mode,
                                                             unsigned* __restrict__ amax_out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * C;
    float mx = 0.f;                      // absmax of what this thread stores (the operand scale of the f16x3 consumers of dx)
    if (idx < total) {
        const int c = (int)(idx % C);
        const long long bh = idx / C;
        const int h = (int)(bh % H);
        const int b = (int)(bh / H);
        const float* p = x + (size_t)bh * W * C + c;
        float* q = dx + (size_t)bh * W * C + c;
        if (mode == 0) {
            const float g = gout[(size_t)b * C * H + (size_t)c * H + h] / (float)W;
            for (int w = 0; w < W; ++w) q[(size_t)w * C] = g;
            mx = spk_finite_abs(g);
        } else {
            float s = 0.f;
            for (int w = 0; w < W; ++w) s += p[(size_t)w * C];
            const float mean = s / (float)W;
            const float* g = gout + (size_t)b * C * 2 * H + (size_t)c * 2 * H;
            const float gv = g[h] * (2.f / (float)(W - 1));
            const float gm = g[H + h] / (2.f * sqrtf(mean)) / (float)W;
            for (int w = 0; w < W; ++w) {
                const float v = fmaf(gv, p[(size_t)w * C] - mean, gm);
                q[(size_t)w * C] = v;
                // sqrt'(0) = inf: a row whose mean over time is exactly 0 (a dead post-ReLU channel row) gets inf / NaN here, as in
                // the reference (torch.sqrt backward); the ReLU mask of that row is 0, so the select in the BatchNorm backward
                // drops those values (threshold_backward semantics) - they must not poison the operand scale of everything else
                mx = fmaxf(mx, spk_finite_abs(v));
            }
        }
    }
    if (amax_out) spk_wave_amax_commit(mx, amax_out);     // every lane of the wave takes part (no early return above)
}

extern "C" int spk_stats_pool_fwd(const float* x, float* out, int B, int H, int W, int C, int mode, void* stream) {
    SPK_REQUIRE(x && out, "spk_stats_pool_fwd: null pointer");
    SPK_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "spk_stats_pool_fwd: empty input");
    SPK_REQUIRE(mode == 0 || mode == 1, "spk_stats_pool_fwd: mode=%d", mode);
    const long long total = (long long)B * H * C;
    hipLaunchKernelGGL(stats_pool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                       B, H, W, C, mode);
    SPK_LAUNCH_CHECK("spk_stats_pool_fwd");
    return 0;
}

extern "C" int spk_stats_pool_bwd(const float* x, const float* gout, float* dx, int B, int H, int W, int C, int mode,
                                  unsigned* amax_out, void* stream) {
    SPK_REQUIRE(x && gout && dx, "spk_stats_pool_bwd: null pointer");
    SPK_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "spk_stats_pool_bwd: empty input");
    SPK_REQUIRE(mode == 0 || mode == 1, "spk_stats_pool_bwd: mode=%d", mode);
    const long long total = (long long)B * H * C;
    hipLaunchKernelGGL(stats_pool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, gout,
                       dx, B, H, W, C, mode, amax_out);
    SPK_LAUNCH_CHECK("spk_stats_pool_bwd");
    return 0;
}
