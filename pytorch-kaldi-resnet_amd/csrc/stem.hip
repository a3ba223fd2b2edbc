// Stem convolution Conv2d(1,32,3,stride 1,pad 1,bias=False) (reference scripts/model.py:210,249) and its
// weight gradient.  Cin = 1, K = 9: this layer is HBM-bound (AI ~ 4 flop/B), so it is a direct
// convolution, not a GEMM: a thread owns (pixel, 8 output channels) with its 72 weights in registers;
// four neighbouring threads write one pixel's 32 channels = 128 contiguous bytes of the NHWC output.
// Input x is [B][F][T] fp32 (freq-major, time innermost; viewed as NCHW [B,1,F,T] by the reference).
#include "spk_common.h"

#define STEM_C 32

__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ out, float* __restrict__ stats,
                                                       const float* __restrict__ epi_scale,
                                                       const float* __restrict__ epi_shift, int B, int F, int T,
                                                       int flags, unsigned* __restrict__ amax_out) {
    __shared__ float red[4][STEM_C][2];
    float mx = 0.f;
    const int tid = threadIdx.x, cg = tid & 3, pl = tid >> 2;
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 8; ++c) wr[t][c] = w[(cg * 8 + c) * 9 + t];
    float es[8], eh[8], ssum[8], ssq[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        es[c] = (flags & SPK_EPI_AFFINE) ? epi_scale[cg * 8 + c] : 1.f;
        eh[c] = (flags & SPK_EPI_AFFINE) ? epi_shift[cg * 8 + c] : 0.f;
        ssum[c] = 0.f;
        ssq[c] = 0.f;
    }
    const int FT = F * T;
    const long long NP = (long long)B * FT;
    for (long long p = (long long)blockIdx.x * 64 + pl; p < NP; p += (long long)gridDim.x * 64) {
        const int b = (int)(p / FT);
        const int rem = (int)(p - (long long)b * FT);
        const int f = rem / T, t0 = rem - f * T;
        const float* xb = x + (size_t)b * FT;
        float xv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ff = f + kh - 1, tt = t0 + kw - 1;
                xv[kh * 3 + kw] = (ff >= 0 && ff < F && tt >= 0 && tt < T) ? xb[ff * T + tt] : 0.f;
            }
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) v = fmaf(xv[t], wr[t][c], v);
            if (flags & SPK_EPI_AFFINE) v = v * es[c] + eh[c];
            if (flags & SPK_EPI_RELU) v = fmaxf(v, 0.f);
            o[c] = v;
            mx = fmaxf(mx, fabsf(v));
            ssum[c] += v;
            ssq[c] += v * v;
        }
        f32x4* dst = (f32x4*)(out + (size_t)p * STEM_C + cg * 8);
        dst[0] = (f32x4){o[0], o[1], o[2], o[3]};
        dst[1] = (f32x4){o[4], o[5], o[6], o[7]};
    }
    if (amax_out) spk_wave_amax_commit(mx, amax_out);     // absmax(out): the operand scale of its f16x3 consumers
    if (flags & SPK_EPI_STATS) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) {
                ssum[c] += __shfl_xor(ssum[c], off, 64);
                ssq[c] += __shfl_xor(ssq[c], off, 64);
            }
        }
        const int lane = tid & 63, wave = tid >> 6;
        if (lane < 4) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                red[wave][lane * 8 + c][0] = ssum[c];
                red[wave][lane * 8 + c][1] = ssq[c];
            }
        }
        __syncthreads();
        if (tid < STEM_C) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                s0 += red[wv][tid][0];
                s1 += red[wv][tid][1];
            }
            stats[((size_t)blockIdx.x * STEM_C + tid) * 2 + 0] = s0;
            stats[((size_t)blockIdx.x * STEM_C + tid) * 2 + 1] = s1;
        }
    }
}

extern "C" int spk_stem_fwd_blocks(int B, int F, int T) {
    long long np = (long long)B * F * T;
    long long nb = (np + 63) / 64;
    return (int)(nb < 2048 ? nb : 2048);
}

extern "C" int spk_stem_conv_fwd(const float* x, const float* w, float* out, float* stats, const float* epi_scale,
                                 const float* epi_shift, int B, int F, int T, int flags, unsigned* amax_out, void* stream) {
    SPK_REQUIRE(x && w && out, "spk_stem_conv_fwd: null pointer");
    SPK_REQUIRE(B > 0 && F > 0 && T > 0, "spk_stem_conv_fwd: empty input");
    SPK_REQUIRE(!(flags & SPK_EPI_STATS) || stats, "spk_stem_conv_fwd: EPI_STATS needs a stats buffer");
    SPK_REQUIRE(!(flags & SPK_EPI_AFFINE) || (epi_scale && epi_shift), "spk_stem_conv_fwd: EPI_AFFINE needs scale/shift");
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(spk_stem_fwd_blocks(B, F, T)), dim3(256), 0, (hipStream_t)stream, x, w, out,
                       stats, epi_scale, epi_shift, B, F, T, flags, amax_out);
    SPK_LAUNCH_CHECK("spk_stem_conv_fwd");
    return 0;
}

// dW[c][tap] = sum_p x[p + tap] * draw[p][c]; per-block partials [nblk][32][9] then a fixed-order sum.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ draw,
                                                         float* __restrict__ partial, int B, int F, int T) {
    __shared__ float red[4][STEM_C * 9];
    const int tid = threadIdx.x, cg = tid & 3, pl = tid >> 2;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[t][c] = 0.f;
    const int FT = F * T;
    const long long NP = (long long)B * FT;
    for (long long p = (long long)blockIdx.x * 64 + pl; p < NP; p += (long long)gridDim.x * 64) {
        const int b = (int)(p / FT);
        const int rem = (int)(p - (long long)b * FT);
        const int f = rem / T, t0 = rem - f * T;
        const float* xb = x + (size_t)b * FT;
        float xv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ff = f + kh - 1, tt = t0 + kw - 1;
                xv[kh * 3 + kw] = (ff >= 0 && ff < F && tt >= 0 && tt < T) ? xb[ff * T + tt] : 0.f;
            }
        const f32x4* src = (const f32x4*)(draw + (size_t)p * STEM_C + cg * 8);
        const f32x4 d0 = src[0], d1 = src[1];
        const float d[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[t][c] = fmaf(xv[t], d[c], acc[t][c]);
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = acc[t][c];
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            if (lane < 4) red[wave][(lane * 8 + c) * 9 + t] = v;
        }
    __syncthreads();
    for (int i = tid; i < STEM_C * 9; i += 256)
        partial[(size_t)blockIdx.x * (STEM_C * 9) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// one block per weight element: 256 threads fold the per-block partials in fp64, fixed order
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                int nblk, int accumulate) {
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) s += (double)partial[(size_t)k * (STEM_C * 9) + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) dw[i] = accumulate ? dw[i] + (float)red[0] : (float)red[0];
}

extern "C" int spk_stem_wgrad_blocks(int B, int F, int T) {
    long long np = (long long)B * F * T;
    long long nb = (np + 63) / 64;
    return (int)(nb < 1024 ? nb : 1024);
}

extern "C" int spk_stem_conv_wgrad(const float* x, const float* draw, float* dw, float* partial, int B, int F, int T,
                                   int accumulate, void* stream) {
    SPK_REQUIRE(x && draw && dw && partial, "spk_stem_conv_wgrad: null pointer");
    SPK_REQUIRE(B > 0 && F > 0 && T > 0, "spk_stem_conv_wgrad: empty input");
    const int nblk = spk_stem_wgrad_blocks(B, F, T);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, draw, partial, B, F, T);
    SPK_LAUNCH_CHECK("spk_stem_conv_wgrad");
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(STEM_C * 9), dim3(256), 0, (hipStream_t)stream, partial, dw, nblk, accumulate);
    SPK_LAUNCH_CHECK("spk_stem_wgrad_reduce");
    return 0;
}
