// BatchNorm over NHWC tensors viewed as [N rows][C channels] (nn.BatchNorm2d / BatchNorm1d of the
// reference, scripts/model.py:41,44,212,235,361; eps 1e-5, momentum 0.1).  All kernels are HBM-bound
// float4 streams; reductions are two-level (fp32 per block -> fp64 fixed-order finalize), so they are
// deterministic and do not lose precision over the 6.1 M-element layer-1 reductions.
//
// Forward (train):  conv epilogue or bn_stats_partial -> per-block (sum, sumsq) -> bn_finalize ->
//                   (mean, invstd, scale = gamma*invstd, shift = beta - mean*scale, running stats)
//                   -> bn_apply [+ residual] [+ ReLU]  (or fused into the next conv's input staging)
// Backward:         bn_bwd_reduce -> (sum dz, sum dz*xhat) -> bn_bwd_finalize -> dgamma, dbeta, coefficients
//                   -> bn_bwd_apply: draw = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat))
#include "spk_common.h"

enum { MASK_NONE = 0, MASK_ACT = 1, MASK_RAW = 2, MASK_BITS = 3 };   // MASK_BITS: `act` holds sign bits, [pixel][C/32] words

// ---- statistics of a plain [N][C] tensor ---------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                               long long N, int C, int rows_per_block) {
    __shared__ float red[256][8];
    const int tid = threadIdx.x;
    const int qpr = C >> 2;            // float4 quads per row
    const int quad = tid % qpr, prow = tid / qpr, rstep = 256 / qpr;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
    for (long long r = r0 + prow; r < r1; r += rstep) {
        const f32x4 v = *(const f32x4*)(x + r * C + quad * 4);
        s += v;
        ss += v * v;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[tid][k] = s[k];
        red[tid][4 + k] = ss[k];
    }
    __syncthreads();
    if (tid < qpr) {
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = 0.f;
        for (int p = 0; p < rstep; ++p)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += red[p * qpr + tid][k];
        float* dst = partial + ((size_t)blockIdx.x * C + tid * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[k * 2 + 0] = a[k];
            dst[k * 2 + 1] = a[4 + k];
        }
    }
}

static bool bn_c_ok(int C) { return C >= 4 && C <= 1024 && (C & (C - 1)) == 0; }

#ifndef BN_STATS_BLOCKS
#define BN_STATS_BLOCKS 4096
#endif
extern "C" int spk_bn_stats_blocks(long long N, int C) {
    (void)C;
    long long nb = (N + 255) / 256;
    if (nb > BN_STATS_BLOCKS) nb = BN_STATS_BLOCKS;
    return (int)(nb < 1 ? 1 : nb);
}

extern "C" int spk_bn_stats_partial(const float* x, float* partial, long long N, int C, void* stream) {
    SPK_REQUIRE(x && partial, "spk_bn_stats_partial: null pointer");
    SPK_REQUIRE(N > 0 && bn_c_ok(C), "spk_bn_stats_partial: N=%lld C=%d (C must be a power of two in [4,1024])", N, C);
    const int nb = spk_bn_stats_blocks(N, C);
    const int rpb = (int)((N + nb - 1) / nb);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, partial, N, C, rpb);
    SPK_LAUNCH_CHECK("spk_bn_stats_partial");
    return 0;
}

// ---- finalize: partial [nblk][C][2] -> mean/invstd/scale/shift (+ running statistics) ---------------
// Layer-1 convolutions emit ~100 k partial rows (25 MB); a first stage folds them to 256 fp64 rows in parallel
// (256 x C/32 blocks), the second stage (one block per 32 channels) finishes in a fixed order.
#define BN_STAGE_ROWS 256
__global__ __launch_bounds__(256) void bn_fold_partials_kernel(const float* __restrict__ partial, double* __restrict__ ws,
                                                               int nblk, int C) {
    __shared__ double red[8][32][2];
    const int tid = threadIdx.x, c = tid & 31, row = tid >> 5;
    const int ch = blockIdx.x * 32 + c;
    const int per = (nblk + BN_STAGE_ROWS - 1) / BN_STAGE_ROWS;
    const int k0 = blockIdx.y * per;
    int k1 = k0 + per;
    if (k1 > nblk) k1 = nblk;
    double s = 0.0, ss = 0.0;
    if (ch < C) {
        const size_t pitch = (size_t)8 * C * 2;
        const float* q = partial + ((size_t)(k0 + row) * C + ch) * 2;
        int k = k0 + row;
        for (; k + 24 < k1; k += 32, q += 4 * pitch) {      // four loads in flight, serial order of the sums
            const float2 a = *(const float2*)q, b = *(const float2*)(q + pitch);
            const float2 c2 = *(const float2*)(q + 2 * pitch), d = *(const float2*)(q + 3 * pitch);
            s += (double)a.x; ss += (double)a.y;
            s += (double)b.x; ss += (double)b.y;
            s += (double)c2.x; ss += (double)c2.y;
            s += (double)d.x; ss += (double)d.y;
        }
        for (; k < k1; k += 8, q += pitch) {
            const float2 p = *(const float2*)q;
            s += (double)p.x;
            ss += (double)p.y;
        }
    }
    red[row][c][0] = s;
    red[row][c][1] = ss;
    __syncthreads();
    if (row == 0 && ch < C) {
        for (int k = 1; k < 8; ++k) {
            s += red[k][c][0];
            ss += red[k][c][1];
        }
        ws[((size_t)blockIdx.y * C + ch) * 2 + 0] = s;
        ws[((size_t)blockIdx.y * C + ch) * 2 + 1] = ss;
    }
}

// Rows row, row + 8, ... of one channel.  Four loads are issued before the first is consumed (a dependent chain of 32 loads was
// 10 of the 13 us of a 256-row finalize); the sums keep the serial order, so the result does not depend on the unrolling.
template <typename T>
__device__ inline void bn_sum_rows(const T* __restrict__ partial, int nrows, int C, int ch, int row, double& s, double& ss) {
    const size_t pitch = (size_t)8 * C * 2;
    const T* p = partial + ((size_t)row * C + ch) * 2;
    int k = row;
    for (; k + 24 < nrows; k += 32, p += 4 * pitch) {
        const T a0 = p[0], a1 = p[1], b0 = p[pitch], b1 = p[pitch + 1];
        const T c0 = p[2 * pitch], c1 = p[2 * pitch + 1], d0 = p[3 * pitch], d1 = p[3 * pitch + 1];
        s += (double)a0; ss += (double)a1;
        s += (double)b0; ss += (double)b1;
        s += (double)c0; ss += (double)c1;
        s += (double)d0; ss += (double)d1;
    }
    for (; k < nrows; k += 8, p += pitch) {
        s += (double)p[0];
        ss += (double)p[1];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const T* __restrict__ partial, int nblk, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          long long* __restrict__ nbt, float* __restrict__ mean_out,
                                                          float* __restrict__ invstd_out, float* __restrict__ scale,
                                                          float* __restrict__ shift, float momentum, float eps,
                                                          const unsigned* __restrict__ amax_in, unsigned* __restrict__ est_out) {
    __shared__ double red[8][32][2];
    const int tid = threadIdx.x, c = tid & 31, row = tid >> 5;
    const int ch = blockIdx.x * 32 + c;
    double s = 0.0, ss = 0.0;
    if (ch < C) bn_sum_rows<T>(partial, nblk, C, ch, row, s, ss);
    red[row][c][0] = s;
    red[row][c][1] = ss;
    __syncthreads();
    if (row == 0 && ch < C) {
        for (int k = 1; k < 8; ++k) {
            s += red[k][c][0];
            ss += red[k][c][1];
        }
        const double mean = s / count;
        double var = ss / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma[ch], b = beta[ch];
        const float sc = g * invstd;
        mean_out[ch] = (float)mean;
        invstd_out[ch] = invstd;
        const float shv = b - (float)mean * sc;
        scale[ch] = sc;
        shift[ch] = shv;
        // f16x3 hand-off: upper bound of |relu(raw * scale_c + shift_c)| for the convolution that applies this BatchNorm while
        // staging, from A = absmax(raw) (complete: the producing convolution ran before this launch): the maximum over
        // channels of |scale_c| A + |shift_c|, one atomicMax per channel lane that can raise the slot
        if (est_out) {
            const float e = fabsf(sc) * __uint_as_float(*amax_in) + fabsf(shv);
            const unsigned bits = __float_as_uint(e);
            if (bits > __hip_atomic_load(est_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(est_out, bits);
        }
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mean;
            running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
        }
    }
    if (nbt && blockIdx.x == 0 && tid == 0) *nbt += 1;
}

// (two stages from 256 partial rows on: the single-stage finalize walks nblk / 8 rows serially per channel block - measured
//  42-49 us at 1024 rows against 9 + 13 us for the fold + the 256-row finalize)
extern "C" size_t spk_bn_finalize_workspace(int nblk, int C) {
    return nblk > BN_STAGE_ROWS ? (size_t)BN_STAGE_ROWS * C * 2 * sizeof(double) : 0;
}

extern "C" int spk_bn_finalize(const float* partial, int nblk, int C, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, long long* num_batches_tracked, float* mean,
                               float* invstd, float* scale, float* shift, float momentum, float eps, double* ws,
                               const unsigned* amax_in, unsigned* est_out, void* stream) {
    SPK_REQUIRE(partial && gamma && beta && mean && invstd && scale && shift, "spk_bn_finalize: null pointer");
    SPK_REQUIRE(nblk > 0 && C > 0 && count > 0, "spk_bn_finalize: bad sizes");
    SPK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "spk_bn_finalize: running stats must come in pairs");
    SPK_REQUIRE(!est_out || amax_in, "spk_bn_finalize: est_out needs amax_in");
    hipStream_t st = (hipStream_t)stream;
    if (spk_bn_finalize_workspace(nblk, C)) {
        SPK_REQUIRE(ws, "spk_bn_finalize: %d partial rows need the fp64 workspace", nblk);
        hipLaunchKernelGGL(bn_fold_partials_kernel, dim3(spk_ceil_div(C, 32), BN_STAGE_ROWS), dim3(256), 0, st, partial, ws, nblk, C);
        SPK_LAUNCH_CHECK("spk_bn_finalize(fold)");
        hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(spk_ceil_div(C, 32)), dim3(256), 0, st, ws, BN_STAGE_ROWS, C, count,
                           gamma, beta, running_mean, running_var, num_batches_tracked, mean, invstd, scale, shift, momentum, eps,
                           amax_in, est_out);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3(spk_ceil_div(C, 32)), dim3(256), 0, st, partial, nblk, C, count, gamma,
                           beta, running_mean, running_var, num_batches_tracked, mean, invstd, scale, shift, momentum, eps, amax_in,
                           est_out);
    }
    SPK_LAUNCH_CHECK("spk_bn_finalize");
    return 0;
}

// ---- eval-mode coefficients from running statistics ------------------------------------------------
__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float* scale, float* shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

extern "C" int spk_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float* scale, float* shift, int C, float eps, void* stream) {
    SPK_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, "spk_bn_eval_coeffs: null pointer");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(spk_ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, scale, shift, C, eps);
    SPK_LAUNCH_CHECK("spk_bn_eval_coeffs");
    return 0;
}

// ---- apply: out = [relu]( raw*scale + shift [+ res | + res*rscale + rshift] ) -----------------------
// 16-byte groups per thread and input stream in the two apply kernels, and their grids (CAP 0: exactly enough blocks to cover the
// tensor once, no grid-stride loop; > 0: that many persistent blocks at most) - each pair measured inside the training step
// (profiles/r03_ab_runs.log).  tools/probe/stream_probe.hip: three 786 MB tensors stream at 5.8-6.05 TB/s from a covering grid and
// at 4.5-5.0 TB/s from 2048 / 8192 persistent blocks - but a covering grid repeats the per-thread work (per-channel vectors, the
// absmax commit) for every group, so it pays only with several groups per thread.  bn_apply: four groups, covering grid.
// bn_bwd_apply (seven per-channel vectors, mask words, the f16 pair conversion): one group per thread from 2048 persistent blocks =
// exactly the 8 blocks a CU holds, no block is ever launched behind another one (3.6 ms over its 30 launches; 8192 blocks 4.3,
// 32768 blocks 5.7, covering grids 4.6-5.8).
#ifndef BN_APPLY_U
#define BN_APPLY_U 4
#endif
#ifndef BN_APPLY_CAP
#define BN_APPLY_CAP 0
#endif
#ifndef BN_BWD_APPLY_U
#define BN_BWD_APPLY_U 1
#endif
#ifndef BN_BWD_APPLY_CAP
#define BN_BWD_APPLY_CAP 2048
#endif
static int stream_grid(long long nquads, int per_thread = 1, int cap = 8192) {
    long long nb = (nquads + 256LL * per_thread - 1) / (256LL * per_thread);
    if (cap > 0 && nb > cap) nb = cap;
    if (nb > 2147483647LL) nb = 2147483647LL;
    return (int)(nb < 1 ? 1 : nb);
}
// BN_NT: non-temporal loads / stores of the streamed tensors (the probe: 6.48 against 6.05 TB/s in isolation; in the step
// bn_apply 3.31 -> 3.04 ms, bn_bwd_apply 4.64 -> 4.25 ms over their launches)
#ifndef BN_NT
#define BN_NT 1
#endif
static __device__ __forceinline__ f32x4 ld_stream(const float* p) {
#if BN_NT
    return __builtin_nontemporal_load((const f32x4*)p);
#else
    return *(const f32x4*)p;
#endif
}
static __device__ __forceinline__ void st_stream(float* p, f32x4 v) {
#if BN_NT
    __builtin_nontemporal_store(v, (f32x4*)p);
#else
    *(f32x4*)p = v;
#endif
}
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ res,
                                                       const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                       float* __restrict__ out, unsigned* __restrict__ mask_out, long long nquads,
                                                       int C, int relu, unsigned* __restrict__ amax_out) {
    const int cmask = C - 1;
    float mx = 0.f;
    constexpr int U = BN_APPLY_U;
    const long long stride = (long long)gridDim.x * 256;
    // C <= 1024 (a power of two): a block covers 1024 floats and the stride is whole blocks - a thread stays on the channels of
    // its first group, and the per-channel vectors are loaded once
    const bool fixed_c = C <= 1024;
    const int c0 = (threadIdx.x * 4) & cmask;
    f32x4 psc = *(const f32x4*)(scale + c0), psh = *(const f32x4*)(shift + c0), prs = psc, prh = psh;
    if (rscale) {
        prs = *(const f32x4*)(rscale + c0);
        prh = *(const f32x4*)(rshift + c0);
    }
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < nquads; i0 += stride * U) {
        // U independent groups per thread: all their loads are issued before the first use (a group past the end re-reads
        // group i0 and is dropped) - more bytes in flight per CU on these HBM-bound streams
        f32x4 rv[U], sv[U];
        long long idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + stride * u;
            idx[u] = i < nquads ? i : i0;
            rv[u] = ld_stream(raw + idx[u] * 4);
            if (res) sv[u] = ld_stream(res + idx[u] * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + stride * u;
            if (i >= nquads) continue;
            if (!fixed_c) {
                const int c = (int)((i * 4) & cmask);
                psc = *(const f32x4*)(scale + c);
                psh = *(const f32x4*)(shift + c);
                if (rscale) {
                    prs = *(const f32x4*)(rscale + c);
                    prh = *(const f32x4*)(rshift + c);
                }
            }
            f32x4 v = rv[u] * psc + psh;
            if (res) {
                f32x4 r = sv[u];
                if (rscale) r = r * prs + prh;
                v += r;
            }
            if (relu) {
                v[0] = fmaxf(v[0], 0.f);
                v[1] = fmaxf(v[1], 0.f);
                v[2] = fmaxf(v[2], 0.f);
                v[3] = fmaxf(v[3], 0.f);
            }
            st_stream(out + i * 4, v);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            if (mask_out) {
                // sign mask of the output, one bit per value, 32 channels per word: the backward pass reads these bits instead
                // of the whole activated tensor.  Eight consecutive lanes hold the 32 channels of one word (C % 32 == 0, the
                // grid stride is a multiple of the wave size: the eight lanes are always active together).
                unsigned bits = (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
                bits <<= 4 * (threadIdx.x & 7);
                bits |= __shfl_xor(bits, 1, 64);
                bits |= __shfl_xor(bits, 2, 64);
                bits |= __shfl_xor(bits, 4, 64);
                if ((threadIdx.x & 7) == 0) mask_out[i >> 3] = bits;
            }
        }
    }
    if (amax_out) spk_wave_amax_commit(mx, amax_out);     // absmax(out): the operand scale of its f16x3 consumers
}

extern "C" int spk_bn_apply(const float* raw, const float* scale, const float* shift, const float* res,
                            const float* res_scale, const float* res_shift, float* out, unsigned* mask_out, long long N, int C,
                            int relu, unsigned* amax_out, void* stream) {
    SPK_REQUIRE(raw && scale && shift && out, "spk_bn_apply: null pointer");
    SPK_REQUIRE(N > 0 && bn_c_ok(C), "spk_bn_apply: N=%lld C=%d", N, C);
    SPK_REQUIRE((res_scale == nullptr) == (res_shift == nullptr), "spk_bn_apply: residual affine must come in pairs");
    SPK_REQUIRE(!res_scale || res, "spk_bn_apply: residual affine without residual");
    SPK_REQUIRE(!mask_out || C % 32 == 0, "spk_bn_apply: the sign mask needs C %% 32 == 0 (C=%d)", C);
    const long long nquads = N * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(nquads, BN_APPLY_U, BN_APPLY_CAP)), dim3(256), 0, (hipStream_t)stream, raw, scale, shift, res,
                       res_scale, res_shift, out, mask_out, nquads, C, relu, amax_out);
    SPK_LAUNCH_CHECK("spk_bn_apply");
    return 0;
}

// ---- backward reduce: partial[blk][c] = (sum dz, sum dz*xhat), dz = dy * mask ---------------------------
__device__ inline f32x4 bn_mask(f32x4 dy, int mode, const float* act, const float* raw_p, f32x4 rawv, f32x4 sc, f32x4 sh,
                                long long off, int C = 0) {
    if (mode == MASK_BITS) {
        // sign bits written by spk_bn_apply (mask_out): one word per pixel and 32 channels; off = pixel * C + c
        const int lg = 31 - __builtin_clz((unsigned)C);          // C is a power of two
        const long long pix = off >> lg;
        const int c = (int)(off & (C - 1));
        const unsigned bits = ((const unsigned*)act)[pix * (C >> 5) + (c >> 5)] >> (c & 31);
#pragma unroll
        for (int k = 0; k < 4; ++k) dy[k] = ((bits >> k) & 1u) ? dy[k] : 0.f;
    } else if (mode == MASK_ACT) {
        const f32x4 a = *(const f32x4*)(act + off);
#pragma unroll
        for (int k = 0; k < 4; ++k) dy[k] = a[k] > 0.f ? dy[k] : 0.f;
    } else if (mode == MASK_RAW) {
        const f32x4 z = rawv * sc + sh;
#pragma unroll
        for (int k = 0; k < 4; ++k) dy[k] = z[k] > 0.f ? dy[k] : 0.f;
    }
    (void)raw_p;
    return dy;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ raw,
                                                            const float* __restrict__ act, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ partial,
                                                            long long N, int C, int rows_per_block, int mode,
                                                            unsigned* __restrict__ chan_amax) {
    __shared__ float red[256][8];
    const int tid = threadIdx.x;
    const int qpr = C >> 2;
    const int quad = tid % qpr, prow = tid / qpr, rstep = 256 / qpr;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    const f32x4 mu = *(const f32x4*)(mean + quad * 4), is = *(const f32x4*)(invstd + quad * 4);
    const f32x4 sc = *(const f32x4*)(scale + quad * 4), sh = *(const f32x4*)(shift + quad * 4);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f}, mx = {0.f, 0.f, 0.f, 0.f};
    for (long long r = r0 + prow; r < r1; r += rstep) {
        const long long off = r * C + quad * 4;
        const f32x4 rv = ld_stream(raw + off);
        f32x4 d = ld_stream(dy + off);
        d = bn_mask(d, mode, act, raw, rv, sc, sh, off, C);
        const f32x4 xh = (rv - mu) * is;
        s += d;
        ss += d * xh;
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], spk_finite_abs(d[k]));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[tid][k] = s[k];
        red[tid][4 + k] = ss[k];
    }
    __syncthreads();
    if (tid < qpr) {
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = 0.f;
        for (int p = 0; p < rstep; ++p)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += red[p * qpr + tid][k];
        float* dst = partial + ((size_t)blockIdx.x * C + tid * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[k * 2 + 0] = a[k];
            dst[k * 2 + 1] = a[4 + k];
        }
    }
    if (chan_amax) {
        // per-CHANNEL absmax of dz (float bits, atomicMax): the BatchNorm-backward bound then pairs every channel's own |dz| with
        // its own k1 (bn_bwd_finalize chan_amax).  One tensor-wide absmax couples the channels: a gradient that is huge in a channel
        // whose gamma is tiny (the pooling layer's sqrt'(mean) at a tiny mean, scripts/model.py:453) is multiplied by the LARGEST
        // |k1| of the layer and the bound overshoots the values by that ratio - everything then sits low in its fp16 window.
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) red[tid][k] = mx[k];
        __syncthreads();
        if (tid < qpr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float m = 0.f;
                for (int p = 0; p < rstep; ++p) m = fmaxf(m, red[p * qpr + tid][k]);
                const unsigned bits = __float_as_uint(m);
                unsigned* slot = chan_amax + tid * 4 + k;
                if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
            }
        }
    }
}

extern "C" int spk_bn_bwd_reduce(const float* dy, const float* raw, const float* act, const float* mean, const float* invstd,
                                 const float* scale, const float* shift, float* partial, long long N, int C, int mask_mode,
                                 unsigned* chan_amax, void* stream) {
    SPK_REQUIRE(dy && raw && mean && invstd && scale && shift && partial, "spk_bn_bwd_reduce: null pointer");
    SPK_REQUIRE(N > 0 && bn_c_ok(C), "spk_bn_bwd_reduce: N=%lld C=%d", N, C);
    SPK_REQUIRE(mask_mode >= 0 && mask_mode <= 3, "spk_bn_bwd_reduce: mask_mode=%d", mask_mode);
    SPK_REQUIRE((mask_mode != MASK_ACT && mask_mode != MASK_BITS) || act, "spk_bn_bwd_reduce: MASK_ACT / MASK_BITS need the activated tensor / its sign bits");
    const int nb = spk_bn_stats_blocks(N, C);
    const int rpb = (int)((N + nb - 1) / nb);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, raw, act, mean, invstd, scale,
                       shift, partial, N, C, rpb, mask_mode, chan_amax);
    SPK_LAUNCH_CHECK("spk_bn_bwd_reduce");
    return 0;
}

// Upper bound of |k1 (dz - m1 - xhat m2)| over one channel, for |dz| <= A and |raw| <= R: |xhat| <= (R + |mean|) invstd.
// The value is evaluated in fp32 as k1 * (dz - m1 - ((raw - mean) * invstd) * m2): five roundings of relative size 2^-24,
// covered (with the roundings of this expression) by the factor 1 + 2^-16.
__device__ inline float bnbwd_bound(float k1, float m1, float m2, float mean, float invstd, float A, float R) {
    const float xh = (R + fabsf(mean)) * fabsf(invstd);
    return fabsf(k1) * (A + fabsf(m1) + xh * fabsf(m2)) * (1.f + 1.52587890625e-05f);
}

// ---- backward finalize: dgamma, dbeta and the apply coefficients coef[3][C] = (gamma*invstd, dbeta/n, dgamma/n)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const T* __restrict__ partial, int nblk, int C, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef, int accumulate,
                                                              const unsigned* __restrict__ amax_in, const unsigned* __restrict__ raw_amax,
                                                              const float* __restrict__ mean, unsigned* __restrict__ est_out,
                                                              const unsigned* __restrict__ chan_amax) {
    __shared__ double red[8][32][2];
    const int tid = threadIdx.x, c = tid & 31, row = tid >> 5;
    const int ch = blockIdx.x * 32 + c;
    double s = 0.0, ss = 0.0;
    if (ch < C) bn_sum_rows<T>(partial, nblk, C, ch, row, s, ss);
    red[row][c][0] = s;
    red[row][c][1] = ss;
    __syncthreads();
    if (row == 0 && ch < C) {
        for (int k = 1; k < 8; ++k) {
            s += red[k][c][0];
            ss += red[k][c][1];
        }
        dbeta[ch] = accumulate ? dbeta[ch] + (float)s : (float)s;
        dgamma[ch] = accumulate ? dgamma[ch] + (float)ss : (float)ss;
        const float k1 = gamma[ch] * invstd[ch], m1 = (float)(s / count), m2 = (float)(ss / count);
        coef[ch] = k1;
        coef[C + ch] = m1;
        coef[2 * C + ch] = m2;
        // f16x3 hand-off: RIGOROUS upper bound of the values k1 (dz - m1 - xhat m2) the BatchNorm backward produces (staged by the
        // fused data gradient, or written as f16 pairs by spk_bn_bwd_apply), maximum over channels by atomicMax.
        // |dz| <= A = absmax of the incoming gradient, |xhat| = |raw - mean| invstd <= (R + |mean|) invstd with R = absmax(raw),
        // both slots complete before this launch; bnbwd_bound() adds the rounding slack of the fp32 evaluation.
        if (est_out) {
            // chan_amax (when the reduction pass produced it): A is this channel's own absmax of dz instead of the tensor's
            const float A = chan_amax ? __uint_as_float(chan_amax[ch]) : __uint_as_float(*amax_in);
            const float e = bnbwd_bound(k1, m1, m2, mean[ch], invstd[ch], A, __uint_as_float(*raw_amax));
            const unsigned bits = __float_as_uint(e);
            if (bits > __hip_atomic_load(est_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(est_out, bits);
        }
    }
}

extern "C" int spk_bn_bwd_finalize(const float* partial, int nblk, int C, double count, const float* gamma,
                                   const float* invstd, float* dgamma, float* dbeta, float* coef, int accumulate,
                                   double* ws, const unsigned* amax_in, const unsigned* raw_amax, const float* mean,
                                   unsigned* est_out, const unsigned* chan_amax, void* stream) {
    SPK_REQUIRE(partial && gamma && invstd && dgamma && dbeta && coef, "spk_bn_bwd_finalize: null pointer");
    SPK_REQUIRE(nblk > 0 && C > 0 && count > 0, "spk_bn_bwd_finalize: bad sizes");
    SPK_REQUIRE(!est_out || ((amax_in || chan_amax) && raw_amax && mean),
                "spk_bn_bwd_finalize: est_out needs amax_in (or chan_amax), raw_amax and mean");
    hipStream_t st = (hipStream_t)stream;
    if (spk_bn_finalize_workspace(nblk, C)) {
        SPK_REQUIRE(ws, "spk_bn_bwd_finalize: %d partial rows need the fp64 workspace", nblk);
        hipLaunchKernelGGL(bn_fold_partials_kernel, dim3(spk_ceil_div(C, 32), BN_STAGE_ROWS), dim3(256), 0, st, partial, ws, nblk, C);
        SPK_LAUNCH_CHECK("spk_bn_bwd_finalize(fold)");
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3(spk_ceil_div(C, 32)), dim3(256), 0, st, ws, BN_STAGE_ROWS, C, count,
                           gamma, invstd, dgamma, dbeta, coef, accumulate, amax_in, raw_amax, mean, est_out, chan_amax);
    } else {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<float>, dim3(spk_ceil_div(C, 32)), dim3(256), 0, st, partial, nblk, C, count, gamma,
                           invstd, dgamma, dbeta, coef, accumulate, amax_in, raw_amax, mean, est_out, chan_amax);
    }
    SPK_LAUNCH_CHECK("spk_bn_bwd_finalize");
    return 0;
}

// ---- backward apply: draw = k1*(dz - m1 - xhat*m2); optionally also stores dz (the residual-path gradient)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ raw,
                                                           const float* __restrict__ act, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ coef,
                                                           float* __restrict__ draw, float* __restrict__ dz_out,
                                                           unsigned nquads, int C, int mode, unsigned* __restrict__ amax_out,
                                                           const unsigned* __restrict__ pair_scale) {
    const int cmask = C - 1;
    float mx = 0.f;
    const float sig = pair_scale ? spk_sigma_from_amax_bits(*pair_scale) : 1.f;
    constexpr int U = BN_BWD_APPLY_U;
    // 32-bit group indices (the host checks nquads < 2^31): the pixel of a group is a shift, its mask word one multiply-add
    const unsigned stride = gridDim.x * 256u;
    const int lgq = 29 - __builtin_clz((unsigned)C);         // log2(C / 4): 16-byte groups per pixel (C is a power of two)
    const unsigned cw = (unsigned)C >> 5;
    // (C <= 1024: a thread stays on the channels of its first group - see bn_apply_kernel - the seven per-channel vectors once)
    const bool fixed_c = C <= 1024;
    const int c0 = (threadIdx.x * 4) & cmask;
    f32x4 pmu = *(const f32x4*)(mean + c0), pis = *(const f32x4*)(invstd + c0), pk1 = *(const f32x4*)(coef + c0);
    f32x4 pm1 = *(const f32x4*)(coef + C + c0), pm2 = *(const f32x4*)(coef + 2 * C + c0), psc = pmu, psh = pmu;
    if (mode == MASK_RAW) {
        psc = *(const f32x4*)(scale + c0);
        psh = *(const f32x4*)(shift + c0);
    }
    for (unsigned i0 = blockIdx.x * 256u + threadIdx.x; i0 < nquads; i0 += stride * U) {
        f32x4 rvv[U], dv[U], av[U];
        unsigned mw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {          // all loads of the U groups first (a group past the end re-reads group i0)
            const unsigned i = i0 + stride * u;
            const unsigned ii = i < nquads && i >= i0 ? i : i0;
            const size_t off = (size_t)ii * 4;
            rvv[u] = ld_stream(raw + off);
            dv[u] = ld_stream(dy + off);
            if (mode == MASK_BITS) mw[u] = ((const unsigned*)act)[(size_t)(ii >> lgq) * cw + (((ii * 4u) & (unsigned)cmask) >> 5)];
            else if (mode == MASK_ACT) av[u] = *(const f32x4*)(act + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + stride * u;
            if (i >= nquads || i < i0) continue;
            const int c = fixed_c ? c0 : (int)((i * 4u) & (unsigned)cmask);
            if (!fixed_c) {
                pmu = *(const f32x4*)(mean + c);
                pis = *(const f32x4*)(invstd + c);
                pk1 = *(const f32x4*)(coef + c);
                pm1 = *(const f32x4*)(coef + C + c);
                pm2 = *(const f32x4*)(coef + 2 * C + c);
                if (mode == MASK_RAW) {
                    psc = *(const f32x4*)(scale + c);
                    psh = *(const f32x4*)(shift + c);
                }
            }
            const size_t off = (size_t)i * 4;
            const f32x4 rv = rvv[u];
            f32x4 d = dv[u];
            if (mode == MASK_BITS) {
                const unsigned bits = mw[u] >> (c & 31);
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = ((bits >> k) & 1u) ? d[k] : 0.f;
            } else if (mode == MASK_ACT) {
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = av[u][k] > 0.f ? d[k] : 0.f;
            } else if (mode == MASK_RAW) {
                const f32x4 z = rv * psc + psh;
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = z[k] > 0.f ? d[k] : 0.f;
            }
            const f32x4 xh = (rv - pmu) * pis;
            const f32x4 o = pk1 * (d - pm1 - xh * pm2);
            if (dz_out) st_stream(dz_out + off, d);
            if (pair_scale) {      // f16 pair tensor: converted here once, staged by plain copy in the data and weight gradients
                uint2 t0, t1;
                split2h(o, sig, t0, t1);
                st_stream(draw + off, spk_pair_pack(t0, t1));
            } else
                st_stream(draw + off, o);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        }
    }
    if (amax_out) spk_wave_amax_commit(mx, amax_out);     // absmax(draw): the operand scale of its f16x3 consumers
}

extern "C" int spk_bn_bwd_apply(const float* dy, const float* raw, const float* act, const float* mean, const float* invstd,
                                const float* scale, const float* shift, const float* coef, float* draw, float* dz_out,
                                long long N, int C, int mask_mode, unsigned* amax_out, const unsigned* pair_scale, void* stream) {
    SPK_REQUIRE(dy && raw && mean && invstd && scale && shift && coef && draw, "spk_bn_bwd_apply: null pointer");
    SPK_REQUIRE(N > 0 && bn_c_ok(C), "spk_bn_bwd_apply: N=%lld C=%d", N, C);
    SPK_REQUIRE(mask_mode >= 0 && mask_mode <= 3, "spk_bn_bwd_apply: mask_mode=%d", mask_mode);
    SPK_REQUIRE((mask_mode != MASK_ACT && mask_mode != MASK_BITS) || act, "spk_bn_bwd_apply: MASK_ACT / MASK_BITS need the activated tensor / its sign bits");
    const long long nquads = N * C / 4;
    SPK_REQUIRE(nquads < 2147483647LL / 8, "spk_bn_bwd_apply: %lld values exceed the 32-bit group index of the kernel", N * C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(nquads, BN_BWD_APPLY_U, BN_BWD_APPLY_CAP)), dim3(256), 0, (hipStream_t)stream, dy, raw, act, mean,
                       invstd, scale, shift, coef, draw, dz_out, (unsigned)nquads, C, mask_mode, amax_out, pair_scale);
    SPK_LAUNCH_CHECK("spk_bn_bwd_apply");
    return 0;
}

// ---- operand-scale hand-offs of the f16x3 mode ---------------------------------------------------------------------
// *slot = max(*slot, float bits of max|x|)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ slot) {
    float mx = 0.f;
    const long long nq = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(x + i * 4);
        mx = fmaxf(fmaxf(mx, fmaxf(spk_finite_abs(v[0]), spk_finite_abs(v[1]))), fmaxf(spk_finite_abs(v[2]), spk_finite_abs(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) mx = fmaxf(mx, spk_finite_abs(x[nq * 4 + threadIdx.x]));
    spk_wave_amax_commit(mx, slot);
}

extern "C" int spk_absmax(const float* x, unsigned* slot, long long n, void* stream) {
    SPK_REQUIRE(x && slot && n > 0, "spk_absmax: bad arguments");
    SPK_REQUIRE((((size_t)x) & 15) == 0, "spk_absmax: x must be 16-byte aligned");
    hipLaunchKernelGGL(absmax_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, n, slot);
    SPK_LAUNCH_CHECK("spk_absmax");
    return 0;
}

// Rigorous upper bound of max |k1 (dz - m1 - xhat m2)| - the values of a BatchNorm backward - from the coefficient rows
// [k1, m1, m2][C], the BatchNorm's mean / invstd rows, A = absmax of the incoming gradient and R = absmax of the raw tensor
// (bnbwd_bound; the same value spk_bn_bwd_finalize(est_out=) gives without this launch).  *est = float bits (plain store).
__global__ __launch_bounds__(256) void bnbwd_estimate_kernel(const float* __restrict__ coef, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int C,
                                                             const unsigned* __restrict__ amax_in, const unsigned* __restrict__ raw_amax,
                                                             unsigned* __restrict__ est) {
    __shared__ float red[4];
    const float A = __uint_as_float(*amax_in), R = __uint_as_float(*raw_amax);
    float mx = 0.f;
    for (int c = threadIdx.x; c < C; c += 256)
        mx = fmaxf(mx, bnbwd_bound(coef[c], coef[C + c], coef[2 * C + c], mean[c], invstd[c], A, R));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) *est = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// Upper bound of |max(raw * scale_c + shift_c, 0)| - what a convolution with the fused input BatchNorm+ReLU stages - from
// A = absmax(raw): max_c |scale_c| * A + max_c |shift_c|.
__global__ __launch_bounds__(256) void affine_estimate_kernel(const float* __restrict__ scale, const float* __restrict__ shift, int C,
                                                              const unsigned* __restrict__ amax_in, unsigned* __restrict__ est) {
    __shared__ float red[2][4];
    float ms = 0.f, mh = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        ms = fmaxf(ms, fabsf(scale[c]));
        mh = fmaxf(mh, fabsf(shift[c]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ms = fmaxf(ms, __shfl_xor(ms, off, 64));
        mh = fmaxf(mh, __shfl_xor(mh, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = ms;
        red[1][threadIdx.x >> 6] = mh;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ms = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        mh = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        *est = __float_as_uint(ms * __uint_as_float(*amax_in) + mh);
    }
}

extern "C" int spk_affine_estimate(const float* scale, const float* shift, int C, const unsigned* amax_in, unsigned* est,
                                   void* stream) {
    SPK_REQUIRE(scale && shift && amax_in && est && C > 0, "spk_affine_estimate: bad arguments");
    hipLaunchKernelGGL(affine_estimate_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scale, shift, C, amax_in, est);
    SPK_LAUNCH_CHECK("spk_affine_estimate");
    return 0;
}

extern "C" int spk_bnbwd_estimate(const float* coef, const float* mean, const float* invstd, int C, const unsigned* amax_in,
                                  const unsigned* raw_amax, unsigned* est, void* stream) {
    SPK_REQUIRE(coef && mean && invstd && amax_in && raw_amax && est && C > 0, "spk_bnbwd_estimate: bad arguments");
    hipLaunchKernelGGL(bnbwd_estimate_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, coef, mean, invstd, C, amax_in, raw_amax, est);
    SPK_LAUNCH_CHECK("spk_bnbwd_estimate");
    return 0;
}

// ---- f16x3 diagnostics: how a tensor sits in the two-term fp16 window of its scale slot ------------------------------------
// counts[0] += values looked at, [1] += values that SATURATE (|v sigma| > 65504: must be 0, every slot is an absmax or a rigorous
// bound), [2] += values whose low term is an fp16 subnormal (0 < |lo| < 2^-14: kept by the matrix instruction with an absolute
// resolution of 2^-24 - the value carries between 11 and 22 significand bits, absolute error <= bound * 2^-39), [3] += values
// whose HIGH term is subnormal (|v sigma| < 2^-14: fewer than 11 bits, same absolute error bound).  v = x, or max(x*scale[c]+shift[c], 0) when scale is
// given (what a convolution with the fused input BatchNorm+ReLU stages); pairs != 0: x is an f16 pair tensor (the window
// is then read back from the stored terms; a stored high term of +-65504 counts as saturated).  Debug / test export: never
// on the training path.
__global__ __launch_bounds__(256) void f16_window_count_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, long long nquads, int C,
                                                               const unsigned* __restrict__ slot, int pairs,
                                                               unsigned long long* __restrict__ counts) {
    const float sig = spk_sigma_from_amax_bits(*slot);
    const int cmask = C - 1;
    unsigned n = 0, sat = 0, lo_lost = 0, hi_sub = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (long long)gridDim.x * 256) {
        f32x4 v = *(const f32x4*)(x + i * 4);
        float hi[4], lo[4];
        if (pairs) {
            const uint4 u = __builtin_bit_cast(uint4, v);
            const f16x4 a = __builtin_bit_cast(f16x4, (uint2){u.x, u.y}), b = __builtin_bit_cast(f16x4, (uint2){u.z, u.w});
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hi[k] = (float)a[k];
                lo[k] = (float)b[k];
                sat += fabsf(hi[k]) >= 65504.f;
            }
        } else {
            if (scale) {
                const int c = (int)((i * 4) & cmask);
                v = v * *(const f32x4*)(scale + c) + *(const f32x4*)(shift + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = v[k] * sig;
                sat += fabsf(u) > 65504.f;
                const float uc = __builtin_amdgcn_fmed3f(u, -65504.f, 65504.f);
                const _Float16 a = (_Float16)uc;
                hi[k] = (float)a;
                lo[k] = (float)(_Float16)(uc - (float)a);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            n += 1;
            lo_lost += lo[k] != 0.f && fabsf(lo[k]) < 6.103515625e-05f;
            hi_sub += hi[k] != 0.f && fabsf(hi[k]) < 6.103515625e-05f;
        }
    }
    unsigned vals[4] = {n, sat, lo_lost, hi_sub};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned v = vals[j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(counts + j, (unsigned long long)v);
    }
}

extern "C" int spk_f16_window_count(const float* x, const float* scale, const float* shift, long long n, int C,
                                    const unsigned* slot, int pairs, unsigned long long* counts, void* stream) {
    SPK_REQUIRE(x && slot && counts && n > 0 && n % 4 == 0, "spk_f16_window_count: bad arguments (n must be a multiple of 4)");
    SPK_REQUIRE((scale == nullptr) == (shift == nullptr), "spk_f16_window_count: scale / shift come in pairs");
    SPK_REQUIRE(!scale || (bn_c_ok(C) && !pairs), "spk_f16_window_count: the affine form needs a power-of-two C and an fp32 tensor");
    hipLaunchKernelGGL(f16_window_count_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, n / 4,
                       scale ? C : 4, slot, pairs, counts);
    SPK_LAUNCH_CHECK("spk_f16_window_count");
    return 0;
}
