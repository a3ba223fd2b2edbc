// 3x3 weight gradient, f16x3 operands, with the four waves of a block laid out 2 input-channel groups x 2 output-channel
// groups (conv_wgrad_split_kernel lays them out 2 pixel halves x 2 output-channel groups).
//
// The staging of the weight gradient is VALU-bound and does not overlap the matrix instructions, so what
// counts is staged elements per MFMA.  A block here owns 64 input x 64 output channels: it stages 64-channel X pixels
// ([pixel][2 groups][terms][32 ch]) and the 64-channel dY tile once, and every wave runs ALL k-steps of the region on its
// own 32 x 32 x 9-tap tile - 108 MFMAs per wave and region against 11 staging items per thread, where the pixel-split layout has
// 54 against 9 (the dY tile was converted Cin/32 times, now Cin/64; the X tile Cout/64 times as before).  No cross-wave fold
// at the end: a wave's accumulators are final.  Same MFMA order along the pixels of a region as the pixel-split layout
// would use with WK = 1, but NOT the same summation order as WK = 2 (one accumulator instead of two partial ones), so
// results agree with conv_wgrad_split_kernel to fp32 rounding, not bit for bit.
#include "conv_wgrad.h"
#ifndef WGRAD_XCD_BAND
#define WGRAD_XCD_BAND 1
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ s16x8 tr_read8m(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

#define WM_NX 7      // X halo float4 per thread: halo_pix <= 16 * WM_NX pixels of 64 channels
#define WM_ND 4      // dY float4 per thread: npix <= 16 * WM_ND

// VAR: compile-time variant of the two run-time switches of the staging pass (so that its per-item branches fold away): bit 0 = fused
// input BatchNorm + ReLU on X, bit 1 = dY is an f16 pair tensor; VAR < 0: both read from the argument block.
// SH (stride 1, TW % 8 == 0; experimental): the eight pixels a lane holds of a k-step are consecutive in one tile row, so the fragments
// of the three taps of a filter row are windows [0,8) [1,9) [2,10) of TEN consecutive halo pixels: pixels [0,8) and [2,10) read as two
// register quads per term (taps 0 and 2), the middle tap built from both in registers (v_alignbit: 4 per term) - 28 transposed LDS reads
// per k-step instead of 40, requested one tap ahead.  Same products in the same order: bit-identical to the plain form on the same tile.
// (A first version with three reads per term and row - 22 per k-step - needed register copies for the unaligned quad of tap 2 and
// spilled 12-16 registers at the 256-register bound: the reloads sit in the per-region path and cost 25 %.)
template <int VAR, bool SH>
__global__ __launch_bounds__(256, 2) void conv_wgrad_wm_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    constexpr int NTAPS = 9, KS = 3, WN = 2;
    constexpr int PX = 2 * 192;                           // X bytes per staged pixel: [2 groups][3-term pitch][32 ch fp16]
    constexpr int PD = WN * 192 + 64;                     // dY bytes per pixel (as conv_wgrad_split_kernel)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int g, ci0, co0;                                      // block -> (region slice, channel groups), slices on one XCD
    {
        const int ncgi = a.Cin >> 6, M = ncgi * (a.Cout >> 6);
        const int bid = blockIdx.x;
        int m;
        if ((a.nsplit & 7) == 0) {
            const int k = bid >> 3;
            m = k % M;
            g = WGRAD_XCD_BAND ? (bid & 7) * (a.nsplit >> 3) + k / M : (k / M) * 8 + (bid & 7);      // conv_wgrad_split.hip: wgrad_block
        } else {
            m = bid % M;
            g = bid / M;
        }
        ci0 = (m % ncgi) * 64;
        co0 = (m / ncgi) * 64;
    }
    const int halo_pix = a.halo_h * a.halo_w;
    const int npix = a.TH * a.TW;
    const int nsteps_all = (npix + 15) >> 4;
    const int npix_pad = nsteps_all << 4;
    unsigned char* xs = ldsb;
    unsigned char* dys = ldsb + halo_pix * PX;
    const int flags = VAR >= 0 ? ((VAR & 1) ? SPK_IN_AFFINE_RELU : 0) : a.flags;
    const float sig_x = a.x_amax ? spk_sigma_from_amax_bits(*a.x_amax) : SPK_F16_ACT_SIGMA;
    const float sig_d = a.dy_amax ? spk_sigma_from_amax_bits(*a.dy_amax) : 1.f;
    const bool dy_pairs = VAR >= 0 ? (VAR & 2) != 0 : (a.flags & SPK_DY_PRESPLIT) != 0;      // dY is an f16 pair tensor: staged by plain copy

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int q16 = tid & 15;                             // this thread's float4 of the 64 channels of a pixel (X and dY alike)
    f32x4 px[WM_NX], pd[WM_ND];
    unsigned inx = 0, ind = 0;
    const unsigned x_row = (unsigned)a.IW * a.Cin * 4u, x_px = (unsigned)a.Cin * 4u;
    const unsigned d_row = (unsigned)a.OW * a.Cout * 4u, d_px = (unsigned)a.Cout * 4u;
    const unsigned x_c = (unsigned)(ci0 + q16 * 4) * 4u, d_c = (unsigned)(co0 + q16 * 4) * 4u;
    auto prefetch = [&](int region) {
        int pt = region;
        const int tx = pt % a.tiles_x;
        pt /= a.tiles_x;
        const int ty = pt % a.tiles_y;
        const int b = pt / a.tiles_y;
        const int oy0 = ty * a.TH, ox0 = tx * a.TW;
        const int iy0 = oy0 * a.S - a.pad, ix0 = ox0 * a.S - a.pad;
        const char* xb = (const char*)(a.x + (size_t)b * a.IH * a.IW * a.Cin);
        const char* db = (const char*)(a.dy + (size_t)b * a.OH * a.OW * a.Cout);
        const unsigned x_safe = (unsigned)(oy0 * a.S) * x_row + (unsigned)(ox0 * a.S) * x_px + x_c;
        const unsigned d_safe = (unsigned)oy0 * d_row + (unsigned)ox0 * d_px + d_c;
        inx = 0;
        ind = 0;
#pragma unroll
        for (int u = 0; u < WM_NX; ++u) {
            const int p = (tid >> 4) + 16 * u;
            const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
            const int hx = p - hy * a.halo_w;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW && p < halo_pix;
            if (ok) inx |= 1u << u;
            const unsigned off = ok ? (unsigned)iy * x_row + (unsigned)ix * x_px + x_c : x_safe;
            px[u] = wgrad_ld<2>(xb + off);
        }
#pragma unroll
        for (int u = 0; u < WM_ND; ++u) {
            const int p = (tid >> 4) + 16 * u;
            const int ly = (int)__umulhi((unsigned)p, a.tw_magic);
            const int lx = p - ly * a.TW;
            const int oy = oy0 + ly, ox = ox0 + lx;
            const bool ok = p < npix && oy < a.OH && ox < a.OW;
            if (ok) ind |= 1u << u;
            const unsigned off = ok ? (unsigned)oy * d_row + (unsigned)ox * d_px + d_c : d_safe;
            pd[u] = wgrad_ld<1>(db + off);
        }
    };
    auto publish = [&]() {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (flags & SPK_IN_AFFINE_RELU) {
            sc = *(const f32x4*)(a.in_scale + ci0 + q16 * 4);
            sh = *(const f32x4*)(a.in_shift + ci0 + q16 * 4);
        }
#pragma unroll
        for (int u = 0; u < WM_NX; ++u) {
            const int p = (tid >> 4) + 16 * u;
            f32x4 w = px[u];
            if (flags & SPK_IN_AFFINE_RELU) {
                w = w * sc + sh;
                w[0] = fmaxf(w[0], 0.f);
                w[1] = fmaxf(w[1], 0.f);
                w[2] = fmaxf(w[2], 0.f);
                w[3] = fmaxf(w[3], 0.f);
            }
            if (!((inx >> u) & 1)) w = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p < halo_pix) {
                uint2* dst = (uint2*)(xs + p * PX + (q16 >> 3) * 192) + (q16 & 7);
                uint2 t0, t1;
                split2h(w, sig_x, t0, t1);
                dst[0] = t0;
                dst[8] = t1;
            }
        }
#pragma unroll
        for (int u = 0; u < WM_ND; ++u) {
            const int p = (tid >> 4) + 16 * u;
            const f32x4 w = ((ind >> u) & 1) ? pd[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p < npix_pad) {
                uint2* dst = (uint2*)(dys + p * PD + (q16 >> 3) * 192) + (q16 & 7);
                uint2 t0, t1;
                spk_terms(w, sig_d, dy_pairs, t0, t1);
                dst[0] = t0;
                dst[8] = t1;
            }
        }
    };

    const int g16 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const int col_off = (g16 & 1) * 32 + p4 * 8;
    auto mma = [&](f32x16& c, const s16x8* af, const s16x8* bf) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[1]), __builtin_bit_cast(f16x8, bf[0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[0]), c, 0, 0, 0);
    };

    int region = g;
    if (region < a.nregions) prefetch(region);
    for (; region < a.nregions; region += a.nsplit) {
        __syncthreads();
        publish();
        __syncthreads();
        if (region + a.nsplit < a.nregions) prefetch(region + a.nsplit);
        if constexpr (SH) {
            // lane -> LDS addresses of k-step j: the X window of its half-step's eight pixels (filter row 0) and its dY fragment
            auto step_ptrs = [&](int j, const unsigned char*& xr, const unsigned char*& dr) {
                const int pix0 = j * 16 + 8 * h;
                const int pc0 = pix0 < npix ? pix0 : npix - 8;          // npix % 8 == 0; the padded half-step multiplies zeros of dY
                const int ly = (int)__umulhi((unsigned)pc0, a.tw_magic);
                const int lx0 = pc0 - ly * a.TW;
                xr = xs + (ly * a.halo_w + lx0 + q) * PX + wm * 192 + col_off;
                dr = dys + (pix0 + q) * PD + wn * 192 + col_off;
            };
            // pixels [0,8) and [2,10) of the window as two aligned register quads per term (taps 0 and 2 need no copies); the middle
            // tap [1,9) is built from both with four v_alignbit / v_perm per term, in place of the first quad
            auto load_q = [&](s16x8* af, const unsigned char* p) {
#pragma unroll
                for (int s = 0; s < 2; ++s) af[s] = tr_read8m(p + s * 64, p + s * 64 + 4 * PX);
            };
            const int row_b = a.halo_w * PX;
            const unsigned char *xr, *dr;
            step_ptrs(0, xr, dr);
            s16x8 bf[2], a0[2], a2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) bf[s] = tr_read8m(dr + s * 64, dr + 4 * PD + s * 64);
            load_q(a0, xr);
            load_q(a2, xr + 2 * PX);
            for (int j = 0; j < nsteps_all; ++j) {
#pragma unroll
                for (int ty = 0; ty < KS; ++ty) {
                    const unsigned char *pn, *dn;
                    if (ty + 1 < KS) pn = xr + (ty + 1) * row_b;
                    else step_ptrs(j + 1 < nsteps_all ? j + 1 : j, pn, dn);      // after the last k-step: a harmless re-read
                    __builtin_amdgcn_sched_barrier(0);
                    mma(acc[3 * ty], a0, bf);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const u32x4 lo = __builtin_bit_cast(u32x4, a0[s]), hi = __builtin_bit_cast(u32x4, a2[s]);
                        a0[s] = __builtin_bit_cast(s16x8, (u32x4){__builtin_amdgcn_alignbit(lo[1], lo[0], 16), __builtin_amdgcn_alignbit(lo[2], lo[1], 16),
                                                                  __builtin_amdgcn_alignbit(lo[3], lo[2], 16), __builtin_amdgcn_alignbit(hi[3], lo[3], 16)});
                    }
                    mma(acc[3 * ty + 1], a0, bf);
                    __builtin_amdgcn_sched_barrier(0);
                    load_q(a0, pn);                                     // the next filter row's first quad, under the products of tap 2
                    __builtin_amdgcn_sched_barrier(0);
                    mma(acc[3 * ty + 2], a2, bf);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ty + 1 == KS) {
#pragma unroll
                        for (int s = 0; s < 2; ++s) bf[s] = tr_read8m(dn + s * 64, dn + 4 * PD + s * 64);
                    }
                    load_q(a2, pn + 2 * PX);                            // ... and its second quad under the products of the next tap 0
                    if (ty + 1 == KS) xr = pn;
                }
            }
        } else {
            for (int j = 0; j < nsteps_all; ++j) {
                int xa[2], da[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const int pix = j * 16 + 8 * h + 4 * blk + q;
                    const int pc = pix < npix ? pix : npix - 1;
                    const int ly = (int)__umulhi((unsigned)pc, a.tw_magic);
                    const int lx = pc - ly * a.TW;
                    xa[blk] = ((ly * a.S) * a.halo_w + lx * a.S) * PX + wm * 192 + col_off;
                    da[blk] = pix * PD + wn * 192 + col_off;
                }
                s16x8 bf[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) bf[s] = tr_read8m(dys + da[0] + s * 64, dys + da[1] + s * 64);
                s16x8 a0[2], a1[2];
                auto load_a = [&](s16x8* af, int t) {
                    const int toff = ((t / KS) * a.halo_w + (t % KS)) * PX;
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s] = tr_read8m(xs + xa[0] + toff + s * 64, xs + xa[1] + toff + s * 64);
                };
                load_a(a0, 0);
#pragma unroll
                for (int t = 0; t < NTAPS; t += 2) {
                    if (t + 1 < NTAPS) load_a(a1, t + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(acc[t], a0, bf);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 2 < NTAPS) load_a(a0, t + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < NTAPS) mma(acc[t + 1], a1, bf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    const int r = lane & 31;
    float* slab = a.partial + (size_t)g * NTAPS * a.Cin * a.Cout;
    const float inv_x = 1.f / sig_x, inv_d = 1.f / sig_d;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            slab[((size_t)t * a.Cin + ci0 + wm * 32 + row) * a.Cout + co0 + wn * 32 + r] = acc[t][e] * inv_x * inv_d;
        }
}

int spk_launch_wgrad_wm(const WgradArgs& a, hipStream_t st) {
    SPK_REQUIRE(a.KW == 3 && a.Cin % 64 == 0 && a.Cout % 64 == 0, "spk_conv_wgrad(2x2 waves): 3x3, Cin and Cout multiples of 64 (%d, %d)", a.Cin, a.Cout);
    SPK_REQUIRE(a.halo_h * a.halo_w <= 16 * WM_NX && a.TH * a.TW <= 16 * WM_ND, "spk_conv_wgrad(2x2 waves): tile %dx%d (halo %dx%d) exceeds the prefetch windows",
                a.TH, a.TW, a.halo_h, a.halo_w);
    const int npix_pad = ((a.TH * a.TW + 15) >> 4) << 4;
    const size_t lds_bytes = (size_t)a.halo_h * a.halo_w * 384 + (size_t)npix_pad * (2 * 192 + 64);
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_wgrad(2x2 waves): tile %dx%d needs %zu B of LDS", a.TH, a.TW, lds_bytes);
    dim3 grid(a.nsplit * (a.Cin / 64) * (a.Cout / 64));
    const int var = ((a.flags & SPK_IN_AFFINE_RELU) ? 1 : 0) | ((a.flags & SPK_DY_PRESPLIT) ? 2 : 0);
    // the shifted-window K loop (SH): stride 1, tile rows of whole 8-pixel half-steps.  Bit-identical and measured +-0 inside the step
    // (28 instead of 40 transposed reads per k-step change nothing: DESIGN.md section 7b) - compiled with SPK_EXPERIMENTAL only;
    // SPK_WGRAD_NOSHIFT keeps the plain form there (A/B)
#ifdef SPK_EXPERIMENTAL
    const bool sh = a.S == 1 && a.TW % 8 == 0 && !(a.flags & SPK_WGRAD_NOSHIFT);
    if (var == 3 && sh) hipLaunchKernelGGL((conv_wgrad_wm_kernel<3, true>), grid, dim3(256), lds_bytes, st, a);
    else if (var == 2 && sh) hipLaunchKernelGGL((conv_wgrad_wm_kernel<2, true>), grid, dim3(256), lds_bytes, st, a);
    else
#endif
#ifdef SPK_NO_FL_VARIANTS
    hipLaunchKernelGGL((conv_wgrad_wm_kernel<-1, false>), grid, dim3(256), lds_bytes, st, a);
#else
    if (var == 3) hipLaunchKernelGGL((conv_wgrad_wm_kernel<3, false>), grid, dim3(256), lds_bytes, st, a);
    else if (var == 2) hipLaunchKernelGGL((conv_wgrad_wm_kernel<2, false>), grid, dim3(256), lds_bytes, st, a);
    else hipLaunchKernelGGL((conv_wgrad_wm_kernel<-1, false>), grid, dim3(256), lds_bytes, st, a);
#endif
    SPK_LAUNCH_CHECK("spk_conv_wgrad(2x2 waves)");
    return 0;
}
