// Weight gradient of the 1x1 convolutions (Bottleneck blocks, downsample branches), f16x3 operands: the input-channel
// groups of a block play the part the nine taps play in conv_wgrad_split_kernel.
//
// conv_wgrad_split_kernel<1, ..> gives a block 32 input channels x 32*WN output channels: with one tap a staged element of
// dY feeds only 32 x 3 matrix products, every dY tile is converted and written to LDS Cin/32 times, and the kernel spends ~75
// VALU instructions of staging per MFMA (ResNet-101 step: 15 of 111 ms at 2.0 TB/s algorithmic, bound by that).  Here a block
// owns CG = 2 or 4 groups of 32 input channels: the X tile is [pixel][CG][terms][32 ch], a wave keeps CG accumulators
// (group g = "tap" g: same pixel, channels 32g..32g+31) and one dY fragment serves CG x 3 MFMAs.  dY is staged Cin/(32 CG)
// times instead of Cin/32, ~23 VALU instructions per MFMA.  Only the pixels a strided convolution reads are staged (the
// 3x3 kernel's halo tile would carry the skipped ones).  Otherwise the structure of conv_wgrad_split_kernel: persistent
// over pixel regions, next region prefetched into registers, transposing fragment reads, WK pixel-splits folded through LDS in
// a fixed order, one slab per region slice, spk_wgrad_reduce folds the slabs.
#include "conv_wgrad.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

static __device__ __forceinline__ s16x8 tr_read8g(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// VAR >= 0: compile-time variant of the staging switches (bit 0 fused input BatchNorm + ReLU on X, bit 1 dY is an f16 pair tensor);
// VAR < 0: read from the argument block
template <int WK, int WN, int CG, int VAR = -1>
__global__ __launch_bounds__(256, 2) void conv_wgrad_1x1_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    constexpr int PX = CG * 192;                          // X bytes per staged pixel: [CG][3-term pitch][32 ch fp16]
    constexpr int PD = WN * 192 + (WN > 1 ? 64 : 0);      // dY bytes per pixel (as conv_wgrad_split_kernel)
    constexpr int QX = CG * 8;                            // X float4 quads per pixel
    constexpr int PSX = 256 / QX;                         // X pixels per pass of the block
    constexpr int NXG = (WGRAD_MAX_PIX_1X1 + PSX - 1) / PSX;
    constexpr int QPP = WN * 8;
    constexpr int PSTEP = 256 / QPP;
    constexpr int ND = (WGRAD_MAX_PIX_1X1 + PSTEP - 1) / PSTEP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int wk = wave / WN, wn = wave % WN;
    // block -> (region slice g, channel groups): slices on one XCD, as wgrad_block in conv_wgrad_split.hip
    int g, ci0, co0;
    {
        const int ncgi = a.Cin / (32 * CG), M = ncgi * (a.Cout / (32 * WN));
        const int bid = blockIdx.x;
        int m;
        if ((a.nsplit & 7) == 0) {
            const int k = bid >> 3;
            m = k % M;
            g = (k / M) * 8 + (bid & 7);
        } else {
            m = bid % M;
            g = bid / M;
        }
        ci0 = (m % ncgi) * (32 * CG);
        co0 = (m / ncgi) * (32 * WN);
    }
    const int npix = a.TH * a.TW;
    const int nsteps_all = (npix + 15) >> 4;
    const int npix_pad = nsteps_all << 4;
    unsigned char* xs = ldsb;
    unsigned char* dys = ldsb + npix_pad * PX;
    const int flags = VAR >= 0 ? ((VAR & 1) ? SPK_IN_AFFINE_RELU : 0) : a.flags;
    const float sig_x = a.x_amax ? spk_sigma_from_amax_bits(*a.x_amax) : SPK_F16_ACT_SIGMA;
    const float sig_d = a.dy_amax ? spk_sigma_from_amax_bits(*a.dy_amax) : 1.f;
    const bool dy_pairs = VAR >= 0 ? (VAR & 2) != 0 : (a.flags & SPK_DY_PRESPLIT) != 0;    // dY is an f16 pair tensor: staged by plain copy

    f32x16 acc[CG];
#pragma unroll
    for (int t = 0; t < CG; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int qx = tid % QX, cq = tid % QPP;
    f32x4 px[NXG], pd[ND];
    unsigned inx = 0, ind = 0;
    const unsigned x_row = (unsigned)a.IW * a.Cin * 4u, x_px = (unsigned)a.Cin * 4u;
    const unsigned d_row = (unsigned)a.OW * a.Cout * 4u, d_px = (unsigned)a.Cout * 4u;
    const unsigned x_c = (unsigned)(ci0 + qx * 4) * 4u, d_c = (unsigned)(co0 + cq * 4) * 4u;
    auto prefetch = [&](int region) {
        int pt = region;
        const int tx = pt % a.tiles_x;
        pt /= a.tiles_x;
        const int ty = pt % a.tiles_y;
        const int b = pt / a.tiles_y;
        const int oy0 = ty * a.TH, ox0 = tx * a.TW;
        const char* xb = (const char*)(a.x + (size_t)b * a.IH * a.IW * a.Cin);
        const char* db = (const char*)(a.dy + (size_t)b * a.OH * a.OW * a.Cout);
        const unsigned x_safe = (unsigned)(oy0 * a.S) * x_row + (unsigned)(ox0 * a.S) * x_px + x_c;
        const unsigned d_safe = (unsigned)oy0 * d_row + (unsigned)ox0 * d_px + d_c;
        inx = 0;
        ind = 0;
#pragma unroll
        for (int u = 0; u < NXG; ++u) {
            const int p = tid / QX + PSX * u;
            const int ly = (int)__umulhi((unsigned)p, a.tw_magic);
            const int lx = p - ly * a.TW;
            const int oy = oy0 + ly, ox = ox0 + lx;
            const bool ok = p < npix && oy < a.OH && ox < a.OW;       // (pad 0: the input pixel (oy S, ox S) is inside then)
            if (ok) inx |= 1u << u;
            const unsigned off = ok ? (unsigned)(oy * a.S) * x_row + (unsigned)(ox * a.S) * x_px + x_c : x_safe;
            px[u] = *(const f32x4*)(xb + off);
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int p = tid / QPP + PSTEP * u;
            const int ly = (int)__umulhi((unsigned)p, a.tw_magic);
            const int lx = p - ly * a.TW;
            const int oy = oy0 + ly, ox = ox0 + lx;
            const bool ok = p < npix && oy < a.OH && ox < a.OW;
            if (ok) ind |= 1u << u;
            const unsigned off = ok ? (unsigned)oy * d_row + (unsigned)ox * d_px + d_c : d_safe;
            pd[u] = *(const f32x4*)(db + off);
        }
    };
    auto publish = [&]() {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (flags & SPK_IN_AFFINE_RELU) {
            sc = *(const f32x4*)(a.in_scale + ci0 + qx * 4);
            sh = *(const f32x4*)(a.in_shift + ci0 + qx * 4);
        }
#pragma unroll
        for (int u = 0; u < NXG; ++u) {
            const int p = tid / QX + PSX * u;
            f32x4 w = px[u];
            if (flags & SPK_IN_AFFINE_RELU) {
                w = w * sc + sh;
                w[0] = fmaxf(w[0], 0.f);
                w[1] = fmaxf(w[1], 0.f);
                w[2] = fmaxf(w[2], 0.f);
                w[3] = fmaxf(w[3], 0.f);
            }
            if (!((inx >> u) & 1)) w = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p < npix_pad) {
                uint2* dst = (uint2*)(xs + p * PX + (qx >> 3) * 192) + (qx & 7);
                uint2 t0, t1;
                split2h(w, sig_x, t0, t1);
                dst[0] = t0;
                dst[8] = t1;
            }
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int p = tid / QPP + PSTEP * u;
            const f32x4 w = ((ind >> u) & 1) ? pd[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p < npix_pad) {
                uint2* dst = (uint2*)(dys + p * PD + (cq >> 3) * 192) + (cq & 7);
                uint2 t0, t1;
                spk_terms(w, sig_d, dy_pairs, t0, t1);
                dst[0] = t0;
                dst[8] = t1;
            }
        }
    };

    const int g16 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const int col_off = (g16 & 1) * 32 + p4 * 8;
    auto mma = [&](f32x16& c, const s16x8* af, const s16x8* bf) {      // h1*g2, h2*g1, h1*g1: the order of the other kernels
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
        for (int j = wk; j < nsteps_all; j += WK) {
            int xa[2], da[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int pix = j * 16 + 8 * h + 4 * blk + q;
                xa[blk] = pix * PX + col_off;
                da[blk] = pix * PD + wn * 192 + col_off;
            }
            s16x8 bf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) bf[s] = tr_read8g(dys + da[0] + s * 64, dys + da[1] + s * 64);
            s16x8 a0[2], a1[2];
            auto load_a = [&](s16x8* af, int t) {
#pragma unroll
                for (int s = 0; s < 2; ++s) af[s] = tr_read8g(xs + xa[0] + t * 192 + s * 64, xs + xa[1] + t * 192 + s * 64);
            };
            load_a(a0, 0);
#pragma unroll
            for (int t = 0; t < CG; t += 2) {
                if (t + 1 < CG) load_a(a1, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(acc[t], a0, bf);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 2 < CG) load_a(a0, t + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < CG) mma(acc[t + 1], a1, bf);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    const int r = lane & 31;
    if (WK > 1) {
        float* red = (float*)ldsb;  // [WN][CG][16][64]
#pragma unroll 1
        for (int src = 1; src < WK; ++src) {
            __syncthreads();
            if (wk == src) {
#pragma unroll
                for (int t = 0; t < CG; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((wn * CG + t) * 16 + e) * 64 + lane] = acc[t][e];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int t = 0; t < CG; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] += red[((wn * CG + t) * 16 + e) * 64 + lane];
            }
        }
    }
    if (wk == 0) {
        float* slab = a.partial + (size_t)g * a.Cin * a.Cout;       // one tap: [Cin][Cout]
        const float inv_x = 1.f / sig_x, inv_d = 1.f / sig_d;
#pragma unroll
        for (int t = 0; t < CG; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                slab[(size_t)(ci0 + t * 32 + row) * a.Cout + co0 + wn * 32 + r] = acc[t][e] * inv_x * inv_d;
            }
    }
}

template <int WK, int WN, int CG>
static int launch_g(const WgradArgs& a, hipStream_t st) {
    const int npix_pad = ((a.TH * a.TW + 15) >> 4) << 4;
    constexpr int PD = WN * 192 + (WN > 1 ? 64 : 0);
    size_t lds_bytes = (size_t)npix_pad * (CG * 192 + PD);
    const size_t red_bytes = (WK > 1) ? (size_t)WN * CG * 16 * 64 * sizeof(float) : 0;
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_wgrad(1x1): tile %dx%d needs %zu B of LDS", a.TH, a.TW, lds_bytes);
    dim3 grid(a.nsplit * (a.Cin / (32 * CG)) * (a.Cout / (32 * WN)));
    const int var = ((a.flags & SPK_IN_AFFINE_RELU) ? 1 : 0) | ((a.flags & SPK_DY_PRESPLIT) ? 2 : 0);
    if (var == 3) hipLaunchKernelGGL((conv_wgrad_1x1_kernel<WK, WN, CG, 3>), grid, dim3(256), lds_bytes, st, a);
    else if (var == 2) hipLaunchKernelGGL((conv_wgrad_1x1_kernel<WK, WN, CG, 2>), grid, dim3(256), lds_bytes, st, a);
    else hipLaunchKernelGGL((conv_wgrad_1x1_kernel<WK, WN, CG>), grid, dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_wgrad(1x1)");
    return 0;
}

// CG = input-channel groups per block (2 or 4; Cin % (32 CG) == 0); the tile holds at most WGRAD_MAX_PIX_1X1 pixels
int spk_launch_wgrad_1x1(const WgradArgs& a, int WN, int CG, hipStream_t st) {
    SPK_REQUIRE(a.KW == 1 && (CG == 2 || CG == 4) && a.Cin % (32 * CG) == 0, "spk_conv_wgrad(1x1): Cin=%d, CG=%d", a.Cin, CG);
    SPK_REQUIRE(a.TH * a.TW <= WGRAD_MAX_PIX_1X1, "spk_conv_wgrad(1x1): tile %dx%d exceeds %d pixels", a.TH, a.TW, WGRAD_MAX_PIX_1X1);
    SPK_REQUIRE(WN == 2 || WN == 4, "spk_conv_wgrad(1x1): WN=%d", WN);
    if (WN == 2) return CG == 4 ? launch_g<2, 2, 4>(a, st) : launch_g<2, 2, 2>(a, st);
    return CG == 4 ? launch_g<1, 4, 4>(a, st) : launch_g<1, 4, 2>(a, st);
}
