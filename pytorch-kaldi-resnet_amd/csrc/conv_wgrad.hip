// Weight gradient of the 3x3 / 1x1 convolutions on the fp32 matrix cores.
//   dW[co][ci][kh][kw] = sum_{b,oy,ox} dY[b][oy][ox][co] * X[b][oy*S+kh-pad][ox*S+kw-pad][ci]
// (autograd of nn.Conv2d(bias=False), reference scripts/model.py:12-15,233-234 + train_resnet.py:327).
// GEMM view per tap: D[m = ci][n = co] += sum_k A[ci][k] * B[k][co] with k = output pixel, so both MFMA
// operands are "32 consecutive channels of one pixel": lane l supplies X[pixel(l>>5) shifted by the
// tap][ci = l&31] and dY[pixel(l>>5)][co = l&31], read from LDS tiles stored [pixel][32 channels]
// (conflict-free ds_read_b32).  One dY read feeds all NTAPS MFMAs of the k-step.
// A wave owns one 32(ci) x 32(co) tile for all taps (NTAPS x 16 accumulator registers); the four
// waves of a block are WK pixel-splits x WN cout tiles.  Blocks are persistent over pixel regions
// (grid.x = nsplit) and emit one partial slab each; spk_wgrad_reduce sums the slabs in a fixed
// order (deterministic) and writes OIHW.
// (ABL_NO_* macros select diagnostic ablation builds - wrong results by construction - used to price each phase of the
//  kernel: SPK_CXXFLAGS="-DABL_NO_STAGE" python build.py, then SPK_LIB=<variant.so> tools/conv_bench.py.)
#include "conv_wgrad.h"

// bf16-split instantiations live in conv_wgrad_split.hip
int spk_launch_wgrad_split(const WgradArgs& a, int WN, int split, hipStream_t st);
int spk_launch_wgrad_ws(const WgradArgs& a, int WN, hipStream_t st);
int spk_launch_wgrad_pipe(const WgradArgs& a, int WN, hipStream_t st);
int spk_launch_wgrad_1x1(const WgradArgs& a, int WN, int CG, hipStream_t st);
int spk_launch_wgrad_wm16(const WgradArgs& a, hipStream_t st);                 // conv_wgrad_wm16.hip: the same on 16x16x32, dy by LDS DMA
int spk_launch_wgrad_c32m16(const WgradArgs& a, hipStream_t st);               //   ... and its layout for 32-channel groups (four slabs per block)
int spk_launch_wgrad_wm(const WgradArgs& a, hipStream_t st);                   // conv_wgrad_wm.hip: 2 x 2 wave layout (3x3)   // conv_wgrad_1x1.hip: input-channel groups as "taps"      // conv_wgrad_pipe.hip: in-wave pipelined form        // conv_wgrad_split.hip: producer / consumer form

template <int NTAPS, int WK, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> scalar registers, scalar branches
    const int r = lane & 31, h = lane >> 5;
    const int wk = wave / WN, wn = wave % WN;
    const int ci0 = blockIdx.y * 32;
    const int co0 = blockIdx.z * (32 * WN);
    const int halo_pix = a.halo_h * a.halo_w;
    const int npix = a.TH * a.TW;
    float* xs = lds;                    // [halo_pix][32]
    float* dys = lds + halo_pix * 32;   // [npix][WN*32]
    const int flags = a.flags;

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    constexpr int KS = (NTAPS == 9) ? 3 : 1;   // kernel size
    const int row_stride = a.halo_w * 32;      // floats between two halo rows

    const int quad = tid & 7;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (flags & SPK_IN_AFFINE_RELU) {
        sc = *(const f32x4*)(a.in_scale + ci0 + quad * 4);
        sh = *(const f32x4*)(a.in_shift + ci0 + quad * 4);
    }
    const int half_w = a.TW >> 1;
    const int ksteps = a.TH * half_w;

    // Region pipeline ("issue early / write late"): the global loads of region i+1 are issued into registers right
    // after the barrier that publishes region i's LDS tiles, fly during region i's MFMAs, and are written to LDS
    // (with the fused BN+ReLU) after the barrier that ends region i.  The host caps the tiles so that NX + ND
    // 16-byte registers per thread hold one region (spk_conv_wgrad: halo_pix <= 32*NX, npix <= PSTEP*ND).
    constexpr int NX = WGRAD_NX;                 // X halo float4 per thread
    constexpr int ND = WGRAD_ND;                 // dY float4 per thread
    constexpr int QPP = WN * 8;                  // dY float4 quads per pixel
    constexpr int PSTEP = 256 / QPP;             // dY pixels per pass
    const int cq = tid % QPP;
    f32x4 px[NX], pd[ND];
    unsigned inx = 0, ind = 0;                   // in-image bit masks of the prefetched slots

    auto prefetch = [&](int region) {
        int pt = region;
        const int tx = pt % a.tiles_x;
        pt /= a.tiles_x;
        const int ty = pt % a.tiles_y;
        const int b = pt / a.tiles_y;
        const int oy0 = ty * a.TH, ox0 = tx * a.TW;
        const int iy0 = oy0 * a.S - a.pad, ix0 = ox0 * a.S - a.pad;
        inx = 0;
        ind = 0;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            int p = (tid >> 3) + 32 * u;
            p = p < halo_pix ? p : halo_pix - 1;
            const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
            const int hx = p - hy * a.halo_w;
            const int iy = iy0 + hy, ix = ix0 + hx;
            if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) inx |= 1u << u;
            const int cy = min(max(iy, 0), a.IH - 1), cx = min(max(ix, 0), a.IW - 1);
#ifndef ABL_NO_STAGE
            px[u] = *(const f32x4*)(a.x + (size_t)((b * a.IH + cy) * a.IW + cx) * a.Cin + ci0 + quad * 4);
#endif
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            int p = tid / QPP + PSTEP * u;
            p = p < npix ? p : npix - 1;
            const int ly = (int)__umulhi((unsigned)p, a.tw_magic);
            const int lx = p - ly * a.TW;
            const int oy = oy0 + ly, ox = ox0 + lx;
            if (oy < a.OH && ox < a.OW) ind |= 1u << u;
            const int cy = min(oy, a.OH - 1), cx = min(ox, a.OW - 1);
#ifndef ABL_NO_STAGE
            pd[u] = *(const f32x4*)(a.dy + (size_t)((b * a.OH + cy) * a.OW + cx) * a.Cout + co0 + cq * 4);
#endif
        }
    };
    auto publish = [&]() {      // registers -> LDS (zero outside the image; BN+ReLU of the producing layer fused)
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int p = (tid >> 3) + 32 * u;
            f32x4 w = px[u];
            if (flags & SPK_IN_AFFINE_RELU) {
                w = w * sc + sh;
                w[0] = fmaxf(w[0], 0.f);
                w[1] = fmaxf(w[1], 0.f);
                w[2] = fmaxf(w[2], 0.f);
                w[3] = fmaxf(w[3], 0.f);
            }
            if (!((inx >> u) & 1)) w = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p < halo_pix) *(f32x4*)(xs + p * 32 + quad * 4) = w;
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int p = tid / QPP + PSTEP * u;
            const f32x4 w = ((ind >> u) & 1) ? pd[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p < npix) *(f32x4*)(dys + p * (WN * 32) + cq * 4) = w;
        }
    };
#ifdef ABL_NO_STAGE
    for (int u = 0; u < NX; ++u) px[u] = (f32x4){1.f, 1.f, 1.f, 1.f};
    for (int u = 0; u < ND; ++u) pd[u] = (f32x4){1.f, 1.f, 1.f, 1.f};
#endif

    int region = blockIdx.x;
    if (region < a.nregions) prefetch(region);
    for (; region < a.nregions; region += a.nsplit) {
        __syncthreads();   // previous region fully consumed
        publish();
        __syncthreads();
        if (region + a.nsplit < a.nregions) prefetch(region + a.nsplit);

        // k-steps of this wave: kk = wk, wk + WK, ...; operands of step i+1 are read from LDS while the NTAPS MFMAs
        // of step i issue (ping-pong register sets), so the matrix pipe does not wait on LDS latency.
        const int nsteps = (ksteps - wk + WK - 1) / WK;
        int qy = wk / half_w, j = wk - qy * half_w;   // position of the next step to load
        auto load_step = [&](float& bval, float* av) {
            const int qx = 2 * j + h;
#ifdef ABL_NO_LDSREAD
            asm volatile("" : "+v"(bval) : "s"(qy));
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) asm volatile("" : "+v"(av[t]) : "s"(j));
#else
            bval = dys[(qy * a.TW + qx) * (WN * 32) + wn * 32 + r];
            const int xb = ((qy * a.S) * a.halo_w + qx * a.S) * 32 + r;
            // taps of one kernel row are 32 floats apart: the compiler pairs them into ds_read2_b32
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) av[kh * KS + kw] = xs[xb + kh * row_stride + kw * 32];
#endif
            j += WK;
            while (j >= half_w) {
                j -= half_w;
                ++qy;
            }
        };
        auto mma = [&](float bval, const float* av) {
#pragma unroll
            for (int t = 0; t < NTAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bval, acc[t], 0, 0, 0);
        };
        float bv0 = 1.f, bv1 = 1.f, av0[NTAPS], av1[NTAPS];
#ifdef ABL_NO_LDSREAD
        for (int t = 0; t < NTAPS; ++t) av0[t] = av1[t] = 1.f;
#endif
        if (nsteps > 0) load_step(bv0, av0);
        for (int i = 0; i < nsteps; i += 2) {
            const bool has1 = i + 1 < nsteps;
            if (has1) load_step(bv1, av1);
            __builtin_amdgcn_sched_barrier(0);
            mma(bv0, av0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < nsteps) load_step(bv0, av0);
            __builtin_amdgcn_sched_barrier(0);
            if (has1) mma(bv1, av1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // fold the WK pixel-splits into wk == 0 through LDS (fixed order -> deterministic)
    if (WK > 1) {
        float* red = lds;  // [WN][NTAPS][16][64]
#pragma unroll 1
        for (int src = 1; src < WK; ++src) {
            __syncthreads();
            if (wk == src) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((wn * NTAPS + t) * 16 + e) * 64 + lane] = acc[t][e];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] += red[((wn * NTAPS + t) * 16 + e) * 64 + lane];
            }
        }
    }
    if (wk == 0) {
        float* slab = a.partial + (size_t)blockIdx.x * NTAPS * a.Cin * a.Cout;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                slab[((size_t)t * a.Cin + ci0 + row) * a.Cout + co0 + wn * 32 + r] = acc[t][e];
            }
    }
}

// partial [nslab][ntaps][Cin][Cout] -> dW OIHW [Cout][Cin][ntaps]; optional accumulate
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nslab, int ntaps,
                                    int Cin, int Cout, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = ntaps * Cin * Cout;
    if (idx >= total) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= nslab; k += 4) {
        s0 += partial[(size_t)(k + 0) * total + idx];
        s1 += partial[(size_t)(k + 1) * total + idx];
        s2 += partial[(size_t)(k + 2) * total + idx];
        s3 += partial[(size_t)(k + 3) * total + idx];
    }
    for (; k < nslab; ++k) s0 += partial[(size_t)k * total + idx];
    const float s = (s0 + s1) + (s2 + s3);
    const int co = idx % Cout;
    const int ci = (idx / Cout) % Cin;
    const int t = idx / (Cout * Cin);
    float* dst = dw + ((size_t)co * Cin + ci) * ntaps + t;
    *dst = accumulate ? *dst + s : s;
}

template <int NTAPS, int WK, int WN>
static int launch_wgrad(const WgradArgs& a, size_t lds_bytes, hipStream_t st) {
    dim3 grid(a.nsplit, a.Cin / 32, a.Cout / (32 * WN));
    hipLaunchKernelGGL((conv_wgrad_kernel<NTAPS, WK, WN>), grid, dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_wgrad");
    return 0;
}

extern "C" int spk_conv_wgrad_limits(int WN, int* max_halo_pix, int* max_tile_pix) {
    if (WN != 1 && WN != 2 && WN != 4) return -1;
    *max_halo_pix = 32 * WGRAD_NX;
    *max_tile_pix = (256 / (8 * WN)) * WGRAD_ND;
    return 0;
}

extern "C" size_t spk_conv_wgrad_workspace(int nsplit, int ksize, int Cin, int Cout) {
    return (size_t)nsplit * ksize * ksize * Cin * Cout * sizeof(float);
}

extern "C" int spk_wgrad_reduce(const float* partial, float* dw, int nslab, int ksize, int Cin, int Cout, int accumulate,
                                void* stream) {
    SPK_REQUIRE(partial && dw && nslab >= 1 && (ksize == 1 || ksize == 3), "spk_wgrad_reduce: bad arguments");
    const int ntaps = ksize * ksize, total = ntaps * Cin * Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(spk_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, partial, dw, nslab,
                       ntaps, Cin, Cout, accumulate);
    SPK_LAUNCH_CHECK("spk_wgrad_reduce");
    return 0;
}

extern "C" int spk_conv_wgrad(const float* x, const float* dy, float* dw, float* partial, const float* in_scale,
                              const float* in_shift, int B, int IH, int IW, int Cin, int OH, int OW, int Cout,
                              int ksize, int stride, int TH, int TW, int WN, int nsplit, int flags, int accumulate,
                              int split, const unsigned* dy_amax, const unsigned* x_amax, void* stream) {
    SPK_REQUIRE(x && dy && dw && partial, "spk_conv_wgrad: null pointer");
    SPK_REQUIRE(ksize == 1 || ksize == 3, "spk_conv_wgrad: ksize=%d unsupported", ksize);
    SPK_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "spk_conv_wgrad: channels (%d,%d) must be multiples of 32", Cin, Cout);
    SPK_REQUIRE(WN == 1 || WN == 2 || WN == 4, "spk_conv_wgrad: WN=%d", WN);
    SPK_REQUIRE(Cout % (32 * WN) == 0, "spk_conv_wgrad: Cout=%d not a multiple of 32*WN", Cout);
    SPK_REQUIRE(TH >= 1 && TW >= 2 && (TW % 2) == 0, "spk_conv_wgrad: TW=%d must be even", TW);
    SPK_REQUIRE(nsplit >= 1, "spk_conv_wgrad: nsplit");
    SPK_REQUIRE(!(flags & SPK_IN_AFFINE_RELU) || (in_scale && in_shift), "spk_conv_wgrad: IN_AFFINE_RELU needs scale/shift");
    SPK_REQUIRE((long long)B * IH * IW * Cin < 2147483647LL * 4 && (long long)B * OH * OW * Cout < 2147483647LL * 4,
                "spk_conv_wgrad: tensor too large");
    SPK_REQUIRE((long long)IH * IW * Cin * 4 < 4294967295LL && (long long)OH * OW * Cout * 4 < 4294967295LL,
                "spk_conv_wgrad: one image exceeds 32-bit byte offsets");
    SPK_REQUIRE((long long)IW * Cin * 4 < (1 << 24) && (long long)OW * Cout * 4 < (1 << 24) && IH < (1 << 24) && OH < (1 << 24),
                "spk_conv_wgrad: a tensor row exceeds the 24-bit multiplier range");
    WgradArgs a;
    a.x = x; a.dy = dy; a.partial = partial; a.in_scale = in_scale; a.in_shift = in_shift;
    a.B = B; a.IH = IH; a.IW = IW; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout;
    a.S = stride; a.KW = ksize; a.pad = (ksize == 3) ? 1 : 0;
    a.TH = TH; a.TW = TW; a.tiles_y = spk_ceil_div(OH, TH); a.tiles_x = spk_ceil_div(OW, TW);
    a.nregions = B * a.tiles_y * a.tiles_x;
    SPK_REQUIRE(nsplit <= a.nregions, "spk_conv_wgrad: nsplit=%d exceeds the %d pixel regions", nsplit, a.nregions);
    a.nsplit = nsplit;
    a.halo_h = (TH - 1) * stride + ksize; a.halo_w = (TW - 1) * stride + ksize;
    a.halo_w_magic = (unsigned)((0x100000000ULL + (unsigned long long)a.halo_w - 1) / (unsigned long long)a.halo_w);
    a.tw_magic = (unsigned)((0x100000000ULL + (unsigned long long)TW - 1) / (unsigned long long)TW);
    a.flags = flags;
    a.dy_amax = dy_amax; a.x_amax = x_amax;
    SPK_REQUIRE(split == 0 || split == 3 || ((split == 6 || split == 9) && ksize == 3),
                "spk_conv_wgrad: split=%d (0; 3 = f16x3, any kernel size; 6 / 9 = bf16 terms, 3x3 only)", split);
    // f16x3: both operand scales ALWAYS come from slots (a NULL dy_amax meant scale 1 - typical gradients of 1e-5..1e-8
    // then sat in or below fp16's subnormal range - a handful of significand bits at best: dw silently wrong)
    SPK_REQUIRE(split != 3 || (dy_amax && x_amax), "spk_conv_wgrad: the f16x3 operand mode needs dy_amax and x_amax (slots with the float bits of the operands' absmax or of upper bounds)");
    SPK_REQUIRE(!(flags & SPK_DY_PRESPLIT) || (split == 3 && !(flags & (SPK_CONV_PIPE | SPK_CONV_WS))),
                "spk_conv_wgrad: DY_PRESPLIT (dy as an f16 pair tensor) needs the f16x3 mode (not the opt-in pipelined / wave-specialised forms)");
    if ((flags & SPK_WGRAD_M16) && !(flags & SPK_WGRAD_GROUPS)) {      // 32-channel groups on 16x16x32, dy by LDS DMA
        SPK_REQUIRE(split == 3 && ksize == 3 && WN == 1, "spk_conv_wgrad: SPK_WGRAD_M16 without SPK_WGRAD_GROUPS is the 32-channel-group 3x3 kernel (f16x3, WN = 1)");
        a.flags = flags & (SPK_IN_AFFINE_RELU | SPK_DY_PRESPLIT);
        return spk_launch_wgrad_c32m16(a, (hipStream_t)stream);
    }
    if (flags & SPK_WGRAD_GROUPS) {       // f16x3: bits 12-13 of flags = log2 of the input-channel groups per block
        SPK_REQUIRE(split == 3, "spk_conv_wgrad: the grouped kernels exist in the f16x3 mode");
        const int cg = 1 << ((flags >> 12) & 3);
        a.flags = flags & (SPK_IN_AFFINE_RELU | SPK_DY_PRESPLIT | SPK_WGRAD_NOSHIFT);
        if (ksize == 3) {                 // 3x3: two groups = the 2 x 2 wave layout
            SPK_REQUIRE(cg == 2 && WN == 2, "spk_conv_wgrad: the 3x3 grouped kernel has 2 input-channel groups and WN = 2");
            if (flags & SPK_WGRAD_M16) return spk_launch_wgrad_wm16(a, (hipStream_t)stream);
            return spk_launch_wgrad_wm(a, (hipStream_t)stream);
        }
        return spk_launch_wgrad_1x1(a, WN, cg, (hipStream_t)stream);
    }
    SPK_REQUIRE(a.halo_h * a.halo_w <= 32 * WGRAD_NX, "spk_conv_wgrad: halo %dx%d exceeds the %d-pixel prefetch window",
                a.halo_h, a.halo_w, 32 * WGRAD_NX);
    SPK_REQUIRE(TH * TW <= (256 / (8 * WN)) * WGRAD_ND, "spk_conv_wgrad: tile %dx%d exceeds the %d-pixel dY prefetch window (WN=%d)",
                TH, TW, (256 / (8 * WN)) * WGRAD_ND, WN);
#ifndef SPK_EXPERIMENTAL
    SPK_REQUIRE(!(flags & (SPK_CONV_PIPE | SPK_CONV_WS)),
                "spk_conv_wgrad: the pipelined / producer-consumer weight gradients are experimental forms: build with SPK_EXPERIMENTAL=1");
#else
    if (flags & SPK_CONV_PIPE) {
        SPK_REQUIRE(split == 3 && ksize == 3, "spk_conv_wgrad: the pipelined kernel exists for 3x3 in the f16x3 mode");
        a.flags = flags & ~SPK_CONV_PIPE;
        return spk_launch_wgrad_pipe(a, WN, (hipStream_t)stream);
    }
    if (flags & SPK_CONV_WS) {
        SPK_REQUIRE(split == 3 && ksize == 3, "spk_conv_wgrad: the producer / consumer kernel exists for 3x3 in the f16x3 mode");
        a.flags = flags & ~SPK_CONV_WS;
        return spk_launch_wgrad_ws(a, WN, (hipStream_t)stream);
    }
#endif
    if (split) return spk_launch_wgrad_split(a, WN, split, (hipStream_t)stream);
    const int ntaps = ksize * ksize;
    const int WK = 4 / WN;
    size_t lds_bytes = ((size_t)a.halo_h * a.halo_w * 32 + (size_t)TH * TW * WN * 32) * sizeof(float);
    const size_t red_bytes = (WK > 1) ? (size_t)WN * ntaps * 16 * 64 * sizeof(float) : 0;
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_wgrad: tile %dx%d needs %zu B of LDS", TH, TW, lds_bytes);
    hipStream_t st = (hipStream_t)stream;
    int rc = -1;
    if (ntaps == 9) {
        if (WN == 1) rc = launch_wgrad<9, 4, 1>(a, lds_bytes, st);
        else if (WN == 2) rc = launch_wgrad<9, 2, 2>(a, lds_bytes, st);
        else rc = launch_wgrad<9, 1, 4>(a, lds_bytes, st);
    } else {
        if (WN == 1) rc = launch_wgrad<1, 4, 1>(a, lds_bytes, st);
        else if (WN == 2) rc = launch_wgrad<1, 2, 2>(a, lds_bytes, st);
        else rc = launch_wgrad<1, 1, 4>(a, lds_bytes, st);
    }
    return rc;
}
