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
#include "spk_common.h"

struct WgradArgs {
    const float* x;
    const float* dy;
    float* partial;
    const float* in_scale;
    const float* in_shift;
    int B, IH, IW, Cin, OH, OW, Cout;
    int S, KW, pad;
    int TH, TW, tiles_y, tiles_x, nregions, nsplit;
    int halo_h, halo_w;
    int flags;
};

template <int NTAPS, int WK, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    int toff[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) toff[t] = ((t / a.KW) * a.halo_w + (t % a.KW)) * 32;

    const int quad = tid & 7;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (flags & SPK_IN_AFFINE_RELU) {
        sc = *(const f32x4*)(a.in_scale + ci0 + quad * 4);
        sh = *(const f32x4*)(a.in_shift + ci0 + quad * 4);
    }
    const int half_w = a.TW >> 1;
    const int ksteps = a.TH * half_w;

    for (int region = blockIdx.x; region < a.nregions; region += a.nsplit) {
        int pt = region;
        const int tx = pt % a.tiles_x;
        pt /= a.tiles_x;
        const int ty = pt % a.tiles_y;
        const int b = pt / a.tiles_y;
        const int oy0 = ty * a.TH, ox0 = tx * a.TW;
        const int iy0 = oy0 * a.S - a.pad, ix0 = ox0 * a.S - a.pad;
        __syncthreads();  // previous region fully consumed
        {   // X halo tile, 32 channels, optional fused BN+ReLU of the producing layer
            int p = tid >> 3;
            int hy = p / a.halo_w, hx = p - hy * a.halo_w;
            for (; p < halo_pix; p += 32) {
                const int iy = iy0 + hy, ix = ix0 + hx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) {
                    v = *(const f32x4*)(a.x + (size_t)((b * a.IH + iy) * a.IW + ix) * a.Cin + ci0 + quad * 4);
                    if (flags & SPK_IN_AFFINE_RELU) {
                        v = v * sc + sh;
                        v[0] = fmaxf(v[0], 0.f);
                        v[1] = fmaxf(v[1], 0.f);
                        v[2] = fmaxf(v[2], 0.f);
                        v[3] = fmaxf(v[3], 0.f);
                    }
                }
                *(f32x4*)(xs + p * 32 + quad * 4) = v;
                hx += 32;
                while (hx >= a.halo_w) {
                    hx -= a.halo_w;
                    ++hy;
                }
            }
        }
        {   // dY tile: zero outside the image, so border pixels contribute nothing
            constexpr int QPP = WN * 8;         // float4 quads per pixel
            constexpr int PSTEP = 256 / QPP;    // pixels per pass
            const int cq = tid % QPP;
            int p = tid / QPP;
            int ly = p / a.TW, lx = p - ly * a.TW;
            for (; p < npix; p += PSTEP) {
                const int oy = oy0 + ly, ox = ox0 + lx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (oy < a.OH && ox < a.OW)
                    v = *(const f32x4*)(a.dy + (size_t)((b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + cq * 4);
                *(f32x4*)(dys + p * (WN * 32) + cq * 4) = v;
                lx += PSTEP;
                while (lx >= a.TW) {
                    lx -= a.TW;
                    ++ly;
                }
            }
        }
        __syncthreads();

        int kk = wk;
        int qy = kk / half_w, j = kk - qy * half_w;
        for (; kk < ksteps; kk += WK) {
            const int qx = 2 * j + h;
            const float bval = dys[(qy * a.TW + qx) * (WN * 32) + wn * 32 + r];
            const int xb = ((qy * a.S) * a.halo_w + qx * a.S) * 32 + r;
            float av[NTAPS];
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) av[t] = xs[xb + toff[t]];
#pragma unroll
            for (int t = 0; t < NTAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bval, acc[t], 0, 0, 0);
            j += WK;
            while (j >= half_w) {
                j -= half_w;
                ++qy;
            }
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

extern "C" size_t spk_conv_wgrad_workspace(int nsplit, int ksize, int Cin, int Cout) {
    return (size_t)nsplit * ksize * ksize * Cin * Cout * sizeof(float);
}

extern "C" int spk_conv_wgrad(const float* x, const float* dy, float* dw, float* partial, const float* in_scale,
                              const float* in_shift, int B, int IH, int IW, int Cin, int OH, int OW, int Cout,
                              int ksize, int stride, int TH, int TW, int WN, int nsplit, int flags, int accumulate,
                              void* stream) {
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
    WgradArgs a;
    a.x = x; a.dy = dy; a.partial = partial; a.in_scale = in_scale; a.in_shift = in_shift;
    a.B = B; a.IH = IH; a.IW = IW; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout;
    a.S = stride; a.KW = ksize; a.pad = (ksize == 3) ? 1 : 0;
    a.TH = TH; a.TW = TW; a.tiles_y = spk_ceil_div(OH, TH); a.tiles_x = spk_ceil_div(OW, TW);
    a.nregions = B * a.tiles_y * a.tiles_x;
    if (nsplit > a.nregions) nsplit = a.nregions;
    a.nsplit = nsplit;
    a.halo_h = (TH - 1) * stride + ksize; a.halo_w = (TW - 1) * stride + ksize;
    a.flags = flags;
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
    if (rc) return rc;
    const int total = ntaps * Cin * Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(spk_ceil_div(total, 256)), dim3(256), 0, st, partial, dw, nsplit, ntaps,
                       Cin, Cout, accumulate);
    SPK_LAUNCH_CHECK("spk_wgrad_reduce");
    return 0;
}
