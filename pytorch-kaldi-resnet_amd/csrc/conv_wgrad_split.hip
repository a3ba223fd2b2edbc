// Weight gradient of the 3x3 convolutions with bf16-split operands (see conv_wgrad.hip for the fp32-operand kernel and
// conv_kernel.h for the splitting idea): fp32 X and dY go in, fp32 dW comes out; each operand value is staged into LDS as
// three bf16 terms whose sum is the fp32 value and the 6 most significant (or all 9) cross terms are multiplied on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  SPLIT == 3: the fp16 two-term form (spk_common.h): X scaled by the
// static activation scale, dY by the power of two that its absmax hand-off fixes, three products on
// v_mfma_f32_32x32x16_f16, the slabs scaled back; the LDS images keep the three-term pitch (the third plane stays unused).
//
// GEMM view per tap: D[ci][co] += sum_k X[pixel k shifted by the tap][ci] * dY[pixel k][co], k = 16 output pixels per
// MFMA.  Both operands are K-strided in memory ([pixel][channel]), so the LDS images stay [pixel][term][32 channels]
// (one ds_write_b64 per term while publishing) and the fragments are gathered with the gfx950 transposing read
// ds_read_b64_tr_b16: a 16-lane group reads 4 pixel rows x 16 channels and every lane receives ITS channel of the four
// pixels - two such reads are the 8 consecutive k of a lane's A (or B) operand.  Row pitches of 192 B (X) and
// WN*192 (+64) B (dY) put the four rows of a block 192 B (mod 256) apart: conflict-free.
#include "conv_wgrad.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

static __device__ __forceinline__ s16x8 tr_read8(const unsigned char* p0, const unsigned char* p1) {
#ifdef WG_ABL_NO_FRAG
    s16x8 v = {1, 2, 3, 4, 5, 6, 7, 8};
    asm volatile("" : "+v"(v) : "v"(p0), "v"(p1));
    return v;
#endif
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Block -> (pixel-region slice g, channel-group pair m).  All Cin/32 x Cout/(32 WN) blocks of a slice walk the same regions
// at the same time and re-read the same X and dY tiles, so they are given adjacent places in launch order ON ONE XCD
// (hardware deals consecutive block ids to the eight XCDs in turn): the re-reads hit that XCD's L2 instead of going to
// memory Cout/(32 WN) + Cin/32 times (the 1x1 convolutions of the Bottleneck blocks are bound by exactly that traffic).
// Which slices share an XCD: WGRAD_XCD_BAND = 1 gives every XCD a contiguous BAND of nsplit / 8 slices - slice g walks the regions g,
// g + nsplit, ..., so at any time an XCD works on nsplit / 8 neighbouring tiles of an image and the halo pixels of their X tiles
// (1.4-1.56 x the tile) meet in that XCD's L2; 0: slices dealt round-robin (neighbouring tiles on eight different L2s: every halo pixel
// is fetched from the fabric once per tile that touches it - profiles/pmc_traffic.json before the change).  The slab a slice writes
// and the regions it sums are the same either way: bit-identical results.
#ifndef WGRAD_XCD_BAND
#define WGRAD_XCD_BAND 1
#endif
struct WgradBlock { int g, ci0, co0; };
template <int WN>
static __device__ __forceinline__ WgradBlock wgrad_block(const WgradArgs& a) {
    const int ncgi = a.Cin >> 5, M = ncgi * (a.Cout / (32 * WN));
    const int bid = blockIdx.x;
    int g, m;
    if ((a.nsplit & 7) == 0) {
        const int k = bid >> 3;
        m = k % M;
        g = WGRAD_XCD_BAND ? (bid & 7) * (a.nsplit >> 3) + k / M : (k / M) * 8 + (bid & 7);
    } else {
        m = bid % M;
        g = bid / M;
    }
    WgradBlock r;
    r.g = g;
    r.ci0 = (m % ncgi) * 32;
    r.co0 = (m / ncgi) * (32 * WN);
    return r;
}

// NTAPS = 9: 3x3 (pad 1); NTAPS = 1: 1x1 (pad 0; the f16x3 mode also runs the Bottleneck / downsample 1x1 convolutions here)
// VAR >= 0: compile-time variant of the staging switches (bit 0 fused input BatchNorm + ReLU on X, bit 1 dY is an f16 pair tensor): the
// per-item branches of the staging pass fold away; VAR < 0: read from the argument block
template <int NTAPS, int WK, int WN, int SPLIT, int NX, int VAR = -1>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    constexpr int KS = NTAPS == 9 ? 3 : 1;
    constexpr int PX = 192;                               // X bytes per staged pixel: [3 terms][32 ch bf16]
    constexpr int PD = WN * 192 + (WN > 1 ? 64 : 0);      // dY bytes per pixel: [WN][3 terms][32 ch bf16] (+ pad)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int wk = wave / WN, wn = wave % WN;
    const WgradBlock wb = wgrad_block<WN>(a);
    const int ci0 = wb.ci0, co0 = wb.co0;
    const int halo_pix = a.halo_h * a.halo_w;
    const int npix = a.TH * a.TW;
    const int nsteps_all = (npix + 15) >> 4;              // k-steps of 16 pixels; the padding rows of dY are zero
    const int npix_pad = nsteps_all << 4;
    unsigned char* xs = ldsb;
    unsigned char* dys = ldsb + halo_pix * PX;
    const int flags = VAR >= 0 ? (((VAR & 1) ? SPK_IN_AFFINE_RELU : 0) | ((VAR & 2) ? SPK_DY_PRESPLIT : 0)) : a.flags;
    constexpr int NTERM = SPLIT == 3 ? 2 : 3;
    float sig_x = 1.f, sig_d = 1.f;
    if constexpr (SPLIT == 3) {
        sig_x = a.x_amax ? spk_sigma_from_amax_bits(*a.x_amax) : SPK_F16_ACT_SIGMA;
        if (a.dy_amax) sig_d = spk_sigma_from_amax_bits(*a.dy_amax);
    }

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int quad = tid & 7;

    // region pipeline as in conv_wgrad.hip: next region's global loads fly during this region's MFMAs
    constexpr int ND = WGRAD_ND;   // NX (X-halo float4 per thread) is a template parameter: 4 when the halo fits 128 pixels
    constexpr int QPP = WN * 8;
    constexpr int PSTEP = 256 / QPP;
    const int cq = tid % QPP;
    f32x4 px[NX], pd[ND];
    unsigned inx = 0, ind = 0;

    // Address arithmetic of the prefetch: the image base is a scalar (64-bit, SALU), the per-thread part a 32-bit byte offset
    // inside the image; a slot outside the image reads the region's first output pixel instead (always inside) and is
    // zeroed when published.  (The clamped 64-bit form cost ~26 VALU instructions per load, a quarter of the staging.)
    const unsigned x_row = (unsigned)a.IW * a.Cin * 4u, x_px = (unsigned)a.Cin * 4u;
    const unsigned d_row = (unsigned)a.OW * a.Cout * 4u, d_px = (unsigned)a.Cout * 4u;
    const unsigned x_c = (unsigned)(ci0 + quad * 4) * 4u, d_c = (unsigned)(co0 + cq * 4) * 4u;
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
        for (int u = 0; u < NX; ++u) {
            const int p = (tid >> 3) + 32 * u;
            const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
            const int hx = p - hy * a.halo_w;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW && p < halo_pix;
            if (ok) inx |= 1u << u;
            const unsigned off = ok ? (unsigned)iy * x_row + (unsigned)ix * x_px + x_c : x_safe;
#ifdef WG_ABL_NO_PREFETCH
            px[u] = (f32x4){(float)off, 0.f, 1.f, 2.f};
#else
            px[u] = wgrad_ld<2>(xb + off);
#endif
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
#ifdef WG_ABL_NO_PREFETCH
            pd[u] = (f32x4){(float)off, 0.f, 1.f, 2.f};
#else
            pd[u] = wgrad_ld<1>(db + off);
#endif
        }
    };
    auto publish = [&]() {      // registers -> three bf16 terms in LDS (zero outside the image / tile; fused BN+ReLU on X)
#ifdef WG_ABL_NO_PUBLISH
        if (a.flags != 12345) return;
#endif
        // the BN coefficients are re-read here (L1-resident) rather than held in registers across the MFMA loop
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (flags & SPK_IN_AFFINE_RELU) {
            sc = *(const f32x4*)(a.in_scale + ci0 + quad * 4);
            sh = *(const f32x4*)(a.in_shift + ci0 + quad * 4);
        }
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
            if (p < halo_pix) {
                uint2* dst = (uint2*)(xs + p * PX) + quad;
                if constexpr (SPLIT == 3) {
                    uint2 t0, t1;
                    split2h(w, sig_x, t0, t1);
                    dst[0] = t0;
                    dst[8] = t1;
                } else {
                    uint2 t0, t1, t2;
                    split3(w, t0, t1, t2);
                    dst[0] = t0;
                    dst[8] = t1;
                    dst[16] = t2;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int p = tid / QPP + PSTEP * u;
            const f32x4 w = ((ind >> u) & 1) ? pd[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p < npix_pad) {
                uint2* dst = (uint2*)(dys + p * PD + (cq >> 3) * 192) + (cq & 7);
                if constexpr (SPLIT == 3) {
                    uint2 t0, t1;
                    spk_terms(w, sig_d, (flags & SPK_DY_PRESPLIT) != 0, t0, t1);     // f16 pair tensor: plain copy
                    dst[0] = t0;
                    dst[8] = t1;
                } else {
                    uint2 t0, t1, t2;
                    split3(w, t0, t1, t2);
                    dst[0] = t0;
                    dst[8] = t1;
                    dst[16] = t2;
                }
            }
        }
    };

    // per-lane roles in the transposing reads: 16-lane group g16 = (channel half, k half); within it lane 4q+p addresses
    // pixel row q, channels 4p..4p+3 of the group's 16 channels
    const int g16 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const int col_off = (g16 & 1) * 32 + p4 * 8;

    auto mma = [&](f32x16& c, const s16x8* af, const s16x8* bf) {
#ifdef WG_ABL_NO_MMA
        asm volatile("" : "+v"(c) : "v"(af[0]), "v"(bf[0]), "v"(af[1]), "v"(bf[1]));
        return;
#endif
#pragma unroll
        for (int sum = (SPLIT == 9 ? 4 : (SPLIT == 6 ? 2 : 1)); sum >= 0; --sum)
#pragma unroll
            for (int sa = 0; sa < NTERM; ++sa) {
                const int sb = sum - sa;
                if (sb < 0 || sb >= NTERM) continue;
                if constexpr (SPLIT == 3)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[sa]), __builtin_bit_cast(f16x8, bf[sb]), c,
                                                               0, 0, 0);
                else
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[sa]), __builtin_bit_cast(bf16x8, bf[sb]), c,
                                                                0, 0, 0);
            }
    };

    int region = wb.g;
    if (region < a.nregions) prefetch(region);
    for (; region < a.nregions; region += a.nsplit) {
        __syncthreads();   // previous region fully consumed
        publish();
        __syncthreads();
        if (region + a.nsplit < a.nregions) prefetch(region + a.nsplit);

#ifdef WG_ABL_NO_KLOOP
        for (int j = wk; j < (a.flags == 12345 ? nsteps_all : 0); j += WK) {
#else
        for (int j = wk; j < nsteps_all; j += WK) {
#endif
            int xa[2], da[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int pix = j * 16 + 8 * h + 4 * blk + q;
                const int pc = pix < npix ? pix : npix - 1;      // padding rows: any valid X address (their dY is zero)
                const int ly = (int)__umulhi((unsigned)pc, a.tw_magic);
                const int lx = pc - ly * a.TW;
                xa[blk] = ((ly * a.S) * a.halo_w + lx * a.S) * PX + col_off;
                da[blk] = pix * PD + wn * 192 + col_off;
            }
            s16x8 bf[NTERM];
#pragma unroll
            for (int s = 0; s < NTERM; ++s) bf[s] = tr_read8(dys + da[0] + s * 64, dys + da[1] + s * 64);
            s16x8 a0[NTERM], a1[NTERM];
            auto load_a = [&](s16x8* af, int t) {
                const int toff = ((t / KS) * a.halo_w + (t % KS)) * PX;
#pragma unroll
                for (int s = 0; s < NTERM; ++s) af[s] = tr_read8(xs + xa[0] + toff + s * 64, xs + xa[1] + toff + s * 64);
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

    // fold the WK pixel-splits into wk == 0 through LDS (fixed order -> deterministic), then one slab per block
    const int r = lane & 31;
    if (WK > 1) {
        float* red = (float*)ldsb;  // [WN][NTAPS][16][64]
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
        float* slab = a.partial + (size_t)wb.g * NTAPS * a.Cin * a.Cout;
        const float inv_x = 1.f / sig_x, inv_d = 1.f / sig_d;           // powers of two: exact; applied one after the other
                                                                         // (their product may leave the fp32 range)
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                slab[((size_t)t * a.Cin + ci0 + row) * a.Cout + co0 + wn * 32 + r] = SPLIT == 3 ? acc[t][e] * inv_x * inv_d : acc[t][e];
            }
    }
}


#ifdef SPK_EXPERIMENTAL
// ---- producer / consumer form (f16x3 operands) ---------------------------------------------------------------------------
// Measured on MI355X (tools/wg_abl.sh, 128-channel 20x75 layer): the kernel above takes 0.43 ms, of which the K loop alone
// (fragment reads + MFMAs) is 0.28 ms and the staging alone (address arithmetic, loads, fp16 split, LDS writes, barriers)
// 0.26 ms - each wave does one after the other and the two resident blocks of a CU drift into the same phase, so the
// matrix pipe is busy 39 % of the time.  Here a block has eight waves: waves 0-3 only run the K loop, waves 4-7 only stage
// the NEXT region into the other half of a two-slot LDS ring; one barrier per region hands a slot over in both directions
// (slot i & 1 is published before barrier i and read after it; the producers overwrite it after barrier i + 1, which the
// consumers reach only when they are done reading).  Every SIMD holds one wave of each kind, so the VALU / memory work of
// the staging runs under the MFMAs of the same SIMD.  Same MFMA order per accumulator as the kernel above: bit-identical slabs.
template <int NTAPS, int WK, int WN, int NX>
__global__ __launch_bounds__(512, 1) void conv_wgrad_ws_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    constexpr int KS = NTAPS == 9 ? 3 : 1;
    constexpr int PX = 192;
    constexpr int PD = WN * 192 + (WN > 1 ? 64 : 0);
    const int tid = threadIdx.x & 255, lane = tid & 63;       // index within the role group (consumers / producers)
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave8 >= 4;
    const int wave = wave8 & 3;
    const int h = lane >> 5;
    const int wk = wave / WN, wn = wave % WN;
    const WgradBlock wb = wgrad_block<WN>(a);
    const int ci0 = wb.ci0, co0 = wb.co0;
    const int halo_pix = a.halo_h * a.halo_w;
    const int npix = a.TH * a.TW;
    const int nsteps_all = (npix + 15) >> 4;
    const int npix_pad = nsteps_all << 4;
    const int slot_bytes = halo_pix * PX + npix_pad * PD;
    const int flags = a.flags;
    const float sig_x = a.x_amax ? spk_sigma_from_amax_bits(*a.x_amax) : SPK_F16_ACT_SIGMA;
    const float sig_d = a.dy_amax ? spk_sigma_from_amax_bits(*a.dy_amax) : 1.f;
    const int nmine = (a.nregions - wb.g + a.nsplit - 1) / a.nsplit;     // regions of this block (>= 1)

    f32x16 acc[NTAPS];
    if (producer) {
        // ---- producers: global -> registers -> (BN + ReLU, zero padding, fp16 split) -> LDS slot ----
        constexpr int ND = WGRAD_ND;
        constexpr int QPP = WN * 8;
        constexpr int PSTEP = 256 / QPP;
        const int quad = tid & 7, cq = tid % QPP;
        // two register sets: the loads of region i + 2 are issued as soon as region i has been published, so a region's
        // global-memory latency has two region times to pass (one is not enough: the producers set the pace then)
        f32x4 pxA[NX], pdA[ND], pxB[NX], pdB[ND];
        unsigned inxA = 0, indA = 0, inxB = 0, indB = 0;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (flags & SPK_IN_AFFINE_RELU) {
            sc = *(const f32x4*)(a.in_scale + ci0 + quad * 4);
            sh = *(const f32x4*)(a.in_shift + ci0 + quad * 4);
        }
        auto prefetch = [&](f32x4 (&px)[NX], f32x4 (&pd)[ND], unsigned& inx, unsigned& ind, int region) {
#ifdef WGWS_ABL_NO_L
            if (a.flags != 12345) return;
#endif
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
                px[u] = wgrad_ld<2>(a.x + (size_t)((b * a.IH + cy) * a.IW + cx) * a.Cin + ci0 + quad * 4);
            }
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const int p0 = tid / QPP + PSTEP * u;
                const int p = p0 < npix ? p0 : npix - 1;
                const int ly = (int)__umulhi((unsigned)p, a.tw_magic);
                const int lx = p - ly * a.TW;
                const int oy = oy0 + ly, ox = ox0 + lx;
                if (p0 < npix && oy < a.OH && ox < a.OW) ind |= 1u << u;
                const int cy = min(oy, a.OH - 1), cx = min(ox, a.OW - 1);
                pd[u] = wgrad_ld<1>(a.dy + (size_t)((b * a.OH + cy) * a.OW + cx) * a.Cout + co0 + cq * 4);
            }
        };
        auto publish = [&](const f32x4 (&px)[NX], const f32x4 (&pd)[ND], unsigned inx, unsigned ind, unsigned char* xs) {
#ifdef WGWS_ABL_NO_P
            if (a.flags != 12345) return;
#endif
            unsigned char* dys = xs + halo_pix * PX;
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
                if (p < halo_pix) {
                    uint2* dst = (uint2*)(xs + p * PX) + quad;
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
                    split2h(w, sig_d, t0, t1);
                    dst[0] = t0;
                    dst[8] = t1;
                }
            }
        };
        // region of this block's i-th turn: g + i * nsplit.  Barrier i: slot i & 1 is complete and the consumers
        // are done with slot (i + 1) & 1.
        const int r0 = wb.g, rs = a.nsplit;
        prefetch(pxA, pdA, inxA, indA, r0);
        if (nmine > 1) prefetch(pxB, pdB, inxB, indB, r0 + rs);
        for (int i = 0; i < nmine; i += 2) {
            publish(pxA, pdA, inxA, indA, ldsb);
            if (i + 2 < nmine) prefetch(pxA, pdA, inxA, indA, r0 + (i + 2) * rs);
            __syncthreads();
            if (i + 1 < nmine) {
                publish(pxB, pdB, inxB, indB, ldsb + slot_bytes);
                if (i + 3 < nmine) prefetch(pxB, pdB, inxB, indB, r0 + (i + 3) * rs);
                __syncthreads();
            }
        }
    } else {
        // ---- consumers: LDS slot -> transposing fragment reads -> MFMA ----
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        const int g16 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
        const int col_off = (g16 & 1) * 32 + p4 * 8;
        auto mma = [&](f32x16& c, const s16x8* af, const s16x8* bf) {
            // h1*g2 + h2*g1 first, h1*g1 last (the order of conv_wgrad_split_kernel<.., 3, ..>)
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[1]), __builtin_bit_cast(f16x8, bf[0]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[0]), c, 0, 0, 0);
        };
        for (int i = 0; i < nmine; ++i) {
            __syncthreads();       // barrier i
            const unsigned char* xs = ldsb + (i & 1) * slot_bytes;
            const unsigned char* dys = xs + halo_pix * PX;
#ifdef WGWS_ABL_NO_K
            for (int j = wk; j < (a.flags == 12345 ? nsteps_all : 0); j += WK) {
#else
            for (int j = wk; j < nsteps_all; j += WK) {
#endif
                int xa[2], da[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const int pix = j * 16 + 8 * h + 4 * blk + q;
                    const int pc = pix < npix ? pix : npix - 1;
                    const int ly = (int)__umulhi((unsigned)pc, a.tw_magic);
                    const int lx = pc - ly * a.TW;
                    xa[blk] = ((ly * a.S) * a.halo_w + lx * a.S) * PX + col_off;
                    da[blk] = pix * PD + wn * 192 + col_off;
                }
                s16x8 bf[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) bf[s] = tr_read8(dys + da[0] + s * 64, dys + da[1] + s * 64);
                s16x8 a0[2], a1[2];
                auto load_a = [&](s16x8* af, int t) {
                    const int toff = ((t / KS) * a.halo_w + (t % KS)) * PX;
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s] = tr_read8(xs + xa[0] + toff + s * 64, xs + xa[1] + toff + s * 64);
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

    // fold the WK pixel-splits into wk == 0 through LDS (fixed order), then one slab per block; every wave of the block
    // takes part in the barriers
    const int r = lane & 31;
    if (WK > 1) {
        float* red = (float*)ldsb;  // [WN][NTAPS][16][64]
#pragma unroll 1
        for (int src = 1; src < WK; ++src) {
            __syncthreads();
            if (!producer && wk == src) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((wn * NTAPS + t) * 16 + e) * 64 + lane] = acc[t][e];
            }
            __syncthreads();
            if (!producer && wk == 0) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] += red[((wn * NTAPS + t) * 16 + e) * 64 + lane];
            }
        }
    }
    if (!producer && wk == 0) {
        float* slab = a.partial + (size_t)wb.g * NTAPS * a.Cin * a.Cout;
        const float inv_x = 1.f / sig_x, inv_d = 1.f / sig_d;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                slab[((size_t)t * a.Cin + ci0 + row) * a.Cout + co0 + wn * 32 + r] = acc[t][e] * inv_x * inv_d;
            }
    }
}

// LDS bytes of the producer / consumer kernel for a tile (0: does not fit)
template <int NTAPS, int WK, int WN>
static int launch_ws(const WgradArgs& a, hipStream_t st) {
    const int npix_pad = ((a.TH * a.TW + 15) >> 4) << 4;
    constexpr int PD = WN * 192 + (WN > 1 ? 64 : 0);
    size_t lds_bytes = 2 * ((size_t)a.halo_h * a.halo_w * 192 + (size_t)npix_pad * PD);
    const size_t red_bytes = (WK > 1) ? (size_t)WN * NTAPS * 16 * 64 * sizeof(float) : 0;
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_wgrad(ws): tile %dx%d needs %zu B of LDS", a.TH, a.TW, lds_bytes);
    dim3 grid(a.nsplit * (a.Cin / 32) * (a.Cout / (32 * WN)));       // see wgrad_block
    if (a.halo_h * a.halo_w <= 32 * 4) hipLaunchKernelGGL((conv_wgrad_ws_kernel<NTAPS, WK, WN, 4>), grid, dim3(512), lds_bytes, st, a);
    else hipLaunchKernelGGL((conv_wgrad_ws_kernel<NTAPS, WK, WN, WGRAD_NX>), grid, dim3(512), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_wgrad(ws)");
    return 0;
}

int spk_launch_wgrad_ws(const WgradArgs& a, int WN, hipStream_t st) {
    SPK_REQUIRE(a.KW == 3, "spk_conv_wgrad(ws): 3x3 only");
    if (WN == 1) return launch_ws<9, 4, 1>(a, st);
    if (WN == 2) return launch_ws<9, 2, 2>(a, st);
    return launch_ws<9, 1, 4>(a, st);
}

#endif   // SPK_EXPERIMENTAL

template <int NTAPS, int WK, int WN>
static int launch_one(const WgradArgs& a, int split, hipStream_t st) {
    const int npix_pad = ((a.TH * a.TW + 15) >> 4) << 4;
    constexpr int PD = WN * 192 + (WN > 1 ? 64 : 0);
    size_t lds_bytes = (size_t)a.halo_h * a.halo_w * 192 + (size_t)npix_pad * PD;
    const size_t red_bytes = (WK > 1) ? (size_t)WN * NTAPS * 16 * 64 * sizeof(float) : 0;
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_wgrad(split): tile %dx%d needs %zu B of LDS", a.TH, a.TW, lds_bytes);
    dim3 grid(a.nsplit * (a.Cin / 32) * (a.Cout / (32 * WN)));       // see wgrad_block
    const bool small = a.halo_h * a.halo_w <= 32 * 4;
    if (split == 3) {
        const int var = ((a.flags & SPK_IN_AFFINE_RELU) ? 1 : 0) | ((a.flags & SPK_DY_PRESPLIT) ? 2 : 0);
        // the layer-1 weight gradients of the training step (3x3, 32 channels: WK 4, WN 1, the large prefetch window) with pair dY
        if constexpr (NTAPS == 9 && WK == 4 && WN == 1) {
            if (!small && var == 3) {
                hipLaunchKernelGGL((conv_wgrad_split_kernel<NTAPS, WK, WN, 3, WGRAD_NX, 3>), grid, dim3(256), lds_bytes, st, a);
                SPK_LAUNCH_CHECK("spk_conv_wgrad(split)");
                return 0;
            }
            if (!small && var == 2) {
                hipLaunchKernelGGL((conv_wgrad_split_kernel<NTAPS, WK, WN, 3, WGRAD_NX, 2>), grid, dim3(256), lds_bytes, st, a);
                SPK_LAUNCH_CHECK("spk_conv_wgrad(split)");
                return 0;
            }
        }
        if (small) hipLaunchKernelGGL((conv_wgrad_split_kernel<NTAPS, WK, WN, 3, 4>), grid, dim3(256), lds_bytes, st, a);
        else hipLaunchKernelGGL((conv_wgrad_split_kernel<NTAPS, WK, WN, 3, WGRAD_NX>), grid, dim3(256), lds_bytes, st, a);
    } else if constexpr (NTAPS == 9) {
        if (split == 6) {
            if (small) hipLaunchKernelGGL((conv_wgrad_split_kernel<9, WK, WN, 6, 4>), grid, dim3(256), lds_bytes, st, a);
            else hipLaunchKernelGGL((conv_wgrad_split_kernel<9, WK, WN, 6, WGRAD_NX>), grid, dim3(256), lds_bytes, st, a);
        } else {
            if (small) hipLaunchKernelGGL((conv_wgrad_split_kernel<9, WK, WN, 9, 4>), grid, dim3(256), lds_bytes, st, a);
            else hipLaunchKernelGGL((conv_wgrad_split_kernel<9, WK, WN, 9, WGRAD_NX>), grid, dim3(256), lds_bytes, st, a);
        }
    } else {
        spk_set_error("spk_conv_wgrad(split): 1x1 weight gradients exist in the f16x3 mode only");
        return -1;
    }
    SPK_LAUNCH_CHECK("spk_conv_wgrad(split)");
    return 0;
}

int spk_launch_wgrad_split(const WgradArgs& a, int WN, int split, hipStream_t st) {
    if (a.KW == 1) {
        if (WN == 1) return launch_one<1, 4, 1>(a, split, st);
        if (WN == 2) return launch_one<1, 2, 2>(a, split, st);
        return launch_one<1, 1, 4>(a, split, st);
    }
    if (WN == 1) return launch_one<9, 4, 1>(a, split, st);
    if (WN == 2) return launch_one<9, 2, 2>(a, split, st);
    return launch_one<9, 1, 4>(a, split, st);
}
