#pragma once
#include "spk_common.h"
#include <type_traits>

#ifndef STAGE_U
#define STAGE_U 4   // staging loads in flight per thread
#endif
// CONV_NT bit 0: non-temporal stores of the convolution output and side outputs; bit 1: non-temporal loads of the epilogue's read
// streams (shortcut, raw tensor of the BatchNorm-backward statistics); bit 2: of the fused BatchNorm-backward staging streams
#ifndef CONV_NT
#define CONV_NT 3
#endif
static __device__ __forceinline__ void conv_st(float* p, f32x4 v) {
    if constexpr ((CONV_NT & 1) != 0) __builtin_nontemporal_store(v, (f32x4*)p);
    else *(f32x4*)p = v;
}
template <int BIT>
static __device__ __forceinline__ f32x4 conv_ld(const float* p) {
    if constexpr ((CONV_NT & BIT) != 0) return __builtin_nontemporal_load((const f32x4*)p);
    else return *(const f32x4*)p;
}
#ifndef STAGE_U2
#define STAGE_U2 2  // pixels in flight per thread in the fused BatchNorm-backward staging (three loads each)
#endif

// SPLIT == 0: operands stay fp32 (v_mfma_f32_32x32x2_f32).  SPLIT == 6 / 9: every fp32 operand is split into three bf16
// terms (x = x1 + x2 + x3 exactly: 3 x 8 significand bits) while it is staged (activations) or packed (weights), and the
// product is formed from the 6 (or all 9) cross terms of weight >= 2^-16 (2^-24) on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation: measured error against fp64 equals that of the native fp32 matrix instruction
// (tools/probe/split_probe.hip), at 16/6 of its rate.
#define SPK_SPLIT_CK 16   // channels per staged plane of the bf16-split kernels (32 = full 128-byte lines per pass was
                          // measured 5 % slower: 208-byte LDS pixels force smaller tiles)
// SPLIT == 3: fp16 two-term operands (spk_common.h, "f16x3"): same structure with two terms per value, three products on
// v_mfma_f32_32x32x16_f16, a power-of-two input scale (static for activations, from the tensor's absmax for gradients)
// and the accumulators scaled back in the epilogue.
template <int SPLIT>
struct ConvCfg {
    static constexpr int NTERM = SPLIT == 3 ? 2 : 3;       // operand terms (split kernels)
    static constexpr int MAXSUM = SPLIT == 9 ? 4 : (SPLIT == 6 ? 2 : 1);   // products (sa, sb) with sa + sb <= MAXSUM
    static constexpr int CK = SPLIT ? SPK_SPLIT_CK : 32;   // channels per staged plane
    static constexpr int TPP = CK / 4;           // threads per staged pixel (one float4 of channels each)
    static constexpr int PPP = 256 / TPP;        // pixels per staging pass of the block
    static constexpr int KG = CK / 16;           // split: 16-channel MFMA groups per plane
    // LDS pixel pitch in 16-byte units: fp32 [32 ch + 4 pad] = 144 B; split [3 terms][CK ch bf16] + 16 pad = 112 / 208 B.
    // All are odd multiples of 16 B, so the 8 lanes of a ds_read_b128 phase (consecutive pixels) hit distinct bank groups.
    static constexpr int LP4 = SPLIT ? (NTERM * CK * 2 + 16) / 16 : 9;
};

struct ConvArgs {
    const float* in;
    const float* wpk;
    float* out;
    const float* in_scale;
    const float* in_shift;
    const float* epi_scale;
    const float* epi_shift;
    const float* epi_add;
    const float* in_raw;   // SPK_IN_BNBWD: the staged input is BatchNorm-backward(in): raw conv output of that BatchNorm,
    const float* in_act;   //   its activated output (mask = act > 0) or NULL (mask = raw*scale+shift > 0),
    const float* in_bn4;   //   [4][Cin]: mean, invstd, scale, shift,
    const float* in_coef;  //   [3][Cin]: gamma*invstd, mean(dz), mean(dz*xhat)  (spk_bn_bwd_finalize)
    const unsigned* in_mask;   //   alternative to in_act: its sign bits, [pixel][Cin/32] words (spk_bn_apply mask_out)
    const unsigned* bn_mask;   // SPK_EPI_BNBWD: alternative to bn_act, [pixel][Cout/32] words
    const unsigned* add_mask;  // SPK_EPI_ADD: add only where the bit is set (epi_add = gradient wrt a ReLU output, bits = its
                               //   sign mask: the shortcut gradient dz = dout*[out > 0] formed here instead of stored and re-read)
    float* side_draw;      //   optional side outputs of the tile's own pixels: the transformed value (gradient wrt the raw
    float* side_dz;        //   conv output, consumed by the weight gradient) and dz = in*mask (the shortcut gradient)
    const float* bn_raw;   // SPK_EPI_BNBWD: raw conv output of the BatchNorm whose backward statistics are reduced here
    const float* bn_act;   //   activated output (mask = act > 0) or NULL (mask = raw*scale+shift > 0)
    const float* bn4;      //   [4][Cout]: mean, invstd, scale, shift of that BatchNorm
    float* stats;
    int B, IH, IW, Cin;    // IH, IW: logical input grid (= physical / ips, rounded up)
    int IHp, IWp, ips;     // physical input dims and pixel stride: logical pixel (y,x) lives at (y*ips, x*ips)
    int OH, OW;            // logical output grid
    int OHf, OWf, Cout;    // physical output tensor
    int IS, OS, ooy, oox;
    int TH, TW, tiles_y, tiles_x;
    int halo_h, halo_w, min_dy, min_dx;
    unsigned halo_w_magic;   // ceil(2^32 / halo_w): p / halo_w == umulhi(p, magic) for p * halo_w < 2^32
    int ntaps, ncg, nblocks, flags;
    int tap_off[9];   // LDS offset (16-byte units) of the (tap, channel plane) inside the staged tile
    int tap_w[9];     // weight tap index
    int tap_g[9];     // weight K-group offset of the channel plane (fp32: 4 groups of 8 channels per plane; split: 1 of 16)
    // f16x3 operand mode (SPLIT == 3)
    const unsigned* in_amax;   // float bits of the staged tensor's absmax (or an upper estimate); NULL: static in_sigma
    float in_sigma;            // static input scale when in_amax is NULL
    const unsigned* w_amax;    // header of the fp16-split packed weights: float bits of max|w| -> the weight scale
    unsigned* out_amax;        // optional: atomicMax of |stored output| (float bits) - the next consumer's in_amax
    unsigned* side_amax;       // optional (IN_BNBWD): atomicMax of |side_draw| - the weight gradient's dY scale
    int tap_boff[9];  // wave-specialised kernel: float offset of (tap, channel plane 0) in the bf16-split packed weights
    int kc;           // channel planes (of 32) staged per barrier: > 1 only for single-tap (1x1) convolutions, whose K loop
                      // per 32-channel chunk is too short to amortise a staging phase
};

// PIPE (f16x3, nine taps, one plane per chunk): the staging of chunk ch + 1 is woven into the K loop of chunk ch IN THE SAME
// WAVE - after the matrix instructions of tap t the wave converts and writes staging item t of the next chunk into the other
// half of a two-slot LDS tile (its global load was issued PIPE_D taps earlier), one barrier per chunk.  Why in the same wave:
// on gfx950 the VALU instructions of one wave do not run under the MFMAs of ANOTHER wave of the SIMD (tools/probe/
// issue_probe.hip: MFMA-only wave + VALU-only wave = the sum of their times, s_setprio or not), only independent VALU work
// that follows an MFMA in the same instruction stream does; staging done as a separate phase - or by separate producer
// waves - is therefore paid in full on top of the matrix time.
#define PIPE_D 3
#ifndef PIPE16_D
#define PIPE16_D 3      // staging items in flight in the 16x16x32 form, pair input (an item is consumed PIPE16_D half steps after
#endif                  // its load; 4 and 5 measured equal: 51.18-51.28 / 51.20 against 51.02-51.21 ms per step)
#ifndef PIPE16_D_CONV
#define PIPE16_D_CONV 3 // the same for the form that converts while staging (fp32 input); 4 is slower (51.7 ms: registers)
#endif
// (M16_NO_STAGE / M16_NO_FINISH / M16_NO_ISSUE: diagnostic builds of the 16x16x32 form without the in-loop staging / without its
//  conversion + LDS write / without its global loads - wrong results by construction; profiles/r03_m16_ablation.log)
#ifndef PIPE_VPM
#define PIPE_VPM 6      // VALU instructions scheduled behind every MFMA of a pipelined tap (2 / 4 / 6 / 8 measured: 57.08 / 57.29 / 56.94 / 56.9 ms per step)
#endif
// PRE (PIPE only): the input is an f16 pair tensor (spk_common.h): staging item = one 16-byte load + two 8-byte LDS writes, no
// conversion - the data gradients whose BatchNorm backward ran as a separate pass (spk_bn_bwd_apply with pair output).
// The non-pipelined f16x3 kernels take pair tensors through the run-time flag SPK_IN_PRESPLIT.
// M16 (PIPE only, plain or pair input): the matrix instruction is v_mfma_f32_16x16x32_f16 instead of 32x32x16 - the same
// multiply-adds per cycle, but in these power-limited loops the chip holds a higher clock on it (tools/probe/shape_probe.hip:
// 1.15 x at one wave per SIMD, 1.25 x at two, in a loop of this kernel's shape).  Its K step is 32 deep; a staged plane stays 16
// channels (two 32-channel tiles would not leave two blocks per CU), so a K step PAIRS TWO TAPS of the plane: lanes 0-31 carry the
// 16 channels of tap t, lanes 32-63 those of tap t'.  Nine taps are odd: two planes are walked together - (0,1)(2,3)(4,5)(6,7) of
// the even plane, then (8 | 0 of the odd plane), then (1,2)(3,4)(5,6)(7,8) - nine steps per plane pair.  The step that straddles
// the planes reads both LDS slots, so the even slot is refilled only after it (one more barrier per plane pair).  The packed
// weights keep their order: a lane picks its tap's 16-byte piece by address.  Inside a pixel the LDS image is [8-channel
// half][term] instead of [term][half], and the 16 rows of a tile are a permutation of 16 consecutive pixels (even pixels on rows
// 4-11): with both, the four 16-lane groups of a ds_read_b128 are conflict-free (the natural order is 2-way).
#ifdef CONV_STAMPS
// diagnostic build (tools/variant.sh ... -DCONV_STAMPS): shader-clock stamps of the phases of a block of the non-pipelined split
// kernel, written by lane 0 of wave 0 to a device array that no other code reads (spk_debug_stamps copies it out)
#define CONV_STAMP_BLOCKS 4096
static __device__ unsigned long long g_conv_stamps[CONV_STAMP_BLOCKS][16];      // (one copy per translation unit)
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < CONV_STAMP_BLOCKS) g_conv_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
// FL >= 0: the launch's flag word is a compile-time constant (the SPK_* bits the kernel sees, plus SPK_FL_ADDMASK / SPK_FL_BNMASK:
// a sign mask accompanies the shortcut add / the BatchNorm-backward statistics; no activation tensor as the statistics' mask) - the
// epilogue and staging branches fold away.  FL < 0: everything is read from the argument block (any combination).
#define SPK_FL_ADDMASK (1 << 20)
#define SPK_FL_BNMASK (1 << 21)
#define SPK_FL_INMASK (1 << 22)      // fused input BatchNorm backward: its ReLU mask as sign bits (else recomputed from the raw tensor); no side_dz
template <int MT, int NT, bool BNBWD, int SPLIT, bool PIPE, bool BITS = false, bool PRE = false, bool M16 = false, int FL = -1>
static __device__ __forceinline__ void conv_body(const ConvArgs& a) {
    using Cfg = ConvCfg<SPLIT>;
    constexpr int CK = Cfg::CK, TPP = Cfg::TPP, PPP = Cfg::PPP, LP4 = Cfg::LP4, NTERM = Cfg::NTERM;
    static_assert(!M16 || (PIPE && !BNBWD && SPLIT == 3), "M16: the pipelined f16x3 kernel with a plain or pair input");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // M16: row of the 16 x 16 tile -> pixel of the 16-pixel group (rows 4..11 <-> even pixels), and the inverse for the epilogue
    auto pix_of_row16 = [](int row) {
        const int q = (row + 12) & 15;
        return ((q & 7) << 1) | (q >> 3);
    };
    // f16x3: input scale (a power of two) and the factor that takes the accumulators back to fp32 units
    // (two factors: the product of the scales may leave the fp32 range, each reciprocal is an exact power of two)
    float sig = 1.f, inv_sig = 1.f, inv_wsig = 1.f;
    if constexpr (SPLIT == 3) {
        sig = a.in_amax ? spk_sigma_from_amax_bits(*a.in_amax) : a.in_sigma;
        inv_sig = 1.f / sig;
        inv_wsig = 1.f / spk_sigma_from_amax_bits(*a.w_amax);
    }
    float side_mx = 0.f;

    // XCD-aware remap (bijective for any grid size): blocks that are adjacent in the logical
    // order (same pixel region, next cout group; then the neighbouring region) share an XCD's L2.
    int bid = blockIdx.x;
    {
        const int n = a.nblocks, q = n >> 3, rr = n & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
    }
    const int cg = bid % a.ncg;
    const int ptile = bid / a.ncg;
    int pt = ptile;
    const int tx = pt % a.tiles_x;
    pt /= a.tiles_x;
    const int ty = pt % a.tiles_y;
    const int b = pt / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int iy0 = oy0 * a.IS + a.min_dy, ix0 = ox0 * a.IS + a.min_dx;
    // Staging addresses: image bases are scalars (64-bit, SALU), the per-thread part is a 32-bit element offset inside image b;
    // a halo slot outside the image reads a pixel of the halo that is inside (and is zeroed before it reaches LDS).
    const int sy = min(max(iy0, 0), a.IH - 1), sx = min(max(ix0, 0), a.IW - 1);
    const unsigned pi_safe = (unsigned)(sy * a.IW + sx), pi_safe_p = (unsigned)((sy * a.IWp + sx) * a.ips);
    const float* img_in_p = a.in + (size_t)b * a.IHp * a.IWp * a.Cin;
    const size_t img_el = (size_t)b * a.IH * a.IW * a.Cin;
    const float* img_in = a.in + img_el;
    const float* img_raw = BNBWD ? a.in_raw + img_el : nullptr;
    const float* img_act = (BNBWD && a.in_act) ? a.in_act + img_el : nullptr;
    const unsigned* img_mask = (BNBWD && a.in_mask) ? a.in_mask + (size_t)b * a.IH * a.IW * (a.Cin >> 5) : nullptr;
    float* img_draw = BNBWD ? a.side_draw + img_el : nullptr;
    float* img_dz = (BNBWD && a.side_dz) ? a.side_dz + img_el : nullptr;
    const int npix_tile = a.TH * a.TW;
    const int n0 = cg * NT * 32;
    const int flags = FL >= 0 ? (FL & 0xFFFFF) : a.flags;
    const bool has_add_mask = FL >= 0 ? (FL & SPK_FL_ADDMASK) != 0 : a.add_mask != nullptr;
    const bool has_bn_mask = FL >= 0 ? (FL & SPK_FL_BNMASK) != 0 : a.bn_mask != nullptr;
    const bool has_bn_act = FL >= 0 ? false : a.bn_act != nullptr;
    const bool has_in_mask = FL >= 0 ? (FL & SPK_FL_INMASK) != 0 : a.in_mask != nullptr;
    const bool has_in_act = FL >= 0 ? false : a.in_act != nullptr;
    const bool has_side_dz = FL >= 0 ? false : a.side_dz != nullptr;

    int lbase[M16 ? 2 * MT : MT], obase[MT], opix[MT];      // opix: output pixel index (the masks are addressed per pixel)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int q = (wave * MT + i) * 32 + r;
        bool v = q < npix_tile;
        const int qq = v ? q : 0;
        const int ly = qq / a.TW, lx = qq - ly * a.TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        v = v && oy < a.OH && ox < a.OW;
        if constexpr (!M16) lbase[i] = ((ly * a.IS) * a.halo_w + lx * a.IS) * LP4 + h;   // 16-byte units
        opix[i] = (b * a.OHf + oy * a.OS + a.ooy) * a.OWf + ox * a.OS + a.oox;
        obase[i] = v ? opix[i] * a.Cout + n0 : -1;
    }
    if constexpr (M16) {
        // A fragment of a 16-row tile: lane l reads the 8 channels of half (l >> 4) & 1 of pixel pix_of_row16(l & 15); the tap
        // (lanes 32-63: the step's second tap) is added per step
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const int q = (wave * 2 * MT + i) * 16 + pix_of_row16(lane & 15);
            const int qq = q < npix_tile ? q : 0;
            const int ly = qq / a.TW, lx = qq - ly * a.TW;
            lbase[i] = ((ly * a.IS) * a.halo_w + lx * a.IS) * LP4 + 2 * ((lane >> 4) & 1);
        }
    }

    using AccT = std::conditional_t<M16, f32x4, f32x16>;
    constexpr int AM = M16 ? 2 * MT : MT, AN = M16 ? 2 * NT : NT, AE = M16 ? 4 : 16;
    AccT acc[AM][AN];
#pragma unroll
    for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int j = 0; j < AN; ++j)
#pragma unroll
            for (int e = 0; e < AE; ++e) acc[i][j][e] = 0.f;

    const int nchunks = a.Cin / (CK * a.kc);
    const int halo_pix = a.halo_h * a.halo_w;
    const int plane_floats = halo_pix * LP4 * 4;
    const int cout32 = a.Cout >> 5;
    const int quad = tid & (TPP - 1);   // this thread's float4 of channels within a staged pixel
    const int prow = tid / TPP;         // and its pixel slot within a staging pass
    auto store_pair_q = [&](float* plane, int p, int qd, uint2 t0, uint2 t1) {      // f16x3: the two terms of float4 `qd` of a plane's channels
        if constexpr (M16) {
            uint2* dst = (uint2*)plane + p * (LP4 * 2) + (qd >> 1) * 4 + (qd & 1);   // [8-channel half][term][8 ch]
            dst[0] = t0;
            dst[2] = t1;
        } else {
            uint2* dst = (uint2*)plane + p * (LP4 * 2) + qd;      // [term][CK ch]: CK*2 bytes per term
            dst[0] = t0;
            dst[CK / 4] = t1;
        }
    };
    auto store_px_q = [&](float* plane, int p, int qd, f32x4 w) {
        if constexpr (SPLIT == 0) {
            *(f32x4*)(plane + p * (LP4 * 4) + qd * 4) = w;
        } else if constexpr (SPLIT == 3) {
            uint2 t0, t1;
            split2h(w, sig, t0, t1);
            store_pair_q(plane, p, qd, t0, t1);
        } else {
            uint2 t0, t1, t2;
            split3(w, t0, t1, t2);
            uint2* dst = (uint2*)plane + p * (LP4 * 2) + qd;   // [term][CK ch]: CK*2 bytes per term
            dst[0] = t0;
            dst[CK / 4] = t1;
            dst[CK / 2] = t2;
        }
    };
    auto store_pair = [&](float* plane, int p, uint2 t0, uint2 t1) { store_pair_q(plane, p, quad, t0, t1); };
    auto store_px = [&](float* plane, int p, f32x4 w) { store_px_q(plane, p, quad, w); };

    // Single-tap (1x1) convolutions stage kc channel planes per barrier.  Plane by plane, a staging pass reads CK channels = 64
    // bytes of a pixel whose channels sit Cin * 4 bytes apart: every 128-byte line is touched by two to eight separate
    // wave-instructions and the texture-address path, not HBM, sets the pace (in-kernel stamps, profiles/r04_conv_stamps.log: the
    // two staging phases are 66 % of a block's life on the 128-channel 1x1 convolutions of ResNet-101).  The wide form below maps
    // the threads across ALL kc planes of a pixel: kc * CK consecutive channels = 256 contiguous bytes per pixel in the f16x3 mode,
    // 16 lanes each, so a wave-instruction reads four whole pixels; the thread's float4 goes to the plane it belongs to.
    auto stage_chunk_wide = [&](int ch) {
        const int tppk = TPP * a.kc;                      // threads per pixel across the planes (kc is 2 or 4: a power of two)
        const int qk = tid & (tppk - 1);
        const int pl = qk / TPP, qd = qk & (TPP - 1);
        const int prowk = tid / tppk, pppk = 256 / tppk;
        const int c = (ch * a.kc + pl) * CK + qd * 4;
        float* ldsp = lds + pl * plane_floats;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (flags & SPK_IN_AFFINE_RELU) {
            sc = *(const f32x4*)(a.in_scale + c);
            sh = *(const f32x4*)(a.in_shift + c);
        }
        constexpr int U = STAGE_U;
        for (int base = prowk; base < halo_pix; base += pppk * U) {
            f32x4 v[U];
            bool inb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int p = base + pppk * u;
                p = p < halo_pix ? p : halo_pix - 1;
                const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
                const int hx = p - hy * a.halo_w;
                const int iy = iy0 + hy, ix = ix0 + hx;
                inb[u] = iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
                const unsigned pi = inb[u] ? (unsigned)((iy * a.IWp + ix) * a.ips) : pi_safe_p;
                v[u] = *(const f32x4*)(img_in_p + pi * (unsigned)a.Cin + (unsigned)c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = base + pppk * u;
                f32x4 w = v[u];
                if (flags & SPK_IN_AFFINE_RELU) {
                    w = w * sc + sh;
                    w[0] = fmaxf(w[0], 0.f);
                    w[1] = fmaxf(w[1], 0.f);
                    w[2] = fmaxf(w[2], 0.f);
                    w[3] = fmaxf(w[3], 0.f);
                }
                if (!inb[u]) w = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (SPLIT == 3) {
                    if (flags & SPK_IN_PRESPLIT) {           // f16 pair tensor: the 16 bytes ARE the two terms
                        uint2 t0, t1;
                        spk_pair_unpack(w, t0, t1);
                        if (p < halo_pix) store_pair_q(ldsp, p, qd, t0, t1);
                        continue;
                    }
                }
                if (p < halo_pix) store_px_q(ldsp, p, qd, w);
            }
        }
    };

    // stage the kc channel planes of chunk `ch` (global -> registers -> fused input transform -> LDS)
    auto stage_chunk = [&](int ch) {
        if constexpr (!BNBWD && !PIPE) {
#ifndef CONV_NO_WIDE_STAGE
            if (a.kc > 1) {
                stage_chunk_wide(ch);
                return;
            }
#endif
        }
        for (int pl = 0; pl < a.kc; ++pl) {
            const int c = (ch * a.kc + pl) * CK + quad * 4;
            float* ldsp = lds + pl * plane_floats;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (flags & SPK_IN_AFFINE_RELU) {
                sc = *(const f32x4*)(a.in_scale + c);
                sh = *(const f32x4*)(a.in_shift + c);
            }
            if constexpr (BNBWD) {
                // The input of this data gradient is BatchNorm-backward of `in`:
                //     dz = in * mask,  xhat = (raw - mean)*invstd,  value = k1*(dz - m1 - xhat*m2)
                // computed while staging, so the separate apply pass over the tensor (and its re-read here) disappears.
                // The block that owns the tile (cout group 0) also writes the values of its own pixels back to memory
                // for the weight gradient, and dz for the shortcut path.
                const f32x4 mu = *(const f32x4*)(a.in_bn4 + c), is = *(const f32x4*)(a.in_bn4 + a.Cin + c);
                const f32x4 bsc = *(const f32x4*)(a.in_bn4 + 2 * a.Cin + c), bsh = *(const f32x4*)(a.in_bn4 + 3 * a.Cin + c);
                const f32x4 k1 = *(const f32x4*)(a.in_coef + c), m1 = *(const f32x4*)(a.in_coef + a.Cin + c);
                const f32x4 m2 = *(const f32x4*)(a.in_coef + 2 * a.Cin + c);
                const bool owner = (cg == 0);
                constexpr int U2 = STAGE_U2;
                for (int base = prow; base < halo_pix; base += PPP * U2) {
                    f32x4 v[U2], rw[U2], ac[U2];
                    unsigned mw[U2];
                    bool inb[U2], core[U2];
                    unsigned off[U2];          // element offset inside image b (32-bit: the image bases are scalars)
#pragma unroll
                    for (int u = 0; u < U2; ++u) {
                        int p = base + PPP * u;
                        p = p < halo_pix ? p : halo_pix - 1;
                        const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
                        const int hx = p - hy * a.halo_w;
                        const int iy = iy0 + hy, ix = ix0 + hx;
                        inb[u] = iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
                        core[u] = inb[u] && iy >= oy0 && iy < oy0 + a.TH && ix >= ox0 && ix < ox0 + a.TW;
                        const unsigned pi = inb[u] ? (unsigned)(iy * a.IW + ix) : pi_safe;      // pixel inside image b
                        off[u] = pi * (unsigned)a.Cin + (unsigned)c;
                        v[u] = conv_ld<4>(img_in + off[u]);
                        rw[u] = conv_ld<4>(img_raw + off[u]);
                        if (has_in_mask) mw[u] = img_mask[pi * (unsigned)(a.Cin >> 5) + (unsigned)(c >> 5)];
                        else if (has_in_act) ac[u] = *(const f32x4*)(img_act + off[u]);
                    }
#pragma unroll
                    for (int u = 0; u < U2; ++u) {
                        const int p = base + PPP * u;
                        f32x4 dz;
                        if (has_in_mask) {
                            const unsigned bits = mw[u] >> (c & 31);
#pragma unroll
                            for (int k = 0; k < 4; ++k) dz[k] = ((bits >> k) & 1u) ? v[u][k] : 0.f;
                        } else {
                            const f32x4 m = has_in_act ? ac[u] : rw[u] * bsc + bsh;
#pragma unroll
                            for (int k = 0; k < 4; ++k) dz[k] = m[k] > 0.f ? v[u][k] : 0.f;
                        }
                        f32x4 w = k1 * (dz - m1 - ((rw[u] - mu) * is) * m2);
                        if (!inb[u]) w = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (p < halo_pix) {
                            f32x4 side = w;
                            if constexpr (SPLIT == 3) {
                                uint2 t0, t1;
                                split2h(w, sig, t0, t1);
                                store_pair(ldsp, p, t0, t1);
                                // the gradient wrt the raw conv output leaves as an f16 pair tensor: the terms just formed, for free
                                if (flags & SPK_SIDE_PRESPLIT) side = spk_pair_pack(t0, t1);
                            } else
                                store_px(ldsp, p, w);
                            if (owner && core[u]) {
                                conv_st(img_draw + off[u], side);
                                if (has_side_dz) conv_st(img_dz + off[u], dz);
                                side_mx = fmaxf(fmaxf(side_mx, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
                            }
                        }
                    }
                }
            } else {
            // U independent 16-byte loads in flight per thread (addresses clamped, zero selected afterwards: no branch
                // around a load, so the compiler issues the whole batch before the first wait)
                constexpr int U = STAGE_U;
#ifdef ABL_NO_STAGE
                for (int base = prow; base < 0; base += PPP * U) {
#else
                for (int base = prow; base < halo_pix; base += PPP * U) {
#endif
                    f32x4 v[U];
                    bool inb[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        int p = base + PPP * u;
                        p = p < halo_pix ? p : halo_pix - 1;
                        const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
                        const int hx = p - hy * a.halo_w;
                        const int iy = iy0 + hy, ix = ix0 + hx;
                        inb[u] = iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
                        const unsigned pi = inb[u] ? (unsigned)((iy * a.IWp + ix) * a.ips) : pi_safe_p;
                        v[u] = *(const f32x4*)(img_in_p + pi * (unsigned)a.Cin + (unsigned)c);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int p = base + PPP * u;
                        f32x4 w = v[u];
                        if (flags & SPK_IN_AFFINE_RELU) {
                            w = w * sc + sh;
                            w[0] = fmaxf(w[0], 0.f);
                            w[1] = fmaxf(w[1], 0.f);
                            w[2] = fmaxf(w[2], 0.f);
                            w[3] = fmaxf(w[3], 0.f);
                        }
                        if (!inb[u]) w = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if constexpr (SPLIT == 3) {
                            if (flags & SPK_IN_PRESPLIT) {           // f16 pair tensor: the 16 bytes ARE the two terms
                                uint2 t0, t1;
                                spk_pair_unpack(w, t0, t1);
                                if (p < halo_pix) store_pair(ldsp, p, t0, t1);
                                continue;
                            }
                        }
                        if (p < halo_pix) store_px(ldsp, p, w);
                    }
                }
            }
        }
    };

    if constexpr (M16) {
        // ---- 16x16x32 K loop (see the note above the template): a.ntaps == 9, a.kc == 1, at most 8 x 64 halo pixels and an
        // even number of planes (checked by the launcher).  A step = 2 * MT row tiles x 2 * NT column tiles x 3 products; the
        // staging items of the next plane are dealt one per HALF step (MT row tiles: the matrix cycles of one tap of the
        // 32x32x16 form), eight items per plane.
        const f32x4* lds4 = (const f32x4*)lds;
        const int plane4 = halo_pix * LP4;                           // 16-byte units per slot
        float* dump = lds + 2 * plane_floats;
        const int hik = lane >> 5;                                   // this lane's unit of a step's pair
        const bool aff = (flags & SPK_IN_AFFINE_RELU) != 0;
        const float floor_v = aff ? 0.f : -__builtin_inff();         // ReLU only with the fused input transform
        // packed weights [tap][Cin/16][term][Cout/32][64 lanes][8 fp16] (pack.hip): the 16 bytes with input channels
        // 8 kh .. 8 kh + 7 of (tap, plane) for output channel n are lane slot (n & 31) + 32 kh of tile n >> 5
        const int grp_stride = NTERM * cout32 * 256, tap_stride = (a.Cin >> 4) * grp_stride, term_stride = cout32 * 256;
        const float* wq = a.wpk + (size_t)(cg * NT) * 256 + ((lane & 15) + 32 * ((lane >> 4) & 1)) * 4;
        // the plane being staged (set per phase)
        int cn = quad * 4;
        float* nxt = dump;
        bool more = false;
        f32x4 scn = {1.f, 1.f, 1.f, 1.f}, shn = {0.f, 0.f, 0.f, 0.f};
        unsigned inbm = 0;
        // staging items in flight: an item is consumed D16 half steps after its load
        constexpr int D16 = PRE ? PIPE16_D : PIPE16_D_CONV;
        f32x4 pre[D16];
        auto phase = [&](int chunk, int slot) {
            more = chunk < nchunks;
            cn = (more ? chunk : nchunks - 1) * CK + quad * 4;
            nxt = lds + slot * plane_floats;
            if (aff) {
                scn = *(const f32x4*)(a.in_scale + cn);
                shn = *(const f32x4*)(a.in_shift + cn);
            }
        };
        // (the item geometry is loop-invariant; formed from an opaque copy of the thread's pixel slot so that the optimizer does
        //  not keep eight items' addresses and flags in registers across the loop)
        auto issue = [&](auto uc) {                                  // global load of staging item u of that plane
            constexpr int u = decltype(uc)::value;
            if constexpr (u >= 8) return;
            int pr = prow;
            asm volatile("" : "+v"(pr));
            const int p = min(pr + PPP * u, halo_pix - 1);
            const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
            const int hx = p - hy * a.halo_w;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = (iy >= 0) & (iy < a.IH) & (ix >= 0) & (ix < a.IW);
            inbm = (inbm & ~(1u << u)) | ((unsigned)ok << u);
            const unsigned pi = ok ? (unsigned)((iy * a.IWp + ix) * a.ips) : pi_safe_p;
            pre[u % D16] = *(const f32x4*)(img_in_p + pi * (unsigned)a.Cin + (unsigned)cn);
        };
        auto finish = [&](auto uc) {                                 // transform + fp16 split + LDS write of item u (branch-free)
            constexpr int u = decltype(uc)::value;
            int pr = prow;
            asm volatile("" : "+v"(pr));
            const int p = pr + PPP * u;
            const bool ok = (inbm >> u) & 1u;
            const bool real = more & (p < halo_pix);
            if constexpr (PRE) {
                const uint4 bb = __builtin_bit_cast(uint4, pre[u % D16]);
                const uint2 t0 = {ok ? bb.x : 0u, ok ? bb.y : 0u}, t1 = {ok ? bb.z : 0u, ok ? bb.w : 0u};
                store_pair(real ? nxt : dump, real ? p : 0, t0, t1);
            } else {
                f32x4 w = pre[u % D16] * scn + shn;
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = ok ? fmaxf(w[k], floor_v) : 0.f;
                store_px(real ? nxt : dump, real ? p : 0, w);
            }
        };
        auto issue_first = [&]() {          // the first D16 items of a plane (issue() ignores items >= 8)
            issue(std::integral_constant<int, 0>{});
            issue(std::integral_constant<int, 1>{});
            issue(std::integral_constant<int, (D16 > 2 ? 2 : 8)>{});
            issue(std::integral_constant<int, (D16 > 3 ? 3 : 8)>{});
            issue(std::integral_constant<int, (D16 > 4 ? 4 : 8)>{});
            issue(std::integral_constant<int, (D16 > 5 ? 5 : 8)>{});
        };
        static_assert(D16 >= 2 && D16 <= 6, "staging items in flight");
        f32x4 aq[2][NTERM];                  // A fragments of the current and of the next 16-row tile
        f32x4 bx[NTERM][AN], by[NTERM][AN];  // B fragments of the current and of the next step
        auto load_b = [&](f32x4 (*bf)[AN], int wsel) {
            const float* wp = wq + wsel;
#pragma unroll
            for (int s = 0; s < NTERM; ++s)
#pragma unroll
                for (int j = 0; j < AN; ++j) bf[s][j] = *(const f32x4*)(wp + s * term_stride + (j >> 1) * 256 + (j & 1) * 64);
        };
        auto load_a = [&](f32x4* af, int i16, int osel) {
#pragma unroll
            for (int s = 0; s < NTERM; ++s) af[s] = lds4[lbase[i16] + osel + s];
        };
        // (the lane's unit is made opaque at every use: the optimizer would otherwise hoist the selects - and lbase[i] + offset for
        //  all nine steps, 54 registers - out of the plane-pair loop; it even factors sel(a + s, b + s) into sel(a, b) + s to do so)
        auto sel = [&](int x0, int x1) {
            int hk = hik;
            asm volatile("" : "+v"(hk));
            return hk ? x1 : x0;
        };
#define IC(n) std::integral_constant<int, n>{}
        // One scheduling region per half step: MT * AN * 3 matrix instructions, the A fragments one row tile ahead, in the first
        // half the B fragments of the next step, and one staging item (finish fin, issue fin + PIPE16_D).  roll: after the last row
        // tile of the step, read row tile 0 of the next step (not across a barrier that completes the slot it reads).
        auto hstep = [&](auto half_c, auto fin_c, auto roll_c, const f32x4 (*bc)[AN], f32x4 (*bn)[AN], int w_next, int o_cur, int o_next) {
            constexpr int half = decltype(half_c)::value, fin = decltype(fin_c)::value;
            constexpr bool roll = decltype(roll_c)::value != 0;
#pragma unroll
            for (int ii = 0; ii < MT; ++ii) {
                // One scheduling region per row tile: its AN * 3 matrix instructions in source order (four independent
                // accumulators in rotation - a dependent 16x16x32 issued back to back waits for its predecessor), the A fragments
                // of the NEXT row tile read first, and one piece of the other work: the B fragments of the next step + the global
                // conversion and LDS write of a staging item (tile 0), the global load of the item PIPE16_D later (tile 1).
                const int i16 = half * MT + ii;
                __builtin_amdgcn_sched_barrier(0);
                if (i16 + 1 < 2 * MT) load_a(aq[(i16 + 1) & 1], i16 + 1, o_cur);
                else if constexpr (roll) load_a(aq[0], 0, o_next);
                __builtin_amdgcn_sched_barrier(0);      // (left to the scheduler the reads sink to their first use: no latency cover)
                if (ii == 0) {
                    if constexpr (half == 0) load_b(bn, w_next);
#ifndef M16_NO_STAGE
#ifndef M16_NO_FINISH
                    if constexpr (fin >= 0) finish(IC(fin));              // (reads pre[fin % D16] before issue below refills it)
#endif
#endif
                }
#ifndef M16_NO_STAGE
                if (ii == 1 % MT) {
#ifndef M16_NO_ISSUE
                    if constexpr (fin >= 0) issue(IC(fin + D16));
#endif
                }
#endif
#pragma unroll
                for (int sum = 1; sum >= 0; --sum)
#pragma unroll
                    for (int sa = 0; sa < NTERM; ++sa) {
                        const int sb = sum - sa;
                        if (sb < 0 || sb >= NTERM) continue;
#pragma unroll
                        for (int j = 0; j < AN; ++j) {
                            acc[i16][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, aq[i16 & 1][sa]),
                                                                                __builtin_bit_cast(f16x8, bc[sb][j]), acc[i16][j], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0x7F7);      // everything but a matrix instruction may cross
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // offsets of a unit (plane c, tap t): A in LDS (16-byte units), B in the packed weights (floats)
        auto ao = [&](int slot, int t) { return slot * plane4 + a.tap_off[t]; };
        auto bo = [&](int c, int t) { return a.tap_w[t] * tap_stride + c * grp_stride; };
        STAMP(0);
        __syncthreads();
        STAMP(1);
        stage_chunk(0);                      // first plane: staged the plain way into slot 0
        STAMP(2);
        __syncthreads();
        STAMP(3);
        // (per-lane offsets of a step: formed where they are used - nine of them held across the loop cost nine registers)
#define O0 sel(ao(0, 0), ao(0, 1))
#define O1 sel(ao(0, 2), ao(0, 3))
#define O2 sel(ao(0, 4), ao(0, 5))
#define O3 sel(ao(0, 6), ao(0, 7))
#define O4 sel(ao(0, 8), ao(1, 0))
#define O5 sel(ao(1, 1), ao(1, 2))
#define O6 sel(ao(1, 3), ao(1, 4))
#define O7 sel(ao(1, 5), ao(1, 6))
#define O8 sel(ao(1, 7), ao(1, 8))
        load_b(bx, sel(bo(0, 0), bo(0, 1)));
        load_a(aq[0], 0, O0);
        for (int cp = 0; cp < nchunks; cp += 2) {
            const int cn2 = cp + 2 < nchunks ? cp + 2 : cp + 1;      // after the last pair: a harmless re-fetch
            // ---- even plane (slot 0): taps (0,1) (2,3) (4,5) (6,7); plane cp + 1 is staged into slot 1
            phase(cp + 1, 1);
            issue_first();
            hstep(IC(0), IC(0), IC(1), bx, by, sel(bo(cp, 2), bo(cp, 3)), O0, O1);
            hstep(IC(1), IC(1), IC(1), bx, by, 0, O0, O1);
            hstep(IC(0), IC(2), IC(1), by, bx, sel(bo(cp, 4), bo(cp, 5)), O1, O2);
            hstep(IC(1), IC(3), IC(1), by, bx, 0, O1, O2);
            hstep(IC(0), IC(4), IC(1), bx, by, sel(bo(cp, 6), bo(cp, 7)), O2, O3);
            hstep(IC(1), IC(5), IC(1), bx, by, 0, O2, O3);
            hstep(IC(0), IC(6), IC(1), by, bx, sel(bo(cp, 8), bo(cp + 1, 0)), O3, O4);
            hstep(IC(1), IC(7), IC(0), by, bx, 0, O3, O4);
            __syncthreads();                 // slot 1 is complete
            load_a(aq[0], 0, O4);
            // ---- (8 | 0 of the odd plane), then the odd plane (slot 1): (1,2) (3,4) (5,6) (7,8); plane cp + 2 goes to slot 0 once
            // every wave is past the straddling step
            phase(cp + 2, 0);
            issue_first();
            hstep(IC(0), IC(-1), IC(1), bx, by, sel(bo(cp + 1, 1), bo(cp + 1, 2)), O4, O5);
            hstep(IC(1), IC(-1), IC(1), bx, by, 0, O4, O5);
            __syncthreads();                 // slot 0 is free
            hstep(IC(0), IC(0), IC(1), by, bx, sel(bo(cp + 1, 3), bo(cp + 1, 4)), O5, O6);
            hstep(IC(1), IC(1), IC(1), by, bx, 0, O5, O6);
            hstep(IC(0), IC(2), IC(1), bx, by, sel(bo(cp + 1, 5), bo(cp + 1, 6)), O6, O7);
            hstep(IC(1), IC(3), IC(1), bx, by, 0, O6, O7);
            hstep(IC(0), IC(4), IC(1), by, bx, sel(bo(cp + 1, 7), bo(cp + 1, 8)), O7, O8);
            hstep(IC(1), IC(5), IC(1), by, bx, 0, O7, O8);
            hstep(IC(0), IC(6), IC(1), bx, by, sel(bo(cn2, 0), bo(cn2, 1)), O8, O0);
            hstep(IC(1), IC(7), IC(0), bx, by, 0, O8, O0);
            __syncthreads();                 // slot 0 is complete, slot 1 is free
            load_a(aq[0], 0, O0);
            // nine steps per pair: the two B buffers have swapped roles
#pragma unroll
            for (int s2 = 0; s2 < NTERM; ++s2)
#pragma unroll
                for (int j = 0; j < AN; ++j) bx[s2][j] = by[s2][j];
        }
        STAMP(4);
#undef O0
#undef O1
#undef O2
#undef O3
#undef O4
#undef O5
#undef O6
#undef O7
#undef O8
#undef IC
    } else {
    f32x4 pb0[PIPE ? 2 : 1][NT], pb1[PIPE ? 2 : 1][NT], pb2[PIPE ? 2 : 1][NT];     // PIPE: B fragments, carried from chunk to chunk
    for (int ch = 0; ch < nchunks; ++ch) {
        if constexpr (SPLIT == 0) {
            // K loop over (tap, 8-cin group), software-pipelined one group ahead: while the 4*MT*NT MFMAs of a group
            // issue (>= 1 k cycles), the B fragments (L2 -> VGPR) and A fragments (LDS -> VGPR) of the next group are
            // already in flight, and the next tap's table entries (scalar loads) are fetched a whole tap early, so the
            // matrix pipe never waits on a memory round trip inside a wave.
            const f32x4* lds4 = (const f32x4*)lds;
            const float* wbase = a.wpk + ((size_t)(ch * a.kc * 4) * cout32 + cg * NT) * 256 + lane * 4;
            const size_t tap_stride = (size_t)(a.Cin >> 3) * cout32 * 256;
            const size_t grp_stride = (size_t)cout32 * 256;
            auto load_b = [&](f32x4* bf, int tw, int g) {
                const float* wp = wbase + (size_t)tw * tap_stride + (size_t)g * grp_stride;
    #ifdef ABL_NO_BLOAD
    #pragma unroll
                for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(bf[j]) : "s"(wp));
    #else
    #pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = *(const f32x4*)(wp + j * 256);
    #endif
            };
            auto load_a = [&](f32x4* af, int toff4, int g) {
    #ifdef ABL_NO_ALOAD
    #pragma unroll
                for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(af[i]) : "s"(toff4 + g));
    #else
    #pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = lds4[lbase[i] + toff4 + g * 2];
    #endif
            };
            auto mma = [&](const f32x4* af, const f32x4* bf) {
    #pragma unroll
                for (int s = 0; s < 4; ++s)
    #pragma unroll
                    for (int i = 0; i < MT; ++i)
    #pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            };
            // B fragments are prefetched a whole tap (4 groups >= 4 k MFMA cycles) ahead into a 4-deep register ring;
            // A fragments one group ahead (LDS latency is ~100 cycles).
            f32x4 bq[4][NT], a0[MT], a1[MT];
    #if defined(ABL_NO_BLOAD) || defined(ABL_NO_ALOAD)
            for (int g = 0; g < 4; ++g)
                for (int j = 0; j < NT; ++j) bq[g][j] = (f32x4){1.f, 2.f, 3.f, 4.f};
            for (int i = 0; i < MT; ++i) a0[i] = a1[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
    #endif
            int tw = a.tap_w[0], toff = a.tap_off[0], tg = a.tap_g[0];
    #pragma unroll
            for (int g = 0; g < 4; ++g) load_b(bq[g], tw, tg + g);
            __syncthreads();  // every wave is done reading the previous chunk's tile
            stage_chunk(ch);
            __syncthreads();

            load_a(a0, toff, 0);
            for (int t = 0; t < a.ntaps; ++t) {
                // the prefetches below are unconditional (the last tap re-fetches itself) so that the loop body is
                // branch-free and the compiler can emit counted vmcnt waits instead of vmcnt(0) at every tap
                const int tn = t + 1 < a.ntaps ? t + 1 : t;
                const int tw_n = a.tap_w[tn], toff_n = a.tap_off[tn], tg_n = a.tap_g[tn];
                // group 0
                load_a(a1, toff, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, bq[0]);
                __builtin_amdgcn_sched_barrier(0);
                load_b(bq[0], tw_n, tg_n + 0);
                // group 1
                load_a(a0, toff, 2);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, bq[1]);
                __builtin_amdgcn_sched_barrier(0);
                load_b(bq[1], tw_n, tg_n + 1);
                // group 2
                load_a(a1, toff, 3);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, bq[2]);
                __builtin_amdgcn_sched_barrier(0);
                load_b(bq[2], tw_n, tg_n + 2);
                // group 3
                load_a(a0, toff_n, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, bq[3]);
                __builtin_amdgcn_sched_barrier(0);
                load_b(bq[3], tw_n, tg_n + 3);
                tw = tw_n;
                toff = toff_n;
                tg = tg_n;
            }
        } else {
            // ---- bf16-split K loop: one step = one (tap, 16-channel plane): 3 A reads per m-tile (one per term), 3 B loads
            // per n-tile, SPLIT MFMAs per (m-tile, n-tile).
            const f32x4* lds4 = (const f32x4*)lds;
            // packed weights: [tap][Cin/16][term][Cout/32][64 lanes][8 bf16]
            const float* wbase = a.wpk + ((size_t)(ch * a.kc) * NTERM * cout32 + cg * NT) * 256 + lane * 4;
            const size_t tap_stride = (size_t)(a.Cin >> 4) * NTERM * cout32 * 256;
            const size_t grp_stride = (size_t)NTERM * cout32 * 256;
            const size_t term_stride = (size_t)cout32 * 256;
            auto load_b = [&](f32x4 (*bf)[NT], int tw, int g) {
#ifdef ABL_B_SAMEADDR
                const float* wp = wbase + (size_t)(tw & 1) * tap_stride;   // diagnostic: a cache-hot address every step
#else
                const float* wp = wbase + (size_t)tw * tap_stride + (size_t)g * grp_stride;
#endif
#pragma unroll
                for (int s = 0; s < NTERM; ++s)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
#ifdef ABL_NO_BLOAD
                        asm volatile("" : "+v"(bf[s][j]) : "s"(wp));
#elif defined(ABL_B_HALF)
                        // diagnostic (wrong numerics): half the weight-fragment bytes through the vector L1, same matrix work
                        if (s == 0) bf[s][j] = *(const f32x4*)(wp + j * 256);
                        else { bf[s][j] = bf[0][j]; asm volatile("" : "+v"(bf[s][j])); }
#else
                        bf[s][j] = *(const f32x4*)(wp + s * term_stride + j * 256);
#endif
                    }
            };
            auto load_a = [&](f32x4 (*af)[MT], int toff4, int g) {
#pragma unroll
                for (int s = 0; s < NTERM; ++s)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
#ifdef ABL_NO_ALOAD
                        asm volatile("" : "+v"(af[s][i]) : "s"(toff4 + g));
#else
                        af[s][i] = lds4[lbase[i] + toff4 + s * (CK / 8) + g * 2];
#endif
                    }
            };
            if constexpr (PIPE) {
                static_assert(SPLIT == 3, "PIPE: f16x3 operands");
                // a.ntaps == 9, a.kc == 1 (checked by the launcher); LDS holds two tiles of plane_floats
                const int plane4 = halo_pix * LP4;                       // 16-byte units per tile
                const f32x4* cur4 = lds4 + (ch & 1) * plane4;
                float* nxt = lds + ((ch + 1) & 1) * plane_floats;
                const bool more = ch + 1 < nchunks;
                // Branch-free staging (a branch would end the scheduling region and with it the interleave): every item is
                // loaded, transformed and written; items that do not exist (beyond the halo, or after the last chunk) read
                // a safe address and write a dump pixel behind the two tiles.
                const int cn = (more ? ch + 1 : ch) * CK + quad * 4;     // this thread's channels in the next chunk
                float* dump = lds + 2 * plane_floats;
                unsigned inbm = 0;
                // ---- plain input (optionally BN + ReLU of the producing layer) ----
                const bool aff = (flags & SPK_IN_AFFINE_RELU) != 0;
                f32x4 scn = {1.f, 1.f, 1.f, 1.f}, shn = {0.f, 0.f, 0.f, 0.f};
                if (!BNBWD && aff) {
                    scn = *(const f32x4*)(a.in_scale + cn);
                    shn = *(const f32x4*)(a.in_shift + cn);
                }
                const float floor_v = aff ? 0.f : -__builtin_inff();     // ReLU only with the fused input transform
                f32x4 pre[PIPE_D];
                // ---- fused BatchNorm backward (see stage_chunk): value = k1*(dz - m1 - xhat*m2), dz = in*mask ----
                f32x4 prr[BNBWD ? PIPE_D : 1];                           // raw conv output of that BatchNorm
                unsigned prm[BNBWD && BITS ? PIPE_D : 1], pof[BNBWD ? PIPE_D : 1];   // mask word, element offset in the image
                unsigned corem = 0;                                      // bit u: item u is one of the tile's own pixels
                f32x4 bmu = {}, bis = {}, bk1 = {}, bm1 = {}, bm2 = {}, bbsc = {}, bbsh = {};
                if constexpr (BNBWD) {
                    bmu = *(const f32x4*)(a.in_bn4 + cn);
                    bis = *(const f32x4*)(a.in_bn4 + a.Cin + cn);
                    if constexpr (!BITS) {
                        bbsc = *(const f32x4*)(a.in_bn4 + 2 * a.Cin + cn);
                        bbsh = *(const f32x4*)(a.in_bn4 + 3 * a.Cin + cn);
                    }
                    bk1 = *(const f32x4*)(a.in_coef + cn);
                    bm1 = *(const f32x4*)(a.in_coef + a.Cin + cn);
                    bm2 = *(const f32x4*)(a.in_coef + 2 * a.Cin + cn);
                }
                const bool owner = (cg == 0);
                // side outputs of an item wait for the end of its tap (their stores are conditional: a branch)
                f32x4 sd_w = {}, sd_dz = {};
                unsigned sd_off = 0;
                bool sd_go = false;
                auto issue = [&](auto uc) {                              // global load(s) of staging item u of the next chunk
                    constexpr int u = decltype(uc)::value;
                    if constexpr (u >= 9) return;
                    const int p = min(prow + PPP * u, halo_pix - 1);
                    const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
                    const int hx = p - hy * a.halo_w;
                    const int iy = iy0 + hy, ix = ix0 + hx;
                    const bool ok = (iy >= 0) & (iy < a.IH) & (ix >= 0) & (ix < a.IW);
                    inbm = (inbm & ~(1u << u)) | ((unsigned)ok << u);
                    if constexpr (BNBWD) {
                        const bool core = ok & (iy >= oy0) & (iy < oy0 + a.TH) & (ix >= ox0) & (ix < ox0 + a.TW);
                        corem = (corem & ~(1u << u)) | ((unsigned)core << u);
                        const unsigned pi = ok ? (unsigned)(iy * a.IW + ix) : pi_safe;
                        const unsigned off = pi * (unsigned)a.Cin + (unsigned)cn;
                        pof[u % PIPE_D] = off;
                        pre[u % PIPE_D] = *(const f32x4*)(img_in + off);
                        prr[u % PIPE_D] = *(const f32x4*)(img_raw + off);
                        if constexpr (BITS) prm[u % PIPE_D] = img_mask[pi * (unsigned)(a.Cin >> 5) + (unsigned)(cn >> 5)];
                    } else {
                        const unsigned pi = ok ? (unsigned)((iy * a.IWp + ix) * a.ips) : pi_safe_p;
                        pre[u % PIPE_D] = *(const f32x4*)(img_in_p + pi * (unsigned)a.Cin + (unsigned)cn);
                    }
                };
                auto finish = [&](auto uc) {                             // transform + fp16 split + LDS write of item u
                    constexpr int u = decltype(uc)::value;
                    const int p = prow + PPP * u;
                    const bool ok = (inbm >> u) & 1u;
                    const bool real = more & (p < halo_pix);
                    f32x4 w;
                    if constexpr (BNBWD) {
                        const f32x4 v = pre[u % PIPE_D], rw = prr[u % PIPE_D];
                        f32x4 dz;
                        if constexpr (BITS) {
                            const unsigned bits = prm[u % PIPE_D] >> (cn & 31);
#pragma unroll
                            for (int k = 0; k < 4; ++k) dz[k] = ((bits >> k) & 1u) ? v[k] : 0.f;
                        } else {
                            const f32x4 m = rw * bbsc + bbsh;
#pragma unroll
                            for (int k = 0; k < 4; ++k) dz[k] = m[k] > 0.f ? v[k] : 0.f;
                        }
                        w = bk1 * (dz - bm1 - ((rw - bmu) * bis) * bm2);
#pragma unroll
                        for (int k = 0; k < 4; ++k) w[k] = ok ? w[k] : 0.f;
                        sd_go = real & owner & (((corem >> u) & 1u) != 0);
                        sd_w = w;
                        sd_dz = dz;
                        sd_off = pof[u % PIPE_D];
                        const float mx = fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fmaxf(fabsf(w[2]), fabsf(w[3])));
                        side_mx = sd_go ? fmaxf(side_mx, mx) : side_mx;
                    } else if constexpr (PRE) {
                        // f16 pair tensor: select zero bits outside the image, write the two terms as they are
                        const uint4 b = __builtin_bit_cast(uint4, pre[u % PIPE_D]);
                        const uint2 t0 = {ok ? b.x : 0u, ok ? b.y : 0u}, t1 = {ok ? b.z : 0u, ok ? b.w : 0u};
                        store_pair(real ? nxt : dump, real ? p : 0, t0, t1);
                        return;
                    } else {
                        w = pre[u % PIPE_D] * scn + shn;
#pragma unroll
                        for (int k = 0; k < 4; ++k) w[k] = ok ? fmaxf(w[k], floor_v) : 0.f;
                    }
                    store_px(real ? nxt : dump, real ? p : 0, w);
                };
                auto flush = [&]() {                                     // after the tap: the item's side outputs (tile's own pixels)
                    if constexpr (BNBWD) {
                        if (sd_go) {
                            *(f32x4*)(img_draw + sd_off) = sd_w;
                            if (img_dz) *(f32x4*)(img_dz + sd_off) = sd_dz;
                        }
                    }
                };
                f32x4 aq[NTERM][MT];
                auto pstep = [&](const f32x4 (*bc)[NT], f32x4 (*bn)[NT], int o_next, int w_n2, int g_n2, auto uc) {
                    constexpr int u = decltype(uc)::value;
                    // One scheduling region per tap: MT*NT*3 matrix instructions, the A fragments of the next tap, the B
                    // fragments two taps ahead, and one staging item of the next chunk (finish u, issue u + PIPE_D).  A
                    // wave's VALU instruction only overlaps an MFMA that directly precedes it in its own stream, so the
                    // item's ~60 VALU instructions are dealt out a few after every MFMA (sched_group_barrier pattern below)
                    // instead of following the whole group.
                    __builtin_amdgcn_sched_barrier(0);
                    load_b(bn, w_n2, g_n2);
                    finish(uc);                                           // (reads pre[u % PIPE_D] before the next line refills it)
                    issue(std::integral_constant<int, u + PIPE_D>{});
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
#pragma unroll
                        for (int sum = Cfg::MAXSUM; sum >= 0; --sum)
#pragma unroll
                            for (int sa = 0; sa < NTERM; ++sa) {
                                const int sb = sum - sa;
                                if (sb < 0 || sb >= NTERM) continue;
#pragma unroll
                                for (int j = 0; j < NT; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aq[sa][i]),
                                                                                      __builtin_bit_cast(f16x8, bc[sb][j]), acc[i][j], 0, 0, 0);
                            }
#pragma unroll
                        for (int s = 0; s < NTERM; ++s) aq[s][i] = cur4[lbase[i] + o_next + s * (CK / 8)];
                    }
#pragma unroll
                    for (int k = 0; k < MT * NT * 3; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, PIPE_VPM, 0);     // a few VALU instructions in its shadow
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // (an LDS write of the item when one is ready)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    flush();
                };
                if (ch == 0) {                   // first chunk: staged the plain way into slot 0
                    __syncthreads();
                    stage_chunk(0);
                    __syncthreads();
                }
                if (ch == 0) {
                    load_b(pb0, a.tap_w[0], a.tap_g[0]);
                    load_b(pb1, a.tap_w[1], a.tap_g[1]);
                }
#pragma unroll
                for (int s = 0; s < NTERM; ++s)
#pragma unroll
                    for (int i = 0; i < MT; ++i) aq[s][i] = cur4[lbase[i] + a.tap_off[0] + s * (CK / 8)];
                issue(std::integral_constant<int, 0>{});
                issue(std::integral_constant<int, 1>{});
                issue(std::integral_constant<int, 2>{});
#define IC(n) std::integral_constant<int, n>{}
                // the last two taps fetch the B fragments of the NEXT chunk's taps 0 and 1 (group offset + one chunk) into the
                // buffers that chunk starts from; after the last chunk they re-read its own (unused)
                const int gn = more ? 1 : 0;           // one K group = one 16-channel plane (kc == 1)
                pstep(pb0, pb2, a.tap_off[1], a.tap_w[2], a.tap_g[2], IC(0));
                pstep(pb1, pb0, a.tap_off[2], a.tap_w[3], a.tap_g[3], IC(1));
                pstep(pb2, pb1, a.tap_off[3], a.tap_w[4], a.tap_g[4], IC(2));
                pstep(pb0, pb2, a.tap_off[4], a.tap_w[5], a.tap_g[5], IC(3));
                pstep(pb1, pb0, a.tap_off[5], a.tap_w[6], a.tap_g[6], IC(4));
                pstep(pb2, pb1, a.tap_off[6], a.tap_w[7], a.tap_g[7], IC(5));
                pstep(pb0, pb2, a.tap_off[7], a.tap_w[8], a.tap_g[8], IC(6));
                pstep(pb1, pb0, a.tap_off[8], a.tap_w[0], a.tap_g[0] + gn, IC(7));
                pstep(pb2, pb1, a.tap_off[8], a.tap_w[1], a.tap_g[1] + gn, IC(8));
#undef IC
                __syncthreads();                 // slot (ch + 1) & 1 is complete, slot ch & 1 is free
            } else
            {
                // Three-deep operand pipeline.  A step (one tap of a 16-channel plane) is only 6*MT*NT MFMAs of 32 cycles -
                // about half a microsecond, less than an L2 round trip - so the B fragments are requested TWO steps
                // ahead (three register buffers in rotation), while the A fragments (LDS, ~100 cycles) roll inside one
                // buffer: as soon as the MFMAs of m-tile i have issued, the next step's fragments of that m-tile are read
                // into the same registers.  Tap-table entries (scalar loads) are fetched three steps early.
                f32x4 b0[NTERM][NT], b1[NTERM][NT], b2[NTERM][NT], aq[NTERM][MT];
#if defined(ABL_NO_BLOAD) || defined(ABL_NO_ALOAD)
                for (int s = 0; s < NTERM; ++s) {
                    for (int j = 0; j < NT; ++j) b0[s][j] = b1[s][j] = b2[s][j] = (f32x4){1.f, 2.f, 3.f, 4.f};
                    for (int i = 0; i < MT; ++i) aq[s][i] = (f32x4){1.f, 2.f, 3.f, 4.f};
                }
#endif
                const int last = a.ntaps - 1;
                auto step = [&](const f32x4 (*bc)[NT], f32x4 (*bn)[NT], int o_next, int w_n2, int g_n2) {
                    load_b(bn, w_n2, g_n2);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int sum = Cfg::MAXSUM; sum >= 0; --sum)
#pragma unroll
                            for (int sa = 0; sa < NTERM; ++sa) {
                                const int sb = sum - sa;
                                if (sb < 0 || sb >= NTERM) continue;
#pragma unroll
                                for (int j = 0; j < NT; ++j) {
                                    if constexpr (SPLIT == 3)
                                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aq[sa][i]),
                                                                                          __builtin_bit_cast(f16x8, bc[sb][j]),
                                                                                          acc[i][j], 0, 0, 0);
                                    else
                                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq[sa][i]),
                                                                                           __builtin_bit_cast(bf16x8, bc[sb][j]),
                                                                                           acc[i][j], 0, 0, 0);
                                }
                            }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < NTERM; ++s) {
#ifdef ABL_NO_ALOAD
                            asm volatile("" : "+v"(aq[s][i]) : "s"(o_next));
#else
                            aq[s][i] = lds4[lbase[i] + o_next + s * (CK / 8)];
#endif
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                load_b(b0, a.tap_w[0], a.tap_g[0]);
                if (ch == 0) STAMP(0);
                __syncthreads();  // every wave is done reading the previous chunk's tile
                if (ch < 2) STAMP(1 + 4 * ch);
                stage_chunk(ch);
                if (ch < 2) STAMP(2 + 4 * ch);
                __syncthreads();
                if (ch < 2) STAMP(3 + 4 * ch);
                load_b(b1, a.tap_w[min(1, last)], a.tap_g[min(1, last)]);   // (after staging: its registers are busy there)
                load_a(aq, a.tap_off[0], 0);
                int o1 = a.tap_off[min(1, last)], w2 = a.tap_w[min(2, last)], g2 = a.tap_g[min(2, last)];
                int t = 0;
                for (; t + 2 < a.ntaps; t += 3) {
                    const int i2 = min(t + 2, last), i3 = min(t + 3, last), i4 = min(t + 4, last), i5 = min(t + 5, last);
                    const int o2 = a.tap_off[i2], w3 = a.tap_w[i3], g3 = a.tap_g[i3];
                    const int o3 = a.tap_off[i3], w4 = a.tap_w[i4], g4 = a.tap_g[i4];
                    const int o4 = a.tap_off[i4], w5 = a.tap_w[i5], g5 = a.tap_g[i5];
                    step(b0, b2, o1, w2, g2);
                    step(b1, b0, o2, w3, g3);
                    step(b2, b1, o3, w4, g4);
                    o1 = o4;
                    w2 = w5;
                    g2 = g5;
                }
                if (t < a.ntaps) {      // one or two steps left: they are in b0 (and b1); prefetches re-fetch the last tap
                    step(b0, b2, o1, w2, g2);
                    if (t + 1 < a.ntaps) step(b1, b0, a.tap_off[last], a.tap_w[last], a.tap_g[last]);
                }
                if (ch < 2) STAMP(4 + 4 * ch);
            }
        }
    }

    }   // !M16

    // ---- epilogue.  C/D layout of a 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), i.e. a lane
    // holds ONE channel of 16 pixels.  Each wave transposes one m-tile at a time through a private LDS slab
    // [32 pixels][NT*32 + 4] so that a lane then owns 4 consecutive channels of one pixel: 16-byte global stores,
    // 16-byte residual / scale loads, and 4x fewer store instructions.  No block barrier after the first one:
    // statistics are reduced per wave (shuffles) and written as one partial row per wave.
    constexpr int LW = NT * 32 + 4;        // slab row pitch in floats (16-byte aligned, bank-staggered)
    constexpr int Q = NT * 8;              // float4 quads per pixel row
    constexpr int RPP = 64 / Q;            // pixel rows covered by one 64-lane pass
    __syncthreads();                       // all waves are done with the input tile
    float* slab = lds + wave * (32 * LW);
    const int qc = lane % Q;               // this lane's channel quad
    const int qr = lane / Q;               // and its row within a pass
    f32x4 es = {1.f, 1.f, 1.f, 1.f}, eh = {0.f, 0.f, 0.f, 0.f};
    if (flags & SPK_EPI_AFFINE) {
        es = *(const f32x4*)(a.epi_scale + n0 + qc * 4);
        eh = *(const f32x4*)(a.epi_shift + n0 + qc * 4);
    }
    f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bis = bmu, bsc = bmu, bsh = bmu;
    if (flags & SPK_EPI_BNBWD) {
        bmu = *(const f32x4*)(a.bn4 + n0 + qc * 4);
        bis = *(const f32x4*)(a.bn4 + a.Cout + n0 + qc * 4);
        bsc = *(const f32x4*)(a.bn4 + 2 * a.Cout + n0 + qc * 4);
        bsh = *(const f32x4*)(a.bn4 + 3 * a.Cout + n0 + qc * 4);
    }
    f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
    float out_mx = 0.f;
    auto to_slab = [&](int i) {          // the accumulators of m-tile i (32 pixels x NT * 32 channels) -> this wave's slab
        if constexpr (M16) {
            // 16 x 16 tiles: col = lane & 15, row = 4 (lane >> 4) + e; tile rows are a permutation of the pixels (pix_of_row16)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * ii + pix_of_row16(4 * (lane >> 4) + e);
#pragma unroll
                    for (int j = 0; j < 2 * NT; ++j) slab[row * LW + j * 16 + (lane & 15)] = acc[2 * i + ii][j][e] * inv_sig * inv_wsig;
                }
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    slab[row * LW + j * 32 + r] = SPLIT == 3 ? acc[i][j][e] * inv_sig * inv_wsig : acc[i][j][e];
                }
        }
    };
#ifndef SPK_EPI_BATCH
#define SPK_EPI_BATCH 1      // 0: A/B builds with the pass-by-pass epilogue everywhere (tools/variant.sh)
#endif
    // Which instantiations batch their epilogue reads (below): the data gradients - the fused BatchNorm-backward form of
    // conv_mfma_kernel and the pair-input form of conv_pipe_kernel - whose launches always carry the shortcut add and / or the
    // BatchNorm-backward statistics.  The forward instantiations keep the pass-by-pass form: they have no epilogue reads, the
    // batch costs registers (84 -> 126 on the 32-channel forward kernel: one wave per SIMD less on an HBM-bound kernel,
    // measured +30 % time), and conv_pipe_kernel with conversion while staging sits at its 256-register launch bound.
    constexpr bool EPB = SPK_EPI_BATCH && (PRE || (BNBWD && !PIPE));
    if constexpr (!EPB) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            to_slab(i);
            // same-wave LDS traffic is ordered; the compiler inserts the lgkmcnt wait for the reads below
#pragma unroll
            for (int k = 0; k < 32 / RPP; ++k) {
                const int row = k * RPP + qr;
                const int ob = __shfl(obase[i], row, 64);
                f32x4 v = *(const f32x4*)(slab + row * LW + qc * 4);
#ifdef ABL_NO_EPI
                asm volatile("" ::"v"(v));
                if (ob == -12345) {
#else
                if (ob >= 0) {
#endif
                    float* dst = a.out + ob + qc * 4;
                    if (flags & SPK_EPI_AFFINE) v = v * es + eh;
                    if (flags & SPK_EPI_ADD) {
                        f32x4 ad = *(const f32x4*)(a.epi_add + ob + qc * 4);
                        if (has_add_mask) {
                            const int ch0 = n0 + qc * 4;
                            const unsigned bits = a.add_mask[(size_t)((ob - n0) / a.Cout) * (a.Cout >> 5) + (ch0 >> 5)] >> (ch0 & 31);
#pragma unroll
                            for (int c = 0; c < 4; ++c) ad[c] = ((bits >> c) & 1u) ? ad[c] : 0.f;
                        }
                        v += ad;
                    }
                    if (flags & SPK_EPI_RELU) {
                        v[0] = fmaxf(v[0], 0.f);
                        v[1] = fmaxf(v[1], 0.f);
                        v[2] = fmaxf(v[2], 0.f);
                        v[3] = fmaxf(v[3], 0.f);
                    }
                    conv_st(dst, v);
                    out_mx = fmaxf(fmaxf(out_mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                    if (flags & SPK_EPI_BNBWD) {
                        // v is the gradient wrt a BatchNorm(+ReLU) output: accumulate (sum dz, sum dz*xhat) of that BN so
                        // its backward needs no separate reduction pass over this tensor
                        const f32x4 rw = *(const f32x4*)(a.bn_raw + ob + qc * 4);
                        f32x4 dz;
                        if (has_bn_mask) {
                            const int ch0 = n0 + qc * 4;
                            const unsigned bits = a.bn_mask[(size_t)((ob - n0) / a.Cout) * (a.Cout >> 5) + (ch0 >> 5)] >> (ch0 & 31);
#pragma unroll
                            for (int c = 0; c < 4; ++c) dz[c] = ((bits >> c) & 1u) ? v[c] : 0.f;
                        } else {
                            f32x4 m;
                            if (has_bn_act) m = *(const f32x4*)(a.bn_act + ob + qc * 4);
                            else m = rw * bsc + bsh;
#pragma unroll
                            for (int c = 0; c < 4; ++c) dz[c] = m[c] > 0.f ? v[c] : 0.f;
                        }
                        ssum += dz;
                        { const f32x4 xh_ = (rw - bmu) * bis; for (int c_ = 0; c_ < 4; ++c_) ssq[c_] = __builtin_fmaf(dz[c_], xh_[c_], ssq[c_]); }
                    } else {
                        ssum += v;
                        for (int c_ = 0; c_ < 4; ++c_) ssq[c_] = __builtin_fmaf(v[c_], v[c_], ssq[c_]);      // (explicit: the same rounding in every instantiation)
                    }
                }
            }
        }
    } else {
        // The epilogue's global READS (shortcut / masked gradient add, raw tensor and mask of the fused BatchNorm-backward
        // statistics) are issued branch-free for a whole batch of passes BEFORE the accumulators of the m-tile take their round
        // trip through the LDS slab, from clamped addresses (a pixel outside the grid reads the tile's first pixel - always inside
        // - and its value is dropped): one exposed memory latency per m-tile instead of a chain of one per pass (a data gradient
        // with both fusions has 8 passes x MT m-tiles of them per wave; measured round 3: 0.40 -> see DESIGN.md).  Results are
        // bit-identical to the pass-by-pass form (same values, same order of the per-lane statistics sums).
        constexpr int NP = 32 / RPP;                     // passes per m-tile
        // passes whose operands are in flight together: 4 x 11 registers, a budget that keeps the 32-channel (HBM-bound, occupancy-
        // sensitive) instantiations at their occupancy and the large tiles under their launch bound
        constexpr int PB = NP < 4 ? NP : 4;
        const int pix_safe = (b * a.OHf + oy0 * a.OS + a.ooy) * a.OWf + ox0 * a.OS + a.oox;      // the tile's first pixel
        const int cw32 = a.Cout >> 5, ch0 = n0 + qc * 4;
        const bool f_add = (flags & SPK_EPI_ADD) != 0, f_bnb = (flags & SPK_EPI_BNBWD) != 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            to_slab(i);
            // same-wave LDS traffic is ordered; the compiler inserts the lgkmcnt wait for the reads below
            if (!f_add && !f_bnb) {
                // no global reads in this epilogue (forward convolutions): pass by pass
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int row = k * RPP + qr;
                    const int ob = __shfl(obase[i], row, 64);
                    f32x4 v = *(const f32x4*)(slab + row * LW + qc * 4);
#ifdef ABL_NO_EPI
                    asm volatile("" ::"v"(v));
                    if (ob == -12345) {
#else
                    if (ob >= 0) {
#endif
                        if (flags & SPK_EPI_AFFINE) v = v * es + eh;
                        if (flags & SPK_EPI_RELU) {
                            v[0] = fmaxf(v[0], 0.f);
                            v[1] = fmaxf(v[1], 0.f);
                            v[2] = fmaxf(v[2], 0.f);
                            v[3] = fmaxf(v[3], 0.f);
                        }
                        conv_st(a.out + ob + qc * 4, v);
                        out_mx = fmaxf(fmaxf(out_mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                        ssum += v;
                        for (int c_ = 0; c_ < 4; ++c_) ssq[c_] = __builtin_fmaf(v[c_], v[c_], ssq[c_]);      // (explicit: the same rounding in every instantiation)
                    }
                }
                continue;
            }
#pragma unroll
            for (int k0 = 0; k0 < NP; k0 += PB) {
                int obv[PB];
                f32x4 adv[PB], rwv[PB];
                unsigned amw[PB], bmw[PB];
#pragma unroll
                for (int kk = 0; kk < PB; ++kk) {
                    const int row = (k0 + kk) * RPP + qr;
                    const int ob = __shfl(obase[i], row, 64);
                    const int px = __shfl(opix[i], row, 64);          // (shuffles stay outside any lane-dependent condition)
                    obv[kk] = ob;
                    const int pix = ob >= 0 ? px : pix_safe;
                    const int o = pix * a.Cout + ch0;                                     // element offset of this lane's quad
                    if (f_add) {
                        adv[kk] = conv_ld<2>(a.epi_add + o);
                        if (has_add_mask) amw[kk] = a.add_mask[(size_t)pix * cw32 + (ch0 >> 5)];
                    }
                    if (f_bnb) {
                        rwv[kk] = conv_ld<2>(a.bn_raw + o);
                        if (has_bn_mask) bmw[kk] = a.bn_mask[(size_t)pix * cw32 + (ch0 >> 5)];
                    }
                }
#pragma unroll
                for (int kk = 0; kk < PB; ++kk) {
                    const int row = (k0 + kk) * RPP + qr;
                    const int ob = obv[kk];
                    f32x4 v = *(const f32x4*)(slab + row * LW + qc * 4);
#ifdef ABL_NO_EPI
                    asm volatile("" ::"v"(v));
                    if (ob == -12345) {
#else
                    if (ob >= 0) {
#endif
                        float* dst = a.out + ob + qc * 4;
                        if (flags & SPK_EPI_AFFINE) v = v * es + eh;
                        if (f_add) {
                            f32x4 ad = adv[kk];
                            if (has_add_mask) {
                                const unsigned bits = amw[kk] >> (ch0 & 31);
#pragma unroll
                                for (int c = 0; c < 4; ++c) ad[c] = ((bits >> c) & 1u) ? ad[c] : 0.f;
                            }
                            v += ad;
                        }
                        if (flags & SPK_EPI_RELU) {
                            v[0] = fmaxf(v[0], 0.f);
                            v[1] = fmaxf(v[1], 0.f);
                            v[2] = fmaxf(v[2], 0.f);
                            v[3] = fmaxf(v[3], 0.f);
                        }
                        conv_st(dst, v);
                        out_mx = fmaxf(fmaxf(out_mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                        if (f_bnb) {
                            // v is the gradient wrt a BatchNorm(+ReLU) output: accumulate (sum dz, sum dz*xhat) of that BN so
                            // its backward needs no separate reduction pass over this tensor
                            const f32x4 rw = rwv[kk];
                            f32x4 dz;
                            if (has_bn_mask) {
                                const unsigned bits = bmw[kk] >> (ch0 & 31);
#pragma unroll
                                for (int c = 0; c < 4; ++c) dz[c] = ((bits >> c) & 1u) ? v[c] : 0.f;
                            } else {
                                f32x4 m;
                                if (has_bn_act) m = *(const f32x4*)(a.bn_act + ob + qc * 4);      // (activated tensor instead of mask bits: tests / tools)
                                else m = rw * bsc + bsh;
#pragma unroll
                                for (int c = 0; c < 4; ++c) dz[c] = m[c] > 0.f ? v[c] : 0.f;
                            }
                            ssum += dz;
                            { const f32x4 xh_ = (rw - bmu) * bis; for (int c_ = 0; c_ < 4; ++c_) ssq[c_] = __builtin_fmaf(dz[c_], xh_[c_], ssq[c_]); }
                        } else {
                            ssum += v;
                            for (int c_ = 0; c_ < 4; ++c_) ssq[c_] = __builtin_fmaf(v[c_], v[c_], ssq[c_]);      // (explicit: the same rounding in every instantiation)
                        }
                    }
                }
            }
        }
    }
    STAMP(9);
    // absmax of what this launch stored (the scale of the next f16x3 consumer of the tensor): one atomicMax per wave
    if (a.out_amax) spk_wave_amax_commit(out_mx, a.out_amax);
    if constexpr (BNBWD) {
        if (a.side_amax) spk_wave_amax_commit(side_mx, a.side_amax);
    }
    if (flags & SPK_EPI_STATS) {
        // lanes with equal qc hold the same 4 channels: fold them (lane strides Q, 2Q, ... < 64)
#pragma unroll
        for (int off = Q; off < 64; off <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ssum[c] += __shfl_xor(ssum[c], off, 64);
                ssq[c] += __shfl_xor(ssq[c], off, 64);
            }
        }
        if (lane < Q) {
            float* dst = a.stats + ((size_t)(ptile * 4 + wave) * a.Cout + n0 + lane * 4) * 2;
            *(f32x4*)dst = (f32x4){ssum[0], ssq[0], ssum[1], ssq[1]};
            *(f32x4*)(dst + 4) = (f32x4){ssum[2], ssq[2], ssum[3], ssq[3]};
        }
    }
}

template <int MT, int NT, bool BNBWD, int SPLIT, int FL = -1>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    conv_body<MT, NT, BNBWD, SPLIT, false, false, false, false, FL>(a);
}

// the in-wave pipelined form (f16x3 operands; conv_pipe.hip)
// BITS (fused BatchNorm backward only): the ReLU mask comes as sign bits (in_mask); otherwise it is recomputed from the raw
// conv output (in_act is not supported here: such launches stay on conv_mfma_kernel)
template <int MT, int NT, bool BNBWD, bool BITS = false, bool PRE = false, bool M16 = false, int FL = -1>
__global__ __launch_bounds__(256, 2) void conv_pipe_kernel(ConvArgs a) {        // two blocks per CU: 256 registers per lane
    conv_body<MT, NT, BNBWD, 3, true, BITS, PRE, M16, FL>(a);
}
