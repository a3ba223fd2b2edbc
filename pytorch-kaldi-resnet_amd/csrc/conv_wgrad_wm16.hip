// 3x3 weight gradient, f16x3 operands, 2 x 2 wave layout (as conv_wgrad_wm.hip) on v_mfma_f32_16x16x32_f16, with dY - an f16 pair
// tensor - brought into LDS by the DMA path (global_load_lds_dwordx4), never through registers.
//
// Why: conv_wgrad_wm_kernel runs at 85 % of what its bare K loop reaches on the 32x32x16 shape at the board's power cap
// (tools/probe/shape_probe.hip, two waves per SIMD: 1 232 TFLOP/s of fp16 issue = 0.277 ms for one launch's products; the kernel takes
// 0.32-0.33 ms) - the 16x16x32 shape holds a higher clock (1 545 TFLOP/s in the same probe).  A first 16x16x32 form of that kernel
// (round 4, not kept) spilled: a k-step of 32 pixels doubles the fragment registers (A 16 + B 16 instead of 8 + 8 + 8 for the double
// buffer) at the 256-register bound, and a spilled register is reloaded once per REGION here (64 pixels = 1.6 us of matrix
// instructions).  The registers come from the dY prefetch: a pair tensor is staged by plain copy, so its 16 KB per region go global ->
// LDS directly (two LDS buffers, the next region's tile in flight under the K loop of the current one) and the 16 prefetch registers
// and the dY half of the publish pass disappear.
//
// LDS image of dY (per buffer): chunks of 4 consecutive tile pixels x 64 channels = 1024 bytes = one wave-instruction of the DMA path
// (a lane's 16 bytes land at base + 16 lane).  A piece = the [4 x fp16 high][4 x fp16 low] of 4 channels of one pixel, as stored in
// the pair tensor.  Inside a chunk piece (pixel q, channel group c = 0..15) sits in slot (c >> 2) * 16 + (c & 3) * 4 + q: the sixteen
// lanes of one ds_read_b64_tr_b16 group read pixels q = 0..3 x channel groups (c & 3) = 0..3 of one 16-channel tile - sixteen
// different 16-byte slots of one 256-byte row: no bank conflict.  The DMA lane L therefore fetches pixel L & 3, channel group
// ((L >> 4) << 2) | ((L >> 2) & 3); pixels outside the image (ragged edge tiles, the padding of the last k-step) fetch a 16-byte block
// of zeros instead.
//
// Products and scales as conv_wgrad_wm_kernel; summation order differs (32 pixels per matrix instruction, the term-1 x term-0 product
// first): equal to the other weight-gradient kernels within fp32 accumulation error, not bit for bit.
#include "conv_wgrad.h"
#include <type_traits>
#ifndef WGRAD_XCD_BAND
#define WGRAD_XCD_BAND 1
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

static __device__ __forceinline__ s16x8 tr_read8w(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// (m0 is named as clobbered below so that the statement documents itself; clang warns that it does not track reserved registers. The
// statement sets m0 right before its only use and nothing else in this kernel reads m0 - checked in the generated code.)
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __attribute__((aligned(16))) unsigned g_wm16_zero[4];      // what the DMA lanes of pixels outside the image fetch

// X pixel pitch and the pixels of a 32-lane half.  A transposed read is banked per 32-lane half (bank = (address / 4) mod 64): the half
// reads 8 pixels x 32 bytes (4 lanes x 8 bytes of one 16-channel row tile of one term).  First layout (WM16_ODD_PITCH = 0): pixel pitch 384
// (192) bytes = 12 (6) x 32 and the half's pixels 0-3 and 8-11 of the k-step - pixel h of the halo starts in 32-byte bank group 4 h mod 8
// (6 h mod 8, and the second four pixels one tile row further: the same groups again): SQ_LDS_BANK_CONFLICT 9.2e7 of 1.30e8
// SQ_LDS_IDX_ACTIVE cycles per launch (3.1e7 of 7.2e7 in the 32-channel layout) - every X read took four (two) LDS cycles per half
// instead of one (profiles/r04_sq_counters/final_kernels_*).  Now: an odd number of 32-byte units per pixel (416 = 13 x 32 bytes; 160 =
// 5 x 32 in the 32-channel layout) and the half reads 8 CONSECUTIVE tile pixels (lane group kg holds pixels 16 (kg >> 1) + 4 (kg & 1) + q
// and + 8 of the step; dY chunks follow the same assignment) - one tile row when the tile is 8 or 16 wide: eight different bank groups.
// (dY keeps its two-way conflict - 8 of the 80 reads of a k-step: both halves of a 16-byte piece cannot be read by one half-wave of
// lanes that all want the same term.)
#ifndef WM16_ODD_PITCH
#define WM16_ODD_PITCH 1
#endif
// WM16_ROW_SKEW (bytes added per halo ROW, default 0 = not built into the library): tiles that are not 8 or 16 pixels wide - 10 x 6 in the
// 256-channel layer - put the 8 consecutive pixels of a half-wave on two tile rows of a halo 8 pixels wide, where two of them meet occupied
// bank groups whatever the pixel pitch; 192 bytes per halo row move them onto the two groups a 6-pixel row leaves free
// (tools/lds_bank_model.py: 1.9 -> 1.0 x the conflict-free cycles on 10 x 6, 1.0 stays 1.0 on 8 x 8 and 4 x 16).  A variant build for the
// A/B (tools/variant.sh skew conv_wgrad_wm16.hip -DWM16_ROW_SKEW=192); the default code is unchanged by the macro.
#ifndef WM16_ROW_SKEW
#define WM16_ROW_SKEW 0
#endif
#define WM16_PX_WIDE (WM16_ODD_PITCH ? 416 : 384)       /* X bytes per staged pixel, 2 x 2 wave layout: [2 groups][3-term pitch][32 ch fp16] (+ 32) */
#define WM16_PX_C32 (WM16_ODD_PITCH ? 160 : 192)        /* ... 32-channel layout: [2 terms][32 ch fp16] (+ 32) */
#ifndef WM16_CVT_IN_LOOP
#define WM16_CVT_IN_LOOP 0      // 1: the conversion of the next region's X dealt out between the matrix instructions of the last k-step
#endif                          //    (built and measured +-0 inside the step - DESIGN.md section 7b; the default is the plain pass after the K loop)
// VAR: bit 0 = fused input BatchNorm + ReLU on X (compile-time, as conv_wgrad_wm_kernel); dY is always an f16 pair tensor.
// C32: the layout for 32-channel groups (the first layer: Cin = Cout = 32).  A block owns 32 x 32 channels, every wave holds the whole
// 9 x 32 x 32 tile and the four waves split the k-steps of a region (wave w takes steps w, w + 4, ...); at the end of the block the four
// tiles are folded through LDS in a fixed order (a first version wrote four slabs per block and left the fold to the slab reduce: 2 048
// slabs made that kernel - 36 blocks, a serial walk over the slabs - 0.14 ms instead of 0.02).  X pixels are 128 bytes (8 threads, 32
// pixels per staging pass), a DMA chunk is 8 pixels x 8 channel groups: piece (pixel p8, group c) in slot (p8 >> 2) * 32 + (c >> 2) * 16 +
// (c & 3) * 4 + (p8 & 3) - again sixteen different 16-byte slots of a 256-byte row for the sixteen lanes of a transposed read.
template <int VAR, bool C32>
static __device__ __forceinline__ void wgrad_m16_body(const WgradArgs& a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    constexpr int NTAPS = 9, KS = 3;
    constexpr int CGL = C32 ? 5 : 6;                      // log2 of the channels per block side
    constexpr int PX = C32 ? WM16_PX_C32 : WM16_PX_WIDE;  // X bytes per staged pixel
    constexpr int QXL = C32 ? 3 : 4;                      // log2 of the threads (float4) per X pixel
    constexpr int PXP = 256 >> QXL;                       // X pixels per staging pass
    constexpr int WM16_NX = C32 ? 6 : 7;                  // X halo float4 per thread: halo_pix <= PXP * WM16_NX (192 / 112 pixels)
    constexpr int CPL = C32 ? 3 : 2;                      // log2 of the pixels per DMA chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = C32 ? 0 : wave >> 1, wn = C32 ? 0 : wave & 1;
    int g, ci0, co0;                                      // block -> (region slice, channel groups): conv_wgrad_split.hip, wgrad_block
    {
        const int ncgi = a.Cin >> CGL, M = ncgi * (a.Cout >> CGL);
        const int bid = blockIdx.x;
        int m;
        if ((a.nsplit & 7) == 0) {
            const int k = bid >> 3;
            m = k % M;
            g = WGRAD_XCD_BAND ? (bid & 7) * (a.nsplit >> 3) + k / M : (k / M) * 8 + (bid & 7);
        } else {
            m = bid % M;
            g = bid / M;
        }
        ci0 = (m % ncgi) << CGL;
        co0 = (m / ncgi) << CGL;
    }
    const int halo_pix = a.halo_h * a.halo_w;
    const int npix = a.TH * a.TW;
    const int nsteps = (npix + 31) >> 5;                  // k-steps of 32 pixels; the padding pixels of dY are zero
    const int nchunks = nsteps << (5 - CPL);              // DMA chunks (1024 bytes: 4 pixels x 64 channels or 8 x 32) of a dY buffer
    unsigned char* xs = ldsb;
    unsigned char* dys = ldsb + halo_pix * PX + a.halo_h * WM16_ROW_SKEW;      // two buffers of nchunks * 1024 bytes
    const int dbuf = nchunks << 10;
    const float sig_x = spk_sigma_from_amax_bits(*a.x_amax);
    const float sig_d = spk_sigma_from_amax_bits(*a.dy_amax);

    f32x4 acc[NTAPS][2][2];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int q16 = tid & ((1 << QXL) - 1);               // this thread's float4 of the 64 (32) channels of an X pixel
    f32x4 px[WM16_NX];
    unsigned inx = 0;
    const unsigned x_row = (unsigned)a.IW * a.Cin * 4u, x_px = (unsigned)a.Cin * 4u;
    const unsigned d_row = (unsigned)a.OW * a.Cout * 4u, d_px = (unsigned)a.Cout * 4u;
    const unsigned x_c = (unsigned)(ci0 + q16 * 4) * 4u;
    const unsigned d_c = (unsigned)(co0 + (((((lane >> 4) & (C32 ? 1 : 3)) << 2) | ((lane >> 2) & 3)) << 2)) * 4u;     // the DMA lane's channel group
    const int d_pl = C32 ? ((lane >> 5) << 2) | (lane & 3) : lane & 3;                                              // ... and pixel of the chunk
    auto region_origin = [&](int region, int& b, int& oy0, int& ox0) {
        int pt = region;
        const int tx = pt % a.tiles_x;
        pt /= a.tiles_x;
        const int ty = pt % a.tiles_y;
        b = pt / a.tiles_y;
        oy0 = ty * a.TH;
        ox0 = tx * a.TW;
    };
    auto prefetch_x = [&](int region) {
        int b, oy0, ox0;
        region_origin(region, b, oy0, ox0);
        int t4 = tid >> QXL;
        asm volatile("" : "+v"(t4));      // opaque: the halo coordinates of the seven items are recomputed per region, not held in registers across the K loop
        const int iy0 = oy0 * a.S - a.pad, ix0 = ox0 * a.S - a.pad;
        const char* xb = (const char*)(a.x + (size_t)b * a.IH * a.IW * a.Cin);
        const unsigned x_safe = (unsigned)(oy0 * a.S) * x_row + (unsigned)(ox0 * a.S) * x_px + x_c;
        inx = 0;
#pragma unroll
        for (int u = 0; u < WM16_NX; ++u) {
            const int p = t4 + PXP * u;
            const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
            const int hx = p - hy * a.halo_w;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW && p < halo_pix;
            if (ok) inx |= 1u << u;
            const unsigned off = ok ? (unsigned)iy * x_row + (unsigned)ix * x_px + x_c : x_safe;
            px[u] = wgrad_ld<2>(xb + off);
        }
    };
    // the dY tile of a region, global -> LDS buffer `buf`: wave w brings chunks w, w + 4, ...
    auto dma_dy = [&](int region, int buf) {
        int b, oy0, ox0;
        region_origin(region, b, oy0, ox0);
        const char* db = (const char*)(a.dy + (size_t)b * a.OH * a.OW * a.Cout);
        unsigned char* dst = dys + buf * dbuf;
        for (int ch = wave; ch < nchunks; ch += 4) {
            const int p = (ch << CPL) + d_pl;
            const int ly = (int)__umulhi((unsigned)p, a.tw_magic);
            const int lx = p - ly * a.TW;
            const int oy = oy0 + ly, ox = ox0 + lx;
            const bool ok = p < npix && oy < a.OH && ox < a.OW;
            const char* src = ok ? db + ((unsigned)oy * d_row + (unsigned)ox * d_px + d_c) : (const char*)g_wm16_zero;
            // (as inline assembly: the compiler models the builtin as a store to LDS through the vector-memory counter and puts s_waitcnt
            // vmcnt(0) in front of every later LDS read - the fragment reads of the CURRENT buffer at the top of each k-step - which waits
            // for the whole prefetch.  The buffers alternate; the vmcnt(0) that matters is the explicit one before the barrier below.
            // Vector-memory operations return in order, so the compiler's own vmcnt arithmetic for the X loads stays on the safe side.)
            const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(dst + (ch << 10));
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_addr), "v"(src) : "memory", "m0");
        }
    };
    // X goes through registers (fp32 activations, optionally BatchNorm + ReLU on the way, two fp16 terms).  The conversion works IN PLACE -
    // a float4 becomes the [4 x high][4 x low] pair of the staged value, the same four registers - right after the K loop, so that only the
    // LDS writes are left between the two barriers.  WM16_CVT_IN_LOOP = 1 deals it out between the matrix instructions of the last k-step
    // instead (vector instructions overlap matrix instructions only inside one wave's instruction stream): the
    // instructions interleave as intended in the generated code, and the step does not move (48.62 / 48.71 against 48.78 / 48.65 ms) -
    // at the power cap a vector instruction costs what it costs wherever it issues.
    auto convert_item = [&](int u, const f32x4& sc, const f32x4& sh) {
        f32x4 w = px[u];
        if constexpr ((VAR & 1) != 0) {
            w = w * sc + sh;
            w[0] = fmaxf(w[0], 0.f);
            w[1] = fmaxf(w[1], 0.f);
            w[2] = fmaxf(w[2], 0.f);
            w[3] = fmaxf(w[3], 0.f);
        }
        if (!((inx >> u) & 1)) w = (f32x4){0.f, 0.f, 0.f, 0.f};
        uint2 t0, t1;
        split2h(w, sig_x, t0, t1);
        px[u] = spk_pair_pack(t0, t1);
    };
    auto load_affine = [&](f32x4& sc, f32x4& sh) {
        if constexpr ((VAR & 1) != 0) {
            sc = *(const f32x4*)(a.in_scale + ci0 + q16 * 4);
            sh = *(const f32x4*)(a.in_shift + ci0 + q16 * 4);
        }
    };
    auto publish_x = [&]() {
#pragma unroll
        for (int u = 0; u < WM16_NX; ++u) {
            const int p = (tid >> QXL) + PXP * u;
            if (p < halo_pix) {
                unsigned char* pp = xs + p * PX + (q16 >> 3) * 192;
                if constexpr (WM16_ROW_SKEW != 0) pp += (int)__umulhi((unsigned)p, a.halo_w_magic) * WM16_ROW_SKEW;
                uint2* dst = (uint2*)pp + (q16 & 7);
                uint2 t0, t1;
                spk_pair_unpack(px[u], t0, t1);
                dst[0] = t0;
                dst[8] = t1;
            }
        }
    };

    // 16 x 16 x 32 operands: lane = row (A: input channel) or column (B: output channel) lane & 15, pixels 8 (lane >> 4) .. + 7 of the step.
    // One transposed read hands a lane 4 pixels of its channel from the 16 lanes (pixel q, 4-channel piece p4) of its group.
    const int kg = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const int a_lane = wm * 192 + p4 * 8;                                  // X: [group wm][term][32 ch]: row tile rt at + 32 rt, term s at + 64 s
    // dY: pixels 8 kg + 4 blk + q of the step: chunk 2 kg + blk, slot (2 wn + ct) * 16 + 4 p4 + q (C32: chunk kg, slot 32 blk + 16 ct + 4 p4 + q); term s at + 8 s
    // (WM16_ODD_PITCH: pixels 16 (kg >> 1) + 8 blk + 4 (kg & 1) + q: chunk 4 (kg >> 1) + 2 blk + (kg & 1); C32: chunk 2 (kg >> 1) + blk, slot 32 (kg & 1) + ...)
    const int d_lane = (WM16_ODD_PITCH ? (C32 ? 2 * (kg >> 1) * 1024 + (kg & 1) * 512 : (4 * (kg >> 1) + (kg & 1)) * 1024) : (C32 ? kg : 2 * kg) * 1024) +
                       wn * 512 + p4 * 64 + q * 16;
    constexpr int D_STEP = C32 ? 4096 : 8192, D_BLK = WM16_ODD_PITCH ? (C32 ? 1024 : 2048) : (C32 ? 512 : 1024);

    int region = g, buf = 0;
    if (region < a.nregions) {
        dma_dy(region, 0);                                // (before the X loads: older operations complete first, the compiler's counts for px stay exact)
        prefetch_x(region);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        load_affine(sc, sh);
#pragma unroll
        for (int u = 0; u < WM16_NX; ++u) convert_item(u, sc, sh);      // the first region's conversion has no K loop to hide under
    }
    for (; region < a.nregions; region += a.nsplit, buf ^= 1) {
        __syncthreads();                                  // every wave is past the previous region's K loop
        publish_x();
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): this region's dY tile has landed in LDS
        __syncthreads();
        const bool has_next = region + a.nsplit < a.nregions;
        if (has_next) {
            dma_dy(region + a.nsplit, buf ^ 1);           // last read by the K loop of the region before this one
            prefetch_x(region + a.nsplit);
        }
        const unsigned char* dcur = dys + buf * dbuf + d_lane;
        // (always_inline: as a call the body would take the accumulators through memory)
        auto kstep = [&](int j, auto cvt_c) __attribute__((always_inline)) {
            constexpr bool CVT = decltype(cvt_c)::value;      // the last k-step of a region carries the conversion of the next region's X
            s16x8 bf[2][2];                               // [column tile][term]
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int s = 0; s < 2; ++s) bf[ct][s] = tr_read8w(dcur + j * D_STEP + ct * 256 + s * 8, dcur + j * D_STEP + D_BLK + ct * 256 + s * 8);
            int xa[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int pix = j * 32 + (WM16_ODD_PITCH ? 16 * (kg >> 1) + 8 * blk + 4 * (kg & 1) : 8 * kg + 4 * blk) + q;
                const int pc = pix < npix ? pix : npix - 1;
                const int ly = (int)__umulhi((unsigned)pc, a.tw_magic);
                const int lx = pc - ly * a.TW;
                xa[blk] = ((ly * a.S) * a.halo_w + lx * a.S) * PX + (ly * a.S) * WM16_ROW_SKEW + a_lane;
            }
            s16x8 af[2][2];                               // [term][row tile]
            auto load_a = [&](int s, int t) {
                const int toff = ((t / KS) * a.halo_w + (t % KS)) * PX + (t / KS) * WM16_ROW_SKEW + s * 64;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) af[s][rt] = tr_read8w(xs + xa[0] + toff + rt * 32, xs + xa[1] + toff + rt * 32);
            };
            auto mm = [&](int t, int sa, int sb) {        // four independent accumulators in rotation
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[t][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[sa][rt]), __builtin_bit_cast(f16x8, bf[ct][sb]),
                                                                                acc[t][rt][ct], 0, 0, 0);
            };
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if constexpr (CVT) load_affine(sc, sh);
            load_a(1, 0);
            load_a(0, 0);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                // term 1 of A is used by one product only: its registers take the next tap's term 1 under the other two products,
                // term 0 follows under the next tap's first product
                __builtin_amdgcn_sched_barrier(0);
                mm(t, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < NTAPS) load_a(1, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                mm(t, 0, 1);
                if constexpr (CVT) {                                           // the next region's X, one item per tap, in the last taps: its
                    if (t >= NTAPS - WM16_NX) convert_item(t - (NTAPS - WM16_NX), sc, sh);      // loads have had the whole region to arrive.
                }                                                              // No branch: one scheduling region with the eight products around it
                mm(t, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < NTAPS) load_a(0, t + 1);
            }
        };
        if constexpr (C32) {
            for (int j = wave; j < nsteps; j += 4) kstep(j, std::false_type{});       // the four waves split the k-steps
        } else {
            for (int j = 0; j + 1 < nsteps; ++j) kstep(j, std::false_type{});
            // (also after a block's last region, where it converts stale registers: harmless, and the loop keeps two bodies instead of three)
            kstep(nsteps - 1, std::integral_constant<bool, WM16_CVT_IN_LOOP != 0>{});
        }
        if constexpr (C32 || !WM16_CVT_IN_LOOP) {
            if (has_next) {
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                load_affine(sc, sh);
#pragma unroll
                for (int u = 0; u < WM16_NX; ++u) convert_item(u, sc, sh);
            }
        }
    }

    if constexpr (C32) {
        // fold the four waves' tiles (each a partial sum over its k-steps) through LDS in a fixed order: waves 1-3 write three taps at a
        // time (12 float4 per lane, 36 KB), wave 0 adds them - w1, w2, w3 - and writes the block's one slab
        f32x4* red = (f32x4*)ldsb;
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            __syncthreads();                              // (first round: every wave is past its last K loop)
            if (wave > 0) {
#pragma unroll
                for (int i = 0; i < 12; ++i) red[((wave - 1) * 12 + i) * 64 + lane] = acc[3 * tt + i / 4][(i >> 1) & 1][i & 1];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 0; w < 3; ++w)
#pragma unroll
                    for (int i = 0; i < 12; ++i) acc[3 * tt + i / 4][(i >> 1) & 1][i & 1] += red[(w * 12 + i) * 64 + lane];
            }
        }
        if (wave != 0) return;
    }
    float* slab = a.partial + (size_t)g * NTAPS * a.Cin * a.Cout;
    const float inv = (1.f / sig_x) * (1.f / sig_d);
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = rt * 16 + 4 * kg + e, col = ct * 16 + (lane & 15);       // 16 x 16 tile: col = lane & 15, row = 4 (lane >> 4) + e
                    slab[((size_t)t * a.Cin + ci0 + wm * 32 + row) * a.Cout + co0 + wn * 32 + col] = acc[t][rt][ct][e] * inv;
                }
}

// (two kernel names, one body: profiles and labels tell the layouts apart)
template <int VAR>
__global__ __launch_bounds__(256, 2) void conv_wgrad_wm16_kernel(WgradArgs a) { wgrad_m16_body<VAR, false>(a); }
template <int VAR>
__global__ __launch_bounds__(256, 2) void conv_wgrad_c32m16_kernel(WgradArgs a) { wgrad_m16_body<VAR, true>(a); }

int spk_launch_wgrad_wm16(const WgradArgs& a, hipStream_t st) {
    SPK_REQUIRE(a.KW == 3 && a.Cin % 64 == 0 && a.Cout % 64 == 0, "spk_conv_wgrad(2x2 waves, 16x16x32): 3x3, Cin and Cout multiples of 64 (%d, %d)", a.Cin, a.Cout);
    SPK_REQUIRE(a.flags & SPK_DY_PRESPLIT, "spk_conv_wgrad(2x2 waves, 16x16x32): dy must be an f16 pair tensor (SPK_DY_PRESPLIT): it is staged by the DMA path");
    SPK_REQUIRE(a.halo_h * a.halo_w <= 16 * 7, "spk_conv_wgrad(2x2 waves, 16x16x32): tile %dx%d (halo %dx%d) exceeds the prefetch window", a.TH, a.TW,
                a.halo_h, a.halo_w);
    SPK_REQUIRE((long long)a.OH * a.OW * a.Cout * 4 < 0x7fffffffLL && (long long)a.IH * a.IW * a.Cin * 4 < 0x7fffffffLL,
                "spk_conv_wgrad(2x2 waves, 16x16x32): an image exceeds 32-bit byte offsets");
    const size_t lds_bytes = (size_t)a.halo_h * a.halo_w * WM16_PX_WIDE + (size_t)a.halo_h * WM16_ROW_SKEW + 2 * (size_t)(((a.TH * a.TW + 31) >> 5) << 5) * 256;
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_wgrad(2x2 waves, 16x16x32): tile %dx%d needs %zu B of LDS", a.TH, a.TW, lds_bytes);
    dim3 grid(a.nsplit * (a.Cin / 64) * (a.Cout / 64));
    if (a.flags & SPK_IN_AFFINE_RELU) hipLaunchKernelGGL(conv_wgrad_wm16_kernel<1>, grid, dim3(256), lds_bytes, st, a);
    else hipLaunchKernelGGL(conv_wgrad_wm16_kernel<0>, grid, dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_wgrad(2x2 waves, 16x16x32)");
    return 0;
}

// the 32-channel-group layout
int spk_launch_wgrad_c32m16(const WgradArgs& a, hipStream_t st) {
    SPK_REQUIRE(a.KW == 3 && a.Cin % 32 == 0 && a.Cout % 32 == 0, "spk_conv_wgrad(32-channel groups, 16x16x32): 3x3, channels multiples of 32 (%d, %d)", a.Cin, a.Cout);
    SPK_REQUIRE(a.flags & SPK_DY_PRESPLIT, "spk_conv_wgrad(32-channel groups, 16x16x32): dy must be an f16 pair tensor (SPK_DY_PRESPLIT): it is staged by the DMA path");
    SPK_REQUIRE(a.halo_h * a.halo_w <= 32 * 6, "spk_conv_wgrad(32-channel groups, 16x16x32): tile %dx%d (halo %dx%d) exceeds the prefetch window", a.TH, a.TW,
                a.halo_h, a.halo_w);
    SPK_REQUIRE((long long)a.OH * a.OW * a.Cout * 4 < 0x7fffffffLL && (long long)a.IH * a.IW * a.Cin * 4 < 0x7fffffffLL,
                "spk_conv_wgrad(32-channel groups, 16x16x32): an image exceeds 32-bit byte offsets");
    size_t lds_bytes = (size_t)a.halo_h * a.halo_w * WM16_PX_C32 + (size_t)a.halo_h * WM16_ROW_SKEW + 2 * (size_t)(((a.TH * a.TW + 31) >> 5) << 5) * 128;
    if (lds_bytes < 3 * 12 * 64 * 16) lds_bytes = 3 * 12 * 64 * 16;      // the fold of the four waves' tiles at the end of a block
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_wgrad(32-channel groups, 16x16x32): tile %dx%d needs %zu B of LDS", a.TH, a.TW, lds_bytes);
    dim3 grid(a.nsplit * (a.Cin / 32) * (a.Cout / 32));
    if (a.flags & SPK_IN_AFFINE_RELU) hipLaunchKernelGGL(conv_wgrad_c32m16_kernel<1>, grid, dim3(256), lds_bytes, st, a);
    else hipLaunchKernelGGL(conv_wgrad_c32m16_kernel<0>, grid, dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_wgrad(32-channel groups, 16x16x32)");
    return 0;
}
