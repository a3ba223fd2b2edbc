// Streaming 3x3 forward convolution of the 32-channel layer (32 -> 32 channels, stride 1, padding 1, f16x3 operands): conv1 / conv2
// of the three BasicBlocks of layer 1 (reference scripts/model.py:12-15,48-64; the Bottleneck trunk has three more).
//
// Layer 1 carries the largest tensors of the network (786 MB each at batch 256 x 80 x 300): its convolutions are HBM-side work, and
// the general kernel (conv_kernel.h) reaches 3 TB/s on them - its staging phases are 44 % of a block's life, its 97 k partial
// statistics rows cost another pass (profiles/r04_conv_stamps.log).  Same recipe as conv1x1_stream.hip, with a halo:
//   * persistent blocks walk 8 x 16-pixel tiles (x fastest, so neighbouring halos meet in L2); the 10 x 18 halo of all 32 channels
//     is staged once per tile - fused BatchNorm + ReLU, fp16 split - from eight 16-byte loads per pixel (a whole 128-byte line),
//     six per thread, issued one tile ahead and in flight through the previous tile's matrix instructions and epilogue;
//   * the geometry of a staging item (halo row / column, global and LDS offsets) is a per-thread constant computed once; tiles
//     that touch no image border skip every bounds test;
//   * the 9 x 32 x 32 weights live in LDS as matrix-core B fragments for the life of the block (36 KB, read with immediate
//     offsets); the K loop is 9 taps x (4 A reads + 4 B reads + 6 matrix instructions), fully unrolled, no address arithmetic;
//     halo rows are padded to a multiple of 256 bytes so that the ds_read_b128 of a 2 x 16-pixel row tile is conflict-free;
//   * stores straight from the accumulator layout (a lane = one channel of 16 pixels; a wave-instruction writes two whole lines);
//     statistics as per-lane sums kept in fp64 across tiles: 4 partial rows per block instead of one per wave and tile.
// Same arithmetic as the general kernel (terms under the same slot, three cross products per tap and 16-channel group in the same
// order); the statistics partials sum in another order.
#include "spk_common.h"

struct Conv3x3C32Args {
    const float* in;          // [B][H][W][32]
    const float* wpk;         // f16x3 pack of the forward weights behind its header: [tap][2 groups][2 terms][64 lanes][8 fp16]
    const unsigned* w_amax;
    float* out;               // [B][H][W][32] raw conv output
    const float* in_scale;    // SPK_IN_AFFINE_RELU
    const float* in_shift;
    float* stats;             // [gridDim.x * 4][32][2] (sum, sum of squares) partial rows
    const unsigned* in_amax;
    unsigned* out_amax;
    int B, H, W, tiles_x, tiles_y, ntiles, flags;
};

#define C32_TH 8
#define C32_TW 16
#define C32_HW 18                    // halo width
#define C32_HP (10 * 18)             // halo pixels
#define C32_PIX 144                  // LDS bytes per halo pixel: [2 groups][2 terms][16 ch fp16] + 16
#define C32_ROW 2816                 // LDS bytes per halo row: 18 x 144 = 2592, padded to 11 x 256 (conflict-free fragment reads)
#define C32_WBYTES (9 * 2 * 2 * 1024)
#ifndef C32_KD
#define C32_KD 2                     // K-loop steps whose fragment reads are in flight ahead of the matrix instructions
#endif
#ifndef C32_DEPTH
#define C32_DEPTH 2                  // tiles in flight (register sets of six 16-byte loads): 1 measured 0.498 ms per launch, 2 0.438
#endif

#ifdef C32_STAMPS
// diagnostic build (tools/variant.sh c32stamps conv3x3_c32_stream.hip -DC32_STAMPS): per-block sums of the shader-clock time wave 0
// spends in each phase of a tile; read back by spk_debug_stamps_c32 (tools/c32_stamps.py)
static __device__ unsigned long long g_c32_stamps[1024][8];
#define C32_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[i] += now_ - last; last = now_; } while (0)
#else
#define C32_T(i) do { } while (0)
#endif

// VAR: 1 = fused input BatchNorm + ReLU, 0 = plain input (compile-time: the staging loop has no branch on it)
#ifndef C32_XCD_BAND
#define C32_XCD_BAND 1
#endif
template <int VAR>
__global__ __launch_bounds__(256, 2) void conv3x3_c32_stream_kernel(Conv3x3C32Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    unsigned char* wl = lds3;                          // weights: [tap][group][term][64 lanes][16 B]
    unsigned char* hl = lds3 + C32_WBYTES;             // halo tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    constexpr bool AFF = VAR != 0;
    const float sig = spk_sigma_from_amax_bits(*a.in_amax);
    const float inv_sig = 1.f / sig, inv_wsig = 1.f / spk_sigma_from_amax_bits(*a.w_amax);      // (two factors: their product may leave the fp32 range)
    const int H = a.H, W = a.W;

    // weights -> LDS once (the packed order IS the fragment order: 36 x 1 KB)
    for (int i = tid; i < C32_WBYTES / 16; i += 256) *(f32x4*)(wl + i * 16) = *(const f32x4*)(a.wpk + i * 4);

    // staging items of this thread: float4 q of the 32 channels of halo pixel (tid / 8) + 32 u, u < 6 (180 pixels: the last pass is partial)
    const int q = tid & 7, prow = tid >> 3;
    int hy[6], hx[6];
    unsigned goff[6], loff[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int p = prow + 32 * u;
        const int pc = p < C32_HP ? p : C32_HP - 1;
        hy[u] = pc / C32_HW;
        hx[u] = pc - hy[u] * C32_HW;
        goff[u] = (unsigned)((hy[u] * W + hx[u]) * 32 + q * 4);                    // elements from the halo's origin pixel
        loff[u] = (unsigned)(hy[u] * C32_ROW + hx[u] * C32_PIX + (q >> 2) * 64 + (q & 3) * 8);
    }
    const bool item5 = prow + 32 * 5 < C32_HP;                                         // the partial sixth pass
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (AFF) {
        sc = *(const f32x4*)(a.in_scale + q * 4);
        sh = *(const f32x4*)(a.in_shift + q * 4);
    }

    // two tiles in flight: the loads of tile t + 2 are issued while tile t computes (one tile ahead was measured short of the HBM
    // latency under load with two blocks per CU: 0.498 ms per launch)
    f32x4 va[6], vb[6], vc[6];
    unsigned okma = 0, okmb = 0, okmc = 0;               // bit u: item u of the tile in flight lies inside the image
    auto tile_origin = [&](int tile, int& b, int& y0, int& x0) {
        const int tx = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty = t2 % a.tiles_y;
        b = t2 / a.tiles_y;
        y0 = ty * C32_TH;
        x0 = tx * C32_TW;
    };
    auto issue = [&](int tile, f32x4* v, unsigned& okm) {
        int b, y0, x0;
        tile_origin(tile, b, y0, x0);
        const bool edge = y0 == 0 || y0 + C32_TH >= H || x0 == 0 || x0 + C32_TW >= W;     // wave-uniform
        // element offset of the halo's origin pixel (y0 - 1, x0 - 1) of image b; may be "negative" for border tiles: their
        // out-of-image items are redirected to the tile's first pixel, which always exists
        const int base = ((b * H + y0 - 1) * W + x0 - 1) * 32;
        const int safe = ((b * H + y0) * W + x0) * 32 + q * 4;
        okm = 0x3f;
        if (!edge) {
#pragma unroll
            for (int u = 0; u < 6; ++u) v[u] = *(const f32x4*)(a.in + (unsigned)(base + (int)goff[u]));
        } else {
            okm = 0;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int iy = y0 - 1 + hy[u], ix = x0 - 1 + hx[u];
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                okm |= (unsigned)ok << u;
                v[u] = *(const f32x4*)(a.in + (unsigned)(ok ? base + (int)goff[u] : safe));
            }
        }
    };

    double s_sum = 0.0, s_sq = 0.0;
    float out_mx = 0.f;
    // this lane's A-fragment base: pixel r of the wave's 2 x 16 row tile (rows 2 wave, 2 wave + 1 of the tile), channel half h
    const unsigned char* rd = hl + (2 * wave + (r >> 4)) * C32_ROW + (r & 15) * C32_PIX + h * 16;
    const unsigned char* wb = wl + lane * 16;

    const int G = (int)gridDim.x;
#ifdef C32_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = __builtin_amdgcn_s_memtime();
    const unsigned long long born = last;
#endif
    auto step = [&](int tile, f32x4* v, unsigned& okm) {
        int b, y0, x0;
        tile_origin(tile, b, y0, x0);
        C32_T(0);
        __syncthreads();                                 // every wave has read the previous tile (and, first time, the weights are written)
        C32_T(1);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
#ifdef C32_ABL_NOSTAGE
            break;
#endif
            if (u == 5 && !item5) break;
            f32x4 w = v[u];
            if constexpr (AFF) {
                w = w * sc + sh;
                w[0] = fmaxf(w[0], 0.f);
                w[1] = fmaxf(w[1], 0.f);
                w[2] = fmaxf(w[2], 0.f);
                w[3] = fmaxf(w[3], 0.f);
            }
            if (!((okm >> u) & 1u)) w = (f32x4){0.f, 0.f, 0.f, 0.f};
            uint2 t0, t1;
            split2h(w, sig, t0, t1);
            *(uint2*)(hl + loff[u]) = t0;
            *(uint2*)(hl + loff[u] + 32) = t1;
        }
        C32_T(2);
        __syncthreads();
        C32_T(3);
#ifndef C32_ABL_NOLOAD   // (C32_ABL_*: timing-only diagnostic builds with wrong results - tools/gpu/c32_abl.sh)
        if (tile + C32_DEPTH * G < a.ntiles) issue(tile + C32_DEPTH * G, v, okm);  // in flight through the next tiles' K loops and epilogues
#endif
        C32_T(4);

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // (group-major like the general kernel, which walks the two 16-channel planes one after the other: the same accumulation
        //  order, bit-identical outputs)
#ifdef C32_ABL_HALFK
        constexpr int NG = 1;
#else
        constexpr int NG = 2;
#endif
        // software-pipelined by hand: the fragments of step s + C32_KD are requested before the matrix instructions of step s (the
        // compiler left every step's four reads directly in front of its products: 18 exposed LDS latencies per tile)
        constexpr int NS = NG * 9;
        f32x4 fr[C32_KD + 1][4];
        auto frag = [&](int s2, f32x4* f) {
            const int g = s2 / 9, t = s2 % 9;
            const int toff = (t / 3) * C32_ROW + (t % 3) * C32_PIX;
            f[0] = *(const f32x4*)(rd + toff + g * 64);
            f[1] = *(const f32x4*)(rd + toff + g * 64 + 32);
            f[2] = *(const f32x4*)(wb + ((t * 2 + g) * 2 + 0) * 1024);
            f[3] = *(const f32x4*)(wb + ((t * 2 + g) * 2 + 1) * 1024);
        };
#pragma unroll
        for (int s2 = 0; s2 < C32_KD; ++s2) frag(s2, fr[s2]);
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (s2 + C32_KD < NS) frag(s2 + C32_KD, fr[(s2 + C32_KD) % (C32_KD + 1)]);
            __builtin_amdgcn_sched_barrier(0);           // (the scheduler would sink these reads back in front of their use)
            const f32x4* f = fr[s2 % (C32_KD + 1)];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[0]), __builtin_bit_cast(f16x8, f[3]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[1]), __builtin_bit_cast(f16x8, f[2]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[0]), __builtin_bit_cast(f16x8, f[2]), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        C32_T(5);
        // epilogue: register e = pixel m = (e & 3) + 8 (e >> 2) + 4 h of the row tile -> tile row 2 wave + (m >> 4), column m & 15; channel r
        const bool ragged = y0 + C32_TH > H || x0 + C32_TW > W;
        float ts = 0.f, tq = 0.f;
        float* ob = a.out + (size_t)((b * H + y0 + 2 * wave) * W + x0) * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = e >> 3, lx = (e & 3) + 8 * ((e >> 2) & 1) + 4 * h;
            if (ragged && (y0 + 2 * wave + row >= H || x0 + lx >= W)) continue;
            const float val = acc[e] * inv_sig * inv_wsig;
#ifndef C32_ABL_NOSTORE
            __builtin_nontemporal_store(val, ob + (row * W + lx) * 32);
#endif
            out_mx = fmaxf(out_mx, fabsf(val));
            ts += val;
            tq = __builtin_fmaf(val, val, tq);
        }
        s_sum += (double)ts;
        s_sq += (double)tq;
        C32_T(6);
    };
    // Virtual block id: block v walks the tiles v, v + G, ...  With C32_XCD_BAND the G / 8 blocks that run on one XCD (the hardware
    // deals block ids to the eight XCDs in turn) take a contiguous band of virtual ids, i.e. G / 8 neighbouring tiles at a time: the
    // halo rows and columns they share (10 x 18 pixels staged per 8 x 16 tile: 1.41 x) meet in that XCD's L2 instead of being fetched
    // from the fabric once per tile.  Tile sets and statistics rows go by the virtual id: results do not depend on the switch.
    const int vblk = (C32_XCD_BAND && (G & 7) == 0) ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    int tile = vblk;
    if (tile < a.ntiles) issue(tile, va, okma);
    if (tile + G < a.ntiles) issue(tile + G, vb, okmb);
    if (C32_DEPTH > 2 && tile + 2 * G < a.ntiles) issue(tile + 2 * G, vc, okmc);
    for (; tile < a.ntiles; tile += C32_DEPTH * G) {
        step(tile, va, okma);
        if (tile + G < a.ntiles) step(tile + G, vb, okmb);
        if (C32_DEPTH > 2 && tile + 2 * G < a.ntiles) step(tile + 2 * G, vc, okmc);
    }

#ifdef C32_STAMPS
    if (tid == 0 && blockIdx.x < 1024) {
        for (int i = 0; i < 7; ++i) g_c32_stamps[blockIdx.x][i] = ph[i];
        g_c32_stamps[blockIdx.x][7] = __builtin_amdgcn_s_memtime() - born;
    }
#endif
    if (a.out_amax) spk_wave_amax_commit(out_mx, a.out_amax);
    if (a.flags & SPK_EPI_STATS) {
        s_sum += __shfl_xor(s_sum, 32, 64);              // lanes l and l + 32 hold the same channel
        s_sq += __shfl_xor(s_sq, 32, 64);
        if (h == 0) {
            float* dst = a.stats + ((size_t)(vblk * 4 + wave) * 32 + r) * 2;
            dst[0] = (float)s_sum;
            dst[1] = (float)s_sq;
        }
    }
}

extern "C" int spk_conv3x3_c32_stream(const float* in, const float* wpk, float* out, const float* in_scale, const float* in_shift,
                                      float* stats /* [4 nblocks][32][2] */, int B, int H, int W, int flags, const unsigned* in_amax,
                                      unsigned* out_amax, int nblocks, void* stream) {
    SPK_REQUIRE(in && wpk && out && in_amax, "spk_conv3x3_c32_stream: null pointer (in, wpk, out and the in_amax slot are required)");
    SPK_REQUIRE(B > 0 && H > 0 && W > 0 && (long long)B * H * W * 32 < 2147483647LL - 65536, "spk_conv3x3_c32_stream: %d x %d x %d pixels exceed 32-bit element offsets", B, H, W);
    SPK_REQUIRE((flags & ~(SPK_IN_AFFINE_RELU | SPK_EPI_STATS)) == 0, "spk_conv3x3_c32_stream: unsupported flags 0x%x", flags);
    SPK_REQUIRE(!(flags & SPK_IN_AFFINE_RELU) || (in_scale && in_shift), "spk_conv3x3_c32_stream: IN_AFFINE_RELU needs scale / shift");
    SPK_REQUIRE(!(flags & SPK_EPI_STATS) || stats, "spk_conv3x3_c32_stream: EPI_STATS needs a stats buffer");
    Conv3x3C32Args a;
    a.in = in; a.w_amax = (const unsigned*)wpk; a.wpk = wpk + 4; a.out = out; a.in_scale = in_scale; a.in_shift = in_shift;
    a.stats = stats; a.in_amax = in_amax; a.out_amax = out_amax; a.B = B; a.H = H; a.W = W; a.flags = flags;
    a.tiles_x = spk_ceil_div(W, C32_TW); a.tiles_y = spk_ceil_div(H, C32_TH);
    a.ntiles = B * a.tiles_y * a.tiles_x;
    SPK_REQUIRE(nblocks >= 1 && nblocks <= a.ntiles, "spk_conv3x3_c32_stream: nblocks=%d for %d tiles", nblocks, a.ntiles);
    const size_t lds_bytes = C32_WBYTES + 10 * C32_ROW;
    hipStream_t st = (hipStream_t)stream;
    if (flags & SPK_IN_AFFINE_RELU) hipLaunchKernelGGL(conv3x3_c32_stream_kernel<1>, dim3(nblocks), dim3(256), lds_bytes, st, a);
    else hipLaunchKernelGGL(conv3x3_c32_stream_kernel<0>, dim3(nblocks), dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv3x3_c32_stream");
    return 0;
}

#ifdef C32_STAMPS
extern "C" int spk_debug_stamps_c32(unsigned long long* out, int nblocks) {
    if (nblocks < 0) {
        static unsigned long long zeros[1024][8];
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_c32_stamps), zeros, sizeof(zeros));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c32_stamps), sizeof(unsigned long long) * 8 * (size_t)nblocks);
}
#endif
