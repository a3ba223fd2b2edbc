// Streaming 1x1 convolution (C -> C channels, stride 1, f16x3 operands): forward and data gradient of the 1x1 convolutions of the
// Bottleneck blocks (reference scripts/model.py:104-110,118-126: conv1 / conv3 of every block; 2 x 33 convolutions in ResNet-101).
//
// A 1x1 convolution is a plain GEMM [P pixels x Cin] x [Cin x Cout] with K = Cin <= 128: 32 flop per byte, HBM-bound by a factor of
// four on this chip.  The general kernel (conv_kernel.h) treats it as a convolution with one tap - tile, stage, barrier, 24 matrix
// instructions, barrier, twice per block, then the epilogue - and spends two thirds of a block's life in the staging phases
// (profiles/r04_conv_stamps.log); each 64-output-channel block re-stages and re-converts the pixels.  This kernel is a stream:
//   * persistent blocks walk tiles of TP consecutive pixels of the flattened [B H W] axis (a 1x1 convolution has no halo: a pixel
//     is a row of a matrix, no image geometry, no bounds but the tail);
//   * the whole weight matrix lives in REGISTERS as matrix-core B fragments for the life of the block (Cin/16 x 2 terms x 4
//     registers per wave: 64 at 128 channels) - loaded once, not once per tile;
//   * the loads of tile t + 1 (eight 16-byte loads per thread: whole 128-byte lines, Cin/4 consecutive lanes per pixel) are issued
//     before the matrix instructions of tile t and stay in flight through its K loop and epilogue; they are converted (fused
//     BatchNorm + ReLU, fp16 split) or copied (f16 pair input) into ONE LDS image between two barriers;
//   * a block covers ALL output channels: the four waves split the output channels (and, below 128 channels, the pixels), so a
//     pixel is staged and converted exactly once;
//   * the epilogue stores straight from the accumulator layout - a lane holds one output channel of 16 pixels, a wave-instruction
//     writes two whole 128-byte lines - with no LDS transpose; the BatchNorm statistics (or the BatchNorm-backward statistics of a
//     data gradient) are per-lane sums, kept in fp64 across the tiles of the block: one partial row per wave group and block
//     instead of one per wave and tile.
// Same arithmetic as the general kernel (two fp16 terms of value x sigma, three cross products, fp32 accumulation); another
// summation order of the statistics partials (they end in the same fp64 finalize).
#include "spk_common.h"

#ifndef C11_KD
#define C11_KD 0    // K-loop steps (2 reads + 3 matrix instructions each) whose A fragments are requested ahead of their use; 0 = each
                    // step's reads directly in front of its products (what the compiler schedules by itself).  3 measured +-0 inside the
                    // ResNet-101 step (128 channels 0.107 vs 0.106 ms, 64 channels 0.209 vs 0.212): the other waves of the SIMD fill the
                    // LDS latency here; the 3x3 kernel (conv3x3_c32_stream.hip, 18 steps per tile) gains 4 % from the same change
#endif

struct Conv1x1Args {
    const float* in;           // [P][Cin] fp32, or f16 pair tensor (SPK_IN_PRESPLIT)
    const float* wpk;          // f16x3 packed weights behind their 16-byte header
    const unsigned* w_amax;    // the header: float bits of max|w|
    float* out;                // [P][Cout]
    const float* in_scale;     // SPK_IN_AFFINE_RELU
    const float* in_shift;
    const float* epi_add;      // SPK_EPI_ADD: [P][Cout]
    const unsigned* add_mask;  //   add only where the bit is set ([P][Cout/32] words)
    const float* bn_raw;       // SPK_EPI_BNBWD: raw tensor of the BatchNorm whose backward statistics are reduced
    const unsigned* bn_mask;   //   its ReLU mask as sign bits, or NULL: recomputed from bn_raw * scale + shift > 0
    const float* bn4;          //   [4][Cout]: mean, invstd, scale, shift
    float* stats;              // SPK_EPI_STATS: [gridDim.x * WM][Cout][2]
    const unsigned* in_amax;
    unsigned* out_amax;
    long long P;
    int ntiles, flags;
};

// C = Cin = Cout: 32, 64 or 128.  Waves: WN along the output channels (32 each), WM = 4 / WN along the pixels; a wave owns two
// 32-pixel row tiles x one 32-channel column tile of the block's TP = 64 WM pixels.
// VAR: the launch's variant, known at compile time so that no per-row branch is left in the loops - the SPK_* flag bits plus
// V_ADDMASK / V_BNMASK (a sign mask accompanies the shortcut add / the BatchNorm-backward statistics); VAR < 0: the generic
// instantiation that reads everything from the argument block (any other combination).
#define V_ADDMASK (1 << 20)
#define V_BNMASK (1 << 21)
template <int C, int VAR>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(Conv1x1Args a) {
    constexpr int KG = C / 16;                       // K groups of 16 input channels
    constexpr int WN = C / 32 < 4 ? C / 32 : 4, WM = 4 / WN;
    constexpr int TP = 64 * WM;                      // pixels per tile
    constexpr int TPPX = C / 4;                      // threads per pixel (one float4 of channels each): whole pixels per pass
    constexpr int PPP = 256 / TPPX;                  // pixels per staging pass
    constexpr int NI = TP / PPP;                     // staging items per thread and tile (8)
    constexpr int PITCH = KG * 64 + 16;              // LDS bytes per pixel: [group][term][16 ch fp16] + 16 (odd multiple of 16 bytes)
    static_assert(NI == 8, "eight 16-byte loads per thread and tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds1[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int r = lane & 31, h = lane >> 5;
    const int flags = VAR >= 0 ? (VAR & ~(V_ADDMASK | V_BNMASK)) : a.flags;
    const bool has_addmask = VAR >= 0 ? (VAR & V_ADDMASK) != 0 : a.add_mask != nullptr;
    const bool has_bnmask = VAR >= 0 ? (VAR & V_BNMASK) != 0 : a.bn_mask != nullptr;
    const float sig = spk_sigma_from_amax_bits(*a.in_amax);
    const float inv_sig = 1.f / sig, inv_wsig = 1.f / spk_sigma_from_amax_bits(*a.w_amax);      // (two factors: their product may leave the fp32 range)
    const bool pairs = (flags & SPK_IN_PRESPLIT) != 0, aff = (flags & SPK_IN_AFFINE_RELU) != 0;

    // the weights of this wave's 32 output channels: B fragments [group][term], 16 bytes per lane each
    // (packed order [tap][Cin/16][term][Cout/32][64 lanes][8 fp16], pack.hip)
    f32x4 bw[KG][2];
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int s = 0; s < 2; ++s) bw[g][s] = *(const f32x4*)(a.wpk + ((size_t)(g * 2 + s) * (C / 32) + wn) * 256 + lane * 4);

    // staging geometry of this thread: float4 q of the channels of pixel (tid / TPPX) + PPP * u of a tile
    const int q = tid % TPPX, prow = tid / TPPX;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (aff) {
        sc = *(const f32x4*)(a.in_scale + q * 4);
        sh = *(const f32x4*)(a.in_shift + q * 4);
    }
    unsigned char* wr = lds1 + prow * PITCH + (q >> 2) * 64 + (q & 3) * 8;          // term 0; term 1 at + 32

    f32x4 v[NI];
    const int Pn = (int)a.P;                                    // (the launcher holds P * C below 2^31: 32-bit element offsets)
    auto issue = [&](int tile) {
        const int p0 = tile * TP;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            int p = p0 + prow + PPP * u;
            p = p < Pn ? p : Pn - 1;                            // (tail: a valid address, zeroed below)
            v[u] = *(const f32x4*)(a.in + (unsigned)p * (unsigned)C + (unsigned)(q * 4));
        }
    };
    // epilogue vectors of this lane's output channel
    const int ch = wn * 32 + r;
    float bmu = 0.f, bis = 0.f, bsc = 0.f, bsh = 0.f;
    if (flags & SPK_EPI_BNBWD) {
        bmu = a.bn4[ch];
        bis = a.bn4[C + ch];
        bsc = a.bn4[2 * C + ch];
        bsh = a.bn4[3 * C + ch];
    }
    double s_sum = 0.0, s_sq = 0.0;
    float out_mx = 0.f;

    int tile = blockIdx.x;
    if (tile < a.ntiles) issue(tile);
    for (; tile < a.ntiles; tile += gridDim.x) {
        const int p0 = tile * TP;
        const bool ragged = p0 + TP > Pn;                       // (only the last tile: wave-uniform)
        __syncthreads();                                        // every wave has read the previous tile's fragments
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            uint2 t0, t1;
            const bool live = !ragged || p0 + prow + PPP * u < Pn;
            if (pairs) {
                spk_pair_unpack(v[u], t0, t1);
                if (!live) t0 = t1 = (uint2){0u, 0u};
            } else {
                f32x4 w = v[u];
                if (aff) {
                    w = w * sc + sh;
                    w[0] = fmaxf(w[0], 0.f);
                    w[1] = fmaxf(w[1], 0.f);
                    w[2] = fmaxf(w[2], 0.f);
                    w[3] = fmaxf(w[3], 0.f);
                }
                if (!live) w = (f32x4){0.f, 0.f, 0.f, 0.f};
                split2h(w, sig, t0, t1);
            }
            *(uint2*)(wr + u * (PPP * PITCH)) = t0;
            *(uint2*)(wr + u * (PPP * PITCH) + 32) = t1;
        }
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) issue(tile + gridDim.x);      // in flight through the K loop and the epilogue

        // K loop: 2 row tiles x KG groups x 3 products; A fragment of (row tile i, group g, term s): 16 bytes at
        // pixel * PITCH + g * 64 + s * 32 + h * 16 (lane: pixel r of the row tile, channels 8 h .. 8 h + 7 of the group)
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        const unsigned char* rd = lds1 + (wm * 64 + r) * PITCH + h * 16;
        // software-pipelined by hand: the A fragments of step s + C11_KD are requested before the matrix instructions of step s
        // (step = (group g, row tile i), g-major: the accumulation order of a row tile is unchanged); left to the compiler every
        // step's two reads sit directly in front of its products - 2 KG exposed LDS latencies per tile
        constexpr int NS = 2 * KG;
        f32x4 fr[C11_KD + 1][2];
        auto frag = [&](int s2, f32x4* f) {
            const int g = s2 >> 1, i = s2 & 1;
            f[0] = *(const f32x4*)(rd + i * (32 * PITCH) + g * 64);
            f[1] = *(const f32x4*)(rd + i * (32 * PITCH) + g * 64 + 32);
        };
#pragma unroll
        for (int s2 = 0; s2 < C11_KD; ++s2) frag(s2, fr[s2]);
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (s2 + C11_KD < NS) frag(s2 + C11_KD, fr[(s2 + C11_KD) % (C11_KD + 1)]);
            __builtin_amdgcn_sched_barrier(0);           // (the scheduler would sink these reads back in front of their use)
            const f32x4* f = fr[s2 % (C11_KD + 1)];
            const int g = s2 >> 1, i = s2 & 1;
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[0]), __builtin_bit_cast(f16x8, bw[g][1]), acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[1]), __builtin_bit_cast(f16x8, bw[g][0]), acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[0]), __builtin_bit_cast(f16x8, bw[g][0]), acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // epilogue from the accumulator layout: register e of row tile i = pixel (e & 3) + 8 (e >> 2) + 4 h of the tile, channel r
        const int pw = p0 + wm * 64;
        float ts = 0.f, tq = 0.f;                                // this tile's statistics (fp32 over 32 values, then fp64)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            // eight rows (half a row tile) at a time: the epilogue's reads (shortcut gradient, raw tensor and mask words) first - one
            // batch per stream inside ONE uniform branch each, so that the loads are issued back to back - then their use
            // (sixteen rows at a time spill at 128 channels, where the weights hold 64 registers)
            const int i = ib >> 1, e0 = (ib & 1) * 8;
            float adv[8], rwv[8];
            unsigned amw[8], bmw[8];
            unsigned eo[8];                                      // element offset of (row, this lane's channel), clamped at the tail
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + k;
                int p = pw + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                p = p < Pn ? p : Pn - 1;
                eo[k] = (unsigned)p * (unsigned)C + (unsigned)ch;
            }
            if (flags & SPK_EPI_ADD) {
#pragma unroll
                for (int k = 0; k < 8; ++k) adv[k] = __builtin_nontemporal_load(a.epi_add + eo[k]);
                if (has_addmask) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) amw[k] = a.add_mask[eo[k] >> 5];           // word of (pixel, channels 32 wn ..): (p C + ch) / 32
                }
            }
            if (flags & SPK_EPI_BNBWD) {
#pragma unroll
                for (int k = 0; k < 8; ++k) rwv[k] = __builtin_nontemporal_load(a.bn_raw + eo[k]);
                if (has_bnmask) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) bmw[k] = a.bn_mask[eo[k] >> 5];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + k;
                const int p = pw + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (ragged && p >= Pn) continue;
                float val = acc[i][e] * inv_sig * inv_wsig;
                if (flags & SPK_EPI_ADD) {
                    float ad = adv[k];
                    if (has_addmask) ad = ((amw[k] >> r) & 1u) ? ad : 0.f;
                    val += ad;
                }
                __builtin_nontemporal_store(val, a.out + eo[k]);
                out_mx = fmaxf(out_mx, fabsf(val));
                if (flags & SPK_EPI_BNBWD) {
                    const float rw = rwv[k];
                    const bool on = has_bnmask ? ((bmw[k] >> r) & 1u) != 0 : (rw * bsc + bsh > 0.f);
                    const float dz = on ? val : 0.f;
                    ts += dz;
                    tq += dz * ((rw - bmu) * bis);
                } else {
                    ts += val;
                    tq += val * val;
                }
            }
        }
        s_sum += (double)ts;
        s_sq += (double)tq;
    }

    if (a.out_amax) spk_wave_amax_commit(out_mx, a.out_amax);
    if (flags & SPK_EPI_STATS) {
        // lanes l and l + 32 hold the same channel (rows 4 h + ...): fold them, one partial row per (block, pixel wave group)
        s_sum += __shfl_xor(s_sum, 32, 64);
        s_sq += __shfl_xor(s_sq, 32, 64);
        if (h == 0) {
            float* dst = a.stats + ((size_t)(blockIdx.x * WM + wm) * C + ch) * 2;
            dst[0] = (float)s_sum;
            dst[1] = (float)s_sq;
        }
    }
}

extern "C" int spk_conv1x1_stream_rows(int nblocks, int C) { return nblocks * (C >= 128 ? 1 : (C == 64 ? 2 : 4)); }

extern "C" int spk_conv1x1_stream(const float* in, const float* wpk, float* out, const float* in_scale, const float* in_shift,
                                  const float* epi_add, const unsigned* add_mask, const float* bn_raw, const unsigned* bn_mask,
                                  const float* bn4, float* stats, long long P, int C, int flags, const unsigned* in_amax,
                                  unsigned* out_amax, int nblocks, void* stream) {
    SPK_REQUIRE(in && wpk && out && in_amax, "spk_conv1x1_stream: null pointer (in, wpk, out and the in_amax slot are required)");
    SPK_REQUIRE(C == 32 || C == 64 || C == 128, "spk_conv1x1_stream: C=%d (32, 64 or 128 channels in and out)", C);
    SPK_REQUIRE(P > 0 && P * C < 2147483647LL, "spk_conv1x1_stream: P=%lld pixels x %d channels exceed 32-bit element offsets", P, C);
    const int known = SPK_IN_AFFINE_RELU | SPK_IN_PRESPLIT | SPK_EPI_STATS | SPK_EPI_ADD | SPK_EPI_BNBWD;
    SPK_REQUIRE((flags & ~known) == 0, "spk_conv1x1_stream: unsupported flags 0x%x", flags & ~known);
    SPK_REQUIRE(!(flags & SPK_IN_AFFINE_RELU) || (in_scale && in_shift), "spk_conv1x1_stream: IN_AFFINE_RELU needs scale / shift");
    SPK_REQUIRE(!((flags & SPK_IN_AFFINE_RELU) && (flags & SPK_IN_PRESPLIT)), "spk_conv1x1_stream: a pair input has no input transform");
    SPK_REQUIRE(!(flags & SPK_EPI_ADD) || epi_add, "spk_conv1x1_stream: EPI_ADD needs epi_add");
    SPK_REQUIRE(!add_mask || (flags & SPK_EPI_ADD), "spk_conv1x1_stream: add_mask needs EPI_ADD");
    SPK_REQUIRE(!(flags & SPK_EPI_STATS) || stats, "spk_conv1x1_stream: EPI_STATS needs a stats buffer");
    SPK_REQUIRE(!(flags & SPK_EPI_BNBWD) || ((flags & SPK_EPI_STATS) && bn_raw && bn4), "spk_conv1x1_stream: EPI_BNBWD needs EPI_STATS, bn_raw and bn4");
    Conv1x1Args a;
    a.in = in; a.w_amax = (const unsigned*)wpk; a.wpk = wpk + 4; a.out = out; a.in_scale = in_scale; a.in_shift = in_shift;
    a.epi_add = epi_add; a.add_mask = add_mask; a.bn_raw = bn_raw; a.bn_mask = bn_mask; a.bn4 = bn4; a.stats = stats;
    a.in_amax = in_amax; a.out_amax = out_amax; a.P = P; a.flags = flags;
    const int WM = C >= 128 ? 1 : (C == 64 ? 2 : 4), TP = 64 * WM;
    a.ntiles = (int)((P + TP - 1) / TP);
    SPK_REQUIRE(nblocks >= 1 && nblocks <= a.ntiles, "spk_conv1x1_stream: nblocks=%d for %d tiles", nblocks, a.ntiles);
    const size_t lds_bytes = (size_t)TP * ((C / 16) * 64 + 16);
    hipStream_t st = (hipStream_t)stream;
    const int var = flags | (add_mask ? V_ADDMASK : 0) | (bn_mask ? V_BNMASK : 0);
    // the combinations the training step uses (engine.py): forward with or without the fused input BatchNorm + ReLU, raw output +
    // statistics; data gradient of the last convolution of a block (pair input, BatchNorm-backward statistics with the mask
    // recomputed from the raw tensor); data gradient of the first one (pair input, masked shortcut add, with or without the
    // statistics of the previous block's last BatchNorm by sign bits); anything else: the generic instantiation
#define LAUNCH(CC, VV) hipLaunchKernelGGL((conv1x1_stream_kernel<CC, VV>), dim3(nblocks), dim3(256), lds_bytes, st, a)
#define PICK(CC)                                                                                                         \
    switch (var) {                                                                                                       \
        case SPK_IN_AFFINE_RELU | SPK_EPI_STATS: LAUNCH(CC, SPK_IN_AFFINE_RELU | SPK_EPI_STATS); break;                  \
        case SPK_EPI_STATS: LAUNCH(CC, SPK_EPI_STATS); break;                                                            \
        case SPK_IN_PRESPLIT | SPK_EPI_STATS | SPK_EPI_BNBWD: LAUNCH(CC, SPK_IN_PRESPLIT | SPK_EPI_STATS | SPK_EPI_BNBWD); break; \
        case SPK_IN_PRESPLIT | SPK_EPI_ADD | V_ADDMASK: LAUNCH(CC, SPK_IN_PRESPLIT | SPK_EPI_ADD | V_ADDMASK); break;    \
        case SPK_IN_PRESPLIT | SPK_EPI_ADD | V_ADDMASK | SPK_EPI_STATS | SPK_EPI_BNBWD | V_BNMASK:                       \
            LAUNCH(CC, SPK_IN_PRESPLIT | SPK_EPI_ADD | V_ADDMASK | SPK_EPI_STATS | SPK_EPI_BNBWD | V_BNMASK); break;     \
        default: LAUNCH(CC, -1);                                                                                         \
    }
    if (C == 128) { PICK(128) }
    else if (C == 64) { PICK(64) }
    else { PICK(32) }
#undef PICK
#undef LAUNCH
    SPK_LAUNCH_CHECK("spk_conv1x1_stream");
    return 0;
}
