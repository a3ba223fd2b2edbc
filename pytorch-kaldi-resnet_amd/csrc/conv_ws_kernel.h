#pragma once
// Producer / consumer ("wave-specialised") form of the bf16-split implicit-GEMM convolution for 3x3 kernels.
//
// Why: in conv_mfma_kernel every wave alternates between staging a channel chunk (global -> registers -> bf16 split ->
// LDS) and the MFMA tap loop over it, with block barriers in between; the matrix pipe of a SIMD idles whenever both of
// its resident waves are staging, waiting at a barrier or in an epilogue (measured: SQ_VALU_MFMA_BUSY 41-60 % of SIMD
// cycles, SQ_WAIT_ANY 35-54 % of wave cycles).  Here a 512-thread workgroup (one per CU) is split by role:
//   waves 4-7 (producers): stage chunk k+1 into the other slot of a two-slot LDS ring - all global loads, the fused input
//                          transforms (BN+ReLU, or the whole BatchNorm backward with its side outputs) and the bf16 split
//                          run here, on the VALU / memory pipes, with deep load batches (these waves own no accumulators);
//   waves 0-3 (consumers): one per SIMD, run the MFMA tap loop of chunk k back to back - their instruction stream is
//                          MFMAs, LDS fragment reads and the L2 -> VGPR weight-fragment prefetch, nothing else.
// One s_barrier per chunk hands slot k+1 over and slot k back (B_k below); both roles execute exactly one barrier per
// (tile, chunk) work item of the block, so the counts always match and every wave leaves the loop after the same number.
//   producer: stage W_0 | B_0 | stage W_1 | B_1 | stage W_2 | ...
//   consumer:           | B_0 | mfma  W_0 | B_1 | mfma  W_1 | ...         (W_k uses slot k & 1)
// The four consumer waves form a WP x WC grid (WP * WC = 4): wave (wp, wc) owns MT m-tiles of 32 pixels of pixel group wp
// and NT n-tiles of 32 output channels of channel group wc.  WC > 1 makes the block tile wide in channels instead of
// pixels: the waves of one pixel group read the SAME A fragments from LDS (LDS has the bandwidth) and each loads only
// its own weight fragments, so a weight fragment is fetched once per block instead of once per wave (the weight stream
// through L1 was the largest single cost of the WP = 4 form: -23 % with it removed) and a staged halo tile serves up
// to 128 output channels (less staging per MFMA, no re-staging per channel group).
// Blocks are persistent (grid = #CUs): a block walks tiles v = blockIdx.x, +gridDim.x, ... so the producers stage the
// next tile's first chunk while the consumers run the current tile's epilogue; the per-block prologue is paid once.
// Same ConvArgs, tap table, packed weights, LDS pixel layout, epilogue and results as conv_mfma_kernel<.., SPLIT>.
#include "conv_kernel.h"

#ifndef WS_STAGE_U
#define WS_STAGE_U 8     // plain staging: 16-byte loads in flight per producer thread
#endif
#ifndef WS_STAGE_U2
#define WS_STAGE_U2 4    // fused BatchNorm-backward staging: pixels in flight per producer thread (three loads each)
#endif

static __device__ __forceinline__ void ws_barrier() {
    // LDS traffic of this wave is complete (reads returned / writes performed), then rendezvous with the other role.
    // No vmcnt wait: the consumers' weight-fragment prefetch stays in flight across the hand-over.
#ifdef WS_ABL_NO_SYNC
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // diagnostic: roles never wait for each other (wrong results)
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

template <int MT, int NT, int WC, bool BNBWD, int SPLIT>
__global__ __launch_bounds__(512) void conv_ws_kernel(ConvArgs a) {
    static_assert(WC == 1 || WC == 2 || WC == 4, "consumer waves: WP x WC = 4");
    constexpr int WP = 4 / WC;
    static_assert(SPLIT == 3 || SPLIT == 6 || SPLIT == 9, "wave-specialised kernel: split operands only");
    using Cfg = ConvCfg<SPLIT>;
    constexpr int CK = Cfg::CK, TPP = Cfg::TPP, PPP = Cfg::PPP, LP4 = Cfg::LP4, NTERM = Cfg::NTERM;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int flags = a.flags;
    const int nchunks = a.Cin / CK;
    const int halo_pix = a.halo_h * a.halo_w;
    const int slot_floats = halo_pix * LP4 * 4;
    const int cout32 = a.Cout >> 5;
    const int ntiles = a.nblocks;                 // logical tiles (pixel tile x cout group), walked persistently
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    // XCD-aware bijective remap of the virtual block id (see conv_mfma_kernel): neighbours in tile order share an L2
    auto tile_of = [&](int v) {
        const int xcd = v & 7, slot = v >> 3;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    };
    int k = 0;                                    // work-item counter of this block: ring slot = k & 1
    // f16x3: input scale (a power of two) and the factor that takes the accumulators back to fp32 units
    float sig = 1.f, inv_sig = 1.f, inv_wsig = 1.f;
    if constexpr (SPLIT == 3) {
        sig = a.in_amax ? spk_sigma_from_amax_bits(*a.in_amax) : a.in_sigma;
        inv_sig = 1.f / sig;
        inv_wsig = 1.f / spk_sigma_from_amax_bits(*a.w_amax);
    }
    // per-tap table in LDS, behind the ring and the epilogue slabs: [0..11] A-fragment offset of the tap inside a slot
    // (16-byte units), [16..27] float offset of the tap's weight fragments; entries 9..11 repeat tap 8 (harmless prefetches)
    int* tapt = (int*)(lds + 2 * slot_floats + 4 * 32 * (NT * 32 + 4));
    if (tid < 12) {
        const int t = tid < 8 ? tid : 8;
        tapt[tid] = a.tap_off[t];
        tapt[16 + tid] = a.tap_boff[t];
    }
    __syncthreads();

    if (wave >= 4) {
        // ================================================= producers ==================================================
        const int ptid = tid - 256;
        const int quad = ptid & (TPP - 1);        // this thread's float4 of channels within a staged pixel
        const int prow = ptid / TPP;              // and its pixel slot within a staging pass
        float side_mx = 0.f;
        auto store_px = [&](float* plane, int p, f32x4 w) {
            uint2* dst = (uint2*)plane + p * (LP4 * 2) + quad;   // [term][CK ch]: CK*2 bytes per term
            if constexpr (SPLIT == 3) {
                uint2 t0, t1;
                split2h(w, sig, t0, t1);
                dst[0] = t0;
                dst[CK / 4] = t1;
            } else {
                uint2 t0, t1, t2;
                split3(w, t0, t1, t2);
                dst[0] = t0;
                dst[CK / 4] = t1;
                dst[CK / 2] = t2;
            }
        };
        for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
            const int bid = tile_of(v);
            const int cg = bid % a.ncg;
            int pt = bid / a.ncg;
            const int tx = pt % a.tiles_x;
            pt /= a.tiles_x;
            const int ty = pt % a.tiles_y;
            const int b = pt / a.tiles_y;
            const int oy0 = ty * a.TH, ox0 = tx * a.TW;
            const int iy0 = oy0 * a.IS + a.min_dy, ix0 = ox0 * a.IS + a.min_dx;
            // scalar image bases + 32-bit offsets inside image b (see conv_kernel.h)
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
            for (int ch = 0; ch < nchunks; ++ch, ++k) {
                float* ldsp = lds + (k & 1) * slot_floats;
                const int c = ch * CK + quad * 4;
                if constexpr (BNBWD) {
                    // staged value = BatchNorm-backward of `in` (see conv_mfma_kernel): dz = in*mask,
                    // value = k1*(dz - m1 - xhat*m2); the cout-group-0 block writes the tile's own pixels back as side outputs
                    const f32x4 mu = *(const f32x4*)(a.in_bn4 + c), is = *(const f32x4*)(a.in_bn4 + a.Cin + c);
                    const f32x4 bsc = *(const f32x4*)(a.in_bn4 + 2 * a.Cin + c), bsh = *(const f32x4*)(a.in_bn4 + 3 * a.Cin + c);
                    const f32x4 k1 = *(const f32x4*)(a.in_coef + c), m1 = *(const f32x4*)(a.in_coef + a.Cin + c);
                    const f32x4 m2 = *(const f32x4*)(a.in_coef + 2 * a.Cin + c);
                    const bool owner = (cg == 0);
                    constexpr int U2 = WS_STAGE_U2;
                    for (int base = prow; base < halo_pix; base += PPP * U2) {
                        f32x4 vv[U2], rw[U2], ac[U2];
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
                            vv[u] = *(const f32x4*)(img_in + off[u]);
                            rw[u] = *(const f32x4*)(img_raw + off[u]);
                            if (a.in_mask) mw[u] = img_mask[pi * (unsigned)(a.Cin >> 5) + (unsigned)(c >> 5)];
                            else if (a.in_act) ac[u] = *(const f32x4*)(img_act + off[u]);
                        }
#pragma unroll
                        for (int u = 0; u < U2; ++u) {
                            const int p = base + PPP * u;
                            f32x4 dz;
                            if (a.in_mask) {
                                const unsigned bits = mw[u] >> (c & 31);
#pragma unroll
                                for (int e = 0; e < 4; ++e) dz[e] = ((bits >> e) & 1u) ? vv[u][e] : 0.f;
                            } else {
                                const f32x4 m = a.in_act ? ac[u] : rw[u] * bsc + bsh;
#pragma unroll
                                for (int e = 0; e < 4; ++e) dz[e] = m[e] > 0.f ? vv[u][e] : 0.f;
                            }
                            f32x4 w = k1 * (dz - m1 - ((rw[u] - mu) * is) * m2);
                            if (!inb[u]) w = (f32x4){0.f, 0.f, 0.f, 0.f};
                            if (p < halo_pix) {
                                store_px(ldsp, p, w);
                                if (owner && core[u]) {
                                    *(f32x4*)(img_draw + off[u]) = w;
                                    if (a.side_dz) *(f32x4*)(img_dz + off[u]) = dz;
                                    side_mx = fmaxf(fmaxf(side_mx, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
                                }
                            }
                        }
                    }
                } else {
                    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                    if (flags & SPK_IN_AFFINE_RELU) {
                        sc = *(const f32x4*)(a.in_scale + c);
                        sh = *(const f32x4*)(a.in_shift + c);
                    }
                    constexpr int U = WS_STAGE_U;
#ifdef WS_ABL_NO_STAGE
                    for (int base = prow; base < 0; base += PPP * U) {
#else
                    for (int base = prow; base < halo_pix; base += PPP * U) {
#endif
                        f32x4 vv[U];
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
                            vv[u] = *(const f32x4*)(img_in_p + pi * (unsigned)a.Cin + (unsigned)c);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int p = base + PPP * u;
                            f32x4 w = vv[u];
                            if (flags & SPK_IN_AFFINE_RELU) {
                                w = w * sc + sh;
                                w[0] = fmaxf(w[0], 0.f);
                                w[1] = fmaxf(w[1], 0.f);
                                w[2] = fmaxf(w[2], 0.f);
                                w[3] = fmaxf(w[3], 0.f);
                            }
                            if (!inb[u]) w = (f32x4){0.f, 0.f, 0.f, 0.f};
                            if (p < halo_pix) store_px(ldsp, p, w);
                        }
                    }
                }
                ws_barrier();      // B_k: slot k & 1 is full; the consumers are done with the other slot
            }
        }
        if constexpr (BNBWD) {
            if (a.side_amax) spk_wave_amax_commit(side_mx, a.side_amax);
        }
        return;
    }

    // ===================================================== consumers =====================================================
    __builtin_amdgcn_s_setprio(2);         // the matrix stream wins VALU-issue arbitration against the staging wave of its SIMD
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave / WC, wc = wave % WC;
    constexpr int LW = NT * 32 + 4;        // epilogue slab row pitch in floats (16-byte aligned, bank-staggered)
    constexpr int Q = NT * 8;              // float4 quads per pixel row
    constexpr int RPP = 64 / Q;            // pixel rows covered by one 64-lane pass
    float* slab = lds + 2 * slot_floats + wave * (32 * LW);   // private to this wave, outside the ring
    const int qc = lane % Q, qr = lane / Q;
    const int npix_tile = a.TH * a.TW;
    const int term_stride = cout32 * 256;
    const unsigned lane4 = (unsigned)lane * 4u;
    const int o0 = a.tap_off[0], o1 = a.tap_off[1], bo0 = a.tap_boff[0], bo1 = a.tap_boff[1], bo2 = a.tap_boff[2];

    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const int bid = tile_of(v);
        const int cg = bid % a.ncg;
        const int ptile = bid / a.ncg;
        int pt = ptile;
        const int tx = pt % a.tiles_x;
        pt /= a.tiles_x;
        const int ty = pt % a.tiles_y;
        const int b = pt / a.tiles_y;
        const int oy0 = ty * a.TH, ox0 = tx * a.TW;
        const int n0 = (cg * WC + wc) * NT * 32;
        int lbase[MT], obase[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int q = (wp * MT + i) * 32 + r;
            bool ok = q < npix_tile;
            const int qq = ok ? q : 0;
            const int ly = qq / a.TW, lx = qq - ly * a.TW;
            const int oy = oy0 + ly, ox = ox0 + lx;
            ok = ok && oy < a.OH && ox < a.OW;
            lbase[i] = ((ly * a.IS) * a.halo_w + lx * a.IS) * LP4 + h;   // 16-byte units
            obase[i] = ok ? ((b * a.OHf + oy * a.OS + a.ooy) * a.OWf + ox * a.OS + a.oox) * a.Cout + n0 : -1;
        }
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // K loop.  One chunk = 16 input channels x 9 taps; a step = one tap = MT "positions" (one m-tile each).
        // The per-tap table entries are read from a small LDS copy one step ahead and moved to scalar registers: a scalar
        // (kernel-argument) load inside this loop would need s_waitcnt lgkmcnt(0) - scalar loads return out of order - and
        // that also drains the LDS fragment reads in flight; LDS reads return in order, so their waits are counted.
        // Operand pipeline: the weight fragments of tap t + 2 are requested at the first position of tap t (three register
        // buffers in rotation, ~2 x 6*MT*NT MFMAs of cover for the L2 round trip), the A fragments of position n + 2 at
        // position n (a ring of three m-tile fragment sets: LDS latency ~130 cycles against 2 x 6*NT MFMAs of 32 cycles).
        for (int ch = 0; ch < nchunks; ++ch, ++k) {
            const f32x4* lds4 = (const f32x4*)(lds + (k & 1) * slot_floats);
            // packed weights: [tap][Cin/16][term][Cout/32][64 lanes][8 bf16]; tap_boff[t] = float offset of (tap, plane 0)
            const float* wchunk = a.wpk + ((size_t)ch * NTERM * cout32 + (cg * WC + wc) * NT) * 256;
            f32x4 bq[3][NTERM][NT], aq[3][NTERM];
            auto load_b = [&](int buf, int boff) {
#ifdef WS_ABL_B_SAMEADDR
                const float* wp = a.wpk + (boff & 1);      // diagnostic: an L1-hot address every step (wrong results)
#else
                const float* wp = wchunk + boff;
#endif
#pragma unroll
                for (int s = 0; s < NTERM; ++s)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
#ifdef WS_ABL_NO_BLOAD
                        asm volatile("" : "+v"(bq[buf][s][j]) : "s"(wp));
#else
                        bq[buf][s][j] = *(const f32x4*)(wp + (unsigned)(s * term_stride + j * 256) + lane4);
#endif
                    }
            };
            auto load_a = [&](int set, int toff, int i) {
#pragma unroll
                for (int s = 0; s < NTERM; ++s) {
#ifdef WS_ABL_NO_ALOAD
                    asm volatile("" : "+v"(aq[set][s]) : "s"(toff));
#else
                    aq[set][s] = lds4[lbase[i] + toff + s * (CK / 8)];
#endif
                }
            };
#if defined(WS_ABL_NO_BLOAD) || defined(WS_ABL_NO_ALOAD)
            for (int u = 0; u < 3; ++u)
                for (int s = 0; s < NTERM; ++s) {
                    for (int j = 0; j < NT; ++j) bq[u][s][j] = (f32x4){1.f, 2.f, 3.f, 4.f};
                    aq[u][s] = (f32x4){1.f, 2.f, 3.f, 4.f};
                }
#endif
            load_b(0, bo0);                                                // weights do not depend on the hand-over
            load_b(1, bo1);
            ws_barrier();                                                  // B_k: slot k & 1 is full
            load_a(0, o0, 0);
            load_a(1, MT > 1 ? o0 : o1, 1 % MT);
            int o_cur = o0, o_next = o1, b_n2 = bo2;
            for (int t0 = 0; t0 < 9; t0 += 3) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int vo = tapt[t0 + u + 2], vb = tapt[16 + t0 + u + 3];     // entries past tap 8 repeat tap 8
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (i == 0) load_b((u + 2) % 3, b_n2);
                        if (i + 2 < MT) load_a((MT * u + i + 2) % 3, o_cur, i + 2);
                        else load_a((MT * u + i + 2) % 3, o_next, i + 2 - MT);
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
                                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                            __builtin_bit_cast(f16x8, aq[(MT * u + i) % 3][sa]), __builtin_bit_cast(f16x8, bq[u % 3][sb][j]),
                                            acc[i][j], 0, 0, 0);
                                    else
                                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                            __builtin_bit_cast(bf16x8, aq[(MT * u + i) % 3][sa]), __builtin_bit_cast(bf16x8, bq[u % 3][sb][j]),
                                            acc[i][j], 0, 0, 0);
                                }
                            }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    o_cur = o_next;
                    o_next = __builtin_amdgcn_readfirstlane(vo);
                    b_n2 = __builtin_amdgcn_readfirstlane(vb);
                }
            }
        }

        // ---- epilogue (identical to conv_mfma_kernel's): transpose each 32x32 tile through the wave's private LDS slab
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
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    slab[row * LW + j * 32 + r] = SPLIT == 3 ? acc[i][j][e] * inv_sig * inv_wsig : acc[i][j][e];
                }
            // All global loads of the m-tile's epilogue (shortcut gradient, raw conv output of the BatchNorm whose backward
            // statistics are reduced here, sign-mask words) are issued first, unconditionally (rows outside the tensor read
            // element 0 and are discarded): with one workgroup per CU nothing else would cover their latency.
            constexpr int P = 32 / RPP;
            int obv[P];
            f32x4 adv[P], rwv[P], acv[P];
            unsigned amv[P], bmv[P];
#pragma unroll
            for (int kk = 0; kk < P; ++kk) {
                obv[kk] = __shfl(obase[i], kk * RPP + qr, 64);
                const int oc = obv[kk] >= 0 ? obv[kk] : n0;
                const int ch0 = n0 + qc * 4;
                const size_t mword = (size_t)((oc - n0) / a.Cout) * (a.Cout >> 5) + (ch0 >> 5);
                if (flags & SPK_EPI_ADD) {
                    adv[kk] = *(const f32x4*)(a.epi_add + oc + qc * 4);
                    if (a.add_mask) amv[kk] = a.add_mask[mword];
                }
                if (flags & SPK_EPI_BNBWD) {
                    rwv[kk] = *(const f32x4*)(a.bn_raw + oc + qc * 4);
                    if (a.bn_mask) bmv[kk] = a.bn_mask[mword];
                    else if (a.bn_act) acv[kk] = *(const f32x4*)(a.bn_act + oc + qc * 4);
                }
            }
#pragma unroll
            for (int kk = 0; kk < P; ++kk) {
                const int row = kk * RPP + qr;
                const int ob = obv[kk];
                f32x4 vv = *(const f32x4*)(slab + row * LW + qc * 4);
#ifdef WS_ABL_NO_EPI
                asm volatile("" ::"v"(vv));
                if (ob == -12345) {
#else
                if (ob >= 0) {
#endif
                    float* dst = a.out + ob + qc * 4;
                    const int sh5 = (n0 + qc * 4) & 31;
                    if (flags & SPK_EPI_AFFINE) vv = vv * es + eh;
                    if (flags & SPK_EPI_ADD) {
                        f32x4 ad = adv[kk];
                        if (a.add_mask) {
                            const unsigned bits = amv[kk] >> sh5;
#pragma unroll
                            for (int c = 0; c < 4; ++c) ad[c] = ((bits >> c) & 1u) ? ad[c] : 0.f;
                        }
                        vv += ad;
                    }
                    if (flags & SPK_EPI_RELU) {
                        vv[0] = fmaxf(vv[0], 0.f);
                        vv[1] = fmaxf(vv[1], 0.f);
                        vv[2] = fmaxf(vv[2], 0.f);
                        vv[3] = fmaxf(vv[3], 0.f);
                    }
                    *(f32x4*)dst = vv;
                    out_mx = fmaxf(fmaxf(out_mx, fmaxf(fabsf(vv[0]), fabsf(vv[1]))), fmaxf(fabsf(vv[2]), fabsf(vv[3])));
                    if (flags & SPK_EPI_BNBWD) {
                        const f32x4 rw = rwv[kk];
                        f32x4 dz;
                        if (a.bn_mask) {
                            const unsigned bits = bmv[kk] >> sh5;
#pragma unroll
                            for (int c = 0; c < 4; ++c) dz[c] = ((bits >> c) & 1u) ? vv[c] : 0.f;
                        } else {
                            const f32x4 m = a.bn_act ? acv[kk] : rw * bsc + bsh;
#pragma unroll
                            for (int c = 0; c < 4; ++c) dz[c] = m[c] > 0.f ? vv[c] : 0.f;
                        }
                        ssum += dz;
                        ssq += dz * ((rw - bmu) * bis);
                    } else {
                        ssum += vv;
                        ssq += vv * vv;
                    }
                }
            }
        }
        if (a.out_amax) spk_wave_amax_commit(out_mx, a.out_amax);
        if (flags & SPK_EPI_STATS) {
#pragma unroll
            for (int off = Q; off < 64; off <<= 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ssum[c] += __shfl_xor(ssum[c], off, 64);
                    ssq[c] += __shfl_xor(ssq[c], off, 64);
                }
            }
            if (lane < Q) {
                float* dst = a.stats + ((size_t)(ptile * WP + wp) * a.Cout + n0 + lane * 4) * 2;     // one partial row per pixel group
                *(f32x4*)dst = (f32x4){ssum[0], ssq[0], ssum[1], ssq[1]};
                *(f32x4*)(dst + 4) = (f32x4){ssum[2], ssq[2], ssum[3], ssq[3]};
            }
        }
    }
}
