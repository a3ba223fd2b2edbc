// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// One kernel serves: 3x3 / 1x1 forward at stride 1 / 2 (scripts/model.py:12-15,105-110,233-234
// of the reference, i.e. nn.Conv2d(bias=False)), and the data gradient of each of them
// (a stride-2 data gradient runs as four parity classes, each a dense conv with a tap subset).
// The caller describes the conv as a tap table: for logical output pixel (oy,ox) and tap t the
// input pixel is (oy*IS + dy[t], ox*IS + dx[t]); the logical output pixel is stored at
// (oy*OS + ooy, ox*OS + oox) of the physical NHWC output.
//
// Data layout: activations NHWC fp32.  Weights are pre-packed (spk_pack_conv_weight) in MFMA
// B-fragment order [tap][Cin/8][Cout/32][lane 0..63][4]: lane l holds, for cout = 32*nt + (l&31),
// the four cin values 8*g + 4*(l>>5) + {0,1,2,3}.  A block stages the input halo tile of a
// TH x TW pixel region for a 32-channel chunk into LDS once ([pixel][32+4 pad] floats) and all
// taps read it with shifted pixel offsets; B fragments stream straight from L2 into registers
// (1 KiB coalesced per wave-instruction, shared by the block's four waves through L1).
// GEMM view: M = pixels (rows of the 32x32 tile), N = cout (lanes), K = (tap, cin).
// K is consumed 8 at a time: one ds_read_b128 + one global_load_dwordx4 feed four MFMAs, the
// two lane halves taking cin {0..3} and {4..7} of the group (any K order is a valid sum).
// (ABL_NO_* macros select diagnostic ablation builds - wrong results by construction - used to price each phase of the
//  kernel: SPK_CXXFLAGS="-DABL_NO_STAGE" python build.py, then SPK_LIB=<variant.so> tools/conv_bench.py.)
#include "spk_common.h"

#define CK 32
#ifndef STAGE_U
#define STAGE_U 4   // staging loads in flight per thread
#endif
#define LPS 36  // LDS floats per staged pixel: 32 + 4 pad -> conflict-free ds_read_b128

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
    int tap_off[9];   // LDS offset (float4 units) of the (tap, channel plane) inside the staged tile
    int tap_w[9];     // weight tap index
    int tap_g[9];     // weight K-group offset of the channel plane (4 groups of 8 channels per plane)
    int kc;           // channel planes (of 32) staged per barrier: > 1 only for single-tap (1x1) convolutions, whose K loop
                      // per 32-channel chunk is too short to amortise a staging phase
};

template <int MT, int NT, bool BNBWD>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

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
    const int npix_tile = a.TH * a.TW;
    const int n0 = cg * NT * 32;
    const int flags = a.flags;

    int lbase[MT], obase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int q = (wave * MT + i) * 32 + r;
        bool v = q < npix_tile;
        const int qq = v ? q : 0;
        const int ly = qq / a.TW, lx = qq - ly * a.TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        v = v && oy < a.OH && ox < a.OW;
        lbase[i] = ((ly * a.IS) * a.halo_w + lx * a.IS) * (LPS / 4) + h;   // float4 units
        obase[i] = v ? ((b * a.OHf + oy * a.OS + a.ooy) * a.OWf + ox * a.OS + a.oox) * a.Cout + n0 : -1;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunks = a.Cin / (CK * a.kc);
    const int plane_floats = a.halo_h * a.halo_w * LPS;
    const int halo_pix = a.halo_h * a.halo_w;
    const int cout32 = a.Cout >> 5;
    const int quad = tid & 7;

    for (int ch = 0; ch < nchunks; ++ch) {
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
                constexpr int U2 = 2;
                for (int base = tid >> 3; base < halo_pix; base += 32 * U2) {
                    f32x4 v[U2], rw[U2], ac[U2];
                    bool inb[U2], core[U2];
                    size_t off[U2];
#pragma unroll
                    for (int u = 0; u < U2; ++u) {
                        int p = base + 32 * u;
                        p = p < halo_pix ? p : halo_pix - 1;
                        const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
                        const int hx = p - hy * a.halo_w;
                        const int iy = iy0 + hy, ix = ix0 + hx;
                        inb[u] = iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
                        core[u] = inb[u] && iy >= oy0 && iy < oy0 + a.TH && ix >= ox0 && ix < ox0 + a.TW;
                        const int cy = min(max(iy, 0), a.IH - 1), cx = min(max(ix, 0), a.IW - 1);
                        off[u] = (size_t)((b * a.IH + cy) * a.IW + cx) * a.Cin + c;
                        v[u] = *(const f32x4*)(a.in + off[u]);
                        rw[u] = *(const f32x4*)(a.in_raw + off[u]);
                        if (a.in_act) ac[u] = *(const f32x4*)(a.in_act + off[u]);
                    }
#pragma unroll
                    for (int u = 0; u < U2; ++u) {
                        const int p = base + 32 * u;
                        const f32x4 m = a.in_act ? ac[u] : rw[u] * bsc + bsh;
                        f32x4 dz;
#pragma unroll
                        for (int k = 0; k < 4; ++k) dz[k] = m[k] > 0.f ? v[u][k] : 0.f;
                        f32x4 w = k1 * (dz - m1 - ((rw[u] - mu) * is) * m2);
                        if (!inb[u]) w = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (p < halo_pix) {
                            *(f32x4*)(ldsp + p * LPS + quad * 4) = w;
                            if (owner && core[u]) {
                                *(f32x4*)(a.side_draw + off[u]) = w;
                                if (a.side_dz) *(f32x4*)(a.side_dz + off[u]) = dz;
                            }
                        }
                    }
                }
            } else {
            // U independent 16-byte loads in flight per thread (addresses clamped, zero selected afterwards: no branch
                // around a load, so the compiler issues the whole batch before the first wait)
                constexpr int U = STAGE_U;
#ifdef ABL_NO_STAGE
                for (int base = tid >> 3; base < 0; base += 32 * U) {
#else
                for (int base = tid >> 3; base < halo_pix; base += 32 * U) {
#endif
                    f32x4 v[U];
                    bool inb[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        int p = base + 32 * u;
                        p = p < halo_pix ? p : halo_pix - 1;
                        const int hy = (int)__umulhi((unsigned)p, a.halo_w_magic);
                        const int hx = p - hy * a.halo_w;
                        const int iy = iy0 + hy, ix = ix0 + hx;
                        inb[u] = iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
                        const int cy = min(max(iy, 0), a.IH - 1), cx = min(max(ix, 0), a.IW - 1);
                        v[u] = *(const f32x4*)(a.in + (size_t)((b * a.IHp + cy * a.ips) * a.IWp + cx * a.ips) * a.Cin + c);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int p = base + 32 * u;
                        f32x4 w = v[u];
                        if (flags & SPK_IN_AFFINE_RELU) {
                            w = w * sc + sh;
                            w[0] = fmaxf(w[0], 0.f);
                            w[1] = fmaxf(w[1], 0.f);
                            w[2] = fmaxf(w[2], 0.f);
                            w[3] = fmaxf(w[3], 0.f);
                        }
                        if (!inb[u]) w = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (p < halo_pix) *(f32x4*)(ldsp + p * LPS + quad * 4) = w;
                    }
                }
            }
        }
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
    }

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
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                slab[row * LW + j * 32 + r] = acc[i][j][e];
            }
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
                if (flags & SPK_EPI_ADD) v += *(const f32x4*)(a.epi_add + ob + qc * 4);
                if (flags & SPK_EPI_RELU) {
                    v[0] = fmaxf(v[0], 0.f);
                    v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f);
                    v[3] = fmaxf(v[3], 0.f);
                }
                *(f32x4*)dst = v;
                if (flags & SPK_EPI_BNBWD) {
                    // v is the gradient wrt a BatchNorm(+ReLU) output: accumulate (sum dz, sum dz*xhat) of that BN so
                    // its backward needs no separate reduction pass over this tensor
                    const f32x4 rw = *(const f32x4*)(a.bn_raw + ob + qc * 4);
                    f32x4 m;
                    if (a.bn_act) m = *(const f32x4*)(a.bn_act + ob + qc * 4);
                    else m = rw * bsc + bsh;
                    f32x4 dz;
#pragma unroll
                    for (int c = 0; c < 4; ++c) dz[c] = m[c] > 0.f ? v[c] : 0.f;
                    ssum += dz;
                    ssq += dz * ((rw - bmu) * bis);
                } else {
                    ssum += v;
                    ssq += v * v;
                }
            }
        }
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

template <int MT, int NT>
static int launch_conv(const ConvArgs& a, size_t lds_bytes, hipStream_t st) {
    // the BatchNorm-backward input mode is a separate instantiation: its staging registers must not cost the plain
    // kernel an occupancy step
    if (a.flags & SPK_IN_BNBWD)
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, true>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, false>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_mfma");
    return 0;
}

extern "C" int spk_conv_mfma(const float* in, const float* wpk, float* out, const float* in_scale,
                             const float* in_shift, const float* epi_scale, const float* epi_shift,
                             const float* epi_add, const float* in_raw, const float* in_act, const float* in_bn4,
                             const float* in_coef, float* side_draw, float* side_dz, const float* bn_raw,
                             const float* bn_act, const float* bn4, float* stats, int B, int IH, int IW, int Cin, int OH,
                             int OW, int OHf, int OWf, int Cout, int IS, int OS, int ooy, int oox, int ntaps,
                             const int* tap_dy, const int* tap_dx, const int* tap_w, int TH, int TW, int MT,
                             int NT, int kc, int ips, int flags, void* stream) {
    SPK_REQUIRE(in && wpk && out, "spk_conv_mfma: null pointer");
    SPK_REQUIRE(B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "spk_conv_mfma: empty tensor");
    SPK_REQUIRE(Cin % 32 == 0 && Cin > 0, "spk_conv_mfma: Cin=%d must be a multiple of 32", Cin);
    SPK_REQUIRE(NT >= 1 && Cout % (32 * NT) == 0, "spk_conv_mfma: Cout=%d not a multiple of 32*NT (NT=%d)", Cout, NT);
    SPK_REQUIRE(ntaps >= 1 && ntaps <= 9, "spk_conv_mfma: ntaps=%d out of range", ntaps);
    SPK_REQUIRE(kc >= 1 && ntaps * kc <= 9 && Cin % (32 * kc) == 0, "spk_conv_mfma: kc=%d incompatible with ntaps=%d, Cin=%d", kc, ntaps, Cin);
    SPK_REQUIRE(TH >= 1 && TW >= 1 && TH * TW <= 128 * MT, "spk_conv_mfma: tile %dx%d exceeds 128*MT (MT=%d)", TH, TW, MT);
    SPK_REQUIRE(IS >= 1 && OS >= 1 && ooy >= 0 && oox >= 0, "spk_conv_mfma: bad strides/offsets");
    SPK_REQUIRE((OH - 1) * OS + ooy < OHf && (OW - 1) * OS + oox < OWf, "spk_conv_mfma: logical grid exceeds the output tensor");
    SPK_REQUIRE((long long)B * OHf * OWf * Cout < 2147483647LL && (long long)B * IH * IW * Cin < 2147483647LL * 4,
                "spk_conv_mfma: tensor too large for 32-bit element offsets");
    SPK_REQUIRE(!(flags & SPK_IN_AFFINE_RELU) || (in_scale && in_shift), "spk_conv_mfma: IN_AFFINE_RELU needs scale/shift");
    SPK_REQUIRE(!(flags & SPK_EPI_AFFINE) || (epi_scale && epi_shift), "spk_conv_mfma: EPI_AFFINE needs scale/shift");
    SPK_REQUIRE(!(flags & SPK_EPI_ADD) || epi_add, "spk_conv_mfma: EPI_ADD needs epi_add");
    SPK_REQUIRE(!(flags & SPK_EPI_STATS) || stats, "spk_conv_mfma: EPI_STATS needs a stats buffer");
    SPK_REQUIRE(!(flags & SPK_IN_BNBWD) || (in_raw && in_bn4 && in_coef && side_draw && !(flags & SPK_IN_AFFINE_RELU)),
                "spk_conv_mfma: IN_BNBWD needs in_raw, in_bn4, in_coef and side_draw (and excludes IN_AFFINE_RELU)");
    SPK_REQUIRE(!(flags & SPK_IN_BNBWD) || (IS == 1 && OS == 1 && ips == 1 && OH == IH && OW == IW && OHf == IH && OWf == IW),
                "spk_conv_mfma: IN_BNBWD is defined for stride-1 data gradients (input and output grids coincide)");
    SPK_REQUIRE(!(flags & SPK_EPI_BNBWD) || ((flags & SPK_EPI_STATS) && bn_raw && bn4),
                "spk_conv_mfma: EPI_BNBWD needs EPI_STATS, bn_raw and bn4");
    ConvArgs a;
    a.in = in; a.wpk = wpk; a.out = out; a.in_scale = in_scale; a.in_shift = in_shift;
    a.epi_scale = epi_scale; a.epi_shift = epi_shift; a.epi_add = epi_add; a.stats = stats;
    a.bn_raw = bn_raw; a.bn_act = bn_act; a.bn4 = bn4;
    a.in_raw = in_raw; a.in_act = in_act; a.in_bn4 = in_bn4; a.in_coef = in_coef; a.side_draw = side_draw; a.side_dz = side_dz;
    SPK_REQUIRE(ips >= 1 && ips <= 4, "spk_conv_mfma: ips=%d", ips);
    a.B = B; a.IHp = IH; a.IWp = IW; a.ips = ips; a.IH = (IH + ips - 1) / ips; a.IW = (IW + ips - 1) / ips; a.Cin = Cin; a.OH = OH; a.OW = OW; a.OHf = OHf; a.OWf = OWf; a.Cout = Cout;
    a.IS = IS; a.OS = OS; a.ooy = ooy; a.oox = oox; a.TH = TH; a.TW = TW;
    a.tiles_y = spk_ceil_div(OH, TH); a.tiles_x = spk_ceil_div(OW, TW);
    int mindy = 127, mindx = 127, maxdy = -127, maxdx = -127;
    for (int t = 0; t < ntaps; ++t) {
        SPK_REQUIRE(tap_dy[t] >= -8 && tap_dy[t] <= 8 && tap_dx[t] >= -8 && tap_dx[t] <= 8 && tap_w[t] >= 0 && tap_w[t] < 9,
                    "spk_conv_mfma: tap %d out of range", t);
        a.tap_w[t] = tap_w[t];
        if (tap_dy[t] < mindy) mindy = tap_dy[t];
        if (tap_dy[t] > maxdy) maxdy = tap_dy[t];
        if (tap_dx[t] < mindx) mindx = tap_dx[t];
        if (tap_dx[t] > maxdx) maxdx = tap_dx[t];
    }
    a.min_dy = mindy; a.min_dx = mindx;
    a.halo_h = (TH - 1) * IS + (maxdy - mindy) + 1;
    a.halo_w = (TW - 1) * IS + (maxdx - mindx) + 1;
    a.kc = kc;
    {   // expand (spatial tap) x (channel plane): plane-major, so one plane's taps are consecutive
        int sp_off[9], sp_w[9];
        for (int t = 0; t < ntaps; ++t) {
            sp_off[t] = ((tap_dy[t] - mindy) * a.halo_w + (tap_dx[t] - mindx)) * (LPS / 4);
            sp_w[t] = tap_w[t];
        }
        const int plane4 = a.halo_h * a.halo_w * (LPS / 4);
        for (int tt = 0; tt < ntaps * kc; ++tt) {
            const int t = tt % ntaps, pl = tt / ntaps;
            a.tap_off[tt] = sp_off[t] + pl * plane4;
            a.tap_w[tt] = sp_w[t];
            a.tap_g[tt] = pl * 4;
        }
    }
    a.halo_w_magic = (unsigned)((0x100000000ULL + (unsigned long long)a.halo_w - 1) / (unsigned long long)a.halo_w);
    a.ntaps = ntaps * kc; a.ncg = Cout / (32 * NT);
    a.nblocks = B * a.tiles_y * a.tiles_x * a.ncg;
    a.flags = flags;
    size_t lds_bytes = (size_t)kc * a.halo_h * a.halo_w * LPS * sizeof(float);
    const size_t red_bytes = (size_t)4 * 32 * (NT * 32 + 4) * sizeof(float);   // epilogue transpose slabs, one per wave
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_mfma: halo tile %dx%d needs %zu B of LDS", a.halo_h, a.halo_w, lds_bytes);
    hipStream_t st = (hipStream_t)stream;
#define CASE(M, N) if (MT == M && NT == N) return launch_conv<M, N>(a, lds_bytes, st)
    CASE(1, 1); CASE(2, 1); CASE(3, 1); CASE(4, 1);
    CASE(1, 2); CASE(2, 2); CASE(3, 2); CASE(4, 2);
    CASE(1, 4); CASE(2, 4);
#undef CASE
    spk_set_error("spk_conv_mfma: unsupported tile config MT=%d NT=%d", MT, NT);
    return -1;
}

// ---- weight packing -------------------------------------------------------------------------
// OIHW [Cout][Cin][KH][KW] -> [tap][K/8][N/32][64][4].  transpose == 0: K = cin, N = cout (forward);
// transpose == 1: K = cout, N = cin (data gradient).  Tap index t = kh*KW + kw in both cases.
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cout, int Cin,
                                        int KHW, int transpose, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int K = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
    int i = idx;
    const int s = i & 3; i >>= 2;
    const int lane = i & 63; i >>= 6;
    const int nt = i % (N >> 5); i /= (N >> 5);
    const int g = i % (K >> 3);
    const int t = i / (K >> 3);
    const int n = nt * 32 + (lane & 31);
    const int k = g * 8 + (lane >> 5) * 4 + s;
    const int co = transpose ? k : n, ci = transpose ? n : k;
    wpk[idx] = w[((size_t)co * Cin + ci) * KHW + t];
}

extern "C" int spk_pack_conv_weight(const float* w, float* wpk, int Cout, int Cin, int KH, int KW, int transpose,
                                    void* stream) {
    SPK_REQUIRE(w && wpk, "spk_pack_conv_weight: null pointer");
    SPK_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0, "spk_pack_conv_weight: channels (%d,%d) must be multiples of 32", Cout, Cin);
    SPK_REQUIRE(KH * KW >= 1 && KH * KW <= 9, "spk_pack_conv_weight: kernel %dx%d unsupported", KH, KW);
    const int total = Cout * Cin * KH * KW;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(spk_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       wpk, Cout, Cin, KH * KW, transpose, total);
    SPK_LAUNCH_CHECK("spk_pack_conv_weight");
    return 0;
}
