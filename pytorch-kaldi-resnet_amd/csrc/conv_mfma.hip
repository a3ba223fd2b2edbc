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
#include "conv_kernel.h"

// bf16-split instantiations live in their own translation unit (conv_split.hip)
int spk_launch_conv_split(const ConvArgs& a, size_t lds_bytes, int MT, int NT, int split, hipStream_t st);
// in-wave pipelined form (conv_pipe.hip)
int spk_launch_conv_pipe(const ConvArgs& a, size_t lds_bytes, int MT, int NT, hipStream_t st);
// wave-specialised persistent form (conv_ws.hip)
int spk_launch_conv_ws(const ConvArgs& a, int MT, int NT, int WC, int split, int lp4, hipStream_t st);

template <int MT, int NT>
static int launch_conv(const ConvArgs& a, size_t lds_bytes, hipStream_t st) {
    // the BatchNorm-backward input mode is a separate instantiation: its staging registers must not cost the plain
    // kernel an occupancy step
    if (a.flags & SPK_IN_BNBWD)
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, true, 0>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, false, 0>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_mfma");
    return 0;
}

extern "C" int spk_conv_mfma(const float* in, const float* wpk, float* out, const float* in_scale,
                             const float* in_shift, const float* epi_scale, const float* epi_shift,
                             const float* epi_add, const float* in_raw, const float* in_act, const float* in_bn4,
                             const float* in_coef, const unsigned* in_mask, const unsigned* bn_mask, const unsigned* add_mask, float* side_draw,
                             float* side_dz,
                             const float* bn_raw,
                             const float* bn_act, const float* bn4, float* stats, int B, int IH, int IW, int Cin, int OH,
                             int OW, int OHf, int OWf, int Cout, int IS, int OS, int ooy, int oox, int ntaps,
                             const int* tap_dy, const int* tap_dx, const int* tap_w, int TH, int TW, int MT,
                             int NT, int kc, int ips, int flags, int split, const unsigned* in_amax, unsigned* out_amax,
                             unsigned* side_amax, void* stream) {
    SPK_REQUIRE(in && wpk && out, "spk_conv_mfma: null pointer");
    SPK_REQUIRE(B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "spk_conv_mfma: empty tensor");
    SPK_REQUIRE(Cin % 32 == 0 && Cin > 0, "spk_conv_mfma: Cin=%d must be a multiple of 32", Cin);
    // wave-specialised kernel: bits 8-9 of flags = log2 of WC, the number of consumer-wave channel groups (1, 2, 4)
    const int ws_wc = (flags & SPK_CONV_WS) ? 1 << ((flags >> 8) & 3) : 1;
    SPK_REQUIRE(ws_wc <= 4, "spk_conv_mfma: bad wave layout");
    SPK_REQUIRE(NT >= 1 && Cout % (32 * NT * ws_wc) == 0, "spk_conv_mfma: Cout=%d not a multiple of 32*NT*WC (NT=%d, WC=%d)", Cout, NT, ws_wc);
    SPK_REQUIRE(ntaps >= 1 && ntaps <= 9, "spk_conv_mfma: ntaps=%d out of range", ntaps);
    SPK_REQUIRE(split == 0 || split == 3 || split == 6 || split == 9,
                "spk_conv_mfma: split=%d (0 = fp32 operands, 6 / 9 = bf16 cross terms, 3 = fp16 two-term operands)", split);
    const int ck = split ? SPK_SPLIT_CK : 32;                     // channels per staged plane
    const int nterm = split == 3 ? 2 : 3;
    const int lp4 = split ? (nterm * SPK_SPLIT_CK * 2 + 16) / 16 : 9;    // LDS pixel pitch in 16-byte units (ConvCfg<SPLIT>::LP4)
    SPK_REQUIRE(kc >= 1 && ntaps * kc <= 9 && Cin % (ck * kc) == 0, "spk_conv_mfma: kc=%d incompatible with ntaps=%d, Cin=%d", kc, ntaps, Cin);
    SPK_REQUIRE(TH >= 1 && TW >= 1 && TH * TW <= 128 * MT / ws_wc, "spk_conv_mfma: tile %dx%d exceeds %d pixels (MT=%d)", TH, TW, 128 * MT / ws_wc, MT);
    SPK_REQUIRE(IS >= 1 && OS >= 1 && ooy >= 0 && oox >= 0, "spk_conv_mfma: bad strides/offsets");
    SPK_REQUIRE((OH - 1) * OS + ooy < OHf && (OW - 1) * OS + oox < OWf, "spk_conv_mfma: logical grid exceeds the output tensor");
    SPK_REQUIRE((long long)B * OHf * OWf * Cout < 2147483647LL && (long long)B * IH * IW * Cin < 2147483647LL * 4,
                "spk_conv_mfma: tensor too large for 32-bit element offsets");
    SPK_REQUIRE((long long)IH * IW * Cin < 2147483647LL, "spk_conv_mfma: one image exceeds 32-bit element offsets");
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
    // f16x3: the operand scale ALWAYS comes from a slot (an absmax written by the producer, or a rigorous bound): a static scale
    // would clamp |v| > 1023 and flush small gradients silently
    SPK_REQUIRE(split != 3 || in_amax, "spk_conv_mfma: the f16x3 operand mode needs in_amax (slot with the float bits of the staged tensor's absmax or of an upper bound)");
    SPK_REQUIRE(!(flags & SPK_IN_PRESPLIT) || (split == 3 && !(flags & (SPK_IN_AFFINE_RELU | SPK_IN_BNBWD | SPK_CONV_WS))),
                "spk_conv_mfma: IN_PRESPLIT (f16 pair input) needs the f16x3 mode, a plain input, and not the wave-specialised kernel");
    SPK_REQUIRE(!(flags & SPK_CONV_M16) || (flags & SPK_CONV_PIPE), "spk_conv_mfma: CONV_M16 is a form of the pipelined kernel (CONV_PIPE)");
    SPK_REQUIRE(!(flags & SPK_SIDE_PRESPLIT) || ((flags & SPK_IN_BNBWD) && split == 3 && !(flags & (SPK_CONV_WS | SPK_CONV_PIPE))),
                "spk_conv_mfma: SIDE_PRESPLIT (f16 pair side output) exists for the fused BatchNorm-backward form of conv_mfma_kernel in the f16x3 mode");
    ConvArgs a;
    a.in = in; a.wpk = wpk; a.out = out; a.in_scale = in_scale; a.in_shift = in_shift;
    a.epi_scale = epi_scale; a.epi_shift = epi_shift; a.epi_add = epi_add; a.stats = stats;
    a.bn_raw = bn_raw; a.bn_act = bn_act; a.bn4 = bn4;
    a.in_raw = in_raw; a.in_act = in_act; a.in_bn4 = in_bn4; a.in_coef = in_coef; a.side_draw = side_draw; a.side_dz = side_dz;
    a.in_mask = in_mask; a.bn_mask = bn_mask; a.add_mask = add_mask;
    a.in_amax = in_amax; a.out_amax = out_amax; a.side_amax = side_amax;
    a.in_sigma = SPK_F16_ACT_SIGMA; a.w_amax = nullptr;
    if (split == 3) {       // fp16-split packs start with a 16-byte header: the float bits of max|w|
        a.w_amax = (const unsigned*)wpk;
        a.wpk = wpk + 4;
    }

    SPK_REQUIRE(!add_mask || ((flags & SPK_EPI_ADD) && Cout % 32 == 0), "spk_conv_mfma: add_mask needs EPI_ADD and Cout %% 32 == 0");
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
            sp_off[t] = ((tap_dy[t] - mindy) * a.halo_w + (tap_dx[t] - mindx)) * lp4;
            sp_w[t] = tap_w[t];
        }
        const int plane4 = a.halo_h * a.halo_w * lp4;
        for (int tt = 0; tt < ntaps * kc; ++tt) {
            const int t = tt % ntaps, pl = tt / ntaps;
            a.tap_off[tt] = sp_off[t] + pl * plane4;
            a.tap_w[tt] = sp_w[t];
            a.tap_g[tt] = pl * (split ? SPK_SPLIT_CK / 16 : 4);
        }
    }
    a.halo_w_magic = (unsigned)((0x100000000ULL + (unsigned long long)a.halo_w - 1) / (unsigned long long)a.halo_w);
    a.ntaps = ntaps * kc; a.ncg = Cout / (32 * NT * ws_wc);
    a.nblocks = B * a.tiles_y * a.tiles_x * a.ncg;
    a.flags = flags;
    size_t lds_bytes = (size_t)kc * a.halo_h * a.halo_w * lp4 * 16;
    const size_t red_bytes = (size_t)4 * 32 * (NT * 32 + 4) * sizeof(float);   // epilogue transpose slabs, one per wave
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    SPK_REQUIRE((flags & SPK_CONV_WS) || lds_bytes <= 160 * 1024, "spk_conv_mfma: halo tile %dx%d needs %zu B of LDS", a.halo_h, a.halo_w, lds_bytes);
    hipStream_t st = (hipStream_t)stream;
    if (flags & SPK_CONV_WS) {
#ifndef SPK_EXPERIMENTAL
        SPK_REQUIRE(false, "spk_conv_mfma: the wave-specialised kernel is an experimental form: build with SPK_EXPERIMENTAL=1 (spk_build_flags)");
#else
        SPK_REQUIRE(split != 0 && kc == 1 && ntaps == 9, "spk_conv_mfma: the wave-specialised kernel needs bf16-split operands, 9 taps and kc = 1");
        for (int t = 0; t < 9; ++t) {
            const long long off = (long long)a.tap_w[t] * (Cin >> 4) * nterm * (Cout >> 5) * 256;     // [tap][Cin/16][term][Cout/32][256 floats]
            SPK_REQUIRE(off < 2147483647LL, "spk_conv_mfma: packed weight offset overflows");
            a.tap_boff[t] = (int)off;
        }
        return spk_launch_conv_ws(a, MT, NT, ws_wc, split, lp4, st);
#endif
    }
    if (flags & SPK_CONV_PIPE) {
        SPK_REQUIRE(split == 3 && kc == 1 && ntaps == 9, "spk_conv_mfma: the pipelined kernel needs f16x3 operands, 9 taps and kc = 1");
        SPK_REQUIRE(!(flags & SPK_IN_BNBWD) || !in_act || in_mask,
                    "spk_conv_mfma: the pipelined kernel takes the ReLU mask of a fused BatchNorm backward as sign bits or recomputes it");
        SPK_REQUIRE(a.halo_h * a.halo_w <= 9 * 64, "spk_conv_mfma: the pipelined kernel stages at most 576 halo pixels (%d x %d)", a.halo_h, a.halo_w);
        size_t lds2 = (2 * (size_t)a.halo_h * a.halo_w + 1) * lp4 * 16;      // two tiles + the dump pixel
        if (lds2 < red_bytes) lds2 = red_bytes;
        SPK_REQUIRE(lds2 <= 160 * 1024, "spk_conv_mfma: two halo tiles %dx%d need %zu B of LDS", a.halo_h, a.halo_w, lds2);
        if (flags & SPK_CONV_M16) {
            SPK_REQUIRE(!(flags & SPK_IN_BNBWD), "spk_conv_mfma: CONV_M16 excludes the fused BatchNorm backward");
            SPK_REQUIRE(a.halo_h * a.halo_w <= 8 * 64, "spk_conv_mfma: CONV_M16 stages at most 512 halo pixels (%d x %d)", a.halo_h, a.halo_w);
            SPK_REQUIRE((Cin / SPK_SPLIT_CK) % 2 == 0, "spk_conv_mfma: CONV_M16 walks the 16-channel planes in pairs");
        }
        a.flags = flags & ~SPK_CONV_PIPE;
        return spk_launch_conv_pipe(a, lds2, MT, NT, st);
    }
    if (split) return spk_launch_conv_split(a, lds_bytes, MT, NT, split, st);
#define CASE(M, N) if (MT == M && NT == N) return launch_conv<M, N>(a, lds_bytes, st)
    CASE(1, 1); CASE(2, 1); CASE(3, 1); CASE(4, 1);
    CASE(1, 2); CASE(2, 2); CASE(3, 2); CASE(4, 2);
    CASE(1, 4); CASE(2, 4);
#undef CASE
    spk_set_error("spk_conv_mfma: unsupported tile config MT=%d NT=%d", MT, NT);
    return -1;
}
