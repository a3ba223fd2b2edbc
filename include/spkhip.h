/* libspkhip - C ABI of the MI355X (gfx950) speaker-embedding hot path.
 *
 * Drop-in boundary: the reference (ZihanLiao/pytorch-kaldi-resnet) has no native
 * layer; its seam is the torch.nn.Module protocol of NeuralSpeakerModel (scripts/model.py:334-432) as
 * driven by scripts/train_resnet.py:304-328 and scripts/decode.py:185-208.  Every torch op on that path
 * maps to one export below; the Python mirror of NeuralSpeakerModel binds them with ctypes
 * (pytorch-kaldi-resnet_amd/hip.py).  Each export cites the reference call site it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise
 *   - the caller owns every buffer, including workspaces; the library never allocates device memory
 *     and keeps no pointer across calls
 *   - every call is an asynchronous launch on `stream` (a hipStream_t) and never synchronises
 *   - return value: 0 = ok, < 0 = invalid argument (see spk_last_error), > 0 = hipError_t of the launch
 *   - activations are NHWC fp32: [B][H = mel bins][W = frames][C]; the network input is [B][F][T]
 *     (reference layout, scripts/datasets.py:68, scripts/model.py:247) which is NHWC with C = 1
 *   - arithmetic is fp32 throughout (MFMA v_mfma_f32_32x32x2_f32 = exact fp32 FMA chain)
 */
#ifndef SPKHIP_H
#define SPKHIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int spk_version(void);
const char* spk_last_error(void);
/* bit 0: the library was built with SPK_EXPERIMENTAL=1 and contains the kernel forms that were measured and did not pay
 * (SPK_CONV_WS of spk_conv_mfma / spk_conv_wgrad, SPK_CONV_PIPE of spk_conv_wgrad, SPK_CONV_PIPE + SPK_IN_BNBWD of spk_conv_mfma);
 * without it those flag combinations are argument errors */
int spk_build_flags(void);

/* flags for the fused input transform / epilogue of the convolutions */
#define SPK_IN_AFFINE_RELU 1 /* input tile = max(in*in_scale[c] + in_shift[c], 0): BN+ReLU of the producer, fused */
#define SPK_EPI_AFFINE 2     /* out = acc*epi_scale[c] + epi_shift[c]   (eval-mode BN folded into the conv) */
#define SPK_EPI_ADD 4        /* out += epi_add[same index]              (residual add / gradient accumulation) */
#define SPK_EPI_RELU 8       /* out = max(out, 0) */
#define SPK_EPI_STATS 16     /* stats[4*pixel_tile + wave][c] = (sum, sumsq) of the stored values (train-mode BN) */
#define SPK_IN_BNBWD 64      /* stride-1 data gradients: the staged input is BatchNorm-backward(in) computed on the fly from
                                in, in_raw, the mask (in_act > 0, or in_raw*scale+shift > 0 when in_act is NULL), in_bn4
                                [mean,invstd,scale,shift][Cin] and in_coef [3][Cin] (spk_bn_bwd_finalize); the tile owner
                                writes that value to side_draw (for the weight gradient) and in*mask to side_dz (optional) */
#define SPK_EPI_BNBWD 32     /* with EPI_STATS, data-gradient launches: stats = (sum dz, sum dz*xhat) of the BatchNorm whose
                                output gradient this launch produces (dz = out * mask; mask = bn_act > 0, or
                                bn_raw*scale+shift > 0 when bn_act is NULL; bn4 = [mean, invstd, scale, shift][Cout]) */

/* kernel form (results are bit-identical to the plain kernel on the same tile, SPK_CONV_M16 excepted); "experimental" = needs spk_build_flags() & 1 */
#define SPK_CONV_WS 128      /* experimental: producer / consumer form of spk_conv_mfma (split != 0, 9 taps, kc = 1; flags bits 8-9 = log2 of
                                its consumer-wave channel groups) and of spk_conv_wgrad (split = 3, 3x3) */
#define SPK_CONV_PIPE 1024   /* in-wave pipelined form: spk_conv_mfma (split = 3, 9 taps, kc = 1, <= 576 halo pixels, two halo
                                tiles in LDS; plain or f16 pair input); experimental: with SPK_IN_BNBWD (in_mask given, MT*NT <= 4)
                                and for spk_conv_wgrad (split = 3, 3x3, tile of 1 or 2 k-steps of 16 pixels per wave group) */

#define SPK_CONV_M16 (1 << 17) /* with SPK_CONV_PIPE on spk_conv_mfma (plain or f16 pair input, MT = 3, NT = 2, <= 512 halo pixels,
                                Cin % 32 == 0): the same kernel on v_mfma_f32_16x16x32_f16, two taps per 32-deep K step.  Same products,
                                another summation order: equal to the other forms within fp32 accumulation error, not bit-identical */

#define SPK_WGRAD_GROUPS 2048 /* spk_conv_wgrad, split = 3: 1x1: the kernel that gives a block 1 << (flags bits 12-13) = 2 or 4
                                groups of 32 input channels (Cin % (32 * groups) == 0, WN 2 or 4, tile <= 64 pixels); 3x3: the
                                2 x 2 (input-channel group x output-channel group) wave layout (groups = 2, WN = 2) */

/* f16 pair tensors (split = 3).  A tensor with the shape and addressing of an fp32 NHWC tensor in which every aligned group of
 * four floats (16 bytes) holds [4 x fp16 high term][4 x fp16 low term] of value * sigma, sigma = the power of two that takes
 * the float whose bits the tensor's scale slot holds into [2^14, 2^15).  The slot holds a RIGOROUS upper bound of the tensor's
 * absmax, known before the producer runs (spk_bn_bwd_finalize est_out).  Producers: spk_bn_bwd_apply(pair_scale), and the side
 * output of a fused BatchNorm-backward data gradient (SPK_SIDE_PRESPLIT).  Consumers stage 16-byte groups by plain copy. */
#define SPK_WGRAD_NOSHIFT (1 << 18) /* spk_conv_wgrad + SPK_WGRAD_GROUPS, 3x3: at stride 1 with TW % 8 == 0 the kernel builds the fragments of
                                the three taps of a filter row from one 10-pixel window per lane (22 instead of 40 transposed LDS reads per
                                k-step; bit-identical results); this flag keeps the plain K loop (A/B measurements) */
#define SPK_WGRAD_M16 (1 << 19) /* spk_conv_wgrad + SPK_WGRAD_GROUPS, 3x3, with SPK_DY_PRESPLIT: conv_wgrad_wm16_kernel - the 2 x 2 wave layout on
                                v_mfma_f32_16x16x32_f16 (32 pixels per K step), dy brought into LDS by global_load_lds (no registers, no
                                staging instructions; two LDS buffers).  LDS: halo * 384 + 2 * ceil32(TH * TW) * 256 bytes.  Same products,
                                another summation order: equal to the other weight gradients within fp32 accumulation error.
                                WITHOUT SPK_WGRAD_GROUPS (3x3, WN = 1, SPK_DY_PRESPLIT): conv_wgrad_c32m16_kernel - the same kernel in its
                                layout for 32-channel groups (the first layer): a block owns 32 x 32 channels, its four waves split the
                                32-pixel k-steps of a region and fold their tiles through LDS at the end; halo <= 192 pixels, LDS: halo *
                                192 + 2 * ceil32(TH * TW) * 128 bytes (at least 36 KB); one slab per block as every other form */
#define SPK_IN_PRESPLIT (1 << 14)    /* spk_conv_mfma: `in` is an f16 pair tensor scaled by the sigma of *in_amax (plain input only) */
#define SPK_SIDE_PRESPLIT (1 << 15)  /* spk_conv_mfma + SPK_IN_BNBWD: side_draw leaves as an f16 pair tensor (scale: *in_amax) */
#define SPK_DY_PRESPLIT (1 << 16)    /* spk_conv_wgrad: `dy` is an f16 pair tensor scaled by the sigma of *dy_amax */

/* ---- convolutions --------------------------------------------------------------------------------- */

/* nn.Conv2d weight [Cout][Cin][KH][KW] (scripts/model.py:12-15,105-110,233-234) -> MFMA fragment order
 * [tap][K/8][N/32][64][4]; transpose = 0 for the forward conv (K = Cin), 1 for its data gradient (K = Cout). */
int spk_pack_conv_weight(const float* w, float* wpk, int Cout, int Cin, int KH, int KW, int transpose, void* stream);
/* the same weights for the split operand modes of spk_conv_mfma: split = 6 or 9: every weight as three bf16 terms whose sum
 * is the fp32 value, [tap][K/16][term][N/32][64][8 bf16] = 6 bytes per weight; split = 3: a 16-byte header (word 0 = float
 * bits of max|w|) followed by w * sigma as two fp16 terms in the same order (4 bytes per weight), sigma = the power of two
 * that puts max|w| in [2^14, 2^15) - spk_conv_mfma reads the header and derives the same sigma */
int spk_pack_conv_weight_split(const float* w, void* wpk, int Cout, int Cin, int KH, int KW, int transpose, int split,
                               void* stream);
/* every convolution of the network in one launch: `jobs` is a device array of njobs 48-byte entries
 * { const float* w; void* wpk; int Cout, Cin, KH*KW, transpose, split, total = Cout*Cin*KH*KW, block0, pad; } ordered by
 * block0 (first 256-thread block of the job; total_blocks = sum of ceil(total/256)).  spk_pack_job_bytes() = 48. */
int spk_pack_job_bytes(void);
int spk_pack_conv_weights_batched(const void* jobs, int njobs, int total_blocks, int has_f16 /* the table holds split = 3
                                  jobs: their absmax headers are refreshed first */, void* stream);

/* Implicit-GEMM convolution described by a tap table; replaces F.conv2d forward (scripts/model.py:51,56,
 * 118,122,126,58-59) and, with a transposed pack and mirrored taps, its data gradient (autograd of the same,
 * scripts/train_resnet.py:327).  Logical output pixel (oy,ox) of an OH x OW grid reads input pixel
 * (oy*IS + tap_dy[t], ox*IS + tap_dx[t]) with weight tap tap_w[t] and is stored at (oy*OS + ooy, ox*OS + oox)
 * of the physical [B][OHf][OWf][Cout] output.  TH x TW = pixel region per block (<= 128*MT pixels),
 * MT in 1..4 m-tiles per wave, NT in {1,2,4} 32-channel n-tiles per block, kc = 32-channel planes staged per barrier
 * (1 for 3x3; up to 4 for single-tap 1x1 convolutions; ntaps*kc <= 9, Cin % (32*kc) == 0).  ips = input pixel stride:
 * the taps address a logical input grid whose pixel (y,x) is physical pixel (y*ips, x*ips) - a strided 1x1 convolution
 * is run as IS = 1, ips = 2 so that only the pixels it uses are staged.
 * stats (EPI_STATS): [4*B*ceil(OH/TH)*ceil(OW/TW)][Cout][2] floats (one partial row per wave).
 * split: 0 = fp32 operands on v_mfma_f32_32x32x2_f32 (wpk from spk_pack_conv_weight); 6 or 9 = operands split exactly into
 * three bf16 terms while staged / packed (wpk from spk_pack_conv_weight_split) and the 6 most significant (or all 9) cross
 * terms multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation - inputs, outputs and measured accuracy are fp32
 * (planes are then 16 channels: Cin % (16*kc) == 0).
 * split = 3 ("f16x3", 3x3 and 1x1): operands as two fp16 terms of value * sigma, sigma a power of two (weights: from max|w| at
 * pack time; staged input: 2^(14 - exponent) of the float whose bits *in_amax holds - REQUIRED: the absmax of the staged
 * values, written by the kernel that produced the tensor, or a rigorous upper bound of it (spk_bn_finalize est_out for a fused
 * input BatchNorm+ReLU, spk_bn_bwd_finalize est_out for a fused BatchNorm backward) - so that the largest staged magnitude
 * lands in [2^14, 2^15) and nothing can saturate), three cross products on v_mfma_f32_32x32x16_f16, fp32 accumulation,
 * accumulators scaled back by 1/(sigma_in * sigma_w).  Values below bound * 2^-18 have a subnormal low term (the instruction
 * keeps fp16 subnormals: absolute error <= bound * 2^-39, relative precision falling from 22 to 11 bits at bound * 2^-28).
 * Measured accuracy = the fp32 instruction's (tools/probe/split_probe.hip).  out_amax / side_amax (optional, any split): the
 * launch atomically maxes the float bits of |stored output| / |side_draw| into them - the in_amax of the kernels that consume
 * those tensors (side_amax records the true absmax also when side_draw leaves as an f16 pair tensor).
 * flags: the SPK_* bits above. */
int spk_conv_mfma(const float* in, const float* wpk, float* out, const float* in_scale, const float* in_shift,
                  const float* epi_scale, const float* epi_shift, const float* epi_add, const float* in_raw,
                  const float* in_act, const float* in_bn4, const float* in_coef, const unsigned* in_mask /* sign bits of
                  in_act, [pixel][Cin/32] words: read instead of it */, const unsigned* bn_mask /* same for bn_act,
                  [pixel][Cout/32] */, const unsigned* add_mask /* EPI_ADD adds only where the bit is set */,
                  float* side_draw, float* side_dz,
                  const float* bn_raw, const float* bn_act, const float* bn4, float* stats, int B, int IH,
                  int IW, int Cin, int OH, int OW, int OHf, int OWf, int Cout, int IS, int OS, int ooy, int oox,
                  int ntaps, const int* tap_dy /*host*/, const int* tap_dx /*host*/, const int* tap_w /*host*/, int TH,
                  int TW, int MT, int NT, int kc, int ips, int flags, int split, const unsigned* in_amax, unsigned* out_amax,
                  unsigned* side_amax, void* stream);

/* Weight gradient of a 3x3 (pad 1) or 1x1 (pad 0) conv at stride 1 or 2 (autograd of nn.Conv2d).
 * x: conv input [B][IH][IW][Cin] (optionally raw + fused BN/ReLU via in_scale/in_shift and SPK_IN_AFFINE_RELU),
 * dy: gradient of the raw conv output [B][OH][OW][Cout]; dw: OIHW [Cout][Cin][k][k].
 * partial: workspace of spk_conv_wgrad_workspace(nsplit, ksize, Cin, Cout) bytes. TW must be even; the region must fit
 * the kernel's register prefetch window: halo pixels <= *max_halo_pix, TH*TW <= *max_tile_pix (spk_conv_wgrad_limits). */
int spk_conv_wgrad_limits(int WN, int* max_halo_pix /*host*/, int* max_tile_pix /*host*/);
size_t spk_conv_wgrad_workspace(int nsplit, int ksize, int Cin, int Cout);
/* Streaming 3x3 forward convolution of the 32-channel layer: 32 -> 32 channels, stride 1, padding 1, f16x3 operands - conv1 / conv2 of
 * the BasicBlocks of layer 1 (scripts/model.py:12-15,48-64).  Persistent blocks over 8 x 16-pixel tiles, the weights as matrix-core
 * fragments in LDS for the life of a block, the 10 x 18 halo staged once per tile from whole 128-byte lines one tile ahead (fused
 * BatchNorm + ReLU with SPK_IN_AFFINE_RELU), stores from the accumulator layout.  in / out: [B][H][W][32]; wpk:
 * spk_pack_conv_weight_split(split = 3, transpose = 0) of the [32][32][3][3] weights; flags: SPK_EPI_STATS [| SPK_IN_AFFINE_RELU];
 * stats: [4 * nblocks][32][2] partial rows (sum, sum of squares) for spk_bn_finalize; in_amax REQUIRED, out_amax optional;
 * nblocks: persistent blocks (<= tiles; two per CU).  Same products as spk_conv_mfma; another order of the statistics partials. */
int spk_conv3x3_c32_stream(const float* in, const float* wpk, float* out, const float* in_scale, const float* in_shift, float* stats,
                           int B, int H, int W, int flags, const unsigned* in_amax, unsigned* out_amax, int nblocks, void* stream);
/* Streaming 1x1 convolution, C -> C channels (C = 32, 64 or 128), stride 1, f16x3 operands: forward and data gradient of the 1x1
 * convolutions of the Bottleneck blocks (scripts/model.py:104-110,118-126) - a GEMM [P pixels][C] x [C][C] by persistent blocks that
 * keep the whole weight matrix in registers, stage every pixel once (whole 128-byte lines; fused BatchNorm + ReLU with
 * SPK_IN_AFFINE_RELU, plain copy of an f16 pair tensor with SPK_IN_PRESPLIT) and store from the accumulator layout.  in / out /
 * epi_add / bn_raw: [P][C]; wpk: spk_pack_conv_weight_split(split = 3) of the 1x1 weights (transpose = 1 for the data gradient);
 * flags: SPK_IN_AFFINE_RELU | SPK_IN_PRESPLIT | SPK_EPI_STATS | SPK_EPI_ADD (+ add_mask: add only where the bit is set) |
 * SPK_EPI_BNBWD (statistics of the BatchNorm-backward of bn4 = [mean, invstd, scale, shift][C] over dz = out * mask, mask = the
 * bits of bn_mask, or bn_raw * scale + shift > 0 when bn_mask is NULL).  stats: [spk_conv1x1_stream_rows(nblocks, C)][C][2]
 * partial rows for spk_bn_finalize / spk_bn_bwd_finalize.  in_amax: REQUIRED (slot of the staged values, as spk_conv_mfma);
 * out_amax: optional.  nblocks: persistent blocks (<= tiles of 64 / 128 / 256 pixels at C = 128 / 64 / 32; two per CU is the
 * measured choice).  Same arithmetic as spk_conv_mfma on these launches; another summation order of the statistics partials. */
int spk_conv1x1_stream(const float* in, const float* wpk, float* out, const float* in_scale, const float* in_shift,
                       const float* epi_add, const unsigned* add_mask, const float* bn_raw, const unsigned* bn_mask,
                       const float* bn4, float* stats, long long P, int C, int flags, const unsigned* in_amax, unsigned* out_amax,
                       int nblocks, void* stream);
int spk_conv1x1_stream_rows(int nblocks, int C);
/* spk_conv_wgrad writes nsplit partial slabs [nsplit][k*k][Cin][Cout] (nsplit <= number of pixel regions); spk_wgrad_reduce
 * sums them in a fixed order into dw (OIHW), optionally accumulating.  (`dw`/`accumulate` of spk_conv_wgrad are unused.) */
int spk_wgrad_reduce(const float* partial, float* dw, int nslab, int ksize, int Cin, int Cout, int accumulate, void* stream);
int spk_conv_wgrad(const float* x, const float* dy, float* dw, float* partial, const float* in_scale,
                   const float* in_shift, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int ksize, int stride,
                   int TH, int TW, int WN, int nsplit, int flags, int accumulate, int split /* 0, or 6 / 9 = bf16-split operands
                   (3x3 only), 3 = fp16 two-term operands (3x3 and 1x1), see spk_conv_mfma */, const unsigned* dy_amax /* split 3,
                   required: float bits of absmax(dy) or of an upper bound: fixes the dY operand scale (the scale slot of dy when
                   SPK_DY_PRESPLIT) */, const unsigned* x_amax /* split 3, required: the same for the (transformed) x operand:
                   its absmax or a bound (spk_bn_finalize est_out / spk_affine_estimate) */, void* stream);

/* Stem Conv2d(1,32,3,1,1,bias=False) (scripts/model.py:210,249): x [B][F][T] -> out [B][F][T][32];
 * stats (EPI_STATS): [spk_stem_fwd_blocks()][32][2]. */
int spk_stem_fwd_blocks(int B, int F, int T);
int spk_stem_conv_fwd(const float* x, const float* w, float* out, float* stats, const float* epi_scale,
                      const float* epi_shift, int B, int F, int T, int flags,
                      unsigned* amax_out /* optional: atomicMax of the float bits of |out| */, void* stream);
/* its weight gradient; partial: [spk_stem_wgrad_blocks()][32*9] floats */
int spk_stem_wgrad_blocks(int B, int F, int T);
int spk_stem_conv_wgrad(const float* x, const float* draw, float* dw, float* partial, int B, int F, int T,
                        int accumulate, void* stream);

/* ---- batch normalisation (nn.BatchNorm2d/1d, scripts/model.py:41,44,212,235,361) ------------------- */
/* x viewed as [N][C], C a power of two in [4,1024] */
int spk_bn_stats_blocks(long long N, int C);
int spk_bn_stats_partial(const float* x, float* partial /*[blocks][C][2]*/, long long N, int C, void* stream);
/* partial -> batch mean / invstd, scale = gamma*invstd, shift = beta - mean*scale; running stats updated with
 * momentum and unbiased variance, *num_batches_tracked += 1 (pass NULLs to skip the running update) */
/* ws: fp64 workspace of spk_bn_finalize_workspace(nblk, C) bytes (0 => may be NULL); used to fold many partial
 * rows in parallel before the fixed-order final sum */
size_t spk_bn_finalize_workspace(int nblk, int C);
int spk_bn_finalize(const float* partial, int nblk, int C, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, long long* num_batches_tracked, float* mean, float* invstd,
                    float* scale, float* shift, float momentum, float eps, double* ws,
                    const unsigned* amax_in, unsigned* est_out /* optional (f16x3 hand-off): *est_out = atomicMax over channels
                    of bits(|scale_c| * A + |shift_c|), A = the float in *amax_in = absmax of the normalised tensor: an upper
                    bound of |relu(bn(raw))|, the operand scale input of the convolution that applies this BN while staging */,
                    void* stream);
/* eval mode: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale */
int spk_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float* scale, float* shift, int C, float eps, void* stream);
/* out = [relu](raw*scale + shift [+ res | + res*res_scale + res_shift])  (BasicBlock tail, scripts/model.py:58-62).
 * mask_out (optional, C % 32 == 0): sign bits of `out`, one uint32 per 32 channels of a pixel ([N][C/32]); the backward
 * kernels read these bits (spk_conv_mfma in_mask / bn_mask) instead of the activated tensor. */
int spk_bn_apply(const float* raw, const float* scale, const float* shift, const float* res, const float* res_scale,
                 const float* res_shift, float* out, unsigned* mask_out, long long N, int C, int relu,
                 unsigned* amax_out /* optional: atomicMax of the float bits of |out| */, void* stream);
/* backward; mask_mode 0: dz = dy, 1: dz = dy*(act > 0), 2: dz = dy*(raw*scale+shift > 0), 3: as 1 with `act` pointing at the
 * sign bits of the activated tensor ([pixel][C/32] words, spk_bn_apply mask_out) instead of the tensor */
int spk_bn_bwd_reduce(const float* dy, const float* raw, const float* act, const float* mean, const float* invstd,
                      const float* scale, const float* shift, float* partial, long long N, int C, int mask_mode,
                      unsigned* chan_amax /* optional [C], zeroed by the caller: atomicMax of the float bits of |dz| per CHANNEL
                      (non-finite values left out) - the per-channel A of spk_bn_bwd_finalize */, void* stream);
int spk_bn_bwd_finalize(const float* partial, int nblk, int C, double count, const float* gamma, const float* invstd,
                        float* dgamma, float* dbeta, float* coef /*[3][C]*/, int accumulate, double* ws,
                        const unsigned* amax_in, const unsigned* raw_amax, const float* mean, unsigned* est_out /* optional
                        (needs the three before it): the spk_bnbwd_estimate bound, per channel, by atomicMax - saves that launch */,
                        const unsigned* chan_amax /* optional [C] (spk_bn_bwd_reduce): the bound pairs every channel's own absmax
                        of dz with its own k1 instead of the tensor-wide *amax_in (which may then be NULL).  Needed where the
                        incoming gradient's range is unbounded per channel: the pooling layer's sqrt'(mean) at a tiny mean
                        (scripts/model.py:453) is huge exactly in channels whose BatchNorm gamma is tiny */,
                        void* stream);
int spk_bn_bwd_apply(const float* dy, const float* raw, const float* act, const float* mean, const float* invstd,
                     const float* scale, const float* shift, const float* coef, float* draw, float* dz_out, long long N,
                     int C, int mask_mode, unsigned* amax_out /* optional: atomicMax of the float bits of |draw| */,
                     const unsigned* pair_scale /* optional: draw is written as an f16 pair tensor (see SPK_IN_PRESPLIT) scaled
                     by the sigma of *pair_scale - the est_out slot of spk_bn_bwd_finalize */, void* stream);
/* operand-scale hand-offs of the f16x3 mode (see spk_conv_mfma): *slot = max(*slot, bits(max|x|)); and the RIGOROUS upper bound
 * max_c |k1_c| (A + |m1_c| + (R + |mean_c|) invstd_c |m2_c|) (1 + 2^-16) of the values k1 (dz - m1 - xhat m2) of a BatchNorm
 * backward - staged by a fused data gradient or written as an f16 pair tensor - from the coefficient rows coef[3][C] of
 * spk_bn_bwd_finalize, the BatchNorm's mean / invstd rows, A = the float in *amax_in (absmax of the incoming gradient, so
 * |dz| <= A) and R = the float in *raw_amax (absmax of the raw tensor, so |xhat| <= (R + |mean|) invstd) */
int spk_absmax(const float* x, unsigned* slot, long long n, void* stream);
int spk_bnbwd_estimate(const float* coef, const float* mean, const float* invstd, int C, const unsigned* amax_in,
                       const unsigned* raw_amax, unsigned* est, void* stream);
/* diagnostics of the f16x3 windows (debug / tests, never on the training path): counts[0] += values looked at, [1] += values
 * that saturate fp16 under the slot's scale (must stay 0), [2] += values whose low term is an fp16 subnormal (kept by the
 * matrix instruction: 11..22 significand bits, absolute error <= bound * 2^-39), [3] += values whose high term is subnormal.  The values are x (n floats, n % 4 == 0), or
 * max(x*scale[c]+shift[c], 0) with c = index % C when scale / shift are given, or - pairs != 0 - the stored terms of an f16
 * pair tensor. */
int spk_f16_window_count(const float* x, const float* scale, const float* shift, long long n, int C, const unsigned* slot,
                         int pairs, unsigned long long* counts /*[4], device*/, void* stream);
/* upper bound of |relu(raw * scale_c + shift_c)| from A = absmax(raw): max_c |scale_c| * A + max_c |shift_c| - the operand
 * scale input of a convolution / weight gradient that applies BatchNorm+ReLU while staging (SPK_IN_AFFINE_RELU) */
int spk_affine_estimate(const float* scale, const float* shift, int C, const unsigned* amax_in, unsigned* est, void* stream);

/* ---- statistics pooling (StatsPooling, scripts/model.py:435-457; mode 0 = 'mean', 1 = 'mean+std') -------- */
int spk_stats_pool_fwd(const float* x /*[B][H][W][C]*/, float* out /*[B][C*H*(1+mode)]*/, int B, int H, int W, int C,
                       int mode, void* stream);
int spk_stats_pool_bwd(const float* x, const float* gout, float* dx, int B, int H, int W, int C, int mode,
                       unsigned* amax_out /* optional: atomicMax of the float bits of |dx| */, void* stream);

/* ---- GEMM (fc1 = nn.Linear(5120,256) scripts/model.py:357; cosine F.linear :485; their gradients) ------ */
/* C[m][n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (+ bias[n]) (+ C[m][n]).  64x64 tiles on the fp32 matrix
 * instruction, K cut into spk_gemm_splitk(M,N,K) slices whose partial tiles go through `workspace`
 * (spk_gemm_workspace(M,N,K) bytes, caller-owned) and are folded in a fixed order. */
int spk_gemm_splitk(int M, int N, int K);
size_t spk_gemm_workspace(int M, int N, int K);
int spk_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long long sam,
                 long long sak, long long sbk, long long sbn, long long ldc, float alpha, int accumulate, float* workspace,
                 void* stream);
int spk_colsum(const float* dy, float* db, int M, int N, int accumulate, void* stream);

/* ---- heads ------------------------------------------------------------------------------------------- */
/* F.normalize rows (scripts/model.py:485) and its gradient */
int spk_l2norm_fwd(const float* x, float* y, float* inv_norm, int R, int D, float eps, void* stream);
int spk_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, float* dx, int R, int D, float eps,
                   int accumulate, void* stream);
/* AAM margin on the label column + scale (scripts/model.py:487-499) and its gradient */
int spk_aam_margin_fwd(const float* cosv, const long long* label, float* logits, int B, int S, float m, float s,
                       void* stream);
int spk_aam_margin_bwd(const float* cosv, const long long* label, const float* dlogits, float* dcos, int B, int S,
                       float m, float s, void* stream);
/* nn.CrossEntropyLoss rows (scripts/train_resnet.py:201,317): loss_row, dlogits = (softmax-onehot)*grad_scale,
 * rank[b] = #{j: logit_j > logit_label} (scripts/accuracy.py:4-16: correct@k <=> rank < k) */
int spk_softmax_ce(const float* logits, const long long* label, float* loss_row, float* dlogits, int* rank, int B, int S,
                   float grad_scale, void* stream);
int spk_mean(const float* v, float* out, int n, void* stream);
int spk_relu_bwd(const float* y, const float* dy, float* dx, long long n, void* stream);

/* ---- optimizer (torch.optim.SGD, scripts/train_resnet.py:203-205,328) ------------------------------------ */
int spk_sgd_step(float* p, const float* g, float* buf, long long n, float lr, float momentum, float weight_decay,
                 float grad_scale, int first_step, void* stream);

/* ---- cosine scoring back end (the step after the path) ----------------- */
/* out[n] = (emb[n] - mean) / max(||emb[n] - mean||, eps): mean subtraction of scripts/cosine_score.py:52-60 followed by
 * the normalisation inside F.cosine_similarity (:62, eps 1e-8) / F.normalize (scripts/compute_topk_mean_std.py:13,16) */
int spk_center_normalize(const float* emb /*[N][D]*/, const float* mean /*[D] or NULL*/, float* out, int N, int D, float eps,
                         void* stream);
/* out[t] = <en[ia[t]], te[ib[t]]> over normalised rows: the per-trial loop of scripts/cosine_score.py:57-65 */
int spk_trial_cosine(const float* en /*[n_en][D]*/, const float* te /*[n_te][D]*/, const int* ia, const int* ib, float* out,
                     int T, int D, int n_en, int n_te, void* stream);
/* per row of scores[N][M] (row pitch ld): mean and unbiased std of the k largest entries, M <= 16384
 * (scripts/compute_topk_mean_std.py:18-21: scores.topk(300), torch.std_mean) */
int spk_topk_mean_std(const float* scores, float* mean_out, float* std_out, int N, int M, int k, long long ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif
