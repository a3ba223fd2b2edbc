"""Thin Python wrappers over the libspkhip exports: shape checks, output allocation (torch caching
allocator owns every device buffer), tile selection, tap tables.  No arithmetic happens here."""
import ctypes
import os

import torch

from . import hip, tiling
from .hip import (CONV_M16, CONV_PIPE, CONV_WS, DY_PRESPLIT, IN_PRESPLIT, SIDE_PRESPLIT, WGRAD_GROUPS, EPI_ADD, EPI_AFFINE, EPI_BNBWD, EPI_RELU,
                  EPI_STATS, IN_AFFINE_RELU, IN_BNBWD, MASK_ACT, MASK_NONE, MASK_RAW,
                  call, ptr, stream)

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# When set to a list, every launch appends (label, algorithmic flops, start event, end event); the events are
# recorded on the launch stream (torch's current stream), which is what bench.py's roofline pass reads.
PROFILE = None
_raw_call = call


def call(name, *args, label=None, flops=0.0, nbytes=0.0):   # noqa: F811  (instrumented wrapper around hip.call)
    """nbytes: ALGORITHMIC bytes of the launch (every tensor it must read or write, once; no halo or re-read factors)"""
    if PROFILE is None:
        return _raw_call(name, *args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _raw_call(name, *args)
    e1.record()
    PROFILE.append((label or name, flops, e0, e1, nbytes))


def _iarr(v):
    return (ctypes.c_int * len(v))(*v)


class AmaxPool:
    """Slots for the absmax hand-offs of the f16x3 operand mode: one int32 (float bits, atomicMax'ed by the producing
    kernel) per gradient tensor of a step, taken in a fixed order so that a captured hipGraph sees the same addresses on
    every replay; zeroed once at the start of the step."""

    def __init__(self, device, n=4096):
        self.table = torch.zeros(n, device=device, dtype=torch.int32)
        self.next = 0

    def take_n(self, n):
        """n consecutive slots (a per-channel absmax row, spk_bn_bwd_reduce chan_amax)"""
        i = self.next
        assert i + n <= self.table.numel(), "AmaxPool exhausted"
        self.next = i + n
        return self.table[i:i + n]

    def reset(self):
        self.table.zero_()
        self.next = 0

    def take(self):
        i = self.next
        assert i < self.table.numel(), "AmaxPool exhausted"
        self.next = i + 1
        return self.table[i:i + 1]


def absmax_into(t, slot):
    """slot = max(slot, float bits of max|t|)  (spk_absmax: one pass over the tensor)"""
    call("spk_absmax", ptr(t), ptr(slot), t.numel(), stream())
    return slot


def affine_estimate(scale, shift, amax_in, est):
    """est = float bits of max_c|scale_c| * absmax + max_c|shift_c|: the bound of the values a convolution / weight gradient
    with the fused input BatchNorm+ReLU stages, given the slot with the absmax of the raw tensor"""
    call("spk_affine_estimate", ptr(scale), ptr(shift), scale.numel(), ptr(amax_in), ptr(est), stream())
    return est


def _amax_fwd_fallback(x, in_affine):
    """Operand-scale input of a forward convolution / weight-gradient X operand called without one (tests, tools)."""
    slot = absmax_into(x, torch.zeros(1, device=x.device, dtype=torch.int32))
    if in_affine is None:
        return slot
    return affine_estimate(in_affine[0], in_affine[1], slot, torch.zeros(1, device=x.device, dtype=torch.int32))


def bnbwd_estimate(coef, bn4, amax_in, raw_amax, est=None):
    """est = float bits of the rigorous bound of the BatchNorm-backward values k1 (dz - m1 - xhat m2) (spk_bnbwd_estimate):
    coef [3][C] rows of bn_bwd_coef, bn4 = [mean, invstd, ...] rows, amax_in / raw_amax = slots with absmax(incoming gradient)
    / absmax(raw tensor).  The value bn_bwd_coef(est_out=) produces without this launch."""
    if est is None:
        est = torch.zeros(1, device=coef.device, dtype=torch.int32)
    call("spk_bnbwd_estimate", ptr(coef), ptr(bn4[0]), ptr(bn4[1]), coef.shape[1], ptr(amax_in), ptr(raw_amax), ptr(est), stream())
    return est


def _amax_fallback(dy, in_bnbwd):
    """Operand-scale input of a data gradient called without one (tests, tools): absmax(dy), or for the fused
    BatchNorm-backward form the same bound bn_bwd_coef(est_out=) gives, from absmax(dy), absmax(raw) and the coefficient rows."""
    slot = torch.zeros(1, device=dy.device, dtype=torch.int32)
    absmax_into(dy, slot)
    if in_bnbwd is None:
        return slot
    raw_slot = absmax_into(in_bnbwd[0], torch.zeros(1, device=dy.device, dtype=torch.int32))
    return bnbwd_estimate(in_bnbwd[3], in_bnbwd[2], slot, raw_slot)


def f16_window_count(x, slot, counts, affine=None, pairs=False):
    """counts [4] int64 += (values, saturating, low term lost, high term subnormal) of x - or of relu(x*scale+shift) when
    affine = (scale, shift); pairs: x is an f16 pair tensor - under the operand scale of `slot` (spk_f16_window_count)."""
    C = x.shape[-1]
    call("spk_f16_window_count", ptr(x), ptr(affine[0]) if affine else None, ptr(affine[1]) if affine else None, x.numel(), C,
         ptr(slot), 1 if pairs else 0, ptr(counts), stream())
    return counts


def conv_out_hw(h, w, ksize, stride):
    pad = 1 if ksize == 3 else 0
    return (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1


# Matrix-core operand mode of the 3x3 convolutions (forward, data gradient, weight gradient).  Inputs, outputs, accumulation
# and the measured accuracy are fp32 in every mode (same test tolerances; tools/probe/split_probe.hip):
#   f32     fp32 operands on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak)
#   bf16x6  every fp32 operand split exactly into three bf16 terms, the 6 cross terms of weight >= 2^-16 on
#   bf16x9    v_mfma_f32_32x32x16_bf16 (bf16x9: all nine) - 16/6 of the fp32 matrix rate
#   f16x3   (default) operand value x 2^k as two fp16 terms (2 x 11 significand bits; k from the tensor's absmax, handed from
#           the kernel that writes a tensor to the kernels that read it), the three cross terms h1g1 + h1g2 + h2g1 on
#           v_mfma_f32_32x32x16_f16, accumulators scaled back by 2^-(ka+kb) - half the matrix instructions of bf16x6; the
#           step is power-limited on MI355X, so this is where the time goes (78 -> 59 ms per step)
# csrc/conv_kernel.h, csrc/conv_wgrad_split.hip, csrc/spk_common.h.  Packed weights carry the mode: set it before the first
# forward of a model (or mark its engine dirty).
MFMA_MODES = {"f32": 0, "bf16x6": 6, "bf16x9": 9, "f16x3": 3}
SPLIT = MFMA_MODES[os.environ.get("SPK_MFMA", "f16x3")]
# optional override for the backward kernels (data / weight gradients): None = same mode as the forward
SPLIT_BWD = MFMA_MODES[os.environ["SPK_MFMA_BWD"]] if os.environ.get("SPK_MFMA_BWD") else None


# wave-specialised (producer / consumer, persistent) convolution kernel, csrc/conv_ws_kernel.h.  SPK_CONV_WS = "0" never,
# "1" every eligible 3x3 launch, "auto" (default) only where it wins INSIDE the training step on MI355X (per-layer
# in-step times, profiles/r02_ws_instep.log): the data-gradient launches with the fused BN backward on >= 128 output
# channels in the f16x3 mode (0.576 -> 0.552 ms and 0.466 -> 0.413 ms per launch; those kernels keep the matrix pipe only
# ~25 % busy, so hiding the staging behind it pays).  Everywhere else conv_mfma_kernel is as fast or faster in-step
# (the 32/64-channel layers lose 10-20 % to the wave-specialised layout), and in the bf16x6 mode the step is power-limited
# and the two kernels tie.
WS_CONV = os.environ.get("SPK_CONV_WS", "auto")
assert WS_CONV in ("0", "1", "auto"), "SPK_CONV_WS must be 0, 1 or auto"
WS_AUTO_MIN_COUT = 128
# in-wave pipelined staging (conv_pipe_kernel): on by default for the f16x3 3x3 launches it covers; SPK_CONV_PIPE=0 disables
PIPE_CONV = os.environ.get("SPK_CONV_PIPE", "1") == "1"
# its v_mfma_f32_16x16x32_f16 form (two taps per K step; the chip holds a higher clock on that instruction shape)
PIPE_M16 = os.environ.get("SPK_PIPE_M16", "1") == "1"
# the same for the 3x3 weight gradients (conv_wgrad_pipe_kernel): opt-in.  Bit-identical, but no faster than
# conv_wgrad_split_kernel (+2..6 % on the 32/64-channel layers, -1..-12 % elsewhere, profiles/r02_wgrad_ablation.log): a tap of
# the weight gradient has three matrix instructions against ~60 VALU instructions of a staging item, so the VALU stream sets the
# pace with or without the overlap
PIPE_WGRAD = os.environ.get("SPK_WGRAD_PIPE", "0") == "1"
# ... and for the data gradients with the fused BatchNorm backward (sign-bit masks, register tiles <= 2 x 2): opt-in.  A fused
# item is ~90 VALU instructions (~450 cycles) against the 384 MFMA cycles of a 2 x 2 tap, so the staging is not hidden: per launch
# 0.75 vs 0.81 ms (64 channels) and 0.50 vs 0.54 ms (128 channels, against the wave-specialised kernel), nothing measurable per step
PIPE_BNBWD = os.environ.get("SPK_PIPE_BNBWD", "0") == "1"
GROUPED_1X1 = os.environ.get("SPK_WGRAD_1X1_GROUPS", "1") == "1"   # 1x1 weight gradients: input-channel groups as "taps"
WM16 = os.environ.get("SPK_WM16", "1") == "1"             # 3x3 grouped weight gradient with a pair-tensor dy: 16x16x32 form, dy by LDS DMA (csrc/conv_wgrad_wm16.hip)
C32M16 = os.environ.get("SPK_C32M16", "1") == "1"         # ... and its 32-channel-group layout for the first layer's weight gradients
WM_SHIFT = os.environ.get("SPK_WM_SHIFT", "1") == "1"      # 3x3 grouped weight gradient: shifted-window K loop where it applies (csrc/conv_wgrad_wm.hip, SH)
GROUPED_3X3 = os.environ.get("SPK_WGRAD_3X3_GROUPS", "1") == "1"   # 3x3 weight gradients: 2 x 2 (cin group x cout group) wave layout
GROUPED_1X1_BLOCKS = int(os.environ.get("SPK_WGRAD_1X1_BLOCKS", "512"))
PIPE_MIN_CIN = int(os.environ.get("SPK_PIPE_MIN_CIN", "64"))    # 32 channels = two chunks: nothing to pipeline, and the second tile costs occupancy
PIPE_MAX_LDS = int(os.environ.get("SPK_PIPE_MAX_LDS", str(80 * 1024)))      # two halo tiles; <= 80 KiB keeps two blocks per CU
# the same idea for the 3x3 weight gradients (f16x3 mode): eight-wave blocks, one per CU (conv_wgrad_ws_kernel).  Opt-in:
# bit-identical, but 10-15 % SLOWER than conv_wgrad_split_kernel on every layer (tools/wg_abl3.sh, profiles/r02_wgrad_ablation.log:
# producers alone 0.31 ms, consumers alone 0.28 ms, together 0.49 ms - the staging is VALU-bound and a SIMD does not run one
# wave's VALU stream under another wave's MFMAs to any useful degree; two resident four-wave blocks interleave better).
WS_WGRAD = os.environ.get("SPK_WGRAD_WS", "0") == "1"
WS_WGRAD_BLOCKS = int(os.environ.get("SPK_WGRAD_WS_BLOCKS", "256"))
WS_MIN_TAPS = int(os.environ.get("SPK_WS_MIN_TAPS", "9"))
LABEL_SHAPES = os.environ.get("SPK_LABEL_SHAPES", "0") == "1"      # diagnostic: per-layer lines in bench.py's all_kernels
WS_FORCE = None        # tests / sweeps: (TH, TW, MT, NT, WC) applied to every eligible launch


def _experimental():
    """the kernel forms behind SPK_CONV_WS / SPK_PIPE_BNBWD / SPK_WGRAD_PIPE / SPK_WGRAD_WS exist only in a library built with
    SPK_EXPERIMENTAL=1 (python pytorch-kaldi-resnet_amd/build.py --experimental)"""
    return hip.has_experimental()


SPLIT_1X1 = os.environ.get("SPK_SPLIT_1X1", "1") == "1"
# channel planes a single-tap (1x1) convolution stages per barrier: the first candidate that divides the channels and fits LDS
KC_CANDIDATES = tuple(int(v) for v in os.environ.get("SPK_KC", "4,2").split(","))


def split_for(ksize, bwd=False):
    """operand mode of a convolution launch.  The bf16-term modes cover the 3x3 convolutions only (1x1 convolutions stay on
    fp32 operands there); f16x3 also runs the 1x1 convolutions (Bottleneck blocks, downsample branches).  The stem (Cin = 1)
    is a direct convolution in every mode."""
    mode = SPLIT_BWD if (bwd and SPLIT_BWD is not None) else SPLIT
    if ksize == 3:
        return mode
    return 3 if (mode == 3 and SPLIT_1X1 and ksize == 1) else 0


def pack_conv_weight(w, transpose=False, out=None):
    """OIHW nn.Conv2d weight -> MFMA fragment order (see csrc/conv_mfma.hip, csrc/conv_split.hip)."""
    Cout, Cin, KH, KW = w.shape
    split = split_for(KH, bwd=transpose)     # the transposed pack feeds the data gradient
    n = packed_numel(w, bwd=transpose)
    if out is None or out.numel() != n:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    if split:
        call("spk_pack_conv_weight_split", ptr(w), ptr(out), Cout, Cin, KH, KW, 1 if transpose else 0, split, stream(),
             label="spk_pack_conv_weight")
    else:
        call("spk_pack_conv_weight", ptr(w), ptr(out), Cout, Cin, KH, KW, 1 if transpose else 0, stream(),
             label="spk_pack_conv_weight")
    return out


def packed_numel(w, bwd=False):
    """floats of the packed form of an OIHW weight in the current operand mode (three bf16 terms = 6 bytes per weight;
    fp32 = 4 bytes; two fp16 terms = 4 bytes + a 16-byte header with the tensor's absmax)."""
    sp = split_for(w.shape[2], bwd)
    return w.numel() * 3 // 2 if sp in (6, 9) else (w.numel() + 4 if sp == 3 else w.numel())


class PackTable:
    """Device-resident job table for spk_pack_conv_weights_batched: (weight, packed buffer, transpose) triples of every
    convolution, packed by ONE launch per step.  Rebuilt by the engine whenever a buffer or the operand mode changes."""

    def __init__(self, jobs, device):
        import struct
        assert hip.lib().spk_pack_job_bytes() == 48
        blob, block0 = b"", 0
        self.key = []
        for w, wpk, transpose in jobs:
            Cout, Cin, KH, KW = w.shape
            split = split_for(KH, bwd=transpose)
            assert wpk.numel() == packed_numel(w, bwd=transpose) and w.is_contiguous()
            blob += struct.pack("<QQ8i", w.data_ptr(), wpk.data_ptr(), Cout, Cin, KH * KW, 1 if transpose else 0, split,
                                w.numel(), block0, 0)
            block0 += (w.numel() + 255) // 256
            self.key.append((w.data_ptr(), wpk.data_ptr(), transpose, split))
        self.njobs, self.blocks = len(jobs), block0
        self.has_f16 = any(k[3] == 3 for k in self.key)
        self.table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
        self.launches = 0               # eager launches + launches recorded into a hipGraph
        self.launches_captured = 0

    def run(self):
        self.launches += 1
        if torch.cuda.is_current_stream_capturing():
            self.launches_captured += 1
        call("spk_pack_conv_weights_batched", ptr(self.table), self.njobs, self.blocks, 1 if self.has_f16 else 0, stream(),
             label="spk_pack_conv_weight")


STREAM_1X1 = os.environ.get("SPK_STREAM_1X1", "1") == "1"
STREAM_1X1_BLOCKS = int(os.environ.get("SPK_STREAM_1X1_BLOCKS", "512"))       # persistent blocks: two per CU


def _conv1x1_stream(x, wpk, out, in_affine, epi_add, add_mask, bn_bwd, want_stats, in_amax, out_amax, in_presplit):
    B, H, W, C = x.shape
    P = B * H * W
    tp = 64 * (1 if C >= 128 else (2 if C == 64 else 4))
    nblocks = max(1, min(STREAM_1X1_BLOCKS, -(-P // tp)))
    flags = 0
    if in_affine is not None:
        flags |= IN_AFFINE_RELU
    if in_presplit:
        assert in_affine is None
        flags |= IN_PRESPLIT
    if epi_add is not None:
        assert epi_add.shape == out.shape
        flags |= EPI_ADD
    bn_mask = bn_bwd[3] if (bn_bwd is not None and len(bn_bwd) > 3) else None
    for mk in (bn_mask, add_mask):
        if mk is not None:
            assert mk.dtype == torch.int32 and mk.numel() == out.numel() // 32, "sign mask shape"
    stats = None
    if bn_bwd is not None:
        want_stats = True
        flags |= EPI_BNBWD
        assert bn_bwd[0].shape == out.shape
    if want_stats:
        flags |= EPI_STATS
        stats = torch.empty(hip.lib().spk_conv1x1_stream_rows(nblocks, C), C, 2, device=x.device, dtype=torch.float32)
    call("spk_conv1x1_stream", ptr(x), ptr(wpk), ptr(out), ptr(in_affine[0]) if in_affine else None,
         ptr(in_affine[1]) if in_affine else None, ptr(epi_add), ptr(add_mask), ptr(bn_bwd[0]) if bn_bwd else None, ptr(bn_mask),
         ptr(bn_bwd[2]) if bn_bwd else None, ptr(stats), P, C, flags, ptr(in_amax), ptr(out_amax), nblocks, stream(),
         label="conv1x1_stream_kernel<%d>" % C + (" C%d %dx%d" % (C, H, W) if LABEL_SHAPES else ""),
         flops=2.0 * P * C * C,
         nbytes=4.0 * (P * C * (2 + (1 if epi_add is not None else 0) + (1 if bn_bwd is not None else 0))
                       + (P * C / 32) * ((bn_mask is not None) + (add_mask is not None)) + wpk.numel()))
    return stats


STREAM_C32 = os.environ.get("SPK_STREAM_C32", "1") == "1"
STREAM_C32_BLOCKS = int(os.environ.get("SPK_STREAM_C32_BLOCKS", "512"))


def _conv3x3_c32_stream(x, wpk, out, in_affine, in_amax, out_amax):
    B, H, W, _ = x.shape
    ntiles = B * (-(-H // 8)) * (-(-W // 16))
    nblocks = max(1, min(STREAM_C32_BLOCKS, ntiles))
    stats = torch.empty(4 * nblocks, 32, 2, device=x.device, dtype=torch.float32)
    flags = EPI_STATS | (IN_AFFINE_RELU if in_affine is not None else 0)
    call("spk_conv3x3_c32_stream", ptr(x), ptr(wpk), ptr(out), ptr(in_affine[0]) if in_affine else None,
         ptr(in_affine[1]) if in_affine else None, ptr(stats), B, H, W, flags, ptr(in_amax), ptr(out_amax), nblocks, stream(),
         label="conv3x3_c32_stream_kernel" + (" C32 %dx%d" % (H, W) if LABEL_SHAPES else ""),
         flops=2.0 * B * H * W * 32 * 32 * 9, nbytes=4.0 * (2 * B * H * W * 32 + wpk.numel()))
    return stats


def _conv_launch(x, wpk, out, Cout, taps, IS, OS, ooy, oox, OH, OW, in_affine, epi_affine, epi_add, relu, want_stats,
                 bn_bwd=None, in_bnbwd=None, side=None, split=0, add_mask=None, in_amax=None, out_amax=None, side_amax=None,
                 in_presplit=False, side_presplit=False):
    B, IH, IW, Cin = x.shape
    OHf, OWf = out.shape[1], out.shape[2]
    # Streaming 3x3 forward kernel of the 32-channel layer (csrc/conv3x3_c32_stream.hip): 32 -> 32 channels, stride 1, the forward tap
    # order, raw output + statistics, plain or fused-BatchNorm input.  (A data gradient has mirrored taps: general kernel.)
    if (STREAM_C32 and split == 3 and Cin == 32 and Cout == 32 and IS == 1 and OS == 1 and ooy == 0 and oox == 0
            and (IH, IW) == (OH, OW) == (OHf, OWf) and want_stats and in_bnbwd is None and side is None and epi_affine is None
            and epi_add is None and not relu and bn_bwd is None and not in_presplit and not side_presplit and add_mask is None
            and list(taps) == [(kh - 1, kw - 1, kh * 3 + kw) for kh in range(3) for kw in range(3)] and B * OH * OW * 32 < 2 ** 31 - 65536):
        return _conv3x3_c32_stream(x, wpk, out, in_affine, in_amax, out_amax)
    # Streaming 1x1 kernel (csrc/conv1x1_stream.hip): C -> C channels at stride 1 in the f16x3 mode, training-mode epilogues only
    # (raw output + statistics; data gradients: shortcut add with or without its sign mask, BatchNorm-backward statistics with the
    # mask as sign bits or recomputed from the raw tensor).  Everything else - eval-mode epilogues, strided 1x1, other widths,
    # the fused input BatchNorm backward, an activation tensor as the mask - stays on the general kernel.
    if (STREAM_1X1 and split == 3 and len(taps) == 1 and taps[0][0] == 0 and taps[0][1] == 0 and IS == 1 and OS == 1 and ooy == 0
            and oox == 0 and Cin == Cout and Cin in (32, 64, 128) and (IH, IW) == (OH, OW) == (OHf, OWf) and in_bnbwd is None
            and side is None and epi_affine is None and not relu and not side_presplit and out is not x
            and (bn_bwd is None or bn_bwd[1] is None or len(bn_bwd) > 3) and B * OH * OW * Cin < 2 ** 31):
        return _conv1x1_stream(x, wpk, out, in_affine, epi_add, add_mask, bn_bwd, want_stats, in_amax, out_amax, in_presplit)
    dys = [t[0] for t in taps]
    dxs = [t[1] for t in taps]
    tws = [t[2] for t in taps]
    ips = 1
    if len(taps) == 1 and IS > 1 and dys[0] == 0 and dxs[0] == 0:
        ips, IS = IS, 1          # strided 1x1: address the input through a strided view, stage only the pixels used
    key = (OH, OW, IS, max(dys) - min(dys) + 1, max(dxs) - min(dxs) + 1, len(taps), Cout)
    if tiling.AUTOTUNE and key not in (tiling.FORCE_CONV_SPLIT if split else tiling.FORCE_CONV) and PROFILE is None and not torch.cuda.is_current_stream_capturing():
        _autotune_conv(key, x, wpk, out, Cout, taps, IS, OS, ooy, oox, OH, OW, in_affine, epi_affine, epi_add, relu, split,
                       in_amax=in_amax, in_presplit=in_presplit)
    TH, TW, MT, NT = tiling.conv_tile(*key, mode=1 if in_bnbwd is not None else 0, split=split)
    # Producer / consumer (wave-specialised, persistent) kernel for the bf16-split 3x3 launches (csrc/conv_ws_kernel.h) with
    # its own wave layouts and tiles; everything else stays on conv_mfma_kernel.
    ws, WC = None, 1
    # fused BatchNorm backward with the mask as sign bits on <= 128 output channels: optionally the in-wave pipelined kernel
    pipe_fused = (PIPE_CONV and PIPE_BNBWD and split == 3 and in_bnbwd is not None and len(in_bnbwd) > 4 and MT * NT <= 4
                  and Cout <= 128 and WS_FORCE is None and WS_CONV != "1" and not side_presplit)
    ws_on = (WS_CONV == "1" or WS_FORCE is not None or (
        WS_CONV == "auto" and _experimental() and split == 3 and in_bnbwd is not None and Cout >= WS_AUTO_MIN_COUT
        and not pipe_fused))
    if side_presplit or in_presplit:
        ws_on = False            # f16 pair tensors: conv_mfma_kernel / conv_pipe_kernel only
    if split and ws_on and len(taps) >= WS_MIN_TAPS and ips == 1:
        ws = WS_FORCE or tiling.ws_tile(*key)
    if ws is not None:
        TH, TW, MT, NT, WC = ws
    # single-tap (1x1) convolutions stage several 32-channel planes per barrier: their K loop per plane is only 4 MFMA groups
    kc = 1
    if len(taps) == 1:
        halo = ((TH - 1) * IS + 1) * ((TW - 1) * IS + 1)
        for cand in KC_CANDIDATES:
            if Cin % (32 * cand) == 0 and cand * halo * tiling.LDS_PIX_BYTES <= tiling.LDS_HARD:
                kc = cand
                break
    flags = 0
    # in-wave pipelined kernel (csrc/conv_kernel.h, PIPE): f16x3 3x3 launches with a plain input whose two halo tiles fit
    halo9 = ((TH - 1) * IS + key[3]) * ((TW - 1) * IS + key[4])
    # (fused BatchNorm backward: the ReLU mask as sign bits, or recomputed from the raw tensor - not read from an activation)
    bnbwd_ok = in_bnbwd is None or pipe_fused
    pipe = (PIPE_CONV and ws is None and split == 3 and len(taps) == 9 and kc == 1 and bnbwd_ok and halo9 <= 576
            and (2 * halo9 + 1) * 80 <= PIPE_MAX_LDS and Cin >= PIPE_MIN_CIN)
    if pipe:
        flags |= CONV_PIPE
    # its 16x16x32 form (conv_kernel.h, M16): the (3, 2) register tile, at most eight staging items per plane
    m16 = pipe and PIPE_M16 and in_bnbwd is None and (MT, NT) == (3, 2) and halo9 <= 512 and IS == 1
    if m16:
        flags |= CONV_M16
    if ws is not None:
        flags |= CONV_WS | ({1: 0, 2: 1, 4: 2}[WC] << 8)
    if in_affine is not None:
        flags |= IN_AFFINE_RELU
    if in_presplit:
        assert split == 3 and in_affine is None and in_bnbwd is None, "f16 pair input: f16x3 mode, plain input"
        flags |= IN_PRESPLIT
    if side_presplit:
        assert split == 3 and in_bnbwd is not None, "f16 pair side output: fused BatchNorm-backward data gradient in the f16x3 mode"
        flags |= SIDE_PRESPLIT
    if epi_affine is not None:
        flags |= EPI_AFFINE
    if epi_add is not None:
        flags |= EPI_ADD
        assert epi_add.shape == out.shape
    if relu:
        flags |= EPI_RELU
    # sign masks (uint32 words, 32 channels each, written by bn_apply(mask=True)) replace the activated tensors where given
    in_mask = in_bnbwd[4] if (in_bnbwd is not None and len(in_bnbwd) > 4) else None
    bn_mask = bn_bwd[3] if (bn_bwd is not None and len(bn_bwd) > 3) else None
    for mk, ref in ((in_mask, x), (bn_mask, out), (add_mask, out)):
        if mk is not None:
            assert mk.dtype == torch.int32 and mk.numel() == ref.numel() // 32, "sign mask shape"
    if in_bnbwd is not None:
        flags |= IN_BNBWD
        assert in_affine is None and side is not None and side[0].shape == x.shape
        assert in_bnbwd[0].shape == x.shape and (in_bnbwd[1] is None or in_bnbwd[1].shape == x.shape)
    stats = None
    if bn_bwd is not None:
        want_stats = True
        flags |= EPI_BNBWD
        assert bn_bwd[0].shape == out.shape and (bn_bwd[1] is None or bn_bwd[1].shape == out.shape)
    if want_stats:
        flags |= EPI_STATS
        ntile = (4 // WC) * B * (-(-OH // TH)) * (-(-OW // TW))      # one partial row per wave (per pixel group of waves)
        stats = torch.empty(ntile, Cout, 2, device=x.device, dtype=torch.float32)
    call("spk_conv_mfma", ptr(x), ptr(wpk), ptr(out),
         ptr(in_affine[0]) if in_affine else None, ptr(in_affine[1]) if in_affine else None,
         ptr(epi_affine[0]) if epi_affine else None, ptr(epi_affine[1]) if epi_affine else None,
         ptr(epi_add),
         ptr(in_bnbwd[0]) if in_bnbwd else None, ptr(in_bnbwd[1]) if (in_bnbwd and in_mask is None) else None,
         ptr(in_bnbwd[2]) if in_bnbwd else None, ptr(in_bnbwd[3]) if in_bnbwd else None,
         ptr(in_mask), ptr(bn_mask), ptr(add_mask),
         ptr(side[0]) if side else None, ptr(side[1]) if side else None,
         ptr(bn_bwd[0]) if bn_bwd else None, ptr(bn_bwd[1]) if (bn_bwd and bn_mask is None) else None,
         ptr(bn_bwd[2]) if bn_bwd else None, ptr(stats), B, IH, IW, Cin, OH, OW, OHf, OWf, Cout, IS, OS, ooy, oox, len(taps),
         _iarr(dys), _iarr(dxs), _iarr(tws), TH, TW, MT, NT, kc, ips, flags, split, ptr(in_amax), ptr(out_amax), ptr(side_amax),
         stream(),
         label=(("conv_ws_kernel<%d,%d,%d,%s,%d>" % (MT, NT, WC, "true" if in_bnbwd is not None else "false", split)) if ws is not None
                else ("conv_pipe_kernel<%d,%d,false,false%s>" % (MT, NT, (",true,true" if in_presplit else ",false,true") if m16 else (",true" if in_presplit else "")) if in_bnbwd is None
                      else "conv_pipe_kernel<%d,%d,true,true>" % (MT, NT)) if pipe
                else "conv_mfma_kernel<%d,%d,%s,%d>" % (MT, NT, "true" if in_bnbwd is not None else "false", split)) + (
             " C%d %dx%d" % (Cout, OH, OW) if LABEL_SHAPES else ""),
         flops=2.0 * B * OH * OW * Cout * Cin * len(taps),
         # algorithmic bytes: the input pixels this launch reads (all of them for a stride-1 / full-tap launch), the output it
         # writes, and every fused side stream once: shortcut add, raw + side draw of the fused BatchNorm backward, raw of the
         # BatchNorm whose backward statistics are reduced, the 1-bit masks; packed weights
         nbytes=4.0 * (B * (IH // ips) * (IW // ips) * Cin * (min(1.0, len(taps) * OH * OW / max(1, (IH // ips) * (IW // ips))) if IS == 1 and OS > 1 else 1.0)
                       + B * OH * OW * Cout * (1 + (1 if epi_add is not None else 0) + (1 if bn_bwd is not None else 0))
                       + (2 * x.numel() if in_bnbwd is not None else 0)
                       + (x.numel() / 32 if in_mask is not None else 0) + (B * OH * OW * Cout / 32) * ((bn_mask is not None) + (add_mask is not None))
                       + wpk.numel()))
    return stats


def _time_launch(fn, reps=2):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def _autotune_conv(key, x, wpk, out, Cout, taps, IS, OS, ooy, oox, OH, OW, in_affine, epi_affine, epi_add, relu, split=0,
                   in_amax=None, in_presplit=False):
    """Time candidate tiles on the real operands (scratch output) and pin the fastest for this launch shape.  The operand-scale
    slot is the caller's (or computed ONCE here): no absmax pass or allocation inside a timed candidate."""
    scratch = torch.empty_like(out)
    if split == 3 and in_amax is None:
        in_amax = _amax_fwd_fallback(x, in_affine)
    add = epi_add if (epi_add is None or epi_add.data_ptr() != out.data_ptr()) else scratch
    best = None
    table = tiling.FORCE_CONV_SPLIT if split else tiling.FORCE_CONV
    for cand in tiling.conv_candidates(*key, split=split):
        table[key] = cand
        try:
            ms = _time_launch(lambda: _conv_launch(x, wpk, scratch, Cout, taps, IS, OS, ooy, oox, OH, OW, in_affine,
                                                   epi_affine, add, relu, True, split=split, in_amax=in_amax,
                                                   in_presplit=in_presplit))
        except RuntimeError:
            continue
        if best is None or ms < best[0]:
            best = (ms, cand)
    if best is None:
        del table[key]
    else:
        table[key] = best[1]


def _autotune_wgrad(key, x, dy, ksize, stride, in_affine, dy_amax=None, x_amax=None, dy_presplit=False):
    """(the operand-scale slots come from the caller: computed once, outside the timed candidates)"""
    scratch = torch.empty(dy.shape[3], x.shape[3], ksize, ksize, device=x.device, dtype=torch.float32)
    best = None
    for cand in tiling.wgrad_candidates(*key):
        tiling.FORCE_WGRAD[key] = cand
        try:
            ms = _time_launch(lambda: conv_wgrad(x, dy, scratch, ksize, stride, in_affine=in_affine, dy_amax=dy_amax,
                                                 x_amax=x_amax, dy_presplit=dy_presplit))
        except RuntimeError:
            continue
        if best is None or ms < best[0]:
            best = (ms, cand)
    if best is None:
        del tiling.FORCE_WGRAD[key]
    else:
        tiling.FORCE_WGRAD[key] = best[1]


def conv_fwd(x, wpk, Cout, ksize, stride, in_affine=None, epi_affine=None, epi_add=None, relu=False, stats=False,
             out=None, in_amax=None, out_amax=None):
    """Forward conv on NHWC x with packed weights. Returns (out, stats_partial or None).
    f16x3 operand mode: in_amax = slot with the float bits of the absmax of the STAGED values (x itself, or the
    affine_estimate of relu(x*scale+shift) when in_affine is given); computed here with extra passes when omitted (tests)."""
    B, IH, IW, Cin = x.shape
    if split_for(ksize) == 3 and in_amax is None:
        in_amax = _amax_fwd_fallback(x, in_affine)
    OH, OW = conv_out_hw(IH, IW, ksize, stride)
    if out is None:
        out = torch.empty(B, OH, OW, Cout, device=x.device, dtype=torch.float32)
    if ksize == 3:
        taps = [(kh - 1, kw - 1, kh * 3 + kw) for kh in range(3) for kw in range(3)]
    else:
        taps = [(0, 0, 0)]
    st = _conv_launch(x, wpk, out, Cout, taps, stride, 1, 0, 0, OH, OW, in_affine, epi_affine, epi_add, relu, stats,
                      split=split_for(ksize), in_amax=in_amax, out_amax=out_amax)
    return out, st


def conv_dgrad(dy, wpk_t, Cin, ksize, stride, in_hw, add=None, out=None, accumulate=False, bn_bwd=None, in_bnbwd=None,
               side=None, add_mask=None, in_amax=None, out_amax=None, side_amax=None, in_presplit=False, side_presplit=False):
    """Data gradient of conv_fwd: dy [B][OH][OW][Cout] -> dx [B][IH][IW][Cin].
    `add` (same shape as dx) is summed in the epilogue (only where the bits of `add_mask`, sign-mask words of the same
    shape, are set when that is given); accumulate=True adds onto the existing `out`.
    bn_bwd = (raw, act or None, bn4[4][Cin]) (stride-1 only): dx is the gradient wrt the output of that BatchNorm
    (+ReLU); the launch also returns the BatchNorm-backward partial sums -> (dx, partial).
    in_bnbwd = (raw, act or None, bn4[4][Cout], coef[3][Cout]) + side = (draw_out, dz_out or None) (stride-1 only): `dy` is
    the gradient wrt a BatchNorm(+ReLU) OUTPUT; the BatchNorm backward is applied while staging and the gradient wrt
    the raw conv output is also written to draw_out (and dy*mask to dz_out).
    f16x3 operand mode: in_amax = 1-element int32 tensor with the float bits of absmax(staged tensor) - for in_bnbwd an upper
    estimate of the BatchNorm-backward values (bn_bwd_coef(..., est_out=)) - that fixes the power-of-two operand scale; when
    omitted it is computed here with an extra pass (tests / tools; the engine always passes it).  out_amax / side_amax: slots
    the launch atomically maxes |dx| / |draw_out| into (float bits) for the kernels that consume those tensors.
    in_presplit: `dy` is an f16 pair tensor (bn_backward(pair_scale=)) scaled by the sigma of in_amax, staged by plain copy.
    side_presplit (with in_bnbwd): draw_out leaves as an f16 pair tensor scaled by the sigma of in_amax (the rigorous bound)."""
    B, OH, OW, Cout = dy.shape
    assert not in_presplit or in_amax is not None, "an f16 pair tensor comes with its scale slot"
    if split_for(ksize, True) == 3 and in_amax is None:
        in_amax = _amax_fallback(dy, in_bnbwd)
    IH, IW = in_hw
    if out is None:
        assert not accumulate
        out = torch.empty(B, IH, IW, Cin, device=dy.device, dtype=torch.float32)
    if accumulate:
        assert add is None
        add = out
    if stride == 1:
        if ksize == 3:
            taps = [(1 - kh, 1 - kw, kh * 3 + kw) for kh in range(3) for kw in range(3)]
        else:
            taps = [(0, 0, 0)]
        st = _conv_launch(dy, wpk_t, out, Cin, taps, 1, 1, 0, 0, IH, IW, None, None, add, False, False, bn_bwd, in_bnbwd,
                          side, split=split_for(ksize, True), add_mask=add_mask, in_amax=in_amax, out_amax=out_amax,
                          side_amax=side_amax, in_presplit=in_presplit, side_presplit=side_presplit)
        return (out, st) if bn_bwd is not None else out
    assert stride == 2 and bn_bwd is None and in_bnbwd is None and add_mask is None
    if ksize == 1:
        # only even input pixels receive gradient from a strided 1x1 conv
        if not accumulate:
            if add is not None:
                out.copy_(add)
            else:
                out.zero_()
        _conv_launch(dy, wpk_t, out, Cin, [(0, 0, 0)], 1, 2, 0, 0, (IH + 1) // 2, (IW + 1) // 2, None, None, out, False,
                     False, split=split_for(1, True), in_amax=in_amax, out_amax=out_amax, in_presplit=in_presplit)
        return out
    for cy in range(2):
        for cx in range(2):
            LH, LW = (IH - cy + 1) // 2, (IW - cx + 1) // 2
            if LH <= 0 or LW <= 0:
                continue
            taps = []
            for kh in range(3):
                if (cy + 1 - kh) % 2:
                    continue
                for kw in range(3):
                    if (cx + 1 - kw) % 2:
                        continue
                    taps.append(((cy + 1 - kh) // 2, (cx + 1 - kw) // 2, kh * 3 + kw))
            _conv_launch(dy, wpk_t, out, Cin, taps, 1, 2, cy, cx, LH, LW, None, None, add, False, False, split=split_for(3, True),
                         in_amax=in_amax, out_amax=out_amax, in_presplit=in_presplit)
    return out


_ws_cache = {}
_ws_retired = []


def _workspace(nbytes, device):
    """Scratch buffer per (device, launch stream): kernels on different streams never share one."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    w = _ws_cache.get(key)
    if w is None or w.numel() * 4 < nbytes:
        if w is not None:
            _ws_retired.append(w)       # a captured hipGraph may still hold its address: never hand it back
        w = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        _ws_cache[key] = w
    return w


def _workspace_named(name, nbytes, device):
    """like _workspace, with its own buffer per purpose (a GEMM may run between a weight gradient and its slab reduce)"""
    key = (name, str(device), torch.cuda.current_stream().cuda_stream)
    w = _ws_cache.get(key)
    if w is None or w.numel() * 4 < nbytes:
        if w is not None:
            _ws_retired.append(w)
        w = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        _ws_cache[key] = w
    return w


def conv_wgrad(x, dy, dw, ksize, stride, in_affine=None, accumulate=False, dy_amax=None, x_amax=None, dy_presplit=False):
    """dw (OIHW view, contiguous) <- weight gradient of conv(x) given dy.  f16x3 operand mode: dy_amax = slot with the float
    bits of absmax(dy) (computed here with an extra pass when omitted: tests / tools).  dy_presplit: dy is an f16 pair tensor
    scaled by the sigma of dy_amax (its scale slot), staged by plain copy."""
    B, IH, IW, Cin = x.shape
    _, OH, OW, Cout = dy.shape
    split = split_for(ksize, True)
    assert not dy_presplit or (split == 3 and dy_amax is not None), "an f16 pair tensor comes with its scale slot (f16x3 mode)"
    if split == 3 and dy_amax is None:
        dy_amax = absmax_into(dy, torch.zeros(1, device=dy.device, dtype=torch.int32))
    if split == 3 and x_amax is None:
        x_amax = _amax_fwd_fallback(x, in_affine)
    wkey = (OH, OW, Cin, Cout, ksize, stride)
    if tiling.AUTOTUNE and wkey not in tiling.FORCE_WGRAD and PROFILE is None and not torch.cuda.is_current_stream_capturing():
        tiling.FORCE_WGRAD[wkey] = tiling._wgrad_tile(*wkey)      # placeholder: stops the recursion below
        _autotune_wgrad(wkey, x, dy, ksize, stride, in_affine, dy_amax, x_amax, dy_presplit)
    TH, TW, WN = tiling.wgrad_tile(OH, OW, Cin, Cout, ksize, stride, split=split)
    if C32M16 and split == 3 and ksize == 3 and WN == 1 and dy_presplit:
        TH, TW, WN = tiling.c32m16_tile(OH, OW, Cin, Cout, ksize, stride) or (TH, TW, WN)      # the 32-channel-group 16x16x32 kernel has its own tile rule
    nreg = B * (-(-OH // TH)) * (-(-OW // TW))
    # producer / consumer kernel (csrc/conv_wgrad_split.hip: conv_wgrad_ws_kernel): f16x3 3x3 launches whose two LDS slots fit
    halo = ((TH - 1) * stride + ksize) * ((TW - 1) * stride + ksize)
    wgws = (WS_WGRAD and split == 3 and ksize == 3 and not dy_presplit
            and 2 * (halo * 192 + -(-(TH * TW) // 16) * 16 * (WN * 192 + (64 if WN > 1 else 0))) <= 160 * 1024)
    # in-wave pipelined kernel (csrc/conv_wgrad_pipe.hip): two planar LDS slots of 64-byte rows
    nst = -(-(TH * TW) // 16)                                   # k-steps of the tile: one or two per wave group
    wgp = (PIPE_WGRAD and not wgws and not dy_presplit and split == 3 and ksize == 3 and nst % (4 // WN) == 0 and nst // (4 // WN) in (1, 2)
           and 2 * (2 * halo * 64 + 2 * WN * -(-(TH * TW) // 16) * 16 * 64) + 128 <= PIPE_MAX_LDS)
    # 1x1, f16x3: conv_wgrad_1x1_kernel with 2 or 4 input-channel groups per block (csrc/conv_wgrad_1x1.hip)
    cg = 0
    if GROUPED_1X1 and split == 3 and ksize == 1 and WN in (2, 4) and TH * TW <= 64:
        cg = 4 if Cin % 128 == 0 else (2 if Cin % 64 == 0 else 0)
        if cg and -(-(TH * TW) // 16) * 16 * (cg * 192 + WN * 192 + 64) > 80 * 1024:
            cg = 2 if cg == 4 else 0
    # 3x3, f16x3: 2 input-channel groups x 2 output-channel groups of waves (csrc/conv_wgrad_wm.hip)
    wm = (GROUPED_3X3 and not wgws and not wgp and split == 3 and ksize == 3 and WN == 2 and Cin % 64 == 0 and Cout % 64 == 0
          and halo <= 112 and TH * TW <= 64 and halo * 384 + -(-(TH * TW) // 16) * 16 * 448 <= 80 * 1024)
    wm16 = wm and WM16 and dy_presplit and halo * 416 + 2 * -(-(TH * TW) // 32) * 32 * 256 <= 80 * 1024     # X pixel pitch 416 B (csrc/conv_wgrad_wm16.hip)
    # ... and the same kernel in its 32-channel-group layout (the first layer): four waves split the k-steps and fold at the end of the block
    c32m16 = (C32M16 and not wm and not wgws and not wgp and split == 3 and ksize == 3 and WN == 1 and dy_presplit and halo <= 192
              and halo * 192 + 2 * -(-(TH * TW) // 32) * 32 * 128 <= 80 * 1024)
    if wm:
        cg = 2
    nsplit = min(nreg, tiling.wgrad_nsplit(nreg, Cin, Cout, WN, WS_WGRAD_BLOCKS if wgws else (GROUPED_1X1_BLOCKS if cg else None),
                                           cin_groups=cg or 1))
    nslab = nsplit
    nbytes = hip.lib().spk_conv_wgrad_workspace(nslab, ksize, Cin, Cout)
    ws = _workspace(nbytes, x.device)
    flags = (IN_AFFINE_RELU if in_affine is not None else 0) | (CONV_WS if wgws else 0) | (CONV_PIPE if wgp else 0)
    if cg:
        flags |= WGRAD_GROUPS | ({2: 1, 4: 2}[cg] << 12)
    if dy_presplit:
        flags |= DY_PRESPLIT
    if wm and not WM_SHIFT:
        flags |= hip.WGRAD_NOSHIFT
    if wm16 or c32m16:
        flags |= hip.WGRAD_M16
    call("spk_conv_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws),
         ptr(in_affine[0]) if in_affine else None, ptr(in_affine[1]) if in_affine else None,
         B, IH, IW, Cin, OH, OW, Cout, ksize, stride, TH, TW, WN, nsplit, flags, 1 if accumulate else 0, split,
         ptr(dy_amax) if split == 3 else None, ptr(x_amax) if split == 3 else None, stream(),
         label=("conv_wgrad_wm16_kernel" if wm16 else "conv_wgrad_c32m16_kernel" if c32m16 else "conv_wgrad_wm_kernel" if wm
         else ("conv_wgrad_1x1_kernel<%d,%d,%d>" % (4 // WN, WN, cg)) if cg
         else ("conv_wgrad_ws_kernel<%d,%d,%d,%d>" % (ksize * ksize, 4 // WN, WN, 4 if halo <= 128 else 5)) if wgws
         else ("conv_wgrad_pipe_kernel<%d,%d,%d,%d>" % (4 // WN, WN, 4 if halo <= 128 else 5, nst // (4 // WN))) if wgp
         else ("conv_wgrad_split_kernel<%d,%d,%d,%d,%d>" % (
             ksize * ksize, 4 // WN, WN, split, 4 if ((TH - 1) * stride + ksize) * ((TW - 1) * stride + ksize) <= 128 else 5)) if split
         else "conv_wgrad_kernel<%d,%d,%d>" % (ksize * ksize, 4 // WN, WN)) + (
             " C%d %dx%d" % (Cout, OH, OW) if LABEL_SHAPES else ""),
         flops=2.0 * B * OH * OW * Cout * Cin * ksize * ksize, nbytes=4.0 * (x.numel() + dy.numel() + nbytes / 4))
    call("spk_wgrad_reduce", ptr(ws), ptr(dw), nslab, ksize, Cin, Cout, 1 if accumulate else 0, stream())
    return dw


def stem_fwd(x, w, epi_affine=None, relu=False, stats=False, amax_out=None):
    """x [B][F][T] -> [B][F][T][32] (+ stats partial [nblk][32][2])."""
    B, F, T = x.shape
    out = torch.empty(B, F, T, 32, device=x.device, dtype=torch.float32)
    flags = (EPI_AFFINE if epi_affine is not None else 0) | (EPI_RELU if relu else 0) | (EPI_STATS if stats else 0)
    st = None
    if stats:
        st = torch.empty(hip.lib().spk_stem_fwd_blocks(B, F, T), 32, 2, device=x.device, dtype=torch.float32)
    call("spk_stem_conv_fwd", ptr(x), ptr(w), ptr(out), ptr(st),
         ptr(epi_affine[0]) if epi_affine else None, ptr(epi_affine[1]) if epi_affine else None, B, F, T, flags, ptr(amax_out),
         stream())
    return out, st


def stem_wgrad(x, draw, dw, accumulate=False):
    B, F, T = x.shape
    nblk = hip.lib().spk_stem_wgrad_blocks(B, F, T)
    ws = _workspace(nblk * 288 * 4, x.device)
    call("spk_stem_conv_wgrad", ptr(x), ptr(draw), ptr(dw), ptr(ws), B, F, T, 1 if accumulate else 0, stream())
    return dw


def bn_stats_partial(x2d):
    N, C = x2d.shape
    nblk = hip.lib().spk_bn_stats_blocks(N, C)
    part = torch.empty(nblk, C, 2, device=x2d.device, dtype=torch.float32)
    call("spk_bn_stats_partial", ptr(x2d), ptr(part), N, C, stream())
    return part


_ws64_cache = {}


def _ws64(device):
    """fp64 scratch of the two-stage BN finalize (spk_bn_finalize_workspace: 256 rows x C x 2 doubles; sized for C = 2048)."""
    w = _ws64_cache.get(str(device))
    if w is None:
        w = torch.zeros(256 * 2048 * 2 + 64, device=device, dtype=torch.float64)
        _ws64_cache[str(device)] = w
    return w


def bn_finalize(partial, count, gamma, beta, running_mean, running_var, nbt, out4, amax_in=None, est_out=None):
    """out4: tensor [4][C] = (mean, invstd, scale, shift).  amax_in + est_out (f16x3 mode): also the upper bound of
    |relu(raw*scale+shift)| from the slot with absmax(raw) (the affine_estimate value, without its launch)."""
    C = gamma.numel()
    call("spk_bn_finalize", ptr(partial), partial.shape[0], C, float(count), ptr(gamma), ptr(beta), ptr(running_mean),
         ptr(running_var), ptr(nbt), ptr(out4[0]), ptr(out4[1]), ptr(out4[2]), ptr(out4[3]), BN_MOMENTUM, BN_EPS,
         ptr(_ws64(partial.device)), ptr(amax_in), ptr(est_out), stream())


def bn_eval_coeffs(gamma, beta, rm, rv, out2):
    C = gamma.numel()
    call("spk_bn_eval_coeffs", ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(out2[0]), ptr(out2[1]), C, BN_EPS, stream())


def bn_apply(raw, scale, shift, res=None, res_affine=None, relu=True, out=None, mask=False, amax_out=None):
    """out = [relu](raw*scale + shift [+ res | + res*rscale + rshift]).  mask=True also returns the sign bits of `out`
    ([N][C/32] int32 words): the backward pass reads those instead of the activated tensor."""
    C = raw.shape[-1]
    N = raw.numel() // C
    if out is None:
        out = torch.empty_like(raw)
    mk = torch.empty(N * (C // 32), device=raw.device, dtype=torch.int32) if mask else None
    call("spk_bn_apply", ptr(raw), ptr(scale), ptr(shift), ptr(res),
         ptr(res_affine[0]) if res_affine else None, ptr(res_affine[1]) if res_affine else None, ptr(out), ptr(mk), N, C,
         1 if relu else 0, ptr(amax_out), stream(),
         nbytes=4.0 * raw.numel() * (2 + (1 if res is not None else 0)) + (raw.numel() / 8 if mask else 0))
    return (out, mk) if mask else out


def _bn_bwd_bytes(raw, mask_mode, passes):
    """algorithmic bytes of a BatchNorm-backward stream kernel: `passes` fp32 tensors (dy, raw, draw / dz) + the mask source"""
    n = raw.numel()
    return 4.0 * n * passes + {MASK_NONE: 0.0, MASK_RAW: 0.0, MASK_ACT: 4.0 * n}.get(mask_mode, n / 8.0)


def bn_backward(dy, raw, act, bn4, gamma, dgamma, dbeta, mask_mode, draw_out=None, dz_out=None, accumulate=False,
                partial=None, amax_out=None, pair=None, chan_amax=None):
    """Full BN backward (reduce -> finalize -> apply). bn4 = [mean, invstd, scale, shift] rows.
    `partial`: (sum dz, sum dz*xhat) rows already produced by the data-gradient epilogue (EPI_BNBWD) - skips the
    reduction pass.  Returns draw (gradient wrt the raw conv output).
    pair = (amax_in, raw_amax, est) (f16x3 mode): draw is written as an f16 PAIR tensor (include/spkhip.h) scaled by the sigma
    of `est`, the slot the finalize fills with the rigorous bound of |draw| from amax_in (absmax of dy) and raw_amax (absmax of
    raw): the data gradient and the weight gradient that consume draw stage it by plain copy with that slot as their scale.
    chan_amax ([C] int32 zeros, with pair): the reduction pass also takes the absmax of dz per CHANNEL and the bound uses it
    instead of the tensor-wide amax_in (needs partial=None: the reduction runs here)."""
    C = raw.shape[-1]
    N = raw.numel() // C
    coef = torch.empty(3, C, device=raw.device, dtype=torch.float32)
    if partial is None:
        nblk = hip.lib().spk_bn_stats_blocks(N, C)
        part = torch.empty(nblk, C, 2, device=raw.device, dtype=torch.float32)
        call("spk_bn_bwd_reduce", ptr(dy), ptr(raw), ptr(act), ptr(bn4[0]), ptr(bn4[1]), ptr(bn4[2]), ptr(bn4[3]), ptr(part),
             N, C, mask_mode, ptr(chan_amax), stream(), nbytes=_bn_bwd_bytes(raw, mask_mode, 2))
    else:
        assert chan_amax is None, "chan_amax comes out of the reduction pass: partial must be None"
        part, nblk = partial, partial.shape[0]
    call("spk_bn_bwd_finalize", ptr(part), nblk, C, float(N), ptr(gamma), ptr(bn4[1]), ptr(dgamma), ptr(dbeta), ptr(coef),
         1 if accumulate else 0, ptr(_ws64(raw.device)), ptr(pair[0]) if pair else None, ptr(pair[1]) if pair else None,
         ptr(bn4[0]) if pair else None, ptr(pair[2]) if pair else None, ptr(chan_amax) if pair else None, stream())
    if draw_out is None:
        draw_out = torch.empty_like(raw)
    call("spk_bn_bwd_apply", ptr(dy), ptr(raw), ptr(act), ptr(bn4[0]), ptr(bn4[1]), ptr(bn4[2]), ptr(bn4[3]), ptr(coef),
         ptr(draw_out), ptr(dz_out), N, C, mask_mode, ptr(amax_out), ptr(pair[2]) if pair else None, stream(),
         nbytes=_bn_bwd_bytes(raw, mask_mode, 3 + (1 if dz_out is not None else 0)))
    return draw_out


def bn_bwd_coef(partial, count, gamma, bn4, dgamma, dbeta, accumulate=False, amax_in=None, raw_amax=None, est_out=None,
                chan_amax=None):
    """BatchNorm-backward finalize only: dgamma, dbeta and the coefficient rows [gamma*invstd, mean(dz), mean(dz*xhat)].
    amax_in + raw_amax + est_out (f16x3 mode): also the RIGOROUS upper bound of the values the fused BatchNorm-backward data
    gradient will stage (spk_bnbwd_estimate), from the absmax of the incoming gradient and of the raw tensor."""
    C = gamma.numel()
    coef = torch.empty(3, C, device=gamma.device, dtype=torch.float32)
    assert est_out is None or (amax_in is not None and raw_amax is not None)
    call("spk_bn_bwd_finalize", ptr(partial), partial.shape[0], C, float(count), ptr(gamma), ptr(bn4[1]), ptr(dgamma),
         ptr(dbeta), ptr(coef), 1 if accumulate else 0, ptr(_ws64(gamma.device)), ptr(amax_in) if est_out is not None else None,
         ptr(raw_amax) if est_out is not None else None, ptr(bn4[0]) if est_out is not None else None, ptr(est_out),
         ptr(chan_amax) if est_out is not None else None, stream())
    return coef


def bn_bwd_partial(dy, raw, act, bn4, mask_mode, chan_amax=None):
    """Stand-alone reduction (sum dz, sum dz*xhat) when no data-gradient epilogue produced it.  chan_amax ([C] int32 zeros):
    also the absmax of dz per channel (float bits)."""
    C = raw.shape[-1]
    N = raw.numel() // C
    nblk = hip.lib().spk_bn_stats_blocks(N, C)
    part = torch.empty(nblk, C, 2, device=raw.device, dtype=torch.float32)
    call("spk_bn_bwd_reduce", ptr(dy), ptr(raw), ptr(act), ptr(bn4[0]), ptr(bn4[1]), ptr(bn4[2]), ptr(bn4[3]), ptr(part),
         N, C, mask_mode, ptr(chan_amax), stream(), nbytes=_bn_bwd_bytes(raw, mask_mode, 2))
    return part


def stats_pool_fwd(x, mode):
    B, H, W, C = x.shape
    out = torch.empty(B, C * H * (2 if mode else 1), device=x.device, dtype=torch.float32)
    call("spk_stats_pool_fwd", ptr(x), ptr(out), B, H, W, C, mode, stream())
    return out


def stats_pool_bwd(x, gout, mode, amax_out=None):
    """amax_out: slot the launch atomically maxes |dx| into (float bits): the operand scale of dx's f16x3 consumers"""
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    call("spk_stats_pool_bwd", ptr(x), ptr(gout), ptr(dx), B, H, W, C, mode, ptr(amax_out), stream())
    return dx


def gemm(A, Bm, M, N, K, sam, sak, sbk, sbn, bias=None, out=None, alpha=1.0, accumulate=False):
    """out[M][N] = alpha * sum_k A[m*sam+k*sak] * B[k*sbk+n*sbn] (+bias[n]) (+out)."""
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    ws = _workspace_named("gemm", hip.lib().spk_gemm_workspace(M, N, K), A.device)
    call("spk_gemm_f32", ptr(A), ptr(Bm), ptr(out), ptr(bias), M, N, K, sam, sak, sbk, sbn, out.stride(0), float(alpha),
         1 if accumulate else 0, ptr(ws), stream(), label="gemm_f32_kernel", flops=2.0 * M * N * K)
    return out


def linear_fwd(x, w, bias=None):
    """x [M][K] @ w[N][K]^T + bias."""
    M, K = x.shape
    N = w.shape[0]
    return gemm(x, w, M, N, K, K, 1, 1, K, bias=bias)


def linear_bwd(x, w, dy, dw, db=None, need_dx=True, accumulate=False):
    """Gradients of linear_fwd: dx [M][K] = dy @ w; dw [N][K] = dy^T @ x; db = colsum(dy)."""
    M, K = x.shape
    N = w.shape[0]
    dx = None
    if need_dx:
        dx = gemm(dy, w, M, K, N, N, 1, K, 1)
    gemm(dy, x, N, K, M, 1, N, K, 1, out=dw, accumulate=accumulate)
    if db is not None:
        call("spk_colsum", ptr(dy), ptr(db), M, N, 1 if accumulate else 0, stream())
    return dx


def l2norm_fwd(x):
    R, D = x.shape
    y = torch.empty_like(x)
    inv = torch.empty(R, device=x.device, dtype=torch.float32)
    call("spk_l2norm_fwd", ptr(x), ptr(y), ptr(inv), R, D, 1e-12, stream())
    return y, inv


def l2norm_bwd(y, inv, dy, out=None, accumulate=False):
    R, D = y.shape
    if out is None:
        out = torch.empty_like(y)
    call("spk_l2norm_bwd", ptr(y), ptr(inv), ptr(dy), ptr(out), R, D, 1e-12, 1 if accumulate else 0, stream())
    return out


def aam_margin_fwd(cosv, label, m, s):
    B, S = cosv.shape
    logits = torch.empty_like(cosv)
    call("spk_aam_margin_fwd", ptr(cosv), ptr(label), ptr(logits), B, S, float(m), float(s), stream())
    return logits


def aam_margin_bwd(cosv, label, dlogits, m, s):
    B, S = cosv.shape
    dcos = torch.empty_like(cosv)
    call("spk_aam_margin_bwd", ptr(cosv), ptr(label), ptr(dlogits), ptr(dcos), B, S, float(m), float(s), stream())
    return dcos


def softmax_ce(logits, label, grad_scale=None, want_rank=True):
    """-> (loss_row [B], dlogits or None, rank [B] int32 or None)."""
    B, S = logits.shape
    loss_row = torch.empty(B, device=logits.device, dtype=torch.float32)
    dl = torch.empty_like(logits) if grad_scale is not None else None
    rank = torch.empty(B, device=logits.device, dtype=torch.int32) if want_rank else None
    call("spk_softmax_ce", ptr(logits), ptr(label), ptr(loss_row), ptr(dl), ptr(rank), B, S,
         float(grad_scale if grad_scale is not None else 0.0), stream())
    return loss_row, dl, rank


def mean(v):
    out = torch.empty(1, device=v.device, dtype=torch.float32)
    call("spk_mean", ptr(v), ptr(out), v.numel(), stream())
    return out


def relu_bwd(y, dy):
    dx = torch.empty_like(dy)
    call("spk_relu_bwd", ptr(y), ptr(dy), ptr(dx), y.numel(), stream())
    return dx


def sgd_step(p, g, buf, lr, momentum, weight_decay, grad_scale, first):
    call("spk_sgd_step", ptr(p), ptr(g), ptr(buf), p.numel(), float(lr), float(momentum), float(weight_decay),
         float(grad_scale), 1 if first else 0, stream())


# ---- scoring back end ---------------------------------------------------------------------------------------------
def center_normalize(emb, mean=None, eps=1e-8):
    """rows of emb [N][D] minus mean, scaled to unit length (max(norm, eps))."""
    N, D = emb.shape
    out = torch.empty_like(emb)
    call("spk_center_normalize", ptr(emb), ptr(mean), ptr(out), N, D, float(eps), stream())
    return out


def trial_cosine(en, te, ia, ib):
    """out[t] = <en[ia[t]], te[ib[t]]>; ia / ib int32 device tensors (validated by the caller)."""
    T = ia.numel()
    out = torch.empty(T, device=en.device, dtype=torch.float32)
    assert ia.dtype == torch.int32 and ib.dtype == torch.int32 and ib.numel() == T and en.shape[1] == te.shape[1]
    call("spk_trial_cosine", ptr(en), ptr(te), ptr(ia), ptr(ib), ptr(out), T, en.shape[1], en.shape[0], te.shape[0], stream())
    return out


def topk_mean_std(scores, k):
    """mean / unbiased std of the k largest entries of every row of scores [N][M]."""
    N, M = scores.shape
    mean = torch.empty(N, device=scores.device, dtype=torch.float32)
    std = torch.empty_like(mean)
    call("spk_topk_mean_std", ptr(scores), ptr(mean), ptr(std), N, M, int(k), scores.stride(0), stream())
    return mean, std
