"""f16 pair tensors and rigorous operand-scale bounds of the f16x3 mode, on a real MI355X.

The gradient wrt a raw conv output (draw) is written ONCE in the two-term fp16 form - by spk_bn_bwd_apply(pair_scale=) or as
the side output of a fused BatchNorm-backward data gradient (SPK_SIDE_PRESPLIT) - under a scale slot that holds a rigorous
upper bound of |draw| (spk_bn_bwd_finalize est_out), and its consumers stage it by plain copy.  Because the conversion is the
same arithmetic (split2h with the same power-of-two scale) wherever it happens, every pair path must reproduce the fp32-draw
path that uses the same slot BIT FOR BIT; the bound must hold on heavy-tailed gradients and outlier channels (nothing
saturates, counted by spk_f16_window_count); and the C ABI refuses an f16x3 launch without its scale slots.
Reference semantics: autograd of nn.BatchNorm2d + nn.Conv2d (scripts/model.py:41-44, scripts/train_resnet.py:327)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import weights as W  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd import ops as _ops
    return _ops


@pytest.fixture(autouse=True)
def f16x3(ops):
    old = ops.SPLIT
    ops.SPLIT = ops.MFMA_MODES["f16x3"]
    yield
    ops.SPLIT = old


def rnd(seed, *shape, scale=1.0, shift=0.0):
    n = int(np.prod(shape))
    return torch.from_numpy(((W.hash_uniform(seed, 1, n) * 2 - 1) * scale + shift).astype(np.float32).reshape(shape))


from helpers import decode_pairs, encode_pairs, sigma_of, slot, slot_value  # noqa: E402,F401


SHAPES = [
    # B, Cin (channels of x / of the data gradient's output), Cout (channels of draw), H, W, ksize, stride
    (2, 64, 64, 23, 41, 3, 1),      # in-wave pipelined data gradient with pair staging
    (2, 128, 256, 20, 37, 3, 1),
    (2, 32, 64, 20, 75, 3, 2),      # stride 2: four parity-class launches of conv_mfma_kernel, conv_wgrad_split_kernel
    (2, 64, 64, 7, 9, 1, 1),        # 1x1: several channel planes per barrier, conv_wgrad_1x1_kernel
    (2, 32, 32, 19, 45, 3, 1),      # the layer-1 shape class (conv_mfma_kernel, conv_wgrad_split_kernel<9,4,1>)
]


@pytest.mark.parametrize("shape", SHAPES)
def test_pair_tensor_paths_equal_the_fp32_draw_paths_bit_for_bit(ops, shape):
    B, Cin, Cout, H, Wd, k, s = shape
    pad = 1 if k == 3 else 0
    OH, OW = (H + 2 * pad - k) // s + 1, (Wd + 2 * pad - k) // s + 1
    x = torch.relu(rnd(1, B, H, Wd, Cin) + 0.2).cuda()
    w = rnd(2, Cout, Cin, k, k, scale=0.1).cuda()
    raw = rnd(3, B, OH, OW, Cout, scale=2.0, shift=0.3).cuda()
    g = (rnd(4, B, OH, OW, Cout) * torch.exp(2.0 * rnd(5, B, OH, OW, Cout))).cuda() * 1e-4      # heavy-ish tails, small values
    N = B * OH * OW
    r2 = raw.reshape(N, Cout)
    mean, var = r2.mean(0), r2.var(0, unbiased=False)
    gamma, beta = rnd(6, Cout, scale=0.3, shift=1.0).cuda(), rnd(7, Cout, scale=0.2).cuda()
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    bn4 = torch.stack([mean, invstd, gamma * invstd, beta - mean * gamma * invstd]).contiguous()
    wpk_t = ops.pack_conv_weight(w, transpose=True)
    act = rnd(8, B, OH, OW, Cout).cuda()                   # the ReLU mask comes from an explicit activation: act > 0
    g_amax = ops.absmax_into(g, slot())
    raw_amax = ops.absmax_into(raw, slot())
    x_amax = ops.absmax_into(x, slot())
    part = ops.bn_bwd_partial(g, raw, act, bn4, ops.MASK_ACT)
    dg, db = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")

    # ---- separate BatchNorm-backward pass: pair output vs fp32 output under the same slot
    est = slot()
    draw_p = ops.bn_backward(g, raw, act, bn4, gamma, dg, db, ops.MASK_ACT, partial=part, pair=(g_amax, raw_amax, est))
    true_amax = slot()
    draw_f = ops.bn_backward(g, raw, act, bn4, gamma, dg, db, ops.MASK_ACT, partial=part, amax_out=true_amax)
    torch.cuda.synchronize()
    bound, truth, sig = slot_value(est), slot_value(true_amax), sigma_of(est)
    assert truth == float(draw_f.abs().max()) and bound >= truth > 0, (bound, truth)
    assert 16384.0 <= bound * sig < 32768.0          # the bound sits in [2^14, 2^15): nothing can saturate
    print("shape %s: BatchNorm-backward bound / true absmax = %.2f" % (shape, bound / truth))
    assert torch.equal(draw_p.cpu().view(torch.int32), encode_pairs(draw_f.cpu(), sig).view(torch.int32))
    dec = decode_pairs(draw_p.cpu(), sig)
    assert float((dec - draw_f.cpu().double()).abs().max()) <= 2.0 ** -21 * truth + 2.0 ** -25 / sig
    cnt = torch.zeros(4, device="cuda", dtype=torch.int64)
    ops.f16_window_count(draw_p, est, cnt, pairs=True)
    cnt2 = torch.zeros(4, device="cuda", dtype=torch.int64)
    ops.f16_window_count(draw_f, est, cnt2)
    assert cnt.tolist() == cnt2.tolist() and cnt[0] == draw_f.numel() and cnt[1] == 0, (cnt.tolist(), cnt2.tolist())

    # ---- consumers: data gradient and weight gradient, pair staging vs conversion while staging
    dx_f = ops.conv_dgrad(draw_f, wpk_t, Cin, k, s, (H, Wd), in_amax=est)
    dx_p = ops.conv_dgrad(draw_p, wpk_t, Cin, k, s, (H, Wd), in_amax=est, in_presplit=True)
    assert torch.equal(dx_f, dx_p), "data gradient: pair staging differs from conversion while staging"
    dw_f, dw_p = torch.empty(Cout, Cin, k, k, device="cuda"), torch.empty(Cout, Cin, k, k, device="cuda")
    # (same kernel for both: the 16x16x32 form - csrc/conv_wgrad_wm16.hip - exists for pair tensors only and sums in another order)
    old_m16, ops.WM16, ops.C32M16 = (ops.WM16, ops.C32M16), False, False
    try:
        ops.conv_wgrad(x, draw_f, dw_f, k, s, dy_amax=est, x_amax=x_amax)
        ops.conv_wgrad(x, draw_p, dw_p, k, s, dy_amax=est, x_amax=x_amax, dy_presplit=True)
    finally:
        ops.WM16, ops.C32M16 = old_m16
    assert torch.equal(dw_f, dw_p), "weight gradient: pair staging differs from conversion while staging"
    ops.conv_wgrad(x, draw_p, dw_p, k, s, dy_amax=est, x_amax=x_amax, dy_presplit=True)      # the default pair path
    assert float((dw_p - dw_f).abs().max()) <= 2e-6 * float(dw_f.abs().max()), "weight gradient: the default pair path"
    # ... and both are the gradients autograd gives for draw (fp64 yardstick)
    xc = x.cpu().permute(0, 3, 1, 2).double().requires_grad_(True)
    wc = w.cpu().double().requires_grad_(True)
    gx, gw = torch.autograd.grad(F.conv2d(xc, wc, None, s, pad), [xc, wc], grad_outputs=draw_f.cpu().permute(0, 3, 1, 2).double())
    e1 = float((dx_p.cpu().permute(0, 3, 1, 2).double() - gx).abs().max() / gx.abs().max())
    e2 = float((dw_p.cpu().double() - gw).abs().max() / gw.abs().max())
    assert e1 < 2e-5 and e2 < 3e-5, (e1, e2)

    # ---- fused BatchNorm-backward data gradient (stride 1): the side output as a pair tensor
    if s == 1:
        est2 = slot()
        coef = ops.bn_bwd_coef(part, N, gamma, bn4, dg, db, amax_in=g_amax, raw_amax=raw_amax, est_out=est2)
        torch.cuda.synchronize()
        assert slot_value(est2) == bound                       # the finalize gives the same bound with or without the apply
        sd_f, sd_p = torch.empty_like(raw), torch.empty_like(raw)
        a_f, a_p = slot(), slot()
        fx_f = ops.conv_dgrad(g, wpk_t, Cin, k, 1, (H, Wd), in_bnbwd=(raw, act, bn4, coef), side=(sd_f, None), in_amax=est2,
                              side_amax=a_f)
        fx_p = ops.conv_dgrad(g, wpk_t, Cin, k, 1, (H, Wd), in_bnbwd=(raw, act, bn4, coef), side=(sd_p, None), in_amax=est2,
                              side_amax=a_p, side_presplit=True)
        torch.cuda.synchronize()
        assert torch.equal(fx_f, fx_p)
        assert slot_value(a_f) == slot_value(a_p) == float(sd_f.abs().max()) <= bound
        assert torch.equal(sd_p.cpu().view(torch.int32), encode_pairs(sd_f.cpu(), sig).view(torch.int32))
        # the fused form and the separate pass compute the same draw up to the contraction of the fp32 expression
        assert float((sd_f - draw_f).abs().max()) <= 4e-6 * truth
        old_m16, ops.WM16, ops.C32M16 = (ops.WM16, ops.C32M16), False, False      # same kernel for both (the 16x16x32 forms take pair tensors only)
        try:
            ops.conv_wgrad(x, sd_p, dw_p, k, 1, dy_amax=est2, x_amax=x_amax, dy_presplit=True)
            ops.conv_wgrad(x, sd_f, dw_f, k, 1, dy_amax=est2, x_amax=x_amax)
        finally:
            ops.WM16, ops.C32M16 = old_m16
        assert torch.equal(dw_f, dw_p)


def _bn_rows(raw, gamma):
    """[mean, invstd, scale, shift] rows of a train-mode BatchNorm over the rows of raw [N][C] (fp64 statistics)"""
    r2 = raw.reshape(-1, raw.shape[-1]).double()
    mean, var = r2.mean(0), r2.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    return torch.stack([mean, invstd, gamma.double() * invstd, -mean * gamma.double() * invstd]).float().contiguous()


def _heavy_case(B, H, Wd, C, outlier):
    raw = rnd(1, B, H, Wd, C, scale=1.5, shift=0.2)
    raw[0, 3, 4, 5] = outlier                 # one value far outside its channel: |xhat| of that element ~ sqrt(N)
    raw[..., 9] *= 1e-3                       # a channel with a tiny variance: large invstd
    g = rnd(2, B, H, Wd, C) * torch.exp(4.0 * rnd(3, B, H, Wd, C)) * 1e-6        # gradients over six decades
    g[1, 2, 3, 7] = 0.5                       # and one element 10^5 times the rest
    act = rnd(6, B, H, Wd, C, shift=0.3)      # explicit activation: mask = act > 0
    gamma = rnd(5, C, scale=0.3, shift=1.0)
    return raw, g, act, gamma


def test_bnbwd_bound_holds_on_heavy_tails_and_outlier_channels(ops):
    """the operand scale of the BatchNorm-backward values used to come from a heuristic (|xhat| <= 8,
    x 64 headroom = 512) with a silent clamp beyond it.  Now it is a rigorous bound.  (a) full-size statistics: 288 000 values
    per channel, one of them a 10^6-sigma outlier (|xhat| = 536 > 512: the old scale would have clamped), gradients over six
    decades with a 10^5 x outlier: the pair tensor must not saturate and must carry every value with two-term accuracy or the
    documented absolute floor.  (b) the same kind of data through the fused data gradient + side output against fp64."""
    # ---- (a) separate pass, large N
    B, H, Wd, C = 12, 80, 300, 32
    raw, g, act, gamma = _heavy_case(B, H, Wd, C, outlier=1e6)
    bn4 = _bn_rows(raw, gamma).cuda()
    rawg, gg, actg, gam = raw.cuda(), g.cuda(), act.cuda(), gamma.cuda()
    g_amax, raw_amax = ops.absmax_into(gg, slot()), ops.absmax_into(rawg, slot())
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    est, true_amax = slot(), slot()
    draw_p = ops.bn_backward(gg, rawg, actg, bn4, gam, dg, db, ops.MASK_ACT, pair=(g_amax, raw_amax, est))
    draw_f = ops.bn_backward(gg, rawg, actg, bn4, gam, dg, db, ops.MASK_ACT, amax_out=true_amax)
    torch.cuda.synchronize()
    b64 = bn4.double().cpu()
    xhat_max = float((((raw.double() - b64[0]) * b64[1]).abs()).max())
    bound, truth, sig = slot_value(est), slot_value(true_amax), sigma_of(est)
    print("N = %d per channel: max |xhat| %.0f (old heuristic: 8, x 64 headroom); bound / true absmax of draw = %.2f" % (
        B * H * Wd, xhat_max, bound / truth))
    assert xhat_max > 512.0 and bound >= truth > 0
    cnt = torch.zeros(4, device="cuda", dtype=torch.int64)
    ops.f16_window_count(draw_p, est, cnt, pairs=True)
    total, sat, lo_sub, hi_sub = cnt.tolist()
    print("window: %d values, %d saturated, %.2f %% low term subnormal, %.2f %% high term subnormal" % (
        total, sat, 100.0 * lo_sub / total, 100.0 * hi_sub / total))
    assert sat == 0 and total == draw_f.numel()
    err = (decode_pairs(draw_p.cpu(), sig) - draw_f.cpu().double()).abs()
    # two fp16 terms: 2^-22 relative; fp16 subnormals survive (tools/probe: kept by the matrix instruction), so below the
    # normal range the terms have an ABSOLUTE resolution of 2^-24 / sigma: error <= 2^-25 / sigma <= bound * 2^-39
    assert bool((err <= 2.0 ** -21 * draw_f.cpu().double().abs() + 2.0 ** -24 / sig).all())
    assert 2.0 ** -24 / sig <= bound * 2.0 ** -38

    # ---- (b) fused data gradient with the side output as a pair tensor, against fp64
    B, H, Wd, C = 2, 12, 17, 64
    raw, g, act, gamma = _heavy_case(B, H, Wd, C, outlier=4000.0)
    w = rnd(4, C, C, 3, 3, scale=0.05)
    N = B * H * Wd
    bn4 = _bn_rows(raw, gamma).cuda()
    rawg, gg, wg, gam, actg = raw.cuda(), g.cuda(), w.cuda(), gamma.cuda(), act.cuda()
    g_amax, raw_amax = ops.absmax_into(gg, slot()), ops.absmax_into(rawg, slot())
    part = ops.bn_bwd_partial(gg, rawg, actg, bn4, ops.MASK_ACT)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    est = slot()
    coef = ops.bn_bwd_coef(part, N, gam, bn4, dg, db, amax_in=g_amax, raw_amax=raw_amax, est_out=est)
    # fp64 yardstick of draw from the SAME coefficients / statistics (the rounding under test is the conv's operand form)
    c64, b64 = coef.double().cpu(), bn4.double().cpu()
    dz = g.double() * (act > 0)
    draw64 = c64[0] * (dz - c64[1] - ((raw.double() - b64[0]) * b64[1]) * c64[2])
    sd = torch.empty_like(rawg)
    true_amax = slot()
    wpk_t = ops.pack_conv_weight(wg, transpose=True)
    dx = ops.conv_dgrad(gg, wpk_t, C, 3, 1, (H, Wd), in_bnbwd=(rawg, actg, bn4, coef), side=(sd, None), in_amax=est,
                        side_amax=true_amax, side_presplit=True)
    torch.cuda.synchronize()
    bound, truth = slot_value(est), slot_value(true_amax)
    assert bound >= truth and bound >= float(draw64.abs().max()) * (1 - 1e-6)
    cnt = torch.zeros(4, device="cuda", dtype=torch.int64)
    ops.f16_window_count(sd, est, cnt, pairs=True)
    torch.cuda.synchronize()
    assert cnt[1] == 0
    sig = sigma_of(est)
    err = (decode_pairs(sd.cpu(), sig) - draw64).abs()
    assert bool((err <= 2.0 ** -20 * draw64.abs() + 2.0 ** -24 / sig + 4e-7 * float(draw64.abs().max())).all())
    xin = torch.zeros(B, C, H, Wd, dtype=torch.float64, requires_grad=True)
    gx, = torch.autograd.grad(F.conv2d(xin, w.double(), None, 1, 1), [xin], grad_outputs=draw64.permute(0, 3, 1, 2))
    mag, = torch.autograd.grad(F.conv2d(xin, w.double().abs(), None, 1, 1), [xin], grad_outputs=draw64.abs().permute(0, 3, 1, 2))
    e = (dx.cpu().permute(0, 3, 1, 2).double() - gx).abs()
    floor = 576 * bound * float(w.abs().max()) * 2.0 ** -30
    assert bool((e <= 3e-6 * mag + floor).all()), float((e - 3e-6 * mag).max() / floor)


def test_f16x3_launches_without_scale_slots_are_refused(ops):
    """through the public C ABI a NULL dy_amax meant scale 1 (gradients of 1e-6 became fp16 subnormals: dw
    silently ~0) and a NULL in_amax a static 2^6 (|v| > 1023 clamped).  Both are argument errors now."""
    from pytorch_kaldi_resnet_amd import hip
    B, C, H, Wd = 1, 64, 8, 8
    x, dy = torch.zeros(B, H, Wd, C, device="cuda"), torch.zeros(B, H, Wd, C, device="cuda")
    dw, ws = torch.zeros(C, C, 3, 3, device="cuda"), torch.zeros(9 * C * C, device="cuda")
    s = slot()
    st = torch.cuda.current_stream().cuda_stream
    for dy_amax, x_amax in ((None, s), (s, None), (None, None)):
        rc = hip.lib().spk_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), None, None, B, H, Wd, C, H, Wd, C,
                                      3, 1, 4, 8, 2, 1, 0, 0, 3, dy_amax.data_ptr() if dy_amax is not None else None,
                                      x_amax.data_ptr() if x_amax is not None else None, st)
        assert rc != 0 and b"dy_amax and x_amax" in hip.lib().spk_last_error()
    wpk = ops.pack_conv_weight(torch.zeros(C, C, 3, 3, device="cuda"))
    out = torch.zeros(B, H, Wd, C, device="cuda")
    z = ctypes.c_int * 9
    taps = [(kh - 1, kw - 1, kh * 3 + kw) for kh in range(3) for kw in range(3)]
    args = [x.data_ptr(), wpk.data_ptr(), out.data_ptr()] + [None] * 18 + [B, H, Wd, C, H, Wd, H, Wd, C, 1, 1, 0, 0, 9,
                                                                            z(*[t[0] for t in taps]), z(*[t[1] for t in taps]),
                                                                            z(*[t[2] for t in taps]), 8, 8, 1, 2, 1, 1, 0, 3]
    rc = hip.lib().spk_conv_mfma(*(args + [None, None, None, st]))
    assert rc != 0 and b"needs in_amax" in hip.lib().spk_last_error()
    rc = hip.lib().spk_conv_mfma(*(args + [s.data_ptr(), None, None, st]))
    assert rc == 0
    torch.cuda.synchronize()
