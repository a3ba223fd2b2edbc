"""Per-kernel parity on a real MI355X: every libspkhip export against the CPU oracle (oracle/spk_oracle.py)
or, for bare ops the oracle only calls through torch (conv2d / batch_norm), against the same torch-CPU
fp32 op.  All inputs are seeded; sizes finish in seconds on CPU.  Tolerances are stated per assertion
(fp32 MFMA = exact fp32 FMA chain; only the summation order differs from ATen's)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import spk_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd import ops as _ops
    return _ops


def needs_experimental():
    """the producer / consumer and the other measured-not-faster kernel forms are compiled only with SPK_EXPERIMENTAL=1
    (python pytorch-kaldi-resnet_amd/build.py --experimental, then SPK_LIB=.../variants/libspkhip_exp.so)"""
    from pytorch_kaldi_resnet_amd import hip
    if not hip.has_experimental():
        pytest.skip("library built without SPK_EXPERIMENTAL")


def rnd(seed, *shape, scale=1.0, shift=0.0):
    n = int(np.prod(shape))
    return torch.from_numpy(((W.hash_uniform(seed, 1, n) * 2 - 1) * scale + shift).astype(np.float32).reshape(shape))


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def nhwc(x):   # NCHW cpu -> NHWC cuda
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(x):   # NHWC cuda -> NCHW cpu
    return x.cpu().permute(0, 3, 1, 2).contiguous()


def sign_mask(act_nchw):
    """[B,C,H,W] cpu tensor -> the int32 sign-mask words spk_bn_apply emits for it ([B*H*W][C/32], bit k = channel 32j+k > 0)."""
    B, C, H, Wd = act_nchw.shape
    bits = (act_nchw.permute(0, 2, 3, 1).reshape(-1, C // 32, 32) > 0).numpy().astype(np.uint64)
    words = (bits << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    return torch.from_numpy(words.view(np.int32).reshape(-1)).cuda()


CONV_CASES = [
    # B, Cin, Cout, H, W, ksize, stride
    (2, 32, 32, 9, 13, 3, 1),
    (1, 32, 32, 80, 300, 3, 1),
    (2, 64, 128, 10, 38, 3, 1),
    (2, 32, 64, 20, 75, 3, 2),
    (2, 128, 256, 20, 75, 3, 2),
    (3, 32, 64, 21, 14, 1, 2),
    (2, 64, 64, 7, 9, 1, 1),
    (2, 256, 256, 10, 38, 3, 1),
    (1, 256, 256, 5, 3, 3, 1),
    (2, 128, 256, 10, 19, 1, 2),    # 1x1: four input-channel groups per block (conv_wgrad_1x1_kernel), strided
    (2, 256, 64, 9, 11, 1, 1),
    # C -> C at stride 1: the streaming 1x1 kernel (conv1x1_stream.hip), with a ragged last tile each
    (2, 128, 128, 20, 75, 1, 1),
    (3, 32, 32, 13, 11, 1, 1),
    (1, 64, 64, 40, 150, 1, 1),
]


@pytest.fixture(params=["f32", "bf16x6", "bf16x9", "f16x3"])
def mfma_mode(request, ops):
    """operand mode of the 3x3 convolutions: native fp32 MFMA, or fp32 operands split into three bf16 terms with the 6 /
    9 cross terms on the bf16 MFMA (fp32 accumulate).  Every mode must meet the SAME tolerances below."""
    old = ops.SPLIT
    ops.SPLIT = ops.MFMA_MODES[request.param]
    yield request.param
    ops.SPLIT = old


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(ops, case, mfma_mode):
    B, Cin, Cout, H, Wd, k, s = case
    if k == 1 and mfma_mode in ("bf16x6", "bf16x9"):
        pytest.skip("1x1 convolutions use fp32 operands in the bf16-term modes (they run on fp16 terms in f16x3)")
    pad = 1 if k == 3 else 0
    x = rnd(1, B, Cin, H, Wd)
    w = rnd(2, Cout, Cin, k, k, scale=0.2)
    ref = F.conv2d(x, w, None, s, pad)
    xg, wg = nhwc(x), w.cuda()
    wpk = ops.pack_conv_weight(wg)
    out, _ = ops.conv_fwd(xg, wpk, Cout, k, s)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (B, ref.shape[2], ref.shape[3], Cout)
    e = relerr(nchw(out), ref)
    assert e < 2e-5, "fwd %g" % e
    # stats epilogue (train-mode BN partial sums)
    out2, st = ops.conv_fwd(xg, wpk, Cout, k, s, stats=True)
    tot = st.double().sum(0).cpu()
    np.testing.assert_allclose(tot[:, 0].numpy(), ref.double().sum((0, 2, 3)).numpy(), rtol=1e-4,
                               atol=1e-4 * float(ref.abs().sum((0, 2, 3)).max()))
    np.testing.assert_allclose(tot[:, 1].numpy(), (ref.double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-4)
    # fused input BN+ReLU and inference epilogue (affine + residual + relu)
    isc, ish = rnd(3, Cin, scale=0.5, shift=1.0), rnd(4, Cin, scale=0.3)
    esc, esh = rnd(5, Cout, scale=0.5, shift=1.0), rnd(6, Cout, scale=0.3)
    res = rnd(7, *ref.shape)
    xa = F.relu(x * isc.view(1, -1, 1, 1) + ish.view(1, -1, 1, 1))
    ref2 = F.relu(F.conv2d(xa, w, None, s, pad) * esc.view(1, -1, 1, 1) + esh.view(1, -1, 1, 1) + res)
    out3, _ = ops.conv_fwd(xg, wpk, Cout, k, s, in_affine=(isc.cuda(), ish.cuda()), epi_affine=(esc.cuda(), esh.cuda()),
                           epi_add=nhwc(res), relu=True)
    e = relerr(nchw(out3), ref2)
    assert e < 2e-5, "fused fwd %g" % e
    # data gradient
    dy = rnd(8, *ref.shape)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    gx, gw = torch.autograd.grad((F.conv2d(xr, wr, None, s, pad) * dy).sum(), [xr, wr])
    wpk_t = ops.pack_conv_weight(wg, transpose=True)
    dx = ops.conv_dgrad(nhwc(dy), wpk_t, Cin, k, s, (H, Wd))
    e = relerr(nchw(dx), gx)
    assert e < 2e-5, "dgrad %g" % e
    addt = rnd(9, B, Cin, H, Wd)
    dx2 = ops.conv_dgrad(nhwc(dy), wpk_t, Cin, k, s, (H, Wd), add=nhwc(addt))
    e = relerr(nchw(dx2), gx + addt)
    assert e < 2e-5, "dgrad+add %g" % e
    if s == 1:
        gate = rnd(16, B, Cin, H, Wd)
        dx2m = ops.conv_dgrad(nhwc(dy), wpk_t, Cin, k, s, (H, Wd), add=nhwc(addt), add_mask=sign_mask(gate))
        e = relerr(nchw(dx2m), gx + addt * (gate > 0))
        assert e < 2e-5, "dgrad + masked add %g" % e
    dx3 = nhwc(addt).clone()
    ops.conv_dgrad(nhwc(dy), wpk_t, Cin, k, s, (H, Wd), out=dx3, accumulate=True)
    e = relerr(nchw(dx3), gx + addt)
    assert e < 2e-5, "dgrad accumulate %g" % e
    # data gradient with the fused BatchNorm-backward reduction (EPI_BNBWD): partial rows must sum to
    # (sum dz, sum dz*xhat) with dz = dx * mask, for both mask sources
    if s == 1:
        rawt = rnd(10, B, Cin, H, Wd, scale=2.0, shift=0.3)
        bn4 = torch.stack([rnd(11, Cin, scale=0.3), rnd(12, Cin, scale=0.2, shift=1.0), rnd(13, Cin, scale=0.5, shift=1.0),
                           rnd(14, Cin, scale=0.4)])
        actt = rnd(15, B, Cin, H, Wd)
        for act, use_mask in ((None, False), (actt, False), (actt, True)):
            bnb = (nhwc(rawt), None if act is None else nhwc(act), bn4.cuda())
            if use_mask:
                bnb = bnb + (sign_mask(act),)        # 1-bit form of the same mask: identical result
            dxb, part = ops.conv_dgrad(nhwc(dy), wpk_t, Cin, k, s, (H, Wd), add=nhwc(addt), bn_bwd=bnb)
            assert relerr(nchw(dxb), gx + addt) < 2e-5
            v = (gx + addt).double()
            mask = ((rawt * bn4[2].view(1, -1, 1, 1) + bn4[3].view(1, -1, 1, 1)) > 0) if act is None else (act > 0)
            dzr = v * mask
            xh = (rawt.double() - bn4[0].view(1, -1, 1, 1)) * bn4[1].view(1, -1, 1, 1)
            tot = part.double().sum(0).cpu()
            ref0, ref1 = dzr.sum((0, 2, 3)), (dzr * xh).sum((0, 2, 3))
            assert float((tot[:, 0] - ref0).abs().max() / ref0.abs().max()) < 1e-4
            assert float((tot[:, 1] - ref1).abs().max() / ref1.abs().max()) < 1e-4
    # data gradient whose input is BatchNorm-backward(dy) applied in the staging (IN_BNBWD), with side outputs
    if s == 1:
        Co = Cout
        rawo = rnd(20, B, Co, ref.shape[2], ref.shape[3], scale=2.0, shift=0.3)
        acto = rnd(21, B, Co, ref.shape[2], ref.shape[3])
        bn4o = torch.stack([rnd(22, Co, scale=0.3), rnd(23, Co, scale=0.2, shift=1.0), rnd(24, Co, scale=0.5, shift=1.0),
                            rnd(25, Co, scale=0.4)])
        coef = torch.stack([rnd(26, Co, scale=0.3, shift=1.0), rnd(27, Co, scale=0.05), rnd(28, Co, scale=0.05)])
        v4 = lambda t: t.view(1, -1, 1, 1)
        for act, use_mask in ((None, False), (acto, False), (acto, True)):
            mask = ((rawo * v4(bn4o[2]) + v4(bn4o[3])) > 0) if act is None else (act > 0)
            dzr = dy * mask
            xh = (rawo - v4(bn4o[0])) * v4(bn4o[1])
            draw_ref = v4(coef[0]) * (dzr - v4(coef[1]) - xh * v4(coef[2]))
            gx_ref, = torch.autograd.grad(F.conv2d(xr, wr, None, s, pad), [xr], grad_outputs=draw_ref)
            sd, sz = torch.zeros(B, ref.shape[2], ref.shape[3], Co, device="cuda"), torch.zeros(B, ref.shape[2], ref.shape[3], Co, device="cuda")
            inb = (nhwc(rawo), None if act is None else nhwc(act), bn4o.cuda(), coef.cuda())
            if use_mask:
                inb = inb + (sign_mask(act),)
            dxf = ops.conv_dgrad(nhwc(dy), wpk_t, Cin, k, s, (H, Wd), add=nhwc(addt), in_bnbwd=inb, side=(sd, sz))
            assert relerr(nchw(dxf), gx_ref + addt) < 3e-5, "dgrad IN_BNBWD"
            assert relerr(nchw(sd), draw_ref) < 1e-5, "side draw"
            assert relerr(nchw(sz), dzr) == 0.0, "side dz"
    # weight gradient (plain and with the fused input transform)
    dw = torch.empty(Cout, Cin, k, k, device="cuda")
    ops.conv_wgrad(xg, nhwc(dy), dw, k, s)
    e = relerr(dw.cpu(), gw)
    assert e < 3e-5, "wgrad %g" % e
    xr2 = x.clone()
    gw2, = torch.autograd.grad((F.conv2d(F.relu(xr2 * isc.view(1, -1, 1, 1) + ish.view(1, -1, 1, 1)), wr, None, s, pad) * dy).sum(), [wr])
    ops.conv_wgrad(xg, nhwc(dy), dw, k, s, in_affine=(isc.cuda(), ish.cuda()))
    e = relerr(dw.cpu(), gw2)
    assert e < 3e-5, "wgrad fused %g" % e
    ops.conv_wgrad(xg, nhwc(dy), dw, k, s, in_affine=(isc.cuda(), ish.cuda()), accumulate=True)
    e = relerr(dw.cpu(), 2 * gw2)
    assert e < 3e-5, "wgrad accumulate %g" % e


@pytest.mark.parametrize("layout", [(2, 1, 1), (4, 1, 1), (3, 2, 1), (3, 1, 2), (6, 1, 2), (3, 1, 4), (6, 1, 4)])
@pytest.mark.parametrize("shape", [(2, 128, 11, 23, 1), (3, 128, 20, 27, 2)])
@pytest.mark.parametrize("wsmode", [6, 3])
def test_wave_specialised_conv_equals_the_reference_kernel(ops, layout, shape, wsmode):
    """csrc/conv_ws_kernel.h (producer / consumer waves, persistent blocks, LDS ring) against conv_mfma_kernel on ragged
    maps - tiles that overhang both edges, more tiles than blocks and fewer: forward (stride 1 and 2, fused input
    BN+ReLU, BN statistics), plain data gradient with shortcut add, and the fused BatchNorm-backward data gradient with
    sign masks, side output and BN-backward statistics.  Same MFMA order per accumulator => the convolution outputs are
    bit-identical; the per-wave statistics rows are grouped differently, so they are compared after reduction."""
    needs_experimental()
    MT, NT, WC = layout
    B, C, H, Wd, stride = shape
    old = (ops.SPLIT, ops.WS_CONV, ops.WS_FORCE)
    ops.SPLIT = wsmode
    try:
        tile = (5, 12, MT, NT, WC)       # 60 pixels: fits the smallest layout (96), leaves every layout's padding rows idle
        torch.manual_seed(7)
        x = torch.randn(B, H, Wd, C, device="cuda")
        w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
        wpk, wpk_t = ops.pack_conv_weight(w), ops.pack_conv_weight(w, True)
        sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
        OH, OW = ops.conv_out_hw(H, Wd, 3, stride)
        dy = torch.randn(B, H, Wd, C, device="cuda")
        raw, raw_p, dout = (torch.randn(B, H, Wd, C, device="cuda") for _ in range(3))
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        m1 = torch.randint(-2 ** 31, 2 ** 31 - 1, (B * H * Wd * (C // 32),), device="cuda", dtype=torch.int32, generator=g)
        m2 = torch.randint(-2 ** 31, 2 ** 31 - 1, (B * H * Wd * (C // 32),), device="cuda", dtype=torch.int32, generator=g)
        bn4 = torch.stack([torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5,
                           torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1])
        coef = torch.stack([torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.01,
                            torch.randn(C, device="cuda") * 0.01])
        e2 = torch.rand(2, C, device="cuda") + 0.5
        res = {}
        for ws in (False, True):
            ops.WS_CONV, ops.WS_FORCE = ("1" if ws else "0"), (tile if ws else None)
            out, st = ops.conv_fwd(x, wpk, C, 3, stride, in_affine=(sc, sh), stats=True)
            res_in = out * 0.5
            out2, _ = ops.conv_fwd(x, wpk, C, 3, stride, epi_affine=(e2[0], e2[1]), epi_add=res_in, relu=True)
            draw = torch.full_like(raw, 7.0)
            dx, part = ops.conv_dgrad(dy, wpk_t, C, 3, 1, (H, Wd), add=dout, add_mask=m2, bn_bwd=(raw_p, None, bn4, m2),
                                      in_bnbwd=(raw, None, bn4, coef, m1), side=(draw, None))
            dxp = ops.conv_dgrad(dy, wpk_t, C, 3, 1, (H, Wd), add=dout)
            res[ws] = (out, st, out2, dx, draw, part, dxp)
        a, b = res[False], res[True]
        for i in (0, 2, 3, 4, 6):
            assert torch.equal(a[i], b[i]), i
        for i in (1, 5):
            assert torch.allclose(a[i].double().sum(0), b[i].double().sum(0), rtol=1e-5, atol=1e-3), i
    finally:
        ops.SPLIT, ops.WS_CONV, ops.WS_FORCE = old


@pytest.mark.parametrize("shape", [(3, 32, 32, 16, 40, 1), (2, 64, 128, 21, 37, 1), (2, 128, 128, 20, 75, 1), (4, 64, 128, 20, 31, 2),
                                   (2, 256, 256, 10, 38, 1)])
def test_wave_specialised_wgrad_equals_the_reference_kernel(ops, shape):
    """conv_wgrad_ws_kernel (eight waves: four stage the next region, four run the MFMAs; csrc/conv_wgrad_split.hip) and
    conv_wgrad_pipe_kernel (the next region staged inside the K loop of the current one, csrc/conv_wgrad_pipe.hip)
    against conv_wgrad_split_kernel on the same tile and the same slab count: same MFMA order per accumulator and the
    same slab reduce => the weight gradients are bit-identical (with and without the fused BN + ReLU on X)."""
    needs_experimental()
    from pytorch_kaldi_resnet_amd import tiling
    B, Cin, Cout, H, Wd, stride = shape
    old = (ops.SPLIT, ops.WS_WGRAD, ops.WS_WGRAD_BLOCKS, ops.PIPE_WGRAD, ops.GROUPED_3X3)
    ops.SPLIT, ops.GROUPED_3X3 = 3, False
    try:
        torch.manual_seed(11)
        x = torch.randn(B, H, Wd, Cin, device="cuda")
        OH, OW = ops.conv_out_hw(H, Wd, 3, stride)
        dy = torch.randn(B, OH, OW, Cout, device="cuda") * 1e-3
        sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
        res = {}
        for ws in (False, True, "pipe"):
            ops.WS_WGRAD, ops.WS_WGRAD_BLOCKS, ops.PIPE_WGRAD = ws is True, tiling.WGRAD_TARGET_BLOCKS, ws == "pipe"
            dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
            ops.conv_wgrad(x, dy, dw, 3, stride)
            dw2 = torch.full((Cout, Cin, 3, 3), 0.25, device="cuda")
            ops.conv_wgrad(x, dy, dw2, 3, stride, in_affine=(sc, sh), accumulate=True)
            res[ws] = (dw, dw2)
        assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
        assert torch.equal(res[False][0], res["pipe"][0]) and torch.equal(res[False][1], res["pipe"][1])      # conv_wgrad_pipe_kernel
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double().cpu(), (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double().cpu(),
                                          stride=stride, padding=1)
        err = (res[True][0].double().cpu() - ref).norm() / ref.norm()
        assert err < 1e-5, err
        # the 2 x 2 wave layout (conv_wgrad_wm_kernel, the default where Cin and Cout are multiples of 64) sums the pixels of a
        # region in one accumulator instead of two: same values to fp32 rounding, and as close to fp64 as the others
        ops.WS_WGRAD, ops.PIPE_WGRAD, ops.GROUPED_3X3 = False, False, True
        dwm = torch.empty(Cout, Cin, 3, 3, device="cuda")
        ops.conv_wgrad(x, dy, dwm, 3, stride)
        dwm2 = torch.full((Cout, Cin, 3, 3), 0.25, device="cuda")
        ops.conv_wgrad(x, dy, dwm2, 3, stride, in_affine=(sc, sh), accumulate=True)
        assert (dwm.double().cpu() - ref).norm() / ref.norm() < 1e-5
        assert torch.allclose(dwm, res[False][0], rtol=1e-4, atol=1e-6 * float(ref.abs().max()))
        assert torch.allclose(dwm2, res[False][1], rtol=1e-4, atol=1e-6 * float((res[False][1] - 0.25).abs().max()))
    finally:
        ops.SPLIT, ops.WS_WGRAD, ops.WS_WGRAD_BLOCKS, ops.PIPE_WGRAD, ops.GROUPED_3X3 = old


@pytest.mark.parametrize("shape", [(2, 32, 32, 19, 45, 1), (2, 64, 64, 23, 41, 1), (3, 64, 128, 20, 27, 2), (2, 128, 128, 20, 75, 1),
                                   (2, 256, 256, 10, 38, 1), (1, 128, 64, 9, 13, 2)])
def test_pipelined_conv_equals_the_reference_kernel(ops, shape):
    """conv_pipe_kernel (next chunk staged inside the current chunk's K loop, two LDS tiles, csrc/conv_kernel.h PIPE)
    against conv_mfma_kernel on the same tiles: same MFMA order per accumulator => bit-identical outputs and statistics,
    for the forward (fused input BN + ReLU, epilogue affine / add / ReLU) and the plain data gradient (stride 1 and 2)."""
    B, Cin, Cout, H, Wd, stride = shape
    from pytorch_kaldi_resnet_amd import tiling
    from pytorch_kaldi_resnet_amd import hip
    old = (ops.SPLIT, ops.PIPE_CONV, ops.WS_CONV, ops.PIPE_BNBWD, ops.PIPE_M16)
    # (the pipelined form of the FUSED BatchNorm-backward data gradient is an experimental kernel: compared only when built;
    #  the 16x16x32 form sums in another order and has its own test below)
    ops.SPLIT, ops.WS_CONV, ops.PIPE_BNBWD, ops.PIPE_M16 = 3, "0", hip.has_experimental(), False
    real_tile = tiling.conv_tile

    def tile_2x2_for_fused(*key, mode=0, split=0):           # the fused pipelined kernel exists for register tiles <= 2 x 2
        return (min(key[0], 10), min(key[1], 25), 2, 2 if key[6] >= 64 else 1) if mode == 1 else real_tile(*key, mode=mode, split=split)
    tiling.conv_tile = tile_2x2_for_fused
    try:
        torch.manual_seed(3)
        x = torch.randn(B, H, Wd, Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        wpk, wpk_t = ops.pack_conv_weight(w), ops.pack_conv_weight(w, True)
        sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
        e2 = torch.rand(2, Cout, device="cuda") + 0.5
        OH, OW = ops.conv_out_hw(H, Wd, 3, stride)
        dy = torch.randn(B, OH, OW, Cout, device="cuda")
        res_in = torch.randn(B, OH, OW, Cout, device="cuda")
        dadd = torch.randn(B, H, Wd, Cin, device="cuda")
        res = {}
        for pipe in (False, True):
            ops.PIPE_CONV = pipe
            out, st = ops.conv_fwd(x, wpk, Cout, 3, stride, in_affine=(sc, sh), stats=True)
            out2, _ = ops.conv_fwd(x, wpk, Cout, 3, stride, epi_affine=(e2[0], e2[1]), epi_add=res_in, relu=True)
            dx = ops.conv_dgrad(dy, wpk_t, Cin, 3, stride, (H, Wd), add=dadd)
            res[pipe] = (out, st, out2, dx)
            if stride == 1 and Cin == Cout:
                # data gradient with the BatchNorm backward fused into its input staging: mask as sign bits / recomputed
                # from the raw tensor, side outputs draw (+ dz), shortcut add with mask, BN-backward statistics in the epilogue
                g = torch.Generator(device="cuda")
                g.manual_seed(5)
                m1, m2 = (torch.randint(-2 ** 31, 2 ** 31 - 1, (B * H * Wd * (Cin // 32),), device="cuda", dtype=torch.int32,
                                        generator=g) for _ in range(2))
                torch.manual_seed(17)
                raw, raw_p, dout = (torch.randn(B, H, Wd, Cin, device="cuda") for _ in range(3))
                bn4 = torch.stack([torch.randn(Cin, device="cuda") * 0.1, torch.rand(Cin, device="cuda") + 0.5,
                                   torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1])
                coef = torch.stack([torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.01,
                                    torch.randn(Cin, device="cuda") * 0.01])
                draw, draw2, dzb = torch.full_like(raw, 7.0), torch.full_like(raw, 7.0), torch.full_like(raw, 7.0)
                fx, fpart = ops.conv_dgrad(dy, wpk_t, Cin, 3, 1, (H, Wd), add=dout, add_mask=m2, bn_bwd=(raw_p, None, bn4, m2),
                                           in_bnbwd=(raw, None, bn4, coef, m1), side=(draw, None))
                fx2 = ops.conv_dgrad(dy, wpk_t, Cin, 3, 1, (H, Wd), in_bnbwd=(raw, None, bn4, coef), side=(draw2, dzb))
                res[pipe] = res[pipe] + (fx, draw, fpart.double().sum(0), fx2, draw2, dzb)
        for i, (a, b) in enumerate(zip(res[False], res[True])):
            if i == 6:
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-3), i       # statistics: per-wave rows, compared after reduction
            else:
                assert torch.equal(a, b), i
        ref = torch.nn.functional.conv2d(torch.relu(x * sc + sh).permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), stride=stride, padding=1)
        err = (res[True][0].permute(0, 3, 1, 2).double().cpu() - ref).norm() / ref.norm()
        assert err < 1e-5, err
    finally:
        ops.SPLIT, ops.PIPE_CONV, ops.WS_CONV, ops.PIPE_BNBWD, ops.PIPE_M16 = old
        tiling.conv_tile = real_tile


@pytest.mark.parametrize("shape", [(2, 64, 64, 40, 75, (20, 19)), (2, 128, 128, 20, 75, (20, 19)), (2, 256, 256, 10, 38, (10, 38)),
                                   (1, 64, 128, 23, 41, (16, 24)), (1, 128, 64, 17, 50, (8, 48)), (3, 64, 192, 9, 20, (10, 38))])
def test_pipelined_conv_16x16x32_form(ops, shape):
    """conv_pipe_kernel<3,2,..,M16> (v_mfma_f32_16x16x32_f16, two taps per K step, planes walked in pairs; csrc/conv_kernel.h)
    against the 32x32x16 form on the same tile: the same products in another summation order - equal within fp32 accumulation
    error (2e-6 of the output range per element), as close to fp64 as the other form, and NOT required bit-identical.  Forward with
    fused input BN + ReLU and statistics, forward with epilogue affine / add / ReLU, plain data gradient with shortcut add, and
    the pair-input data gradient with masked add + BatchNorm-backward statistics (the batched-read epilogue).  Tiles with ragged
    edges, 6 / 8 / 16 planes of 16 channels."""
    B, Cin, Cout, H, Wd, tile = shape
    from pytorch_kaldi_resnet_amd import tiling
    from test_pairs_gpu import encode_pairs, sigma_of
    old = (ops.SPLIT, ops.PIPE_CONV, ops.WS_CONV, ops.PIPE_M16, ops.PROFILE)
    ops.SPLIT, ops.WS_CONV, ops.PIPE_CONV = 3, "0", True
    real_tile = tiling.conv_tile
    tiling.conv_tile = lambda *key, mode=0, split=0: (tile[0], tile[1], 3, 2)
    try:
        torch.manual_seed(3)
        x = torch.randn(B, H, Wd, Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        wpk, wpk_t = ops.pack_conv_weight(w), ops.pack_conv_weight(w, True)
        sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
        e2 = torch.rand(2, Cout, device="cuda") + 0.5
        dy = torch.randn(B, H, Wd, Cout, device="cuda") * 1e-3
        res_in = torch.randn(B, H, Wd, Cout, device="cuda")
        dadd = torch.randn(B, H, Wd, Cin, device="cuda") * 1e-3
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        m2 = torch.randint(-2 ** 31, 2 ** 31 - 1, (B * H * Wd * (Cin // 32),), device="cuda", dtype=torch.int32, generator=g)
        raw_p = torch.randn(B, H, Wd, Cin, device="cuda")
        bn4 = torch.stack([torch.randn(Cin, device="cuda") * 0.1, torch.rand(Cin, device="cuda") + 0.5,
                           torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1])
        slot = ops.absmax_into(dy, torch.zeros(1, device="cuda", dtype=torch.int32))
        dy_pairs = encode_pairs(dy.cpu(), sigma_of(slot)).cuda()
        res, labels = {}, {}
        for m16 in (False, True):
            ops.PIPE_M16 = m16
            ops.PROFILE = []
            out, st = ops.conv_fwd(x, wpk, Cout, 3, 1, in_affine=(sc, sh), stats=True)
            out2, _ = ops.conv_fwd(x, wpk, Cout, 3, 1, epi_affine=(e2[0], e2[1]), epi_add=res_in, relu=True)
            dx = ops.conv_dgrad(dy, wpk_t, Cin, 3, 1, (H, Wd), add=dadd)
            dxp, part = ops.conv_dgrad(dy_pairs, wpk_t, Cin, 3, 1, (H, Wd), add=dadd, add_mask=m2, bn_bwd=(raw_p, None, bn4, m2),
                                       in_amax=slot, in_presplit=True)
            torch.cuda.synchronize()
            labels[m16] = [p[0] for p in ops.PROFILE if p[0].startswith("conv_")]
            ops.PROFILE = None
            res[m16] = (out, st.double().sum(0), out2, dx, dxp, part.double().sum(0))
        assert labels[True] == ["conv_pipe_kernel<3,2,false,false,false,true>"] * 3 + ["conv_pipe_kernel<3,2,false,false,true,true>"], labels[True]
        assert labels[False] == ["conv_pipe_kernel<3,2,false,false>"] * 3 + ["conv_pipe_kernel<3,2,false,false,true>"], labels[False]
        for i, (u, v) in enumerate(zip(res[False], res[True])):
            tol = (2e-6 if i not in (1, 5) else 2e-5) * float(u.abs().max())       # (statistics: sums over all pixels)
            assert float((u.double() - v.double()).abs().max()) <= tol, (i, float((u.double() - v.double()).abs().max()), tol)
        ref = torch.nn.functional.conv2d(torch.relu(x * sc + sh).permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), padding=1)
        errs = [float((res[k][0].permute(0, 3, 1, 2).double().cpu() - ref).norm() / ref.norm()) for k in (False, True)]
        assert errs[1] < 1e-6 and errs[1] < 1.5 * errs[0] + 1e-8, errs
        gref = torch.nn.grad.conv2d_input((B, Cin, H, Wd), w.double().cpu(), dy.permute(0, 3, 1, 2).double().cpu(), padding=1)
        gerr = [float((res[k][3].permute(0, 3, 1, 2).double().cpu() - dadd.permute(0, 3, 1, 2).double().cpu() - gref).norm() / gref.norm()) for k in (False, True)]
        assert gerr[1] < 1e-6 and gerr[1] < 1.5 * gerr[0] + 1e-8, gerr
    finally:
        ops.SPLIT, ops.PIPE_CONV, ops.WS_CONV, ops.PIPE_M16, ops.PROFILE = old
        tiling.conv_tile = real_tile


def test_bn_apply_sign_mask(ops):
    """spk_bn_apply's optional 1-bit output: bit k of word j of a pixel = (out[pixel][32 j + k] > 0)."""
    torch.manual_seed(1)
    for C, N in ((32, 999), (128, 77), (256, 1031)):
        raw = torch.randn(N, 1, 1, C)
        res = torch.randn(N, 1, 1, C)
        sc, sh = torch.rand(C) + 0.5, torch.randn(C) * 0.3
        out, mk = ops.bn_apply(raw.cuda(), sc.cuda(), sh.cuda(), res=res.cuda(), relu=True, mask=True)
        ref = torch.relu(raw * sc + sh + res)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)
        want = sign_mask(out.cpu().permute(0, 3, 1, 2))
        assert torch.equal(mk, want)


@pytest.mark.parametrize("C,rows", [(32, 97280), (256, 4100), (64, 1025), (96, 1500)])
def test_bn_finalize_many_partial_rows(ops, C, rows):
    """The two-stage fold + finalize path (more than 1024 partial rows: conv-epilogue statistics of the big layers):
    values against fp64 sums, bitwise repeatability, and reuse of the shared fp64 workspace by back-to-back launches
    with different data and different channel counts."""
    rng = np.random.RandomState(C + rows)
    count = float(rows * 37)
    outs = []
    for rep in range(3):
        part = torch.from_numpy((rng.randn(rows, C, 2) * (1.0 + rep)).astype(np.float32))
        part[:, :, 1] = part[:, :, 1].abs() * 40 + 5                      # sum of squares side: keep the variance positive
        gamma = torch.from_numpy(rng.rand(C).astype(np.float32) + 0.5)
        beta = torch.from_numpy(rng.randn(C).astype(np.float32))
        s = part.double().sum(0)
        mean = s[:, 0] / count
        var = (s[:, 1] / count - mean * mean).clamp_min(0)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        for again in range(2):
            bn4 = torch.zeros(4, C, device="cuda")
            rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
            nbt = torch.zeros((), dtype=torch.long, device="cuda")
            ops.bn_finalize(part.cuda(), count, gamma.cuda(), beta.cuda(), rm, rv, nbt, bn4)
            dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
            coef = ops.bn_bwd_coef(part.cuda(), count, gamma.cuda(), bn4, dg, db)
            torch.cuda.synchronize()
            outs.append((rep, bn4.cpu(), rm.cpu(), rv.cpu(), dg.cpu(), db.cpu(), coef.cpu()))
            assert int(nbt) == 1
        a, b = outs[-2], outs[-1]
        for u, v in zip(a[1:], b[1:]):
            assert torch.equal(u, v)
        bn4c = b[1].double()
        np.testing.assert_allclose(bn4c[0].numpy(), mean.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(bn4c[1].numpy(), invstd.numpy(), rtol=2e-6)
        np.testing.assert_allclose(bn4c[2].numpy(), (gamma.double() * invstd).numpy(), rtol=3e-6)
        np.testing.assert_allclose(b[3].numpy(), (0.9 + 0.1 * var * count / (count - 1)).numpy(), rtol=1e-5)
        np.testing.assert_allclose(b[5].numpy(), s[:, 0].numpy(), rtol=1e-6, atol=1e-3)      # dbeta = sum of column 0
        np.testing.assert_allclose(b[4].numpy(), s[:, 1].numpy(), rtol=1e-6)                 # dgamma = sum of column 1
        np.testing.assert_allclose(b[6][1].numpy(), (s[:, 0] / count).numpy(), rtol=1e-6, atol=1e-7)


def test_batched_weight_pack_equals_per_conv_pack(ops, mfma_mode):
    """spk_pack_conv_weights_batched (one launch for the whole network) against the per-convolution exports, both operand
    modes, mixed shapes, forward and transposed orders: bit-identical buffers."""
    torch.manual_seed(3)
    ws = [torch.randn(co, ci, k, k, device="cuda") for co, ci, k in ((32, 32, 3), (64, 32, 3), (64, 32, 1), (128, 64, 3), (256, 256, 3), (256, 128, 1))]
    jobs, refs = [], []
    for w in ws:
        for tr in (False, True):
            buf = torch.full((ops.packed_numel(w),), float("nan"), device="cuda")
            jobs.append((w, buf, tr))
            refs.append(ops.pack_conv_weight(w, tr))
    tab = ops.PackTable(jobs, "cuda")
    tab.run()
    torch.cuda.synchronize()
    for (w, buf, tr), ref in zip(jobs, refs):
        assert torch.equal(buf.view(torch.int32), ref.view(torch.int32)), (tuple(w.shape), tr)


def test_split_operands_are_as_accurate_as_fp32_operands(ops):
    """The bf16-split operand modes against an fp64 convolution on inputs with a wide dynamic range (exp(3 N(0,1))
    magnitudes: 6 decades) and on ReLU-like activations, for forward, data gradient and weight gradient: the rms error
    must stay within 3x of the native fp32 matrix instruction's (measured 0.6x .. 2.3x: either can be ahead) and below
    1e-6 relative - i.e. fp32-class (eps = 6e-8), four orders of magnitude below a bf16 result."""
    torch.manual_seed(7)
    B, C, H, Wd = 2, 64, 12, 19
    for name, x, w in (
            ("wide", torch.randn(B, C, H, Wd) * torch.exp(3 * torch.randn(B, C, H, Wd)),
             torch.randn(C, C, 3, 3) * torch.exp(3 * torch.randn(C, C, 3, 3)) * 0.05),
            ("relu", torch.relu(torch.randn(B, C, H, Wd) * 1.3 + 0.2), torch.randn(C, C, 3, 3) * 0.03)):
        dy = torch.randn(B, C, H, Wd) * torch.exp(torch.randn(B, C, H, Wd))
        x64, w64, dy64 = (t.double().requires_grad_(True) for t in (x, w, dy))
        ref = F.conv2d(x64, w64, None, 1, 1)
        gx, gw = torch.autograd.grad(ref, [x64, w64], grad_outputs=dy64)
        errs = {}
        old = ops.SPLIT
        try:
            for mode, split in ops.MFMA_MODES.items():
                ops.SPLIT = split
                wg = w.cuda()
                out, _ = ops.conv_fwd(nhwc(x), ops.pack_conv_weight(wg), C, 3, 1)
                dx = ops.conv_dgrad(nhwc(dy), ops.pack_conv_weight(wg, transpose=True), C, 3, 1, (H, Wd))
                dw = torch.empty(C, C, 3, 3, device="cuda")
                ops.conv_wgrad(nhwc(x), nhwc(dy), dw, 3, 1)
                torch.cuda.synchronize()
                errs[mode] = tuple(float((a.double() - b.detach()).pow(2).mean().sqrt() / b.detach().pow(2).mean().sqrt())
                                   for a, b in ((nchw(out), ref), (nchw(dx), gx), (dw.cpu(), gw)))
        finally:
            ops.SPLIT = old
        print(name, {k: ["%.2e" % e for e in v] for k, v in errs.items()})
        for mode in ("bf16x6", "bf16x9", "f16x3"):
            for e_split, e_f32 in zip(errs[mode], errs["f32"]):
                assert e_split <= 3.0 * e_f32 + 1e-9 and e_split < 1e-6, (name, mode, errs)
        assert max(errs["f32"]) < 2e-5        # rms error relative to the rms of the exact result


@pytest.mark.parametrize("case", ["tiny", "huge", "outlier", "zeros", "tiny_grad_big_act"])
def test_split_operands_at_extreme_magnitudes(ops, case):
    """Worst cases of the split operand modes - where a term could underflow (bf16's third term, the second fp16 term) or
    overflow (fp16 tops out at 65504): magnitudes near the bottom and the top of the fp32 range, one outlier 10^6 times the
    rest, all-zero tensors, 1e-12 gradients against O(10) activations.  f16x3 rescales every operand tensor by a power of
    two from its absmax, so none of these may cost accuracy: per output element the error must stay within a few fp32
    rounding steps of the exact result's scale, in every mode, for forward, data gradient and weight gradient; and the
    maximum error (not only the rms) is bounded against the native fp32 instruction's."""
    torch.manual_seed(11)
    B, C, H, Wd = 2, 64, 9, 14
    x = torch.relu(torch.randn(B, C, H, Wd) + 0.3)
    w = torch.randn(C, C, 3, 3) * 0.04
    dy = torch.randn(B, C, H, Wd)
    if case == "tiny":
        x, dy = x * 1e-14, dy * 1e-17          # (their products, 1e-31, still sit inside the fp32 range)
    elif case == "huge":
        x, w, dy = x * 1e15, w * 1e3, dy * 1e12
    elif case == "outlier":           # 10^3 x the typical magnitude in every operand: elements down to 1 % of typical still
        x[0, 3, 4, 5] = 1e3           # sit inside f16x3's 2^18 full-precision window below the tensor's absmax
        dy[1, 7, 2, 2] = -1e3
        w[5, 6, 1, 1] = 40.0
    elif case == "zeros":
        x, dy = x * 0, dy * 0
    elif case == "tiny_grad_big_act":
        x, dy = x * 20, dy * 1e-12
    x64, w64, dy64 = (t.double().requires_grad_(True) for t in (x, w, dy))
    ref = F.conv2d(x64, w64, None, 1, 1)
    gx, gw = torch.autograd.grad(ref, [x64, w64], grad_outputs=dy64)
    # |error| is judged against sum |a||b| of each output (what fp32 rounding of the terms is proportional to)
    mag = [F.conv2d(x64.abs(), w64.abs(), None, 1, 1).detach(),
           torch.autograd.grad(F.conv2d(x64, w64.abs(), None, 1, 1), [x64], grad_outputs=dy64.abs())[0],
           torch.autograd.grad(F.conv2d(x64.abs(), w64, None, 1, 1), [w64], grad_outputs=dy64.abs())[0]]
    worst = {}
    old = ops.SPLIT
    try:
        for mode, split in ops.MFMA_MODES.items():
            ops.SPLIT = split
            wg = w.cuda()
            out, _ = ops.conv_fwd(nhwc(x), ops.pack_conv_weight(wg), C, 3, 1)
            dx = ops.conv_dgrad(nhwc(dy), ops.pack_conv_weight(wg, transpose=True), C, 3, 1, (H, Wd))
            dw = torch.empty(C, C, 3, 3, device="cuda")
            ops.conv_wgrad(nhwc(x), nhwc(dy), dw, 3, 1)
            torch.cuda.synchronize()
            res = []
            for got, exact, m in ((nchw(out), ref, mag[0]), (nchw(dx), gx, mag[1]), (dw.cpu(), gw, mag[2])):
                assert torch.isfinite(got).all(), (case, mode)
                scale = float(m.max()) if float(m.max()) > 0 else 1.0
                # per element against its own magnitude sum, with a floor of 1e-6 of the largest one (elements that are
                # sums of nothing but vanishing terms are judged on the tensor's scale)
                rel = (got.double() - exact.detach()).abs() / torch.clamp(m, min=1e-6 * scale)
                res.append(float(rel.max()))
            worst[mode] = res
    finally:
        ops.SPLIT = old
    print(case, {k: ["%.2e" % e for e in v] for k, v in worst.items()})
    for mode, res in worst.items():
        for e, e32 in zip(res, worst["f32"]):
            assert e < 2e-6, (case, mode, worst)                       # ~30 fp32 rounding steps of sum |a||b|, K = 576
            assert e <= 4.0 * e32 + 2e-7, (case, mode, worst)          # and never far from the fp32 instruction itself


def test_f16x3_precision_floor_below_the_scale_window(ops):
    """The documented limit of the f16x3 mode (csrc/spk_common.h): operand scales are per TENSOR, so one element 10^7 times
    the rest pushes the rest below bound * 2^-18, where the second fp16 term leaves the normal range.  The matrix instruction
    KEEPS fp16 subnormals (tools/probe/run_split_probe.py, profiles/r03_split_probe.log - round 2 documented the opposite from
    a probe that read freed memory), so those elements degrade gracefully: absolute error <= 2^-25 / sigma <= amax * 2^-39
    each.  The bf16 modes (fp32's exponent range) have no such floor.  Asserted here: the error of every output <= fp32-class
    relative part + K * amax_x * amax_w * 2^-29 (the bound round 2 stated; the measured floor is three decades lower),
    nothing overflows, and the outputs that involve the outlier itself stay fp32-accurate."""
    torch.manual_seed(12)
    B, C, H, Wd = 1, 64, 9, 14
    x = torch.relu(torch.randn(B, C, H, Wd) + 0.3)
    x[0, 3, 4, 5] = 3e7
    w = torch.randn(C, C, 3, 3) * 0.04
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
    old = ops.SPLIT
    try:
        ops.SPLIT = ops.MFMA_MODES["f16x3"]
        out, _ = ops.conv_fwd(nhwc(x), ops.pack_conv_weight(w.cuda()), C, 3, 1)
    finally:
        ops.SPLIT = old
    got = nchw(out).double()
    assert torch.isfinite(got).all()
    floor = 576 * float(x.abs().max()) * float(w.abs().max()) * 2.0 ** -29
    err = (got - ref).abs()
    assert bool((err <= 2e-6 * mag + floor).all()), float((err - 2e-6 * mag).max() / floor)
    near = mag > 1e3                     # outputs whose sum contains the outlier
    assert bool(near.any()) and float((err[near] / mag[near]).max()) < 2e-6


@pytest.mark.parametrize("shape", [(2, 8, 13), (3, 80, 200), (1, 40, 37)])
def test_stem(ops, shape):
    B, Fd, T = shape
    x = rnd(11, B, Fd, T)
    w = rnd(12, 32, 1, 3, 3, scale=0.5)
    ref = F.conv2d(x.view(B, 1, Fd, T), w, None, 1, 1)
    out, st = ops.stem_fwd(x.cuda(), w.cuda(), stats=True)
    assert relerr(nchw(out), ref) < 1e-6
    tot = st.double().sum(0).cpu()
    np.testing.assert_allclose(tot[:, 0].numpy(), ref.double().sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(tot[:, 1].numpy(), (ref.double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-5)
    esc, esh = rnd(13, 32, scale=0.5, shift=1.0), rnd(14, 32, scale=0.3)
    out2, _ = ops.stem_fwd(x.cuda(), w.cuda(), epi_affine=(esc.cuda(), esh.cuda()), relu=True)
    assert relerr(nchw(out2), F.relu(ref * esc.view(1, -1, 1, 1) + esh.view(1, -1, 1, 1))) < 1e-6
    dy = rnd(15, *ref.shape)
    wr = w.clone().requires_grad_(True)
    gw, = torch.autograd.grad((F.conv2d(x.view(B, 1, Fd, T), wr, None, 1, 1) * dy).sum(), [wr])
    dw = torch.empty(32, 1, 3, 3, device="cuda")
    ops.stem_wgrad(x.cuda(), nhwc(dy), dw)
    assert relerr(dw.cpu(), gw) < 1e-5


@pytest.mark.parametrize("C,N", [(32, 1000), (64, 777), (256, 64), (128, 5000)])
def test_batchnorm_train_fwd_bwd(ops, C, N):
    x = rnd(21, N, C, scale=2.0, shift=0.5)
    gamma, beta = rnd(22, C, scale=0.3, shift=1.0), rnd(23, C, scale=0.2)
    rm, rv = rnd(24, C, scale=0.1), rnd(25, C, scale=0.2, shift=1.0)
    res = rnd(26, N, C)
    dy = rnd(27, N, C)
    # oracle: F.batch_norm train mode on [N,C] + residual + relu, autograd for the backward
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.relu(F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5) + res)
    gx, gg, gb = torch.autograd.grad((y * dy).sum(), [xr, gr, br])
    xg = x.cuda()
    part = ops.bn_stats_partial(xg)
    bn4 = torch.empty(4, C, device="cuda")
    rmg, rvg = rm.cuda(), rv.cuda()
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    ops.bn_finalize(part, N, gamma.cuda(), beta.cuda(), rmg, rvg, nbt, bn4)
    out = ops.bn_apply(xg, bn4[2], bn4[3], res=res.cuda(), relu=True)
    assert relerr(out.cpu(), y.detach()) < 2e-6
    assert relerr(rmg.cpu(), rm_ref) < 1e-6 and relerr(rvg.cpu(), rv_ref) < 1e-6
    assert int(nbt) == 1
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dz = torch.empty(N, C, device="cuda")
    draw = ops.bn_backward(dy.cuda(), xg, out, bn4, gamma.cuda(), dgam, dbet, ops.MASK_ACT, dz_out=dz)
    assert relerr(draw.cpu(), gx) < 2e-5
    assert relerr(dgam.cpu(), gg) < 2e-5 and relerr(dbet.cpu(), gb) < 2e-5
    assert relerr(dz.cpu(), dy * (y.detach() > 0)) == 0.0
    # MASK_RAW: relu directly after BN, mask recomputed from raw
    y2 = F.relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5))
    gx2, = torch.autograd.grad((y2 * dy).sum(), [xr])
    draw2 = ops.bn_backward(dy.cuda(), xg, None, bn4, gamma.cuda(), dgam, dbet, ops.MASK_RAW)
    assert relerr(draw2.cpu(), gx2) < 2e-5
    # MASK_NONE: no relu (downsample BN)
    y3 = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    gx3, = torch.autograd.grad((y3 * dy).sum(), [xr])
    draw3 = ops.bn_backward(dy.cuda(), xg, None, bn4, gamma.cuda(), dgam, dbet, ops.MASK_NONE)
    assert relerr(draw3.cpu(), gx3) < 2e-5
    # eval coefficients
    ev = torch.empty(2, C, device="cuda")
    ops.bn_eval_coeffs(gamma.cuda(), beta.cuda(), rmg, rvg, ev)
    y4 = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, False, 0.1, 1e-5)
    out4 = ops.bn_apply(xg, ev[0], ev[1], relu=False)
    assert relerr(out4.cpu(), y4) < 2e-6


def test_stats_pool_golden_and_random(ops, gold_dir):
    g = np.load(os.path.join(gold_dir, "kernels.npz"))
    x = torch.from_numpy(g["pool_x"])            # NCHW [2,4,3,13]
    for mode, name in [(0, "mean"), (1, "mean+std")]:
        y = ops.stats_pool_fwd(nhwc(x), mode)
        ref = g["pool_%s_y" % name].reshape(2, -1)
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=2e-6, atol=1e-7)
        gy = torch.from_numpy(g["pool_%s_gy" % name].reshape(2, -1)).cuda()
        slot = torch.zeros(1, device="cuda", dtype=torch.int32)
        dx = ops.stats_pool_bwd(nhwc(x), gy, mode, amax_out=slot)
        np.testing.assert_allclose(nchw(dx).numpy(), g["pool_%s_gx" % name], rtol=1e-5, atol=1e-7)
        assert float(slot.cpu().view(torch.float32)[0]) == float(dx.abs().max())      # absmax hand-off of the f16x3 mode
    xr = rnd(31, 3, 256, 10, 38, scale=0.5, shift=0.6)
    yo = O.stats_pool(xr, "mean+std").flatten(1)
    y = ops.stats_pool_fwd(nhwc(xr), 1)
    assert relerr(y.cpu(), yo) < 2e-6
    slot = torch.zeros(1, device="cuda", dtype=torch.int32)              # a grid that does not fill its last wave
    dx = ops.stats_pool_bwd(nhwc(xr), rnd(32, 3, 256 * 10 * 2).cuda(), 1, amax_out=slot)
    assert float(slot.cpu().view(torch.float32)[0]) == float(dx.abs().max())
    # a dead row (all zeros over time: mean 0): sqrt'(0) gives inf / NaN there exactly as torch.sqrt's backward does in the
    # reference (scripts/model.py:453: the ReLU mask select drops them one step later); the absmax hand-off leaves them out
    xz = xr.clone()
    xz[1, 7, 3, :] = 0.0
    xz[2, 100, 0, :] = 0.0
    gy = rnd(33, 3, 256 * 10 * 2)
    gy[2, 100 * 20 + 10 + 0] = 0.0                  # 0 / 0 -> NaN in that row, g / 0 -> inf in the other
    xt = xz.clone().requires_grad_(True)
    gref, = torch.autograd.grad(O.stats_pool(xt, "mean+std").flatten(1), [xt], grad_outputs=gy)
    slot = torch.zeros(1, device="cuda", dtype=torch.int32)
    dx = nchw(ops.stats_pool_bwd(nhwc(xz), gy.cuda(), 1, amax_out=slot))
    bad = ~torch.isfinite(gref)
    assert int(bad.sum()) == 2 * 38 and torch.equal(bad, ~torch.isfinite(dx))
    np.testing.assert_allclose(dx[~bad].numpy(), gref[~bad].numpy(), rtol=2e-5, atol=1e-7)
    assert float(slot.cpu().view(torch.float32)[0]) == float(dx[~bad].abs().max())


@pytest.mark.parametrize("M,N,K", [(256, 256, 5120), (6, 11, 256), (256, 1211, 256), (37, 70, 129)])
def test_gemm_and_linear(ops, M, N, K):
    x, w, b = rnd(41, M, K), rnd(42, N, K, scale=0.1), rnd(43, N)
    ref = F.linear(x, w, b)
    out = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda())
    assert relerr(out.cpu(), ref) < 1e-5
    dy = rnd(44, M, N)
    dw, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
    dx = ops.linear_bwd(x.cuda(), w.cuda(), dy.cuda(), dw, db)
    assert relerr(dx.cpu(), dy @ w) < 1e-5
    assert relerr(dw.cpu(), dy.t() @ x) < 1e-5
    assert relerr(db.cpu(), dy.sum(0)) < 1e-5


def test_aam_head_golden(ops, gold_dir):
    g = np.load(os.path.join(gold_dir, "kernels.npz"))
    e, w, lab = torch.from_numpy(g["aam_e"]).cuda(), torch.from_numpy(g["aam_w"]).cuda(), torch.from_numpy(g["aam_lab"]).cuda()
    B, D = e.shape
    S = w.shape[0]
    en, einv = ops.l2norm_fwd(e)
    wn, winv = ops.l2norm_fwd(w)
    cosv = ops.gemm(en, wn, B, S, D, D, 1, 1, D)
    logits = ops.aam_margin_fwd(cosv, lab, 0.2, 30.0)
    # row 2 is engineered with cos ~ 0.999 on its label (sine small): that logit is conditioned ~cos/sine worse
    np.testing.assert_allclose(logits.cpu().numpy(), g["aam_logits"], rtol=2e-5, atol=5e-5)
    loss_row, dl, rank = ops.softmax_ce(logits, lab, grad_scale=1.0 / B)
    assert abs(float(ops.mean(loss_row)) - float(g["aam_loss"])) < 1e-4
    dcos = ops.aam_margin_bwd(cosv, lab, dl, 0.2, 30.0)
    den = ops.gemm(dcos, wn, B, D, S, S, 1, D, 1)
    dwn = ops.gemm(dcos, en, S, D, B, 1, S, D, 1)
    de = ops.l2norm_bwd(en, einv, den)
    dw = ops.l2norm_bwd(wn, winv, dwn)
    assert relerr(de.cpu(), torch.from_numpy(g["aam_ge"])) < 1e-4
    assert relerr(dw.cpu(), torch.from_numpy(g["aam_gw"])) < 1e-4
    # rank vs the oracle's top-k
    lg = torch.from_numpy(g["aam_logits"])
    acc1, acc5 = O.accuracy(lg, torch.from_numpy(g["aam_lab"]), (1, 5))
    r = rank.cpu()
    assert abs(float((r < 1).float().mean() * 100) - float(acc1)) < 1e-4
    assert abs(float((r < 5).float().mean() * 100) - float(acc5)) < 1e-4


def test_softmax_ce_large(ops):
    B, S = 64, 5994
    lg = rnd(51, B, S, scale=12.0)
    y = torch.from_numpy((W.hash_uniform(52, 1, B) * S).astype(np.int64))
    lr = lg.clone().requires_grad_(True)
    loss = O.cross_entropy(lr, y)
    g, = torch.autograd.grad(loss, [lr])
    loss_row, dl, rank = ops.softmax_ce(lg.cuda(), y.cuda(), grad_scale=1.0 / B)
    assert abs(float(ops.mean(loss_row)) - float(loss)) < 1e-5 * abs(float(loss))
    assert relerr(dl.cpu(), g) < 1e-5


def test_sgd_matches_oracle(ops):
    n = 10007
    p0, g1, g2 = rnd(61, n), rnd(62, n), rnd(63, n)
    st = {"p": p0.clone()}
    bufs = {}
    O.sgd_step(st, {"p": g1}, bufs, 0.1, 0.9, 5e-4)
    O.sgd_step(st, {"p": g2}, bufs, 0.05, 0.9, 5e-4)
    n_pad = (n + 3) // 4 * 4
    p = torch.zeros(n_pad, device="cuda")[:n]
    p.copy_(p0)
    buf = torch.zeros(n, device="cuda")
    ops.sgd_step(p, g1.cuda(), buf, 0.1, 0.9, 5e-4, 1.0, True)
    ops.sgd_step(p, g2.cuda(), buf, 0.05, 0.9, 5e-4, 1.0, False)
    assert relerr(p.cpu(), st["p"]) < 1e-6
    assert relerr(buf.cpu(), bufs["p"]) < 1e-6


def test_scoring_backend_on_device(ops, gold_dir, tmp_path):
    """csrc/score.hip (normalise, per-trial dot, cohort top-k statistics) against the reference's own score files and
    against the host back end on a larger random problem."""
    from pytorch_kaldi_resnet_amd import kaldi_io, scoring
    d = os.path.join(gold_dir, "io")
    mean = kaldi_io.read_vec_flt(os.path.join(d, "mean.vec"))
    emb = scoring.read_embeddings(os.path.join(d, "emb.iv"))
    coh = scoring.read_embeddings(os.path.join(d, "cohort.iv"))
    sc, lab = scoring.cosine_score(emb, emb, os.path.join(d, "trials"), mean, str(tmp_path / "scores"), backend="hip")
    ref = [float(l.split()[2]) for l in open(os.path.join(d, "scores"))]
    np.testing.assert_allclose(sc, ref, rtol=1e-6, atol=3e-7)
    assert "{0:.2%}".format(scoring.compute_eer(sc, lab)) == open(os.path.join(d, "eer.txt")).read().strip()
    st = scoring.topk_mean_std(emb, coh, mean, 300, backend="hip")
    gold = scoring.read_mean_std(os.path.join(d, "topk_mean_std"))
    for k in gold:
        np.testing.assert_allclose(st[k], gold[k], rtol=5e-6, atol=2e-7)
    # model-sized problem: 256-dim embeddings, 3000 utterances, 1500-vector cohort, 20k trials (two distinct tables)
    rng = np.random.RandomState(5)
    spk = rng.randn(60, 256).astype(np.float32)
    en = {"e%d" % i: spk[i % 60] + 0.8 * rng.randn(256).astype(np.float32) for i in range(3000)}
    te = {"t%d" % i: spk[i % 60] + 0.8 * rng.randn(256).astype(np.float32) for i in range(2000)}
    co = {"c%d" % i: rng.randn(256).astype(np.float32) + 0.3 * spk[i % 60] for i in range(1500)}
    m = rng.randn(256).astype(np.float32) * 0.1
    tp = str(tmp_path / "trials")
    with open(tp, "w") as f:
        for _ in range(20000):
            a, b = rng.randint(3000), rng.randint(2000)
            f.write("e%d t%d %s\n" % (a, b, "target" if a % 60 == b % 60 else "nontarget"))
    s_h, l_h = scoring.cosine_score(en, te, tp, m, backend="host")
    s_d, l_d = scoring.cosine_score(en, te, tp, m, backend="hip")
    np.testing.assert_allclose(s_d, s_h, rtol=0, atol=5e-7)
    assert abs(scoring.compute_eer(s_d, l_d) - scoring.compute_eer(s_h, l_h)) < 1e-3
    k_h = scoring.topk_mean_std(te, co, m, 300, backend="host")
    k_d = scoring.topk_mean_std(te, co, m, 300, backend="hip")
    for k in k_h:
        np.testing.assert_allclose(k_d[k], k_h[k], rtol=2e-5, atol=5e-7)
    # k == M, non power-of-two M, duplicate scores
    x = torch.tensor([[3.0, 1.0, 2.0, 2.0, -1.0], [0.5, 0.5, 0.5, 0.5, 0.5]], device="cuda")
    mu, sd = ops.topk_mean_std(x, 5)
    np.testing.assert_allclose(mu.cpu().numpy(), [1.4, 0.5], rtol=1e-6)
    np.testing.assert_allclose(sd.cpu().numpy(), [np.std([3, 1, 2, 2, -1], ddof=1), 0.0], rtol=1e-6, atol=1e-7)
    mu, sd = ops.topk_mean_std(x, 2)
    np.testing.assert_allclose(mu.cpu().numpy(), [2.5, 0.5], rtol=1e-6)
    with pytest.raises(RuntimeError):
        ops.topk_mean_std(x, 6)


@pytest.mark.parametrize("shape,tile", [
    ((3, 64, 64, 40, 150, 1), (8, 8)),        # the step's own shapes and tiles (tile_table.json) ...
    ((3, 128, 128, 20, 75, 1), (4, 16)),
    ((2, 256, 256, 10, 38, 1), (10, 6)),      # ... 60 pixels = 1.9 k-steps: the padding pixels fetch the zero block
    ((2, 64, 128, 40, 150, 2), (3, 6)),       # stride 2 (halo 7 x 13): 18 pixels, one k-step, almost half of it padding
    ((2, 64, 128, 13, 29, 1), (2, 24)),       # ragged right / bottom edges
    ((2, 128, 64, 3, 5, 1), (4, 16)),         # an image smaller than one tile
    ((5, 64, 64, 9, 33, 1), (3, 10)),         # tile width not a multiple of 4: the four pixels of a DMA chunk straddle tile rows
])
def test_weight_gradient_16x16x32_with_dy_by_lds_dma(ops, shape, tile):
    """conv_wgrad_wm16_kernel (csrc/conv_wgrad_wm16.hip; the default for the 3x3 weight gradients with Cin, Cout multiples of 64 whose dY
    is an f16 pair tensor): the 2 x 2 wave layout on v_mfma_f32_16x16x32_f16, dY brought into a swizzled, double-buffered LDS image by
    global_load_lds.  Same products as conv_wgrad_wm_kernel in another summation order: equal within fp32 accumulation error (2e-6 of
    the largest element), as close to the fp64 gradient, with and without the fused BatchNorm + ReLU on X and with accumulation.
    Reference semantics: autograd of nn.Conv2d (scripts/model.py:41-44 under scripts/train_resnet.py:327 loss.backward())."""
    from helpers import encode_pairs, sigma_of, slot
    from pytorch_kaldi_resnet_amd import tiling
    B, Cin, Cout, H, Wd, stride = shape
    OH, OW = ops.conv_out_hw(H, Wd, 3, stride)
    key = (OH, OW, Cin, Cout, 3, stride)
    old_force, old_wm16, old_split = tiling.FORCE_WGRAD_SPLIT.get(key), ops.WM16, ops.SPLIT
    tiling.FORCE_WGRAD_SPLIT[key] = (tile[0], tile[1], 2)
    ops.SPLIT = ops.MFMA_MODES["f16x3"]
    try:
        x = rnd(31, B, H, Wd, Cin).cuda()
        dy = (rnd(32, B, OH, OW, Cout, scale=3e-4) * (1 + 50 * (rnd(33, B, OH, OW, 1) > 0.97))).cuda()
        sc, sh = rnd(34, Cin, scale=0.4, shift=1.0).cuda(), rnd(35, Cin, scale=0.3).cuda()
        dy_amax = ops.absmax_into(dy, slot())
        x_amax = ops.absmax_into(x, slot())
        xa_amax = ops.absmax_into(torch.relu(x * sc + sh), slot())
        dy_p = encode_pairs(dy.cpu(), sigma_of(dy_amax)).cuda()
        res = {}
        for m16 in (True, False):
            ops.WM16 = m16
            dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
            ops.conv_wgrad(x, dy_p, dw, 3, stride, dy_amax=dy_amax, x_amax=x_amax, dy_presplit=True)
            dw2 = torch.full((Cout, Cin, 3, 3), 0.25, device="cuda")
            ops.conv_wgrad(x, dy_p, dw2, 3, stride, in_affine=(sc, sh), accumulate=True, dy_amax=dy_amax, x_amax=xa_amax, dy_presplit=True)
            dw3 = torch.empty(Cout, Cin, 3, 3, device="cuda")       # a second launch: both LDS buffers and the zero block reused
            ops.conv_wgrad(x, dy_p, dw3, 3, stride, dy_amax=dy_amax, x_amax=x_amax, dy_presplit=True)
            res[m16] = (dw, dw2, dw3)
        torch.cuda.synchronize()
        assert torch.equal(res[True][0], res[True][2]), "not deterministic"
        xc, dc = x.cpu().permute(0, 3, 1, 2).double(), dy.cpu().permute(0, 3, 1, 2).double()
        ref = torch.nn.grad.conv2d_weight(xc, (Cout, Cin, 3, 3), dc, stride=stride, padding=1)
        xa = torch.relu(xc * sc.cpu().double().view(1, -1, 1, 1) + sh.cpu().double().view(1, -1, 1, 1))
        ref2 = torch.nn.grad.conv2d_weight(xa, (Cout, Cin, 3, 3), dc, stride=stride, padding=1)
        e16, e32 = [(res[m][0].double().cpu() - ref).norm() / ref.norm() for m in (True, False)]
        print("shape %s tile %s: |dw - fp64| / |fp64| = %.2e (16x16x32, dY by DMA)  %.2e (32x32x16)" % (shape, tile, e16, e32))
        assert e16 < 1e-5 and e16 < 2 * e32 + 1e-7, (e16, e32)
        assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-6 * float(ref.abs().max())
        d2 = (res[True][1].double().cpu() - 0.25) - ref2
        assert d2.norm() <= 1e-5 * ref2.norm() + 1.5e-8 * ref2.numel() ** 0.5
        assert float((res[True][1] - res[False][1]).abs().max()) <= 2e-6 * float(ref2.abs().max()) + 3e-8
    finally:
        ops.WM16, ops.SPLIT = old_wm16, old_split
        if old_force is None:
            tiling.FORCE_WGRAD_SPLIT.pop(key, None)
        else:
            tiling.FORCE_WGRAD_SPLIT[key] = old_force


@pytest.mark.parametrize("shape,tile", [
    ((3, 32, 32, 80, 300, 1), None),          # the first layer's shape with the tile its rule picks (8 x 16: four k-steps, one per wave)
    ((2, 32, 32, 19, 45, 1), None),           # the rule's tile on an odd image (10 x 12: ragged edges, DMA chunks straddling tile rows)
    ((2, 32, 32, 19, 45, 1), (14, 8)),        # the general kernel's tile: 112 pixels = 3.5 k-steps
    ((2, 64, 32, 13, 29, 1), (2, 24)),        # two input-channel groups, 48 pixels: two waves have no k-step
    ((2, 32, 32, 3, 5, 1), (4, 16)),          # an image smaller than one tile
    ((2, 32, 32, 40, 150, 2), (4, 10)),       # stride 2
])
def test_weight_gradient_16x16x32_32_channel_groups(ops, shape, tile):
    """conv_wgrad_wm16_kernel<., C32> (csrc/conv_wgrad_wm16.hip): the layout for 32-channel groups - the first layer's weight gradients
    whenever dY is an f16 pair tensor.  Every wave holds the whole 9 x 32 x 32 tile, the four waves split the 32-pixel k-steps of a
    region and write four slabs per block; dY by LDS DMA in chunks of 8 pixels.  Against conv_wgrad_split_kernel (SPK_C32M16=0) and the
    fp64 gradient, with and without the fused BatchNorm + ReLU on X and with accumulation (reference: scripts/model.py:41-44 nn.Conv2d
    under scripts/train_resnet.py:327)."""
    from helpers import encode_pairs, sigma_of, slot
    from pytorch_kaldi_resnet_amd import tiling
    B, Cin, Cout, H, Wd, stride = shape
    OH, OW = ops.conv_out_hw(H, Wd, 3, stride)
    key = (OH, OW, Cin, Cout, 3, stride)
    old_c32, old_split = ops.C32M16, ops.SPLIT
    if tile is not None:
        tiling.FORCE_WGRAD_C32M16[key] = (tile[0], tile[1], 1)
    ops.SPLIT = ops.MFMA_MODES["f16x3"]
    try:
        x = rnd(41, B, H, Wd, Cin).cuda()
        dy = (rnd(42, B, OH, OW, Cout, scale=3e-4) * (1 + 50 * (rnd(43, B, OH, OW, 1) > 0.97))).cuda()
        sc, sh = rnd(44, Cin, scale=0.4, shift=1.0).cuda(), rnd(45, Cin, scale=0.3).cuda()
        dy_amax = ops.absmax_into(dy, slot())
        x_amax = ops.absmax_into(x, slot())
        xa_amax = ops.absmax_into(torch.relu(x * sc + sh), slot())
        dy_p = encode_pairs(dy.cpu(), sigma_of(dy_amax)).cuda()
        res = {}
        for m16 in (True, False):
            ops.C32M16 = m16
            dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
            ops.conv_wgrad(x, dy_p, dw, 3, stride, dy_amax=dy_amax, x_amax=x_amax, dy_presplit=True)
            dw2 = torch.full((Cout, Cin, 3, 3), 0.25, device="cuda")
            ops.conv_wgrad(x, dy_p, dw2, 3, stride, in_affine=(sc, sh), accumulate=True, dy_amax=dy_amax, x_amax=xa_amax, dy_presplit=True)
            dw3 = torch.empty(Cout, Cin, 3, 3, device="cuda")
            ops.conv_wgrad(x, dy_p, dw3, 3, stride, dy_amax=dy_amax, x_amax=x_amax, dy_presplit=True)
            res[m16] = (dw, dw2, dw3)
        torch.cuda.synchronize()
        assert torch.equal(res[True][0], res[True][2]), "not deterministic"
        xc, dc = x.cpu().permute(0, 3, 1, 2).double(), dy.cpu().permute(0, 3, 1, 2).double()
        ref = torch.nn.grad.conv2d_weight(xc, (Cout, Cin, 3, 3), dc, stride=stride, padding=1)
        xa = torch.relu(xc * sc.cpu().double().view(1, -1, 1, 1) + sh.cpu().double().view(1, -1, 1, 1))
        ref2 = torch.nn.grad.conv2d_weight(xa, (Cout, Cin, 3, 3), dc, stride=stride, padding=1)
        e16, e32 = [(res[m][0].double().cpu() - ref).norm() / ref.norm() for m in (True, False)]
        print("shape %s tile %s: |dw - fp64| / |fp64| = %.2e (16x16x32, 32-channel groups)  %.2e (general kernel)" % (shape, tile, e16, e32))
        assert e16 < 1e-5 and e16 < 2 * e32 + 1e-7, (e16, e32)
        assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-6 * float(ref.abs().max())
        d2 = (res[True][1].double().cpu() - 0.25) - ref2
        assert d2.norm() <= 1e-5 * ref2.norm() + 1.5e-8 * ref2.numel() ** 0.5
        assert float((res[True][1] - res[False][1]).abs().max()) <= 2e-6 * float(ref2.abs().max()) + 3e-8
    finally:
        ops.C32M16, ops.SPLIT = old_c32, old_split
        tiling.FORCE_WGRAD_C32M16.pop(key, None)


@pytest.mark.parametrize("shape,tile", [
    ((3, 64, 64, 40, 150), (8, 8)),        # the step's own tiles (tile_table.json) ...
    ((3, 128, 128, 20, 75), (4, 16)),
    ((2, 256, 256, 10, 38), (5, 8)),       # ... 40 pixels = 2.5 k-steps: the padded half-step multiplies zeros of dY
    ((2, 64, 128, 13, 29), (2, 24)),       # ragged right / bottom edges, three half-steps per tile row
    ((2, 128, 64, 3, 5), (4, 16)),         # an image smaller than one tile
    ((1, 64, 64, 9, 33), (1, 32)),         # one-row tiles: the third window read of the last halo row reaches past the X image in LDS
])
def test_shifted_window_weight_gradient_is_bit_identical_to_the_plain_k_loop(ops, shape, tile):
    """conv_wgrad_wm_kernel<VAR, SH> (csrc/conv_wgrad_wm.hip): at stride 1 with TW % 8 == 0 the fragments of the three taps of a filter
    row come from one 10-pixel window per lane (22 instead of 40 transposed LDS reads per k-step, the middle tap built in registers).
    Measured +-0 inside the training step: compiled only with SPK_EXPERIMENTAL.
    Same products in the same order as the plain K loop (SPK_WGRAD_NOSHIFT) on the same tile: the weight gradients are bit-identical,
    with and without the fused BatchNorm + ReLU on X, and they are the gradients autograd gives (reference: scripts/model.py:41-44
    nn.Conv2d under scripts/train_resnet.py:327 loss.backward())."""
    needs_experimental()
    from helpers import encode_pairs, sigma_of, slot
    from pytorch_kaldi_resnet_amd import tiling
    B, Cin, Cout, H, Wd = shape
    key = (H, Wd, Cin, Cout, 3, 1)
    old_force, old_shift, old_split = tiling.FORCE_WGRAD_SPLIT.get(key), ops.WM_SHIFT, ops.SPLIT
    tiling.FORCE_WGRAD_SPLIT[key] = (tile[0], tile[1], 2)
    ops.SPLIT = ops.MFMA_MODES["f16x3"]
    try:
        x = rnd(21, B, H, Wd, Cin).cuda()
        dy = (rnd(22, B, H, Wd, Cout, scale=3e-4) * (1 + 50 * (rnd(23, B, H, Wd, 1) > 0.97))).cuda()
        sc, sh = rnd(24, Cin, scale=0.4, shift=1.0).cuda(), rnd(25, Cin, scale=0.3).cuda()
        dy_amax = ops.absmax_into(dy, slot())
        x_amax = ops.absmax_into(x, slot())
        xa_amax = ops.absmax_into(torch.relu(x * sc + sh), slot())
        dy_p = encode_pairs(dy.cpu(), sigma_of(dy_amax)).cuda()
        res = {}
        for shift in (True, False):
            ops.WM_SHIFT = shift
            dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
            ops.conv_wgrad(x, dy_p, dw, 3, 1, dy_amax=dy_amax, x_amax=x_amax, dy_presplit=True)
            dw2 = torch.full((Cout, Cin, 3, 3), 0.25, device="cuda")
            ops.conv_wgrad(x, dy_p, dw2, 3, 1, in_affine=(sc, sh), accumulate=True, dy_amax=dy_amax, x_amax=xa_amax, dy_presplit=True)
            res[shift] = (dw, dw2)
        torch.cuda.synchronize()
        assert torch.equal(res[True][0], res[False][0]), "plain X"
        assert torch.equal(res[True][1], res[False][1]), "fused BatchNorm + ReLU on X"
        xc, dc = x.cpu().permute(0, 3, 1, 2).double(), dy.cpu().permute(0, 3, 1, 2).double()
        ref = torch.nn.grad.conv2d_weight(xc, (Cout, Cin, 3, 3), dc, stride=1, padding=1)
        assert (res[True][0].double().cpu() - ref).norm() / ref.norm() < 1e-5
        xa = torch.relu(xc * sc.cpu().double().view(1, -1, 1, 1) + sh.cpu().double().view(1, -1, 1, 1))
        ref2 = torch.nn.grad.conv2d_weight(xa, (Cout, Cin, 3, 3), dc, stride=1, padding=1)
        # accumulated onto 0.25: every element also carries the rounding of that sum (half an ulp of 0.25 = 1.5e-8)
        d2 = (res[True][1].double().cpu() - 0.25) - ref2
        assert d2.norm() <= 1e-5 * ref2.norm() + 1.5e-8 * ref2.numel() ** 0.5
    finally:
        ops.WM_SHIFT, ops.SPLIT = old_shift, old_split
        if old_force is None:
            tiling.FORCE_WGRAD_SPLIT.pop(key, None)
        else:
            tiling.FORCE_WGRAD_SPLIT[key] = old_force


def test_experimental_kernel_forms_in_their_variant_library():
    """The producer / consumer convolution and weight gradient, the in-wave pipelined weight gradient and the in-wave pipelined
    fused-BatchNorm-backward data gradient were measured and did not pay; they are compiled only with
    SPK_EXPERIMENTAL=1.  __graft_entry__.build() builds that variant library next to the product one; their bit-identity tests
    (skipped above when the loaded library lacks them) run here in a child process that loads it through SPK_LIB."""
    import subprocess
    import sys
    from pytorch_kaldi_resnet_amd import hip
    if hip.has_experimental():
        pytest.skip("the loaded library already contains the experimental forms: their tests ran above")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "pytorch-kaldi-resnet_amd", "variants", "libspkhip_exp.so")
    if not os.path.exists(lib):
        pytest.skip("variants/libspkhip_exp.so not built (python __graft_entry__.py build)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k",
                        "wave_specialised or pipelined_conv or shifted_window"], env=dict(os.environ, SPK_LIB=lib), capture_output=True, text=True,
                       timeout=900)
    tail = r.stdout[-1500:] + r.stderr[-500:]
    assert r.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 36, tail                  # 28 + 5 + 6 + 6 cases; a few layouts skip themselves by design



@pytest.mark.parametrize("C,B,H,Wd", [(128, 5, 20, 75), (64, 3, 33, 41), (32, 2, 47, 53), (128, 1, 3, 5)])
def test_streaming_1x1_kernel_equals_the_general_kernel(ops, C, B, H, Wd):
    """conv1x1_stream_kernel (persistent blocks, weights in registers, every pixel staged once, stores from the accumulator layout)
    against conv_mfma_kernel on the same launches - the 1x1 convolutions of the Bottleneck blocks (reference scripts/model.py:
    104-110,118-126), forward and data gradient with every epilogue the engine uses there.  Same operands (two fp16 terms under
    the same slot), same products; the accumulation order inside the matrix instruction chain differs (K in one pass here), so
    outputs agree to fp32 accumulation error (2e-6 of the output range), not bit for bit; the statistics partials sum to the same
    totals (1e-5: fp32 partial rows of different lengths).  Grids of 1, 7 and the default number of persistent blocks."""
    if ops.SPLIT != 3:
        pytest.skip("the streaming kernel exists for the f16x3 operand mode")
    x = rnd(31, B, C, H, Wd, scale=1.5, shift=0.2)
    w = rnd(32, C, C, 1, 1, scale=0.2)
    xg = nhwc(x)
    wpk, wpk_t = ops.pack_conv_weight(w.cuda()), ops.pack_conv_weight(w.cuda(), transpose=True)
    isc, ish = rnd(33, C, scale=0.5, shift=1.0).cuda(), rnd(34, C, scale=0.3).cuda()
    dy = nhwc(rnd(35, B, C, H, Wd, scale=1e-3))
    addt = nhwc(rnd(36, B, C, H, Wd, scale=1e-3))
    gate = rnd(37, B, C, H, Wd)
    rawt = nhwc(rnd(38, B, C, H, Wd, scale=2.0, shift=0.3))
    bn4 = torch.stack([rnd(39, C, scale=0.3), rnd(40, C, scale=0.2, shift=1.0), rnd(41, C, scale=0.5, shift=1.0), rnd(42, C, scale=0.4)]).cuda()
    # an f16 pair tensor of dy under a slot holding a bound of it (what bn_bwd_apply(pair_scale=) produces)
    slot = ops.absmax_into(dy, torch.zeros(1, device="cuda", dtype=torch.int32))
    amax = float(slot.view(torch.float32))
    sig = 2.0 ** (14 - int(np.floor(np.log2(amax))))
    v = dy.cpu().float() * sig
    hi = v.half()
    lo = (v - hi.float()).half()
    pairs = torch.stack([hi.view(-1, 4), lo.view(-1, 4)], dim=1).contiguous().view(torch.float32).view(dy.shape).cuda()

    def run(stream, blocks):
        old = ops.STREAM_1X1, ops.STREAM_1X1_BLOCKS
        ops.STREAM_1X1, ops.STREAM_1X1_BLOCKS = stream, blocks
        try:
            r = {}
            r["fwd"], r["fwd_st"] = ops.conv_fwd(xg, wpk, C, 1, 1, stats=True)
            r["aff"], r["aff_st"] = ops.conv_fwd(xg, wpk, C, 1, 1, in_affine=(isc, ish), stats=True)
            r["dx"] = ops.conv_dgrad(dy, wpk_t, C, 1, 1, (H, Wd), in_amax=slot)
            r["dx_pair"] = ops.conv_dgrad(pairs, wpk_t, C, 1, 1, (H, Wd), in_amax=slot, in_presplit=True)
            r["dx_add"] = ops.conv_dgrad(pairs, wpk_t, C, 1, 1, (H, Wd), add=addt, add_mask=sign_mask(gate), in_amax=slot, in_presplit=True)
            r["dx_bnb"], r["bnb_st"] = ops.conv_dgrad(pairs, wpk_t, C, 1, 1, (H, Wd), add=addt, add_mask=sign_mask(gate),
                                                      bn_bwd=(rawt, None, bn4, sign_mask(gate)), in_amax=slot, in_presplit=True)
            r["dx_bnr"], r["bnr_st"] = ops.conv_dgrad(pairs, wpk_t, C, 1, 1, (H, Wd), bn_bwd=(rawt, None, bn4), in_amax=slot, in_presplit=True)
            acc = addt.clone()
            ops.conv_dgrad(pairs, wpk_t, C, 1, 1, (H, Wd), out=acc, accumulate=True, in_amax=slot, in_presplit=True)
            r["dx_acc"] = acc
            amx = torch.zeros(1, device="cuda", dtype=torch.int32)
            ops.conv_dgrad(pairs, wpk_t, C, 1, 1, (H, Wd), in_amax=slot, in_presplit=True, out_amax=amx)
            r["amax"] = amx.view(torch.float32).clone()
            torch.cuda.synchronize()
            return r
        finally:
            ops.STREAM_1X1, ops.STREAM_1X1_BLOCKS = old

    ref = run(False, 512)
    assert torch.equal(ref["dx"], ref["dx_pair"])                     # (the pair tensor IS the staged form of dy)
    for blocks in (1, 7, 512):
        got = run(True, blocks)
        for k in ("fwd", "aff", "dx", "dx_pair", "dx_add", "dx_bnb", "dx_bnr", "dx_acc"):
            scale = float(ref[k].abs().max())
            assert float((got[k] - ref[k]).abs().max()) <= 2e-6 * scale, (k, blocks)
        assert torch.equal(got["dx"], got["dx_pair"])
        assert abs(float(got["amax"]) - float(ref["amax"])) <= 2e-6 * float(ref["amax"])
        for k in ("fwd_st", "aff_st", "bnb_st", "bnr_st"):
            a, b = got[k].double().sum(0), ref[k].double().sum(0)
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (k, blocks)
    # and against the definition (torch CPU fp32)
    refc = F.conv2d(x, w)
    assert relerr(nchw(run(True, 512)["fwd"]), refc) < 2e-5


@pytest.mark.parametrize("B,H,Wd", [(3, 80, 300), (2, 13, 45), (2, 5, 9), (1, 8, 16), (4, 80, 203)])
def test_streaming_3x3_forward_of_the_32_channel_layer_equals_the_general_kernel(ops, B, H, Wd):
    """conv3x3_c32_stream_kernel (persistent blocks, weights in LDS, halo staged one tile ahead from whole lines, stores from the
    accumulator layout) against conv_mfma_kernel on the forward convolutions of layer 1 (reference scripts/model.py:48-64), plain
    input and fused BatchNorm + ReLU input: ragged right / bottom edges, images smaller than a tile, grids of 1 / 5 / 512 blocks.
    Same operand terms, the same three products per tap and channel group in the same order: outputs bit-identical; the
    statistics partials sum to the same totals (1e-5); and the definition (torch CPU fp32) within the usual 2e-5."""
    if ops.SPLIT != 3:
        pytest.skip("the streaming kernel exists for the f16x3 operand mode")
    x = rnd(51, B, 32, H, Wd, scale=1.5, shift=0.2)
    w = rnd(52, 32, 32, 3, 3, scale=0.2)
    xg = nhwc(x)
    wpk = ops.pack_conv_weight(w.cuda())
    isc, ish = rnd(53, 32, scale=0.5, shift=1.0).cuda(), rnd(54, 32, scale=0.3).cuda()

    def run(stream, blocks):
        old = ops.STREAM_C32, ops.STREAM_C32_BLOCKS
        ops.STREAM_C32, ops.STREAM_C32_BLOCKS = stream, blocks
        try:
            amx = torch.zeros(1, device="cuda", dtype=torch.int32)
            o0, s0 = ops.conv_fwd(xg, wpk, 32, 3, 1, stats=True, out_amax=amx)
            o1, s1 = ops.conv_fwd(xg, wpk, 32, 3, 1, in_affine=(isc, ish), stats=True)
            torch.cuda.synchronize()
            return o0, s0, o1, s1, amx.view(torch.float32).clone()
        finally:
            ops.STREAM_C32, ops.STREAM_C32_BLOCKS = old

    ref = run(False, 512)
    for blocks in (1, 5, 512):
        got = run(True, blocks)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2]), blocks
        assert float(got[4]) == float(ref[4])
        for i in (1, 3):
            a, b = got[i].double().sum(0), ref[i].double().sum(0)
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (i, blocks)
    assert relerr(nchw(run(True, 512)[0]), F.conv2d(x, w, None, 1, 1)) < 2e-5
