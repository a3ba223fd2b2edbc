"""Helpers shared by the GPU parity tests (test infrastructure only)."""
import torch


def hip_relu_masks(eng, saved):
    """The ReLU masks the HIP training forward chose, in the call order of the reference forward (scripts/model.py:250,
    48-64 / 115-135 per block, head :361-363) - the masks the backward kernels differentiate with.  Inner masks come from the
    same fused multiply-add the kernels use (spk_bn_apply); block-output masks from the stored block outputs."""
    from pytorch_kaldi_resnet_amd import ops
    nchw = lambda t: (t > 0).permute(0, 3, 1, 2).cpu()      # noqa: E731
    masks = [nchw(saved["blocks"][0]["x"])]
    for b, rec in zip(eng.blocks, saved["blocks"]):
        for raw, bn in zip(rec["raws"][:-1], b.bns[:-1]):
            masks.append(nchw(ops.bn_apply(raw, bn.t4[2], bn.t4[3], relu=True)))
        masks.append(nchw(rec["out"]))
    if "h" in saved["head"]:
        masks.append((saved["head"]["h"] > 0).cpu())
    return masks


# ---- f16 pair tensors and scale slots of the f16x3 operand mode (host restatements used by the kernel tests)
def slot():
    return torch.zeros(1, device="cuda", dtype=torch.int32)


def sigma_of(slot_t):
    """host copy of spk_sigma_from_amax_bits"""
    bits = int(slot_t.cpu().view(torch.int32)[0]) & 0xFFFFFFFF
    e = (bits >> 23) & 0xFF
    if e in (0, 255):
        return 1.0
    return 2.0 ** (14 - (e - 127))


def slot_value(slot_t):
    return float(slot_t.cpu().view(torch.float32)[0])


def encode_pairs(t32, sig):
    """host restatement of split2h + the pair layout: [.., 4k..4k+3] floats -> [4 x fp16 hi][4 x fp16 lo] of value * sigma"""
    u = (t32.double() * sig).float().clamp(-65504.0, 65504.0)       # sigma is a power of two: exact
    hi = u.half()
    lo = (u - hi.float()).half()
    g = t32.shape[-1] // 4
    hi = hi.reshape(-1, g, 4)
    lo = lo.reshape(-1, g, 4)
    return torch.cat([hi, lo], dim=-1).reshape(-1).view(torch.float32).reshape(t32.shape)


def decode_pairs(tp, sig):
    h = tp.reshape(-1).view(torch.float16).reshape(-1, 8)
    return ((h[:, :4].double() + h[:, 4:].double()) / sig).reshape(tp.shape)
