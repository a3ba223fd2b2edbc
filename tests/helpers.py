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
