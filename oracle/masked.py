"""Oracle forward with PRESCRIBED ReLU masks (test infrastructure only).

Why: the gradient of this network is a discontinuous function of the forward rounding.  relu'(z) is a step, so an
activation whose pre-activation z is within the forward's fp32 rounding error of 0 (|z| ~ 1e-5 after 30 layers) gets
mask 0 in one correct fp32 implementation and 1 in another; each such flip moves that layer's gradient by one whole
element, ~1/sqrt(N) of its norm (3e-3 in layer 4 of a 4-utterance batch).  The number of flips is Poisson with a mean
of a few per network, for the CPU fp32 path exactly as for the HIP path (tools/diag_fwd.py counts them against fp64).
Comparing a gradient against an fp64 run that chose its OWN masks therefore measures luck, not arithmetic.

`forward(..., masks=...)` replays the reference forward (same functions as spk_oracle.forward, scripts/model.py:246-269,
374-400) with every F.relu replaced by `where(mask, z, 0)`, masks taken in call order from the implementation under test; the
autograd gradient of that is the exact gradient of the function the implementation differentiated, and fp64 makes it
the yardstick: what is left is pure rounding error of the backward arithmetic.
`record_masks(...)` runs a forward and returns the masks it chose itself (for the CPU fp32 path's own yardstick).
"""
import contextlib

import torch

from . import spk_oracle as O


@contextlib.contextmanager
def _patched_relu(fn):
    real = O.F.relu
    O.F.relu = fn
    try:
        yield
    finally:
        O.F.relu = real


def record_masks(st, x, y, pooling, loss, arch):
    """Training forward of spk_oracle on `st`; -> (logits, [bool mask of every F.relu call, in call order])."""
    masks = []
    real = torch.relu

    def relu(z, inplace=False):
        masks.append((z > 0).detach())
        return real(z)

    with _patched_relu(relu):
        logits = O.forward(st, x, y, pooling, loss, arch, train=True)
    return logits, masks


def forward(st, x, y, pooling, loss, arch, masks):
    """Training forward with relu(z) := z * masks[i] for the i-th F.relu call.  Asserts every mask is consumed."""
    it = iter(masks)

    def relu(z, inplace=False):
        mk = next(it)
        assert mk.shape == z.shape, (mk.shape, z.shape)
        return torch.where(mk, z, torch.zeros_like(z))      # a select, like relu's backward: inf * 0 never forms

    with _patched_relu(relu):
        logits = O.forward(st, x, y, pooling, loss, arch, train=True)
    assert next(it, None) is None, "more masks than ReLU calls"
    return logits


def grads(np_state, x, y, pooling, loss, arch, masks, dtype=torch.float64):
    """Gradient of mean cross-entropy wrt every trainable tensor, in `dtype`, under the prescribed masks.
    -> (loss value, {key: grad (float64)})."""
    st = O.to_torch_state(np_state)
    st = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
    keys = O.trainable_keys(st)
    for k in keys:
        st[k].requires_grad_(True)
    logits = forward(st, torch.from_numpy(x).to(dtype), torch.from_numpy(y), pooling, loss, arch, masks)
    lv = O.cross_entropy(logits, torch.from_numpy(y))
    gs = torch.autograd.grad(lv, [st[k] for k in keys])
    return float(lv), {k: g.double() for k, g in zip(keys, gs)}
