"""Fused SGD over the model's flat parameter arena + the per-epoch cosine schedule of the reference.

``FlatSGD`` follows torch.optim.SGD semantics exactly as the reference uses it
(scripts/train_resnet.py:203-205,328: momentum 0.9, weight decay on every parameter, no dampening/nesterov)
but runs as ONE kernel launch (spk_sgd_step) over the 27.8 MB arena instead of ~220 small tensor ops.
It subclasses torch.optim.Optimizer so torch.optim.lr_scheduler.CosineAnnealingLR
(scripts/train_resnet.py:206) drives it unchanged, and its state_dict()/load_state_dict() speak torch SGD's
format so checkpoints interchange with the reference ('optimizer' entry, scripts/train_resnet.py:288).
"""
import torch

from . import ops


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, model, lr, momentum=0.0, weight_decay=0.0, grad_scale=1.0):
        self._model = model
        params = list(model.parameters())
        defaults = dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False,
                        maximize=False, foreach=None, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._buf = None
        self._first = True
        self.grad_scale = grad_scale

    def _arena(self):
        flat = self._model.flat_parameters()
        if self._buf is None or self._buf.device != flat.device or self._buf.numel() != flat.numel():
            self._buf = torch.zeros_like(flat)
            self._first = True
        return flat

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        flat = self._arena()
        self._model.attach_grads()
        ops.sgd_step(flat, self._model.flat_grads(), self._buf, g["lr"], g["momentum"], g["weight_decay"],
                     self.grad_scale, self._first or g["momentum"] == 0)
        self._first = False
        self._model.mark_weights_changed()

    def zero_grad(self, set_to_none=False):
        """Zero the gradient arena in one memset (set_to_none=True drops the views; the next backward re-attaches
        them and overwrites, which skips the memset entirely - the fast path used by train.py)."""
        if set_to_none:
            for p in self._model.parameters():
                p.grad = None
        else:
            self._model.flat_grads().zero_()

    # ---- torch.optim.SGD-compatible (de)serialisation -------------------------------------------------------
    def state_dict(self):
        flat = self._arena()
        state = {}
        if not self._first:
            for i, (p, o) in enumerate(zip(self._model.parameters(), self._model._offsets)):
                state[i] = {"momentum_buffer": self._buf[o:o + p.numel()].view(p.shape).clone()}
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d["params"] = list(range(len(g["params"])))
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        flat = self._arena()
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
        st = sd.get("state", {})
        if len(st):
            for i, (p, o) in enumerate(zip(self._model.parameters(), self._model._offsets)):
                ent = st.get(i, st.get(str(i)))
                if ent is not None and ent.get("momentum_buffer") is not None:
                    self._buf[o:o + p.numel()].copy_(ent["momentum_buffer"].reshape(-1).to(flat.device))
            self._first = False


def cosine_lr(epoch, epochs, lr0, lr_final):
    """Closed form of CosineAnnealingLR(T_max=epochs, eta_min=lr_final) (scripts/train_resnet.py:206,275)."""
    import math
    return lr_final + (lr0 - lr_final) * (1.0 + math.cos(math.pi * epoch / epochs)) / 2.0
