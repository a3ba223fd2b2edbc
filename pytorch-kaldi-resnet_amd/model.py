"""NeuralSpeakerModel: the drop-in boundary.

Mirrors the reference module's interface (scripts/model.py:334-432): same constructor arguments (plus a
real ``arch`` switch, which the reference parses but ignores - train_resnet.py:42,152), ``forward(x, y)``,
``predict(x)``, ``loadParameters(state)``, the 219/225/224-key ``state_dict`` naming, ``parameters()`` whose
``.grad`` is filled by ``loss.backward()``, and ``train()/eval()`` switching BatchNorm behaviour.  All
arithmetic runs in libspkhip.so (hand-written gfx950 kernels) through ``engine.Engine``; the modules below
only hold parameters.  There is no CPU path: calling forward on a CPU tensor raises.
"""
import math

import torch
import torch.nn as nn

from . import engine

ARCH_LAYERS = {
    # reference factories, scripts/model.py:272-331: block kind and blocks per stage
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
    "resnet101": ("bottleneck", [3, 4, 23, 3]),
}
STAGE_WIDTH = [32, 64, 128, 256]     # scripts/model.py:215-218
STAGE_STRIDE = [1, 2, 2, 2]


class ConvP(nn.Module):
    """Parameter holder for nn.Conv2d(bias=False); init kaiming_normal_(fan_out, relu), scripts/model.py:222-224."""

    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")


class BNP(nn.Module):
    """Parameter / buffer holder for nn.BatchNorm2d / BatchNorm1d (weight 1, bias 0; scripts/model.py:225-227)."""

    def __init__(self, c):
        super().__init__()
        self.c = c
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class LinearP(nn.Module):
    """Holder for nn.Linear with torch's default init (kaiming_uniform(a=sqrt 5) / U(+-1/sqrt(fan_in)))."""

    def __init__(self, fin, fout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(fin)
            self.bias = nn.Parameter(torch.empty(fout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)


class AAMP(nn.Module):
    """Holder for AAMLayer.weight [n_classes, in_feats], xavier_normal_ (scripts/model.py:470-471)."""

    def __init__(self, in_feats, n_classes, m, s):
        super().__init__()
        self.m, self.s = m, s
        self.weight = nn.Parameter(torch.empty(n_classes, in_feats))
        nn.init.xavier_normal_(self.weight, gain=1)


class BlockP(nn.Module):
    def __init__(self, kind, inplanes, planes, stride, downsample):
        super().__init__()
        self.kind, self.stride = kind, stride
        if kind == "basic":      # scripts/model.py:38-46
            self.conv1 = ConvP(inplanes, planes, 3, stride)
            self.bn1 = BNP(planes)
            self.conv2 = ConvP(planes, planes, 3, 1)
            self.bn2 = BNP(planes)
        else:                    # Bottleneck with expansion = 1, scripts/model.py:100-113
            self.conv1 = ConvP(inplanes, planes, 1, 1)
            self.bn1 = BNP(planes)
            self.conv2 = ConvP(planes, planes, 3, stride)
            self.bn2 = BNP(planes)
            self.conv3 = ConvP(planes, planes, 1, 1)
            self.bn3 = BNP(planes)
        if downsample:           # scripts/model.py:231-236
            self.downsample = nn.Sequential(ConvP(inplanes, planes, 1, stride), BNP(planes))
        else:
            self.downsample = None


class ResNetP(nn.Module):
    """Parameter tree of the reference ResNet (scripts/model.py:205-244): stem + 4 stages, no max/avg pool."""

    def __init__(self, arch):
        super().__init__()
        kind, layers = ARCH_LAYERS[arch]
        self.conv1 = ConvP(1, 32, 3, 1)
        self.bn1 = BNP(32)
        inplanes = 32
        for li, (planes, nblk, stride) in enumerate(zip(STAGE_WIDTH, layers, STAGE_STRIDE)):
            blocks = []
            for bi in range(nblk):
                s = stride if bi == 0 else 1
                ds = bi == 0 and (stride != 1 or inplanes != planes)
                blocks.append(BlockP(kind, inplanes if bi == 0 else planes, planes, s, ds))
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))
            inplanes = planes


class NeuralSpeakerModel(nn.Module):
    """Same constructor as the reference (scripts/model.py:341) + ``arch``."""

    def __init__(self, spk_num, feat_dim=40, pooling="mean", loss="softmax", m=0.2, s=30, arch="resnet34"):
        super().__init__()
        if arch not in ARCH_LAYERS:
            raise NotImplementedError(arch)
        if pooling not in ("mean", "mean+std"):
            raise NotImplementedError(pooling)
        self.loss = loss
        self.arch = arch
        self.pooling = pooling
        self.feat_dim = feat_dim
        self.spk_num = spk_num
        self.res = ResNetP(arch)
        fdim = (feat_dim + 7) // 8
        self.fc1 = LinearP(fdim * 256 * (2 if pooling == "mean+std" else 1), 256)
        if loss == "softmax":
            self.bn1 = BNP(256)
            self.last = LinearP(256, spk_num)
        elif loss == "AAM":
            self.last = AAMP(256, spk_num, m, s)
        elif loss == "AAM-v1":
            self.bn1 = BNP(256)
            self.last = AAMP(256, spk_num, m, s)
        else:
            raise NotImplementedError
        if loss != "softmax":
            print("Initialised AAM m=%.3f s=%.3f" % (m, s))
        self.m, self.s = m, s
        self._engine = None
        self._flatten()

    # ---- flat parameter / gradient arenas -------------------------------------------------------------
    def _flatten(self):
        """(Re)build the flat fp32 parameter arena on the parameters' current device and re-point every
        Parameter at its slice (16-byte aligned), so SGD, gradient all-reduce and repacking see one buffer."""
        params = [p for p in self.parameters()]
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
        self._flat = flat
        self._offsets = offs
        self._grad_flat = None
        self._engine = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flatten()
        return out

    def flat_parameters(self):
        return self._flat

    def flat_grads(self):
        """Flat gradient arena; every Parameter's .grad is a view into it."""
        if self._grad_flat is None or self._grad_flat.device != self._flat.device:
            self._grad_flat = torch.zeros_like(self._flat)
        return self._grad_flat

    def attach_grads(self):
        """Point every .grad at its arena slice. Returns True when the slices were (re)attached - i.e. the
        previous gradients were dropped (zero_grad(set_to_none=True)) and this backward must overwrite."""
        gf = self.flat_grads()
        fresh = False
        for p, o in zip(self.parameters(), self._offsets):
            g = p.grad
            if g is None or g.data_ptr() != gf.data_ptr() + 4 * o:
                p.grad = gf[o:o + p.numel()].view(p.shape)
                fresh = True
        return fresh

    def engine(self):
        if self._engine is None:
            self._engine = engine.Engine(self)
        return self._engine

    def mark_weights_changed(self):
        if self._engine is not None:
            self._engine.dirty = True

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.mark_weights_changed()
        return out

    # ---- reference interface ------------------------------------------------------------------------------
    def forward(self, x, y=None):
        """Logits [B, spk_num] (scripts/model.py:374-400). Differentiable: loss.backward() fills .grad."""
        return self.engine().forward_logits(x, y)

    def predict(self, x):
        """Embeddings [B, 256] = fc1 output (scripts/model.py:402-409)."""
        return self.engine().predict(x)

    def loadParameters(self, loaded_state):
        """scripts/model.py:415-432: name match, strip 'module.', skip on shape mismatch, same messages."""
        self_state = self.state_dict()
        for name, param in loaded_state.items():
            origname = name
            if name not in self_state:
                name = name.replace("module.", "")
                if name not in self_state:
                    print("%s is not in the model." % origname)
                    continue
            if self_state[name].size() != loaded_state[origname].size():
                print("Wrong parameter length: %s, model: %s, loaded: %s" % (
                    origname, self_state[name].size(), loaded_state[origname].size()))
                continue
            self_state[name].copy_(param)
        self.mark_weights_changed()
