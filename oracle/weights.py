"""Closed-form, seed-addressed parameter fill (test infrastructure, see oracle/__init__.py).

Golden fixtures must not ship 27 MB of weights, and the reference cannot travel to the
GPU box, so every state_dict tensor is filled from a counter hash
``splitmix64(seed, tensor-index, element-index)`` that numpy evaluates identically
everywhere.  ``tools/make_golden.py`` loads the same fill into the reference's
``NeuralSpeakerModel`` (scripts/model.py:341) through ``load_state_dict``; tests rebuild
it here and compare outputs.

The key list restates what ``NeuralSpeakerModel.state_dict()`` yields (SURVEY.md section 5:
219 keys for loss='AAM', 225 for 'softmax', 224 for 'AAM-v1'); it is checked against the
key list recorded from the reference in tests/golden/state_keys_*.json.
"""
import math

import numpy as np

_MASK = (1 << 64) - 1

ARCH_LAYERS = {
    # scripts/model.py:272-331 (factories); block kind, blocks per stage
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
    "resnet101": ("bottleneck", [3, 4, 23, 3]),
}
STAGE_WIDTH = [32, 64, 128, 256]  # scripts/model.py:215-218
STAGE_STRIDE = [1, 2, 2, 2]


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(seed, stream, n):
    """n doubles in [0,1): element i of stream `stream` under `seed`."""
    base = (int(seed) * 0xD1342543DE82EF95 + int(stream) * 0x2545F4914F6CDD1D) & _MASK
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = splitmix64(splitmix64(np.uint64(base)) ^ idx)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def bn_keys(prefix, c):
    return [
        (prefix + ".weight", (c,), "bn_gamma"),
        (prefix + ".bias", (c,), "bn_beta"),
        (prefix + ".running_mean", (c,), "bn_rmean"),
        (prefix + ".running_var", (c,), "bn_rvar"),
        (prefix + ".num_batches_tracked", (), "counter"),
    ]


def state_spec(spk_num, feat_dim=40, pooling="mean", loss="softmax", arch="resnet34"):
    """[(key, shape, kind)] in the order of NeuralSpeakerModel.state_dict()
    (scripts/model.py:341-371, ResNet ctor :207-227, _make_layer :229-244)."""
    kind, layers = ARCH_LAYERS[arch]
    spec = [("res.conv1.weight", (32, 1, 3, 3), "conv")]
    spec += bn_keys("res.bn1", 32)
    inplanes = 32
    for li, (planes, nblk, stride) in enumerate(zip(STAGE_WIDTH, layers, STAGE_STRIDE)):
        for bi in range(nblk):
            p = "res.layer%d.%d" % (li + 1, bi)
            cin = inplanes if bi == 0 else planes
            if kind == "basic":
                spec.append((p + ".conv1.weight", (planes, cin, 3, 3), "conv"))
                spec += bn_keys(p + ".bn1", planes)
                spec.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv"))
                spec += bn_keys(p + ".bn2", planes)
            else:
                spec.append((p + ".conv1.weight", (planes, cin, 1, 1), "conv"))
                spec += bn_keys(p + ".bn1", planes)
                spec.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv"))
                spec += bn_keys(p + ".bn2", planes)
                spec.append((p + ".conv3.weight", (planes, planes, 1, 1), "conv"))
                spec += bn_keys(p + ".bn3", planes)
            if bi == 0 and (stride != 1 or inplanes != planes):
                spec.append((p + ".downsample.0.weight", (planes, inplanes, 1, 1), "conv"))
                spec += bn_keys(p + ".downsample.1", planes)
        inplanes = planes
    fdim = (feat_dim + 7) // 8
    fin = fdim * 256 * (2 if pooling == "mean+std" else 1)
    spec.append(("fc1.weight", (256, fin), "linear_w"))
    spec.append(("fc1.bias", (256,), "linear_b:%d" % fin))
    if loss == "softmax":
        spec += bn_keys("bn1", 256)
        spec.append(("last.weight", (spk_num, 256), "linear_w"))
        spec.append(("last.bias", (spk_num,), "linear_b:256"))
    elif loss == "AAM":
        spec.append(("last.weight", (spk_num, 256), "aam_w"))
    elif loss == "AAM-v1":
        spec += bn_keys("bn1", 256)
        spec.append(("last.weight", (spk_num, 256), "aam_w"))
    else:
        raise NotImplementedError(loss)
    return spec


def fill_tensor(seed, stream, shape, kind):
    n = int(np.prod(shape)) if len(shape) else 1
    if kind == "counter":
        return np.zeros((), dtype=np.int64)
    u = hash_uniform(seed, stream, n)
    if kind == "conv":
        # same std as kaiming_normal_(fan_out, relu) (scripts/model.py:222-224), uniform shape
        fan_out = shape[0] * shape[2] * shape[3]
        std = math.sqrt(2.0 / fan_out)
        v = (2.0 * u - 1.0) * (std * math.sqrt(3.0))
    elif kind == "bn_gamma":
        v = 0.8 + 0.4 * u
    elif kind == "bn_beta":
        v = 0.2 * u - 0.1
    elif kind == "bn_rmean":
        v = 0.2 * u - 0.1
    elif kind == "bn_rvar":
        v = 0.8 + 0.4 * u
    elif kind == "linear_w":
        bound = 1.0 / math.sqrt(shape[1])
        v = (2.0 * u - 1.0) * bound
    elif kind.startswith("linear_b"):
        bound = 1.0 / math.sqrt(int(kind.split(":")[1]))
        v = (2.0 * u - 1.0) * bound
    elif kind == "aam_w":
        std = math.sqrt(2.0 / (shape[0] + shape[1]))  # xavier_normal_ std, scripts/model.py:471
        v = (2.0 * u - 1.0) * (std * math.sqrt(3.0))
    else:
        raise ValueError(kind)
    return v.astype(np.float32).reshape(shape)


def make_state(seed, spk_num, feat_dim=40, pooling="mean", loss="softmax", arch="resnet34"):
    """Ordered dict key -> numpy array for the whole state_dict."""
    out = {}
    for i, (key, shape, kind) in enumerate(state_spec(spk_num, feat_dim, pooling, loss, arch)):
        out[key] = fill_tensor(seed, i, shape, kind)
    return out


def make_input(seed, batch, feat_dim, frames, spk_num):
    """Synthetic fbank [B,F,T] ~ approx N(0,1) (sum of 4 uniforms, rescaled) and labels [B]."""
    n = batch * feat_dim * frames
    acc = np.zeros(n, dtype=np.float64)
    for j in range(4):
        acc += hash_uniform(seed, 1000 + j, n)
    x = ((acc - 2.0) * math.sqrt(3.0)).astype(np.float32).reshape(batch, feat_dim, frames)
    y = (hash_uniform(seed, 2000, batch) * spk_num).astype(np.int64)
    return x, y
