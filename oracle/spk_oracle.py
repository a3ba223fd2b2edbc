"""Torch-CPU fp32 restatement of the reference hot path (test infrastructure only).

Functional style over a plain ``dict`` of tensors keyed like the reference state_dict.
Every function cites the reference lines it follows (paths relative to the reference
repo).  Gradients come from torch autograd over these same functions, which is what the
reference does (scripts/train_resnet.py:327).

Pinning: tests/test_oracle_golden.py checks these functions against arrays produced by
the imported reference (tools/make_golden.py) - logits, loss, embeddings, gradients,
BN running statistics and a 5-step SGD loss curve.
"""
import math

import torch
import torch.nn.functional as F

from .weights import ARCH_LAYERS, STAGE_STRIDE, STAGE_WIDTH

BN_EPS = 1e-5       # nn.BatchNorm2d default, scripts/model.py:41
BN_MOMENTUM = 0.1


def to_torch_state(np_state, requires_grad=False):
    st = {}
    for k, v in np_state.items():
        t = torch.from_numpy(v.copy()) if v.shape != () else torch.tensor(int(v), dtype=torch.int64)
        if requires_grad and t.is_floating_point() and not (
                k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        st[k] = t
    return st


def _bn(st, prefix, x, train):
    """nn.BatchNorm2d / BatchNorm1d forward (scripts/model.py:41,44,212,235,361).
    train: batch mean / biased var, running stats momentum 0.1 with unbiased var,
    num_batches_tracked += 1;  eval: running stats."""
    rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    if train:
        st[prefix + ".num_batches_tracked"] = st[prefix + ".num_batches_tracked"] + 1
    return F.batch_norm(x, rm, rv, st[prefix + ".weight"], st[prefix + ".bias"],
                        training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def _basic_block(st, p, x, stride, has_ds, train):
    """BasicBlock.forward, scripts/model.py:48-64."""
    out = F.conv2d(x, st[p + ".conv1.weight"], None, stride, 1)
    out = F.relu(_bn(st, p + ".bn1", out, train))
    out = F.conv2d(out, st[p + ".conv2.weight"], None, 1, 1)
    out = _bn(st, p + ".bn2", out, train)
    res = x
    if has_ds:  # scripts/model.py:232-236
        res = F.conv2d(x, st[p + ".downsample.0.weight"], None, stride, 0)
        res = _bn(st, p + ".downsample.1", res, train)
    return F.relu(out + res)


def _bottleneck(st, p, x, stride, has_ds, train):
    """Bottleneck.forward (expansion = 1), scripts/model.py:100-135."""
    out = F.conv2d(x, st[p + ".conv1.weight"], None, 1, 0)
    out = F.relu(_bn(st, p + ".bn1", out, train))
    out = F.conv2d(out, st[p + ".conv2.weight"], None, stride, 1)
    out = F.relu(_bn(st, p + ".bn2", out, train))
    out = F.conv2d(out, st[p + ".conv3.weight"], None, 1, 0)
    out = _bn(st, p + ".bn3", out, train)
    res = x
    if has_ds:
        res = F.conv2d(x, st[p + ".downsample.0.weight"], None, stride, 0)
        res = _bn(st, p + ".downsample.1", res, train)
    return F.relu(out + res)


def trunk(st, x, arch="resnet34", train=False):
    """ResNet.forward, scripts/model.py:246-269.  x: [B,F,T] -> [B,256,ceil(F/8),ceil(T/8)]."""
    kind, layers = ARCH_LAYERS[arch]
    x = x.view(x.size(0), 1, x.size(1), x.size(2))           # :247
    x = F.conv2d(x, st["res.conv1.weight"], None, 1, 1)       # :249
    x = F.relu(_bn(st, "res.bn1", x, train))                  # :250-251
    block = _basic_block if kind == "basic" else _bottleneck
    inplanes = 32
    for li, (planes, nblk, stride) in enumerate(zip(STAGE_WIDTH, layers, STAGE_STRIDE)):
        for bi in range(nblk):
            p = "res.layer%d.%d" % (li + 1, bi)
            s = stride if bi == 0 else 1
            has_ds = bi == 0 and (stride != 1 or inplanes != planes)
            x = block(st, p, x, s, has_ds, train)
        inplanes = planes
    return x


def stats_pool(x, pooling):
    """StatsPooling.forward, scripts/model.py:441-457.  'mean+std' reproduces the swapped
    outputs of :450 (`mean, var = torch.var_mean(...)` binds (var, mean)): the layer emits
    cat([unbiased var over T, sqrt(mean over T)], -1)."""
    if pooling == "mean":
        return x.mean(dim=3, keepdim=True)                    # AdaptiveAvgPool2d((None,1)) :439
    if pooling == "mean+std":
        var = x.var(dim=3, unbiased=True)
        mean = x.mean(dim=3)
        return torch.cat([var, torch.sqrt(mean)], dim=-1)     # :453-454
    raise NotImplementedError(pooling)


def embed(st, x, pooling="mean+std", arch="resnet34", train=False):
    """NeuralSpeakerModel.predict, scripts/model.py:402-409."""
    h = trunk(st, x, arch, train)
    h = stats_pool(h, pooling).flatten(1)                     # :352
    return F.linear(h, st["fc1.weight"], st["fc1.bias"])      # :357


def aam_logits(emb, weight, label, m=0.2, s=30.0):
    """AAMLayer.forward, scripts/model.py:483-501 (easy_margin=False)."""
    cos_m, sin_m = math.cos(m), math.sin(m)
    th = math.cos(math.pi - m)
    mm = math.sin(math.pi - m) * m
    cosine = F.linear(F.normalize(emb), F.normalize(weight))
    sine = torch.sqrt((1.0 - cosine * cosine).clamp(0, 1))
    phi = cosine * cos_m - sine * sin_m
    phi = torch.where((cosine - th) > 0, phi, cosine - mm)
    one_hot = torch.zeros_like(cosine)
    one_hot.scatter_(1, label.view(-1, 1), 1)
    return (one_hot * phi + (1.0 - one_hot) * cosine) * s


def forward(st, x, y=None, pooling="mean+std", loss="AAM", arch="resnet34", train=False,
            m=0.2, s=30.0):
    """NeuralSpeakerModel.forward, scripts/model.py:374-400."""
    e = embed(st, x, pooling, arch, train)
    if loss == "softmax":
        h = F.relu(_bn(st, "bn1", e, train))
        return F.linear(h, st["last.weight"], st["last.bias"])
    if loss == "AAM":
        return aam_logits(e, st["last.weight"], y, m, s)
    if loss == "AAM-v1":
        h = F.relu(_bn(st, "bn1", e, train))
        return aam_logits(h, st["last.weight"], y, m, s)
    raise NotImplementedError(loss)


def cross_entropy(logits, y):
    """nn.CrossEntropyLoss() default mean reduction, scripts/train_resnet.py:201,317."""
    return (torch.logsumexp(logits, dim=1) - logits.gather(1, y.view(-1, 1)).squeeze(1)).mean()


def accuracy(output, target, topk=(1,)):
    """scripts/accuracy.py:4-16 (with .reshape so it runs on torch >= 1.7)."""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return [correct[:k].reshape(-1).float().sum() * (100.0 / target.size(0)) for k in topk]


def trainable_keys(st):
    return [k for k, v in st.items() if v.is_floating_point()
            and not (k.endswith("running_mean") or k.endswith("running_var"))]


def sgd_step(st, grads, bufs, lr, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD semantics used at scripts/train_resnet.py:203-205,328:
    g += wd*p; buf = g on the first step else momentum*buf + g; p -= lr*buf."""
    with torch.no_grad():
        for k, g in grads.items():
            p = st[k]
            g = g + weight_decay * p
            if k not in bufs:
                bufs[k] = g.clone()
            else:
                bufs[k].mul_(momentum).add_(g)
            p.sub_(lr * bufs[k])


def cosine_lr(epoch, epochs, lr0, lr_final):
    """CosineAnnealingLR(T_max=epochs, eta_min=lr_final) closed form,
    scripts/train_resnet.py:206,275."""
    return lr_final + (lr0 - lr_final) * (1.0 + math.cos(math.pi * epoch / epochs)) / 2.0


def train_step(st, bufs, x, y, lr, pooling="mean+std", loss="AAM", arch="resnet34",
               m=0.2, s=30.0, momentum=0.9, weight_decay=1e-4):
    """One iteration of the loop at scripts/train_resnet.py:304-328.
    Returns (loss value, logits, grads dict)."""
    keys = trainable_keys(st)
    for k in keys:
        st[k].requires_grad_(True)
        st[k].grad = None
    logits = forward(st, x, y, pooling, loss, arch, train=True, m=m, s=s)
    lv = cross_entropy(logits, y)
    gs = torch.autograd.grad(lv, [st[k] for k in keys])
    grads = {k: g for k, g in zip(keys, gs)}
    sgd_step(st, grads, bufs, lr, momentum, weight_decay)
    return float(lv.detach()), logits.detach(), grads


# ---- scoring back end (the step after the path; reference scripts restated) -------------

def cosine_scores(enroll, test, trials, mean=None):
    """scripts/cosine_score.py:52-68: mean-subtract, cosine per trial (float32)."""
    import numpy as np
    out = []
    for e, t in trials:
        a, b = np.asarray(enroll[e], dtype=np.float64), np.asarray(test[t], dtype=np.float64)
        if mean is not None:
            a, b = a - mean, b - mean
        a32, b32 = torch.from_numpy(a).float(), torch.from_numpy(b).float()
        out.append(float(F.cosine_similarity(a32, b32, dim=0)))
    return out


def compute_eer(scores, labels):
    """scripts/compute_eer.py:35-70,101-105: ComputeErrorRates + nanargmin crossing.
    scores: floats; labels: 1 = target, 0 = nontarget.  Returns the EER as a fraction."""
    import numpy as np
    order = sorted(range(len(scores)), key=lambda i: scores[i])   # stable, like sorted(key=itemgetter(1))
    lab = np.asarray([labels[i] for i in order], dtype=np.float64)
    fnrs = np.cumsum(lab)
    fprs = np.cumsum(1.0 - lab)
    fnrs_norm = lab.sum()
    fprs_norm = len(lab) - fnrs_norm
    fnrs = fnrs / float(fnrs_norm)
    fprs = 1.0 - fprs / float(fprs_norm)
    idx = int(np.nanargmin(np.absolute(fnrs - fprs)))
    return float(max(fprs[idx], fnrs[idx]))
