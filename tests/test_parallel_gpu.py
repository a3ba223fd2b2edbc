"""Data-parallel semantics of the training step, checked on ONE MI355X (reference scripts/train_resnet.py:183-185:
per-rank batch = global / ngpus, DistributedDataParallel averages the gradients over ranks, BatchNorm statistics stay
per rank).

A golden batch is split into two shards ("ranks").  Each shard runs through Engine.loss_and_grad with a recording
on_stage_done hook - the hook the RCCL reducer hangs on - then the two gradient arenas are summed (what all-reduce(sum)
leaves on every rank) and FlatSGD(grad_scale = 1/2) steps.  The resulting parameters must equal the CPU oracle's:
mean of the per-shard gradients (each shard normalised with its OWN batch statistics) -> SGD.  Also checked: the hook
fires head -> layer4 -> ... -> stem, every reported slice is final when it is reported (so an all-reduce enqueued at
that moment reads finished gradients), and the stage-segmented hipGraph replay used when world > 1 is bit-identical to
the eager launch sequence.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import spk_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402

ORDER = ["head", "layer4", "layer3", "layer2", "layer1", "stem"]


def _build(meta):
    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    npst = W.make_state(meta["seed"], meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], meta["arch"])
    m = NeuralSpeakerModel(meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], 0.2, 30, arch=meta["arch"])
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in npst.items()})
    return m.cuda().train(), npst


class _Recorder:
    """Stands where parallel.GradAllReducer.on_stage_done stands: snapshots the reported slice on the launch stream."""

    def __init__(self, model):
        from pytorch_kaldi_resnet_amd.parallel import stage_slices
        self.model, self.slices = model, stage_slices(model)
        self.names, self.snaps = [], {}

    def __call__(self, name):
        lo, hi = self.slices[name]
        self.names.append(name)
        self.snaps[name] = self.model.flat_grads()[lo:hi].clone()


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6", "f32"])
def test_two_shards_average_like_ddp(gold_dir, mode):
    from pytorch_kaldi_resnet_amd import ops
    from pytorch_kaldi_resnet_amd.optim import FlatSGD
    old, ops.SPLIT = ops.SPLIT, ops.MFMA_MODES[mode]
    try:
        meta = json.load(open(os.path.join(gold_dir, "c1_r34_aam.json")))
        m, npst = _build(meta)
        x, y = W.make_input(meta["seed"] + 1, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
        assert meta["batch"] % 2 == 0
        h = meta["batch"] // 2
        shards = [(x[:h], y[:h]), (x[h:], y[h:])]
        lr, wd = 0.05, 5e-4
        opt = FlatSGD(m, lr, momentum=0.9, weight_decay=wd, grad_scale=0.5)      # 1 / world folded into the SGD kernel
        p0 = m.flat_parameters().clone()
        total = torch.zeros_like(m.flat_grads())
        losses = []
        for xs, ys in shards:
            opt.zero_grad(set_to_none=True)
            rec = _Recorder(m)
            loss, _, _ = m.engine().loss_and_grad(torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda(), rec)
            torch.cuda.synchronize()
            assert rec.names == ORDER
            g = m.flat_grads()
            covered = torch.zeros_like(g)
            for name in ORDER:
                lo, hi = rec.slices[name]
                assert torch.equal(rec.snaps[name], g[lo:hi]), "slice %s changed after it was reported" % name
                covered[lo:hi] += 1
            assert bool((covered == 1).all())
            total += g
            losses.append(float(loss))
        m.flat_grads().copy_(total)             # = all-reduce(sum) over the two ranks
        opt.step()
        hip_delta = (m.flat_parameters() - p0).double().cpu()

        # oracle: per-shard gradients (own BN statistics), averaged, one SGD step
        def oracle_delta(dtype):
            grads, lo = None, []
            for xs, ys in shards:
                st = O.to_torch_state(npst)
                st = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
                keys = O.trainable_keys(st)
                for k in keys:
                    st[k].requires_grad_(True)
                lg = O.forward(st, torch.from_numpy(xs).to(dtype), torch.from_numpy(ys), meta["pooling"], meta["loss"],
                               meta["arch"], train=True)
                lv = O.cross_entropy(lg, torch.from_numpy(ys))
                gs = torch.autograd.grad(lv, [st[k] for k in keys])
                lo.append(float(lv))
                grads = [a.detach() for a in gs] if grads is None else [a + b.detach() for a, b in zip(grads, gs)]
            st = O.to_torch_state(npst)
            out = []
            for k, gsum in zip(keys, grads):
                p = st[k].to(dtype)
                out.append((-lr * (gsum / 2 + wd * p)).reshape(-1).double())       # first step: buf = g
            return out, keys, lo

        d32, keys, lo32 = oracle_delta(torch.float32)
        d64, _, _ = oracle_delta(torch.float64)
        names = [n for n, _ in m.named_parameters()]
        assert names == keys
        hd = [hip_delta[o:o + p.numel()] for p, o in zip(m.parameters(), m._offsets)]
        for a, b in zip(losses, lo32):
            assert abs(a - b) < 2e-4
        f64, f32, fh = torch.cat(d64), torch.cat(d32), torch.cat(hd)
        e_oracle = float((f32 - f64).norm() / f64.norm())
        e_hip = float((fh - f64).norm() / f64.norm())
        print("DP update error vs fp64 oracle: oracle-fp32 %.3e hip %.3e" % (e_oracle, e_hip))
        assert e_hip <= 3.0 * e_oracle + GRAD_SLACK
        # NOT the single-batch gradient: BatchNorm statistics are per shard (train_resnet.py:183 - plain BatchNorm2d under DDP)
        st = O.to_torch_state(npst)
        kk = O.trainable_keys(st)
        for k in kk:
            st[k].requires_grad_(True)
        lg = O.forward(st, torch.from_numpy(x), torch.from_numpy(y), meta["pooling"], meta["loss"], meta["arch"], train=True)
        gs = torch.autograd.grad(O.cross_entropy(lg, torch.from_numpy(y)), [st[k] for k in kk])
        whole = torch.cat([(-lr * (g.detach() + wd * st[k].detach())).reshape(-1).double() for k, g in zip(kk, gs)])
        d_whole = float((fh - whole).norm() / whole.norm())
        print("distance to the single-batch (synchronised-BN) update: %.3e" % d_whole)
        assert d_whole > 2 * e_hip
    finally:
        ops.SPLIT = old


# additive slack of the whole-gradient comparison against fp64 (see tests/test_model_gpu.py::test_backward_parity)
GRAD_SLACK = 2e-2


def test_segmented_graph_replay_equals_eager(gold_dir):
    """world > 1 replays the step as six hipGraph segments with the stage hook between them: same numbers as eager,
    hook order head..stem, slices final when reported, weights re-packed every step."""
    from pytorch_kaldi_resnet_amd.engine import GraphedTrainStep
    from pytorch_kaldi_resnet_amd.optim import FlatSGD
    meta = json.load(open(os.path.join(gold_dir, "c1_r34_aam.json")))
    me, _ = _build(meta)
    mg, _ = _build(meta)
    oe = FlatSGD(me, 1e-2, momentum=0.9, weight_decay=5e-4)
    og = FlatSGD(mg, 1e-2, momentum=0.9, weight_decay=5e-4)
    step = GraphedTrainStep(mg.engine(), meta["batch"], meta["frames"], warmup=1, segmented=True)
    assert len(step.segments) == 6
    for s in range(3):
        xs, ys = W.make_input(meta["seed"] + 1 + s, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
        xs, ys = torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda()
        oe.zero_grad(set_to_none=True)
        l1, _, _ = me.engine().loss_and_grad(xs, ys)
        rec = _Recorder(mg)
        l2, _, _ = step(xs, ys, rec)
        torch.cuda.synchronize()
        assert rec.names == ORDER
        for name in ORDER:
            lo, hi = rec.slices[name]
            assert torch.equal(rec.snaps[name], mg.flat_grads()[lo:hi]), name
        assert float(l1) == float(l2)
        assert torch.equal(me.flat_grads(), mg.flat_grads())
        oe.step()
        og.step()
        assert torch.equal(me.flat_parameters(), mg.flat_parameters())
