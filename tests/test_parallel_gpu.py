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
        from helpers import hip_relu_masks
        from oracle import masked
        masks_hip = []
        for xs, ys in shards:
            opt.zero_grad(set_to_none=True)
            rec = _Recorder(m)
            xg, yg = torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda()
            eng = m.engine()
            with torch.no_grad():          # (.grad is None after zero_grad(set_to_none=True): this backward overwrites the arena)
                logits, saved = eng.forward_train(xg, yg)
                masks_hip.append(hip_relu_masks(eng, saved))       # the ReLU masks this shard's backward differentiates with
                loss_row, dl, _ = ops.softmax_ce(logits, yg, grad_scale=1.0 / logits.shape[0])
                loss = ops.mean(loss_row)
            eng.backward(saved, dl, rec)
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
        kw = dict(pooling=meta["pooling"], loss=meta["loss"], arch=meta["arch"])

        def update(grad_dicts, keys):
            st = O.to_torch_state(npst)
            return [(-lr * ((grad_dicts[0][k] + grad_dicts[1][k]) / 2 + wd * st[k].double())).reshape(-1) for k in keys]   # first step: buf = g

        # oracle: per-shard gradients (own BN statistics), averaged, one SGD step - the CPU fp32 path with its own masks
        def oracle_own(dtype):
            gd, lo, mk = [], [], []
            for xs, ys in shards:
                st = O.to_torch_state(npst)
                st = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
                keys = O.trainable_keys(st)
                for k in keys:
                    st[k].requires_grad_(True)
                lg, masks = masked.record_masks(st, torch.from_numpy(xs).to(dtype), torch.from_numpy(ys), **kw)
                lv = O.cross_entropy(lg, torch.from_numpy(ys))
                gs = torch.autograd.grad(lv, [st[k] for k in keys])
                lo.append(float(lv))
                gd.append({k: a.detach().double() for k, a in zip(keys, gs)})
                mk.append(masks)
            return gd, keys, lo, mk

        g32, keys, lo32, masks32 = oracle_own(torch.float32)
        names = [n for n, _ in m.named_parameters()]
        assert names == keys
        # fp64 yardsticks of the SAME piecewise-linear function: the shard forwards replayed with the masks each path chose
        ref_hip = [masked.grads(npst, xs, ys, masks=mk, **kw)[1] for (xs, ys), mk in zip(shards, masks_hip)]
        ref_cpu = [masked.grads(npst, xs, ys, masks=mk, **kw)[1] for (xs, ys), mk in zip(shards, masks32)]
        hd = [hip_delta[o:o + p.numel()] for p, o in zip(m.parameters(), m._offsets)]
        for a, b in zip(losses, lo32):
            assert abs(a - b) < 2e-4
        fh = torch.cat(hd)
        f_ref_hip, f_ref_cpu, f32 = torch.cat(update(ref_hip, keys)), torch.cat(update(ref_cpu, keys)), torch.cat(update(g32, keys))
        e_oracle = float((f32 - f_ref_cpu).norm() / f_ref_cpu.norm())
        e_hip = float((fh - f_ref_hip).norm() / f_ref_hip.norm())
        print("DP update error vs the same-mask fp64 oracle: cpu-fp32 %.3e hip %.3e" % (e_oracle, e_hip))
        # no additive slack: the HIP update is within 3x the CPU fp32 path's own distance to fp64, both judged against the fp64
        # gradient of the function they differentiated (see tests/test_model_gpu.py::test_backward_parity)
        assert e_hip <= 3.0 * e_oracle, (e_hip, e_oracle)
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


def _rank_main(rank, world, port, out_path, gold_dir):
    """One data-parallel rank on device 0 (both ranks share the GPU; gloo carries the collectives - RCCL needs one GPU per
    rank): the real default multi-GPU path of scripts/train_resnet.py - GraphedTrainStep(segmented=True) + GradAllReducer on its
    communication stream + FlatSGD(grad_scale = 1 / world)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from pytorch_kaldi_resnet_amd.engine import GraphedTrainStep
    from pytorch_kaldi_resnet_amd.optim import FlatSGD
    from pytorch_kaldi_resnet_amd.parallel import GradAllReducer
    meta = json.load(open(os.path.join(gold_dir, "c1_r34_aam.json")))
    torch.manual_seed(100 + rank)
    m, _ = _build(meta)
    if rank != 0:                                # DDP constructor semantics: rank 0's weights must win
        with torch.no_grad():
            m.flat_parameters().mul_(1.5)
    red = GradAllReducer(m)
    red.broadcast_parameters(0)
    opt = FlatSGD(m, 1e-2, momentum=0.9, weight_decay=5e-4, grad_scale=1.0 / world)
    h = meta["batch"] // world
    step = GraphedTrainStep(m.engine(), h, meta["frames"], warmup=1, segmented=True)
    losses = []
    for s in range(3):
        x, y = W.make_input(meta["seed"] + 1 + s, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
        xs, ys = torch.from_numpy(x[rank * h:(rank + 1) * h]).cuda(), torch.from_numpy(y[rank * h:(rank + 1) * h]).cuda()
        loss, _, _ = step(xs, ys, red.on_stage_done)
        red.finish()
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    torch.save({"params": m.flat_parameters().cpu(), "losses": losses}, "%s.%d" % (out_path, rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_processes_graph_segments_and_reducer_match_the_two_shard_emulation(gold_dir, tmp_path):
    """VERDICT r02 missing #3: a REAL process group under the segmented-graph path.  Two rank processes (both on device 0,
    gloo) run three steps of segmented hipGraph replay + GradAllReducer (communication stream, events, finish()) +
    FlatSGD(grad_scale = 1/2); the parameters both ranks end with must equal - bit for bit: a two-term sum commutes - the
    single-process emulation (two shards through the eager engine, summed arenas, same SGD), and every rank's loss curve its
    shard's (reference: DistributedDataParallel, scripts/train_resnet.py:148-149,183-185)."""
    import socket

    import torch.multiprocessing as mp
    from pytorch_kaldi_resnet_amd.optim import FlatSGD
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    out = str(tmp_path / "rank")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out, gold_dir)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    got = [torch.load("%s.%d" % (out, r), weights_only=True) for r in range(2)]
    assert torch.equal(got[0]["params"], got[1]["params"])
    # emulation in this process
    meta = json.load(open(os.path.join(gold_dir, "c1_r34_aam.json")))
    m, _ = _build(meta)
    opt = FlatSGD(m, 1e-2, momentum=0.9, weight_decay=5e-4, grad_scale=0.5)
    h = meta["batch"] // 2
    shard_losses = [[], []]
    for s in range(3):
        x, y = W.make_input(meta["seed"] + 1 + s, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
        total = None
        for r in range(2):
            opt.zero_grad(set_to_none=True)
            loss, _, _ = m.engine().loss_and_grad(torch.from_numpy(x[r * h:(r + 1) * h]).cuda(), torch.from_numpy(y[r * h:(r + 1) * h]).cuda())
            shard_losses[r].append(float(loss))
            total = m.flat_grads().clone() if total is None else total + m.flat_grads()
        m.flat_grads().copy_(total)
        opt.step()
    for r in range(2):
        assert got[r]["losses"] == shard_losses[r], (r, got[r]["losses"], shard_losses[r])
    assert torch.equal(got[0]["params"], m.flat_parameters().cpu())
