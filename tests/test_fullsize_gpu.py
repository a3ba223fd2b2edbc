"""The default operand mode (f16x3) at the size the headline is quoted on - BASELINE configs[1]: ResNet-34 + AAM, 1211
speakers, batch 256 x 300 frames x 80 mel - against the EXACT split mode (bf16x6: every fp32 operand as the exact sum of three
bf16 terms, no scales) from identical weights and inputs (reference step: scripts/train_resnet.py:316-328).

VERDICT r02 item 1: every gradient-level parity check ran at B = 2..4; this file runs the full-size training step in both
modes and bounds the relative difference of the loss, the logits and the gradient arena per ResNet stage, counts the
staged values against the fp16 windows of their scale slots (saturation must be 0: every scale is an absmax or a rigorous
bound), and repeats the comparison after 20 SGD steps at lr 0.1.  The oracle does not run at this size (a CPU step of the
reference path at batch 256 takes minutes); the small-batch tests in test_model_gpu.py pin both modes to the oracle, this
one pins the modes to each other where the headline is measured.

Also here: the lifetime of the operand-scale slots between a training forward and its backward."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import weights as W  # noqa: E402

B, FEAT, FRAMES, SPK = 256, 80, 300, 1211


@pytest.fixture(scope="module")
def P():
    assert torch.cuda.is_available()
    import pytorch_kaldi_resnet_amd as pkg
    return pkg


def _model(spk=SPK, seed=5):
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    m = NeuralSpeakerModel(spk, FEAT, "mean+std", "AAM", 0.2, 30, arch="resnet34")
    npst = W.make_state(seed, spk, FEAT, "mean+std", "AAM", "resnet34")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in npst.items()}, strict=True)
    return m.cuda().train()


def _stage_rel(model, ga, gb):
    from pytorch_kaldi_resnet_amd.parallel import stage_slices
    out = {}
    for name, (lo, hi) in stage_slices(model).items():
        a, b = ga[lo:hi].double(), gb[lo:hi].double()
        out[name] = float((a - b).norm() / b.norm())
    return out


def test_full_size_step_f16x3_against_the_exact_split_mode(P):
    from pytorch_kaldi_resnet_amd import ops
    from pytorch_kaldi_resnet_amd.optim import FlatSGD
    m = _model()
    eng = m.engine()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234)
    x = torch.randn(B, FEAT, FRAMES, device="cuda", generator=gen)
    y = torch.randint(0, SPK, (B,), device="cuda", generator=gen)
    p0 = m.flat_parameters().clone()
    buf0 = [b.clone() for b in m.buffers()]
    old, old_bwd = ops.SPLIT, ops.SPLIT_BWD
    res = {}

    def run(fwd, bwd=None, count=False):
        ops.SPLIT, ops.SPLIT_BWD = ops.MFMA_MODES[fwd], (ops.MFMA_MODES[bwd] if bwd else None)
        eng.dirty = True
        for b, b0 in zip(m.buffers(), buf0):
            b.copy_(b0)
        for p in m.parameters():
            p.grad = None
        if count:
            eng.window_counts = torch.zeros(4, device="cuda", dtype=torch.int64)
        loss, logits, rank = eng.loss_and_grad(x, y)
        torch.cuda.synchronize()
        counts = None
        if count:
            counts, eng.window_counts = eng.window_counts.tolist(), None
        return float(loss), logits.clone(), m.flat_grads().clone(), counts

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())

    try:
        # (1) SAME forward (exact split mode: same ReLU masks, same BatchNorm statistics, same saved tensors), backward in each
        # operand mode: what differs is the rounding of the backward arithmetic alone
        la, lga, g_ex, _ = run("bf16x6")
        _, _, g_h, counts_bwd = run("bf16x6", "f16x3", count=True)
        _, _, g_f, _ = run("bf16x6", "f32")
        same_h, same_f = rel(g_h, g_ex), rel(g_f, g_ex)
        st_h, st_f = _stage_rel(m, g_h, g_ex), _stage_rel(m, g_f, g_ex)
        print("same forward, backward f16x3 vs bf16x6: gradient arena rel %.2e   (native fp32 instruction vs bf16x6: %.2e)" % (same_h, same_f))
        print("  per stage f16x3: " + ", ".join("%s %.2e" % kv for kv in sorted(st_h.items())))
        print("  per stage f32  : " + ", ".join("%s %.2e" % kv for kv in sorted(st_f.items())))
        # (2) free-running: forward AND backward in each mode (ReLU masks of activations within rounding of zero and the
        # BatchNorm statistics now differ too - the conditioning of the map, tests/test_model_gpu.py)
        lb, lgb, g_hh, counts = run("f16x3", count=True)
        lc, lgc, g_ff, _ = run("f32")
        d_loss = abs(la - lb) / abs(la)
        d_logits = rel(lgb, lga)
        free_h, free_f = rel(g_hh, g_ex), rel(g_ff, g_ex)
        print("free-running step, f16x3 vs bf16x6: loss %.6f vs %.6f (rel %.2e), logits rel %.2e (f32: %.2e), gradient arena rel "
              "%.2e (f32 vs bf16x6: %.2e)" % (lb, la, d_loss, d_logits, rel(lgc, lga), free_h, free_f))
        print("  per stage: " + ", ".join("%s %.2e" % kv for kv in sorted(_stage_rel(m, g_hh, g_ex).items())))
        total, sat, lo_lost, hi_sub = counts
        print("  f16 windows over %d staged values (forward + backward): %d saturated, %.4f %% low term subnormal, %.4f %% high term "
              "subnormal; backward only: %d values, %d saturated" % (total, sat, 100.0 * lo_lost / total, 100.0 * hi_sub / total,
                                                                    counts_bwd[0], counts_bwd[1]))
        assert sat == 0 and counts_bwd[1] == 0 and total > 1e9
        assert d_loss < 2e-6, d_loss
        assert d_logits < 2e-5, d_logits
        # the two-term fp16 form is as close to the exact split as the native fp32 instruction is (bounds: 2x its distance)
        assert same_h <= 2.0 * same_f + 1e-7, (same_h, same_f)
        for name in st_h:
            assert st_h[name] <= 2.0 * st_f[name] + 1e-7, (name, st_h[name], st_f[name])
        assert free_h <= 2.0 * free_f + 1e-7, (free_h, free_f)
        # ---- 20 SGD steps at lr 0.1 from the same start, same batch: trajectories stay together
        traj = {}
        for mode in ("bf16x6", "f16x3"):
            ops.SPLIT, ops.SPLIT_BWD = ops.MFMA_MODES[mode], None
            eng.dirty = True
            m.flat_parameters().copy_(p0)
            m.mark_weights_changed()
            for b, b0 in zip(m.buffers(), buf0):
                b.copy_(b0)
            opt = FlatSGD(m, 0.1, momentum=0.9, weight_decay=5e-4)
            ls = []
            for _ in range(20):
                opt.zero_grad(set_to_none=True)
                loss, _, _ = eng.loss_and_grad(x, y)
                opt.step()
                ls.append(float(loss))
            traj[mode] = (ls, m.flat_parameters().clone())
        (la, pa), (lb, pb) = traj["bf16x6"], traj["f16x3"]
        d_par = float((pa.double() - pb.double()).norm() / (pa.double() - p0.double()).norm())
        d_l = max(abs(a - b) / abs(a) for a, b in zip(la, lb))
        print("20 SGD steps (lr 0.1): losses f16x3 %s" % ["%.4f" % v for v in lb])
        print("                       losses bf16x6 %s" % ["%.4f" % v for v in la])
        print("  max relative loss difference %.2e, parameter displacement difference %.2e of the distance travelled" % (d_l, d_par))
        assert lb[-1] < lb[0] and la[-1] < la[0]
        assert d_l < 5e-3, d_l
        assert d_par < 5e-2, d_par
    finally:
        ops.SPLIT, ops.SPLIT_BWD = old, old_bwd


def test_operand_scale_slots_survive_an_eval_forward_between_forward_and_backward(P, gold_dir):
    """ADVICE r02 (engine.py): the training forward's absmax slots are read again by its backward (weight-gradient X operands,
    the |xhat| bound).  An eval-mode / no-grad forward in between used to reset that table (wrong power-of-two scales, no
    error).  It now has a table of its own, and a second TRAINING forward before the backward is refused."""
    from pytorch_kaldi_resnet_amd import ops
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    meta = json.load(open(os.path.join(gold_dir, "c1_r34_aam.json")))
    npst = W.make_state(meta["seed"], meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], meta["arch"])
    x, y = W.make_input(meta["seed"] + 1, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
    x2, _ = W.make_input(meta["seed"] + 9, 3, meta["feat_dim"], 117, meta["spk_num"])
    xg, yg, x2g = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(x2 * 50.0).cuda()   # very different absmax
    assert ops.SPLIT == ops.MFMA_MODES["f16x3"]
    grads = []
    for interleave in (False, True):
        m = NeuralSpeakerModel(meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], 0.2, 30, arch=meta["arch"])
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in npst.items()})
        m = m.cuda().train()
        logits = m(xg, yg)
        if interleave:
            m.eval()
            with torch.no_grad():
                m.predict(x2g)
            m.train()
            with torch.no_grad():
                m.predict(x2g)                  # train-mode statistics, no autograd: also the other table
        torch.nn.functional.cross_entropy(logits, yg).backward()
        grads.append(m.flat_grads().clone())
    assert torch.equal(grads[0], grads[1])
    logits = m(xg, yg)
    m(xg, yg)                                   # a second training forward reuses the rows and slots of the first
    with pytest.raises(RuntimeError, match="reused"):
        torch.nn.functional.cross_entropy(logits, yg).backward()


def test_dead_channel_rows_at_the_pooling_input_do_not_poison_the_operand_scales(P, gold_dir):
    """Found on a TRAINED checkpoint (tools/window_on_checkpoint.py): a channel row of the last block's output that is zero over
    the whole utterance has mean 0, the pooling layer's sqrt'(0) puts inf / NaN into the gradient there (the reference does the
    same: torch.sqrt backward, scripts/model.py:453), and the ReLU mask select drops them one step later.  In f16x3 the absmax
    hand-off of that gradient used to become inf -> operand scale 1 for the whole tensor -> gradients of 1e-6 carried as fp16
    subnormals: 40 % error in layer 4.  Here: three dead output channels in every layer-4 block; the backward pass in f16x3
    must stay finite and agree with the exact split mode on the same forward like the native fp32 instruction does."""
    from pytorch_kaldi_resnet_amd import ops
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    meta = json.load(open(os.path.join(gold_dir, "c1_r34_aam.json")))
    npst = W.make_state(meta["seed"], meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], meta["arch"])
    sd = {k: torch.from_numpy(np.array(v)) for k, v in npst.items()}
    dead = [5, 77, 200]
    for blk in range(3):                       # relu(bn2(.) + shortcut) == 0 needs the whole residual chain of the channel dead
        for key in ("res.layer4.%d.bn2" % blk,) + (("res.layer4.0.downsample.1",) if blk == 0 else ()):
            sd[key + ".weight"][dead] = 0.0
            sd[key + ".bias"][dead] = -1.0
    m = NeuralSpeakerModel(meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], 0.2, 30, arch=meta["arch"])
    m.load_state_dict(sd)
    m = m.cuda().train()
    eng = m.engine()
    x, y = W.make_input(meta["seed"] + 1, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
    xg, yg = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    buf0 = [b.clone() for b in m.buffers()]
    old, old_bwd = ops.SPLIT, ops.SPLIT_BWD
    grads = {}
    try:
        for bwd in ("bf16x6", "f16x3", "f32"):
            ops.SPLIT, ops.SPLIT_BWD = ops.MFMA_MODES["bf16x6"], ops.MFMA_MODES[bwd]
            eng.dirty = True
            for b, b0 in zip(m.buffers(), buf0):
                b.copy_(b0)
            for p in m.parameters():
                p.grad = None
            if bwd == "f16x3":
                eng.window_counts = torch.zeros(4, device="cuda", dtype=torch.int64)
            loss, _, _ = eng.loss_and_grad(xg, yg)
            torch.cuda.synchronize()
            if bwd == "f16x3":
                counts, eng.window_counts = eng.window_counts.tolist(), None
            grads[bwd] = m.flat_grads().clone()
            assert bool(torch.isfinite(grads[bwd]).all()) and np.isfinite(float(loss)), bwd
    finally:
        ops.SPLIT, ops.SPLIT_BWD = old, old_bwd
    assert counts[1] == 0
    ref = grads["bf16x6"].double()
    e_h = float((grads["f16x3"].double() - ref).norm() / ref.norm())
    e_f = float((grads["f32"].double() - ref).norm() / ref.norm())
    print("dead pooling rows: same-forward backward vs exact split: f16x3 %.2e, native fp32 instruction %.2e" % (e_h, e_f))
    assert e_h <= 2.0 * e_f + 1e-6, (e_h, e_f)
    # the dead channels really produce non-finite pooling gradients (the case under test is exercised)
    with torch.no_grad():
        _, saved = eng.forward_train(xg, yg)
        feat = saved["feat"]
        assert float(feat[..., dead].abs().max()) == 0.0
        d = ops.stats_pool_bwd(feat, torch.ones(feat.shape[0], feat.shape[3] * feat.shape[1] * 2, device="cuda"), 1)
        assert not bool(torch.isfinite(d).all())


def test_tiny_pooled_means_keep_the_last_stage_gradients_in_their_fp16_windows(P, gold_dir):
    """VERDICT r03 item 4.  A channel row of the last block's output whose mean over time is tiny but NOT zero makes the pooling
    layer's sqrt'(mean) (reference scripts/model.py:450-454: torch.sqrt backward) emit a huge, finite gradient - 1e5 / 1e9 / 1e14
    times the rest for means of 1e-12 / 1e-20 / 1e-30 - that is not masked (the row has a positive element) and travels down
    the identity shortcuts of the last stage as `dout`.  The reference carries every other element with full fp32 precision.
    In f16x3 the BatchNorm-backward scale bound used to pair the tensor-wide absmax of dout with the layer's largest
    gamma*invstd: the bound overshot the values by the outlier ratio and the pair tensors of the whole last stage fell out of
    their fp16 windows.  Now the last stage's bounds pair every channel's own absmax with its own gamma*invstd
    (Engine.chan_amax, spk_bn_bwd_reduce chan_amax).

    Construction: three layer-4 channels whose whole residual chain (bn2 of every block + the downsample BatchNorm) is scaled
    by 1e-12 / 1e-20 / 1e-30 (weight and bias), so their block outputs - and row means - are that small and positive.  Same
    forward (exact split mode), backward in bf16x6 (exact split: the yardstick) / f16x3 / the native fp32 instruction.
    Criterion as everywhere else: f16x3's distance to the exact split <= 2 x the native instruction's, per stage - and, sharper,
    per parameter tensor of the last stage (4 x: these are 256-element tensors) - and the same run with the per-channel bound
    switched off must be visibly worse (the test has teeth)."""
    from pytorch_kaldi_resnet_amd import ops
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    meta = json.load(open(os.path.join(gold_dir, "c1_r34_aam.json")))
    npst = W.make_state(meta["seed"], meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], meta["arch"])
    sd = {k: torch.from_numpy(np.array(v)) for k, v in npst.items()}
    tiny = {5: 1e-12, 77: 1e-20, 200: 1e-30}
    for blk in range(3):
        for key in ("res.layer4.%d.bn2" % blk,) + (("res.layer4.0.downsample.1",) if blk == 0 else ()):
            for c, eps in tiny.items():
                sd[key + ".weight"][c] = abs(float(sd[key + ".weight"][c])) * eps
                sd[key + ".bias"][c] = (abs(float(sd[key + ".bias"][c])) + 0.5) * eps        # positive: the rows are not dead
    m = NeuralSpeakerModel(meta["spk_num"], meta["feat_dim"], meta["pooling"], meta["loss"], 0.2, 30, arch=meta["arch"])
    m.load_state_dict(sd)
    m = m.cuda().train()
    eng = m.engine()
    x, y = W.make_input(meta["seed"] + 1, meta["batch"], meta["feat_dim"], meta["frames"], meta["spk_num"])
    xg, yg = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    buf0 = [b.clone() for b in m.buffers()]
    old, old_bwd, old_chan = ops.SPLIT, ops.SPLIT_BWD, eng.chan_amax
    grads, counts = {}, {}
    try:
        for tag, bwd, chan in (("exact", "bf16x6", True), ("f16x3", "f16x3", True), ("f32", "f32", True),
                               ("f16x3_tensor_wide_bound", "f16x3", False)):
            ops.SPLIT, ops.SPLIT_BWD = ops.MFMA_MODES["bf16x6"], ops.MFMA_MODES[bwd]
            eng.chan_amax = chan
            eng.dirty = True
            for b, b0 in zip(m.buffers(), buf0):
                b.copy_(b0)
            for p in m.parameters():
                p.grad = None
            if bwd == "f16x3":
                eng.window_counts = torch.zeros(4, device="cuda", dtype=torch.int64)
            loss, _, _ = eng.loss_and_grad(xg, yg)
            torch.cuda.synchronize()
            if bwd == "f16x3":
                counts[tag], eng.window_counts = eng.window_counts.tolist(), None
            grads[tag] = m.flat_grads().clone()
            assert bool(torch.isfinite(grads[tag]).all()) and np.isfinite(float(loss)), tag
    finally:
        ops.SPLIT, ops.SPLIT_BWD, eng.chan_amax = old, old_bwd, old_chan
    # the case under test is exercised: tiny positive row means, huge finite pooling gradients in those channels only
    with torch.no_grad():
        ops.SPLIT = ops.MFMA_MODES["bf16x6"]
        try:
            eng.dirty = True
            _, saved = eng.forward_train(xg, yg)
            feat = saved["feat"]
            d = ops.stats_pool_bwd(feat, torch.ones(feat.shape[0], feat.shape[3] * feat.shape[1] * 2, device="cuda"), 1)
        finally:
            ops.SPLIT = old
            eng.dirty = True
        row_mean = feat.mean(dim=2)                                           # [B][H][C]
        for c, eps in tiny.items():
            mc = row_mean[..., c]
            assert float(mc.max()) < 10 * eps and float(mc.max()) > 0.0, (c, float(mc.max()))
        others = [c for c in range(feat.shape[3]) if c not in tiny]
        ratio = float(d[..., 200].abs().max() / d[..., others].abs().max())
        assert bool(torch.isfinite(d[..., list(tiny)]).all()) and ratio > 1e10, ratio
    assert counts["f16x3"][1] == 0 and counts["f16x3_tensor_wide_bound"][1] == 0          # a bound never saturates, loose or tight
    ref = grads["exact"]
    st = {t: _stage_rel(m, grads[t], ref) for t in ("f16x3", "f32", "f16x3_tensor_wide_bound")}
    print("tiny pooled means, same-forward backward vs exact split, per stage:")
    for name in st["f32"]:
        print("  %-7s f16x3 %.2e   native fp32 %.2e   f16x3 with the tensor-wide bound %.2e" % (
            name, st["f16x3"][name], st["f32"][name], st["f16x3_tensor_wide_bound"][name]))
    print("  high-term-subnormal share of the staged values: %.3e (per-channel bound) vs %.3e (tensor-wide bound)" % (
        counts["f16x3"][3] / counts["f16x3"][0], counts["f16x3_tensor_wide_bound"][3] / counts["f16x3_tensor_wide_bound"][0]))
    for name in st["f32"]:
        assert st["f16x3"][name] <= 2.0 * st["f32"][name] + 1e-6, (name, st["f16x3"][name], st["f32"][name])
    worst_old = 0.0
    offs = m._offsets
    for (name, p), o in zip(m.named_parameters(), offs):
        if not name.startswith("res.layer4."):
            continue
        sl = slice(o, o + p.numel())
        r = ref[sl].double()
        e = {t: float((grads[t][sl].double() - r).norm() / r.norm()) for t in ("f16x3", "f32", "f16x3_tensor_wide_bound")}
        worst_old = max(worst_old, e["f16x3_tensor_wide_bound"] / max(e["f32"], 1e-7))
        assert e["f16x3"] <= 4.0 * e["f32"] + 5e-6, (name, e)
    print("  worst last-stage parameter tensor with the tensor-wide bound: %.1f x the native instruction's distance" % worst_old)
    assert worst_old > 20.0, worst_old


def test_weight_gradients_on_a_side_branch_of_the_captured_step():
    """SPK_GRAPH_SIDE=1 (GraphedTrainStep(side_stream=True)): the weight gradients are captured on a second stream.  The tensors
    they read are released by the host while the capture goes on, and a block freed during a capture is handed to the next
    allocation of the main branch - unless the side stream is recorded as a user of it.  Before that was done inside captures
    too, this configuration gave a different loss on every run at bench size.  Here: batch 64 x 200 frames, three SGD steps,
    parameters bit-identical to the single-stream capture."""
    from pytorch_kaldi_resnet_amd.engine import GraphedTrainStep
    from pytorch_kaldi_resnet_amd.optim import FlatSGD
    B, T = 64, 200
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    batches = [(torch.randn(B, FEAT, T, device="cuda", generator=gen), torch.randint(0, 64, (B,), device="cuda", generator=gen))
               for _ in range(3)]
    finals, losses = [], []
    for side in (False, True):
        m = _model(spk=64)
        opt = FlatSGD(m, 0.05, momentum=0.9, weight_decay=5e-4)
        step = GraphedTrainStep(m.engine(), B, T, warmup=1, side_stream=side)
        ls = []
        for x, y in batches:
            loss, _, _ = step(x, y)
            opt.step()
            ls.append(float(loss))
        torch.cuda.synchronize()
        finals.append(m.flat_parameters().clone())
        losses.append(ls)
        del step, opt, m
    assert losses[0] == losses[1], losses
    assert torch.equal(finals[0], finals[1])
