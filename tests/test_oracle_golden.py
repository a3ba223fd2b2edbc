"""The CPU oracle against arrays recorded from the reference itself (tools/make_golden.py).

Tolerances: the oracle and the reference issue the same ATen CPU calls, so agreement is
expected to ~1e-6 relative; stated per assertion.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import spk_oracle as O
from oracle import weights as W

CASES = ["c1_r34_aam", "r34_aam_t203", "r34_aam_t300", "r34_softmax_mean_f40", "r34_aamv1_f40",
         "r101_aam"]


def close(a, b, rel):
    """max |a-b| <= rel * max|b| (scale-relative: same ATen calls, thread-order noise only)."""
    assert np.abs(a - b).max() <= rel * np.abs(b).max(), (np.abs(a - b).max(), np.abs(b).max())


def load_case(gold_dir, name):
    meta = json.load(open(os.path.join(gold_dir, name + ".json")))
    arrs = np.load(os.path.join(gold_dir, name + ".npz"))
    return meta, arrs


@pytest.mark.parametrize("loss,arch", [("AAM", "resnet34"), ("softmax", "resnet34"),
                                       ("AAM-v1", "resnet34"), ("AAM", "resnet101")])
def test_state_keys_match_reference(gold_dir, loss, arch):
    keys = json.load(open(os.path.join(gold_dir, "state_keys_%s_%s.json" % (arch, loss))))
    spec = W.state_spec(7, 80, "mean+std", loss, arch)
    assert [k for k, _, _ in spec] == [k for k, _ in keys]
    assert [list(s) for _, s, _ in spec] == [s for _, s in keys]
    assert len(keys) == {"AAM": 219, "softmax": 225, "AAM-v1": 224}[loss] or arch != "resnet34"


@pytest.mark.parametrize("name", CASES)
def test_forward_backward_vs_reference(gold_dir, name):
    meta, g = load_case(gold_dir, name)
    kw = dict(pooling=meta["pooling"], loss=meta["loss"], arch=meta["arch"])
    npst = W.make_state(meta["seed"], meta["spk_num"], meta["feat_dim"], **kw)
    x, y = W.make_input(meta["seed"] + 1, meta["batch"], meta["feat_dim"], meta["frames"],
                        meta["spk_num"])
    x, y = torch.from_numpy(x), torch.from_numpy(y)
    st = O.to_torch_state(npst)
    with torch.no_grad():
        emb = O.embed(st, x, meta["pooling"], meta["arch"], train=False).numpy()
        lg = O.forward(st, x, y, train=False, **kw).numpy()
    close(emb, g["emb_eval"], 1e-6)
    close(lg, g["logits_eval"], 1e-6)
    # training step
    bufs = {}
    lv, logits, grads = O.train_step(st, bufs, x, y, meta["lr"], weight_decay=meta["wd"], **kw)
    close(logits.numpy(), g["logits_train"], 1e-6)
    assert abs(lv - float(g["loss_train"])) <= 1e-5
    names = meta["param_names"]
    assert names == O.trainable_keys(st)
    for i, n in enumerate(names):
        gi = grads[n].reshape(-1).numpy()
        ref_norm = g["grad_norm"][i]
        # analytically-zero gradients (e.g. a bias in front of a BatchNorm) are pure rounding noise:
        # absolute floor 1e-5 on norms, and samples are judged against the tensor's rms gradient
        assert abs(np.sqrt((gi.astype(np.float64) ** 2).sum()) - ref_norm) <= 1e-4 * ref_norm + 1e-5, n
        idx = np.minimum((W.hash_uniform(77, i, 16) * gi.size).astype(np.int64), gi.size - 1)
        rms = ref_norm / np.sqrt(gi.size)
        assert np.abs(gi[idx] - g["grad_samples"][i]).max() <= 2e-3 * rms + 1e-6, n
    for key in g.files:
        if key.startswith("rm:"):
            np.testing.assert_allclose(st[key[3:] + ".running_mean"].numpy(), g[key], rtol=1e-5, atol=1e-6)
        if key.startswith("rv:"):
            np.testing.assert_allclose(st[key[3:] + ".running_var"].numpy(), g[key], rtol=1e-5, atol=1e-6)
        if key.startswith("nbt:"):
            assert int(st[key[4:] + ".num_batches_tracked"]) == int(g[key])
    if meta["steps"]:
        losses = [lv]
        for s in range(1, meta["steps"]):
            xs, ys = W.make_input(meta["seed"] + 1 + s, meta["batch"], meta["feat_dim"],
                                  meta["frames"], meta["spk_num"])
            l, _, _ = O.train_step(st, bufs, torch.from_numpy(xs), torch.from_numpy(ys),
                                   meta["lr"], weight_decay=meta["wd"], **kw)
            losses.append(l)
        np.testing.assert_allclose(losses, g["loss_curve"], rtol=0, atol=1e-3)  # see DESIGN.md "gradient conditioning"
        with torch.no_grad():
            e2 = O.embed(st, x, meta["pooling"], meta["arch"], train=False).numpy()
        cos = (e2 * g["emb_after"]).sum(1) / (np.linalg.norm(e2, axis=1) * np.linalg.norm(g["emb_after"], axis=1))
        assert (1 - cos).max() < 5e-4   # after SGD steps: fp32 gradient noise (DESIGN.md "gradient conditioning")


def test_pool_and_aam_kernels_vs_reference(gold_dir):
    g = np.load(os.path.join(gold_dir, "kernels.npz"))
    x = torch.from_numpy(g["pool_x"]).requires_grad_(True)
    for mode in ["mean", "mean+std"]:
        y = O.stats_pool(x, mode)
        np.testing.assert_allclose(y.detach().numpy(), g["pool_%s_y" % mode], rtol=1e-6, atol=1e-7)
        gx, = torch.autograd.grad((y * torch.from_numpy(g["pool_%s_gy" % mode])).sum(), x)
        np.testing.assert_allclose(gx.numpy(), g["pool_%s_gx" % mode], rtol=1e-5, atol=1e-7)
    e = torch.from_numpy(g["aam_e"]).requires_grad_(True)
    w = torch.from_numpy(g["aam_w"]).requires_grad_(True)
    lab = torch.from_numpy(g["aam_lab"])
    lg = O.aam_logits(e, w, lab, 0.2, 30.0)
    np.testing.assert_allclose(lg.detach().numpy(), g["aam_logits"], rtol=1e-6, atol=1e-5)
    ce = O.cross_entropy(lg, lab)
    assert abs(float(ce) - float(g["aam_loss"])) < 1e-5
    ge, gw = torch.autograd.grad(ce, [e, w])
    np.testing.assert_allclose(ge.numpy(), g["aam_ge"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gw.numpy(), g["aam_gw"], rtol=1e-4, atol=1e-6)


def test_scoring_vs_reference(gold_dir):
    d = os.path.join(gold_dir, "io")
    emb = {}
    for line in open(os.path.join(d, "emb.iv")):
        t = line.split()
        emb[t[0]] = np.array([float(v) for v in t[2:-1]])
    mean = np.array([float(v) for v in open(os.path.join(d, "mean.vec")).read().split()[1:-1]])
    trials, labels = [], []
    for line in open(os.path.join(d, "trials")):
        a, b, t = line.split()
        trials.append((a, b))
        labels.append(1 if t == "target" else 0)
    sc = O.cosine_scores(emb, emb, trials, mean)
    ref = [float(l.split()[2]) for l in open(os.path.join(d, "scores"))]
    np.testing.assert_allclose(sc, ref, rtol=1e-6, atol=1e-7)
    eer = O.compute_eer(ref, labels)
    assert "{0:.2%}".format(eer) == open(os.path.join(d, "eer.txt")).read().strip()


def test_cosine_lr_matches_torch_scheduler():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], 0.1)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 30, eta_min=1e-4)
    for e in range(30):
        assert abs(opt.param_groups[0]["lr"] - O.cosine_lr(e, 30, 0.1, 1e-4)) < 1e-9
        opt.step()
        sch.step()


def test_masked_oracle_reproduces_the_plain_oracle(gold_dir):
    """oracle/masked.py (the same-mask yardstick of the GPU gradient tests): replaying the forward with the masks it chose
    itself gives the same logits and the same gradient as the plain oracle - for both block kinds and the BN1d head."""
    from oracle import masked
    for arch, loss, pooling in (("resnet18", "AAM-v1", "mean+std"), ("resnet50", "softmax", "mean")):
        S, Fd, T, B = 5, 24, 40, 3
        npst = W.make_state(31, S, Fd, pooling, loss, arch)
        x, y = W.make_input(32, B, Fd, T, S)
        st = O.to_torch_state(npst)
        keys = O.trainable_keys(st)
        for k in keys:
            st[k].requires_grad_(True)
        lo, masks = masked.record_masks(st, torch.from_numpy(x), torch.from_numpy(y), pooling, loss, arch)
        g_plain = torch.autograd.grad(O.cross_entropy(lo, torch.from_numpy(y)), [st[k] for k in keys])
        nrelu = {"resnet18": 1 + 8 * 2, "resnet50": 1 + 16 * 3}[arch] + 1      # stem + per block + head BN1d/ReLU
        assert len(masks) == nrelu
        lv, g_masked = masked.grads(npst, x, y, pooling, loss, arch, masks, dtype=torch.float32)
        assert abs(lv - float(O.cross_entropy(lo, torch.from_numpy(y)))) < 1e-6
        for k, a in zip(keys, g_plain):
            assert float((g_masked[k] - a.double()).norm()) <= 1e-6 * float(a.norm()) + 1e-9, k
        # flipping one mask element changes the gradient: the masks really steer the backward
        masks[3] = masks[3].clone()
        idx = tuple(int(v[0]) for v in torch.nonzero(masks[3], as_tuple=True))
        masks[3][idx] = False
        _, g_flip = masked.grads(npst, x, y, pooling, loss, arch, masks, dtype=torch.float32)
        assert any(float((g_flip[k] - g_masked[k]).norm()) > 0 for k in keys)


def test_reference_sensitivity_fixture(gold_dir):
    """tests/golden/ref_sensitivity.json (tools/ref_sensitivity.py: the imported reference against itself) backs the
    loss-curve budget of the GPU tests: well formed, and it says what DESIGN.md says it says."""
    import json
    d = json.load(open(os.path.join(gold_dir, "ref_sensitivity.json")))
    for name in ("c1_r34_aam", "r34_softmax_mean_f40"):
        ent = d["cases"][name]
        rec = np.load(os.path.join(gold_dir, name + ".npz"))["loss_curve"]
        assert np.allclose(ent["recorded"], rec)
        n = len(rec)
        assert len(ent["fp64_minus_recorded"]) == n and len(ent["perturb_ulp"]["max_abs_dloss"]) == n
        assert ent["perturb_ulp"]["max_abs_dloss"][0] < 1e-4 and abs(ent["fp64_minus_recorded"][0]) < 1e-4
    c1 = d["cases"]["c1_r34_aam"]
    assert c1["perturb_ulp"]["max_abs_dloss"][4] > 1e-3      # the reference does not track itself to 1e-3 at step 4
