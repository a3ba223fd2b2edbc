#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE itself on CPU (build container only).

Imports /root/reference/scripts/model.py (+ kaldi_io.py, cosine_score/compute_mean/
compute_eer as subprocesses) unmodified, loads the closed-form weights of oracle/weights.py
into ``NeuralSpeakerModel`` through ``load_state_dict`` and records arrays: logits, loss,
embeddings, gradient norms + sampled gradient entries, BN running statistics, and a 5-step
SGD loss curve driven by a loop that follows scripts/train_resnet.py:304-328.
Only arrays / text data files are written; no reference source travels.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "scripts"))
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import weights as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.manual_seed(0)
torch.set_num_threads(8)


def ref_model(spk_num, feat_dim, pooling, loss, arch, seed):
    import model as refmodel
    m = refmodel.NeuralSpeakerModel(spk_num=spk_num, feat_dim=feat_dim, pooling=pooling,
                                    loss=loss, m=0.2, s=30)
    if arch == "resnet101":
        m.res = refmodel.resnet101()       # oracle for config 4
    elif arch != "resnet34":
        m.res = getattr(refmodel, arch)()
    st = W.make_state(seed, spk_num, feat_dim, pooling, loss, arch)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in st.items()}
    m.load_state_dict(sd, strict=True)
    return m


def sample_idx(n, stream, k=16):
    u = W.hash_uniform(77, stream, k)
    return np.minimum((u * n).astype(np.int64), n - 1)


def record_case(name, spk_num, feat_dim, frames, batch, pooling, loss, arch, seed, steps=0,
                lr=2e-5, wd=5e-4):
    print("case", name)
    m = ref_model(spk_num, feat_dim, pooling, loss, arch, seed)
    x_np, y_np = W.make_input(seed + 1, batch, feat_dim, frames, spk_num)
    x, y = torch.from_numpy(x_np), torch.from_numpy(y_np)
    out = {}
    # eval-mode embeddings and logits (scripts/decode.py:186,198; validate() :349,359)
    m.eval()
    with torch.no_grad():
        out["emb_eval"] = m.predict(x).numpy()
        out["logits_eval"] = m(x, y).numpy()
    # one training step following scripts/train_resnet.py:304-328
    m.train()
    crit = nn.CrossEntropyLoss()
    opt = torch.optim.SGD(m.parameters(), lr, momentum=0.9, weight_decay=wd)
    logits = m(x, y)
    lossv = crit(logits, y)
    opt.zero_grad()
    lossv.backward()
    out["logits_train"] = logits.detach().numpy()
    out["loss_train"] = np.array(float(lossv))
    names = [n for n, _ in m.named_parameters()]
    gnorm, gsamp = [], []
    for i, (n, p) in enumerate(m.named_parameters()):
        g = p.grad.detach().reshape(-1).numpy()
        gnorm.append(np.sqrt((g.astype(np.float64) ** 2).sum()))
        gsamp.append(g[sample_idx(g.size, i)])
    out["grad_norm"] = np.array(gnorm)
    out["grad_samples"] = np.stack(gsamp).astype(np.float32)
    # BN running statistics after the first forward
    sd = m.state_dict()
    for key in ["res.bn1", "res.layer1.0.bn1", "res.layer2.0.downsample.1", "res.layer4.2.bn2"
                if arch in ("resnet18", "resnet34") else "res.layer4.2.bn3"]:
        out["rm:" + key] = sd[key + ".running_mean"].numpy().copy()
        out["rv:" + key] = sd[key + ".running_var"].numpy().copy()
        out["nbt:" + key] = sd[key + ".num_batches_tracked"].numpy().copy()
    losses = [float(lossv)]
    if steps:
        opt.step()
        for s in range(1, steps):
            xs, ys = W.make_input(seed + 1 + s, batch, feat_dim, frames, spk_num)
            xs, ys = torch.from_numpy(xs), torch.from_numpy(ys)
            o = m(xs, ys)
            l = crit(o, ys)
            opt.zero_grad()
            l.backward()
            opt.step()
            losses.append(float(l))
        out["loss_curve"] = np.array(losses)
        m.eval()
        with torch.no_grad():
            out["emb_after"] = m.predict(x).numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    meta = dict(name=name, spk_num=spk_num, feat_dim=feat_dim, frames=frames, batch=batch,
                pooling=pooling, loss=loss, arch=arch, seed=seed, steps=steps, lr=lr, wd=wd,
                param_names=names, torch=torch.__version__, numpy=np.__version__)
    with open(os.path.join(GOLD, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)


def record_keys():
    import model as refmodel
    for loss in ["AAM", "softmax", "AAM-v1"]:
        m = refmodel.NeuralSpeakerModel(spk_num=7, feat_dim=80, pooling="mean+std", loss=loss)
        keys = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        with open(os.path.join(GOLD, "state_keys_resnet34_%s.json" % loss), "w") as f:
            json.dump(keys, f)
    m = refmodel.NeuralSpeakerModel(spk_num=7, feat_dim=80, pooling="mean+std", loss="AAM")
    m.res = refmodel.resnet101()
    keys = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    with open(os.path.join(GOLD, "state_keys_resnet101_AAM.json"), "w") as f:
        json.dump(keys, f)


def record_kernels():
    """Per-op fixtures from the reference's own layer classes (tiny shapes, fwd + bwd)."""
    import model as refmodel
    out = {}
    # StatsPooling both modes, with the swap (scripts/model.py:441-457)
    u = W.hash_uniform(5, 1, 2 * 4 * 3 * 13).reshape(2, 4, 3, 13).astype(np.float32) + 0.05
    xin = torch.from_numpy(u).requires_grad_(True)
    for mode in ["mean", "mean+std"]:
        y = refmodel.StatsPooling(mode)(xin)
        gy = torch.from_numpy(W.hash_uniform(5, 2, y.numel()).reshape(y.shape).astype(np.float32))
        gx, = torch.autograd.grad((y * gy).sum(), xin)
        out["pool_%s_y" % mode] = y.detach().numpy()
        out["pool_%s_gy" % mode] = gy.numpy()
        out["pool_%s_gx" % mode] = gx.numpy()
    out["pool_x"] = u
    # AAMLayer (scripts/model.py:459-501): rows engineered to hit both where-branches
    S, D, B = 11, 256, 6
    layer = refmodel.AAMLayer(in_feats=D, n_classes=S, m=0.2, s=30)
    wv = (W.hash_uniform(6, 1, S * D).reshape(S, D) * 2 - 1).astype(np.float32)
    ev = (W.hash_uniform(6, 2, B * D).reshape(B, D) * 2 - 1).astype(np.float32)
    lab = np.array([0, 3, 5, 7, 10, 2], dtype=np.int64)
    ev[1] = -3.0 * wv[3] + 0.03 * ev[1]  # cos ~ -0.999: (cos - th) <= 0 branch on the target column (exactly -1 is nan in torch)
    ev[2] = 2.0 * wv[5] + 0.03 * ev[2]   # cos ~ 0.999: small sine, large dphi/dcos (cos == 1 exactly is inf/nan in torch too)
    with torch.no_grad():
        layer.weight.copy_(torch.from_numpy(wv))
    e = torch.from_numpy(ev).requires_grad_(True)
    lg = layer(e, torch.from_numpy(lab))
    ce = nn.CrossEntropyLoss()(lg, torch.from_numpy(lab))
    ge, gw = torch.autograd.grad(ce, [e, layer.weight])
    out.update(aam_w=wv, aam_e=ev, aam_lab=lab, aam_logits=lg.detach().numpy(),
               aam_loss=np.array(float(ce)), aam_ge=ge.numpy(), aam_gw=gw.numpy())
    np.savez_compressed(os.path.join(GOLD, "kernels.npz"), **out)


def record_io():
    """ark/scp, text-ark embeddings, mean.vec, trials and the reference scoring outputs."""
    import kaldi_io
    d = os.path.join(GOLD, "io")
    os.makedirs(d, exist_ok=True)
    ark = os.path.join(d, "feats.ark")
    if os.path.exists(ark):
        os.remove(ark)
    lines = []
    utts = ["spk%d-utt%d" % (s, u) for s in range(3) for u in range(2)]
    with open(ark, "wb") as f:
        for i, utt in enumerate(utts):
            T = 20 + 3 * i
            mat = (W.hash_uniform(9, i, T * 8).reshape(T, 8) * 4 - 2).astype(np.float32)
            f.write((utt + " ").encode())
            off = f.tell()
            kaldi_io.write_mat(f, mat)
            lines.append("%s tests/golden/io/feats.ark:%d" % (utt, off))
    open(os.path.join(d, "feats.scp"), "w").write("\n".join(lines) + "\n")
    # text-ark embeddings in the format of scripts/decode.py:206
    emb = {}
    with open(os.path.join(d, "emb.iv"), "w") as f:
        for i, utt in enumerate(utts):
            v = (W.hash_uniform(10, i, 16) * 2 - 1).astype(np.float32)
            v += (i // 2) * 0.7
            emb[utt] = v
            f.write(utt + " [ " + " ".join(map(str, v)) + " ]\n")
    trials = []
    for i, a in enumerate(utts):
        for b in utts[i + 1:]:
            trials.append("%s %s %s" % (a, b, "target" if a.split("-")[0] == b.split("-")[0]
                                          else "nontarget"))
    open(os.path.join(d, "trials"), "w").write("\n".join(trials) + "\n")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    sc = os.path.join(REF, "scripts")
    subprocess.check_call([sys.executable, os.path.join(sc, "compute_mean.py"),
                           os.path.join(d, "emb.iv"), os.path.join(d, "mean.vec")], env=env, cwd=sc)
    subprocess.check_call([sys.executable, os.path.join(sc, "cosine_score.py"),
                           "--mean", os.path.join(d, "mean.vec"),
                           "--enroll", os.path.join(d, "emb.iv"), "--test", os.path.join(d, "emb.iv"),
                           "--trials", os.path.join(d, "trials"),
                           "--score-file", os.path.join(d, "scores")], env=env, cwd=sc)
    eer = subprocess.check_output([sys.executable, os.path.join(sc, "compute_eer.py"),
                                   os.path.join(d, "scores"), os.path.join(d, "trials")],
                                  env=env, cwd=sc, stderr=subprocess.DEVNULL)
    open(os.path.join(d, "eer.txt"), "w").write(eer.decode())
    # adaptive S-norm (scripts/compute_topk_mean_std.py + scripts/adaptive_snorm.py of the reference): a 320-vector cohort
    # so that the reference's hard-wired top-300 selection is exercised
    with open(os.path.join(d, "cohort.iv"), "w") as f:
        for i in range(320):
            v = (W.hash_uniform(12, i, 16) * 2 - 1).astype(np.float32)
            v += (i % 5) * 0.35
            f.write("coh%03d [ " % i + " ".join(map(str, v)) + " ]\n")
    subprocess.check_call([sys.executable, os.path.join(sc, "compute_topk_mean_std.py"),
                           "--mean", os.path.join(d, "mean.vec"), "--ark-file", os.path.join(d, "emb.iv"),
                           "--cohort-file", os.path.join(d, "cohort.iv"),
                           "--mean-std-file", os.path.join(d, "topk_mean_std")], env=env, cwd=sc,
                          stdout=subprocess.DEVNULL)
    subprocess.check_call([sys.executable, os.path.join(sc, "adaptive_snorm.py"),
                           "--enroll", os.path.join(d, "topk_mean_std"), "--test", os.path.join(d, "topk_mean_std"),
                           "--score-in", os.path.join(d, "scores"), "--score-out", os.path.join(d, "scores_snorm")],
                          env=env, cwd=sc, stdout=subprocess.DEVNULL)
    # matrices as read back by the reference reader
    mats = {u: kaldi_io.read_mat(l.split()[1].replace("tests/golden/io", d))
            for u, l in zip(utts, lines)}
    np.savez_compressed(os.path.join(d, "feats_expected.npz"), **mats)


if __name__ == "__main__":
    what = sys.argv[1:] or ["keys", "kernels", "io", "cases"]
    if "keys" in what:
        record_keys()
    if "kernels" in what:
        record_kernels()
    if "io" in what:
        record_io()
    if "cases" in what:
        record_case("c1_r34_aam", 10, 80, 200, 4, "mean+std", "AAM", "resnet34", 11, steps=5)
        record_case("r34_aam_t203", 10, 80, 203, 2, "mean+std", "AAM", "resnet34", 12)
        record_case("r34_aam_t300", 1211, 80, 300, 2, "mean+std", "AAM", "resnet34", 13)
        record_case("r34_softmax_mean_f40", 9, 40, 120, 3, "mean", "softmax", "resnet34", 14, steps=3)
        record_case("r34_aamv1_f40", 9, 40, 96, 3, "mean+std", "AAM-v1", "resnet34", 15)
        record_case("r101_aam", 12, 80, 200, 2, "mean+std", "AAM", "resnet101", 16, steps=2)
