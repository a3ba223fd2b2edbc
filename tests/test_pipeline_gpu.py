"""BASELINE configs[0]-shaped plumbing on the GPU: synthetic fbank ark/scp -> SequenceDataset/DataLoader ->
scripts/train_resnet.py (2 epochs) -> checkpoint -> scripts/decode.py -> compute_mean / cosine_score /
compute_eer, with the extracted embeddings compared against the CPU oracle loaded from the same checkpoint
(cosine >= 1 - 1e-4, the BASELINE bar) and the EER compared against the oracle's embeddings' EER."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_data(d, n_spk=5, per_spk=12, feat=80, frames=208):
    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd import kaldi_io
    rs = np.random.RandomState(1234)
    spk_mean = 0.5 * rs.randn(n_spk, feat).astype(np.float32)
    scp, u2s = [], []
    with open(os.path.join(d, "feats.ark"), "wb") as f:
        for s in range(n_spk):
            for u in range(per_spk):
                utt = "spk%02d-utt%02d" % (s, u)
                T = frames if u < per_spk - 2 else frames + 24      # decode set: equal-length batches + a different length
                mat = (rs.randn(T, feat).astype(np.float32) + spk_mean[s])
                off = kaldi_io.write_mat(f, mat, key=utt)
                scp.append("%s %s:%d" % (utt, os.path.join(d, "feats.ark"), off))
                u2s.append("%s %d" % (utt, s))
    open(os.path.join(d, "train.scp"), "w").write("\n".join(scp[:-10]) + "\n")
    open(os.path.join(d, "cv.scp"), "w").write("\n".join(scp[-10:]) + "\n")
    open(os.path.join(d, "decode.scp"), "w").write("\n".join(scp) + "\n")
    open(os.path.join(d, "utt2spkid"), "w").write("\n".join(u2s) + "\n")
    return n_spk


def test_train_decode_score_pipeline(tmp_path):
    d = str(tmp_path)
    n_spk = _make_data(d)
    env = dict(os.environ, PYTHONPATH=ROOT)
    log = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train_resnet.py"), "--gpu", "0", "--workers", "0",
                          "--batch-size", "16", "--print-freq", "1", "--arch", "resnet34", "--input-dim", "80",
                          "--loss-type", "AAM", "--pooling", "mean+std", "--margin", "0.2", "--scale", "30", "--epochs", "2",
                          "--lr", "0.01", "--lr-final", "0.001", "--wd", "5e-4", "--max-chunk-size", "200",
                          "--train-list", os.path.join(d, "train.scp"), "--cv-list", os.path.join(d, "cv.scp"),
                          "--spk-num", str(n_spk), "--utt2spkid", os.path.join(d, "utt2spkid"), "--seed", "7",
                          "--native-reader", "--log-dir", os.path.join(d, "exp")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert log.returncode == 0, log.stdout[-3000:] + log.stderr[-3000:]
    assert "Epoch: [1][" in log.stdout and " * Acc@1 " in log.stdout
    # model_best.pth.tar only appears once cv Acc@1 > 0 (reference semantics: is_best = acc1 > best_acc1 = 0)
    ckpt_path = os.path.join(d, "exp", "checkpoint_epoch1.pth.tar")
    assert os.path.exists(ckpt_path) and os.path.exists(os.path.join(d, "exp", "checkpoint_epoch0.pth.tar"))
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    assert set(ckpt) == {"epoch", "arch", "state_dict", "best_acc1", "optimizer"} and len(ckpt["state_dict"]) == 219
    dec = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "decode.py"), "--gpu", "0", "--workers", "0",
                          "--batch-size", "1", "--chunk-size", "-1", "--spk_num", str(n_spk), "--arch", "resnet34",
                          "--input-dim", "80", "--pooling", "mean+std", "--model-path", ckpt_path,
                          "--decode-scp", os.path.join(d, "decode.scp"), "--out-path", os.path.join(d, "emb")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert dec.returncode == 0, dec.stdout[-3000:] + dec.stderr[-3000:]
    iv = os.path.join(d, "emb", "alone")
    from pytorch_kaldi_resnet_amd import kaldi_io, scoring
    emb = scoring.read_embeddings(iv)
    # the native, length-bucketed reader at batch 8 (text and binary output) must give the same embeddings
    for fmt in ("text", "fv"):
        dec2 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "decode.py"), "--gpu", "0", "--workers", "2",
                               "--batch-size", "8", "--chunk-size", "-1", "--spk_num", str(n_spk), "--arch", "resnet34",
                               "--input-dim", "80", "--pooling", "mean+std", "--model-path", ckpt_path, "--native-reader",
                               "--out-format", fmt, "--decode-scp", os.path.join(d, "decode.scp"),
                               "--out-path", os.path.join(d, "emb_" + fmt)], env=env, capture_output=True, text=True, timeout=600)
        assert dec2.returncode == 0, dec2.stdout[-3000:] + dec2.stderr[-3000:]
        emb2 = scoring.read_embeddings(os.path.join(d, "emb_" + fmt, "alone"))
        assert sorted(emb2) == sorted(emb)
        for k in emb:
            a, b = np.asarray(emb[k], dtype=np.float64), np.asarray(emb2[k], dtype=np.float64)
            # batch composition changes which tile/summation order a sample sees only through BN-free eval kernels: identical
            assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max(), k
    assert len(emb) == sum(1 for _ in open(os.path.join(d, "decode.scp")))
    # the CPU oracle on the same checkpoint
    from oracle import spk_oracle as O
    st = {k: v.clone() for k, v in ckpt["state_dict"].items()}
    o_emb = {}
    with torch.no_grad():
        for line in open(os.path.join(d, "decode.scp")):
            utt, rx = line.split()
            x = torch.from_numpy(np.ascontiguousarray(kaldi_io.read_mat(rx).T))[None]
            o_emb[utt] = O.embed(st, x, "mean+std", "resnet34", train=False)[0].numpy().astype(np.float64)
    worst = 0.0
    for utt, v in emb.items():
        a, b = np.asarray(v), o_emb[utt]
        worst = max(worst, 1 - float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))))
    print("max 1-cos HIP vs oracle over %d utts: %.3e" % (len(emb), worst))
    assert worst < 1e-4
    # scoring scripts
    trials = os.path.join(d, "trials")
    utts = sorted(emb)
    with open(trials, "w") as f:
        for i, a in enumerate(utts):
            for b in utts[i + 1:]:
                f.write("%s %s %s\n" % (a, b, "target" if a[:5] == b[:5] else "nontarget"))
    for cmd in (["compute_mean.py", iv, os.path.join(d, "mean.vec")],
                ["cosine_score.py", "--mean", os.path.join(d, "mean.vec"), "--enroll", iv, "--test", iv, "--trials", trials,
                 "--score-file", os.path.join(d, "scores")]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", cmd[0])] + cmd[1:], env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "compute_eer.py"), os.path.join(d, "scores"), trials],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("%")
    mean_o = np.mean(np.stack([o_emb[u] for u in utts]).astype(np.float32), axis=0)
    sc_o, lab = scoring.cosine_score(o_emb, o_emb, trials, mean_o)
    eer_o = scoring.compute_eer(sc_o, lab)
    eer_h = float(r.stdout.strip().rstrip("%")) / 100
    print("EER hip %.4f oracle %.4f" % (eer_h, eer_o))
    assert abs(eer_h - eer_o) <= 0.005 + 1e-9


def test_resume_continues_the_uninterrupted_run(tmp_path):
    """--resume (reference scripts/train_resnet.py:209-229): epoch counter, weights, BatchNorm buffers, momentum buffers and
    the learning-rate schedule come back from the checkpoint - a run resumed from checkpoint_epoch0 must write the same
    checkpoint_epoch1 as the uninterrupted 2-epoch run (the default --lr-final 1e-4 is also what the resume path
    hard-codes, :225).  Tile autotuning is pinned off: it is timing-based and would change summation orders between runs."""
    d = str(tmp_path)
    n_spk = _make_data(d)
    env = dict(os.environ, PYTHONPATH=ROOT, SPK_AUTOTUNE="0")

    def run(logdir, extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train_resnet.py"), "--gpu", "0", "--workers", "0",
                            "--batch-size", "16", "--print-freq", "1", "--arch", "resnet34", "--input-dim", "80",
                            "--loss-type", "AAM", "--pooling", "mean+std", "--epochs", "2", "--lr", "0.01", "--wd", "5e-4",
                            "--max-chunk-size", "200", "--train-list", os.path.join(d, "train.scp"),
                            "--cv-list", os.path.join(d, "cv.scp"), "--spk-num", str(n_spk),
                            "--utt2spkid", os.path.join(d, "utt2spkid"), "--seed", "7", "--native-reader",
                            "--log-dir", os.path.join(d, logdir)] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r.stdout

    run("full", [])
    out = run("resumed", ["--resume", os.path.join(d, "full", "checkpoint_epoch0.pth.tar")])
    assert "=> loaded checkpoint" in out and "(epoch 1)" in out and "Epoch: [0][" not in out and "Epoch: [1][" in out
    a = torch.load(os.path.join(d, "full", "checkpoint_epoch1.pth.tar"), map_location="cpu", weights_only=True)
    b = torch.load(os.path.join(d, "resumed", "checkpoint_epoch1.pth.tar"), map_location="cpu", weights_only=True)
    first = torch.load(os.path.join(d, "full", "checkpoint_epoch0.pth.tar"), map_location="cpu", weights_only=True)
    assert a["epoch"] == b["epoch"] == 2
    moved = 0
    for k, v in a["state_dict"].items():
        assert torch.equal(v, b["state_dict"][k]), k
        moved += int(not torch.equal(v, first["state_dict"][k]))
    assert moved > 200                       # epoch 1 really trained (every tensor but a few moved)
    assert float(a["best_acc1"]) == float(b["best_acc1"])
    sa, sb = a["optimizer"], b["optimizer"]
    assert sa["param_groups"][0]["lr"] == pytest.approx(sb["param_groups"][0]["lr"], rel=1e-12)
    for i, ent in sa["state"].items():
        assert torch.equal(ent["momentum_buffer"], sb["state"][i]["momentum_buffer"]), i


@pytest.mark.parametrize("reader", ["native", "dataloader"])
def test_variable_length_training_through_the_entry_point(tmp_path, reader):
    """BASELINE configs[3] (variable-length batches) through scripts/train_resnet.py: --var-chunk draws ONE chunk length per batch
    in [--min-chunk-size, --max-chunk-size] (reference scripts/datasets.py:40-43,53-57 draws per sample, which cannot be
    batched), the step replays one captured hipGraph per distinct length out of a cache that shares one memory pool
    (engine.GraphedStepCache), and the host cost of a step whose length has been seen before stays below 2 ms.  Both ingest
    paths: the native reader and Dataset + DataLoader worker processes (the length rides on the index)."""
    import re
    d = str(tmp_path)
    n_spk = _make_data(d, per_spk=14, frames=240)
    env = dict(os.environ, PYTHONPATH=ROOT, SPK_AUTOTUNE="0")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_resnet.py"), "--gpu", "0", "--workers", "2",
           "--batch-size", "8", "--print-freq", "1", "--arch", "resnet34", "--input-dim", "80", "--loss-type", "AAM",
           "--pooling", "mean+std", "--epochs", "3", "--lr", "0.01", "--lr-final", "0.001", "--wd", "5e-4",
           "--var-chunk", "--min-chunk-size", "200", "--max-chunk-size", "232", "--chunk-quantum", "16",
           "--train-list", os.path.join(d, "train.scp"), "--cv-list", os.path.join(d, "cv.scp"), "--spk-num", str(n_spk),
           "--utt2spkid", os.path.join(d, "utt2spkid"), "--seed", "7", "--log-dir", os.path.join(d, "exp")]
    if reader == "native":
        cmd.append("--native-reader")
    log = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert log.returncode == 0, log.stdout[-3000:] + log.stderr[-3000:]
    assert "variable-length training: one chunk length per batch in [200, 232] step 16" in log.stdout
    rep = re.findall(r"epoch (\d+) captured steps: (\d+) \(chunk lengths \[([0-9, ]+)\]\), host enqueue ([0-9.]+) ms/step over (\d+) replays",
                     log.stdout)
    assert rep, log.stdout[-2000:]
    lengths = sorted(int(v) for v in rep[-1][2].split(","))
    assert set(lengths) <= {200, 216, 232} and len(lengths) >= 2, lengths        # several distinct lengths were trained on
    assert int(rep[-1][1]) == len(lengths)                                      # one captured step per length, reused across epochs
    assert float(rep[-1][3]) < 2.0, rep                                         # host enqueue per replayed step
    assert " * Acc@1 " in log.stdout and os.path.exists(os.path.join(d, "exp", "checkpoint_epoch2.pth.tar"))
    losses = [float(v) for v in re.findall(r"Loss ([0-9.e+-]+) \(", log.stdout)]
    assert all(np.isfinite(losses)) and len(losses) > 10


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _strip(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def test_reference_launch_path_on_rccl_with_one_rank(tmp_path):
    """The launch path every stage of the reference's recipes uses (run_aam_v2.sh:86-102,114-126; scripts/train_resnet.py:
    122-128,148-149; scripts/decode.py:55-57): --multiprocessing-distributed --world-size 1 --rank 0 --dist-url tcp://... ->
    mp.spawn -> init_process_group('nccl') = RCCL, here with --gpu-num 1 (RCCL refuses two ranks on one device, so one rank
    is what a one-GPU box can run).  Three runs of the same two epochs (tile autotuning pinned off, it is timing-based):
      plain     --gpu 0: no process group, one captured graph per step
      spawned   the reference's flags: RCCL communicator, DistributedSampler / rank-sharded reader, 'module.' checkpoint keys
      forced    spawned + SPK_FORCE_REDUCER=1: the world > 1 machinery on that one rank - six stage-segmented hipGraph replays
                sharing one memory pool with a real RCCL collective enqueued on the communication stream between them
    All three must write bit-identical checkpoints (weights, BatchNorm buffers, momentum).  decode.py through the same launch
    path must write the embeddings of the plain run."""
    d = str(tmp_path)
    n_spk = _make_data(d)
    env = dict(os.environ, PYTHONPATH=ROOT, SPK_AUTOTUNE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = ["--workers", "0", "--batch-size", "16", "--print-freq", "1", "--arch", "resnet34", "--input-dim", "80",
            "--loss-type", "AAM", "--pooling", "mean+std", "--epochs", "2", "--lr", "0.01", "--wd", "5e-4",
            "--max-chunk-size", "200", "--train-list", os.path.join(d, "train.scp"), "--cv-list", os.path.join(d, "cv.scp"),
            "--spk-num", str(n_spk), "--utt2spkid", os.path.join(d, "utt2spkid"), "--seed", "7", "--native-reader"]

    def dist_flags():
        return ["--multiprocessing-distributed", "--world-size", "1", "--rank", "0", "--gpu-num", "1",
                "--dist-url", "tcp://127.0.0.1:%d" % _free_port(), "--dist-backend", "nccl"]

    def train(logdir, extra, env_extra=None):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train_resnet.py")] + base + extra +
                           ["--log-dir", os.path.join(d, logdir)], env=dict(env, **(env_extra or {})), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r.stdout

    train("plain", ["--gpu", "0"])
    out_s = train("spawned", dist_flags())
    out_f = train("forced", dist_flags(), {"SPK_FORCE_REDUCER": "1"})
    assert "gradient all-reduce per ResNet stage" not in out_s
    assert "gradient all-reduce per ResNet stage on the communication stream: backend nccl, 1 rank(s)" in out_f
    import re
    calls = [int(v) for v in re.findall(r"collectives issued so far: (\d+)", out_f)]
    steps = len(re.findall(r"Epoch: \[\d+\]\[", out_f))
    assert calls and calls[-1] == 6 * steps, (calls, steps)            # head, layer4..1, stem per step
    ck = {n: torch.load(os.path.join(d, n, "checkpoint_epoch1.pth.tar"), map_location="cpu", weights_only=True)
          for n in ("plain", "spawned", "forced")}
    assert all(k.startswith("module.") for k in ck["spawned"]["state_dict"])          # DDP-wrapped naming (train_resnet.py:283-289)
    assert not any(k.startswith("module.") for k in ck["plain"]["state_dict"])
    ref_sd = ck["plain"]["state_dict"]
    for n in ("spawned", "forced"):
        sd = _strip(ck[n]["state_dict"])
        assert sorted(sd) == sorted(ref_sd)
        for k, v in ref_sd.items():
            assert torch.equal(v, sd[k]), (n, k)
        for i, ent in ck["plain"]["optimizer"]["state"].items():
            assert torch.equal(ent["momentum_buffer"], ck[n]["optimizer"]["state"][i]["momentum_buffer"]), (n, i)
        assert float(ck[n]["best_acc1"]) == float(ck["plain"]["best_acc1"])
    # extraction through the same launch path: rank file '0' instead of 'alone', same vectors
    dec = ["--workers", "0", "--batch-size", "1", "--chunk-size", "-1", "--spk_num", str(n_spk), "--arch", "resnet34",
           "--input-dim", "80", "--pooling", "mean+std", "--decode-scp", os.path.join(d, "decode.scp")]
    for name, extra, model in (("emb_plain", ["--gpu", "0"], os.path.join(d, "plain", "checkpoint_epoch1.pth.tar")),
                               ("emb_spawned", dist_flags()[:-2] + ["--dist-backend", "nccl"],
                                os.path.join(d, "spawned", "checkpoint_epoch1.pth.tar"))):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "decode.py")] + dec + extra +
                           ["--model-path", model, "--out-path", os.path.join(d, name)], env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    a = open(os.path.join(d, "emb_plain", "alone")).read().splitlines()
    b = open(os.path.join(d, "emb_spawned", "0")).read().splitlines()
    assert sorted(a) == sorted(b) and len(a) == sum(1 for _ in open(os.path.join(d, "decode.scp")))


def test_run_aam_cpu_sh_end_to_end(tmp_path):
    """BASELINE configs[0] at its stated size through the shell entry point itself (reference run_aam_cpu.sh:60-110: 1 000
    utterances of 200..260 frames x 80 mel, 10 speakers, batch 32, ResNet-34 + AAM, train -> decode -> mean -> cosine -> EER).
    The recipe must run to the end, leave the checkpoints, one embedding per utterance, a score per trial and an EER file
    whose number equals the EER recomputed here from the embeddings it wrote; the synthetic speakers (0.5 sigma mean offsets)
    separate, so the EER must be away from chance.  The recipe is 60 SGD steps from a random initialisation: unseeded (as the
    reference's recipe runs) it ended at 12.0 / 15.0 / 20.0 / 21.3 / 25.1 / 29.6 / 33.3 / 33.9 / 36.7 % over nine runs on two
    builds of the weight-gradient kernel (profiles/r04_c1_eer_spread.log) - a bound of 35 % sat inside that spread and failed one
    run in nine.  The test seeds the run (SPK_SEED: one trajectory per build) and asks for < 45 %: chance is 50 %."""
    d = str(tmp_path / "exp")
    r = subprocess.run(["bash", os.path.join(ROOT, "run_aam_cpu.sh"), d], cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT, SPK_SEED="0"),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for f in ("checkpoint_epoch0.pth.tar", "checkpoint_epoch1.pth.tar", "train.log", "mean.vec", "scores", "eer_cosine"):
        assert os.path.exists(os.path.join(d, f)), f
    log = open(os.path.join(d, "train.log")).read()
    assert "Epoch: [1][" in log and " * Acc@1 " in log and "train_loader samples: 30" in log          # 950 utterances / 32
    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd import scoring
    emb = scoring.read_embeddings(os.path.join(d, "embeddings", "alone"))
    assert len(emb) == 1000 and all(len(v) == 256 for v in emb.values())
    n_trials = sum(1 for _ in open(os.path.join(d, "data", "trials")))
    assert sum(1 for _ in open(os.path.join(d, "scores"))) == n_trials
    txt = open(os.path.join(d, "eer_cosine")).read().strip()
    assert txt.startswith("EER: ") and txt.endswith("%"), txt
    eer_file = float(txt[len("EER: "):-1]) / 100
    mean = scoring.compute_mean(os.path.join(d, "embeddings", "alone"))
    sc, lab = scoring.cosine_score(emb, emb, os.path.join(d, "data", "trials"), mean)
    eer_here = scoring.compute_eer(sc, lab)
    print("run_aam_cpu.sh: EER %.4f (recomputed %.4f) over %d trials" % (eer_file, eer_here, n_trials))
    assert abs(eer_file - eer_here) <= 1e-4 + 1e-9
    assert eer_file < 0.45


def test_resnet101_variable_length_through_the_entry_point(tmp_path):
    """BASELINE configs[3] is ResNet-101 WITH variable-length batches: Bottleneck blocks (reference scripts/model.py:100-135,
    resnet101 :314-321) through scripts/train_resnet.py --arch resnet101 --var-chunk, two chunk lengths, a few steps per epoch
    (--max-steps).  One captured step per length out of the shared pool; finite, decreasing-on-average loss; a 1x1/3x3/1x1
    checkpoint with the reference's key naming."""
    import re
    d = str(tmp_path)
    n_spk = _make_data(d, per_spk=14, frames=240)
    env = dict(os.environ, PYTHONPATH=ROOT, SPK_AUTOTUNE="0")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_resnet.py"), "--gpu", "0", "--workers", "2",
           "--batch-size", "8", "--print-freq", "1", "--arch", "resnet101", "--input-dim", "80", "--loss-type", "AAM",
           "--pooling", "mean+std", "--epochs", "2", "--lr", "0.01", "--lr-final", "0.001", "--wd", "5e-4", "--max-steps", "6",
           "--var-chunk", "--min-chunk-size", "200", "--max-chunk-size", "216", "--chunk-quantum", "16",
           "--train-list", os.path.join(d, "train.scp"), "--cv-list", os.path.join(d, "cv.scp"), "--spk-num", str(n_spk),
           "--utt2spkid", os.path.join(d, "utt2spkid"), "--seed", "11", "--native-reader", "--log-dir", os.path.join(d, "exp")]
    log = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert log.returncode == 0, log.stdout[-3000:] + log.stderr[-3000:]
    assert "=> creating model 'resnet101'" in log.stdout
    rep = re.findall(r"epoch (\d+) captured steps: (\d+) \(chunk lengths \[([0-9, ]+)\]\)", log.stdout)
    assert rep, log.stdout[-2000:]
    lengths = sorted(int(v) for v in rep[-1][2].split(","))
    assert lengths == [200, 216], lengths
    losses = [float(v) for v in re.findall(r"Epoch: \[\d+\].*Loss ([0-9.e+-]+) \(", log.stdout)]
    assert len(losses) == 12 and all(np.isfinite(losses))
    ck = torch.load(os.path.join(d, "exp", "checkpoint_epoch1.pth.tar"), map_location="cpu", weights_only=True)
    import json
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys_resnet101_AAM.json")))     # recorded from the reference
    assert sorted(ck["state_dict"]) == sorted(k for k, _ in keys)
    for k, shape in keys:
        if k != "last.weight":                                                                        # (fixture: another speaker count)
            assert list(ck["state_dict"][k].shape) == shape, k
    assert ck["arch"] == "resnet101"
