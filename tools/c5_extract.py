#!/usr/bin/env python3
"""End-to-end extraction + scoring run (case C5) on one GPU box: synthetic ark on local disk ->
scripts/decode.py --native-reader (bs 512, predict only, text ark) -> compute_mean -> cosine_score --backend hip ->
compute_eer.  Prints one JSON object with the stage timings; parity of the same pipeline against the CPU oracle is
tests/test_pipeline_gpu.py."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--dir", default="/tmp/c5")
ap.add_argument("--speakers", type=int, default=200)
ap.add_argument("--utts-per-speaker", type=int, default=100)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--trials", type=int, default=100000)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--workers", type=int, default=8)
ap.add_argument("--out-format", default="text", help="text (the reference's line format) or fv (binary Kaldi float vectors)")
ap.add_argument("--big-trials", type=int, default=5000000,
                help="also score N seeded trials in-process on both back ends (no score file: the per-line Python formatting is common "
                     "to both and would hide the difference) - the size at which the arithmetic, not the file parsing, decides; 0 = skip")
ap.add_argument("--oracle-subset", type=int, default=2048,
                help="the CPU oracle (the checker) extracts the first N utterances from the same checkpoint: max 1 - cos against the "
                     "HIP embeddings and the EER of both on 20 k seeded trials inside the subset (SURVEY.md section 8d, C5); 0 = skip")
a = ap.parse_args()
py = sys.executable
sc = os.path.join(ROOT, "scripts")


def run(cmd):
    t = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout)
        raise SystemExit("failed: %s" % " ".join(cmd))
    return time.time() - t, r.stdout


res = {"utterances": a.speakers * a.utts_per_speaker, "frames": a.frames, "batch": a.batch}
res["make_data_s"], _ = run([py, os.path.join(ROOT, "tools", "make_synth_data.py"), "--out", a.dir, "--speakers", str(a.speakers),
                             "--utts-per-speaker", str(a.utts_per_speaker), "--min-frames", str(a.frames), "--max-frames",
                             str(a.frames), "--trials", str(a.trials)])
import torch  # noqa: E402
import pytorch_kaldi_resnet_amd  # noqa: E402,F401
from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel  # noqa: E402
torch.manual_seed(0)
m = NeuralSpeakerModel(a.speakers, 80, "mean+std", "AAM", 0.2, 30)
ck = os.path.join(a.dir, "init.pth.tar")
torch.save({"epoch": 0, "arch": "resnet34", "state_dict": m.state_dict()}, ck)
out = os.path.join(a.dir, "emb")
res["decode_wall_s"], log = run([py, os.path.join(sc, "decode.py"), "--spk_num", str(a.speakers), "--arch", "resnet34",
                                 "--input-dim", "80", "--pooling", "mean+std", "--model-path", ck, "--decode-scp",
                                 os.path.join(a.dir, "all.scp"), "--out-path", out, "-b", str(a.batch), "--gpu", "0",
                                 "--native-reader", "-j", str(a.workers), "--out-format", a.out_format])
for line in log.splitlines():
    if line.startswith("=> extracted"):
        res["decode_loop"] = line
iv = os.path.join(out, "alone")
res["embedding_file_mb"] = round(os.path.getsize(iv) / 1e6, 1)
res["compute_mean_s"], _ = run([py, os.path.join(sc, "compute_mean.py"), iv, os.path.join(a.dir, "mean.vec")])
res["cosine_score_hip_s"], _ = run([py, os.path.join(sc, "cosine_score.py"), "--mean", os.path.join(a.dir, "mean.vec"), "--enroll", iv,
                                    "--test", iv, "--trials", os.path.join(a.dir, "trials"), "--score-file",
                                    os.path.join(a.dir, "scores"), "--backend", "hip"])
res["cosine_score_host_s"], _ = run([py, os.path.join(sc, "cosine_score.py"), "--mean", os.path.join(a.dir, "mean.vec"), "--enroll", iv,
                                     "--test", iv, "--trials", os.path.join(a.dir, "trials"), "--score-file",
                                     os.path.join(a.dir, "scores_host")])
import numpy as np  # noqa: E402
s_d = np.array([float(l.split()[2]) for l in open(os.path.join(a.dir, "scores"))])
s_h = np.array([float(l.split()[2]) for l in open(os.path.join(a.dir, "scores_host"))])
res["hip_vs_host_max_abs_score_diff"] = float(np.abs(s_d - s_h).max())
res["compute_eer_s"], eer = run([py, os.path.join(sc, "compute_eer.py"), os.path.join(a.dir, "scores"), os.path.join(a.dir, "trials")])
res["eer"] = eer.strip().splitlines()[-1]
_, eer_h = run([py, os.path.join(sc, "compute_eer.py"), os.path.join(a.dir, "scores_host"), os.path.join(a.dir, "trials")])
res["eer_host_scores"] = eer_h.strip().splitlines()[-1]
res["trials"] = len(s_d)
res["out_format"] = a.out_format
if a.big_trials:
    # compute-bound size: both back ends in this process on the same table (embeddings parsed once, natively)
    from pytorch_kaldi_resnet_amd import kaldi_io as _kio, scoring as _sc  # noqa: E402
    emb_t = _sc.read_embeddings(iv)
    mean_v = _kio.read_vec_flt(os.path.join(a.dir, "mean.vec"))
    rs_b = np.random.RandomState(5)
    names_b = emb_t.keys_list
    ia_b, ib_b = rs_b.randint(0, len(names_b), a.big_trials), rs_b.randint(0, len(names_b), a.big_trials)
    big = os.path.join(a.dir, "trials_big")
    t = time.time()
    with open(big, "w") as f:
        f.write("".join("%s %s %s\n" % (names_b[i], names_b[j], "target" if names_b[i][:7] == names_b[j][:7] else "nontarget")
                        for i, j in zip(ia_b.tolist(), ib_b.tolist())))
    res["big_trials"] = a.big_trials
    res["big_trials_write_s"] = round(time.time() - t, 2)
    _sc.cosine_score(emb_t, emb_t, os.path.join(a.dir, "trials_subset") if os.path.exists(os.path.join(a.dir, "trials_subset"))
                     else os.path.join(a.dir, "trials"), mean_v, backend="hip")          # device warm-up (context, library load)
    for backend in ("hip", "host"):
        t = time.time()
        sc_b, _ = _sc.cosine_score(emb_t, emb_t, big, mean_v, backend=backend)
        res["big_trials_cosine_%s_s" % backend] = round(time.time() - t, 2)
        res["big_trials_score_sum_%s" % backend] = float(np.asarray(sc_b, dtype=np.float64).sum())
if a.oracle_subset:
    # the checker: CPU oracle on a subset, same checkpoint (oracle/ is test infrastructure: only compared against here)
    from oracle import spk_oracle as O  # noqa: E402
    from pytorch_kaldi_resnet_amd import kaldi_io, scoring  # noqa: E402
    # 16 utterances of every k-th speaker (utterance ids are spkNNNN-uttNNNN, speakers contiguous in all.scp): a subset with
    # target trials in it
    alll = [l.split() for l in open(os.path.join(a.dir, "all.scp"))]
    per = min(16, a.utts_per_speaker)
    nspk_sub = max(1, a.oracle_subset // per)
    spk_step = max(1, a.speakers // nspk_sub)
    sub = []
    for sidx in range(0, a.speakers, spk_step):
        sub += alll[sidx * a.utts_per_speaker:sidx * a.utts_per_speaker + per]
    sub = sub[:a.oracle_subset]
    st = {k: v.clone() for k, v in torch.load(ck, map_location="cpu", weights_only=True)["state_dict"].items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t = time.time()
    o_emb = {}
    with torch.no_grad():
        for i in range(0, len(sub), 32):
            xs = torch.from_numpy(np.stack([np.ascontiguousarray(kaldi_io.read_mat(rx).T) for _, rx in sub[i:i + 32]]))
            e = O.embed(st, xs, "mean+std", "resnet34", train=False).numpy().astype(np.float64)
            for (utt, _), v in zip(sub[i:i + 32], e):
                o_emb[utt] = v
    res["oracle_subset_utts"], res["oracle_extract_s"] = len(sub), round(time.time() - t, 2)
    hip_emb = scoring.read_embeddings(iv)
    worst = 0.0
    for utt, v in o_emb.items():
        h = np.asarray(hip_emb[utt], dtype=np.float64)
        worst = max(worst, 1.0 - float(h @ v / (np.linalg.norm(h) * np.linalg.norm(v))))
    res["oracle_subset_max_1_minus_cos"] = worst
    rs = np.random.RandomState(77)
    names = [u for u, _ in sub]
    tr = os.path.join(a.dir, "trials_subset")
    with open(tr, "w") as f:
        for _ in range(20000):
            i, j = rs.randint(0, len(names), 2)
            if i != j:
                f.write("%s %s %s\n" % (names[i], names[j], "target" if names[i][:7] == names[j][:7] else "nontarget"))
    h_sub = {u: np.asarray(hip_emb[u], dtype=np.float64) for u in names}
    for tag, emb in (("hip", h_sub), ("oracle", o_emb)):
        mean = np.mean(np.stack([emb[u] for u in names]).astype(np.float32), axis=0)
        sc_, lab_ = scoring.cosine_score(emb, emb, tr, mean)
        res["subset_eer_" + tag] = round(float(scoring.compute_eer(sc_, lab_)), 5)
        res["subset_targets"] = int(lab_.sum())
print(json.dumps(res))
