#!/usr/bin/env python3
"""Headline benchmark: training throughput of the ResNet-34 + AAM-softmax hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without WORLD_SIZE in the environment: this process spawns its N ranks itself (python -m torch.distributed.run
  --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...) BEFORE touching the GPU, relays rank 0's
  JSON line and exits with the children's code - the reference spawns its own ranks too (scripts/train_resnet.py:122-128).
  Launched by torch.distributed.run (RANK / WORLD_SIZE set) it is one of the ranks.

A "step" is one full training iteration of BASELINE.json configs[1] on one batch of synthetic input that is
already resident in HBM: forward (train-mode BN) + AAM margin + cross-entropy + hand-written backward +
fused SGD (+ overlapped RCCL gradient all-reduce when N > 1; per-GPU batch fixed at 256 = weak scaling).
Prints ONE JSON line (see the repo contract) with two extra objects:
  roofline     - dominant kernel (by total device time in an instrumented pass of the same steps): its algorithmic
                 FLOPs and bytes per launch / average launch duration measured with HIP events on the launch stream,
                 priced against BOTH ceilings of MI355X_MICROARCH.md - the dense matrix peak of its operand mode (fp16 /
                 bf16 2.5 PFLOP/s divided by the matrix instructions per fp32 multiply-add; fp32 157.3 TFLOP/s) and
                 8 TB/s of HBM; the larger fraction names the binding one
  f16_window   - (f16x3 mode) counters of the operand-scale windows of one full-size step: staged values that would
                 saturate fp16 (must be 0: every scale comes from an absmax or a rigorous bound) and the share that
                 has a subnormal low term (the matrix instruction keeps fp16 subnormals: 11..22 significand bits, absolute
                 error <= bound * 2^-39)
  cpu_baseline - the CPU oracle (oracle/spk_oracle.py, a torch-CPU port of the reference path) timed on this
                 host's cores on a bounded sample of the same workload (rank 0, N = 1 only): 2 warm-ups + median of 5
                 steps of the C2-micro (bs 32, T = 300) and the C1 shape (bs 32, T = 200), train step and predict
  eer          - cosine-score EER of HIP embeddings of a seeded 2048-utterance synthetic set, and of the CPU oracle's
                 embeddings of a 256-utterance subset next to the HIP EER on the same trials (reference test.sh:33-39,65-74)

  --config c1|c2|c4|c5 selects a BASELINE.json configuration by name (c2 = the headline, default):
     c1  ResNet-34 + AAM, 10 speakers, bs 32, 200 frames       (the run_aam_cpu.sh shape, on the GPU)
     c4  ResNet-101 + AAM, 5994 speakers, one chunk length per step in [200, 400], bs 256 (per-length hipGraph cache)
     c5  eval-mode embedding extraction (predict), bs 512, 300 frames
  --ingest   loader-inclusive throughput: a seeded fbank ark is written once, then every step's batch comes from the native
             reader (libspkio: pread of the cropped frames on --ingest-threads threads into a pinned ring) -> H2D copy on a
             copy stream -> the same graph replay; reports utt/s and its ratio to the resident-input figure of the same run
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, FEAT, FRAMES, SPK = 256, 80, 300, 1211
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters (dense fp32 matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table: dense bf16 / fp16 matrix (no sparsity)
PEAK_HBM_GBS = 8000.0           # same table: HBM3E peak (6.3 TB/s is what a float4 copy achieves)


def mfma_peak(kernel_label):
    """Peak rate of ALGORITHMIC fp32 FLOPs for a kernel: the fp32 matrix peak for fp32-operand kernels; for the bf16-split
    kernels every fp32 multiply-add costs 6 (or 9) bf16 ones, so the dense bf16 peak divided by that."""
    inner = kernel_label[kernel_label.find("<") + 1:kernel_label.rfind(">")].split(",") if "<" in kernel_label else []
    terms = 0
    if kernel_label.startswith("conv_mfma_kernel") and len(inner) == 4:
        terms = int(inner[3])
    elif kernel_label.startswith("conv_ws_kernel") and len(inner) == 5:
        terms = int(inner[4])
    elif kernel_label.startswith("conv_wgrad_split_kernel") and len(inner) == 5:
        terms = int(inner[3])
    elif kernel_label.startswith(("conv_pipe_kernel", "conv_wgrad_wm_kernel", "conv_wgrad_wm16_kernel", "conv_wgrad_c32m16_kernel", "conv_wgrad_1x1_kernel", "conv_wgrad_pipe_kernel",
                                  "conv_wgrad_ws_kernel")):
        terms = 3           # these forms exist for the f16x3 operand mode only
    if terms:
        return PEAK_BF16_MFMA_TFLOPS / terms, "%s dense peak / %d cross products per fp32 multiply-add" % (
            "fp16" if terms == 3 else "bf16", terms)
    return PEAK_F32_MFMA_TFLOPS, "fp32 dense matrix peak"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="per-GPU batch (256 = the BASELINE config)")
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--mode", choices=["train", "predict"], default="train",
                    help="train = the BASELINE metric (default); predict = eval-mode embedding extraction (decode.py path)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--arch", choices=["resnet34", "resnet101"], default="resnet34",
                    help="resnet101 + --speakers 5994 + --frames-range 200 400 = BASELINE configs[3] (parity/coverage case, not the headline)")
    ap.add_argument("--speakers", type=int, default=SPK)
    ap.add_argument("--frames-range", type=int, nargs=2, metavar=("LO", "HI"), default=None,
                    help="one chunk length per step, uniform in [LO, HI] (seeded) like the reference's variable-length "
                         "batches (scripts/datasets.py:178-193); implies eager launches")
    ap.add_argument("--mfma", choices=["f16x3", "bf16x6", "bf16x9", "f32"], default=None,
                    help="operand mode of the 3x3 convolutions (default: the package default, f16x3 = fp32 operands as two fp16 "
                         "terms of value x 2^k, 3 cross products on the fp16 MFMA, fp32 accumulate; bf16x6 = three exact bf16 "
                         "terms, 6 cross products; f32 = native fp32 MFMA)")
    ap.add_argument("--autotune", action="store_true", help="time candidate tiles on first use of a launch shape (SPK_AUTOTUNE=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed CPU steps per leg (median reported) after 2 warm-ups")
    ap.add_argument("--no-eer", action="store_true")
    ap.add_argument("--config", choices=["c1", "c2", "c4", "c5"], default=None,
                    help="BASELINE.json configuration preset (c2 = headline = the defaults)")
    ap.add_argument("--ingest", action="store_true",
                    help="also time the loader-inclusive step: seeded ark -> libspkio reader -> pinned ring -> H2D -> graph replay")
    ap.add_argument("--ingest-utts", type=int, default=4096, help="utterances in the synthetic ark of --ingest")
    ap.add_argument("--ingest-threads", type=int, default=8, help="reader threads of --ingest")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the comparison run on the native fp32 matrix instruction (profiled "
                                                                "runs: keeps its kernels out of the trace)")
    ap.add_argument("--no-f16-window", action="store_true", help="skip the operand-scale window counters (f16x3 mode)")
    ap.add_argument("--lengths", type=int, default=8, help="distinct chunk lengths of --frames-range (one cached hipGraph each)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the bounded extra legs of the default N = 1 line (after the timed region, never part of `value`): "
                         "`ingest` (loader-inclusive step), `extra.c5` (predict bs 512 with its own roofline object), `extra.c4` "
                         "(ResNet-101, S 5994, three cached chunk lengths in [200, 400])")
    a = ap.parse_args()
    if a.config == "c1":
        a.batch, a.frames, a.speakers = 32, 200, 10
    elif a.config == "c4":
        a.arch, a.speakers, a.frames_range = "resnet101", 5994, [200, 400]
    elif a.config == "c5":
        a.mode, a.batch = "predict", 512
    return a


def cpu_baseline(args):
    """The CPU oracle on the host cores, bounded sample (SURVEY.md section 8d / BASELINE.md section 3): per leg 2 warm-ups, then
    the median of --cpu-steps steps.  Legs: C2-micro (bs 32 x 300 frames, S = 1211) train step + predict - the same
    workload as the metric, `value` - and the C1 shape (bs 32 x 200 frames, S = 10) train step + predict."""
    import statistics

    from oracle import spk_oracle as O
    from oracle import weights as W
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))   # the GPU box's CPU share is 16 cores per GPU
    bs = args.cpu_batch

    def leg(frames, spk, what):
        st = O.to_torch_state(W.make_state(0, spk, FEAT, "mean+std", "AAM", "resnet34"))
        x, y = W.make_input(1234, bs, FEAT, frames, spk)
        x, y = torch.from_numpy(x), torch.from_numpy(y)
        bufs = {}

        def one():
            t = time.perf_counter()
            if what == "train":
                O.train_step(st, bufs, x, y, 1e-4, weight_decay=5e-4)
            else:
                with torch.no_grad():
                    O.embed(st, x, "mean+std", "resnet34", train=False)
            return time.perf_counter() - t
        for _ in range(2):
            one()
        ts = [one() for _ in range(args.cpu_steps)]
        med = statistics.median(ts)
        log("cpu %s bs%d x %d frames: median %.3f s/step over %d steps" % (what, bs, frames, med, len(ts)))
        return {"utt_per_s": round(bs / med, 3), "s_per_step_median": round(med, 4), "steps": len(ts), "warmup": 2}

    c2t = leg(args.frames, SPK, "train")
    c2p = leg(args.frames, SPK, "predict")
    c1t = leg(200, 10, "train")
    c1p = leg(200, 10, "predict")
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": c2t["utt_per_s"], "unit": "utt/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "median of %d train steps (after 2 warm-ups) of bs%d x %d frames x %d mel, ResNet-34+AAM S=%d, torch %s "
                      "CPU oracle (%.2f s/step)" % (args.cpu_steps, bs, args.frames, FEAT, SPK, torch.__version__,
                                                   c2t["s_per_step_median"]),
            "host_cpu": cpu_model, "torch_threads": torch.get_num_threads(),
            "c2_micro_train": c2t, "c2_micro_predict": c2p, "c1_train_bs32_t200": c1t, "c1_predict_bs32_t200": c1p}


def eer_leg(dev):
    """Cosine-score EER next to the CPU path's (north_star; reference test.sh:33-39,65-74: compute_mean -> cosine_score
    --mean -> compute_eer).  Seeded synthetic set: 64 speakers x 32 utterances x 200 frames x 80 mel, N(0,1) plus a
    per-speaker mean offset 0.5*N(0,1)[80], hashed weights, eval mode.  HIP and the CPU oracle - the
    checker - both extract all 2048 utterances (the "2 k-utt subset" of SURVEY.md section 8d; ~8 s of CPU) and are scored on
    the same 20 000 seeded trials (HIP on the device back end, the oracle on the host back end); a 256-utterance / 4 000-trial
    subset is reported as well."""
    import contextlib

    import numpy as np
    from oracle import spk_oracle as O
    from oracle import weights as W
    from pytorch_kaldi_resnet_amd import scoring
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    nspk, per, T, sub = 64, 32, 200, 256
    ncpu = nspk * per
    rs = np.random.RandomState(1234)
    spk_mean = 0.5 * rs.randn(nspk, FEAT, 1).astype(np.float32)
    lab = np.repeat(np.arange(nspk), per)
    x = rs.randn(nspk * per, FEAT, T).astype(np.float32) + spk_mean[lab]
    npst = W.make_state(11, 10, FEAT, "mean+std", "AAM", "resnet34")
    with contextlib.redirect_stdout(sys.stderr):
        m = NeuralSpeakerModel(10, FEAT, "mean+std", "AAM", 0.2, 30)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in npst.items()})
    m = m.to(dev).eval()
    t0 = time.perf_counter()
    with torch.no_grad():
        emb = torch.cat([m.predict(torch.from_numpy(x[i:i + 512]).to(dev)) for i in range(0, len(x), 512)]).cpu().numpy()
    t_hip = time.perf_counter() - t0
    st = O.to_torch_state(npst)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = torch.cat([O.embed(st, torch.from_numpy(x[i:i + 32]), "mean+std", "resnet34", train=False)
                         for i in range(0, ncpu, 32)]).numpy()
    t_cpu = time.perf_counter() - t0
    names = ["u%04d" % i for i in range(len(x))]

    def eer_of(vecs, n, ntrials, seed, backend):
        import tempfile
        r = np.random.RandomState(seed)
        a, b = r.randint(0, n, ntrials), r.randint(0, n, ntrials)
        keep = a != b
        a, b = a[keep], b[keep]
        table = {names[i]: vecs[i] for i in range(n)}
        mean = np.mean(np.stack([vecs[i] for i in range(n)]).astype(np.float32), axis=0)
        with tempfile.NamedTemporaryFile("w", suffix=".trials", delete=False) as f:
            for i, j in zip(a, b):
                f.write("%s %s %s\n" % (names[i], names[j], "target" if lab[i] == lab[j] else "nontarget"))
            path = f.name
        try:
            sc, lb = scoring.cosine_score(table, table, path, mean, backend=backend)
        finally:
            os.unlink(path)
        return scoring.compute_eer(sc, lb), int(lb.sum()), len(lb)

    e_all, tgt_all, n_all = eer_of(emb, len(x), 20000, 77, "hip")
    e_call, _, _ = eer_of(ref, ncpu, 20000, 77, "host")
    e_hs, tgt_s, n_s = eer_of(emb[:sub], sub, 4000, 78, "hip")
    e_cs, _, _ = eer_of(ref[:sub], sub, 4000, 78, "host")
    e64, r64 = emb.astype(np.float64), ref.astype(np.float64)
    cos = (e64 * r64).sum(1) / (np.linalg.norm(e64, axis=1) * np.linalg.norm(r64, axis=1))
    return {"hip_2048_utts": round(e_all, 5), "cpu_oracle_2048_utts": round(e_call, 5), "trials_2048": n_all, "targets_2048": tgt_all,
            "hip_subset": round(e_hs, 5), "cpu_oracle_subset": round(e_cs, 5), "subset_utts": sub, "trials_subset": n_s,
            "targets_subset": tgt_s, "max_1_minus_cos_2048_utts": float((1.0 - cos).max()),
            "hip_extract_s": round(t_hip, 3), "cpu_oracle_extract_s": round(t_cpu, 3),
            "data": "64 speakers x 32 utts x 200 frames x 80 mel, N(0,1) + 0.5*N(0,1) per-speaker offset, seed 1234; hashed "
                    "weights (random init: EER reflects the input offsets, not a trained model)"}


def embedding_parity(dev):
    """BASELINE's second metric: max (1 - cosine) between HIP predict() and the CPU oracle on a seeded batch
    (hashed weights, 4 utterances x 200 frames x 80 mel, eval mode).  The oracle is only the checker here."""
    import contextlib

    import numpy as np
    from oracle import spk_oracle as O
    from oracle import weights as W
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    npst = W.make_state(11, 10, FEAT, "mean+std", "AAM", "resnet34")
    with contextlib.redirect_stdout(sys.stderr):
        m = NeuralSpeakerModel(10, FEAT, "mean+std", "AAM", 0.2, 30)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in npst.items()})
    m = m.to(dev).eval()
    x, _ = W.make_input(12, 4, FEAT, 200, 10)
    with torch.no_grad():
        e = m.predict(torch.from_numpy(x).to(dev)).cpu().numpy().astype(np.float64)
        r = O.embed(O.to_torch_state(npst), torch.from_numpy(x), "mean+std", "resnet34", train=False).numpy().astype(np.float64)
    cos = (e * r).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(r, axis=1))
    return float((1.0 - cos).max())


def roofline_object(recs, npass):
    """The `roofline` object from an instrumented pass (ops.PROFILE records of `npass` steps, train or predict): the dominant
    kernel by device time priced against BOTH ceilings - matrix FLOP/s of its operand mode and HBM bytes/s of its algorithmic
    traffic (every tensor it must read or write, once) - the larger fraction names the binding one."""
    # an event pair around NOTHING still measures the marker packets themselves (tens of microseconds on ROCm):
    # calibrate that bracket overhead on the same stream and subtract it from every bracketed launch
    empty = []
    for _ in range(64):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        a1.record()
        empty.append((a0, a1))
    torch.cuda.synchronize()
    overhead_ms = sorted(p0.elapsed_time(p1) for p0, p1 in empty)[len(empty) // 2]
    agg = {}
    for name, flops, e0, e1, nbytes in recs:
        a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
        a[0] += max(e0.elapsed_time(e1) - overhead_ms, 1e-3) * 1e-3
        a[1] += flops
        a[2] += 1
        a[3] += nbytes
    name, (tsum, fsum, n, bsum) = max(agg.items(), key=lambda kv: kv[1][0])
    ach = fsum / tsum / 1e12
    ach_gbs = bsum / tsum / 1e9
    # HBM traffic of that kernel: rocprofv3 cannot run inside this process, so this is the COMMITTED PMC pass of the
    # same command (profiles/pmc_traffic.json) - used only while the device sources still hash to what it measured
    traffic, traffic_note = None, "no committed PMC pass for this kernel"
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        import importlib.util
        spec = importlib.util.spec_from_file_location("spk_build", os.path.join(ROOT, "pytorch-kaldi-resnet_amd", "build.py"))
        bmod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bmod)
        pj = json.load(open(pmc))
        # the profiler's kernel names carry template arguments the label does not (the compile-time flag variants of round 4:
        # "conv_pipe_kernel<3, 2, false, false, true, true, 16432>"): every instance of the labelled kernel, weighted by launches
        key = name.split(" C")[0].replace(",", ", ")
        ents = [v for k, v in pj["kernels"].items() if k == key or (key.endswith(">") and k.startswith(key[:-1] + ","))
                or (not key.endswith(">") and k.startswith(key + "<"))]
        if not ents and key.startswith("spk_"):  # C-ABI entry point labels: spk_bn_bwd_apply launches bn_bwd_apply_kernel
            kk = key[4:] + "_kernel"
            ents = [v for k, v in pj["kernels"].items() if k == kk or k.startswith(kk + "<")]
        ent = None
        if ents:
            nl = sum(e["launches"] for e in ents)
            ent = {"hbm_bytes_per_launch": sum(e["hbm_bytes_per_launch"] * e["launches"] for e in ents) / max(nl, 1)}
        if pj.get("csrc_fingerprint") != bmod.csrc_fingerprint():
            traffic_note = "committed PMC pass is stale (csrc changed since %s): refused" % pj.get("csrc_fingerprint")
        elif ent:
            traffic = round(ent["hbm_bytes_per_launch"])
            traffic_note = ("committed PMC pass @ csrc %s: HBM bytes per launch = FETCH_SIZE x2 + WRITE_SIZE, separate "
                            "rocprofv3 --pmc passes of this command (profiles/pmc_traffic.json)" % pj["csrc_fingerprint"])
    peak, peak_note = mfma_peak(name)
    f_mfma, f_hbm = ach / peak, ach_gbs / PEAK_HBM_GBS
    hbm_bound = f_hbm >= f_mfma
    total_ms = sum(v[0] for v in agg.values()) / npass * 1e3
    total_flops = sum(v[1] for v in agg.values()) / npass
    return {"bound": "hbm" if hbm_bound else "mfma", "kernel": name,
            "achieved": round(ach_gbs if hbm_bound else ach, 2), "peak": PEAK_HBM_GBS if hbm_bound else round(peak, 1),
            "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round(max(f_hbm, f_mfma), 4),
            "mfma": {"achieved_tflops": round(ach, 2), "peak_tflops": round(peak, 1), "frac": round(f_mfma, 4),
                     "peak_note": peak_note, "frac_of_fp32_matrix_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4)},
            "hbm": {"achieved_gbs": round(ach_gbs, 1), "peak_gbs": PEAK_HBM_GBS, "frac": round(f_hbm, 4),
                    "algorithmic_bytes_per_launch": round(bsum / n),
                    "note": "algorithmic bytes = inputs read once + outputs written once + every fused side stream "
                            "(shortcut, BatchNorm-backward raw / side gradient, masks) + packed weights"},
            "traffic": traffic, "traffic_note": traffic_note,
            "launches_per_step": n // npass, "avg_launch_ms": round(tsum / n * 1e3, 4),
            "event_bracket_overhead_us": round(overhead_ms * 1e3, 1),
            "gflop_per_launch": round(fsum / n / 1e9, 3),
            "step_level": {"kernel_ms_sum": round(total_ms, 3), "algorithmic_tflop": round(total_flops / 1e12, 3),
                           "tflops": round(total_flops / 1e9 / max(total_ms, 1e-9), 1)},
            "all_kernels": {k: {"ms_per_step": round(v[0] / npass * 1e3, 3),
                                "tflops": round(v[1] / v[0] / 1e12, 2) if v[1] else None,
                                "algorithmic_gbs": round(v[3] / v[0] / 1e9, 1) if v[3] else None,
                                "launches_per_step": v[2] // npass} for k, v in sorted(agg.items())}}


def extra_leg(name, argv, timeout):
    """A bounded extra leg of the default line: `python bench.py <argv>` as a child process (fresh state, the very command a
    user would type), its JSON line parsed and trimmed.  Never part of `value`; a failure is reported in the object, not hidden."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + argv
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "timeout after %d s" % timeout, "command": "python bench.py " + " ".join(argv)}
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    try:
        j = json.loads(line)
    except ValueError:
        return {"error": "rc %d: %s" % (r.returncode, (r.stderr or "")[-400:]), "command": "python bench.py " + " ".join(argv)}
    rf = j.get("roofline")
    if rf:
        rf = {k: v for k, v in rf.items() if k != "all_kernels"}
    out = {k: j.get(k) for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "first_loss",
                                 "final_loss", "graph_replay_repacks_weights", "kernel_launches_per_step")}
    out["roofline"] = rf
    out["command"] = "python bench.py " + " ".join(argv)
    out["wall_s"] = round(time.perf_counter() - t0, 1)
    log("extra leg %s: %s %s in %.1f s" % (name, out.get("value"), out.get("unit"), out["wall_s"]))
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as fresh children - one process per GPU,
    torch.distributed.run on 127.0.0.1 - before this process has touched the GPU (it never does), relay their output
    (rank 0 prints the JSON line to the inherited stdout) and leave with their exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("spawning %d ranks: %s" % (args.gpus, " ".join(cmd)))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def write_ingest_ark(d, n_utts, frames, nspk, seed):
    """Synthetic training corpus of --ingest: n_utts utterances x (frames + 0..31) frames x 80 mel, N(0,1), in one `FM `
    ark + scp (offset form) + utt2spkid, written with the package's own kaldi_io writer."""
    import numpy as np
    from pytorch_kaldi_resnet_amd import kaldi_io
    rs = np.random.RandomState(seed)
    ark = os.path.join(d, "feats.ark")
    scp, u2s = [], []
    with open(ark, "wb") as f:
        for i in range(n_utts):
            T = frames + int(rs.randint(0, 32))
            mat = rs.standard_normal((T, FEAT)).astype(np.float32)
            utt = "utt%06d" % i
            off = kaldi_io.write_mat(f, mat, key=utt)
            scp.append("%s %s:%d" % (utt, ark, off))
            u2s.append("%s %d" % (utt, i % nspk))
    open(os.path.join(d, "train.scp"), "w").write("\n".join(scp) + "\n")
    open(os.path.join(d, "utt2spkid"), "w").write("\n".join(u2s) + "\n")
    return os.path.join(d, "train.scp"), os.path.join(d, "utt2spkid"), os.path.getsize(ark)


def log(msg):
    sys.stderr.write("[bench %.1fs] %s\n" % (time.time() - T_START, msg))
    sys.stderr.flush()


T_START = time.time()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)          # never returns; nothing above has initialised the GPU
    # rehearsal hooks (one-GPU box): SPK_FORCE_DEVICE pins every rank to one device, SPK_DIST_BACKEND=gloo replaces
    # RCCL so the N > 1 control flow can be exercised where only one GPU exists.  Never set by the driver.
    if "SPK_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["SPK_FORCE_DEVICE"])
    backend = os.environ.get("SPK_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # SPK_FORCE_REDUCER=1 (parallel.reducer_forced): the N > 1 machinery - RCCL communicator, stage-segmented graphs, the
    # all-reduce on the communication stream between the replays - on ONE rank, for a box with a single GPU
    forced = world == 1 and os.environ.get("SPK_FORCE_REDUCER", "0") == "1"
    if forced:
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
        s_.close()
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd import ops
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    from pytorch_kaldi_resnet_amd.optim import FlatSGD
    from pytorch_kaldi_resnet_amd.parallel import GradAllReducer

    torch.manual_seed(0)
    import contextlib
    if args.mfma:
        ops.SPLIT = ops.MFMA_MODES[args.mfma]
    mfma_mode = {v: k for k, v in ops.MFMA_MODES.items()}[ops.SPLIT]
    if args.autotune:
        from pytorch_kaldi_resnet_amd import tiling
        tiling.AUTOTUNE = True
    nspk = args.speakers
    with contextlib.redirect_stdout(sys.stderr):   # the model announces itself like the reference does; keep stdout = 1 JSON line
        model = NeuralSpeakerModel(nspk, FEAT, "mean+std", "AAM", 0.2, 30, arch=args.arch).to(dev)
    model.train()
    opt = FlatSGD(model, 0.1, momentum=0.9, weight_decay=5e-4, grad_scale=1.0 / world)
    red = GradAllReducer(model)
    red.broadcast_parameters(0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    x = torch.randn(args.batch, FEAT, args.frames, device=dev, generator=gen)
    y = torch.randint(0, nspk, (args.batch,), device=dev, generator=gen)
    var_x = None
    if args.frames_range:
        # a fixed, seeded schedule of chunk lengths (every rank draws the same length per step, as the reference's
        # batch sampler does); inputs of each distinct length are generated once and stay resident in HBM
        import random
        rng = random.Random(4321)
        lo, hi = args.frames_range
        lens = [rng.randint(lo, hi) for _ in range(args.lengths)]      # --lengths distinct lengths, cycled
        sched = [lens[i % len(lens)] for i in range(args.warmup + args.steps)]
        var_x = {t: torch.randn(args.batch, FEAT, t, device=dev, generator=gen) for t in sorted(set(sched))}
    step_no = [0]
    # the BASELINE configs[1] label belongs to exactly that configuration (a batch-64 rehearsal carried it)
    headline = (args.arch == "resnet34" and nspk == SPK and var_x is None and args.batch == B_PER_GPU and args.frames == FRAMES
                and args.mode == "train")
    eng = model.engine()
    if os.environ.get("SPK_SIDE_STREAM", "1") == "0":
        eng.use_side_stream = False

    if args.mode == "predict":
        model.eval()

        def step():
            return model.predict(cur_x()).sum()
    else:
        step = None

    # Default launch mode: weight re-pack + forward + CE + backward replayed as hipGraphs (host-load independent).  N = 1: one
    # graph.  N > 1: six graph segments cut where backward finishes a ResNet stage; between two replays the host enqueues
    # that stage's RCCL all-reduce on the communication stream, so the collective overlaps the remaining backward kernels
    # (engine.GraphedTrainStep(segmented=True)).  --no-graph launches every kernel eagerly with the same overlap.
    graphed = None
    graphs = {}          # chunk length -> GraphedTrainStep (variable-length mode: one graph per distinct length, one shared pool)
    if args.mode == "train" and not args.no_graph:
        from pytorch_kaldi_resnet_amd.engine import GraphedTrainStep
        # a capture failure is a real fault (HIP error, shape bug): surface it instead of silently timing another path
        if var_x is None:
            graphed = graphs[args.frames] = GraphedTrainStep(eng, args.batch, args.frames, segmented=red.active)
        else:
            for t in sorted(var_x, reverse=True):           # longest first: it sizes the shared pool
                graphs[t] = GraphedTrainStep(eng, args.batch, t, segmented=red.active,
                                             pool=graphed.pool() if graphed is not None else None)
                graphed = graphed or graphs[t]
                log("captured the step for %d frames" % t)

    def cur_x():
        if var_x is None:
            return x
        t = sched[step_no[0] % len(sched)]
        step_no[0] += 1
        return var_x[t]

    def train_step():
        xb = cur_x()
        if graphed is not None and PROFILE_OFF():
            loss, _, _ = graphs[xb.shape[2]](xb, y, red.on_stage_done if red.active else None)
            red.finish()
            opt.step()
            return loss
        opt.zero_grad(set_to_none=True)
        loss, _, _ = eng.loss_and_grad(xb, y, red.on_stage_done if red.active else None)
        red.finish()
        opt.step()
        return loss

    def PROFILE_OFF():
        return ops.PROFILE is None

    if step is None:
        step = train_step
    log("model built, starting warm-up")
    first_loss = None
    for i in range(args.warmup):
        l0 = step()
        torch.cuda.synchronize()
        if first_loss is None:
            first_loss = float(l0)
        log("warm-up step %d done" % i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_host = time.perf_counter() - t0          # host time to enqueue the K steps (no sync inside)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    lossv = float(loss)
    if first_loss is None:
        first_loss = lossv
    repack_ok = None
    if graphed is not None and args.mode == "train":
        # Proof that every replay trains on the weights the optimizer just wrote (scripts/train_resnet.py:316-328): the last
        # opt.step() changed the parameter arena; one more replay must leave the packed operands of the convolutions equal to
        # a fresh pack of the CURRENT weights.  A graph that does not contain the re-pack fails here.
        g0 = graphs[sorted(graphs)[0]]
        g0(var_x[sorted(graphs)[0]] if var_x is not None else x, y)
        bad = []
        for cv in list(eng._all_convs())[::7]:
            for transpose, buf in ((False, cv.wpk), (True, cv.wpk_t)):
                if not torch.equal(ops.pack_conv_weight(cv.h.weight.data, transpose), buf):
                    bad.append((cv.cin, cv.cout, cv.k, transpose))
        torch.cuda.synchronize()
        repack_ok = not bad
        if bad and not all(bool(torch.isfinite(p_).all()) for p_ in model.parameters()):
            # NaN != NaN: not a stale pack.  The timed steps train on ONE fixed batch; once it is memorised (loss ~0.03, ~250 steps
            # at lr 0.1) a target cosine reaches 1.0 and the gradient of sqrt(1 - cos^2) (scripts/model.py:487) is infinite - in
            # the reference's own formula, in the native-fp32 operand mode as in f16x3 (tools/long_run.py, profiles/r04_long_run.log)
            raise SystemExit("bench: the weights are no longer finite after %d steps on one fixed batch (AAM's sqrt(1 - cos^2) has an "
                             "unbounded gradient at cos = 1, reached once the batch is memorised): use fewer than ~200 steps"
                             % (args.warmup + args.steps))
        if bad:
            raise SystemExit("bench: the replayed hipGraph uses stale packed conv weights %s - not a valid training step" % bad)
    log("timed region done: %.3f s for %d steps (host enqueue %.1f ms/step)" % (dt, args.steps, t_host / args.steps * 1e3))

    ingest = None
    want_extra = rank == 0 and world == 1 and headline and not args.no_extra
    if (args.ingest or want_extra) and args.mode == "train" and var_x is None:
        # Loader-inclusive step (reference scripts/train_resnet.py:307-313: loader -> pinned H2D -> step): every rank reads
        # its own seeded ark through libspkio (pread of the cropped frames on a thread pool into a pinned ring, guarded by
        # copy events), the copy runs on the loader's copy stream under the previous step's kernels, then the same replay.
        import tempfile
        import contextlib as _cl
        from pytorch_kaldi_resnet_amd.ingest import NativeTrainLoader
        d = tempfile.mkdtemp(prefix="spk_ingest_r%d_" % rank)
        t_w = time.perf_counter()
        scp, u2s, ark_bytes = write_ingest_ark(d, args.ingest_utts, args.frames, nspk, 99 + rank)
        log("ingest: wrote %d utterances, %.0f MB ark in %.1f s" % (args.ingest_utts, ark_bytes / 1e6, time.perf_counter() - t_w))
        with _cl.redirect_stdout(sys.stderr):
            loader = NativeTrainLoader(scp, u2s, args.frames, args.batch, rank=0, world=1, seed=5, threads=args.ingest_threads,
                                       drop_last=True, prefetch=3, device=str(dev))

        def batches():
            ep = 0
            while True:
                loader.set_epoch(ep)
                ep += 1
                for xb, yb in loader:
                    yield xb, yb
        it = batches()

        def ingest_step():
            xb, yb = next(it)
            if graphed is not None:
                lossi, _, _ = graphed(xb, yb, red.on_stage_done if red.active else None)
            else:
                opt.zero_grad(set_to_none=True)
                lossi, _, _ = eng.loss_and_grad(xb, yb, red.on_stage_done if red.active else None)
            red.finish()
            opt.step()
            return lossi
        for _ in range(max(2, args.warmup)):
            ingest_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ingest_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dti = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dti], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dti = float(t)
        import shutil
        shutil.rmtree(d, ignore_errors=True)
        ingest = {"value": round(args.batch * world * args.steps / dti, 2), "unit": "utt/s",
                  "ms_per_step": round(dti / args.steps * 1e3, 3),
                  "ratio_to_resident_input": round(dt / dti, 4), "reader_threads": args.ingest_threads,
                  "path": "seeded FM ark (%d utts x %d..%d frames x %d mel, %.0f MB, page cache) -> libspkio pread+crop+transpose -> "
                          "pinned ring (5 slots, copy-event guarded) -> H2D on a copy stream -> step" % (
                              args.ingest_utts, args.frames, args.frames + 31, FEAT, ark_bytes / 1e6)}
        log("ingest-inclusive: %.1f utt/s (%.3f of the resident-input rate)" % (ingest["value"], ingest["ratio_to_resident_input"]))

    f16_window = None
    if rank == 0 and args.mode == "train" and ops.SPLIT == 3 and not args.no_f16_window:
        # one eager step (outside every timed region) with the window counters on: every tensor an f16x3 matrix-core kernel
        # stages is also run through spk_f16_window_count under the scale slot its consumer uses
        eng.window_counts = torch.zeros(4, device=dev, dtype=torch.int64)
        opt.zero_grad(set_to_none=True)
        eng.loss_and_grad(x if var_x is None else var_x[sorted(var_x)[0]], y, None)
        torch.cuda.synchronize()
        tot, sat, lo_lost, hi_sub = eng.window_counts.tolist()
        eng.window_counts = None
        opt.zero_grad(set_to_none=True)
        f16_window = {"staged_values": tot, "saturated": sat, "low_term_subnormal_frac": round(lo_lost / max(tot, 1), 8),
                      "high_term_subnormal_frac": round(hi_sub / max(tot, 1), 8),
                      "note": "values staged by f16x3 matrix-core kernels in one step, judged under the scale slot of their consumer "
                              "(absmax of the tensor or a rigorous bound): saturated must be 0; a subnormal low term = the value "
                              "carries between 11 and 22 significand bits (the matrix instruction keeps fp16 subnormals: absolute "
                              "error <= bound * 2^-39)"}
        log("f16 windows: %d staged values, %d saturated, %.5f %% low term subnormal" % (tot, sat, 100.0 * lo_lost / max(tot, 1)))
        if sat:
            raise SystemExit("bench: %d staged values saturate fp16 under their scale slot - an operand-scale bound is wrong" % sat)

    roofline = None
    if rank == 0 and not args.no_roofline:
        # instrumented pass: same steps, every launch bracketed by HIP events on its launch stream; the side stream
        # for weight gradients is folded into the main stream here so kernels do not overlap while being timed
        eng.use_side_stream = False
        ops.PROFILE = []
        NPASS = 2
        for _ in range(NPASS):
            # rank-local: no collective may be issued here, the other ranks are not in this pass
            if args.mode == "train":
                opt.zero_grad(set_to_none=True)
                eng.loss_and_grad(x, y, None)
                opt.step()
            else:
                model.predict(x)
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        eng.use_side_stream = True
        roofline = roofline_object(recs, NPASS)
    native = None
    if rank == 0 and world == 1 and args.mode == "train" and headline and graphed is not None and ops.SPLIT != 0 \
            and not args.no_roofline and not args.no_fp32_leg:
        # the same step on the native fp32 matrix instruction, timed the same way, for comparison (N = 1 only)
        from pytorch_kaldi_resnet_amd.engine import GraphedTrainStep
        keep = ops.SPLIT
        ops.SPLIT = 0
        eng.dirty = True
        g32 = GraphedTrainStep(eng, args.batch, args.frames)
        for _ in range(2):
            g32(x, y)
            opt.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            g32(x, y)
            opt.step()
        torch.cuda.synchronize()
        d32 = time.perf_counter() - t1
        native = {"mfma": "f32", "value": round(args.batch * args.steps / d32, 2), "unit": "utt/s",
                  "ms_per_step": round(d32 / args.steps * 1e3, 3)}
        ops.SPLIT = keep
        eng.dirty = True
        log("fp32-operand comparison run done: %.1f utt/s" % native["value"])
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "train" and headline:
        log("roofline pass done; timing the CPU oracle (bounded sample)")
        cpu = cpu_baseline(args)
        log("cpu baseline done")
        parity = embedding_parity(dev)
        log("embedding parity vs oracle: max 1-cos = %.3e" % parity)
    eer = None
    if rank == 0 and world == 1 and not args.no_eer and not args.no_cpu_baseline and headline:
        eer = eer_leg(dev)
        log("EER leg done: hip %.4f vs cpu oracle %.4f (2048 utts, 20 k trials); subset hip %.4f vs cpu oracle %.4f" % (
            eer["hip_2048_utts"], eer["cpu_oracle_2048_utts"], eer["hip_subset"], eer["cpu_oracle_subset"]))
    launch_desc = ("eager" if graphed is None else "hipGraph replay (%d graph%s%s)" % (
        len(graphs) * len(graphed.segments), "s" if len(graphs) * len(graphed.segments) > 1 else "",
        ", stage-segmented with the all-reduce between segments" if red.active else ""))
    extra = None
    if want_extra:
        # release what this process holds on the device before the children allocate (graphs, pools, saved tensors)
        graphed = None
        graphs.clear()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        mf = ["--mfma", mfma_mode]
        extra = {"c5": extra_leg("c5", ["--config", "c5", "--steps", "10", "--warmup", "3", "--no-extra"] + mf, 240),
                 "c4": extra_leg("c4", ["--config", "c4", "--lengths", "3", "--steps", "6", "--warmup", "3", "--no-cpu-baseline",
                                        "--no-eer", "--no-f16-window", "--no-extra"] + mf, 300)}
    if rank == 0:
        gb = args.batch * world
        arch_name = {"resnet34": "ResNet-34", "resnet101": "ResNet-101"}[args.arch]
        frames_desc = ("%d-frame" % args.frames) if var_x is None else "%d..%d-frame (one length per step)" % tuple(args.frames_range)
        out = {
            "metric": ("utterances/sec (%s x 80 fbank, bs%d/GPU), %s + AAM-softmax training step" % (frames_desc, args.batch, arch_name))
            if args.mode == "train" else "utterances/sec, eval-mode embedding extraction (predict), %s" % arch_name,
            "value": round(gb * args.steps / dt, 2), "unit": "utt/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if ops.SPLIT == 0 else (
                "f32 (operands as 2 fp16 terms of value x 2^k, 3 cross products, f32 accumulate)" if ops.SPLIT == 3 else
                "f32 (operands as 3 exact bf16 terms, %d cross products, f32 accumulate)" % ops.SPLIT),
            "data": "synthetic",
            "config": {"workload": ("%s: %s + AAM-softmax (m 0.2, s 30), %d speakers, "
                                    "%s x %d fbank, per-GPU batch %d, fwd+CE+bwd+SGD(0.9, wd 5e-4)"
                                    % ("BASELINE configs[1]" if headline else (
                                        "BASELINE configs[3] shape on %d GPU(s)" % world if (args.arch == "resnet101" and nspk == 5994
                                                                                           and var_x is not None)
                                        else "non-headline case"), arch_name, nspk, frames_desc, FEAT, args.batch))
                       if args.mode == "train" else
                       ("%s: eval-mode predict() of %s (BatchNorm folded into the conv epilogues), %s x %d fbank, batch %d, inputs "
                        "resident in HBM; the ark -> reader -> writer pipeline around it is tools/c5_extract.py"
                        % ("BASELINE configs[4] kernel path" if (args.arch == "resnet34" and args.batch == 512) else "non-headline case",
                           arch_name, frames_desc, FEAT, args.batch)),
                       "global_batch": gb, "frames": args.frames if var_x is None else list(args.frames_range),
                       "feat_dim": FEAT, "speakers": nspk,
                       "parallelism": "dp%d" % world,
                       "launch": launch_desc,
                       "collective": (None if not red.active else "%s all-reduce(%s) of the gradient arena per ResNet stage, "
                                      "%d rank(s), %d calls" % (dist.get_backend(), red.op, world, red.calls)),
                       "mfma": mfma_mode},
            "first_loss": round(first_loss, 4), "final_loss": round(lossv, 4),
            "loss_decreased_on_the_fixed_batch": bool(lossv < first_loss) if args.mode == "train" else None,
            "graph_replay_repacks_weights": repack_ok, "eer": eer,
            "kernel_launches_per_step": (sum(v["launches_per_step"] for v in roofline["all_kernels"].values())
                                         if roofline else None),
            "roofline": roofline, "cpu_baseline": cpu, "fp32_operand_mfma": native, "ingest": ingest, "extra": extra, "f16_window": f16_window,
            "embedding_cosine_delta_vs_oracle": parity,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()          # rank 0 is still in its instrumented pass while the others are done: leave together
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
