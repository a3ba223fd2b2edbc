#!/usr/bin/env python3
"""Headline benchmark: training throughput of the ResNet-34 + AAM-softmax hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one full training iteration of BASELINE.json configs[1] on one batch of synthetic input that is
already resident in HBM: forward (train-mode BN) + AAM margin + cross-entropy + hand-written backward +
fused SGD (+ overlapped RCCL gradient all-reduce when N > 1; per-GPU batch fixed at 256 = weak scaling).
Prints ONE JSON line (see the repo contract) with two extra objects:
  roofline     - dominant kernel (by total device time in an instrumented pass of the same steps), its
                 algorithmic FLOPs per launch / average launch duration measured with HIP events on the launch
                 stream, against the dense fp32 MFMA peak of MI355X_MICROARCH.md (157.3 TFLOP/s)
  cpu_baseline - the CPU oracle (oracle/spk_oracle.py, a torch-CPU port of the reference path) timed on this
                 host's cores on a bounded sample of the same workload (rank 0, N = 1 only)
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
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table: dense bf16 matrix (no sparsity)


def mfma_peak(kernel_label):
    """Peak rate of ALGORITHMIC fp32 FLOPs for a kernel: the fp32 matrix peak for fp32-operand kernels; for the bf16-split
    kernels every fp32 multiply-add costs 6 (or 9) bf16 ones, so the dense bf16 peak divided by that."""
    inner = kernel_label[kernel_label.find("<") + 1:kernel_label.rfind(">")].split(",") if "<" in kernel_label else []
    terms = 0
    if kernel_label.startswith("conv_mfma_kernel") and len(inner) == 4:
        terms = int(inner[3])
    elif kernel_label.startswith("conv_wgrad_split_kernel") and len(inner) == 4:
        terms = int(inner[2])
    if terms:
        return PEAK_BF16_MFMA_TFLOPS / terms, "bf16 dense peak / %d cross products per fp32 multiply-add" % terms
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
    ap.add_argument("--mfma", choices=["bf16x6", "bf16x9", "f32"], default=None,
                    help="operand mode of the 3x3 convolutions (default: the package default, bf16x6 = fp32 operands as three "
                         "exact bf16 terms, 6 cross products on the bf16 MFMA, fp32 accumulate; f32 = native fp32 MFMA)")
    ap.add_argument("--autotune", action="store_true", help="time candidate tiles on first use of a launch shape (SPK_AUTOTUNE=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=3)
    return ap.parse_args()


def cpu_baseline(args):
    """Oracle train step (fwd + CE + autograd bwd + SGD) on the host cores; bounded sample."""
    from oracle import spk_oracle as O
    from oracle import weights as W
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))   # the GPU box's CPU share is 16 cores per GPU
    bs = args.cpu_batch
    st = O.to_torch_state(W.make_state(0, SPK, FEAT, "mean+std", "AAM", "resnet34"))
    x, y = W.make_input(1234, bs, FEAT, args.frames, SPK)
    x, y = torch.from_numpy(x), torch.from_numpy(y)
    bufs = {}
    O.train_step(st, bufs, x, y, 1e-4, weight_decay=5e-4)   # warm-up
    log("cpu warm-up step done (%d threads)" % torch.get_num_threads())
    t0 = time.time()
    for i in range(args.cpu_steps):
        O.train_step(st, bufs, x, y, 1e-4, weight_decay=5e-4)
        log("cpu step %d done" % i)
    dt = (time.time() - t0) / args.cpu_steps
    return {"value": round(bs / dt, 3), "unit": "utt/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d train steps of bs%d x %d frames x %d mel, ResNet-34+AAM S=%d, torch %s CPU oracle "
                      "(%.2f s/step)" % (args.cpu_steps, bs, args.frames, FEAT, SPK, torch.__version__, dt)}


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
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    # rehearsal hooks (one-GPU box): SPK_FORCE_DEVICE pins every rank to one device, SPK_DIST_BACKEND=gloo replaces
    # RCCL so the N > 1 control flow can be exercised where only one GPU exists.  Never set by the driver.
    if "SPK_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["SPK_FORCE_DEVICE"])
    backend = os.environ.get("SPK_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
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
        lens = [rng.randint(lo, hi) for _ in range(8)]      # 8 distinct lengths, cycled: --warmup 8 touches each once
        sched = [lens[i % 8] for i in range(args.warmup + args.steps)]
        var_x = {t: torch.randn(args.batch, FEAT, t, device=dev, generator=gen) for t in sorted(set(sched))}
        args.no_graph = True
    step_no = [0]
    eng = model.engine()
    if os.environ.get("SPK_SIDE_STREAM", "1") == "0":
        eng.use_side_stream = False

    if args.mode == "predict":
        model.eval()

        def step():
            return model.predict(cur_x()).sum()
    else:
        step = None

    # Default launch mode: the whole forward + CE + backward replayed as ONE hipGraph (host-load independent); with
    # N > 1 the flat 27.8 MB gradient arena is then all-reduced in one RCCL call (~0.5 ms of a ~110 ms step) before
    # SGD.  --no-graph launches eagerly and overlaps stage-bucketed all-reduces with the backward kernels instead.
    graphed = None
    if args.mode == "train" and not args.no_graph:
        from pytorch_kaldi_resnet_amd.engine import GraphedTrainStep
        try:
            graphed = GraphedTrainStep(eng, args.batch, args.frames)   # fwd + CE + bwd as one hipGraph
        except Exception as e:     # keep the benchmark alive: same kernels, launched eagerly
            log("hipGraph capture failed (%s: %s); falling back to eager launches" % (type(e).__name__, e))
            graphed = None
            torch.cuda.synchronize()

    def cur_x():
        if var_x is None:
            return x
        t = sched[step_no[0] % len(sched)]
        step_no[0] += 1
        return var_x[t]

    def train_step():
        if graphed is not None and PROFILE_OFF():
            loss, _, _ = graphed(x, y)
            red.allreduce_all()
            opt.step()
            return loss
        opt.zero_grad(set_to_none=True)
        loss, _, _ = eng.loss_and_grad(cur_x(), y, red.on_stage_done if world > 1 else None)
        red.finish()
        opt.step()
        return loss

    def PROFILE_OFF():
        return ops.PROFILE is None

    if step is None:
        step = train_step
    log("model built, starting warm-up")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
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
    log("timed region done: %.3f s for %d steps (host enqueue %.1f ms/step)" % (dt, args.steps, t_host / args.steps * 1e3))

    roofline = None
    if rank == 0 and not args.no_roofline and args.mode == "train":
        # instrumented pass: same steps, every launch bracketed by HIP events on its launch stream; the side stream
        # for weight gradients is folded into the main stream here so kernels do not overlap while being timed
        eng.use_side_stream = False
        ops.PROFILE = []
        for _ in range(2):
            # rank-local: no collective may be issued here, the other ranks are not in this pass
            opt.zero_grad(set_to_none=True)
            eng.loss_and_grad(x, y, None)
            opt.step()
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        eng.use_side_stream = True
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
        for name, flops, e0, e1 in recs:
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += max(e0.elapsed_time(e1) - overhead_ms, 1e-3) * 1e-3
            a[1] += flops
            a[2] += 1
        name, (tsum, fsum, n) = max(agg.items(), key=lambda kv: kv[1][0])
        ach = fsum / tsum / 1e12
        # HBM traffic of that kernel from the committed PMC passes (rocprofv3 cannot run inside this process)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            ent = json.load(open(pmc))["kernels"].get(name.replace(",", ", "))
            if ent:
                traffic = round(ent["hbm_bytes_per_launch"])
        peak, peak_note = mfma_peak(name)
        roofline = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "peak_note": peak_note,
                    "frac_of_fp32_matrix_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic.json)",
                    "launches_per_step": n // 2, "avg_launch_ms": round(tsum / n * 1e3, 4),
                    "event_bracket_overhead_us": round(overhead_ms * 1e3, 1),
                    "gflop_per_launch": round(fsum / n / 1e9, 3),
                    "all_kernels": {k: {"ms_per_step": round(v[0] / 2 * 1e3, 3),
                                        "tflops": round(v[1] / v[0] / 1e12, 2) if v[1] else None,
                                        "launches_per_step": v[2] // 2} for k, v in sorted(agg.items())}}
    headline = args.arch == "resnet34" and nspk == SPK and var_x is None
    native = None
    if rank == 0 and world == 1 and args.mode == "train" and headline and graphed is not None and ops.SPLIT != 0 \
            and not args.no_roofline:
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
            "dtype": "f32" if ops.SPLIT == 0 else "f32 (operands as 3 exact bf16 terms, %d cross products, f32 accumulate)" % ops.SPLIT,
            "data": "synthetic",
            "config": {"workload": "%s: %s + AAM-softmax (m 0.2, s 30), %d speakers, "
                                   "%s x %d fbank, per-GPU batch %d, fwd+CE+bwd+SGD(0.9, wd 5e-4)"
                                   % ("BASELINE configs[1]" if headline else "non-headline case", arch_name, nspk,
                                      frames_desc, FEAT, args.batch),
                       "global_batch": gb, "frames": args.frames if var_x is None else list(args.frames_range),
                       "feat_dim": FEAT, "speakers": nspk,
                       "parallelism": "dp%d" % world, "launch": "hipGraph replay" if graphed is not None else "eager",
                       "mfma": mfma_mode},
            "final_loss": round(lossv, 4),
            "roofline": roofline, "cpu_baseline": cpu, "fp32_operand_mfma": native,
            "embedding_cosine_delta_vs_oracle": parity,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()          # rank 0 is still in its instrumented pass while the others are done: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
