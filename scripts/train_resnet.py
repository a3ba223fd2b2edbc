#!/usr/bin/env python3
"""Training entry point on MI355X - same flags, log line format and checkpoint format as the reference's
scripts/train_resnet.py (argparse :25-91, main_worker :134-289, train :292-335, validate :338-379,
save_checkpoint :382-385), one process per GPU, RCCL gradient all-reduce overlapped with backward.

Differences that are the point of this build: the model is pytorch_kaldi_resnet_amd.NeuralSpeakerModel
(hand-written HIP kernels), the step uses the fused forward+CE+backward path, SGD is one fused kernel over the
flat arena, `--arch` really selects the trunk, and meters accumulate on the device (one host sync per
--print-freq steps instead of `.item()` every step).
"""
import argparse
import os
import random
import shutil
import sys
import time
import warnings

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

parser = argparse.ArgumentParser(description="ResNet speaker-embedding training (MI355X-native)")
parser.add_argument("--train-list", type=str, help="training scp")
parser.add_argument("--cv-list", type=str, help="cv scp")
parser.add_argument("--utt2spkid", type=str, help="utt2spkid")
parser.add_argument("--input-dim", type=int, required=True, help="input feature dimension")
parser.add_argument("--spk-num", type=int, required=True, help="number of speakers")
parser.add_argument("--pooling", type=str, default="mean", help="mean or mean+std")
parser.add_argument("--loss-type", type=str, default="softmax", help="softmax, AAM or AAM-v1")
parser.add_argument("--margin", type=float, default=0.2, help="margin for AAM")
parser.add_argument("--scale", type=float, default=30, help="scale for AAM")
parser.add_argument("--dataset", type=str, default="v1", help="v1 or v2")
parser.add_argument("--min-chunk-size", default=200, type=int)
parser.add_argument("--max-chunk-size", default=400, type=int)
parser.add_argument("--log-dir", type=str, required=True, help="logging directory")
parser.add_argument("-a", "--arch", metavar="ARCH", default="resnet34", help="resnet18/34/50/101")
parser.add_argument("-j", "--workers", default=2, type=int, metavar="N")
parser.add_argument("--epochs", default=10, type=int, metavar="N")
parser.add_argument("--start-epoch", default=0, type=int, metavar="N")
parser.add_argument("-b", "--batch-size", default=128, type=int, metavar="N",
                    help="total batch size over all GPUs of this node")
parser.add_argument("--lr", "--learning-rate", default=0.1, type=float, dest="lr")
parser.add_argument("--lr-final", "--final-learning-rate", default=0.0001, type=float, dest="lr_final")
parser.add_argument("--momentum", default=0.9, type=float)
parser.add_argument("--wd", "--weight-decay", default=1e-4, type=float, dest="weight_decay")
parser.add_argument("-p", "--print-freq", default=10, type=int)
parser.add_argument("--resume", default="", type=str)
parser.add_argument("-e", "--evaluate", dest="evaluate", action="store_true")
parser.add_argument("--pretrained", dest="pretrained", type=str)
parser.add_argument("--world-size", default=-1, type=int, help="number of nodes")
parser.add_argument("--rank", default=-1, type=int, help="node rank")
parser.add_argument("--dist-url", default="tcp://127.0.0.1:23456", type=str)
parser.add_argument("--dist-backend", default="nccl", type=str, help="nccl (= RCCL on ROCm)")
parser.add_argument("--seed", default=None, type=int)
parser.add_argument("--gpu", default=None, type=int)
parser.add_argument("--gpu-num", default=-1, type=int)
parser.add_argument("--multiprocessing-distributed", action="store_true")
parser.add_argument("--max-steps", default=-1, type=int, help="stop each epoch after N steps (smoke runs)")
parser.add_argument("--native-reader", action="store_true",
                    help="read training batches with the native C++ ark reader (libspkio: pread of the cropped frames on a "
                         "thread pool into pinned memory) instead of Dataset/DataLoader worker processes; --dataset v1 only")
parser.add_argument("--var-chunk", action="store_true",
                    help="variable-length training (BASELINE configs[3]): one chunk length per batch, uniform over "
                         "--min-chunk-size, +quantum, ... <= --max-chunk-size, seeded, identical on every rank (the reference "
                         "parses --min-chunk-size and never reads it, scripts/train_resnet.py:237; its Dataset can draw lengths "
                         "per sample, scripts/datasets.py:40-43, which default collation cannot batch).  Off = the reference's "
                         "behaviour: every chunk is --max-chunk-size frames")
parser.add_argument("--chunk-quantum", default=8, type=int, help="step between the chunk lengths of --var-chunk (one captured "
                                                                  "hipGraph per distinct length, one shared memory pool)")
parser.add_argument("--no-graph", action="store_true",
                    help="launch kernels eagerly instead of replaying the step as hipGraph(s); both forms overlap the "
                         "stage-bucketed gradient all-reduce with backward")

best_acc1 = 0


def main():
    args = parser.parse_args()
    if args.seed is not None:
        random.seed(args.seed)
        torch.manual_seed(args.seed)
        warnings.warn("You have chosen to seed training.")
    if args.dist_url == "env://" and args.world_size == -1:
        args.world_size = int(os.environ["WORLD_SIZE"])
    args.distributed = args.world_size > 1 or args.multiprocessing_distributed
    ngpus = torch.cuda.device_count() if args.gpu_num == -1 else min(torch.cuda.device_count(), args.gpu_num)
    if args.multiprocessing_distributed:
        args.world_size = ngpus * args.world_size
        mp.spawn(main_worker, nprocs=ngpus, args=(ngpus, args))
    else:
        main_worker(args.gpu if args.gpu is not None else 0, ngpus, args)


def main_worker(gpu, ngpus_per_node, args):
    global best_acc1
    import numpy as np
    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd.datasets import SequenceDataset, SequenceDataset2
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    from pytorch_kaldi_resnet_amd.optim import FlatSGD
    from pytorch_kaldi_resnet_amd.parallel import GradAllReducer

    args.gpu = gpu
    print("Use GPU: {} for training".format(args.gpu))
    from pytorch_kaldi_resnet_amd import tiling
    # like `cudnn.benchmark = True` (reference train_resnet.py:231): tune tiles on first use.  Timing-based, so two runs
    # may pick different tiles (= summation orders); SPK_AUTOTUNE=0 pins the built-in table for bit-reproducible runs.
    tiling.AUTOTUNE = os.environ.get("SPK_AUTOTUNE", "1") == "1"
    if args.distributed:
        if args.dist_url == "env://" and args.rank == -1:
            args.rank = int(os.environ["RANK"])
        if args.multiprocessing_distributed:
            args.rank = args.rank * ngpus_per_node + gpu
        dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                                rank=args.rank)
    if args.seed is not None:
        np.random.seed(args.seed + max(args.rank, 0))
        # the reference seeds torch in main() only (train_resnet.py:96-104), so its spawned workers initialise from an unseeded
        # generator and rely on DDP's broadcast from rank 0; seeding here too makes a --multiprocessing-distributed run
        # reproduce the --gpu run with the same --seed
        torch.manual_seed(args.seed)
    torch.cuda.set_device(args.gpu)
    print("=> creating model '{}'".format(args.arch))
    model = NeuralSpeakerModel(spk_num=args.spk_num, feat_dim=args.input_dim, pooling=args.pooling, loss=args.loss_type,
                               m=args.margin, s=args.scale, arch=args.arch)
    print("===> Model total parameter: {}".format(sum(p.numel() for p in model.parameters() if p.requires_grad)))
    if args.pretrained:
        if os.path.isfile(args.pretrained):
            print("=> using pre-trained model '{}'".format(args.pretrained))
            ckpt = torch.load(args.pretrained, map_location="cpu", weights_only=True)
            model.loadParameters(ckpt["state_dict"])
        else:
            print("=> no pre-trained model found at '{}'".format(args.pretrained))
            return
    model.cuda(args.gpu)
    world = args.world_size if args.distributed else 1
    if args.distributed:
        args.batch_size = int(args.batch_size / ngpus_per_node)
        args.workers = int((args.workers + ngpus_per_node - 1) / ngpus_per_node)
    print("gpu: {}, batch size: {}, args.workers:{}, ngpus_per_node: {}".format(gpu, args.batch_size, args.workers,
                                                                                ngpus_per_node))
    optimizer = FlatSGD(model, args.lr, momentum=args.momentum, weight_decay=args.weight_decay, grad_scale=1.0 / world)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, args.epochs, eta_min=args.lr_final, last_epoch=-1)
    reducer = GradAllReducer(model)      # active with more than one rank (or SPK_FORCE_REDUCER=1 on a one-rank process group)
    if reducer.active:
        print("=> gradient all-reduce per ResNet stage on the communication stream: backend {}, {} rank(s), op {}".format(
            dist.get_backend(), world, reducer.op))
    if args.resume:
        if os.path.isfile(args.resume):
            print("=> loading checkpoint '{}'".format(args.resume))
            ckpt = torch.load(args.resume, map_location="cpu", weights_only=True)
            args.start_epoch = ckpt["epoch"]
            best_acc1 = ckpt["best_acc1"]
            model.loadParameters(ckpt["state_dict"])
            optimizer.load_state_dict(ckpt["optimizer"])
            # the reference rebuilds the scheduler with eta_min hard-coded to 1e-4 on resume (train_resnet.py:225)
            for g in optimizer.param_groups:
                g.setdefault("initial_lr", args.lr)
            scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, args.epochs, eta_min=0.0001,
                                                                   last_epoch=args.start_epoch - 1)
            print("=> loaded checkpoint '{}' (epoch {})".format(args.resume, ckpt["epoch"]))
        else:
            print("=> no checkpoint found at '{}'".format(args.resume))
    reducer.broadcast_parameters(0)   # DDP constructor semantics: rank 0's weights and buffers

    DS = SequenceDataset2 if args.dataset == "v2" else SequenceDataset
    chunk = args.max_chunk_size if args.dataset == "v2" else [args.max_chunk_size]
    train_sampler = None
    var = (args.min_chunk_size, args.max_chunk_size, args.chunk_quantum) if args.var_chunk else None
    if var is not None:
        assert 0 < var[0] <= var[1], "--var-chunk needs 0 < --min-chunk-size <= --max-chunk-size"
        print("=> variable-length training: one chunk length per batch in [{}, {}] step {}".format(*var))
    if args.native_reader and args.dataset != "v2":
        from pytorch_kaldi_resnet_amd.ingest import NativeTrainLoader
        train_loader = NativeTrainLoader(args.train_list, args.utt2spkid, args.max_chunk_size, args.batch_size,
                                         rank=max(args.rank, 0), world=world, seed=args.seed or 0,
                                         threads=max(1, args.workers), device="cuda:%d" % args.gpu, chunk_range=var)
        train_sampler = train_loader          # set_epoch() reshuffles, like DistributedSampler
    else:
        train_dataset = DS(scp_file=args.train_list, utt2spkid_file=args.utt2spkid, chunk_size=chunk)
        if args.distributed:
            train_sampler = torch.utils.data.distributed.DistributedSampler(train_dataset, num_replicas=args.world_size,
                                                                            rank=args.rank, shuffle=True)
        if var is not None:
            # the chunk length rides on the index ((index, T) pairs), so worker processes need no shared state
            from pytorch_kaldi_resnet_amd.datasets import ChunkBatchSampler
            inner = train_sampler if train_sampler is not None else torch.utils.data.RandomSampler(train_dataset)
            train_sampler = ChunkBatchSampler(inner, args.batch_size, var[0], var[1], var[2], seed=args.seed or 0)
            train_loader = torch.utils.data.DataLoader(train_dataset, batch_sampler=train_sampler, num_workers=args.workers,
                                                       pin_memory=True)
        else:
            train_loader = torch.utils.data.DataLoader(train_dataset, batch_size=args.batch_size,
                                                       shuffle=(train_sampler is None), num_workers=args.workers,
                                                       pin_memory=True, sampler=train_sampler, drop_last=False)
    print("=> args.world_size: {}, args.rank: {}, args.batch_size: {}, train_loader samples: {}".format(
        args.world_size, args.rank, args.batch_size, len(train_loader)))
    val = DS(scp_file=args.cv_list, utt2spkid_file=args.utt2spkid, chunk_size=chunk)
    val_loader = torch.utils.data.DataLoader(val, batch_size=args.batch_size, shuffle=False, num_workers=args.workers,
                                             pin_memory=True)
    if args.evaluate:
        validate(val_loader, model, args)
        return
    os.makedirs(args.log_dir, exist_ok=True)
    for epoch in range(args.start_epoch, args.epochs):
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        train(train_loader, model, optimizer, reducer, epoch, args, world)
        acc1 = validate(val_loader, model, args)
        scheduler.step()
        is_best = acc1 > best_acc1
        best_acc1 = max(acc1, best_acc1)
        if not args.multiprocessing_distributed or (args.multiprocessing_distributed and args.rank % ngpus_per_node == 0):
            sd = model.state_dict()
            if args.distributed:   # the reference saves a DDP-wrapped model: keys carry the 'module.' prefix
                sd = {"module." + k: v for k, v in sd.items()}
            save_checkpoint({"epoch": epoch + 1, "arch": args.arch, "state_dict": sd,
                             "best_acc1": torch.as_tensor(float(best_acc1)), "optimizer": optimizer.state_dict()},
                            is_best, os.path.join(args.log_dir, "checkpoint_epoch{}.pth.tar".format(epoch)))
    if args.distributed:
        dist.destroy_process_group()


class DeviceMeter:
    """AverageMeter (train_resnet.py:388-409) whose running sums live on the device."""

    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.sum = None
        self.count = 0
        self.val = 0.0

    def update(self, val, n=1):
        self.val = val
        self.sum = val * n if self.sum is None else self.sum + val * n
        self.count += n

    @property
    def avg(self):
        return float(self.sum) / max(self.count, 1) if self.sum is not None else 0.0

    def __str__(self):
        f = "{name} {val" + self.fmt + "} ({avg" + self.fmt + "})"
        return f.format(name=self.name, val=float(self.val), avg=self.avg)


def _progress(prefix, i, n, meters):
    digits = len(str(n))
    fmt = "[{:" + str(digits) + "d}/" + ("{:" + str(digits) + "d}").format(n) + "]"
    print("\t".join([prefix + fmt.format(i)] + [str(m) for m in meters]))
    sys.stdout.flush()


def train(loader, model, optimizer, reducer, epoch, args, world):
    bt, dt_ = DeviceMeter("Time", ":6.3f"), DeviceMeter("Data", ":6.3f")
    losses, top1, top5 = DeviceMeter("Loss", ":.4e"), DeviceMeter("Acc@1", ":6.2f"), DeviceMeter("Acc@5", ":6.2f")
    model.train()
    eng = model.engine()
    # one captured step per (batch, chunk length), one shared memory pool (engine.GraphedStepCache): a fixed-length run
    # captures once, --var-chunk once per distinct length
    cache = getattr(eng, "_graph_cache", None)
    if cache is None and not args.no_graph:
        from pytorch_kaldi_resnet_amd.engine import GraphedStepCache
        cache = eng._graph_cache = GraphedStepCache(eng, segmented=reducer.active)
    t_enq, n_enq = 0.0, 0
    end = time.time()
    t_epoch, n_utt = time.time(), 0
    for i, (audios, target) in enumerate(loader):
        if 0 <= args.max_steps <= i:
            break
        dt_.update(time.time() - end)
        audios = audios.cuda(args.gpu, non_blocking=True)
        target = target.cuda(args.gpu, non_blocking=True).long()
        if not args.no_graph and audios.size(0) == args.batch_size:
            # hipGraph replay: weight re-pack + fwd + CE + bwd; with world > 1 six stage segments with the stage's
            # all-reduce enqueued on the communication stream between them (overlaps the remaining backward)
            known = (audios.size(0), audios.size(2)) in cache.steps
            t_q = time.time()
            loss, _, rank = cache(audios, target, reducer.on_stage_done if reducer.active else None)
            reducer.finish()
            if known:
                t_enq, n_enq = t_enq + time.time() - t_q, n_enq + 1
        else:
            optimizer.zero_grad(set_to_none=True)
            loss, _, rank = eng.loss_and_grad(audios, target, reducer.on_stage_done if reducer.active else None)
            reducer.finish()
        optimizer.step()
        n = audios.size(0)
        n_utt += n
        losses.update(loss[0], n)
        top1.update((rank < 1).float().mean() * 100, n)
        top5.update((rank < 5).float().mean() * 100, n)
        bt.update(time.time() - end)
        end = time.time()
        if i % args.print_freq == 0:
            _progress("Epoch: [{}]".format(epoch), i, len(loader), [bt, dt_, losses, top1, top5])
    torch.cuda.synchronize()
    print(" * epoch {} train throughput {:.1f} utt/s (this rank)".format(epoch, n_utt / max(time.time() - t_epoch, 1e-9)))
    if reducer.active:
        print(" * epoch {} collectives issued so far: {}".format(epoch, reducer.calls))
    if cache is not None and n_enq:
        print(" * epoch {} captured steps: {} (chunk lengths {}), host enqueue {:.2f} ms/step over {} replays".format(
            epoch, len(cache), sorted({k[1] for k in cache.steps}), 1e3 * t_enq / n_enq, n_enq))


def validate(loader, model, args):
    from pytorch_kaldi_resnet_amd import ops
    bt = DeviceMeter("Time", ":6.3f")
    losses, top1, top5 = DeviceMeter("Loss", ":.4e"), DeviceMeter("Acc@1", ":6.2f"), DeviceMeter("Acc@5", ":6.2f")
    model.eval()
    with torch.no_grad():
        end = time.time()
        for i, (audios, target) in enumerate(loader):
            if 0 <= args.max_steps <= i:
                break
            audios = audios.cuda(args.gpu, non_blocking=True)
            target = target.cuda(args.gpu, non_blocking=True).long()
            output = model(audios, target)          # eval-mode BN, AAM margin still applied (train_resnet.py:359)
            loss_row, _, rank = ops.softmax_ce(output, target)
            n = audios.size(0)
            losses.update(ops.mean(loss_row)[0], n)
            top1.update((rank < 1).float().mean() * 100, n)
            top5.update((rank < 5).float().mean() * 100, n)
            bt.update(time.time() - end)
            end = time.time()
            if i % args.print_freq == 0:
                _progress("Test: ", i, len(loader), [bt, losses, top1, top5])
        print(" * Acc@1 {:.3f} Acc@5 {:.3f}".format(top1.avg, top5.avg))
    return top1.avg


def save_checkpoint(state, is_best, filename="checkpoint.pth.tar"):
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, os.path.join(os.path.dirname(filename), "model_best.pth.tar"))


if __name__ == "__main__":
    main()
