#!/usr/bin/env python3
"""Embedding extraction on MI355X - same flags and output format as the reference's scripts/decode.py
(argparse :26-61, main_worker :101-182, SequenceGenerator :185-208): load a checkpoint through loadParameters,
run predict() under eval(), write text-ark lines 'utt [ v0 ... v255 ]' with str(np.float32) to <out-path>/<gpu>.
One process per GPU shards the scp with a DistributedSampler; no collective is issued.
Unlike the reference (decode.py:204 is only correct at per-process batch size 1) batches of equal-length
utterances are supported: --chunk-size T crops, --chunk-size -1 needs equal lengths within a batch.
"""
import argparse
import os
import time
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

parser = argparse.ArgumentParser(description="speaker-embedding extraction (MI355X-native)")
parser.add_argument("--spk_num", type=int, help="number of speakers")
parser.add_argument("--arch", type=str, required=True)
parser.add_argument("--input-dim", type=int, required=True)
parser.add_argument("--pooling", type=str, required=True, help="mean or mean+std")
parser.add_argument("--chunk-size", default=-1, type=int)
parser.add_argument("--model-path", help="checkpoint (.pth.tar)")
parser.add_argument("--world-size", default=-1, type=int)
parser.add_argument("--rank", default=-1, type=int)
parser.add_argument("-j", "--workers", default=2, type=int)
parser.add_argument("-b", "--batch-size", default=128, type=int, help="total batch size over the node's GPUs")
parser.add_argument("--dist-url", default="tcp://127.0.0.1:23456", type=str)
parser.add_argument("--dist-backend", default="nccl", type=str)
parser.add_argument("--seed", default=None, type=int)
parser.add_argument("--gpu", type=int)
parser.add_argument("--gpu-num", default=-1, type=int)
parser.add_argument("--decode-scp", help="decode.scp")
parser.add_argument("--out-path", help="output directory")
parser.add_argument("--multiprocessing-distributed", action="store_true")
parser.add_argument("--native-reader", action="store_true",
                    help="read whole utterances with the C++ ark reader, bucketed by length so that any --batch-size works "
                         "with variable-length utterances (the reference is limited to per-process batch 1)")
parser.add_argument("--out-format", default="text", choices=["text", "fv"],
                    help="text = the reference's 'utt [ v0 ... ]' lines (str(np.float32), ~0.2 ms/utt of Python formatting); "
                         "fv = binary Kaldi float-vector ark, read by the same scoring scripts")


def main():
    args = parser.parse_args()
    if args.dist_url == "env://" and args.world_size == -1:
        args.world_size = int(os.environ["WORLD_SIZE"])
    args.distributed = args.world_size > 1 or args.multiprocessing_distributed
    ngpus = torch.cuda.device_count() if args.gpu_num == -1 else min(torch.cuda.device_count(), args.gpu_num)
    if args.multiprocessing_distributed:
        args.world_size = ngpus * args.world_size
        mp.spawn(main_worker, nprocs=ngpus, args=(ngpus, args))
    else:
        main_worker(args.gpu if args.gpu is not None else 0, ngpus, args)


def collate(batch):
    feats = [b[0] for b in batch]
    t0 = feats[0].shape[1]
    if any(f.shape[1] != t0 for f in feats):
        raise RuntimeError("utterances of different length in one batch: use --batch-size = number of GPUs "
                           "(per-process batch 1, as the reference's recipes do) or --chunk-size T")
    return torch.from_numpy(np.stack(feats)), [b[1] for b in batch]


def main_worker(gpu, ngpus_per_node, args):
    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd.datasets import EmbeddingDataset
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    args.gpu = gpu
    if args.distributed:
        if args.dist_url == "env://" and args.rank == -1:
            args.rank = int(os.environ["RANK"])
        if args.multiprocessing_distributed:
            args.rank = args.rank * ngpus_per_node + gpu
        dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                                rank=args.rank)
        args.batch_size = max(1, int(args.batch_size / ngpus_per_node))
        args.workers = int((args.workers + ngpus_per_node - 1) / ngpus_per_node)
    torch.cuda.set_device(args.gpu)
    print("=> creating model '{}'".format(args.arch))
    model = NeuralSpeakerModel(spk_num=args.spk_num, feat_dim=args.input_dim, pooling=args.pooling, arch=args.arch)
    if not (args.model_path and os.path.isfile(args.model_path)):
        print("=> no checkpoint found at '{}'".format(args.model_path))
        return
    print("=> loading checkpoint '{}'".format(args.model_path))
    ckpt = torch.load(args.model_path, map_location="cpu", weights_only=True)
    model.loadParameters(ckpt["state_dict"])
    print("=> loaded checkpoint '{}' (epoch {})".format(args.model_path, ckpt.get("epoch")))
    model.cuda(args.gpu)
    os.makedirs(args.out_path, exist_ok=True)
    if args.native_reader:
        native_generator(model, args)
        if args.distributed:
            dist.destroy_process_group()
        return
    ds = EmbeddingDataset(scp_file=args.decode_scp, chunk_size=args.chunk_size)
    sampler = None
    if args.distributed:
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=args.world_size, rank=args.rank,
                                                                  shuffle=True)
    loader = torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=False, num_workers=args.workers,
                                         pin_memory=True, sampler=sampler, collate_fn=collate)
    print("=> args.world_size: {}, args.rank: {}, loaded embedding samples num: {}".format(args.world_size, args.rank,
                                                                                         len(loader)))
    sequence_generator(loader, model, args.out_path, args)
    if args.distributed:
        dist.destroy_process_group()


def _write(f, utts, pred, fmt):
    from pytorch_kaldi_resnet_amd import kaldi_io
    for i in range(pred.shape[0]):
        if fmt == "text":
            f.write(utts[i] + " [ " + " ".join(map(str, pred[i, :].flatten())) + " ]\n")
        else:
            kaldi_io.write_vec_flt(f, np.ascontiguousarray(pred[i]), key=utts[i])


def sequence_generator(loader, model, out_path, args):
    model.eval()
    name = str(args.gpu) if args.distributed else "alone"
    with open(os.path.join(out_path, name), "w" if args.out_format == "text" else "wb") as f, torch.no_grad():
        for audios, utts in loader:
            pred = model.predict(audios.cuda(args.gpu, non_blocking=True)).cpu().numpy()
            _write(f, utts, pred, args.out_format)


def native_generator(model, args):
    """Length-bucketed extraction through libspkio: utterances of equal frame count share a batch (whole utterance when
    --chunk-size -1, else the first chunk-size frames... the reference crops at random; extraction of a fixed window is
    deterministic here), each rank takes every world-th batch."""
    from pytorch_kaldi_resnet_amd.ingest import ArkTable
    tab = [l.rstrip().split(None, 1) for l in open(args.decode_scp)]
    utts = [u for u, _ in tab]
    table = ArkTable([r for _, r in tab])
    print("Totally " + str(len(utts)) + " samples")
    T_of = table.rows if args.chunk_size < 0 else np.minimum(table.rows, args.chunk_size)
    if args.chunk_size >= 0:
        assert (table.rows >= args.chunk_size).all(), "utterance shorter than --chunk-size"
    order = np.argsort(T_of, kind="stable")
    batches, i = [], 0
    while i < len(order):
        j = i
        while j < len(order) and j - i < args.batch_size and T_of[order[j]] == T_of[order[i]]:
            j += 1
        batches.append(order[i:j])
        i = j
    rank, world = (max(args.rank, 0), max(args.world_size, 1)) if args.distributed else (0, 1)
    model.eval()
    name = str(args.gpu) if args.distributed else "alone"
    F = int(table.cols[0])
    mine = batches[rank::world]
    # three-stage pipeline: a reader thread fills the next pinned batch (pread + transpose in libspkio, GIL released)
    # and a writer thread formats the previous batch's embeddings (natively, byte-identical to the reference's
    # str(np.float32) text) while the GPU works on the current one
    from concurrent.futures import ThreadPoolExecutor
    from pytorch_kaldi_resnet_amd import ingest, kaldi_io
    pinned = {}

    def load(n):
        b = mine[n]
        T = int(T_of[b[0]])
        key = (n & 1, len(b), T)
        buf = pinned.get(key)
        if buf is None:
            for k in [k for k in pinned if k[0] == key[0]]:
                del pinned[k]                    # one live buffer per parity: its previous batch has been consumed
            buf = pinned[key] = torch.empty(len(b), F, T).pin_memory()
        table.read_crop(b, [0] * len(b), T, buf, max(1, args.workers))
        return buf

    def emit(f, keys, pred):
        if args.out_format == "text":
            f.write(ingest.format_text_vectors(keys, pred, max(1, args.workers)))
        else:
            for i in range(pred.shape[0]):
                kaldi_io.write_vec_flt(f, np.ascontiguousarray(pred[i]), key=keys[i])

    with open(os.path.join(args.out_path, name), "wb") as f, torch.no_grad(), ThreadPoolExecutor(1) as rd, \
            ThreadPoolExecutor(1) as wr:
        nxt = rd.submit(load, 0) if mine else None
        pending = None
        t0, done = time.time(), 0
        for n, b in enumerate(mine):
            buf = nxt.result()
            nxt = rd.submit(load, n + 1) if n + 1 < len(mine) else None
            pred = model.predict(buf.cuda(args.gpu, non_blocking=True)).cpu().numpy()
            if pending is not None:
                pending.result()
            pending = wr.submit(emit, f, [utts[k] for k in b], pred)
            done += len(b)
        if pending is not None:
            pending.result()
        dt = time.time() - t0
        print("=> extracted {} utterances in {:.2f} s ({:.0f} utt/s, read + predict + write)".format(done, dt, done / max(dt, 1e-9)))


if __name__ == "__main__":
    main()
