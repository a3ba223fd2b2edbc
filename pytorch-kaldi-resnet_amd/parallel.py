"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

Reference behaviour (scripts/train_resnet.py:148-149,183-185: DistributedDataParallel over NCCL): gradients
are summed over ranks and divided by the world size; BatchNorm statistics stay per rank; rank 0's state is
what gets checkpointed.  Here the gradients already live in one flat arena ordered stem -> layer1..4 -> head,
and backward finishes stages in the reverse order, so each stage's slice is all-reduced on a side stream as
soon as its last weight gradient is enqueued (deepest first), overlapping the remaining backward convs.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 27.8 MB ring all-reduce moves 48.6 MB per GPU,
~0.3-0.6 ms, against >= 100 ms of backward - a handful of large slices beats many small buckets.
The division by world size is folded into the SGD kernel (grad_scale), not a separate pass.
"""
import os

import torch
import torch.distributed as dist


def reducer_forced():
    """SPK_FORCE_REDUCER=1: run the whole data-parallel machinery - stage-segmented graph replay, the all-reduce on the
    communication stream between the replays, finish() - on a process group of ONE rank.  It exists so that the RCCL path
    (communicator init, collectives enqueued between hipGraph replays that share one memory pool, stream hand-offs) can be
    exercised on a box with a single GPU; RCCL refuses two ranks on one device ("Duplicate GPU detected")."""
    return os.environ.get("SPK_FORCE_REDUCER", "0") == "1"


def stage_slices(model):
    """name -> (start, end) element ranges of the flat arena for: stem, layer1..layer4, head."""
    names = [n for n, _ in model.named_parameters()]
    params = list(model.parameters())
    offs = model._offsets
    total = model.flat_parameters().numel()
    out = {}

    def stage_of(n):
        if n.startswith("res.layer"):
            return n.split(".")[1]
        if n.startswith("res."):
            return "stem"
        return "head"

    for n, p, o in zip(names, params, offs):
        s = stage_of(n)
        end = o + (p.numel() + 3) // 4 * 4
        lo, hi = out.get(s, (o, end))
        out[s] = (min(lo, o), max(hi, end))
    assert max(hi for _, hi in out.values()) == total
    return out


class GradAllReducer:
    """Overlapped, stage-bucketed all-reduce(sum) of the gradient arena. Works with any torch.distributed
    backend: 'nccl' (= RCCL on ROCm) on GPUs, 'gloo' in the CPU tests."""

    def __init__(self, model, group=None, force=None):
        """force (default: SPK_FORCE_REDUCER): keep the reducer active on a process group of one rank (see reducer_forced)."""
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if force is None:
            force = reducer_forced()
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        # An in-place SUM over one rank launches nothing in RCCL (ncclLaunchOneRank returns for in == out); AVG goes through
        # its one-rank pre-multiply kernel (x * 1/1: bit-exact), so a forced one-rank run really puts an RCCL kernel on the
        # RCCL stream between the graph replays.  More than one rank: SUM, the 1/world is folded into the SGD kernel.
        self.op = dist.ReduceOp.SUM
        if self.active and self.world == 1 and dist.get_backend(group) == "nccl":
            self.op = dist.ReduceOp.AVG
        self.calls = 0                # collectives issued (tests / logs)
        self.slices = stage_slices(model)
        self.comm_stream = torch.cuda.Stream() if model.flat_parameters().is_cuda else None
        self._works = []

    def on_stage_done(self, name):
        """Called by Engine.backward right after the stage's last gradient kernel was enqueued."""
        if not self.active:
            return
        self.calls += 1
        lo, hi = self.slices[name]
        g = self.model.flat_grads()[lo:hi]
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._works.append(dist.all_reduce(g, op=self.op, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(g, op=self.op, group=self.group, async_op=True))

    def finish(self):
        """Make the compute stream wait for every outstanding all-reduce (no host sync on GPU)."""
        for w in self._works:
            w.wait()
        self._works = []
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def allreduce_all(self):
        """Non-overlapped variant: one all-reduce over the whole arena (used by the autograd path)."""
        if not self.active:
            return
        self.calls += 1
        dist.all_reduce(self.model.flat_grads(), op=self.op, group=self.group)

    def broadcast_parameters(self, src=0):
        """DDP constructor semantics: rank 0's parameters and buffers win."""
        if not self.active:
            return
        dist.broadcast(self.model.flat_parameters(), src, group=self.group)
        for b in self.model.buffers():
            dist.broadcast(b, src, group=self.group)
        self.model.mark_weights_changed()
