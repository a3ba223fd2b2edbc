"""The N > 1 data-parallel path on CPU: world_size 2 over gloo (127.0.0.1).  Checks the stage-bucketed gradient
all-reduce (sum over ranks, every element covered exactly once, any completion order), the initial parameter
broadcast and the DistributedSampler sharding used by train/decode."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
    from pytorch_kaldi_resnet_amd.parallel import GradAllReducer, stage_slices
    torch.manual_seed(100 + rank)            # different init per rank, like independent processes
    m = NeuralSpeakerModel(6, 40, "mean+std", "AAM")
    red = GradAllReducer(m)
    red.broadcast_parameters(0)
    flat = m.flat_parameters().clone()
    ref = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(ref, flat)
    same_params = all(torch.equal(r, ref[0]) for r in ref)
    # gradients: rank-dependent ramp; stages complete deepest-first like Engine.backward reports them
    m.attach_grads()
    g = m.flat_grads()
    g.copy_(torch.arange(g.numel(), dtype=torch.float32) * (rank + 1) * 1e-3)
    for name in ["head", "layer4", "layer3", "layer2", "layer1", "stem"]:
        red.on_stage_done(name)
    red.finish()
    expect = torch.arange(g.numel(), dtype=torch.float32) * 1e-3 * sum(r + 1 for r in range(world))
    ok_sum = torch.allclose(g, expect, rtol=1e-6)
    sl = stage_slices(m)
    cover = torch.zeros(g.numel())
    for lo, hi in sl.values():
        cover[lo:hi] += 1
    ok_cover = bool((cover == 1).all())
    # .grad views still alias the arena after the reduction
    ok_alias = m.fc1.weight.grad.data_ptr() == g.data_ptr() + 4 * m._offsets[[n for n, _ in m.named_parameters()].index("fc1.weight")]
    sampler = torch.utils.data.distributed.DistributedSampler(list(range(11)), num_replicas=world, rank=rank, shuffle=True)
    sampler.set_epoch(3)
    idx = list(sampler)
    q.put((rank, same_params, ok_sum, ok_cover, ok_alias, idx))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_allreduce_and_broadcast():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx_all = []
    for rank, same_params, ok_sum, ok_cover, ok_alias, idx in res:
        assert same_params and ok_sum and ok_cover and ok_alias
        idx_all += idx
    assert sorted(set(idx_all)) == list(range(11)) and len(idx_all) == 12   # padded to equal shards
