"""CPU, world_size 2, gloo: the data-parallel exchange of the N>1 path -- one flat gradient
all-reduce per step (mucon_amd/mucon/trainers.all_reduce_gradients; bench.py uses the same
flatten / all-reduce / scatter-back pattern), disjoint video shards per rank, and the all-reduced
MoF counters of the evaluator.  The kernels themselves need a GPU; everything around them is
exercised here with a stand-in model."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mucon_amd.mucon.trainers import GradBucket, all_reduce_gradients, broadcast_parameters

    torch.manual_seed(rank)  # replicas that start DIFFERENT: the broadcast has to make them equal
    model = nn.Sequential(nn.Linear(16, 8), nn.ReLU(), nn.Linear(8, 4), nn.Linear(4, 2), nn.Linear(3, 3))
    broadcast_parameters(model, 0)
    start = [p.detach().clone() for p in model.parameters()]
    bucket = GradBucket(model.parameters())
    out = {"start": start, "steps": []}
    for step in range(2):                # twice: the second step meets .grad tensors that are views of the flat buffer
        for p in model.parameters():
            p.grad = None
        x = torch.randn(5, 16, generator=torch.Generator().manual_seed(100 + rank + 10 * step))
        h = model[1](model[0](x))
        # model[2], model[3] receive a gradient on rank 0 only; model[4] on no rank; which tensors share a storage differs too
        y = model[3](model[2](h)) if rank == 0 else h[:, :2] * 1.0
        y.sum().backward()
        local = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
        if rank == 1:    # like the HIP encoder's backward: this rank's first-layer gradients are views of ONE buffer
            ps = list(model[0].parameters())
            flat = torch.cat([p.grad.reshape(-1) for p in ps])
            off = 0
            for p in ps:
                p.grad = flat[off: off + p.numel()].view_as(p)
                off += p.numel()
        all_reduce_gradients(model, world, bucket)
        for p in model[4].parameters():
            assert p.grad is None                       # unused everywhere: stays None, as in a single process
        for p in list(model.parameters())[:6]:
            assert p.grad.untyped_storage().data_ptr() == bucket.flat.untyped_storage().data_ptr()
        out["steps"].append({"local": local, "avg": [None if p.grad is None else p.grad.clone() for p in model.parameters()]})
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))

    # shard arithmetic of SimpleTrainer.train_epoch: disjoint, same count on every rank
    n = 11
    order = torch.randperm(n, generator=torch.Generator().manual_seed(1)).tolist()
    steps = n // world
    mine = [order[s * world + rank] for s in range(steps)]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        flat = sum(gathered, [])
        assert len(flat) == len(set(flat)) == steps * world
    # evaluator counters: a few scalars summed over ranks
    counts = torch.tensor([3.0 + rank, 10.0], dtype=torch.float64)
    dist.all_reduce(counts)
    assert counts.tolist() == [7.0, 20.0]
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    for a, b in zip(r0["start"], r1["start"]):
        assert torch.equal(a, b)                        # broadcast_parameters: identical replicas
    for s0, s1 in zip(r0["steps"], r1["steps"]):
        for a0, a1, l0, l1 in zip(s0["avg"], s1["avg"], s0["local"], s1["local"]):
            if l0 is None and l1 is None:
                assert a0 is None and a1 is None        # no gradient anywhere: None on every rank
                continue
            assert torch.equal(a0, a1)                  # every rank ends with the same averaged gradient
            z = torch.zeros_like(a0)
            want = ((l0 if l0 is not None else z) + (l1 if l1 is not None else z)) / 2
            torch.testing.assert_close(a0, want, rtol=1e-6, atol=1e-7)
