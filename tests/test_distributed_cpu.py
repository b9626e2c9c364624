"""CPU, world_size 2, gloo: the data-parallel gradient synchronisation used by bench.py --gpus N."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepphysinet_amd.distributed import GradientAllReduce, broadcast_parameters, shard_range


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    broadcast_parameters(lin, src=0)
    w0 = torch.cat([p.detach().reshape(-1) for p in lin.parameters()])
    # every rank: its own shard of a common batch
    torch.manual_seed(7)
    xs, ys = torch.randn(10, 7), torch.randn(10, 3)
    lo, hi = shard_range(10, rank, world)
    loss = ((lin(xs[lo:hi]) - ys[lo:hi]) ** 2).mean()
    loss.backward()
    GradientAllReduce(bucket_mb=0.0001)(list(lin.parameters()))     # tiny buckets: exercises the multi-bucket path
    g = torch.cat([p.grad.reshape(-1) for p in lin.parameters()])
    q.put((rank, w0.tolist(), g.tolist()))      # by value: a tensor would travel as a shared-memory handle owned by this (exiting) process
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_equals_union_batch():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w_a, g_a), (_, w_b, g_b) = [(r, torch.tensor(w), torch.tensor(g)) for r, w, g in out]
    assert torch.equal(w_a, w_b)                    # broadcast made the replicas identical
    assert torch.allclose(g_a, g_b, rtol=0, atol=0)
    # reference: single process, whole batch (equal shard sizes -> mean of shard means == batch mean)
    torch.manual_seed(100)
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    torch.manual_seed(7)
    xs, ys = torch.randn(10, 7), torch.randn(10, 3)
    ((lin(xs) - ys) ** 2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in lin.parameters()])
    assert torch.allclose(g_a, ref, rtol=1e-5, atol=1e-7)


def test_shard_range_partitions():
    for n in (1, 7, 61, 37265):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
