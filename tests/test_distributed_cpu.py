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


class _FlatOwner:
    """CPU stand-in for optim.FusedClipAdam's gradient side: one flat buffer, bucket bounds, gather (the real one needs a GPU)."""

    def __init__(self, params, bucket_sizes):
        self.params = params
        self._g_flat = torch.zeros(sum(p.numel() for p in params))
        self._slots, off = [], 0
        for p in params:
            self._slots.append(self._g_flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.bucket_bounds, a, k = [], 0, 0
        for nb in bucket_sizes:
            b = a + sum(p.numel() for p in params[k:k + nb])
            self.bucket_bounds.append((a, b))
            a, k = b, k + nb

    def flat_gradients(self):
        return self._g_flat

    def gather_gradients(self, zero_missing=False):
        for p, s in zip(self.params, self._slots):
            if p.grad is None:
                assert zero_missing
                s.zero_()
            elif p.grad.data_ptr() != s.data_ptr():
                s.copy_(p.grad)
            p.grad = s


def _worker_flat(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(5)
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    params = list(lin.parameters())
    owner = _FlatOwner(params, [2, 3, 1])
    sync = GradientAllReduce(owner)
    torch.manual_seed(50 + rank)
    x = torch.randn(6, 7)
    # rank 1 never uses the last layer: its gradients are None there -- the flat buckets still have the same size on every rank
    out = lin(x) if rank == 0 else lin[1](lin[0](x))
    out.pow(2).mean().backward()
    mine = [None if p.grad is None else p.grad.clone() for p in params]
    sync()                                                       # gather (zeros for the missing ones) + three bucket all-reduces + wait
    assert all(p.grad.data_ptr() == s.data_ptr() for p, s in zip(params, owner._slots))
    q.put((rank, [None if g is None else g.tolist() for g in mine], [p.grad.tolist() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_with_uneven_gradient_sets():
    """The in-place bucketed reducer on a flat gradient buffer: ranks whose sets of p.grad None / non-None differ must not hang
     and get the mean with zeros for the missing gradients, like DistributedDataParallel(find_unused_parameters)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flat, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, own0, red0), (_, own1, red1) = out
    assert own1[4] is None and own1[5] is None and own0[4] is not None
    for i in range(6):
        a = torch.tensor(own0[i])
        b = torch.zeros_like(a) if own1[i] is None else torch.tensor(own1[i])
        want = (a + b) / 2
        assert torch.equal(torch.tensor(red0[i]), torch.tensor(red1[i]))
        assert torch.allclose(torch.tensor(red0[i]), want, rtol=1e-6, atol=1e-8)


def _worker_mismatch(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    params = [torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(6))]
    owner = _FlatOwner(params, [1, 1] if rank == 0 else [2])          # differently cut buffers
    for p in params:
        p.grad = torch.ones_like(p)
    try:
        GradientAllReduce(owner)()
        q.put((rank, 'no error'))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.destroy_process_group()


def _worker_ranges(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(5)
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    params = list(lin.parameters())
    owner = _FlatOwner(params, [2, 3, 1])
    sync = GradientAllReduce(owner)
    torch.manual_seed(60 + rank)
    lin(torch.randn(6, 7)).pow(2).mean().backward()
    owner.gather_gradients()
    mine = owner.flat_gradients().clone()
    sync.reduce_bucket(0, 2)                                     # the staged step's schedule: layout buckets 0 and 1 as ONE collective ...
    sync.wait()
    after_first = owner.flat_gradients().clone()
    sync.reduce_bucket(2)                                        # ... then the last one
    sync.wait()
    q.put((rank, mine.tolist(), after_first.tolist(), owner.flat_gradients().tolist(), owner.bucket_bounds))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_ranges_reduce_as_one_collective():
    """reduce_bucket(i, end): the contiguous slice of layout buckets i .. end-1 is averaged by one all-reduce (StagedPdeStep.stage_buckets:
    the statics' and the heads' buckets travel together), the buckets outside the range are untouched until their own call."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ranges, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, first0, fin0, bounds), (_, g1, first1, fin1, _) = out
    g0, g1, first0, fin0, fin1 = (torch.tensor(v) for v in (g0, g1, first0, fin0, fin1))
    mean = (g0 + g1) / 2
    cut = bounds[1][1]                                           # end of layout bucket 1
    assert torch.allclose(first0[:cut], mean[:cut], rtol=1e-6, atol=1e-8)
    assert torch.equal(first0[cut:], g0[cut:])                   # bucket 2 not reduced yet
    assert torch.allclose(fin0, mean, rtol=1e-6, atol=1e-8) and torch.equal(fin0, fin1)


def test_flat_bucket_allreduce_refuses_mismatched_layouts():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mismatch, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all('differently laid out' in msg for _, msg in out), out


def test_shard_range_partitions():
    for n in (1, 7, 61, 37265):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sample_sharding_pads_like_distributed_sampler():
    """run_train_interface_dist's per-rank samples (reference: DistributedSampler, interface_physics.py:936, drop_last=False): every rank
    gets ceil(n / world) samples -- the same count, so the same number of all-reduces (an uneven split deadlocks RCCL) -- rank r
    takes r, r + world, ..., the tail wraps around; identical for a sequence (indexed: a rank touches only its own samples) and for a
    generator (consumed round by round)."""
    from torch.utils.data.distributed import DistributedSampler
    from deepphysinet_amd.interface.interface_physics import InterfacePhysics
    for n in (1, 2, 3, 5, 7, 8):
        for world in (1, 2, 3, 4):
            if n < world:
                continue                                    # DistributedSampler pads differently below one sample per rank; not a training setup
            counts = set()
            for rank in range(world):
                want = list(DistributedSampler(list(range(n)), num_replicas=world, rank=rank, shuffle=False, drop_last=False))
                touched = []

                class Seq:                                  # a sequence that records which samples were materialised
                    def __len__(self): return n
                    def __getitem__(self, i):
                        if not 0 <= i < n:
                            raise IndexError(i)
                        touched.append(i)
                        return i
                got_seq = list(InterfacePhysics._shard_samples(Seq(), rank, world))
                got_gen = list(InterfacePhysics._shard_samples((i for i in range(n)), rank, world))
                assert got_seq == want and got_gen == want, (n, world, rank, got_seq, got_gen, want)
                assert touched == want                      # nothing drawn for the other ranks
                counts.add(len(got_seq))
                # the distributed loop's default: the reference's `DistributedSampler(train_dataset)` (:936) = shuffle=True, seed 0, and no set_epoch
                # call anywhere -> the same seed-0 permutation every epoch
                want_sh = list(DistributedSampler(list(range(n)), num_replicas=world, rank=rank, drop_last=False))
                touched.clear()
                assert list(InterfacePhysics._shard_samples(Seq(), rank, world, shuffle_seed=0)) == want_sh, (n, world, rank)
                assert touched == want_sh
            assert len(counts) == 1                         # every rank runs the same number of steps


def test_rank_aware_sample_callable_is_used_as_is():
    from deepphysinet_amd.interface.interface_physics import InterfacePhysics

    class Dummy(InterfacePhysics):
        def __init__(self):                                 # no model: only the sample plumbing is under test
            self.train_cfg = {}
    d = Dummy()
    calls = []
    own = lambda epoch, rank, world: calls.append((epoch, rank, world)) or ['r%d' % rank]
    # the per-rank protocol is selected explicitly: by keyword, or by an attribute of the callable -- never by its arity
    assert list(d._epoch_samples({'samples': own, 'samples_per_rank': True}, 3, 1, 4)) == ['r1'] and calls == [(3, 1, 4)]
    own.per_rank = True
    assert list(d._epoch_samples({'samples': own}, 5, 0, 2)) == ['r0'] and calls[-1] == (5, 0, 2)
    # sharding inside the loop: the distributed loop shuffles like DistributedSampler's defaults in every epoch, `shuffle=False` keeps the order,
    # the single-process loop never shuffles
    from torch.utils.data.distributed import DistributedSampler
    seq = list('abcdefg')
    for epoch in (0, 3):
        want = [seq[i] for i in DistributedSampler(seq, num_replicas=2, rank=1)]
        assert list(d._epoch_samples({'samples': seq}, epoch, 1, 2, True)) == want
    assert list(d._epoch_samples({'samples': seq, 'shuffle': False}, 0, 1, 2, True)) == ['b', 'd', 'f', 'a']
    assert list(d._epoch_samples({'samples': seq}, 0, 1, 2)) == ['b', 'd', 'f', 'a']
    every = lambda epoch: ['a', 'b', 'c']
    assert list(d._epoch_samples({'samples': every}, 0, 1, 2)) == ['b', 'b'] or list(d._epoch_samples({'samples': every}, 0, 1, 2)) == ['b', 'a']


def test_watchdog_ends_a_stuck_rank_with_its_name_and_bucket():
    """bench.py --gpus N / the data-parallel loop: a collective that never completes must end the job with the stuck rank and the bucket it last
    queued named (exit code 13), not hang it.  The watchdog only ever EXITS the process -- no exec."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys, time; sys.path.insert(0, %r)\n'
            'from deepphysinet_amd.distributed import Watchdog\n'
            'class S: last_queued = (2, 3)\n'
            'w = Watchdog(0.4, rank=5, sync=S())\n'
            'w.beat("a block of 20 steps")\n'
            'time.sleep(30)\n' % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 13, (r.returncode, r.stderr[-500:])
    assert 'rank 5' in r.stderr and 'a block of 20 steps' in r.stderr and '(2, 3)' in r.stderr
    # a loop that keeps beating is left alone, and stop() ends the thread
    code2 = ('import sys, time; sys.path.insert(0, %r)\n'
             'from deepphysinet_amd.distributed import Watchdog\n'
             'w = Watchdog(0.5, rank=0)\n'
             'for _ in range(12):\n'
             '    time.sleep(0.1); w.beat("x")\n'
             'w.stop(); time.sleep(1.0); print("alive")\n' % ROOT)
    r2 = subprocess.run([sys.executable, '-c', code2], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0 and 'alive' in r2.stdout, (r2.returncode, r2.stderr[-500:])
    # a stall with a handler (bench.py's trial of the one-graph form): the handler decides -- it prints what it holds and ends the process with bench.TRIAL_STALL_EXIT = 14:
    # NON-ZERO, a process that has touched the GPU and gave up on a collective must not look like a clean run
    code3 = ('import os, sys, time; sys.path.insert(0, %r)\n'
             'from deepphysinet_amd.distributed import Watchdog\n'
             'def handler(msg):\n'
             '    print("stashed line", flush=True); os._exit(14)\n'
             'w = Watchdog(0.4, rank=1, on_stall=handler)\n'
             'w.beat("first replays of the one-graph form")\n'
             'time.sleep(30)\n' % ROOT)
    r3 = subprocess.run([sys.executable, '-c', code3], capture_output=True, text=True, timeout=120)
    assert r3.returncode == 14 and 'stashed line' in r3.stdout and 'first replays of the one-graph form' in r3.stderr, (r3.returncode, r3.stderr[-500:])
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert 'TRIAL_STALL_EXIT = 14' in src and 'os._exit(0)' not in src          # bench.py's own handler follows the convention
