"""Data-parallel gradient synchronisation: one process per GPU, collocation batches (field samples) sharded across
ranks, the 22.4 MB gradient set averaged over RCCL/xGMI (backend "nccl" on ROCm) once per step.

Replaces the DistributedDataParallel wrap of the reference (interface/interface_physics.py:901-907, fired inside backward at :1056);
the only collective on the path is this gradient all-reduce.

`GradientAllReduce(optimizer)` works IN PLACE on the optimiser's flat gradient buffer (optim.FusedClipAdam / grad_arena.py): the
buffer is cut into the optimiser's layout buckets (gradients in the order the backward pass finishes them), each bucket is one
asynchronous all-reduce on a contiguous slice -- no flatten, no copy back -- and `reduce_bucket(i)` lets the caller start bucket i as
soon as its segment of the backward pass has been queued, so the transfer runs under the rest of the backward (xGMI rings are per-link
bound: few large messages, each hidden behind compute).  Without an optimiser it falls back to flattening `p.grad` of the given
parameters (any device / backend; the CPU tests use it with gloo).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank).
    force: create the process group even for one rank (a single-GPU box can then run the real RCCL calls of the N > 1 path)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # this pool's driver only supports dmabuf IPC (RCCL needs it)
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        try:
            if backend == 'nccl':
                torch.cuda.set_device(local)
            import datetime
            # a collective that never completes (a rank died, a link is down) must END the job with the rank named, not hang it: the process
            # group's own timeout (DPN_PG_TIMEOUT_S, default 600 s) backs the host-side watchdog below
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=float(os.environ.get('DPN_PG_TIMEOUT_S', '600'))))
            if backend == 'nccl':
                # the first collective is where a wrong IPC mode shows (hipIpcGetMemHandle: invalid argument): fail HERE, with the setting named
                probe = torch.ones(1, device=torch.device('cuda', local))
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError('the probe all-reduce returned %r instead of %d' % (probe.item(), world))
        except Exception as e:                                       # noqa
            raise RuntimeError(
                'deepphysinet_amd.distributed: the %s process group of %d rank(s) did not come up (%s: %s).  Multi-process GPU work on this '
                'platform needs dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY is %r in this process (this module sets 0 when it is unset); if the '
                "node's driver wants the legacy mode, export HSA_ENABLE_IPC_MODE_LEGACY=1 before starting the ranks.  MASTER_ADDR=%s "
                'MASTER_PORT=%s RANK=%d LOCAL_RANK=%d' % (backend, world, type(e).__name__, str(e)[:300], os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
                                                          os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'), rank, local)) from e
    return rank, world, local


def ipc_mode():
    """The IPC setting the ranks run with (recorded in bench.py's `collective` object)."""
    return os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')


def _all_reduce_mean(flat, world, group, async_op):
    """Average `flat` over the group in place.  RCCL: one AVG all-reduce on the device.  gloo (CPU tests; several ranks on ONE GPU in
    the single-GPU test box): SUM on a host copy, then the division."""
    backend = dist.get_backend(group)
    if backend == 'nccl':
        return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
    if flat.is_cuda:
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host.div_(world))
        return None
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=False)
    flat.div_(world)
    return None


def quiesce_for_capture(seconds=0.5):
    """Call before a stream capture that will CONTAIN collectives of an RCCL ('nccl') group.  ProcessGroupNCCL's watchdog thread polls the end events of
    the collectives it has not reaped yet (every ~100 ms); those events live on the group's RCCL stream, and once that stream is part of a capture HIP
    refuses to query them ("operation not permitted on an event last recorded in a capturing stream") -- the watchdog thread then terminates the whole
    process (exit code 134).  Reproduced 3 / 3 with `tools/pg_capture_probe.py 0`, 0 / 3 with a 0.5-s pause (profiles/round5_pg_watchdog_capture_race.txt).
    So: finish the eager collectives, then give the watchdog a few polls to drop them from its list.  (Collectives issued INSIDE the capture are never put
    on that list.)"""
    import time
    if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'):
        return
    torch.cuda.synchronize()
    time.sleep(float(seconds))


class Watchdog:
    """Host-side progress watchdog of a multi-rank loop: `beat(what)` after every unit of progress; when no beat arrives for `timeout_s`
    the thread prints which rank is stuck, in what (the label of the last beat and the last bucket a GradientAllReduce queued), and ends
    the PROCESS with exit code 13 -- the launcher (torch.distributed.run) then tears the other ranks down.  It never re-executes anything
    (a process that has touched the GPU must not exec on this platform); it only exits."""

    def __init__(self, timeout_s=300.0, rank=0, sync=None, out=None, on_stall=None):
        import sys
        import threading
        import time
        self.timeout_s, self.rank, self.sync = float(timeout_s), rank, sync
        self.out = out or sys.stderr
        self.on_stall = on_stall                   # callable(message): replaces the exit with code 13 (it is expected to end the process itself)
        self._last, self._what, self._stop = time.monotonic(), 'start', threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def beat(self, what=''):
        import time
        self._last, self._what = time.monotonic(), what

    def stop(self):
        self._stop.set()

    def _run(self):
        import time
        while not self._stop.wait(min(1.0, self.timeout_s / 4)):
            idle = time.monotonic() - self._last
            if idle > self.timeout_s:
                bucket = getattr(self.sync, 'last_queued', None)
                msg = ('[deepphysinet_amd.distributed] rank %d made no progress for %.0f s (last: %s; last all-reduce queued: %s) -- a collective '
                       'is not completing' % (self.rank, idle, self._what, 'buckets %s of the flat gradient buffer' % (bucket,) if bucket else 'none'))
                if self.on_stall is not None:
                    print(msg, file=self.out, flush=True)
                    self.on_stall(msg)
                    return
                print(msg + '; exiting with code 13', file=self.out, flush=True)
                os._exit(13)


class GradientAllReduce:
    def __init__(self, optimizer=None, bucket_mb=32.0, group=None, single_rank_too=False):
        self.last_queued = None
        self.opt = optimizer
        self.bucket_bytes = int(bucket_mb * 1024 * 1024)
        self.group = group
        self._work = []
        self._checked = False
        self.single_rank_too = single_rank_too        # run the collectives even in a one-rank group (exercises RCCL on a single-GPU box)

    def active(self):
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.single_rank_too)

    # ---- flat path (the optimiser owns the gradient buffer) ------------------------------------------------------------
    def _check_layout(self):
        """Every rank must reduce identically sized buckets, or the collective hangs: compare the layouts once."""
        if self._checked:
            return
        bounds = [b for bb in self.opt.bucket_bounds for b in bb]
        # a fixed-size fingerprint (an all_gather of differently sized tensors is itself a mismatched collective)
        mine = torch.tensor([len(bounds), bounds[-1], sum((i + 1) * b for i, b in enumerate(bounds)) % (1 << 62)], dtype=torch.int64)
        dev = self.opt.flat_gradients().device if dist.get_backend(self.group) == 'nccl' else 'cpu'
        mine = mine.to(dev)
        every = [torch.empty_like(mine) for _ in range(dist.get_world_size(self.group))]
        dist.all_gather(every, mine, group=self.group)
        if not all(torch.equal(e, mine) for e in every):
            raise RuntimeError('GradientAllReduce: the ranks hold differently laid out gradient buffers')
        self._checked = True

    def n_buckets(self):
        return len(self.opt.bucket_bounds)

    def reduce_bucket(self, i, end=None, async_op=True):
        """Queue the averaging all-reduce of layout bucket i -- or of the buckets i .. end-1 as ONE collective on their contiguous slice of
        the flat buffer (their gradients must already be queued on the current stream)."""
        if not self.active():
            return
        self._check_layout()
        a, b = self.opt.bucket_bounds[i][0], self.opt.bucket_bounds[(i + 1 if end is None else end) - 1][1]
        self.last_queued = (i, i + 1 if end is None else end)
        w = _all_reduce_mean(self.opt.flat_gradients()[a:b], dist.get_world_size(self.group), self.group, async_op)
        if w is not None and async_op:
            self._work.append(w)

    def wait(self, events=None):
        """The current stream waits for every queued bucket.  events (a list): a HIP event pair is recorded around each bucket's wait and
        appended to it -- the time between the two on the waiting stream is the part of that all-reduce the backward pass did NOT hide
        (bench.py: `collective.exposed_us`)."""
        for w in self._work:
            if events is not None and torch.cuda.is_available():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                w.wait()
                e1.record()
                events.append((e0, e1))
            else:
                w.wait()
        self._work = []

    # ---- entry point ---------------------------------------------------------------------------------------------------
    def __call__(self, params=None):
        """Average the gradients over the ranks.  With an optimiser: gather stray gradients into the flat buffer (a rank whose backward
        produced no gradient for a parameter contributes zeros), reduce every bucket, wait."""
        if not self.active():
            return
        if self.opt is not None:
            self.opt.gather_gradients(zero_missing=True)
            self.reduce_bucket(0, self.n_buckets())          # nothing left to overlap with: the whole flat buffer as one collective
            self.wait()
            return
        world = dist.get_world_size(self.group)
        params = list(params)
        # generic path: every rank flattens the same parameter list (None gradients count as zeros, so the sizes agree)
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        buckets, cur, size = [], [], 0
        for g in grads:
            nb = g.numel() * g.element_size()
            if cur and size + nb > self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(g)
            size += nb
        if cur:
            buckets.append(cur)
        for b in buckets:
            flat = torch.cat([g.reshape(-1) for g in b])
            _all_reduce_mean(flat, world, self.group, False)
            off = 0
            for g in b:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        for p, g in zip(params, grads):
            if p.grad is None:
                p.grad = g


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (DDP does this at wrap time)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    stage = dist.get_backend(group) != 'nccl'
    for t in list(module.parameters()) + list(module.buffers()):
        if stage and t.is_cuda:
            host = t.data.cpu()
            dist.broadcast(host, src=src, group=group)
            t.data.copy_(host)
        else:
            dist.broadcast(t.data, src=src, group=group)


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items for `rank` (DistributedSampler-like, no padding)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
