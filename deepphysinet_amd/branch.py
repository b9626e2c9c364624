"""Side branches of the step's dependency chain.

The training step (reference: interface/interface_physics.py:501-515) is one long chain of dependent launches, and a good part of it --
the encoder's backward, the hyper-network heads -- runs on 18-144 of the 256 CUs.  Kernels that nothing further down the chain waits for
(gradients of static tensors, per-layer weight gradients of the encoder) are issued on a SIDE stream: ordered behind the launch that produced
their inputs by an event, joined into the main stream before anybody may read their results.  Inside a hipGraph capture the same calls become
a fork and a join of the graph, so the replayed step keeps the branches.

    with branch.side(keep=(tensors the side kernels read through raw pointers,)):
        ... launches: torch.cuda.current_stream() IS the side stream here ...

Joins: `join()` makes the current stream wait for the side stream and releases the kept tensors.  A fork taken inside an autograd backward
pass joins by itself when the engine has finished that pass (`queue_callback`, the hook DistributedDataParallel uses for its own
finalisation): the caller of `loss.backward()` / `torch.autograd.grad(...)` sees every gradient complete on its stream, as without branches.
Forks outside a backward pass must call `join()` themselves before the results are used.

Which branches are taken is a frozen switch (`config.FROZEN.branches`, DPN_BRANCHES=finish,wgrad16; default: NONE).  Measured (round 5, same box,
profiles/round5_branches_ab.txt): a fork / join pair inside a hipGraph costs 10-16 us of idle chain on this runtime and the kernels that then run side
by side slow each other -- the captured step takes 1.211 ms without branches, 1.253 ms with the finish stage's static half (29 us of kernels) on the
side branch, 1.289 ms with the encoder's weight gradients there as well; five forks (one per layer) 1.407 against 1.301 ms.  The mechanism stays
(correct, tested) for runtimes where a fork is cheap; the product path does not take it.
"""
import torch

from . import config

_tab = {}       # device index -> _Dev.  One Python thread per rank drives the GPU; autograd's worker thread and the caller share this table


def enabled(which=None):
    """Is the side branch `which` ('finish', 'wgrad16') taken?  (no argument: is any)"""
    b = config.FROZEN.branches
    return bool(b) if which is None else which in b


class _Dev:
    __slots__ = ('stream', 'keep', 'open')

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.keep = []
        self.open = False                 # side work queued since the last join


def _dev(device=None):
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    d = _tab.get(idx)
    if d is None:
        d = _tab[idx] = _Dev(torch.device('cuda', idx))
    return d


def join(device=None):
    """The current stream waits for everything queued on the side stream; the tensors kept alive for it are released."""
    d = _dev(device)
    if d.open:
        torch.cuda.current_stream().wait_stream(d.stream)
        d.open = False
    d.keep.clear()


class side:
    """Context: launches inside go to the side stream, ordered behind everything queued so far on the current stream.
    keep: tensors the side kernels read or write through raw pointers (kept alive until the join: the caching allocator -- and a graph's
    private pool -- would otherwise hand their memory to a later allocation of the main stream while the branch is still running).
    in_backward: queue the join as a final callback of the running autograd pass (default); False: the caller joins."""

    def __init__(self, keep=(), in_backward=True, device=None):
        self.keep, self.in_backward, self.device = keep, in_backward, device
        self._ctx = None

    def __enter__(self):
        if not enabled():
            return self
        d = _dev(self.device)
        d.stream.wait_stream(torch.cuda.current_stream())
        d.keep.extend(self.keep)
        d.open = True
        if self.in_backward:              # (one callback per fork: join is idempotent, and a pass that died half-way leaves no stale flag behind)
            torch.autograd.Variable._execution_engine.queue_callback(join)       # raises outside a backward pass: use in_backward=False there
        self._ctx = torch.cuda.stream(d.stream)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
            self._ctx = None
        return False
