"""Fused clip_grad_norm_ + Adam on the HIP library (reference: interface_physics.py:514-515, cfg:151-155).

Same arithmetic as `torch.nn.utils.clip_grad_norm_(params, max_norm)` followed by `torch.optim.Adam(lr, betas, eps,
weight_decay)` (weight decay added to the gradient), but three kernel launches for the whole model instead of ~60, and
hipGraph-capturable: the step counter AND the hyper-parameters (learning rate included) live on the device, so a captured step follows
a learning-rate schedule (`CosineAnnealingLR` stepped once per epoch, interface_physics.py:396-397, :831-833).

It is a `torch.optim.Optimizer`: `param_groups`, `state` (per parameter `step`, `exp_avg`, `exp_avg_sq` -- the keys of
torch.optim.Adam, so the state dicts load both ways: a torch.optim.Adam state dict gets this optimiser's own extras, max_norm, from its
defaults), `state_dict` / `load_state_dict`, and every torch LR scheduler attaches.

Gradients: the optimiser owns ONE flat fp32 gradient buffer laid out like its flat moment buffers (grad_arena.py).  The autograd
nodes of this package write parameter gradients straight into it, the data-parallel all-reduce (distributed.GradientAllReduce) runs in
place on its bucket slices, and the kernels read it -- gradients that arrive elsewhere (foreign autograd nodes) are copied in first.
`layout`: optional list of parameter lists = the buckets in the order their gradients become ready in the backward pass.
"""
import ctypes

import torch

from . import _lib as L
from . import grad_arena

_CHUNK = 2048            # dpn_clip_adam_flat: every tensor is padded to whole 2048-element chunks


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=2.5e7, layout=None):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, max_norm=float(max_norm))
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError('FusedClipAdam takes one parameter group (the reference trains with one, cfg:151-155)')
        given = list(self.param_groups[0]['params'])
        if not given or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in given):
            raise RuntimeError('FusedClipAdam needs contiguous fp32 HIP parameters (no CPU fallback)')
        if layout is None:
            layout = [given]
        flat_order = [p for b in layout for p in b]
        if len(flat_order) != len(given) or {id(p) for p in flat_order} != {id(p) for p in given}:
            raise ValueError('layout must partition exactly the optimised parameters')
        self.params = flat_order                                   # kernel / flat-buffer order (bucket after bucket)
        dev = self.params[0].device
        n = len(self.params)
        lib = L.load()
        self._numel = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
        # [0] = sum of squares of all gradients, then one fp64 partial per 2048-element chunk (fixed-order reduction, no atomics)
        self._sumsq = torch.zeros(int(lib.dpn_clip_adam_scratch_doubles(n, self._numel)), dtype=torch.float64, device=dev)
        total = int(lib.dpn_clip_adam_flat_floats(n, self._numel))
        # moments and gradients as ONE flat buffer each: no per-tensor state pointers in the kernel arguments, one launch per pass
        self._m_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._v_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._g_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._offsets, off = [], 0
        for p in self.params:
            self._offsets.append(off)
            off += ((p.numel() + _CHUNK - 1) // _CHUNK) * _CHUNK
        self.layout_ids = [[id(p) for p in b] for b in layout]       # which parameters each bucket of the flat buffers holds
        self.bucket_bounds, pos, k = [], 0, 0                       # [(start, end)] float offsets of each layout bucket in the flat buffers
        for b in layout:
            end = self._offsets[k + len(b)] if k + len(b) < n else total
            self.bucket_bounds.append((pos, end))
            pos, k = end, k + len(b)
        self.exp_avg = [self._m_flat[o:o + p.numel()].view_as(p) for o, p in zip(self._offsets, self.params)]
        self.exp_avg_sq = [self._v_flat[o:o + p.numel()].view_as(p) for o, p in zip(self._offsets, self.params)]
        self._slots = [self._g_flat[o:o + p.numel()].view_as(p) for o, p in zip(self._offsets, self.params)]
        for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq):       # torch.optim.Adam's state keys
            self.state[p] = {'step': self.step_count, 'exp_avg': m, 'exp_avg_sq': v}
        self._p = (ctypes.c_void_p * n)(*[p.data_ptr() for p in self.params])
        self._g = (ctypes.c_void_p * n)(*[s.data_ptr() for s in self._slots])
        # [lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale] on the device, read by the kernels at run time
        self._hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self._hyper_host = None
        self.grad_scale = 1.0
        self.steps_done = 0                                         # host-side count of step() calls (point_path's forward / backward stamps)
        self._leased = set()                                        # offsets of the slots a live gradient may alias (grad_arena)
        self._ptrs = [p.data_ptr() for p in self.params]            # what the kernels write through: checked against the live tensors every step
        self.sync_hyper()
        grad_arena.register(self, self.params, self._offsets)

    def __del__(self):
        try:
            grad_arena.unregister(self)
        except Exception:                                           # interpreter shutdown
            pass

    def _check_pointers(self):
        """The kernels update the parameters through the raw pointers taken at construction: a `model.to(...)` / `param.data = ...` after
        build_optimizer would leave them writing into freed memory."""
        for p, ptr in zip(self.params, self._ptrs):
            if p.data_ptr() != ptr:
                raise RuntimeError('FusedClipAdam: a parameter of shape %s was re-allocated after the optimiser was built (model.to() / '
                                   'param.data assignment); build the optimiser again' % (tuple(p.shape),))

    # ---- hyper-parameters ----------------------------------------------------------------------------------------------
    @property
    def max_norm(self):
        return self.param_groups[0]['max_norm']

    @max_norm.setter
    def max_norm(self, v):
        self.param_groups[0]['max_norm'] = float(v)

    def _hyper_values(self):
        g = self.param_groups[0]
        return (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), float(g['max_norm']),
                float(self.grad_scale), 0.0)

    def sync_hyper(self):
        """Upload param_groups[0] (lr after a scheduler step, ...) to the device scalars the kernels read.  step() does it by itself
        outside a graph capture; call it after `scheduler.step()` when the optimiser step is replayed from a hipGraph."""
        vals = self._hyper_values()
        if vals != self._hyper_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('FusedClipAdam: hyper-parameters changed during a graph capture; call sync_hyper() before capturing')
            self._hyper.copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=False)
            self._hyper_host = vals

    # ---- gradients -----------------------------------------------------------------------------------------------------
    def flat_gradients(self):
        """The flat gradient buffer (every tensor padded to whole 2048-element chunks, buckets back to back: `bucket_bounds`)."""
        return self._g_flat

    def gather_gradients(self, zero_missing=False):
        """Make `p.grad` of every parameter the view of its arena slot: gradients produced elsewhere are copied in (none on the fused
        path); a parameter without gradient raises, or -- zero_missing, for ranks whose graphs differ -- contributes zeros."""
        for p, s in zip(self.params, self._slots):
            g = p.grad
            if g is None:
                if not zero_missing:
                    raise RuntimeError('FusedClipAdam: a parameter has no gradient')
                s.zero_()
            elif g.data_ptr() != s.data_ptr() or not g.is_contiguous():
                s.copy_(g)
            else:
                continue
            p.grad = s

    def place_gradients(self, params, grads):
        """`p.grad = g` for gradients handed back by torch.autograd.grad (staged backward): a gradient that is not already the
        parameter's slot is copied into it (None: zeros), so that the bucket's all-reduce -- which may start right after -- sees it."""
        if getattr(self, '_slot_of', None) is None:
            self._slot_of = {id(p): s for p, s in zip(self.params, self._slots)}
        for p, g in zip(params, grads):
            s = self._slot_of[id(p)]
            if g is None:
                s.zero_()
            elif g.data_ptr() != s.data_ptr() or not g.is_contiguous():
                s.copy_(g)
            p.grad = s

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
        if set_to_none:
            self._leased.clear()                                    # no param.grad aliases a slot any more: the next backward writes in place

    # ---- step ----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._check_pointers()
        self.gather_gradients()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        else:
            grad_arena.captured_step[0] = True      # replays of this step rewrite the parameters without passing through here
        lib = L.load()
        L.check(lib.dpn_clip_adam_flat_dev(len(self.params), self._p, self._g, self._numel, ctypes.c_void_p(self._m_flat.data_ptr()),
                                           ctypes.c_void_p(self._v_flat.data_ptr()), ctypes.c_void_p(self._sumsq.data_ptr()),
                                           ctypes.c_void_p(self.step_count.data_ptr()), ctypes.c_void_p(self._hyper.data_ptr()),
                                           ctypes.c_void_p(self.grad_norm.data_ptr()), torch.cuda.current_stream().cuda_stream),
                'dpn_clip_adam_flat_dev')
        grad_arena.param_epoch[0] += 1
        self.steps_done += 1
        return self.grad_norm if closure is None else loss

    # ---- checkpoints ---------------------------------------------------------------------------------------------------
    def state_dict(self):
        sd = super().state_dict()
        step = self.step_count.detach().to(torch.float32).reshape(()).cpu()
        # copies, detached from the flat buffers (the packed state shares its per-parameter dicts with self.state); `step` as
        # torch.optim.Adam stores it
        sd['state'] = {k: {'step': step.clone(), 'exp_avg': st['exp_avg'].detach().clone(), 'exp_avg_sq': st['exp_avg_sq'].detach().clone()}
                       for k, st in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        views = {id(p): (m, v) for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq)}
        super().load_state_dict(state_dict)
        for k, v in self.defaults.items():          # a torch.optim.Adam state dict has no 'max_norm' (load_state_dict replaces the group wholesale)
            self.param_groups[0].setdefault(k, v)
        step = None
        for p in self.param_groups[0]['params']:
            st = self.state.get(p, {})
            m, v = views[id(p)]
            if 'exp_avg' in st:
                m.copy_(st['exp_avg'])
                v.copy_(st['exp_avg_sq'])
                step = st.get('step', step)
            self.state[p] = {'step': self.step_count, 'exp_avg': m, 'exp_avg_sq': v}
        if step is not None:
            self.step_count.fill_(int(float(step)))
        self._hyper_host = None
        self.sync_hyper()
