"""Fused clip_grad_norm_ + Adam on the HIP library (reference: interface_physics.py:514-515, cfg:151-155).

Same arithmetic as `torch.nn.utils.clip_grad_norm_(params, max_norm)` followed by `torch.optim.Adam(lr, betas, eps,
weight_decay)` (weight decay added to the gradient), but three kernel launches for the whole model instead of ~60, and
hipGraph-capturable (the step counter lives on the device).
"""
import ctypes

import torch

from . import _lib as L


class FusedClipAdam:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=2.5e7):
        self.params = [p for p in params]
        if not self.params or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in self.params):
            raise RuntimeError('FusedClipAdam needs contiguous fp32 HIP parameters (no CPU fallback)')
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, initial_lr=lr)]
        self.max_norm = float(max_norm)
        dev = self.params[0].device
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        n = len(self.params)
        lib = L.load()
        self._numel = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
        # [0] = sum of squares of all gradients, then one fp64 partial per 2048-element chunk (fixed-order reduction, no atomics)
        self._sumsq = torch.zeros(int(lib.dpn_clip_adam_scratch_doubles(n, self._numel)), dtype=torch.float64, device=dev)
        # both moments as ONE flat buffer each (every tensor padded to whole 2048-element chunks): the kernels need no per-tensor state
        # pointers, and the whole model is one launch per pass.  exp_avg / exp_avg_sq are views into them.
        total = int(lib.dpn_clip_adam_flat_floats(n, self._numel))
        self._m_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._v_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg, self.exp_avg_sq, off = [], [], 0
        for p in self.params:
            k = p.numel()
            self.exp_avg.append(self._m_flat[off:off + k].view_as(p))
            self.exp_avg_sq.append(self._v_flat[off:off + k].view_as(p))
            off += ((k + 2047) // 2048) * 2048
        self._p = (ctypes.c_void_p * n)(*[p.data_ptr() for p in self.params])

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self):
        lib = L.load()
        n = len(self.params)
        grads = []
        for p in self.params:
            if p.grad is None:
                raise RuntimeError('FusedClipAdam.step(): a parameter has no gradient')
            grads.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
        g = (ctypes.c_void_p * n)(*[t.data_ptr() for t in grads])
        grp = self.param_groups[0]
        L.check(lib.dpn_clip_adam_flat(n, self._p, g, self._numel, ctypes.c_void_p(self._m_flat.data_ptr()), ctypes.c_void_p(self._v_flat.data_ptr()),
                                       ctypes.c_void_p(self._sumsq.data_ptr()), ctypes.c_void_p(self.step_count.data_ptr()), float(grp['lr']),
                                       float(grp['betas'][0]), float(grp['betas'][1]), float(grp['eps']), float(grp['weight_decay']), self.max_norm,
                                       ctypes.c_void_p(self.grad_norm.data_ptr()), torch.cuda.current_stream().cuda_stream), 'dpn_clip_adam_flat')
        return self.grad_norm
