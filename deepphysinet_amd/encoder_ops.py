"""Autograd wrappers of the encoder kernels in libdpn_hip.so (csrc/dpn_encoder.hip): attention and add + LayerNorm.
CPU tensors take the equivalent torch expressions (encoder-math tests only)."""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib as L


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _s():
    return torch.cuda.current_stream().cuda_stream


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _AttentionFn(torch.autograd.Function):
    """o = softmax(q k^T / sqrt(32)) v for 8 heads x 32 (model/attn.py:50-68); q, k, v: [L, 256]."""

    @staticmethod
    def forward(ctx, q, k, v):
        lib = L.load()
        q, k, v = _c(q), _c(k), _c(v)
        n = q.shape[0]
        o = torch.empty_like(q)
        P = torch.empty((8, 288, 288), dtype=torch.float32, device=q.device)
        L.check(lib.dpn_attn_fwd(_p(q), _p(k), _p(v), n, _p(o), _p(P), _s()), 'dpn_attn_fwd')
        ctx.save_for_backward(q, k, v, o, P)
        return o

    @staticmethod
    def backward(ctx, go):
        lib = L.load()
        q, k, v, o, P = ctx.saved_tensors
        go = _c(go)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        dS = torch.empty_like(P)
        L.check(lib.dpn_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(P), _p(go), q.shape[0], _p(dq), _p(dk), _p(dv), _p(dS), _s()), 'dpn_attn_bwd')
        return dq, dk, dv


def attention(q, k, v):
    """q, k, v: [1, L, 8, 32] (the reference's FullAttention layout) -> [1, L, 8, 32]."""
    B, Lq, H, E = q.shape
    if q.is_cuda and B == 1 and H == 8 and E == 32 and Lq <= 288 and k.shape[1] == Lq:
        o = _AttentionFn.apply(q.reshape(Lq, H * E), k.reshape(Lq, H * E), v.reshape(Lq, H * E))
        return o.view(1, Lq, H, E)
    out = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    return out.transpose(1, 2).contiguous()


class _AddLayerNormFn(torch.autograd.Function):
    """LayerNorm_256(x + r) * gamma + beta; r may be None."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta):
        lib = L.load()
        shape = x.shape
        x2 = _c(x.reshape(-1, 256))
        r2 = None if r is None else _c(r.reshape(-1, 256))
        rows = x2.shape[0]
        out = torch.empty_like(x2)
        xhat = torch.empty_like(x2)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.check(lib.dpn_add_ln_fwd(_p(x2), _p(r2), _p(gamma), _p(beta), rows, _p(out), _p(xhat), _p(rstd), _s()), 'dpn_add_ln_fwd')
        ctx.save_for_backward(xhat, rstd, gamma)
        ctx.has_r, ctx.shape = r is not None, shape
        return out.view(shape)

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        xhat, rstd, gamma = ctx.saved_tensors
        g2 = _c(g.reshape(-1, 256))
        gx = torch.empty_like(g2)
        dgamma = torch.empty(256, dtype=torch.float32, device=g.device)
        dbeta = torch.empty(256, dtype=torch.float32, device=g.device)
        scratch = torch.empty(((g2.shape[0] + 3) // 4) * 512, dtype=torch.float32, device=g.device)
        L.check(lib.dpn_add_ln_bwd(_p(g2), _p(xhat), _p(rstd), _p(gamma), g2.shape[0], _p(gx), _p(dgamma), _p(dbeta), _p(scratch), _s()),
                'dpn_add_ln_bwd')
        gx = gx.view(ctx.shape)
        return gx, (gx if ctx.has_r else None), dgamma, dbeta


def add_layer_norm(x, r, norm: torch.nn.LayerNorm):
    """norm(x + r) with the module's parameters (r may be None)."""
    if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256 and norm.elementwise_affine and norm.eps == 1e-5:
        return _AddLayerNormFn.apply(x, r, norm.weight, norm.bias)
    return norm(x if r is None else x + r)
