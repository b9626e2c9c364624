"""y = x W^T + b for the per-field tensors (encoder, hyper-network heads) on the library's own small fp32 GEMM.

At these shapes (M <= 288 tokens, K = N = 256) library GEMMs are latency-bound (19-75 us each on MI355X, rocprofv3
profiles/round1); dpn_sgemm is a few microseconds.  CPU tensors take torch's F.linear (tests of the encoder math only).
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib as L


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _sgemm(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, asum=None, accumulate=0):
    lib = L.load()
    ws, ws_bytes = None, 0
    if K >= 1024 and ((M + 31) // 32) * ((N + 31) // 32) < 256:        # long reduction, small output: deterministic split-K scratch
        ws_bytes = 32 * (M * N + M) * 4
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=C.device)
    L.check(lib.dpn_sgemm(ta, tb, M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc, _p(bias), _p(asum), accumulate, _p(ws), ws_bytes,
                          torch.cuda.current_stream().cuda_stream), 'dpn_sgemm')


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        M, K = x2.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        _sgemm(0, 1, M, N, K, x2, K, w, K, y, N, bias=bias)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y.reshape(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        g = gy.reshape(M, N)
        g = g if g.is_contiguous() else g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((M, K), dtype=torch.float32, device=g.device)
            _sgemm(0, 0, M, K, N, g, N, w, K, gx, K)                   # gx = g W
            gx = gx.reshape(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            gw = torch.empty((N, K), dtype=torch.float32, device=g.device)
            gb = torch.empty((N,), dtype=torch.float32, device=g.device) if ctx.has_bias else None
            _sgemm(1, 0, N, K, M, g, N, x2, K, gw, K, asum=gb)          # gw = g^T x ; gb = sum_m g
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


def linear(x, weight, bias=None):
    """Drop-in for F.linear on fp32 HIP tensors; falls back to torch only for CPU tensors."""
    if x.is_cuda and x.dtype == torch.float32:
        return _LinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)
