"""y = x W^T + b for the per-field tensors (encoder, hyper-network heads) on the library's own exact-fp32 MFMA GEMM, and the
problem / launch helpers the encoder nodes (encoder_ops.py) build their launches from.

At these shapes (M = 287 tokens, K = N = 256) a GEMM is latency-bound whatever computes it (rocBLAS / hipBLASLt through torch:
19-75 us each on MI355X); dpn_sgemm_batch runs the independent GEMMs of one step of the layer schedule in a single launch (the
three q/k/v projections; the input- and weight-gradient GEMMs of a linear; ride-along LayerNorm parameter sums) in 8-25 us.
CPU tensors take torch's F.linear (tests of the encoder math only).
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib as L
from .grad_arena import new_grad


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('deepphysinet_amd: a %s tensor on %s was passed to a HIP kernel; there is no CPU fallback' % (tuple(t.shape), t.device))
    return t.data_ptr()


def _problem(M, N, K, terms, C, ldc, ta, tb, bias=None, asum=None, epi=0, aux=None, aux_out=None):
    q = L.DpnGemmProblem()
    q.epi, q.aux, q.aux_out = epi, _p(aux), _p(aux_out)
    for i, term in enumerate(terms):                 # (A, lda, B, ldb) or (A, lda, B, ldb, K_t) when the terms reduce over different lengths
        A, lda, B, ldb = term[:4]
        q.A[i], q.lda[i], q.B[i], q.ldb[i] = _p(A), lda, _p(B), ldb
        q.k_term[i] = term[4] if len(term) > 4 else 0
    q.bias, q.C, q.asum = _p(bias), _p(C), _p(asum)
    q.M, q.N, q.K, q.ldc, q.ta, q.tb, q.nterms = M, N, K, ldc, ta, tb, len(terms)
    return q


def _launch(problems, colsum_jobs=()):
    """One dpn_sgemm_batch launch; colsum_jobs: up to two (scratch, rows, dgamma, dbeta) LayerNorm parameter reductions riding along."""
    lib = L.load()
    arr = (L.DpnGemmProblem * len(problems))(*problems)
    if not colsum_jobs:
        L.check(lib.dpn_sgemm_batch(len(problems), arr, torch.cuda.current_stream().cuda_stream), 'dpn_sgemm_batch')
        return
    # (scratch, rows, dgamma, dbeta): partials of dpn_add_ln_bwd (one block per 4 rows); a fifth element gives the block count itself
    jobs = (L.DpnColsumJob * len(colsum_jobs))(*[L.DpnColsumJob(j[0].data_ptr(), j[2].data_ptr(), j[3].data_ptr(), j[4] if len(j) > 4 else (j[1] + 3) // 4)
                                                for j in colsum_jobs])
    L.check(lib.dpn_sgemm_batch_jobs(len(problems), arr, len(colsum_jobs), jobs, torch.cuda.current_stream().cuda_stream), 'dpn_sgemm_batch_jobs')


def _launch_ln(mode, M, N, x, r, gamma, beta, rstd_in, y_out, xhat_out, rstd_out, partial, B, tb, ldb, C, ldc, bias=None, epi=0, aux=None,
               aux_out=None):
    """dpn_sgemm_ln: LayerNorm (mode 1 forward of x + r, mode 2 backward of g = x with xhat = r) applied to the A tile of the GEMM that
    consumes it -- one launch instead of two (include/dpn_hip.h)."""
    q = L.DpnLnGemm()
    q.mode, q.M, q.N, q.tb, q.ldb, q.ldc, q.epi = mode, M, N, tb, ldb, ldc, epi
    q.x, q.r, q.gamma, q.beta, q.rstd_in = _p(x), _p(r), _p(gamma), _p(beta), _p(rstd_in)
    q.y_out, q.xhat_out, q.rstd_out, q.partial = _p(y_out), _p(xhat_out), _p(rstd_out), _p(partial)
    q.B, q.bias, q.C, q.aux, q.aux_out = _p(B), _p(bias), _p(C), _p(aux), _p(aux_out)
    L.check(L.load().dpn_sgemm_ln(ctypes.byref(q), torch.cuda.current_stream().cuda_stream), 'dpn_sgemm_ln')


def _sgemm_splitk(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, asum=None):
    """Single GEMM with the deterministic two-pass split-K (long reductions, small outputs: token embedding, head input-gradient)."""
    lib = L.load()
    ws_bytes = 32 * (M * N + M) * 4
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=C.device)
    L.check(lib.dpn_sgemm(ta, tb, M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc, _p(bias), _p(asum), 0, ws.data_ptr(),
                          ws_bytes, torch.cuda.current_stream().cuda_stream), 'dpn_sgemm')


def _long_k(M, N, K):
    return K >= 1024 and ((M + 31) // 32) * ((N + 31) // 32) < 256


class _MultiLinearFn(torch.autograd.Function):
    """(y_1..y_n) = (x W_1^T + b_1, ..., x W_n^T + b_n), n <= 3, one launch forward, one launch backward."""

    @staticmethod
    def forward(ctx, x, *wb):
        n = len(wb) // 2
        ws = [w if w.is_contiguous() else w.contiguous() for w in wb[:n]]
        bs = list(wb[n:])
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        M, K = x2.shape
        ys = [torch.empty((M, w.shape[0]), dtype=torch.float32, device=x.device) for w in ws]
        if n == 1 and _long_k(M, ws[0].shape[0], K):
            _sgemm_splitk(0, 1, M, ws[0].shape[0], K, x2, K, ws[0], K, ys[0], ws[0].shape[0], bias=bs[0])
        else:
            _launch([_problem(M, w.shape[0], K, [(x2, K, w, K)], y, w.shape[0], 0, 1, bias=b) for w, b, y in zip(ws, bs, ys)])
        ctx.save_for_backward(x2, *ws)
        ctx.n, ctx.has_bias, ctx.x_shape = n, [b is not None for b in bs], x.shape
        ctx.biases = bs                                  # identify the bias parameters' gradient slots (grad_arena)
        return tuple(y.reshape(x.shape[:-1] + (y.shape[1],)) for y in ys)

    @staticmethod
    def backward(ctx, *gys):
        x2, *ws = ctx.saved_tensors
        n = ctx.n
        M, K = x2.shape
        gs = []
        for gy, w in zip(gys, ws):
            g = gy.reshape(M, w.shape[0])
            gs.append(g if g.is_contiguous() else g.contiguous())
        dev = x2.device
        problems = []
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((M, K), dtype=torch.float32, device=dev)
            if n == 1 and _long_k(M, K, ws[0].shape[0]):
                _sgemm_splitk(0, 0, M, K, ws[0].shape[0], gs[0], ws[0].shape[0], ws[0], K, gx, K)
            else:
                N0 = ws[0].shape[0]
                assert all(w.shape[0] == N0 for w in ws)
                problems.append(_problem(M, K, N0, [(g, N0, w, K) for g, w in zip(gs, ws)], gx, K, 0, 0))      # gx = sum_i g_i W_i
        from .encoder_ops import _wgrad
        gws, gbs = [], []
        for i, (g, w) in enumerate(zip(gs, ws)):
            N = w.shape[0]
            gw = new_grad(w, (N, K))
            gb = new_grad(ctx.biases[i], (N,)) if ctx.has_bias[i] else None
            _wgrad(problems, N, K, M, g, N, x2, K, gw, gb)             # gw = g^T x ; gb = sum_m g (split-K kernel when M is a batch of fields)
            gws.append(gw)
            gbs.append(gb)
        if problems:
            _launch(problems)
        return (gx.reshape(ctx.x_shape) if gx is not None else None, *gws, *gbs)


def linear(x, weight, bias=None):
    """Drop-in for F.linear on fp32 HIP tensors (host tensors: only in reference-math mode, _lib.host_math_or_raise)."""
    if x.is_cuda and x.dtype == torch.float32:
        return _MultiLinearFn.apply(x, weight, bias)[0]
    L.host_math_or_raise(x, 'linear')
    return F.linear(x, weight, bias)


def linear_multi(x, weights, biases):
    """[F.linear(x, w, b) for w, b in ...] as one launch (same output width for all; at most three)."""
    if x.is_cuda and x.dtype == torch.float32 and len(weights) <= 3:
        return _MultiLinearFn.apply(x, *weights, *biases)
    L.host_math_or_raise(x, 'linear_multi')
    return tuple(F.linear(x, w, b) for w, b in zip(weights, biases))
