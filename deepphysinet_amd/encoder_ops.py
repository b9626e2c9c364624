"""Autograd nodes of the grid encoder on libdpn_hip.so (csrc/dpn_encoder.hip, dpn_sgemm_batch): the data embedding, a whole
EncoderLayer and the hyper-network heads as single nodes with hand-scheduled backward passes (a dependent kernel costs >= 4.5 us on
this machine whatever it does, so the schedule minimises the number of launches on the dependency chain), for one field sample or
a batch of them; plus per-op wrappers (attention, add + LayerNorm) for modules that do not fit the fused nodes.
Host tensors raise unless reference math is switched on (_lib.enable_cpu_reference_math: CPU-side encoder-math tests only)."""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib as L
from . import branch, config
from .grad_arena import new_grad


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('deepphysinet_amd: a %s tensor on %s was passed to a HIP kernel; there is no CPU fallback' % (tuple(t.shape), t.device))
    return ctypes.c_void_p(t.data_ptr())


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
        L.check(lib.dpn_attn_fwd(_p(q), _p(k), _p(v), n, 1, _p(o), _p(P), _s()), 'dpn_attn_fwd')
        ctx.save_for_backward(q, k, v, o, P)
        return o

    @staticmethod
    def backward(ctx, go):
        lib = L.load()
        q, k, v, o, P = ctx.saved_tensors
        go = _c(go)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        L.check(lib.dpn_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(P), _p(go), q.shape[0], 1, _p(dq), _p(dk), _p(dv), _s()), 'dpn_attn_bwd')
        return dq, dk, dv


def attention(q, k, v):
    """q, k, v: [1, L, 8, 32] (the reference's FullAttention layout) -> [1, L, 8, 32]."""
    B, Lq, H, E = q.shape
    if q.is_cuda and B == 1 and H == 8 and E == 32 and Lq <= 288 and k.shape[1] == Lq:
        o = _AttentionFn.apply(q.reshape(Lq, H * E), k.reshape(Lq, H * E), v.reshape(Lq, H * E))
        return o.view(1, Lq, H, E)
    L.host_math_or_raise(q, 'attention')
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
        ctx.has_r, ctx.shape, ctx.beta = r is not None, shape, beta
        return out.view(shape)

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        xhat, rstd, gamma = ctx.saved_tensors
        g2 = _c(g.reshape(-1, 256))
        gx = torch.empty_like(g2)
        dgamma, dbeta = new_grad(gamma), new_grad(ctx.beta)
        scratch = torch.empty(((g2.shape[0] + 3) // 4) * 512, dtype=torch.float32, device=g.device)
        L.check(lib.dpn_add_ln_bwd(_p(g2), _p(xhat), _p(rstd), _p(gamma), g2.shape[0], _p(gx), _p(dgamma), _p(dbeta), _p(scratch), _s()),
                'dpn_add_ln_bwd')
        gx = gx.view(ctx.shape)
        return gx, (gx if ctx.has_r else None), dgamma, dbeta


def add_layer_norm(x, r, norm: torch.nn.LayerNorm):
    """norm(x + r) with the module's parameters (r may be None)."""
    if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256 and norm.elementwise_affine and norm.eps == 1e-5:
        return _AddLayerNormFn.apply(x, r, norm.weight, norm.bias)
    L.host_math_or_raise(x, 'add_layer_norm')
    return norm(x if r is None else x + r)


def _wgrad(batch, N, K, n, g, ldg, x, ldx, gw, gb):
    """gw[N][K] = g^T x over the n rows (gb = column sums of g).  Short reductions (one field: n = 287) join the GEMM launch `batch`.
    Long ones (a batch of fields: n = B * 287) are cut into row slices that run side by side as problems of their own MFMA launch;
    the slice results are added in a fixed order by dpn_sum_parts."""
    from .linear import _launch, _problem
    if n < 1024:
        batch.append(_problem(N, K, n, [(g, ldg, x, ldx)], gw, K, 1, 0, asum=gb))
        return
    S = min(16, (n + 1023) // 1024)
    rows = (n + S - 1) // S
    stride = N * K + N                                           # per slice: the weight block, then the bias sums
    parts = torch.empty((S, stride), dtype=torch.float32, device=g.device)
    out = torch.empty(stride, dtype=torch.float32, device=g.device)
    problems = []
    for s_ in range(S):
        r0, r1 = s_ * rows, min(n, (s_ + 1) * rows)
        q = _problem(N, K, r1 - r0, [(g, ldg, x, ldx)], parts, K, 1, 0, asum=parts if gb is not None else None)
        base = parts.data_ptr() + s_ * stride * 4
        q.A[0], q.B[0], q.C = g.data_ptr() + r0 * ldg * 4, x.data_ptr() + r0 * ldx * 4, base
        if gb is not None:
            q.asum = base + N * K * 4
        problems.append(q)
    _launch(problems)
    L.check(L.load().dpn_sum_parts(_p(parts), S, stride if gb is not None else N * K, 0, _p(out), _s()), 'dpn_sum_parts')
    gw.copy_(out[:N * K].view(N, K))                             # the callers own gw / gb (they are returned as gradients)
    if gb is not None:
        gb.copy_(out[N * K:])


class _EncoderLayerFn(torch.autograd.Function):
    """One EncoderLayer (model/transformer_net.py:28-44 + attn.py:177-196) as a single autograd node with a hand-scheduled backward:
        x1 = LN1(x + out_proj(attention(q(x), k(x), v(x))));  out = LN2(x1 + conv2(gelu(conv1(x1))))
    7 launches forward, 5 backward.  GELU and its derivative ride in the GEMM epilogues, each residual-branch gradient joins the
    input gradient inside the GEMM that produces it (DPN_EPI_ADD), both LayerNorm backwards are applied by the consuming GEMM to its
    own A tile (dpn_sgemm_ln), and the LayerNorm parameter sums ride along in the next GEMM launch: no elementwise kernel sits on
    the dependency chain of the backward pass.
    x: [B * L, 256] -- B field samples of L tokens each; attention stays inside a field, everything else is row-wise.
    conv weights as [d_ff, 256] / [256, d_ff] matrices."""

    @staticmethod
    def forward(ctx, x, B, Lt, wq, bq, wk, bk, wv, bv, wo, bo, g1, be1, wc1, bc1, wc2, bc2, g2, be2):
        from .linear import _launch, _launch_ln, _problem
        lib = L.load()
        x = _c(x)
        wq, wk, wv, wo, wc1, wc2 = (_c(w) for w in (wq, wk, wv, wo, wc1, wc2))
        n, D = x.shape
        assert n == B * Lt
        Fh = wc1.shape[0]
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        # BASELINE configs[4] experiment, off in the product: the six forward GEMMs of the layer on the fp8 matrix cores (csrc/dpn_fp8.hip)
        # ('1': per-row scales, v_mfma_f32_32x32x16_fp8_fp8; 'mx': one E8M0 scale per 32 k, v_mfma_scale_f32_32x32x64_f8f6f4)
        fp8_mode = config.FROZEN.encoder_fp8
        fp8 = fp8_mode in ('1', 'mx')

        def gemm8(M, N, K, A, W, bias, C, epi=0, aux_out=None):
            fn = lib.dpn_gemm_fp8_mx if fp8_mode == 'mx' else L.load_experiments().dpn_gemm_fp8     # (the non-scaled form: shelved, experiment library)
            L.check(fn(M, N, K, _p(A), K, _p(W), K, _p(bias), _p(C), N, epi, _p(aux_out), _s()), 'dpn_gemm_fp8')
        q, k, v = new(n, D), new(n, D), new(n, D)
        if fp8:
            for w, b, y in ((wq, bq, q), (wk, bk, k), (wv, bv, v)):
                gemm8(n, D, D, x, w, b, y)
        else:
            _launch([_problem(n, D, D, [(x, D, w, D)], y, D, 0, 1, bias=b) for w, b, y in ((wq, bq, q), (wk, bk, k), (wv, bv, v))])
        o, P = new(n, D), new(B * 8, 288, 288)
        L.check(lib.dpn_attn_fwd(_p(q), _p(k), _p(v), Lt, B, _p(o), _p(P), _s()), 'dpn_attn_fwd')
        a = new(n, D)
        if fp8:
            gemm8(n, D, D, o, wo, bo, a)
        else:
            _launch([_problem(n, D, D, [(o, D, wo, D)], a, D, 0, 1, bias=bo)])
        # (LayerNorm forward folded into the conv1 GEMM -- dpn_sgemm_ln mode 1 -- measured slower than the two launches: 21 vs 13.6 us with four waves, equal with eight)
        x1, xhat1, rstd1 = new(n, D), new(n, D), new(n)
        L.check(lib.dpn_add_ln_fwd(_p(x), _p(a), _p(g1), _p(be1), n, _p(x1), _p(xhat1), _p(rstd1), _s()), 'dpn_add_ln_fwd')
        pre, act = new(n, Fh), new(n, Fh)
        y = new(n, D)
        if fp8:
            gemm8(n, Fh, D, x1, wc1, bc1, act, L.EPI_GELU, pre)
            gemm8(n, D, Fh, act, wc2, bc2, y)
        else:
            _launch([_problem(n, Fh, D, [(x1, D, wc1, D)], act, Fh, 0, 1, bias=bc1, epi=L.EPI_GELU, aux_out=pre)])
            _launch([_problem(n, D, Fh, [(act, Fh, wc2, Fh)], y, D, 0, 1, bias=bc2)])
        out, xhat2, rstd2 = new(n, D), new(n, D), new(n)
        L.check(lib.dpn_add_ln_fwd(_p(x1), _p(y), _p(g2), _p(be2), n, _p(out), _p(xhat2), _p(rstd2), _s()), 'dpn_add_ln_fwd')
        ctx.save_for_backward(x, q, k, v, o, P, x1, pre, act, xhat1, rstd1, xhat2, rstd2, wq, wk, wv, wo, wc1, wc2, g1, g2)
        ctx.B, ctx.Lt = B, Lt
        ctx.biases = (bq, bk, bv, bo, be1, bc1, bc2, be2)            # identify the gradient slots of the bias parameters (grad_arena)
        return out

    @staticmethod
    def backward(ctx, g):
        from .linear import _launch, _launch_ln, _problem
        lib = L.load()
        x, q, k, v, o, P, x1, pre, act, xhat1, rstd1, xhat2, rstd2, wq, wk, wv, wo, wc1, wc2, g1, g2 = ctx.saved_tensors
        n, D = x.shape
        Fh = wc1.shape[0]
        dev = x.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        nb = (n + 31) // 32                                       # row blocks of the LayerNorm-in-GEMM launches (their partial sums)
        scratch2, scratch1 = new(nb * 512), new(nb * 512)
        # LN2 backward is applied by the conv2 input-gradient GEMM to its own A tile: d(pre) = (gs2 W_c2) * gelu'(pre); gs2 kept
        bq, bk, bv, bo, be1, bc1, bc2, be2 = ctx.biases
        gs2, dg2, dbe2 = new(n, D), new_grad(g2), new_grad(be2)
        dpre = new(n, Fh)
        _launch_ln(2, n, Fh, _c(g), xhat2, g2, None, rstd2, gs2, None, None, scratch2, wc2, 0, Fh, dpre, Fh, epi=L.EPI_MUL_GELU_GRAD, aux=pre)
        # conv1: d(x1) = dpre W_c1 + gs2 (the residual branch); with it dW_c2 = gs2^T act, dW_c1 = dpre^T x1 and LN2's parameter sums
        dwc2, dbc2 = new_grad(wc2, (D, Fh)), new_grad(bc2)
        dx1, dwc1, dbc1 = new(n, D), new_grad(wc1, (Fh, D)), new_grad(bc1)
        batch = [_problem(n, D, Fh, [(dpre, Fh, wc1, D)], dx1, D, 0, 0, epi=L.EPI_ADD, aux=gs2)]
        _wgrad(batch, D, Fh, n, gs2, D, act, Fh, dwc2, dbc2)
        _wgrad(batch, Fh, D, n, dpre, Fh, x1, D, dwc1, dbc1)
        _launch(batch, colsum_jobs=[(scratch2, n, dg2, dbe2, nb)])
        # LN1 backward is applied by the out-projection input-gradient GEMM: d(o) = gs1 W_o; gs1 kept
        gs1, dg1, dbe1 = new(n, D), new_grad(g1), new_grad(be1)
        do, dwo, dbo = new(n, D), new_grad(wo), new_grad(bo)
        _launch_ln(2, n, D, dx1, xhat1, g1, None, rstd1, gs1, None, None, scratch1, wo, 0, D, do, D)
        # attention
        dq, dk, dv = new(n, D), new(n, D), new(n, D)
        L.check(lib.dpn_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(P), _p(do), ctx.Lt, ctx.B, _p(dq), _p(dk), _p(dv), _s()), 'dpn_attn_bwd')
        # q/k/v projections: dx = dq Wq + dk Wk + dv Wv + gs1 (the residual branch)
        dx, dwq, dwk, dwv, dbq, dbk, dbv = new(n, D), new_grad(wq), new_grad(wk), new_grad(wv), new_grad(bq), new_grad(bk), new_grad(bv)
        batch = [_problem(n, D, D, [(dq, D, wq, D), (dk, D, wk, D), (dv, D, wv, D)], dx, D, 0, 0, epi=L.EPI_ADD, aux=gs1)]
        for gq, gw, gb in ((dq, dwq, dbq), (dk, dwk, dbk), (dv, dwv, dbv)):
            _wgrad(batch, D, D, n, gq, D, x, D, gw, gb)
        _wgrad(batch, D, D, n, gs1, D, o, D, dwo, dbo)              # dW_o = gs1^T o
        _launch(batch, colsum_jobs=[(scratch1, n, dg1, dbe1, nb)])
        return dx, None, None, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dg1, dbe1, dwc1, dbc1, dwc2, dbc2, dg2, dbe2


def encoder_layer_fused(x, layer):
    """EncoderLayer.forward for [B, L, 256] fp32 device tensors with the shipped shapes (8 heads x 32, gelu, LayerNorm eps 1e-5,
    all biases present); returns None when the layer does not fit, so that the caller takes the per-op path."""
    att = layer.attention
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == 256 and x.shape[1] <= 288):
        return None
    if not (att.n_heads == 8 and not att.mix and layer.activation is F.gelu and layer.norm1.eps == 1e-5 and layer.norm2.eps == 1e-5
            and layer.norm1.elementwise_affine and layer.norm2.elementwise_affine and layer.conv1.weight.shape[1] == 256
            and layer.conv2.weight.shape[0] == 256 and att.query_projection.weight.shape == (256, 256)
            and all(m.bias is not None for m in (att.query_projection, att.key_projection, att.value_projection, att.out_projection,
                                                 layer.conv1, layer.conv2))):
        return None
    B, Lt = x.shape[0], x.shape[1]
    out = _EncoderLayerFn.apply(x.reshape(B * Lt, 256), B, Lt, att.query_projection.weight, att.query_projection.bias, att.key_projection.weight,
                                att.key_projection.bias, att.value_projection.weight, att.value_projection.bias,
                                att.out_projection.weight, att.out_projection.bias, layer.norm1.weight, layer.norm1.bias,
                                layer.conv1.weight.squeeze(-1), layer.conv1.bias, layer.conv2.weight.squeeze(-1), layer.conv2.bias,
                                layer.norm2.weight, layer.norm2.bias)
    return out.view(B, Lt, 256)              # views, not x[0]: a select's backward is a zero fill + a copy


def lead_time_pe(h, freq_bands):
    """SineCosPE(1, include_input=False) of the scalar lead time of each field: h with B elements -> [B, 2 * N_freqs] ([2 * N_freqs] for B = 1)."""
    B = h.numel()
    out = torch.empty((B, 2 * freq_bands.numel()), dtype=torch.float32, device=h.device)
    L.check(L.load().dpn_lead_pe(_p(_c(h.detach().float().reshape(-1))), B, _p(_c(freq_bands)), freq_bands.numel(), _p(out), None, 0, None, _s()),
            'dpn_lead_pe')
    return out.view(-1) if B == 1 else out


class _DataEmbeddingFn(torch.autograd.Function):
    """DataEmbedding + learnable tokens (model/embed.py:60-64, transformer_net.py:124-126) for B field samples:
    x0[b] = cat(token, circular_conv3(field[b])) + pos + time_embedding(h[b]).  Four launches forward (im2col, one 19-way split-K MFMA GEMM
    launch, lead-time PE, assemble + split reduction); backward = one GEMM for the conv weight (already in the parameter's [256][C][3]
    layout) with its bias sum."""

    @staticmethod
    def forward(ctx, field, conv_w, conv_b, token, pos, h, freq_bands, xu=None, te=None, share=None):
        from .linear import _launch, _problem
        lib = L.load()
        B, T, C = field.shape
        D = conv_w.shape[0]
        dev = field.device
        if xu is None:                                               # (encoder_prep has made both when the whole encoder runs fused)
            x = _c(field.detach().reshape(B * T, C).float())
            xu = torch.empty((B * T, 3 * C), dtype=torch.float32, device=dev)
            L.check(lib.dpn_im2col_circ3(_p(x), T, C, B, _p(xu), _s()), 'dpn_im2col_circ3')
        w2 = _c(conv_w).view(D, 3 * C)
        K3 = 3 * C
        c16 = getattr(share, 'conv16', None) if share is not None else None
        if c16 is not None and c16[5].data_ptr() == w2.data_ptr():
            # emb = xu . w2^T on the planes dpn_enc_prep split (f16 hi+lo MFMA, per-row scales): sixteen K-slices, added in order by the assemble
            xs, xe, ws, we, Kp, _ = c16
            n_parts = 16
            emb_parts = torch.empty((n_parts, B * T, D), dtype=torch.float32, device=dev)
            L.check(L.load_experiments().dpn_conv16(_p(xs), _p(xe), _p(ws), _p(we), B * T, D, Kp, n_parts, _p(emb_parts), _s()), 'dpn_conv16')
        elif not config.FROZEN.embed_gemm16:
            # emb = xu . w2^T with K = 3C = 7215: sixteen K-slices as sixteen problems of one exact-fp32 MFMA launch (24 us).  DPN_EMBED_PARTS overrides the
            # number of slices (at most 26 problems per launch): round 6 sweep in tools/embed_parts_bench.py
            # (default since round 6: slices of 384 = six whole 64-deep k-tiles -> 19 slices x 12 output tiles = 228 workgroups, one round of the 256 CUs:
            # GEMM + assemble 24.8 us against 28.5 us for sixteen slices of 451 whose eighth k-tile is 95 % padding; profiles/round6_embed_split_sweep.txt)
            parts = min(26, config.FROZEN.embed_parts) if config.FROZEN.embed_parts > 0 else 19
            ks = (K3 + parts - 1) // parts
            if config.FROZEN.embed_align:                      # slices in whole 64-deep k-tiles of the kernel (no slice ends in a mostly empty tile)
                ks = (ks + 63) // 64 * 64
            bounds = [(k0, min(k0 + ks, K3)) for k0 in range(0, K3, ks)]
            emb_parts = torch.empty((len(bounds), B * T, D), dtype=torch.float32, device=dev)
            problems = []
            for i, (k0, k1) in enumerate(bounds):
                q = _problem(B * T, D, k1 - k0, [(xu, K3, w2, K3)], emb_parts, D, 0, 1)
                q.A[0], q.B[0], q.C = xu.data_ptr() + k0 * 4, w2.data_ptr() + k0 * 4, emb_parts.data_ptr() + i * B * T * D * 4
                problems.append(q)
            _launch(problems)
            n_parts = len(bounds)
        else:
            # (measured experiment, DPN_EMBED_GEMM16=1: the same product on the f16 hi+lo MFMA GEMM dpn_gemm16.  With the kernel's
            # "k is contiguous" load path (two 16-byte buffer loads per operand and block) 17 us at 38 K-slices + 11.6 us assemble, 20.6 + 6.4 us
            # at 16 (DPN_EMBED_PARTS) against 23.7 + 6.2 us for the exact-fp32 split-K launch: the kernel spends ~2 300 cycles per 32-k block on
            # scales and splits for 12 MFMAs, and the tile-level scales make a field's result depend on its batch neighbours at the 1e-7
            # level; not the product path.  What would pay is operands split ONCE -- by dpn_enc_prep, per-row scales -- and a load-only
            # kernel; at 64 x 64 tiles that GEMM moves 74 MB through L2 for 1 GFLOP.)
            tiles = ((B * T + 63) // 64) * ((D + 63) // 64)
            n_parts = config.FROZEN.embed_parts or max(1, min(38, 456 // tiles))
            q = L.DpnGemm16Problem()
            emb_parts = torch.empty((n_parts, B * T, D), dtype=torch.float32, device=dev)
            q.A, q.B, q.C, q.M, q.N, q.K, q.ldc = _p(xu), _p(w2), _p(emb_parts), B * T, D, K3, D
            q.a_sm, q.a_sk, q.b_sn, q.b_sk = K3, 1, K3, 1
            L.check(L.load_experiments().dpn_gemm16(1, ctypes.byref(q), n_parts, _p(emb_parts), 0, _s()), 'dpn_gemm16')
        if te is None:
            te = lead_time_pe(h, freq_bands)
        n_tok = token.shape[-2]
        out = torch.empty((B, n_tok + T, D), dtype=torch.float32, device=dev)
        pending = dict(token=_c(token), n_tok=n_tok, parts=emb_parts, n_parts=n_parts, T=T, B=B, bias=conv_b, pos=_c(pos), te=te, out=out)
        if share is not None and B == 1 and getattr(share, 'defer_assemble', False):
            # one field on the fused path (encoder_forward_fused): the encoder stack's FIRST launch assembles x0 from the split-K slices itself and
            # writes it here -- no launch for it (it was 6.9 us).  `out` is valid once that launch has run; whoever holds `share.pending` and does
            # not hand it to that launch calls assemble_pending() (encoder_forward_fused does, when the stack declines).
            share.pending = pending
        else:
            _assemble(pending)
        ctx.save_for_backward(xu)
        ctx.n_tok, ctx.w_shape, ctx.tok_shape, ctx.B = n_tok, conv_w.shape, token.shape, B
        ctx.params = (conv_w, conv_b)
        ctx.share = share
        if share is not None and B == 1 and not config.FROZEN.embed_own_wgrad:
            # one field: the encoder stack's backward computes the token convolution's weight gradient inside ITS weight-gradient launch
            # (dW = (d x0 rows of the field tokens)^T xu: both operands exist there) and hands it back through `share`
            share.embed = dict(xu=xu, n_tok=n_tok, conv_w=conv_w, conv_b=conv_b, token=token, grads=None, g_tok=None)
        return out

    @staticmethod
    def backward(ctx, g):
        (xu,) = ctx.saved_tensors
        n, K3 = xu.shape
        D, B = ctx.w_shape[0], ctx.B
        g3 = g.reshape(B, -1, D)
        g_emb = _c(g3[:, ctx.n_tok:]).reshape(n, D)              # one field: a contiguous row range (no copy)
        done = getattr(ctx.share, 'embed', None) if ctx.share is not None else None
        # The stack node's weight-gradient launch has computed dW, db (and the learnable tokens' gradient) from ITS d x0 -- valid only when that is
        # the cotangent arriving here, i.e. x0 had exactly one consumer and no hook rewrote its gradient: otherwise (a second use of
        # `last_embedding`, a tensor hook) the incoming g differs and the gradients are computed from it below.
        #  The pointer alone does not prove it: autograd's input buffer may ACCUMULATE a second consumer's gradient in place into d x0 when it
        # holds the only reference, and an in-place tensor hook keeps the pointer too.  The stack node therefore KEEPS a reference to d x0 (the engine
        # then never accumulates into it in place: it adds out of place, a new pointer) and records its version counter, which every in-place write
        # bumps (views share the counter).
        if done is not None and done.get('grads') is not None:
            kept = done.pop('dx0', None)
            same = kept is not None and kept.data_ptr() == g.data_ptr() and g._version == done.get('dx0_version') and kept._version == done.get('dx0_version')
            if not same:
                done['grads'] = done['g_tok'] = None
        if done is not None and done.get('grads') is not None:       # computed by the stack node's weight-gradient launch
            dw, db = done['grads']
            done['grads'] = None
            g_tok = done.get('g_tok')
            done['g_tok'] = None
            g_tok = g_tok.view(ctx.tok_shape) if g_tok is not None else g3[:, :ctx.n_tok].reshape(ctx.tok_shape)
            return None, dw.view(ctx.w_shape), db, g_tok, None, None, None, None, None, None
        dw, db = new_grad(ctx.params[0], (D, K3)), new_grad(ctx.params[1])
        if config.FROZEN.encoder_unfused:
            batch = []
            _wgrad(batch, D, K3, n, g_emb, D, xu, K3, dw, db)
            if batch:
                from .linear import _launch
                _launch(batch)
        else:
            held = wgrad16([(g_emb, xu, dw, db)])                   # (the launch is queued; `held` may go: the stream orders the reuse)
        g_tok = g3[:, :ctx.n_tok]
        g_tok = g_tok.reshape(ctx.tok_shape) if B == 1 else g_tok.sum(dim=0).reshape(ctx.tok_shape)
        return None, dw.view(ctx.w_shape), db, g_tok, None, None, None, None, None, None


def embed_wgrad_rides_with_stack(n_fields=1):
    """Does the token convolution's weight gradient come out of the encoder stack's ONE weight-gradient launch (_EncoderStackFn.backward:
    a single field on the fused path) instead of a launch of the embedding's own backward?  Then the data-parallel step has nothing to
    overlap between the encoder's and the embedding's gradient buckets: they complete together and travel as ONE all-reduce
    (interface_physics.StagedPdeStep, InterfacePhysics.training_step)."""
    return n_fields == 1 and not config.FROZEN.embed_own_wgrad and not config.FROZEN.encoder_unfused


def _assemble(pd):
    """dpn_embed_assemble: x0 = cat(learnable_token, value_embedding) + positional table + lead-time embedding from the split-K slices (its own launch)"""
    L.check(L.load().dpn_embed_assemble(_p(pd['token']), pd['n_tok'], _p(pd['parts']), pd['n_parts'], pd['T'], pd['B'], _p(pd['bias']), _p(pd['pos']),
                                        _p(pd['te']), _p(pd['out']), _s()), 'dpn_embed_assemble')


def assemble_pending(share):
    """Run a deferred assembly now (nobody took it over)."""
    pd = getattr(share, 'pending', None) if share is not None else None
    if pd is not None:
        share.pending = None
        _assemble(pd)


def _params_ok(params, device):
    """Every parameter the fused kernels read through a raw pointer: fp32, on `device`, contiguous (a .half() / .double() model, or one on another
    GPU, would be read as fp32 words of the wrong size -- silently wrong, or out of bounds): otherwise the per-op path handles or rejects it."""
    return all(p is not None and p.is_cuda and p.device == device and p.dtype == torch.float32 and p.is_contiguous() for p in params)


def _embedding_fits(field, emb_module, token, h):
    conv = emb_module.value_embedding.tokenConv
    return (field.is_cuda and field.dim() == 3 and conv.weight.shape[0] == 256 and conv.kernel_size == (3,)
            and conv.bias is not None and token.shape[-1] == 256 and h.numel() == field.shape[0]
            and _params_ok((conv.weight, conv.bias, token, emb_module.position_embedding.pe), field.device))


def data_embedding_fused(field, emb_module, token, h, prep=None):
    """-> [B, n_tok + T, 256] or None when the module does not fit the kernels (then the caller takes the per-op path).
    prep: an EncoderPrep of the same (field, h): its im2col rows and lead-time encoding are used instead of two launches here."""
    conv = emb_module.value_embedding.tokenConv
    if not _embedding_fits(field, emb_module, token, h):
        return None
    n = token.shape[-2] + field.shape[1]
    pos = emb_module.position_embedding.pe[0, :n]
    if prep is not None:
        return _DataEmbeddingFn.apply(field, conv.weight, conv.bias, token, pos, h, emb_module.time_embending.freq_bands, prep.xu, prep.te, prep)
    return _DataEmbeddingFn.apply(field, conv.weight, conv.bias, token, pos, h, emb_module.time_embending.freq_bands)


HEAD_WIDTHS = (193,) * 6 + (257,) * 6                  # coord_input_fc (w1 | b1) of the six nets, then coord_hidden_fc (w2 | b2)
HEADS_COLS = sum(HEAD_WIDTHS)


class _HeadsFn(torch.autograd.Function):
    """The twelve hyper-network heads and the six lead-time embeddings of a PhysicsNet (model/variable_net.py:57-65,75-78), per field
    sample ONE launch forward (18 GEMM problems reading the encoder output transposed in place) and two launches backward (the input
    gradient as six 2-term problems joined by dpn_sum_parts, twelve weight gradients with their bias sums, six outer products).
    inputs: meta [B, L, 256] (tokens 0..255 are used), pe_h [B, 192], 12 head weights, 12 head biases, 6 fore_h_fc weights, 6 biases
    -> heads [B, 256, 2700] = [w1b1 of nets 0..5 | w2b2 of nets 0..5] per hidden channel, evec [B, 6, 256].
    With B > 1 the parameter gradients of the fields are written side by side and added in a fixed order by one dpn_sum_parts."""

    @staticmethod
    def forward(ctx, meta, pe_h, *wb):
        from .linear import _launch, _problem
        hw, hb, fw, fb = wb[0:12], wb[12:24], wb[24:30], wb[30:36]
        hw = [_c(w) for w in hw]
        fw = [_c(w) for w in fw]
        B, Lt = meta.shape[0], meta.shape[1]
        m3 = _c(meta)                                            # [B][L tokens][256 channels]
        pe2 = _c(pe_h.reshape(B, 192))
        dev = m3.device
        heads = torch.empty((B, 256, HEADS_COLS), dtype=torch.float32, device=dev)
        evec = torch.empty((B, 6, 256), dtype=torch.float32, device=dev)
        mode = config.FROZEN.heads_per_field         # A/B measurements: 1 = the per-field launches of rounds 1-3; fwd / bwd = only that pass per field
        ctx.per_field_bwd = B > 1 and mode in ('1', 'bwd')
        if B > 1 and mode not in ('1', 'fwd'):
            # B fields: ONE launch over all of them (it was one per field: 61 launches of 15 us at configs[2]).  The encoder output of the
            # fields is brought into [field * 256 + channel][token] order once (16 MB at B = 61), so that a head is ONE problem with
            # B * 256 rows; the backward pass reuses the copy as the K-operand of the weight gradients.
            acat = m3[:, :256, :].transpose(1, 2).contiguous().view(B * 256, 256)
            problems, off = [], 0
            for w, b in zip(hw, hb):                             # heads[f][c][off + j] = sum_tok meta[f][tok][c] W[j][tok] + b[j]
                n_k = w.shape[0]
                q = _problem(B * 256, n_k, 256, [(acat, 256, w, 256)], heads, HEADS_COLS, 0, 1, bias=b)
                q.C = heads.data_ptr() + off * 4
                problems.append(q)
                off += n_k
            for k, (w, b) in enumerate(zip(fw, fb)):             # evec[f][k] = fore_h_fc_k(pe_h[f])
                q = _problem(B, 256, 192, [(pe2, 192, w, 192)], evec, 6 * 256, 0, 1, bias=b)
                q.C = evec.data_ptr() + k * 256 * 4
                problems.append(q)
            _launch(problems)
            ctx.save_for_backward(m3, pe2, acat, *hw)
            ctx.batched = True
            ctx.has_acat = True
            ctx.params = (hb, fw, fb)
            return heads, evec
        ctx.batched = B > 1
        ctx.has_acat = False
        for f in range(B):
            m_ptr, h_ptr = m3.data_ptr() + f * Lt * 256 * 4, heads.data_ptr() + f * 256 * HEADS_COLS * 4
            problems, off = [], 0
            for w, b in zip(hw, hb):                             # heads[c][off + j] = sum_tok meta[tok][c] W[j][tok] + b[j]
                n_k = w.shape[0]
                q = _problem(256, n_k, 256, [(m3, 256, w, 256)], heads, HEADS_COLS, 1, 1, bias=b)
                q.A[0], q.C = m_ptr, h_ptr + off * 4
                problems.append(q)
                off += n_k
            for k, (w, b) in enumerate(zip(fw, fb)):             # evec[k] = fore_h_fc_k(pe_h)
                q = _problem(1, 256, 192, [(pe2, 192, w, 192)], evec, 256, 0, 1, bias=b)
                q.A[0], q.C = pe2.data_ptr() + f * 192 * 4, evec.data_ptr() + (f * 6 + k) * 256 * 4
                problems.append(q)
            _launch(problems)
        ctx.save_for_backward(m3, pe2, *hw)
        ctx.params = (hb, fw, fb)                                    # identify the gradient slots (grad_arena); fw: the contiguous weights
        return heads, evec

    @staticmethod
    def backward(ctx, g_heads, g_evec):
        from .linear import _launch, _problem
        if ctx.batched and not ctx.per_field_bwd:
            return _HeadsFn._backward_batched(ctx, g_heads, g_evec)
        if ctx.has_acat:
            m3, pe2, _, *hw = ctx.saved_tensors
        else:
            m3, pe2, *hw = ctx.saved_tensors
        dev = m3.device
        B, Lt = m3.shape[0], m3.shape[1]
        lib = L.load()
        gh, ge = _c(g_heads), _c(g_evec)
        d_meta = torch.empty(m3.shape, dtype=torch.float32, device=dev)
        offs, off = [], 0
        for w in hw:
            offs.append(off)
            off += w.shape[0]
        # destinations of the 36 parameter gradients (12 head weights, 12 head biases, 6 fore_h_fc weights, 6 fore_h_fc biases).
        # One field: the parameters' own gradient tensors (slots of the optimiser's flat gradient buffer when one is registered,
        # grad_arena).  B fields: side by side in [B][flat], added in a fixed order by one dpn_sum_parts.
        hb, fw, fb = ctx.params
        shapes = [(w.shape[0], 256) for w in hw] + [(w.shape[0],) for w in hw] + [(256, 192)] * 6 + [(256,)] * 6
        if B == 1:
            dest = [new_grad(t, shp) for t, shp in zip(list(hw) + list(hb) + list(fw) + list(fb), shapes)]
            ptrs = [[d.data_ptr() for d in dest]]
        else:
            starts = [0]
            for shp in shapes:
                starts.append(starts[-1] + int(torch.Size(shp).numel()))
            flat = torch.empty((B, starts[-1]), dtype=torch.float32, device=dev)
            ptrs = [[flat.data_ptr() + (f * starts[-1] + starts[i]) * 4 for i in range(36)] for f in range(B)]
        # d meta as NP accumulated-term problems side by side, joined by dpn_sum_parts: 6 two-term problems (with the 12 weight gradients and
        # the 6 outer products exactly the 24 problems a launch takes).  Same box, 300-step runs, three times each: NP = 2 1.614 ms per
        # step, 3 1.591, 4 1.614 (rounds 1-3), 6 1.580
        NP = config.FROZEN.heads_dmeta_parts
        if NP not in (1, 2, 3, 4, 6, 12):
            raise ValueError('DPN_HEADS_DMETA_PARTS must divide the twelve heads (1, 2, 3, 4, 6 or 12), got %d' % NP)
        parts = torch.empty((NP, 256, 256), dtype=torch.float32, device=dev)
        n_tail = (Lt - 256) * 256                                # tokens >= 256 feed no VariableNet: their gradient rows are zero
        for f in range(B):
            g_ptr, m_ptr = gh.data_ptr() + f * 256 * HEADS_COLS * 4, m3.data_ptr() + f * Lt * 256 * 4
            # d_meta[tok][c] = sum_k sum_j W_k[j][tok] g[c][off_k + j]: a 12-term problem would walk 24 k-tiles in sequence, so NP
            # problems of 12 / NP terms run side by side and dpn_sum_parts joins them
            problems = []
            for p_ in range(NP):
                grp = list(range((12 // NP) * p_, (12 // NP) * (p_ + 1)))
                q0 = _problem(256, 256, 256, [(hw[k], 256, gh, HEADS_COLS, hw[k].shape[0]) for k in grp], parts, 256, 1, 1)
                q0.C = parts.data_ptr() + p_ * 256 * 256 * 4
                for i, k in enumerate(grp):
                    q0.B[i] = g_ptr + offs[k] * 4
                problems.append(q0)
            for k, w in enumerate(hw):                           # dW_k[j][tok] = sum_c g[c][off + j] meta[tok][c] ; db_k[j] = sum_c g[c][off + j]
                n_k = w.shape[0]
                q = _problem(n_k, 256, 256, [(gh, HEADS_COLS, m3, 256)], parts, 256, 1, 1, asum=parts)
                q.A[0], q.B[0], q.C, q.asum = g_ptr + offs[k] * 4, m_ptr, ptrs[f][k], ptrs[f][12 + k]
                problems.append(q)
            for k in range(6):                                   # d fore_h_fc_k.weight = g_evec[k] (outer) pe_h ; its bias gradient = g_evec[k]
                q = _problem(256, 192, 1, [(ge, 256, pe2, 192)], parts, 192, 1, 0, asum=parts)       # = the "row sums" over the K = 1 reduction
                q.A[0], q.B[0], q.C, q.asum = ge.data_ptr() + (f * 6 + k) * 256 * 4, pe2.data_ptr() + f * 192 * 4, ptrs[f][24 + k], ptrs[f][30 + k]
                problems.append(q)
            _launch(problems)
            L.check(lib.dpn_sum_parts(_p(parts), NP, 256 * 256, n_tail, ctypes.c_void_p(d_meta.data_ptr() + f * Lt * 256 * 4), _s()), 'dpn_sum_parts')
        if B > 1:
            total = torch.empty(starts[-1], dtype=torch.float32, device=dev)
            L.check(lib.dpn_sum_parts(_p(flat), B, starts[-1], 0, _p(total), _s()), 'dpn_sum_parts')
            dest = [total[starts[i]:starts[i + 1]].view(shapes[i]) for i in range(36)]
        return (d_meta, None, *dest)

    @staticmethod
    def _backward_batched(ctx, g_heads, g_evec):
        """B > 1: two launches for all fields.  (1) d meta as ONE problem over the B * 256 (field, channel) rows -- twelve accumulated terms,
        one per head --, transposed back into the encoder's [token][channel] order by one copy; (2) the twelve head-weight gradients and the
        six fore_h_fc gradients as reductions over all fields at once (K = B * 256 resp. B): the sum over the fields happens inside the
        GEMM's own fixed-order reduction, nothing is written per field and joined afterwards."""
        from .linear import _launch, _problem
        if ctx.has_acat:
            m3, pe2, acat, *hw = ctx.saved_tensors
        else:
            m3, pe2, *hw = ctx.saved_tensors
            acat = m3[:, :256, :].transpose(1, 2).contiguous().view(m3.shape[0] * 256, 256)
        dev = m3.device
        B, Lt = m3.shape[0], m3.shape[1]
        gh, ge = _c(g_heads), _c(g_evec)                          # [B, 256, 2700] = [(f, c)][j], [B, 6, 256]
        offs, off = [], 0
        for w in hw:
            offs.append(off)
            off += w.shape[0]
        hb, fw, fb = ctx.params
        shapes = [(w.shape[0], 256) for w in hw] + [(w.shape[0],) for w in hw] + [(256, 192)] * 6 + [(256,)] * 6
        dest = [new_grad(t, shp) for t, shp in zip(list(hw) + list(hb) + list(fw) + list(fb), shapes)]
        # (1) dmt[(f, c)][tok] = sum_k sum_j g[(f, c)][off_k + j] W_k[j][tok]
        dmt = torch.empty((B * 256, 256), dtype=torch.float32, device=dev)
        q0 = _problem(B * 256, 256, 256, [(gh, HEADS_COLS, hw[k], 256, hw[k].shape[0]) for k in range(12)], dmt, 256, 0, 0)
        for k in range(12):
            q0.A[k] = gh.data_ptr() + offs[k] * 4
        _launch([q0])
        d_meta = torch.empty(m3.shape, dtype=torch.float32, device=dev)
        d_meta[:, :256, :].copy_(dmt.view(B, 256, 256).transpose(1, 2))
        if Lt > 256:
            d_meta[:, 256:, :].zero_()                           # tokens >= 256 feed no VariableNet
        # (2) dW_k[j][tok] = sum_(f, c) g[(f, c)][off_k + j] acat[(f, c)][tok] ; db_k[j] = sum_(f, c) g[(f, c)][off_k + j]
        problems = []
        for k, w in enumerate(hw):
            n_k = w.shape[0]
            q = _problem(n_k, 256, B * 256, [(gh, HEADS_COLS, acat, 256)], dest[k], 256, 1, 0, asum=dest[12 + k])
            q.A[0] = gh.data_ptr() + offs[k] * 4
            problems.append(q)
        for k in range(6):                                       # d fore_h_fc_k.weight[c][i] = sum_f g_evec[f][k][c] pe_h[f][i] ; bias: sum_f g_evec[f][k][c]
            q = _problem(256, 192, B, [(ge, 6 * 256, pe2, 192)], dest[24 + k], 192, 1, 0, asum=dest[30 + k])
            q.A[0] = ge.data_ptr() + k * 256 * 4
            problems.append(q)
        _launch(problems)
        return (d_meta, None, *dest)


# ------------------------------------------------------------------------------------------------ row-local fused encoder (round 4)
_enc_status = {}


def enc_status(device):
    """Device-side status word of dpn_enc_pack (bit 0: a weight entry outside the f16 split's range, |w| >= 32768)."""
    key = (device.type, device.index)
    if key not in _enc_status:
        _enc_status[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _enc_status[key]


def check_enc_status():
    """Synchronising check of the status words (call where the step synchronises anyway: logging, checkpoints)."""
    for t in _enc_status.values():
        if int(t.item()) != 0:
            raise RuntimeError('deepphysinet_amd: an encoder weight matrix holds an entry with |w| >= 32768 (or a non-finite one): outside the '
                               'range of the f16 hi+lo operand split of dpn_enc_fwd / dpn_enc_bwd')


def reset_enc_status():
    """Clear the (sticky) status words: a caller that has handled the condition -- bench.py discarding a run whose trajectory went non-finite."""
    for t in _enc_status.values():
        t.zero_()


def enc_pack(mats):
    """MFMA-fragment images (x W^T and g W forms) of the [256, 256] fp32 matrices `mats` (dpn_enc_pack): one launch."""
    lib = L.load()
    n = len(mats)
    if not 0 < n <= L.ENC_MAX_MATS:
        raise ValueError('enc_pack takes 1..%d matrices' % L.ENC_MAX_MATS)
    for m in mats:
        if tuple(m.shape) != (256, 256) or m.dtype != torch.float32 or not m.is_contiguous():
            raise ValueError('enc_pack: contiguous [256, 256] fp32 matrices only, got %s' % (tuple(m.shape),))
    dev = mats[0].device
    buf = torch.empty(int(lib.dpn_enc_pack_bytes(n)), dtype=torch.uint8, device=dev)
    arr = (ctypes.c_void_p * n)(*[_p(m).value for m in mats])
    L.check(lib.dpn_enc_pack(n, arr, _p(buf), _p(enc_status(dev)), _s()), 'dpn_enc_pack')
    return buf


def wgrad16(problems, jobs=(), rows=None):
    """dpn_wgrad16: dW = G^T X (+ db = column sums of G) for a list of (G, X, dW, db) 2-D tensors (G [rows, M], X [rows, N], dW [M, N], db [M] or
    None) in one launch, with the LayerNorm parameter-sum jobs (partial, dgamma, dbeta, n_blocks) riding along; long reductions (batches of
    fields) are cut into row slices joined by the library's second launch.  Raw pointers: the caller keeps the tensors alive."""
    lib = L.load()
    n = len(problems)
    arr = (L.DpnWgradProblem * max(n, 1))()
    rows_max = 0
    for i, (G, X, dW, db) in enumerate(problems):
        q = arr[i]
        q.G, q.X, q.dW, q.db = _p(G), _p(X), _p(dW), _p(db)
        q.M, q.N, q.rows, q.ldg, q.ldx, q.ldw = G.shape[1], X.shape[1], G.shape[0], G.stride(0), X.stride(0), dW.stride(0)
        assert X.shape[0] == G.shape[0] and tuple(dW.shape) == (G.shape[1], X.shape[1]) and G.stride(1) == 1 and X.stride(1) == 1 and dW.stride(1) == 1
        rows_max = max(rows_max, G.shape[0])
    slices = 1 if rows_max <= 2048 else min(32, (rows_max + 2047) // 2048)
    partials = None
    if slices > 1 and n:
        dev = problems[0][0].device
        partials = torch.empty(int(lib.dpn_wgrad16_partial_floats(n, arr, slices)), dtype=torch.float32, device=dev)
    jarr = (L.DpnColsumJob * max(len(jobs), 1))(*[L.DpnColsumJob(j[0].data_ptr(), j[1].data_ptr(), j[2].data_ptr(), j[3]) for j in jobs])
    L.check(lib.dpn_wgrad16(n, arr, len(jobs), jarr, slices, _p(partials), _s()), 'dpn_wgrad16')
    return partials


_LAYER_PARAMS = 16     # wq, bq, wk, bk, wv, bv, wo, bo, g1, be1, wc1, bc1, wc2, bc2, g2, be2


class _EncoderStackFn(torch.autograd.Function):
    """nl EncoderLayers (+ encoder.norm and the output projection when `final`) as ONE autograd node on the row-local fused kernels
    (csrc/dpn_encoder_chain.hip): forward = pack + q/k/v + nl x (attention, layer tail with the next layer's q/k/v) launches, backward =
    nl x (row-local chain, attention backward) + the first layer's q/k/v backward + ONE launch for all weight gradients and the
    LayerNorm parameter sums.  Reference: model/transformer_net.py:28-44,54-72,129, model/attn.py:177-196.
    x0: [B * Lt, 256]; parameters per layer: wq, bq, wk, bk, wv, bv, wo, bo, g1, be1, wc1, bc1, wc2, bc2, g2, be2 (conv weights as [256, 256]),
    then gf, bef, wp, bp when final."""

    @staticmethod
    def forward(ctx, x0, B, Lt, nl, final, wpack, share, *params):
        lib = L.load()
        x0 = _c(x0)
        n, D = x0.shape
        assert n == B * Lt and D == 256 and len(params) == _LAYER_PARAMS * nl + (4 if final else 0)
        dev = x0.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        lay = [params[_LAYER_PARAMS * l:_LAYER_PARAMS * (l + 1)] for l in range(nl)]
        fin = params[_LAYER_PARAMS * nl:] if final else None
        n_mats = 6 * nl + (1 if final else 0)                        # 6 l + (q, k, v, o, c1, c2), then the projection
        if wpack is None:                                            # (encoder_prep has packed them when the whole encoder runs fused)
            wpack = enc_pack(_stack_matrices(lay, fin))
        rt = config.FROZEN.enc_row_tiles or (1 if n <= 2048 else 2)
        stream = _s()

        def fwd(**kw):
            f = L.DpnEncFwd()
            f.wpack, f.n_mats, f.rows, f.row_tiles = _p(wpack), n_mats, n, rt
            for k_, v_ in kw.items():
                setattr(f, k_, _p(v_) if isinstance(v_, torch.Tensor) else v_)
            L.check(lib.dpn_enc_fwd(ctypes.byref(f), stream), 'dpn_enc_fwd')
        q, k, v = new(n, D), new(n, D), new(n, D)
        pd = getattr(share, 'pending', None) if share is not None else None
        if pd is not None:
            share.pending = None
        if pd is not None and B == 1 and pd['B'] == 1 and pd['out'].data_ptr() == x0.data_ptr() and n == pd['n_tok'] + pd['T'] and rt == 1:
            # the data embedding left its assembly to this launch (x0 is written by it, then read by everything below as before)
            fwd(tail=0, next=1, xin=x0, m_n0=0, m_n1=1, m_n2=2, bn0=lay[0][1], bn1=lay[0][3], bn2=lay[0][5], y0=q, y1=k, y2=v,
                emb_parts=pd['parts'], emb_bias=pd['bias'], emb_pos=pd['pos'], emb_te=pd['te'], emb_token=pd['token'], emb_out=x0,
                emb_part_stride=pd['parts'].stride(0), emb_n_parts=pd['n_parts'], emb_n_tok=pd['n_tok'])
            del pd
        else:
            if pd is not None:
                _assemble(pd)                                            # (another tensor arrived than the one the embedding deferred: assemble it first)
            fwd(tail=0, next=1, xin=x0, m_n0=0, m_n1=1, m_n2=2, bn0=lay[0][1], bn1=lay[0][3], bn2=lay[0][5], y0=q, y1=k, y2=v)
        x, saved, out = x0, [], None
        xf = xhatf = rstdf = None
        for l in range(nl):
            p_ = lay[l]
            o, P = new(n, D), new(B * 8, 288, 288)
            attn = lib.dpn_attn_fwd if config.FROZEN.attn_fwd_fp32 else lib.dpn_attn16_fwd      # (round 3's exact-fp32 kernel: A/B runs)
            L.check(attn(_p(q), _p(k), _p(v), Lt, B, _p(o), _p(P), stream), 'dpn_attn_fwd')
            x1, xhat1, rstd1, pre, act, x2, xhat2, rstd2 = new(n, D), new(n, D), new(n), new(n, D), new(n, D), new(n, D), new(n, D), new(n)
            kw = dict(tail=1, o=o, x=x, m_o=6 * l + 3, m_c1=6 * l + 4, m_c2=6 * l + 5, bo=p_[7], g1=p_[8], be1=p_[9], bc1=p_[11], bc2=p_[13],
                      g2=p_[14], be2=p_[15], x1=x1, xhat1=xhat1, rstd1=rstd1, pre=pre, act=act, x2=x2, xhat2=xhat2, rstd2=rstd2)
            qn = kn = vn = None
            if l + 1 < nl:
                qn, kn, vn = new(n, D), new(n, D), new(n, D)
                pn = lay[l + 1]
                kw.update(next=1, m_n0=6 * l + 6, m_n1=6 * l + 7, m_n2=6 * l + 8, bn0=pn[1], bn1=pn[3], bn2=pn[5], y0=qn, y1=kn, y2=vn)
            elif final:
                xf, xhatf, rstdf, out = new(n, D), new(n, D), new(n), new(n, D)
                kw.update(next=2, m_n0=6 * nl, gf=fin[0], bef=fin[1], bn0=fin[3], xf=xf, xhatf=xhatf, rstdf=rstdf, y0=out)
            else:
                kw.update(next=0)
                out = x2
            fwd(**kw)
            saved += [x, q, k, v, o, P, x1, xhat1, rstd1, pre, act, xhat2, rstd2]
            x, q, k, v = x2, qn, kn, vn
        ctx.save_for_backward(wpack, *saved, *([xf, xhatf, rstdf] if final else []), *[p_[8] for p_ in lay], *[p_[14] for p_ in lay],
                              *([fin[0]] if final else []))
        ctx.B, ctx.Lt, ctx.nl, ctx.final, ctx.rt, ctx.n_mats = B, Lt, nl, final, rt, n_mats
        ctx.share = share
        ctx.params = params                                          # identify the gradient slots (grad_arena); never read
        return out

    @staticmethod
    def backward(ctx, g):
        from .linear import _launch
        lib = L.load()
        B, Lt, nl, final, rt, n_mats = ctx.B, ctx.Lt, ctx.nl, ctx.final, ctx.rt, ctx.n_mats
        t = list(ctx.saved_tensors)
        wpack, t = t[0], t[1:]
        lsaved = [t[13 * l:13 * (l + 1)] for l in range(nl)]
        t = t[13 * nl:]
        if final:
            xf, xhatf, rstdf = t[:3]
            t = t[3:]
        g1s, g2s, t = t[:nl], t[nl:2 * nl], t[2 * nl:]
        gf = t[0] if final else None
        g = _c(g)
        n, D = g.shape
        dev = g.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        nb = (n + 16 * rt - 1) // (16 * rt)
        stream = _s()
        lay = [ctx.params[_LAYER_PARAMS * l:_LAYER_PARAMS * (l + 1)] for l in range(nl)]
        fin = ctx.params[_LAYER_PARAMS * nl:] if final else None

        def bwd(**kw):
            b = L.DpnEncBwd()
            b.wpack, b.n_mats, b.rows, b.row_tiles = _p(wpack), n_mats, n, rt
            for k_, v_ in kw.items():
                setattr(b, k_, _p(v_) if isinstance(v_, torch.Tensor) else v_)
            L.check(lib.dpn_enc_bwd(ctypes.byref(b), stream), 'dpn_enc_bwd')
        batch, jobs = [], []
        grads = [None] * len(ctx.params)
        keep = []                                                    # the launch at the end reads these through raw pointers

        def wgrad(slot_w, slot_b, N, gmat, xmat):                    # d W = gmat^T xmat, d b = column sums of gmat
            keep.extend((gmat, xmat))
            w = ctx.params[slot_w]
            gw, gb = new_grad(w, (N, D)), new_grad(ctx.params[slot_b])
            batch.append((gmat, xmat, gw, gb))
            grads[slot_w], grads[slot_b] = gw.view(w.shape), gb

        def lnjob(slot_g, slot_b, partial):
            dg, db = new_grad(ctx.params[slot_g]), new_grad(ctx.params[slot_b])
            keep.append(partial)
            jobs.append((partial, dg, db, nb))
            grads[slot_g], grads[slot_b] = dg, db
        res = dq = dk = dv = None
        from . import grad_arena
        miss0 = grad_arena.misses[0]
        # Per-layer weight gradients on the side branch (branch.py): a layer's (G, X) pairs are complete once its attention backward is queued,
        # four launches before the chain ends; only the token convolution's product (it needs d x0) stays behind the last launch.  Only for a
        # single field's row count (batches cut the reductions into slices: their own second launch) and only while every gradient so far is a
        # fresh slot of the optimiser's flat buffer (autograd keeps such a view as param.grad without touching it on the main stream).
        may_fork = branch.enabled('wgrad16') and n <= 2048

        def flush(on_side):
            nonlocal batch, jobs, keep
            if on_side and not (may_fork and grad_arena.misses[0] == miss0):
                return                                               # no branch: everything in ONE launch at the end, as before
            while batch or jobs:
                b_, batch = batch[:L.WGRAD_MAX_PROBLEMS], batch[L.WGRAD_MAX_PROBLEMS:]
                j_, jobs = jobs[:L.GEMM_MAX_JOBS], jobs[L.GEMM_MAX_JOBS:]
                if on_side:
                    # keep: the operands (G, X, LayerNorm partials) -- NOT the gradient outputs: they are slots of the optimiser's flat buffer
                    # (persistent), and a second reference would make autograd copy each of them instead of keeping the view as param.grad
                    with branch.side(keep=tuple(keep)):
                        wgrad16(b_, j_)
                else:
                    keep.append(wgrad16(b_, j_))
            if on_side:
                keep = []
        for l in range(nl - 1, -1, -1):
            x, q, k, v, o, P, x1, xhat1, rstd1, pre, act, xhat2, rstd2 = lsaved[l]
            gs2, dpre, gs1, do = new(n, D), new(n, D), new(n, D), new(n, D)
            p2, p1 = new(nb * 512), new(nb * 512)
            kw = dict(body=1, m_c2=6 * l + 5, m_c1=6 * l + 4, m_o=6 * l + 3, xhat2=xhat2, rstd2=rstd2, pre=pre, xhat1=xhat1, rstd1=rstd1,
                      g2=g2s[l], g1=g1s[l], gs2=gs2, dpre=dpre, gs1=gs1, dout=do, partial2=p2, partial1=p1)
            base = _LAYER_PARAMS * l
            if l == nl - 1:
                if final:
                    pf = new(nb * 512)
                    kw.update(head=2, dmeta=g, m_h0=6 * nl, xhatf=xhatf, rstdf=rstdf, gf=gf, partial_f=pf)
                    fb = _LAYER_PARAMS * nl
                    lnjob(fb, fb + 1, pf)
                    wgrad(fb + 2, fb + 3, D, g, xf)
                else:
                    kw.update(head=0, gin=g)
            else:
                kw.update(head=1, res=res, dq=dq, dk=dk, dv=dv, m_h0=6 * l + 6, m_h1=6 * l + 7, m_h2=6 * l + 8)
            bwd(**kw)
            lnjob(base + 14, base + 15, p2)
            lnjob(base + 8, base + 9, p1)
            wgrad(base + 12, base + 13, D, gs2, act)                 # conv2
            wgrad(base + 10, base + 11, D, dpre, x1)                 # conv1
            wgrad(base + 6, base + 7, D, gs1, o)                     # out projection
            dq, dk, dv = new(n, D), new(n, D), new(n, D)
            L.check(lib.dpn_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(P), _p(do), Lt, B, _p(dq), _p(dk), _p(dv), stream), 'dpn_attn_bwd')
            wgrad(base + 0, base + 1, D, dq, x)
            wgrad(base + 2, base + 3, D, dk, x)
            wgrad(base + 4, base + 5, D, dv, x)
            res = gs1
            if l == 1:
                flush(True)                                          # ONE fork: everything above the first layer (a fork / join pair costs 10-16 us)
        dx0 = new(n, D)
        emb = getattr(ctx.share, 'embed', None) if ctx.share is not None else None
        if emb is not None and B == 1 and n < 1024 and emb.get('token') is not None:
            # the learnable tokens' rows of d x0 also go straight to that parameter's gradient slot (no copy node behind the backward pass)
            g_tok = new_grad(emb['token'], (emb['n_tok'], D))
            bwd(head=1, body=0, res=res, dq=dq, dk=dk, dv=dv, m_h0=0, m_h1=1, m_h2=2, gx=dx0, gx_head=g_tok, gx_head_rows=emb['n_tok'])
            emb['g_tok'] = g_tok
        else:
            bwd(head=1, body=0, res=res, dq=dq, dk=dk, dv=dv, m_h0=0, m_h1=1, m_h2=2, gx=dx0)
        if emb is not None and B == 1 and n < 1024:
            # the token convolution's weight gradient joins the launch: G = the field-token rows of d x0, X = the im2col rows (embed.py:45-47)
            g_emb = dx0[emb['n_tok']:]
            dwt = new_grad(emb['conv_w'], (D, emb['xu'].shape[1]))
            dbt = new_grad(emb['conv_b'])
            batch.append((g_emb, emb['xu'], dwt, dbt))
            keep.append(g_emb)
            emb['grads'] = (dwt, dbt)
            emb['dx0'], emb['dx0_version'] = dx0, dx0._version      # see _DataEmbeddingFn.backward: identity AND version of the cotangent
        # every weight gradient of the stack and the LayerNorm parameter sums: ONE launch (dpn_wgrad16; plus its slice reduction for batches
        # of fields)
        flush(False)
        del keep
        return (dx0, None, None, None, None, None, None, *grads)


def _stack_matrices(lay, fin):
    mats = []
    for p_ in lay:
        mats += [_c(p_[0]), _c(p_[2]), _c(p_[4]), _c(p_[6]), _c(p_[10]), _c(p_[12])]
    if fin is not None:
        mats.append(_c(fin[2]))
    return mats


def _layer_params(layer):
    att = layer.attention
    return (att.query_projection.weight, att.query_projection.bias, att.key_projection.weight, att.key_projection.bias,
            att.value_projection.weight, att.value_projection.bias, att.out_projection.weight, att.out_projection.bias,
            layer.norm1.weight, layer.norm1.bias, layer.conv1.weight.squeeze(-1), layer.conv1.bias, layer.conv2.weight.squeeze(-1),
            layer.conv2.bias, layer.norm2.weight, layer.norm2.bias)


def _layer_fits(layer):
    att = layer.attention
    return (att.n_heads == 8 and not att.mix and layer.activation is F.gelu and layer.norm1.eps == 1e-5 and layer.norm2.eps == 1e-5
            and layer.norm1.elementwise_affine and layer.norm2.elementwise_affine and tuple(layer.conv1.weight.shape) == (256, 256, 1)
            and tuple(layer.conv2.weight.shape) == (256, 256, 1) and att.query_projection.weight.shape == (256, 256)
            and all(m.bias is not None for m in (att.query_projection, att.key_projection, att.value_projection, att.out_projection,
                                                 layer.conv1, layer.conv2)))


def _stack_fits(layers, norm, projection, device=None):
    if config.FROZEN.encoder_fp8 or config.FROZEN.encoder_unfused or len(layers) < 1:
        return False
    if not all(_layer_fits(l_) and not getattr(l_.attention.inner_attention, 'output_attention', False) for l_ in layers):
        return False
    if (norm is None) != (projection is None):
        return False
    final = norm is not None
    if final and not (norm.elementwise_affine and norm.eps == 1e-5 and tuple(projection.weight.shape) == (256, 256) and projection.bias is not None):
        return False
    if device is not None:
        every = [p for l_ in layers for p in l_.parameters()] + ([norm.weight, norm.bias, projection.weight, projection.bias] if final else [])
        if not _params_ok(every, device):
            return False
    return 6 * len(layers) + (1 if final else 0) <= L.ENC_MAX_MATS


class EncoderPrep:
    """dpn_enc_prep's outputs for one forward of the whole encoder: weight images, im2col rows, lead-time encodings."""
    __slots__ = ('wpack', 'xu', 'te', 'pe_extra', 'embed', 'conv16', 'defer_assemble', 'pending')


def encoder_prep(field, h, emb_module, extra_freqs, layers, norm, projection):
    """ONE launch for everything of the encoder forward that depends on the step's inputs only (they were four: dpn_enc_pack,
    dpn_im2col_circ3 and two dpn_lead_pe): the weight images of the stack, the im2col rows of the token convolution, the encoder's lead-time
    encoding and (extra_freqs: the VariableNets' frequency table) the one the hyper-network heads take."""
    lib = L.load()
    lay = [_layer_params(l_) for l_ in layers]
    mats = _stack_matrices(lay, (None, None, projection.weight))
    B, T, C = field.shape
    dev = field.device
    x = _c(field.detach().reshape(B * T, C).float())
    fa = _c(emb_module.time_embending.freq_bands)
    hh = _c(h.detach().float().reshape(-1))
    out = EncoderPrep()
    out.embed = None
    out.defer_assemble, out.pending = False, None
    out.wpack = torch.empty(int(lib.dpn_enc_pack_bytes(len(mats))), dtype=torch.uint8, device=dev)
    out.xu = torch.empty((B * T, 3 * C), dtype=torch.float32, device=dev)
    te = torch.empty((B, 2 * fa.numel()), dtype=torch.float32, device=dev)
    q = L.DpnEncPrep()
    arr = (ctypes.c_void_p * len(mats))(*[_p(m_).value for m_ in mats])
    q.n_mats, q.weights, q.packed, q.status_dev = len(mats), ctypes.cast(arr, ctypes.c_void_p), _p(out.wpack), _p(enc_status(dev))
    q.x, q.T, q.C, q.batch, q.xu = _p(x), T, C, B, _p(out.xu)
    q.h, q.freqs_a, q.n_a, q.out_a = _p(hh), _p(fa), fa.numel(), _p(te)
    out.conv16 = None
    out.pe_extra = None
    if extra_freqs is not None:
        fb = _c(extra_freqs)
        out.pe_extra = torch.empty((B, 2 * fb.numel()), dtype=torch.float32, device=dev)
        q.freqs_b, q.n_b, q.out_b = _p(fb), fb.numel(), _p(out.pe_extra)
    L.check(lib.dpn_enc_prep(ctypes.byref(q), _s()), 'dpn_enc_prep')
    out.te = te.view(-1) if B == 1 else te
    conv = emb_module.value_embedding.tokenConv
    if config.FROZEN.conv16 and conv.weight.is_cuda and conv.weight.dtype == torch.float32 and tuple(conv.weight.shape[1:]) == (C, 3):
        # EXPERIMENT: the token convolution's operands split into f16 hi / lo fragment images with one power-of-two scale per row (dpn_conv16).
        # Measured: the GEMM 23.7 -> 11.2 us, the split 12-16 us -- no gain; not the product path.
        lib = L.load_experiments()
        Kp, n_out = int(lib.dpn_conv16_kp(3 * C)), conv.weight.shape[0]
        if (B * T + 16) * Kp * 4 < 2 ** 31 - 8192 and Kp <= 29 * 256:
            xs = torch.empty(((B * T + 15) // 16 * 16, 2 * Kp), dtype=torch.float16, device=dev)      # fragment images: 16-row strips x (Kp / 32) blocks x 2 KB
            ws = torch.empty(((n_out + 15) // 16 * 16, 2 * Kp), dtype=torch.float16, device=dev)
            xe = torch.empty(B * T, dtype=torch.int32, device=dev)
            we = torch.empty(n_out, dtype=torch.int32, device=dev)
            cw = _c(conv.weight.detach())
            L.check(lib.dpn_conv16_split(_p(x), T, C, B, _p(cw), n_out, _p(xs), _p(xe), _p(ws), _p(we), _s()), 'dpn_conv16_split')
            out.conv16 = (xs, xe, ws, we, Kp, cw)
    return out


def encoder_forward_fused(net, x_enc, forecast_h):
    """TransformerNet.forward (transformer_net.py:123-129) on the fused nodes: prep (1 launch), data embedding, the encoder stack with
    encoder.norm and the projection; None when a module does not fit.  net.extra_lead_freqs (set by PhysicsNet) asks the prep launch
    for the VariableNets' lead-time encoding too: net.extra_lead_pe = (forecast_h, tensor)."""
    enc, emb = net.encoder, net.enc_embedding
    if not (x_enc.is_cuda and x_enc.dim() == 3 and enc.conv_layers is None and enc.norm is not None):
        return None
    layers = list(enc.attn_layers)
    n_tok = net.learnable_token.shape[-2]
    if not (_stack_fits(layers, enc.norm, net.projection, x_enc.device) and _embedding_fits(x_enc, emb, net.learnable_token, forecast_h)
            and n_tok + x_enc.shape[1] <= 288):
        return None
    prep = encoder_prep(x_enc, forecast_h, emb, getattr(net, 'extra_lead_freqs', None), layers, enc.norm, net.projection)
    net.extra_lead_pe = (forecast_h, prep.pe_extra) if prep.pe_extra is not None else None
    # one field: the stack's first launch assembles x0 itself (DPN_EMBED_DEFER=0: the launch of its own, as in rounds 1-5)
    prep.defer_assemble = bool(config.FROZEN.embed_defer and x_enc.shape[0] == 1 and not config.FROZEN.encoder_fp8 and not config.FROZEN.encoder_unfused)
    x0 = data_embedding_fused(x_enc, emb, net.learnable_token, forecast_h, prep=prep)
    # where a staged backward cuts between the stack and the data embedding -- kept only on request (PhysicsNet.encode_field(keep_embedding=True)):
    # it holds this call's autograd graph
    object.__setattr__(net, 'last_embedding', x0 if getattr(net, 'keep_last_embedding', False) else None)
    out = encoder_stack_fused(x0, layers, enc.norm, net.projection, wpack=prep.wpack, share=prep)
    assemble_pending(prep)                                           # (only if the stack declined and nobody took the assembly over)
    return out


def encoder_stack_fused(x, layers, norm=None, projection=None, wpack=None, share=None):
    """[B, L, 256] -> the encoder layers (+ encoder.norm + output projection when both are given) as one autograd node, or None when the
    modules do not fit the kernels (8 heads x 32, d_ff = 256, gelu, affine LayerNorms with eps 1e-5, L <= 288)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == 256 and x.shape[1] <= 288 and len(layers) >= 1):
        return None
    if config.FROZEN.encoder_fp8 or config.FROZEN.encoder_unfused:
        return None
    if not all(_layer_fits(l_) and not getattr(l_.attention.inner_attention, 'output_attention', False) for l_ in layers):
        return None
    final = norm is not None and projection is not None
    if (norm is None) != (projection is None):
        return None
    if 6 * len(layers) + (1 if final else 0) > L.ENC_MAX_MATS:
        return None
    params = []
    for l_ in layers:
        params += list(_layer_params(l_))
    if final:
        if not (norm.elementwise_affine and norm.eps == 1e-5 and tuple(projection.weight.shape) == (256, 256) and projection.bias is not None):
            return None
        params += [norm.weight, norm.bias, projection.weight, projection.bias]
    if not _params_ok([p.detach() for p in params], x.device):
        return None
    B, Lt = x.shape[0], x.shape[1]
    out = _EncoderStackFn.apply(x.reshape(B * Lt, 256), B, Lt, len(layers), final, wpack, share, *params)
    return out.view(B, Lt, 256)
