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
        L.check(lib.dpn_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(P), _p(go), q.shape[0], _p(dq), _p(dk), _p(dv), _s()), 'dpn_attn_bwd')
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


class _EncoderLayerFn(torch.autograd.Function):
    """One EncoderLayer (model/transformer_net.py:28-44 + attn.py:177-196) as a single autograd node with a hand-scheduled backward:
        x1 = LN1(x + out_proj(attention(q(x), k(x), v(x))));  out = LN2(x1 + conv2(gelu(conv1(x1))))
    7 launches forward, 9 backward.  GELU and its derivative ride in the GEMM epilogues, and each residual-branch gradient joins
    the input gradient inside the GEMM that produces it (DPN_EPI_ADD), so no separate elementwise kernel sits on the dependency
    chain of the step.  x: [L, 256]; conv weights as [d_ff, 256] / [256, d_ff] matrices."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo, g1, be1, wc1, bc1, wc2, bc2, g2, be2):
        from .linear import _launch, _problem
        lib = L.load()
        x = _c(x)
        wq, wk, wv, wo, wc1, wc2 = (_c(w) for w in (wq, wk, wv, wo, wc1, wc2))
        n, D = x.shape
        Fh = wc1.shape[0]
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        q, k, v = new(n, D), new(n, D), new(n, D)
        _launch([_problem(n, D, D, [(x, D, w, D)], y, D, 0, 1, bias=b) for w, b, y in ((wq, bq, q), (wk, bk, k), (wv, bv, v))])
        o, P = new(n, D), new(8, 288, 288)
        L.check(lib.dpn_attn_fwd(_p(q), _p(k), _p(v), n, _p(o), _p(P), _s()), 'dpn_attn_fwd')
        a = new(n, D)
        _launch([_problem(n, D, D, [(o, D, wo, D)], a, D, 0, 1, bias=bo)])
        x1, xhat1, rstd1 = new(n, D), new(n, D), new(n)
        L.check(lib.dpn_add_ln_fwd(_p(x), _p(a), _p(g1), _p(be1), n, _p(x1), _p(xhat1), _p(rstd1), _s()), 'dpn_add_ln_fwd')
        pre, act = new(n, Fh), new(n, Fh)
        _launch([_problem(n, Fh, D, [(x1, D, wc1, D)], act, Fh, 0, 1, bias=bc1, epi=L.EPI_GELU, aux_out=pre)])
        y = new(n, D)
        _launch([_problem(n, D, Fh, [(act, Fh, wc2, Fh)], y, D, 0, 1, bias=bc2)])
        out, xhat2, rstd2 = new(n, D), new(n, D), new(n)
        L.check(lib.dpn_add_ln_fwd(_p(x1), _p(y), _p(g2), _p(be2), n, _p(out), _p(xhat2), _p(rstd2), _s()), 'dpn_add_ln_fwd')
        ctx.save_for_backward(x, q, k, v, o, P, x1, pre, act, xhat1, rstd1, xhat2, rstd2, wq, wk, wv, wo, wc1, wc2, g1, g2)
        return out

    @staticmethod
    def backward(ctx, g):
        from .linear import _launch, _problem
        lib = L.load()
        x, q, k, v, o, P, x1, pre, act, xhat1, rstd1, xhat2, rstd2, wq, wk, wv, wo, wc1, wc2, g1, g2 = ctx.saved_tensors
        n, D = x.shape
        Fh = wc1.shape[0]
        dev = x.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        scratch = new(((n + 3) // 4) * 512)
        # LN2 (its parameter sums ride along in the next GEMM launch)
        gs2, dg2, dbe2 = new(n, D), new(D), new(D)
        L.check(lib.dpn_add_ln_bwd(_p(_c(g)), _p(xhat2), _p(rstd2), _p(g2), n, _p(gs2), None, None, _p(scratch), _s()), 'dpn_add_ln_bwd')
        # conv2: d(pre) = (gs2 W_c2) * gelu'(pre) ; dW_c2 = gs2^T act ; db_c2 = sum_rows gs2
        dpre, dwc2, dbc2 = new(n, Fh), new(D, Fh), new(D)
        _launch([_problem(n, Fh, D, [(gs2, D, wc2, Fh)], dpre, Fh, 0, 0, epi=L.EPI_MUL_GELU_GRAD, aux=pre),
                 _problem(D, Fh, n, [(gs2, D, act, Fh)], dwc2, Fh, 1, 0, asum=dbc2)], colsum_jobs=[(scratch, n, dg2, dbe2)])
        # conv1: d(x1) = dpre W_c1 + gs2 (the residual branch) ; dW_c1 = dpre^T x1
        dx1, dwc1, dbc1 = new(n, D), new(Fh, D), new(Fh)
        _launch([_problem(n, D, Fh, [(dpre, Fh, wc1, D)], dx1, D, 0, 0, epi=L.EPI_ADD, aux=gs2),
                 _problem(Fh, D, n, [(dpre, Fh, x1, D)], dwc1, D, 1, 0, asum=dbc1)])
        # LN1
        gs1, dg1, dbe1 = new(n, D), new(D), new(D)
        scratch1 = new(((n + 3) // 4) * 512)
        L.check(lib.dpn_add_ln_bwd(_p(dx1), _p(xhat1), _p(rstd1), _p(g1), n, _p(gs1), None, None, _p(scratch1), _s()), 'dpn_add_ln_bwd')
        # out projection
        do, dwo, dbo = new(n, D), new(D, D), new(D)
        _launch([_problem(n, D, D, [(gs1, D, wo, D)], do, D, 0, 0), _problem(D, D, n, [(gs1, D, o, D)], dwo, D, 1, 0, asum=dbo)],
                colsum_jobs=[(scratch1, n, dg1, dbe1)])
        # attention
        dq, dk, dv = new(n, D), new(n, D), new(n, D)
        L.check(lib.dpn_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(P), _p(do), n, _p(dq), _p(dk), _p(dv), _s()), 'dpn_attn_bwd')
        # q/k/v projections: dx = dq Wq + dk Wk + dv Wv + gs1 (the residual branch)
        dx, dwq, dwk, dwv, dbq, dbk, dbv = new(n, D), new(D, D), new(D, D), new(D, D), new(D), new(D), new(D)
        _launch([_problem(n, D, D, [(dq, D, wq, D), (dk, D, wk, D), (dv, D, wv, D)], dx, D, 0, 0, epi=L.EPI_ADD, aux=gs1),
                 _problem(D, D, n, [(dq, D, x, D)], dwq, D, 1, 0, asum=dbq),
                 _problem(D, D, n, [(dk, D, x, D)], dwk, D, 1, 0, asum=dbk),
                 _problem(D, D, n, [(dv, D, x, D)], dwv, D, 1, 0, asum=dbv)])
        return dx, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dg1, dbe1, dwc1, dbc1, dwc2, dbc2, dg2, dbe2


def encoder_layer_fused(x, layer):
    """EncoderLayer.forward for [1, L, 256] fp32 device tensors with the shipped shapes (8 heads x 32, gelu, LayerNorm eps 1e-5,
    all biases present); returns None when the layer does not fit, so that the caller takes the per-op path."""
    att = layer.attention
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[0] == 1 and x.shape[2] == 256 and x.shape[1] <= 288):
        return None
    if not (att.n_heads == 8 and not att.mix and layer.activation is F.gelu and layer.norm1.eps == 1e-5 and layer.norm2.eps == 1e-5
            and layer.norm1.elementwise_affine and layer.norm2.elementwise_affine and layer.conv1.weight.shape[1] == 256
            and layer.conv2.weight.shape[0] == 256 and att.query_projection.weight.shape == (256, 256)
            and all(m.bias is not None for m in (att.query_projection, att.key_projection, att.value_projection, att.out_projection,
                                                 layer.conv1, layer.conv2))):
        return None
    out = _EncoderLayerFn.apply(x.view(x.shape[1], 256), att.query_projection.weight, att.query_projection.bias, att.key_projection.weight,
                                att.key_projection.bias, att.value_projection.weight, att.value_projection.bias,
                                att.out_projection.weight, att.out_projection.bias, layer.norm1.weight, layer.norm1.bias,
                                layer.conv1.weight.squeeze(-1), layer.conv1.bias, layer.conv2.weight.squeeze(-1), layer.conv2.bias,
                                layer.norm2.weight, layer.norm2.bias)
    return out.view(1, -1, 256)              # views, not x[0]: a select's backward is a zero fill + a copy


def lead_time_pe(h, freq_bands):
    """SineCosPE(1, include_input=False) of the scalar lead time h (a device tensor with one element) -> [2 * N_freqs]."""
    out = torch.empty(2 * freq_bands.numel(), dtype=torch.float32, device=h.device)
    L.check(L.load().dpn_lead_pe(_p(_c(h.detach().float())), _p(_c(freq_bands)), freq_bands.numel(), _p(out), None, 0, None, _s()), 'dpn_lead_pe')
    return out


class _DataEmbeddingFn(torch.autograd.Function):
    """DataEmbedding + learnable tokens (model/embed.py:60-64, transformer_net.py:124-126) for one field sample:
    x0 = cat(token, circular_conv3(field)) + pos + time_embedding(h).  Four launches forward (im2col, one 16-way split-K MFMA GEMM launch, lead-time
    PE, assemble + split reduction); backward = one GEMM for the conv weight (already in the parameter's [256][C][3] layout) with its bias sum."""

    @staticmethod
    def forward(ctx, field, conv_w, conv_b, token, pos, h, freq_bands):
        from .linear import _launch, _problem
        lib = L.load()
        x = _c(field.detach().reshape(field.shape[-2], field.shape[-1]).float())      # [T, C]
        T, C = x.shape
        D = conv_w.shape[0]
        dev = x.device
        xu = torch.empty((T, 3 * C), dtype=torch.float32, device=dev)
        L.check(lib.dpn_im2col_circ3(_p(x), T, C, _p(xu), _s()), 'dpn_im2col_circ3')
        w2 = _c(conv_w).view(D, 3 * C)
        # emb = xu . w2^T with K = 3C = 7215: sixteen K-slices as sixteen problems of one MFMA launch; their partial products are
        # added (fixed order) together with the bias by the assemble kernel
        K3, parts = 3 * C, 16
        ks = (K3 + parts - 1) // parts
        bounds = [(k0, min(k0 + ks, K3)) for k0 in range(0, K3, ks)]
        emb_parts = torch.empty((len(bounds), T, D), dtype=torch.float32, device=dev)
        problems = []
        for i, (k0, k1) in enumerate(bounds):
            q = _problem(T, D, k1 - k0, [(xu, K3, w2, K3)], emb_parts, D, 0, 1)
            q.A[0], q.B[0], q.C = xu.data_ptr() + k0 * 4, w2.data_ptr() + k0 * 4, emb_parts.data_ptr() + i * T * D * 4
            problems.append(q)
        _launch(problems)
        te = lead_time_pe(h, freq_bands)
        n_tok = token.shape[-2]
        out = torch.empty((n_tok + T, D), dtype=torch.float32, device=dev)
        L.check(lib.dpn_embed_assemble(_p(_c(token)), n_tok, _p(emb_parts), len(bounds), T, _p(conv_b), _p(_c(pos)), _p(te), _p(out), _s()),
                'dpn_embed_assemble')
        ctx.save_for_backward(xu)
        ctx.n_tok, ctx.w_shape, ctx.tok_shape = n_tok, conv_w.shape, token.shape
        return out.view(1, n_tok + T, D)

    @staticmethod
    def backward(ctx, g):
        from .linear import _launch, _problem
        (xu,) = ctx.saved_tensors
        T, K3 = xu.shape
        D = ctx.w_shape[0]
        g2 = _c(g.reshape(-1, D))
        g_emb = g2[ctx.n_tok:]                                   # contiguous row range
        dw = torch.empty((D, K3), dtype=torch.float32, device=g.device)
        db = torch.empty((D,), dtype=torch.float32, device=g.device)
        _launch([_problem(D, K3, T, [(g_emb, D, xu, K3)], dw, K3, 1, 0, asum=db)])
        return None, dw.view(ctx.w_shape), db, g2[:ctx.n_tok].view(ctx.tok_shape), None, None, None


def data_embedding_fused(field, emb_module, token, h):
    """-> [1, n_tok + T, 256] or None when the module does not fit the kernels (then the caller takes the per-op path)."""
    conv = emb_module.value_embedding.tokenConv
    if not (field.is_cuda and field.dim() == 3 and field.shape[0] == 1 and conv.weight.shape[0] == 256 and conv.kernel_size == (3,)
            and conv.bias is not None and token.shape[-1] == 256 and h.numel() == 1):
        return None
    n = token.shape[-2] + field.shape[1]
    pos = emb_module.position_embedding.pe[0, :n]
    return _DataEmbeddingFn.apply(field, conv.weight, conv.bias, token, pos, h, emb_module.time_embending.freq_bands)


HEAD_WIDTHS = (193,) * 6 + (257,) * 6                  # coord_input_fc (w1 | b1) of the six nets, then coord_hidden_fc (w2 | b2)
HEADS_COLS = sum(HEAD_WIDTHS)


class _HeadsFn(torch.autograd.Function):
    """The twelve hyper-network heads and the six lead-time embeddings of a PhysicsNet (model/variable_net.py:57-65,75-78) as ONE
    launch forward (18 GEMM problems reading the encoder output transposed in place) and two launches backward (the input gradient as four
    3-term problems joined by dpn_sum_parts, twelve weight gradients with their bias sums, six outer products).
    inputs: meta [1, L, 256] (tokens 0..255 are used), pe_h [192], 12 head weights, 12 head biases, 6 fore_h_fc weights, 6 biases
    -> heads [256, 2700] = [w1b1 of nets 0..5 | w2b2 of nets 0..5] per hidden channel, evec [6, 256]."""

    @staticmethod
    def forward(ctx, meta, pe_h, *wb):
        from .linear import _launch, _problem
        hw, hb, fw, fb = wb[0:12], wb[12:24], wb[24:30], wb[30:36]
        hw = [_c(w) for w in hw]
        fw = [_c(w) for w in fw]
        m2 = _c(meta.reshape(meta.shape[-2], 256))               # [L tokens][256 channels]
        dev = m2.device
        heads = torch.empty((256, HEADS_COLS), dtype=torch.float32, device=dev)
        evec = torch.empty((6, 256), dtype=torch.float32, device=dev)
        problems, off = [], 0
        for w, b in zip(hw, hb):                                 # heads[c][off + j] = sum_tok meta[tok][c] W[j][tok] + b[j]
            n_k = w.shape[0]
            q = _problem(256, n_k, 256, [(m2, 256, w, 256)], heads, HEADS_COLS, 1, 1, bias=b)
            q.C = heads.data_ptr() + off * 4
            problems.append(q)
            off += n_k
        for k, (w, b) in enumerate(zip(fw, fb)):                 # evec[k] = fore_h_fc_k(pe_h)
            q = _problem(1, 256, 192, [(pe_h, 192, w, 192)], evec, 256, 0, 1, bias=b)
            q.C = evec.data_ptr() + k * 256 * 4
            problems.append(q)
        _launch(problems)
        ctx.save_for_backward(m2, pe_h, *hw)
        ctx.meta_shape = meta.shape
        return heads, evec

    @staticmethod
    def backward(ctx, g_heads, g_evec):
        from .linear import _launch, _problem
        m2, pe_h, *hw = ctx.saved_tensors
        dev = m2.device
        gh, ge = _c(g_heads), _c(g_evec)
        lib = L.load()
        d_meta = torch.empty(m2.shape, dtype=torch.float32, device=dev)
        terms, off = [], 0
        offs = []
        for w in hw:                                             # d_meta[tok][c] = sum_k sum_j W_k[j][tok] g[c][off_k + j]
            n_k = w.shape[0]
            terms.append((w, 256, gh, HEADS_COLS, n_k, off))
            offs.append(off)
            off += n_k
        # a 12-term problem would walk 24 k-tiles in sequence: four 3-term problems run side by side and dpn_sum_parts joins them
        parts = torch.empty((4, 256, 256), dtype=torch.float32, device=dev)
        problems = []
        for p_ in range(4):
            grp = terms[3 * p_:3 * p_ + 3]
            q0 = _problem(256, 256, 256, [t[:5] for t in grp], parts, 256, 1, 1)
            q0.C = parts.data_ptr() + p_ * 256 * 256 * 4
            for i, t in enumerate(grp):
                q0.B[i] = gh.data_ptr() + t[5] * 4
            problems.append(q0)
        dws, dbs = [], []
        for w, o in zip(hw, offs):                               # dW_k[j][tok] = sum_c g[c][off + j] meta[tok][c] ; db_k[j] = sum_c g[c][off + j]
            n_k = w.shape[0]
            dw = torch.empty((n_k, 256), dtype=torch.float32, device=dev)
            db = torch.empty((n_k,), dtype=torch.float32, device=dev)
            q = _problem(n_k, 256, 256, [(gh, HEADS_COLS, m2, 256)], dw, 256, 1, 1, asum=db)
            q.A[0] = gh.data_ptr() + o * 4
            problems.append(q)
            dws.append(dw)
            dbs.append(db)
        dfw = []
        for k in range(6):                                       # d fore_h_fc_k.weight = g_evec[k] (outer) pe_h ; bias gradient = g_evec[k]
            dw = torch.empty((256, 192), dtype=torch.float32, device=dev)
            q = _problem(256, 192, 1, [(ge, 256, pe_h, 192)], dw, 192, 1, 0)
            q.A[0] = ge.data_ptr() + k * 256 * 4
            problems.append(q)
            dfw.append(dw)
        _launch(problems)
        n_tail = (m2.shape[0] - 256) * 256                       # tokens >= 256 feed no VariableNet: their gradient rows are zero
        L.check(lib.dpn_sum_parts(_p(parts), 4, 256 * 256, n_tail, _p(d_meta), _s()), 'dpn_sum_parts')
        return (d_meta.view(ctx.meta_shape), None, *dws, *dbs, *dfw, *[ge[k] for k in range(6)])
