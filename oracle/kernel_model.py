"""TEST INFRASTRUCTURE -- a PyTorch model of the *restructured* algorithm the HIP kernels run.

The HIP path (deepphysinet_amd/csrc) does not replay the reference's 28
autograd.grad calls.  It uses the structure of the VariableNet (scalar output,
piecewise-linear in the coordinate features) to cut the work ~3.5x:

  * the 6x3 Jacobian is taken by ONE reverse sweep per net (v, y, gpe) instead of 3 tangents;
  * cat_fc1.fc.2 (W2) is only ever seen through u = W2^T w_out: no 256x256 GEMM with it, fwd or bwd;
  * in the backward pass the four cotangent streams (value + 3 tangents) are scalar multiples of
    the same per-point vectors, so every weight gradient is ONE points-reduction GEMM.

This file states that algorithm with plain torch ops so that tests can (1) prove it
equal to the autograd oracle (oracle/dpn_oracle.py) in fp64 and (2) emulate the bf16
MFMA operand rounding on CPU to set the tolerances quoted in DESIGN.md.  DESIGN.md
section 3 derives the formulas; reference lines: model/variable_net.py:49-87,
interface/interface_physics.py:90-185,232-262,322-332.
"""
import torch

from . import dpn_oracle as O

C_P, L_V, R_V, R_D = 1005.0, 2.5e6, 461.5, 287.0


def _split(x, prec):
    if prec in ('fp32', 'fp64'):
        return [x]
    hi = x.to(torch.bfloat16).to(x.dtype)
    if prec == 'bf16':
        return [hi]
    lo = (x - hi).to(torch.bfloat16).to(x.dtype)
    if prec == 'bf16x2':
        return [hi, lo]
    lo2 = (x - hi - lo).to(torch.bfloat16).to(x.dtype)
    return [hi, lo, lo2]


def mm(a, b, prec):
    """a @ b with MFMA-style operand rounding: bf16 operands (optionally hi/lo split), wide accumulate."""
    if prec in ('fp32', 'fp64'):
        return a @ b
    sa, sb = _split(a, prec), _split(b, prec)
    out = 0
    order = len(sa)
    for i in range(order):
        for j in range(order - i):
            out = out + sa[i] @ sb[j]
    return out


def pe_and_tangent(xi, n_freqs=32):
    """SineCosPE value [N, 2*F*C] and d/d(xi_c) (same layout; non-zero only on channel c)."""
    freq = (2.0 ** torch.linspace(0.0, 4.0, steps=n_freqs)).to(xi.dtype)
    ang = xi[:, None, :] * freq[None, :, None]                       # [N,F,C]
    s, c = torch.sin(ang), torch.cos(ang)
    pe = torch.stack([s, c], 2).reshape(xi.shape[0], -1)             # [N,F,2,C] -> flat f*2C + fn*C + c
    dpe = torch.stack([c * freq[None, :, None], -s * freq[None, :, None]], 2)   # [N,F,2,C]: d/d xi_c of channel (f,fn,c)
    return pe, dpe


def hyper_weights(state, net, meta_out, fore_h):
    """Per-field tensors the point kernels consume (model/variable_net.py:57-65,75-78)."""
    p = lambda k: state[net + '.' + k]
    m = meta_out[0, :256]
    w1b1 = torch.nn.functional.linear(m.T, p('coord_input_fc.weight'), p('coord_input_fc.bias'))     # [256,193]
    w2b2 = torch.nn.functional.linear(m.T, p('coord_hidden_fc.weight'), p('coord_hidden_fc.bias'))   # [256,257]
    e = torch.nn.functional.linear(O.sine_cos_pe(fore_h.squeeze(-1), 96), p('fore_h_fc.weight'), p('fore_h_fc.bias'))[0]
    return w1b1, w2b2, e


def phase_a(W, pe, dpe, pe6, ref, prec, fused=False):
    """fwd value + reverse sweep.  Returns normalised out [N], J_xi [N,3] and the saved per-point state.

    fused (round 5, csrc/dpn_fwd_tiles.h + dpn_layout.h): W1 = cat_fc1.fc.0.weight only ever multiplies c = w2 h1 + Wd pe6 + cvec and W1^T only
    ever meets w2^T on the way back, so with A = W1 w2 and B = W1 Wd formed once per net (exact)
        pre2 = A h1 + B pe6 + (W1 cvec + bf1),   wo . c = (w2^T wo) . h1 + (Wd^T wo) . pe6 + wo . cvec,   y = A^T (m2 (.) u) + 2 w2^T wo:
    five GEMMs per point and net instead of seven; c and v = d out / d c are never formed (the backward pass needs neither: phase_b)."""
    w1, b1, w2, b2 = W['w1b1'][:, :192], W['w1b1'][:, 192], W['w2b2'][:, :256], W['w2b2'][:, 256]
    pre1 = mm(pe, w1.T, prec) + b1
    m1 = (pre1 > 0).to(pe.dtype)
    h1 = pre1 * m1
    cvec = b2 + W['bd'] + W['e']
    u = W['W2'].T @ W['wo']                                            # [256]
    if fused:
        A, B = W['W1'] @ w2, W['W1'] @ W['Wd']
        a2 = w2.T @ W['wo']
        pre2 = mm(h1, A.T, prec) + mm(pe6, B.T, prec) + (W['W1'] @ cvec + W['bf1'])
        m2 = (pre2 > 0).to(pe.dtype)
        out = (pre2 * m2) @ u + 2.0 * (h1 @ a2 + pe6 @ (W['Wd'].T @ W['wo']) + W['wo'] @ cvec) + (W['wo'] @ W['bf2'] + W['bo']) + ref
        y = mm(m2 * u, A, prec) + 2.0 * a2
        v = None
    else:
        c = mm(h1, w2.T, prec) + mm(pe6, W['Wd'].T, prec) + cvec
        pre2 = mm(c, W['W1'].T, prec) + W['bf1']
        m2 = (pre2 > 0).to(pe.dtype)
        a = pre2 * m2
        out = a @ u + 2.0 * (c @ W['wo']) + (W['wo'] @ W['bf2'] + W['bo']) + ref
        t2 = m2 * u
        v = mm(t2, W['W1'], prec) + 2.0 * W['wo']
        y = mm(v, w2, prec)
    t1 = m1 * y
    gpe = mm(t1, w1, prec)                                             # [N,192]
    jxi = (gpe.reshape(pe.shape[0], 32, 2, 3) * dpe).sum(dim=(1, 2))   # [N,3]
    return out, jxi, dict(m1=m1, m2=m2, v=v, t1=t1, u=u)


def residuals(out_n, jn, f, with_clip=True, factors=O.LOSS_FACTOR):
    """De-norm + clip + the six residual losses (mean over points) and their analytic gradients w.r.t. the
    normalised outputs out_n [N,6] and the normalised Jacobian jn [N,6,3] (d out_n / d(x,y,t))."""
    dt = out_n.dtype
    N = out_n.shape[0]
    std = torch.tensor(O.OBS_STD, dtype=dt)
    mean = torch.tensor(O.OBS_MEAN, dtype=dt)
    val = out_n * std + mean
    mask = torch.ones_like(val)
    if with_clip:
        for k in range(2, 6):
            lo, hi = O.CLIP_LO[k], O.CLIP_HI[k]
            mask[:, k] = ((val[:, k] >= lo) & (val[:, k] <= hi)).to(dt)
            val[:, k] = val[:, k].clamp(lo, hi)
    J = jn * (std * mask)[:, :, None] if mask.dim() == 2 else None
    u, v, p, T, q, rho = [val[:, k] for k in range(6)]
    (u_x, u_y, u_t), (v_x, v_y, v_t), (p_x, p_y, p_t) = J[:, 0].unbind(1), J[:, 1].unbind(1), J[:, 2].unbind(1)
    (T_x, T_y, T_t), (q_x, q_y, q_t), (r_x, r_y, r_t) = J[:, 3].unbind(1), J[:, 4].unbind(1), J[:, 5].unbind(1)
    f = f.reshape(-1)
    eps = 1e-6
    omega = p_t + u * p_x + v * p_y
    A = T_t + u * T_x + v * T_y
    B = q_t + u * q_x + v * q_y
    tc = T - 273.15
    e_s = 6.112 * torch.exp(17.67 * tc / (tc + 243.5)) * 100
    q_s = torch.clamp(0.622 * e_s / (p - 0.378 * e_s), min=1e-6)
    delta = ((omega < 0) & (q >= q_s)).to(dt)
    R = (1 + 0.608 * q) * R_D
    Fv = (L_V * R - C_P * R_V * T) / (C_P * R_V + T * T + L_V * L_V * q_s) * q_s * T
    K = delta * Fv / (p + eps)
    r = [u_t + u * u_x + v * u_y + p_x / rho - f * v,
         v_t + u * v_x + v * v_y + p_y / rho + f * u,
         r_t + u * r_x + v * r_y + rho * u_x + rho * v_y,
         C_P * A - omega / (rho + eps) + L_V * B,
         -omega * K + B,
         p - rho * (1 + 0.608 * q) * R_D * T]
    fac = [factors['motion_u_factor'], factors['motion_v_factor'], factors['continuous_factor'],
           factors['energy_factor'], factors['vapor_factor'], factors['gas_factor']]
    losses = torch.stack([fac[i] * (r[i] ** 2).mean() for i in range(6)])
    g = [2.0 * fac[i] * r[i] / N for i in range(6)]                     # dL/dr_i
    z = torch.zeros_like(u)
    gval = torch.stack([
        g[0] * u_x + g[1] * (v_x + f) + g[2] * r_x + g[3] * (C_P * T_x - p_x / (rho + eps) + L_V * q_x) + g[4] * (-p_x * K + q_x),
        g[0] * (u_y - f) + g[1] * v_y + g[2] * r_y + g[3] * (C_P * T_y - p_y / (rho + eps) + L_V * q_y) + g[4] * (-p_y * K + q_y),
        g[4] * omega * delta * Fv / (p + eps) ** 2 + g[5],
        -g[5] * rho * (1 + 0.608 * q) * R_D,
        -g[5] * rho * 0.608 * R_D * T,
        -g[0] * p_x / rho ** 2 - g[1] * p_y / rho ** 2 + g[2] * (u_x + v_y) + g[3] * omega / (rho + eps) ** 2
        - g[5] * (1 + 0.608 * q) * R_D * T], dim=1)
    gJ = torch.stack([
        torch.stack([g[0] * u + g[2] * rho, g[0] * v, g[0]], 1),                                   # u_x,u_y,u_t
        torch.stack([g[1] * u, g[1] * v + g[2] * rho, g[1]], 1),                                   # v_*
        torch.stack([g[0] / rho - g[3] * u / (rho + eps) - g[4] * u * K,
                     g[1] / rho - g[3] * v / (rho + eps) - g[4] * v * K,
                     -g[3] / (rho + eps) - g[4] * K], 1),                                          # p_*
        torch.stack([g[3] * C_P * u, g[3] * C_P * v, g[3] * C_P + z], 1),                          # T_*
        torch.stack([(g[3] * L_V + g[4]) * u, (g[3] * L_V + g[4]) * v, g[3] * L_V + g[4]], 1),     # q_*
        torch.stack([g[2] * u, g[2] * v, g[2]], 1)], dim=1)                                        # rho_*
    sm = std * mask
    return losses, gval * sm, gJ * sm[:, :, None], val, J


def phase_b(W, S, pe, dpe, pe6, gout, gjxi, prec):
    """Parameter gradients of one net from per-point cotangents gout [N] (on out) and gjxi [N,3] (on J_xi).

    Round 5: every product with the second ReLU mask on the X side factors through the mask-side sums
        S1 = M2^T Z1 [256, 256],  S2 = M2^T (g pe6) [256, 192],  mvec = M2^T g,   q1 = colsum(Z1), q6 = colsum(g pe6), sg = sum g
    (the only points-reduction GEMMs left besides dw1 = T1^T Z0):
      * Z = Z1 w2^T + (g pe6) Wd^T + g cvec^T is LINEAR in (Z1, g pe6, g), so G = M2^T Z = S1 w2^T + S2 Wd^T + mvec (x) cvec and
        colsum(Z) = q1 w2^T + q6 Wd^T + sg cvec: neither Z nor the product M2^T Z is ever formed per point (csrc: dpn_finish_gside);
      * v = W1^T (m2 (.) u) + 2 wo is affine in the mask (round 3), so V^T Z1 = W1^T diag(u) S1 + 2 wo (x) q1, likewise with g pe6."""
    w1, b1, w2 = W['w1b1'][:, :192], W['w1b1'][:, 192], W['w2b2'][:, :256]
    N = pe.shape[0]
    pt = (dpe * gjxi[:, None, None, :]).reshape(N, -1)                 # sum_c gJ_c * dpe_c  (disjoint channels)
    Z0 = gout[:, None] * pe + pt
    Z1 = S['m1'] * (mm(Z0, w1.T, prec) + gout[:, None] * b1)
    gpe6 = gout[:, None] * pe6
    cvec = W['w2b2'][:, 256] + W['bd'] + W['e']
    S1 = mm(S['m2'].T, Z1, prec)                                        # [256 o, 256 j]
    S2 = mm(S['m2'].T, gpe6, prec)                                      # [256 o, 192 k]
    mvec = S['m2'].T @ gout
    q1, q6, sg = Z1.sum(0), gpe6.sum(0), gout.sum()
    G = S1 @ w2.T + S2 @ W['Wd'].T + mvec[:, None] * cvec[None, :]      # = M2^T Z, exact-fp32 GEMM once per net
    zsum = w2 @ q1 + W['Wd'] @ q6 + sg * cvec                           # = colsum(Z)
    u = S['u']
    r = (W['W1'] * G).sum(1) + W['bf1'] * mvec
    W1u = W['W1'].T * u[None, :]                                        # W1^T diag(u)
    gcvec = W1u @ mvec + 2.0 * W['wo'] * sg                             # = V^T g
    grads = {
        'W1': u[:, None] * G, 'bf1': u * mvec,
        'W2': W['wo'][:, None] * r[None, :], 'bf2': W['wo'] * sg,
        'wo': W['W2'] @ r + W['bf2'] * sg + 2.0 * zsum, 'bo': sg,
        'w1b1': torch.cat([mm(S['t1'].T, Z0, prec), (S['t1'].T @ gout)[:, None]], 1),
        'w2b2': torch.cat([W1u @ S1 + 2.0 * W['wo'][:, None] * q1[None, :], gcvec[:, None]], 1),
        'Wd': W1u @ S2 + 2.0 * W['wo'][:, None] * q6[None, :], 'bd': gcvec, 'e': gcvec,
    }
    return grads


def net_weights(state, net, meta_out, fore_h):
    p = lambda k: state[net + '.' + k]
    w1b1, w2b2, e = hyper_weights(state, net, meta_out, fore_h)
    return dict(w1b1=w1b1, w2b2=w2b2, e=e, Wd=p('data_input_fc.weight'), bd=p('data_input_fc.bias'),
                W1=p('cat_fc1.fc.0.weight'), bf1=p('cat_fc1.fc.0.bias'), W2=p('cat_fc1.fc.2.weight'),
                bf2=p('cat_fc1.fc.2.bias'), wo=p('out_fc.weight')[0], bo=p('out_fc.bias')[0])


def pde_step(state, x, y, t, f, field, coord_data, forecast_h, geo, with_clip=True, prec='fp32', meta_out=None, fused=True):
    """The whole restructured place_one_batch + backward to the per-net kernel-level gradients."""
    dt = x.dtype
    with torch.no_grad():
        if meta_out is None:
            meta_out = O.meta_net_forward(state, field, forecast_h)
        scale = torch.tensor([1.0 / geo.dx / (geo.lon - 1), 1.0 / geo.dy / (geo.lat - 1), 1.0 / geo.pred_t_span], dtype=dt)
        xi = torch.cat([x / geo.dx / (geo.lon - 1), y / geo.dy / (geo.lat - 1), t / geo.pred_t_span], 1)
        pe, dpe = pe_and_tangent(xi)
        pe6 = O.sine_cos_pe(coord_data, 16)
        Ws, Ss, outs, jxis = [], [], [], []
        for k, net in enumerate(O.NETS):
            W = net_weights(state, net, meta_out, forecast_h)
            out, jxi, S = phase_a(W, pe, dpe, pe6, coord_data[:, k], prec, fused=fused)
            Ws.append(W), Ss.append(S), outs.append(out), jxis.append(jxi)
        out_n = torch.stack(outs, 1)
        jn = torch.stack(jxis, 1) * scale
        losses, gout, gjn, val, J = residuals(out_n, jn, f, with_clip=with_clip)
        grads = [phase_b(Ws[k], Ss[k], pe, dpe, pe6, gout[:, k], gjn[:, k] * scale, prec) for k in range(6)]
    return dict(losses=losses, out_n=out_n, jac_phys=J, val=val, grads=grads, meta_out=meta_out)
