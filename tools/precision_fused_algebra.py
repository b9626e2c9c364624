#!/usr/bin/env python
"""CPU study (round 5): the point forward with the two static-times-hyper matrix products formed ONCE per net,
    A = W1 w2  [256, 256],   B = W1 Wd  [256, 192]     (W1 = cat_fc1.fc.0.weight, w2 = the hyper-network's hidden matrix, Wd = data_input_fc.weight)
so that   pre2 = A h1 + B pe6 + (W1 cvec + bf1),   wo . c = (w2^T wo) . h1 + (Wd^T wo) . pe6 + wo . cvec,   y = A^T (m2 (.) u) + 2 w2^T wo
-- five GEMMs per point and net (L1, A, B, A^T, gpe) instead of seven (L1, L2, Wd, fc1, v, y, gpe): c and v are never formed.
Same losses?  Each algebra is run with the MFMA operand rounding emulated (hi+lo bf16 = the product's mode) and compared with the fp64 run of
the seven-GEMM algebra.  TEST / DESIGN INFRASTRUCTURE: imports oracle/, never imported by the product.

    python tools/precision_fused_algebra.py [--points 2048] [--default-init]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dpn_oracle as O          # noqa: E402
from oracle import kernel_model as KM       # noqa: E402
from oracle.fill import synthetic_inputs    # noqa: E402
from tools.precision_table import mm_fmt    # noqa: E402


def phase_a_fused(W, pe, dpe, pe6, ref, fmt):
    w1, b1, w2, b2 = W['w1b1'][:, :192], W['w1b1'][:, 192], W['w2b2'][:, :256], W['w2b2'][:, 256]
    mm = lambda a, b: mm_fmt(a, b, fmt)
    A = W['W1'] @ w2                     # exact (fp32 MFMA GEMM once per net and step)
    B = W['W1'] @ W['Wd']
    cvec = b2 + W['bd'] + W['e']
    c2 = W['W1'] @ cvec + W['bf1']
    a2 = w2.T @ W['wo']
    bv = W['Wd'].T @ W['wo']
    pre1 = mm(pe, w1.T) + b1
    m1 = (pre1 > 0).to(pe.dtype)
    h1 = pre1 * m1
    pre2 = mm(h1, A.T) + mm(pe6, B.T) + c2
    m2 = (pre2 > 0).to(pe.dtype)
    u = W['W2'].T @ W['wo']
    out = (pre2 * m2) @ u + 2.0 * (h1 @ a2 + pe6 @ bv + W['wo'] @ cvec) + (W['wo'] @ W['bf2'] + W['bo']) + ref
    t2 = m2 * u
    y = mm(t2, A) + 2.0 * a2
    t1 = m1 * y
    gpe = mm(t1, w1)
    jxi = (gpe.reshape(pe.shape[0], 32, 2, 3) * dpe).sum(dim=(1, 2))
    return out, jxi, dict(m1=m1, m2=m2)


def phase_a_seven(W, pe, dpe, pe6, ref, fmt):
    old = KM.mm
    KM.mm = lambda a, b, prec: mm_fmt(a, b, fmt)
    try:
        return KM.phase_a(W, pe, dpe, pe6, ref, 'x')
    finally:
        KM.mm = old


def run(state, inp, geo, meta_out, phase, fmt):
    with torch.no_grad():
        dt = inp['x'].dtype
        scale = torch.tensor([1.0 / geo.dx / (geo.lon - 1), 1.0 / geo.dy / (geo.lat - 1), 1.0 / geo.pred_t_span], dtype=dt)
        xi = torch.cat([inp['x'] / geo.dx / (geo.lon - 1), inp['y'] / geo.dy / (geo.lat - 1), inp['t'] / geo.pred_t_span], 1)
        pe, dpe = KM.pe_and_tangent(xi)
        pe6 = O.sine_cos_pe(inp['coord_data'], 16)
        outs, jxis, masks = [], [], []
        for k, net in enumerate(O.NETS):
            W = KM.net_weights(state, net, meta_out, inp['forecast_h'])
            out, jxi, S = phase(W, pe, dpe, pe6, inp['coord_data'][:, k], fmt)
            outs.append(out), jxis.append(jxi), masks.append((S['m1'] > 0, S['m2'] > 0))
        out_n = torch.stack(outs, 1)
        jn = torch.stack(jxis, 1) * scale
        losses = KM.residuals(out_n, jn, inp['f'])[0]
    return losses, out_n, jn, masks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=2048)
    ap.add_argument('--default-init', action='store_true')
    args = ap.parse_args()
    torch.set_num_threads(8)
    geo = O.Geometry()
    inp = synthetic_inputs(args.points, tag='inter')
    if args.default_init:
        from deepphysinet_amd.configs import ncep_config
        from deepphysinet_amd.interface import builder_models
        torch.manual_seed(1)
        m = builder_models(**ncep_config())
        st = {k: v.detach().clone() for k, v in m.physics_net.state_dict().items()}
    else:
        st = O.make_state()
    st64 = {k: v.double() for k, v in st.items()}
    i64 = {k: v.double() for k, v in inp.items()}
    meta64 = O.meta_net_forward(st64, i64['field_data'], i64['forecast_h'])
    meta32 = O.meta_net_forward(st, inp['field_data'], inp['forecast_h'])
    ref, ref_out, ref_j, ref_m = run(st64, i64, geo, meta64, phase_a_seven, 'fp64')
    chk, chk_out, _, _ = run(st64, i64, geo, meta64, phase_a_fused, 'fp64')
    print('# %d points, %s init; fp64: five-GEMM algebra against seven-GEMM algebra: losses %.1e, fields %.1e' %
          (args.points, 'default' if args.default_init else 'closed-form', float(((chk - ref).abs() / ref.abs()).max()),
           float((chk_out - ref_out).abs().max() / ref_out.abs().max())))
    for label, phase in (('seven GEMMs (rounds 1-4)', phase_a_seven), ('five GEMMs (A = W1 w2, B = W1 Wd)', phase_a_fused)):
        for fmt in ('fp32', 'bf16x2', 'bf16'):
            l, o, j, mk = run(st, inp, geo, meta32, phase, fmt)
            rel = ((l.double() - ref).abs() / ref.abs()).numpy()
            flips = sum(int((a[0] != b[0]).any(dim=1).sum() + 0) for a, b in zip(mk, ref_m)), sum(int((a[1] != b[1]).any(dim=1).sum()) for a, b in zip(mk, ref_m))
            print('%-36s %-7s worst %.1e  losses %s  field %.1e  jac %.1e  (point, net) pairs with a flipped m1 / m2 bit %d / %d' %
                  (label, fmt, rel.max(), ' '.join('%.1e' % r for r in rel), float((o.double() - ref_out).abs().max() / ref_out.abs().max()),
                   float((j.double() - ref_j).abs().max() / ref_j.abs().max()), flips[0], flips[1]), flush=True)


if __name__ == '__main__':
    main()
