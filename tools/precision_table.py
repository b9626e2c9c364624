#!/usr/bin/env python
"""CPU study: which GEMMs of the point chain need which MFMA operand format for the 1e-4 PDE-loss bar?

Per GEMM site of oracle/kernel_model.py's phase_a (forward: L1, L2, Wd, fc1; reverse sweep: v, y, gpe) the operand
format is varied on its own and in the assignments worth building; the six loss scalars are compared with the fp64 run of the same
restructured algorithm.  Formats (cost in MFMAs per product):
    bf16 (1)   f16 (1)   bf16x2 (3: hi*hi + hi*lo + lo*hi)   f16x2 (3)
    f16a2 (2: activation split hi+lo, weight single f16)   f16w2 (2: weight split, activation single)
TEST / DESIGN INFRASTRUCTURE: imports oracle/, never imported by the product.

    python tools/precision_table.py [--points 2048] [--default-init]
"""
import argparse
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dpn_oracle as O          # noqa: E402
from oracle import kernel_model as KM       # noqa: E402
from oracle.fill import synthetic_inputs    # noqa: E402

SITES = ('L1', 'L2', 'Wd', 'fc1', 'v', 'y', 'gpe')
COST = {'fp32': 0, 'bf16': 1, 'f16': 1, 'bf16x2': 3, 'f16x2': 3, 'f16a2': 2, 'f16w2': 2, 'bf16a2': 2, 'bf16w2': 2}
# executed MACs per point per net at each site
MACS = {'L1': 192 * 256, 'L2': 256 * 256, 'Wd': 192 * 256, 'fc1': 256 * 256, 'v': 256 * 256, 'y': 256 * 256, 'gpe': 256 * 192}


def _rt(x, dt):
    return x.to(dt).to(x.dtype)


def mm_fmt(a, b, fmt):
    """a = activations [N,K], b = weights [K,M]."""
    if fmt in ('fp32', 'fp64'):
        return a @ b
    base = torch.float16 if fmt.startswith('f16') else torch.bfloat16
    ah, bh = _rt(a, base), _rt(b, base)
    if fmt in ('bf16', 'f16'):
        return ah @ bh
    al, bl = _rt(a - ah, base), _rt(b - bh, base)
    if fmt.endswith('x2'):
        return ah @ bh + (ah @ bl + al @ bh)
    if fmt.endswith('a2'):
        return ah @ bh + al @ bh
    if fmt.endswith('w2'):
        return ah @ bh + ah @ bl
    raise ValueError(fmt)


def run(state, inp, geo, assign, meta_out):
    """pde_step with a per-site format table (the sites are visited in phase_a's program order)."""
    order = iter(['L1', 'L2', 'Wd', 'fc1', 'v', 'y', 'gpe'] * 6)

    def mm(a, b, prec):
        return mm_fmt(a, b, assign[next(order)])
    old = KM.mm
    KM.mm = mm
    try:
        with torch.no_grad():
            dt = inp['x'].dtype
            scale = torch.tensor([1.0 / geo.dx / (geo.lon - 1), 1.0 / geo.dy / (geo.lat - 1), 1.0 / geo.pred_t_span], dtype=dt)
            xi = torch.cat([inp['x'] / geo.dx / (geo.lon - 1), inp['y'] / geo.dy / (geo.lat - 1), inp['t'] / geo.pred_t_span], 1)
            pe, dpe = KM.pe_and_tangent(xi)
            pe6 = O.sine_cos_pe(inp['coord_data'], 16)
            outs, jxis = [], []
            for k, net in enumerate(O.NETS):
                W = KM.net_weights(state, net, meta_out, inp['forecast_h'])
                out, jxi, _ = KM.phase_a(W, pe, dpe, pe6, inp['coord_data'][:, k], 'x')
                outs.append(out), jxis.append(jxi)
            out_n = torch.stack(outs, 1)
            jn = torch.stack(jxis, 1) * scale
            losses = KM.residuals(out_n, jn, inp['f'])[0]
    finally:
        KM.mm = old
    return losses, out_n, jn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=2048)
    ap.add_argument('--default-init', action='store_true', help='PyTorch default init instead of the closed-form fill')
    ap.add_argument('--tag', default='inter')
    args = ap.parse_args()
    torch.set_num_threads(8)
    geo = O.Geometry()
    inp = synthetic_inputs(args.points, tag=args.tag)
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
    ref, ref_out, ref_j = run(st64, i64, geo, {s: 'fp64' for s in SITES}, meta64)

    def report(label, assign):
        l, o, j = run(st, inp, geo, assign, meta32)
        rel = ((l.double() - ref).abs() / ref.abs()).numpy()
        fe = float((o.double() - ref_out).abs().max() / ref_out.abs().max())
        je = float((j.double() - ref_j).abs().max() / ref_j.abs().max())
        cost = sum(COST[assign[s]] * MACS[s] for s in SITES) / sum(MACS.values())
        print('%-44s cost %.2f  worst %.1e  losses %s  field %.1e  jac %.1e' % (label, cost, rel.max(), ' '.join('%.1e' % r for r in rel), fe, je), flush=True)
        return rel.max()

    print('# %d points, %s init; loss order motion_u motion_v continuous energy vapor gas; cost = MFMAs per product, MAC-weighted' %
          (args.points, 'default' if args.default_init else 'closed-form'))
    report('all fp32', {s: 'fp32' for s in SITES})
    for f in ('bf16', 'f16', 'bf16x2', 'f16x2', 'f16a2', 'f16w2'):
        report('all ' + f, {s: f for s in SITES})
    print('# one site lowered, the rest fp32')
    for f in ('bf16', 'f16', 'f16a2', 'f16w2'):
        for s in SITES:
            a = {k: 'fp32' for k in SITES}
            a[s] = f
            report('%s = %s' % (s, f), a)
    print('# forward sites F = (L1, L2, Wd, fc1), reverse sites R = (v, y, gpe)')
    fmts = ('f16', 'f16a2', 'f16w2', 'f16x2', 'bf16x2')
    for ff, fr in itertools.product(fmts, fmts):
        a = {s: ff for s in ('L1', 'L2', 'Wd', 'fc1')}
        a.update({s: fr for s in ('v', 'y', 'gpe')})
        report('F = %s, R = %s' % (ff, fr), a)


if __name__ == '__main__':
    main()
