#!/usr/bin/env python
"""CPU study: which operands of the four weight-gradient products need the hi+lo split for the 1e-3 gradient bar?

Forward and the cotangent chain (Z0 -> Z1 -> Z) stay in the bf16x2 mode of the product; only the points-reduction GEMMs
(G = M2^T Z, dw1 = T1^T Z0, dw2 = V^T Z1, dWd = V^T G6; oracle/kernel_model.py phase_b) change format:
    x2  : both operands split, 3 MFMAs (the product today)
    a2  : saved activation (X) split, cotangent (Y) single bf16, 2 MFMAs, half the Y bytes
    w2  : X single, Y split
    1   : both single
The gradients are compared with the fp64 run per tensor, max-abs error over max-abs value (the criterion of tests/test_gpu_parity.py).
TEST / DESIGN INFRASTRUCTURE: imports oracle/, never imported by the product.

    python tools/precision_wgrad.py [--points 4096] [--default-init]
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

PHASE_B_SITES = ('Z1', 'Za', 'Zb', 'G', 'dw1', 'dw2', 'dWd')       # program order of phase_b's mm calls


def run(state, inp, geo, meta_out, fmt_chain, fmt_wgrad):
    calls = {'n': 0}

    def mm(a, b, prec):
        k = calls['n']
        calls['n'] += 1
        if k < 42:                                   # phase_a: 7 sites x 6 nets
            return mm_fmt(a, b, fmt_chain)
        site = PHASE_B_SITES[(k - 42) % 7]
        return mm_fmt(a, b, fmt_wgrad.get(site, fmt_chain) if isinstance(fmt_wgrad, dict) else (fmt_wgrad if site in ('G', 'dw1', 'dw2', 'dWd') else fmt_chain))
    old = KM.mm
    KM.mm = mm
    try:
        r = KM.pde_step(state, inp['x'], inp['y'], inp['t'], inp['f'], inp['field_data'], inp['coord_data'], inp['forecast_h'], geo,
                        prec='x', meta_out=meta_out)
    finally:
        KM.mm = old
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=4096)
    ap.add_argument('--default-init', action='store_true')
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
    ref = run(st64, i64, geo, meta64, 'fp64', 'fp64')

    def report(label, fmt_chain, fmt_wgrad):
        r = run(st, inp, geo, meta32, fmt_chain, fmt_wgrad)
        worst, where = 0.0, None
        per = {}
        for k in range(6):
            for name, g in r['grads'][k].items():
                g64 = ref['grads'][k][name]
                e = float((g.double() - g64).abs().max() / (g64.abs().max() + 1e-300))
                per[name] = max(per.get(name, 0.0), e)
                if e > worst:
                    worst, where = e, (O.NETS[k], name)
        print('%-40s worst %.1e at %-22s  ' % (label, worst, where) + ' '.join('%s %.0e' % (n, per[n]) for n in ('W1', 'w1b1', 'w2b2', 'Wd', 'W2', 'wo')), flush=True)

    print('# %d points, %s init; per-tensor max-abs error / max-abs value against fp64, worst over the six nets' %
          (args.points, 'default' if args.default_init else 'closed-form'))
    report('all fp32', 'fp32', 'fp32')
    report('chain bf16x2, products bf16x2 (today)', 'bf16x2', 'bf16x2')
    report('chain bf16x2, products X split / Y single', 'bf16x2', 'bf16a2')
    report('chain bf16x2, products X single / Y split', 'bf16x2', 'bf16w2')
    report('chain bf16x2, products single', 'bf16x2', 'bf16')
    for site in ('G', 'dw1', 'dw2', 'dWd'):
        report('only %s with Y single' % site, 'bf16x2', {site: 'bf16a2'})
    report('chain bf16, products bf16 (bf16 mode)', 'bf16', 'bf16')


if __name__ == '__main__':
    main()
