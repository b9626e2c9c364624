"""CPU emulation: which split-operand matrix-core format keeps the grid encoder fp32-class?

Every 256-wide GEMM of the encoder layers (q/k/v/out projections, the two 1x1 convolutions, the output projection) is replaced by an
emulation of the operand rounding of a candidate MFMA format (fp32 accumulate), the rest of oracle.meta_net_forward is untouched; the
encoder output is compared with the fp64 run and with the fp32 run's own distance from fp64.

    f16x2  : x = hi + 2^-11 lo', hi = f16(x'), lo' = f16(2^11 (x' - hi)), x' = x * 2^-e(row) for activations (row max in [8, 16));
             three products hi.hi + 2^-11 (hi.lo' + lo'.hi)
    bf16x2 : hi + lo, three products            bf16x3 : hi + mid + lo, six products
Usage: python tools/precision_encoder_split.py
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import dpn_oracle as O                      # noqa: E402
from oracle.fill import synthetic_inputs               # noqa: E402


def _split(x, dt, parts, scale_lo):
    out, r = [], x
    for i in range(parts):
        p = r.to(dt).to(torch.float32)
        out.append(p)
        r = (r - p)
        if scale_lo and i == 0:
            r = r * 2048.0
    return out


def emu_linear(mode):
    def lin(x, w, b=None):
        x32, w32 = x.to(torch.float32), w.to(torch.float32)
        if mode == 'f16x2':
            amax = x32.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
            e = torch.floor(torch.log2(amax)) - 3.0
            xs = x32 * torch.exp2(-e)
            xh, xl = _split(xs, torch.float16, 2, True)
            wh, wl = _split(w32, torch.float16, 2, True)
            y = (xh @ wh.T + (xh @ wl.T + xl @ wh.T) * (1.0 / 2048.0)) * torch.exp2(e)
        elif mode == 'bf16x2':
            xh, xl = _split(x32, torch.bfloat16, 2, False)
            wh, wl = _split(w32, torch.bfloat16, 2, False)
            y = xh @ wh.T + (xh @ wl.T + xl @ wh.T)
        elif mode == 'bf16x3':
            xh, xm, xl = _split(x32, torch.bfloat16, 3, False)
            wh, wm, wl = _split(w32, torch.bfloat16, 3, False)
            y = xh @ wh.T + (xh @ wm.T + xm @ wh.T) + (xm @ wm.T + xh @ wl.T + xl @ wh.T)
        else:
            raise ValueError(mode)
        return y if b is None else y + b
    return lin


def run(state, inp, h, lin=None):
    real_lin, real_conv = F.linear, F.conv1d
    if lin is not None:
        def conv(x, w, b=None, *a, **k):
            if w.shape[-1] == 1:
                return lin(x.transpose(1, 2), w.squeeze(-1), b).transpose(1, 2)
            return real_conv(x, w, b, *a, **k)
        O.F.linear, O.F.conv1d = lin, conv
    try:
        with torch.no_grad():
            return O.meta_net_forward(state, inp['field_data'], torch.full((1, 1, 1), h / 360.0, dtype=inp['field_data'].dtype))
    finally:
        O.F.linear, O.F.conv1d = real_lin, real_conv


def main():
    st = O.make_state()
    st64 = {k: v.double() for k, v in st.items()}
    inp = synthetic_inputs(4)
    inp64 = dict(inp, field_data=inp['field_data'].double())
    for h in (0, 24, 336):
        ref = run(st64, inp64, h)
        base = run(st, inp, h)
        rel = lambda a: float((a.double() - ref).abs().max() / ref.abs().max())
        print('lead %3d h: fp32 torch vs fp64 %.2e' % (h, rel(base)), end='')
        for mode in ('f16x2', 'bf16x2', 'bf16x3'):
            y = run(st, inp, h, emu_linear(mode))
            print(' | %s vs fp64 %.2e, vs fp32 %.2e' % (mode, rel(y), float((y - base).abs().max() / base.abs().max())), end='')
        print()


if __name__ == '__main__':
    main()
