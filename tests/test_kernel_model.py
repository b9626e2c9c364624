"""CPU: the restructured algorithm the HIP kernels implement (oracle/kernel_model.py) is
mathematically identical to the reference-faithful autograd oracle (fp64, 1e-9), and the bf16
operand-rounding emulation stays inside the tolerances DESIGN.md quotes."""
import numpy as np
import pytest
import torch

from oracle import dpn_oracle as O
from oracle import kernel_model as KM
from oracle.fill import synthetic_inputs

GEO = O.Geometry()


def full_param_grads_from_kernel_grads(state, meta_out, fore_h, kgrads):
    """Chain the per-net kernel-level gradients through the hyper-network heads and the encoder with autograd,
    exactly as deepphysinet_amd does on the GPU."""
    outs, gouts = [], []
    direct = {}
    for k, net in enumerate(O.NETS):
        w1b1, w2b2, e = KM.hyper_weights(state, net, meta_out, fore_h)
        outs += [w1b1, w2b2, e]
        gouts += [kgrads[k]['w1b1'], kgrads[k]['w2b2'], kgrads[k]['e']]
        direct[net + '.data_input_fc.weight'] = kgrads[k]['Wd']
        direct[net + '.data_input_fc.bias'] = kgrads[k]['bd']
        direct[net + '.cat_fc1.fc.0.weight'] = kgrads[k]['W1']
        direct[net + '.cat_fc1.fc.0.bias'] = kgrads[k]['bf1']
        direct[net + '.cat_fc1.fc.2.weight'] = kgrads[k]['W2']
        direct[net + '.cat_fc1.fc.2.bias'] = kgrads[k]['bf2']
        direct[net + '.out_fc.weight'] = kgrads[k]['wo'][None, :]
        direct[net + '.out_fc.bias'] = kgrads[k]['bo'].reshape(1)
    names = [n for n in O.param_names(state) if n not in direct]
    gs = torch.autograd.grad(outs, [state[n] for n in names], grad_outputs=gouts, allow_unused=True)
    full = dict(direct)
    for n, g in zip(names, gs):
        full[n] = g if g is not None else torch.zeros_like(state[n])
    return full


@pytest.mark.parametrize('fused', [True, False])          # the five-GEMM algebra of the tile-split kernels / the seven GEMMs of the ring kernels
@pytest.mark.parametrize('with_clip,gain', [(True, 1.0), (False, 1.0), (True, 5.0)])
def test_restructured_equals_autograd_fp64(with_clip, gain, fused):
    dt = torch.float64
    st = O.make_state(dtype=dt, requires_grad=True, gain=gain)
    inp = {k: v.to(dt) for k, v in synthetic_inputs(96, tag='inter').items()}
    x, y, t = (inp[k].clone().requires_grad_(True) for k in ('x', 'y', 't'))
    total, parts, fn, ph = O.place_one_batch(st, x, y, t, inp['f'], inp['field_data'], inp['coord_data'], inp['forecast_h'],
                                             GEO, with_clip=with_clip, return_parts=True)
    jac = O.jacobian_fields(x, y, t, ph)
    res = KM.pde_step(st, inp['x'], inp['y'], inp['t'], inp['f'], inp['field_data'], inp['coord_data'], inp['forecast_h'],
                      GEO, with_clip=with_clip, prec='fp64', fused=fused)
    assert torch.allclose(res['out_n'], torch.cat(fn, 1).detach(), rtol=1e-10, atol=1e-10)
    assert torch.allclose(res['jac_phys'], jac.detach(), rtol=1e-8, atol=1e-18)
    mine = res['losses'].numpy()
    ref = np.array([float(p.detach()) for p in parts])
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(mine), ok)
    assert np.all(np.abs(mine[ok] - ref[ok]) <= 2e-7 * np.abs(ref[ok]))     # reference .float()s each scalar
    if not np.all(ok):
        return
    names = O.param_names(st)
    ref_g = dict(zip(names, torch.autograd.grad(total, [st[n] for n in names])))
    meta_out = O.meta_net_forward(st, inp['field_data'], inp['forecast_h'])
    my_g = full_param_grads_from_kernel_grads(st, meta_out, inp['forecast_h'], res['grads'])
    for n in names:
        denom = float(ref_g[n].abs().max()) + 1e-30
        err = float((my_g[n] - ref_g[n]).abs().max()) / denom
        if n.endswith('key_projection.bias'):
            continue
        assert err < 5e-6, (n, err)      # total loss carries the reference's fp32 casts of each scalar


@pytest.mark.parametrize('prec,tol_loss,tol_field', [('fp32', 1e-4, 1e-5), ('bf16x2', 2e-4, 2e-5), ('bf16', 8e-2, 3e-2)])
def test_precision_emulation(prec, tol_loss, tol_field):
    """What bf16 MFMA operands cost, measured on CPU against the fp64 oracle (documents DESIGN.md's tolerances)."""
    st64 = O.make_state(dtype=torch.float64)
    inp = synthetic_inputs(512, tag='inter')
    i64 = {k: v.double() for k, v in inp.items()}
    ref = KM.pde_step(st64, i64['x'], i64['y'], i64['t'], i64['f'], i64['field_data'], i64['coord_data'], i64['forecast_h'], GEO, prec='fp64')
    st = O.make_state()
    res = KM.pde_step(st, inp['x'], inp['y'], inp['t'], inp['f'], inp['field_data'], inp['coord_data'], inp['forecast_h'], GEO, prec=prec)
    rel = (res['losses'].double() - ref['losses']).abs() / ref['losses'].abs()
    ferr = (res['out_n'].double() - ref['out_n']).abs().max() / ref['out_n'].abs().max()
    print(prec, 'loss rel err', rel.numpy(), 'field err', float(ferr))
    assert float(rel.max()) < tol_loss
    assert float(ferr) < tol_field
