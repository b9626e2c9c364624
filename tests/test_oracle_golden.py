"""CPU: the oracle (oracle/dpn_oracle.py) against golden vectors captured from the reference
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU parity tests then
compare the HIP path with the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import dpn_oracle as O
from oracle.fill import synthetic_inputs

GEO = O.Geometry()


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope='module')
def state():
    return O.make_state()


def test_state_dict_contract(golden_dir, state):
    """156 names/shapes (the 155 parameters + the persistent `pe` buffer)."""
    d = _load(golden_dir, 'f0_state_names.npz')
    ref = {str(k): str(s) for k, s in zip(d['names'], d['shapes'])}
    assert len(ref) == 156
    mine = {k: str(tuple(v.shape)) for k, v in state.items()}
    pe = 'meta_net.model.enc_embedding.position_embedding.pe'
    assert ref.pop(pe) == '(1, 5000, 256)'
    assert mine == ref


def test_f1_position_encoding(golden_dir):
    d = _load(golden_dir, 'f1_pe.npz')
    t = torch.from_numpy
    assert np.array_equal(O.sine_cos_pe(t(d['in3']), 32).numpy(), d['pe3'])
    assert np.array_equal(O.sine_cos_pe(t(d['in6']), 16).numpy(), d['pe6'])
    assert np.array_equal(O.sine_cos_pe(t(d['in1']), 96).numpy(), d['pe1_96'])
    assert np.array_equal(O.sine_cos_pe(t(d['in1']).reshape(1, 1, 1), 128).numpy(), d['pe1_128'])
    enc = O.encoding_coord(t(d['x']), t(d['y']), t(d['t']), GEO).numpy()
    assert np.array_equal(enc, d['enc'])


def test_f2_encoder(golden_dir, state):
    d = _load(golden_dir, 'f2_encoder.npz')
    inp = synthetic_inputs(4)
    with torch.no_grad():
        for h in (0, 24, 336):
            mo = O.meta_net_forward(state, inp['field_data'], torch.full((1, 1, 1), h / 360.0)).numpy()
            assert mo.shape == (1, 287, 256)
            assert _rel(mo, d['meta_out_h%d' % h]) < 2e-6


def _run_pde(state, inputs, with_clip, dtype=torch.float32):
    st = {k: v.to(dtype) for k, v in state.items()}
    c = lambda v: v.to(dtype)
    x, y, t = (c(inputs[k]).clone().requires_grad_(True) for k in ('x', 'y', 't'))
    total, parts, fn, ph = O.place_one_batch(st, x, y, t, c(inputs['f']), c(inputs['field_data']), c(inputs['coord_data']),
                                             c(inputs['forecast_h']), GEO, with_clip=with_clip, return_parts=True)
    jac = O.jacobian_fields(x, y, t, ph)
    return total, parts, torch.cat(fn, 1), torch.cat(ph, 1), jac


@pytest.mark.parametrize('with_clip', [True, False])
def test_f345_fields_jacobian_residuals(golden_dir, state, with_clip):
    d = _load(golden_dir, 'f345_pde_clip%d_fp32.npz' % int(with_clip))
    total, parts, fn, ph, jac = _run_pde(state, synthetic_inputs(256, tag='inter'), with_clip)
    assert _rel(fn.detach().numpy(), d['fields_norm']) < 2e-6
    assert _rel(ph.detach().numpy(), d['fields_phys']) < 2e-6
    j, jr = jac.detach().numpy(), d['jac']
    assert np.array_equal(j == 0, jr == 0)              # identical clip masks
    for k in range(6):
        assert _rel(j[:, k], jr[:, k]) < 2e-5
    mine = np.array([float(p.detach()) for p in parts])
    ok = np.isfinite(d['parts'])
    assert np.array_equal(np.isfinite(mine), ok)
    assert np.abs(mine[ok] - d['parts'][ok]).max() / np.abs(d['parts'][ok]).max() < 1e-5
    assert np.all(np.abs(mine[ok] - d['parts'][ok]) <= 2e-5 * np.abs(d['parts'][ok]))
    if np.isfinite(d['total']):
        assert abs(float(total.detach()) - float(d['total'])) <= 1e-5 * abs(float(d['total']))


def _f10_inputs():
    inp = synthetic_inputs(200, tag='f10', margin=True, forecast_h=336.0 / 360.0)
    inp['x'][0, 0], inp['y'][0, 0] = 0.0, 0.0                                     # both corners of the domain (tests/golden/make_golden.py)
    inp['x'][1, 0], inp['y'][1, 0] = 256 * 27000.0, 144 * 27000.0
    return inp


def test_f10_grid_node_points_longest_lead(golden_dir, state):
    """Second, independent pin of the PDE path: the REFERENCE on 200 grid-node points (both domain corners included) at 336 h lead."""
    d = _load(golden_dir, 'f10_grid_nodes_h336_fp32.npz')
    inp = _f10_inputs()
    assert np.array_equal(inp['x'].numpy(), d['x']) and np.array_equal(inp['y'].numpy(), d['y'])
    total, parts, fn, ph, jac = _run_pde(state, inp, True)
    assert _rel(fn.detach().numpy(), d['fields_norm']) < 2e-6
    assert _rel(ph.detach().numpy(), d['fields_phys']) < 2e-6
    j, jr = jac.detach().numpy(), d['jac']
    assert np.array_equal(j == 0, jr == 0)
    for k in range(6):
        assert _rel(j[:, k], jr[:, k]) < 2e-5
    mine = np.array([float(p.detach()) for p in parts])
    assert np.all(np.abs(mine - d['parts']) <= 2e-5 * np.abs(d['parts']))
    assert abs(float(total.detach()) - float(d['total'])) <= 1e-5 * abs(float(d['total']))


F12_CASES = {'l1': dict(crit=('L1Loss', 0.0)), 'mse_sum': dict(crit=('MSELoss', 0.0, 'sum')), 'sl1': dict(crit=('WeightSmoothL1Loss', 0.1)), 'sl1_b2': dict(crit=('WeightSmoothL1Loss', 2.0)),
             'norm': dict(norm='f12_norm_cfg'), 'norm_sq': dict(norm='f12_norm_sq_cfg')}


@pytest.mark.parametrize('case', sorted(F12_CASES))
def test_f12_other_criteria_and_norm_branches(golden_dir, case):
    """The branches the shipped config does not take, against the REFERENCE run on them: the loss builder's other PDE criteria (L1Loss,
    WeightSmoothL1Loss(beta)) and inverse_norm's use_norm False / two- and three-factor min_max branches; six terms, total, all 155 gradient norms."""
    d = _load(golden_dir, 'f12_criteria_and_norm_branches.npz')
    c = F12_CASES[case]
    st = O.make_state(requires_grad=True)
    inp = synthetic_inputs(256, tag='inter')
    x, y, t = (inp[k].clone().requires_grad_(True) for k in ('x', 'y', 't'))
    crit = O.pde_criterion(c['crit'][0], beta=c['crit'][1], reduction=c['crit'][2] if len(c['crit']) > 2 else 'mean') if 'crit' in c else None
    total, parts, fn, ph = O.place_one_batch(st, x, y, t, inp['f'], inp['field_data'], inp['coord_data'], inp['forecast_h'], GEO, return_parts=True,
                                             crit=crit, norm_cfg=getattr(O, c['norm'])() if c.get('norm') else None)
    assert _rel(torch.cat(ph, 1).detach().numpy(), d[case + '.fields_phys']) < 2e-6
    mine = np.array([float(p.detach()) for p in parts])
    assert np.all(np.abs(mine - d[case + '.parts']) <= 2e-5 * np.abs(d[case + '.parts'])), (mine, d[case + '.parts'])
    assert abs(float(total.detach()) - float(d[case + '.total'])) <= 1e-5 * abs(float(d[case + '.total']))
    names = O.param_names(st)
    grads = dict(zip(names, torch.autograd.grad(total, [st[n] for n in names])))
    ref_norm = dict(zip([str(n) for n in d[case + '.grad_names']], d[case + '.grad_norms']))
    assert set(ref_norm) == set(names)
    top = max(ref_norm.values())
    for n in names:
        if n.endswith('key_projection.bias'):
            continue
        assert abs(float(grads[n].double().norm()) - ref_norm[n]) <= 5e-4 * ref_norm[n] + 1e-9 * top, n


@pytest.mark.parametrize('with_clip', [True, False])
def test_f9_wide_outputs_clip_masks(golden_dir, with_clip):
    st = O.make_state(gain=5.0)
    d = _load(golden_dir, 'f9_wide_clip%d_fp32.npz' % int(with_clip))
    total, parts, fn, ph, jac = _run_pde(st, synthetic_inputs(128, tag='f9'), with_clip)
    assert _rel(fn.detach().numpy(), d['fields_norm']) < 2e-6
    j, jr = jac.detach().numpy(), d['jac']
    assert np.array_equal(j == 0, jr == 0)
    assert (jr == 0).all(-1).mean() > 0.1 or not with_clip
    mine = np.array([float(p.detach()) for p in parts])
    ok = np.isfinite(d['parts'])
    assert np.array_equal(np.isfinite(mine), ok)
    assert np.all(np.abs(mine[ok] - d['parts'][ok]) <= 1e-4 * np.abs(d['parts'][ok]))


def test_f5_fp64_agrees_with_reference_fp64(golden_dir, state):
    d = _load(golden_dir, 'f5_pde_clip1_fp64.npz')
    total, parts, fn, ph, jac = _run_pde(state, synthetic_inputs(256, tag='inter'), True, torch.float64)
    mine = np.array([float(p.detach()) for p in parts])
    # the reference casts every loss scalar with .float() (interface_physics.py:104), so its fp64 run carries one fp32 rounding
    assert np.all(np.abs(mine - d['parts']) <= 2e-7 * np.abs(d['parts']))
    assert _rel(jac.detach().numpy()[:32], d['jac']) < 1e-9


def test_f7_parameter_gradients(golden_dir):
    """Second-order correctness: d(total PDE loss)/d(all 155 parameters)."""
    d = _load(golden_dir, 'f7_grads_fp32.npz')
    st = O.make_state(requires_grad=True)
    inp = synthetic_inputs(256, tag='inter')
    x, y, t = (inp[k].clone().requires_grad_(True) for k in ('x', 'y', 't'))
    total = O.place_one_batch(st, x, y, t, inp['f'], inp['field_data'], inp['coord_data'], inp['forecast_h'], GEO)
    names = O.param_names(st)
    grads = dict(zip(names, torch.autograd.grad(total, [st[n] for n in names])))
    ref_norm = dict(zip([str(n) for n in d['names']], d['norms']))
    assert set(ref_norm) == set(names) and len(names) == 155
    for n in names:
        mine = float(grads[n].double().norm())
        if n.endswith('key_projection.bias'):
            # mathematically zero (softmax is invariant to a key bias): pure rounding noise in both codes
            assert mine < 1e-3 and ref_norm[n] < 1e-3, n
            continue
        assert abs(mine - ref_norm[n]) <= 2e-4 * ref_norm[n] + 1e-12, n
    for key in d.files:
        if key.startswith('g.'):
            assert _rel(grads[key[2:]].numpy(), d[key]) < 2e-4, key
        if key.startswith('s.'):
            assert _rel(grads[key[2:]].flatten()[::97].numpy(), d[key]) < 2e-4, key


def test_f6_data_loss(golden_dir):
    d = _load(golden_dir, 'f6_data_loss.npz')
    st = O.make_state(requires_grad=True)
    inp = synthetic_inputs(256, tag='margin', margin=True)
    loss = O.data_loss(st, inp['x'], inp['y'], inp['t'], inp['field_data'], inp['coord_data'], inp['labels'], inp['forecast_h'], GEO)
    assert abs(float(loss.detach()) - float(d['loss'])) <= 1e-6 * float(d['loss'])
    names = O.param_names(st)
    grads = dict(zip(names, torch.autograd.grad(loss, [st[n] for n in names])))
    ref_norm = dict(zip([str(n) for n in d['names']], d['norms']))
    for n in names:
        if n.endswith('key_projection.bias'):
            continue
        assert abs(float(grads[n].double().norm()) - ref_norm[n]) <= 2e-4 * ref_norm[n] + 1e-12, n


def test_f8_optimiser_step(golden_dir):
    """a18: data loss + PDE(inter) + PDE(margin) -> clip_grad_norm_(2.5e7) -> Adam(lr 1e-4, wd 1e-4)."""
    d = _load(golden_dir, 'f8_step.npz')
    st = O.make_state(requires_grad=True)
    inter = synthetic_inputs(256, tag='inter')
    margin = synthetic_inputs(256, tag='margin', margin=True)
    loss = O.data_loss(st, margin['x'], margin['y'], margin['t'], margin['field_data'], margin['coord_data'], margin['labels'],
                       margin['forecast_h'], GEO)
    for inp in (inter, margin):
        x, y, t = (inp[k].clone().requires_grad_(True) for k in ('x', 'y', 't'))
        loss = loss + O.place_one_batch(st, x, y, t, inp['f'], inp['field_data'], inp['coord_data'], inp['forecast_h'], GEO)
    assert abs(float(loss.detach()) - float(d['loss'])) <= 1e-5 * float(d['loss'])
    names = O.param_names(st)
    grads = dict(zip(names, torch.autograd.grad(loss, [st[n] for n in names])))
    before = {n: st[n].detach().clone() for n in names}
    gnorm = O.clip_and_adam_step(st, grads, {})
    assert abs(float(gnorm) - float(d['gnorm'])) <= 2e-4 * float(d['gnorm'])
    ref_post = dict(zip([str(n) for n in d['names']], d['post_norms']))
    ref_delta = dict(zip([str(n) for n in d['names']], d['delta_norms']))
    for n in names:
        assert abs(float(st[n].detach().double().norm()) - ref_post[n]) <= 1e-6 * ref_post[n] + 1e-12, n
        dn = float((st[n].detach() - before[n]).double().norm())
        assert abs(dn - ref_delta[n]) <= 2e-3 * ref_delta[n] + 1e-12, n


def _f13_stride(numel):
    return max(1, numel // 128)


def _step_deltas_agree(name, mine_delta, ref_delta, ref_g, gmax, lr=1e-4, clear=1e-4, frac_allowed=0.0):
    """One Adam step from zero moments moves an entry by -lr * g / (|g| + eps'): lr * sign(g) wherever the gradient is clear of zero.  Entries whose
    reference gradient is above `clear` x the tensor's largest must agree within 2 % of lr; the others (a rounding-level gradient decides the sign of
    a full-size step) are bounded by the step size itself."""
    big = np.abs(ref_g) > clear * gmax
    bad = np.abs(mine_delta - ref_delta)[big] > 0.02 * lr
    assert bad.mean() <= frac_allowed if big.any() else True, (name, int(bad.sum()), int(big.sum()))
    assert np.all(np.abs(mine_delta) <= 1.05 * lr), name
    return int(big.sum())


def test_f13_elementwise_data_loss_gradients_and_optimiser_step(golden_dir):
    """ELEMENT-wise pins where F6 / F8 hold norms: sampled entries of every data-loss gradient, and of every parameter's
    (post - pre) of one clip + Adam step, against the reference's own."""
    d = _load(golden_dir, 'f13_elementwise_data_loss_and_step.npz')
    names = O.param_names(O.make_state())
    assert [str(n) for n in d['names']] == names
    # (a) data loss
    st = O.make_state(requires_grad=True)
    margin = synthetic_inputs(256, tag='margin', margin=True)
    loss = O.data_loss(st, margin['x'], margin['y'], margin['t'], margin['field_data'], margin['coord_data'], margin['labels'], margin['forecast_h'], GEO)
    assert abs(float(loss.detach()) - float(d['dl.loss'])) <= 1e-6 * float(d['dl.loss'])
    grads = dict(zip(names, torch.autograd.grad(loss, [st[n] for n in names])))
    for n in names:
        if n.endswith('key_projection.bias'):
            continue
        mine = grads[n].flatten()[::_f13_stride(grads[n].numel())].numpy()
        assert np.abs(mine - d['dl.g.' + n]).max() <= 2e-4 * float(d['dl.gmax.' + n]) + 1e-30, n
    # (b) the optimiser step
    st = O.make_state(requires_grad=True)
    inter = synthetic_inputs(256, tag='inter')
    loss = O.data_loss(st, margin['x'], margin['y'], margin['t'], margin['field_data'], margin['coord_data'], margin['labels'], margin['forecast_h'], GEO)
    for inp in (inter, margin):
        x, y, t = (inp[k].clone().requires_grad_(True) for k in ('x', 'y', 't'))
        loss = loss + O.place_one_batch(st, x, y, t, inp['f'], inp['field_data'], inp['coord_data'], inp['forecast_h'], GEO)
    assert abs(float(loss.detach()) - float(d['step.loss'])) <= 1e-5 * float(d['step.loss'])
    grads = dict(zip(names, torch.autograd.grad(loss, [st[n] for n in names])))
    before = {n: st[n].detach().clone() for n in names}
    O.clip_and_adam_step(st, grads, {})
    checked = 0
    for n in names:
        if n.endswith('key_projection.bias'):
            continue
        stride = _f13_stride(before[n].numel())
        mine = (st[n].detach() - before[n]).flatten()[::stride].numpy()
        checked += _step_deltas_agree(n, mine, d['step.delta.' + n], d['step.g.' + n], float(d['step.gmax.' + n]))
    assert checked > 5000
