"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box never sees it).
Third-party packages the reference imports but the hot path never touches
(torchvision, GDAL, xarray, tensorboard, mmcv, ...) are replaced by empty
stand-in modules in sys.modules, as SURVEY.md 8(c) describes.  The reference's
own modules are imported unmodified and filled with oracle/fill.py's closed-form
parameters; the outputs are what the reference computes on CPU in fp32 (and
fp64 where noted).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

import numpy as np
import torch
import torch.nn  # noqa
import torch.utils.data  # noqa
import torch._dynamo  # noqa  (torch.optim imports it lazily; must happen before the stand-in modules exist)
import importlib.machinery


def _stub_third_party():
    class _Anything(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return _Anything(self.__name__ + '.' + name)

        def __call__(self, *a, **k):
            return None

    for name in ['torchvision', 'torchvision.models', 'torchvision.ops', 'torchvision.transforms', 'pyproj', 'osgeo', 'osgeo.gdal', 'osgeo.osr',
                 'osgeo.ogr', 'osgeo.gdalconst', 'gdal', 'osr', 'ogr', 'gdalconst', 'bs4', 'cv2', 'skimage', 'skimage.io', 'netCDF4', 'wrf',
                 'xarray', 'tensorboard', 'torch.utils.tensorboard', 'mmcv', 'tqdm', 'matplotlib', 'matplotlib.pyplot',
                 'mpl_toolkits', 'mpl_toolkits.basemap', 'pandas', 'scipy', 'scipy.interpolate', 'PIL', 'PIL.Image']:
        if name not in sys.modules:
            try:
                if name in ('tqdm', 'pandas', 'scipy', 'scipy.interpolate', 'matplotlib', 'matplotlib.pyplot'):
                    __import__(name)
                    continue
            except Exception:
                pass
            mod = _Anything(name)
            mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = mod


def load_reference():
    _stub_third_party()
    sys.path.insert(0, REF)
    from DeepPhysiNet.interface.build import builder_models
    spec = importlib.util.spec_from_file_location('ref_cfg', os.path.join(REF, 'configs', 'DeepPhysiNet_NCEP_cfg.py'))
    cfg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cfg)
    m = builder_models(**cfg.config)
    m.dx = m.dy = 27000.0
    m.dt = 3600.0
    m.pred_t_span = 86400.0
    m.with_clip = True
    return m, cfg


def write_f10(m, run_pde):
    """F10: the PDE path on GRID-NODE points (x, y exact multiples of dx, dy incl. both domain edges: xi = 0 and 1), a ragged batch of 200
    and the longest lead time (336 h) -- a second, independent pin of a2-a16 next to F3-F5 (interior points, 24 h)."""
    from oracle.fill import synthetic_inputs
    inp = synthetic_inputs(200, tag='f10', margin=True, forecast_h=336.0 / 360.0)
    inp['x'][0, 0], inp['y'][0, 0] = 0.0, 0.0                                     # both corners of the domain
    inp['x'][1, 0], inp['y'][1, 0] = 256 * 27000.0, 144 * 27000.0
    rec, _ = run_pde(m, inp, True, torch.float32)
    np.savez_compressed(os.path.join(HERE, 'f10_grid_nodes_h336_fp32.npz'), x=inp['x'].numpy(), y=inp['y'].numpy(), **rec)


def f12_norm_cfg(base):
    """F12's de-normalisation table: u10 not normalised at all (use_norm False), pres in the two-factor min_max form, the rest as shipped."""
    import copy
    c = copy.deepcopy(base)
    c['u10']['use_norm'] = False
    m_, s_ = c['pres']['norm_factor']
    c['pres']['norm_type'] = 'min_max'
    c['pres']['norm_factor'] = [m_ - 0.5 * s_, m_ + 0.5 * s_]
    return c


def f12_norm_sq_cfg(base):
    """F12's third table: q2 in the three-factor min_max form (squared and shifted, interface_physics.py:244-247), the rest as shipped."""
    import copy
    c = copy.deepcopy(base)
    c['q2']['norm_type'] = 'min_max'
    c['q2']['norm_factor'] = [0.08, 0.11, 0.001]
    return c


def write_f12(m, run_pde, builder_loss, inter):
    """F12: the branches of the PDE path the shipped config does not take, run on the reference itself -- the two other criteria its loss builder
    offers for `pde_loss` (L1Loss, WeightSmoothL1Loss(beta), interface_physics.py:384 + losses/builder.py), and inverse_norm's other branches
    (use_norm False, two-factor min_max, :238-243).  256 interior points, clip on, fp32; parameter gradients' norms for the criteria."""
    import copy
    base = copy.deepcopy(m.obs_norm_cfg)
    out = {}
    for tag, crit, ncfg in (('l1', builder_loss(name='L1Loss'), None), ('sl1', builder_loss(name='WeightSmoothL1Loss', beta=0.1), None),
                            ('sl1_b2', builder_loss(name='WeightSmoothL1Loss', beta=2.0), None), ('mse_sum', builder_loss(name='MSELoss', reduction='sum'), None),
                            ('norm', None, f12_norm_cfg(base)),
                            ('norm_sq', None, f12_norm_sq_cfg(base))):
        m.physics_net.zero_grad()
        rec, total = run_pde(m, inter, True, torch.float32, crit=crit, norm_cfg=ncfg if ncfg is not None else copy.deepcopy(base))
        total.backward()
        g = {k: p.grad.detach() for k, p in m.physics_net.named_parameters()}
        out[tag + '.parts'], out[tag + '.total'] = rec['parts'], rec['total']
        out[tag + '.fields_phys'] = rec['fields_phys']
        out[tag + '.grad_norms'] = np.array([float(v.double().norm()) for v in g.values()])
        out[tag + '.grad_names'] = np.array(list(g.keys()))
    m.obs_norm_cfg = base
    np.savez_compressed(os.path.join(HERE, 'f12_criteria_and_norm_branches.npz'), **out)


def f13_stride(numel):
    """Sampling stride of fixture F13: every tensor contributes at most ~128 entries (all of them when it is smaller)."""
    return max(1, numel // 128)


def write_f13(m, net, cfg, builder_loss, inter, margin):
    """F13: ELEMENT-wise pins where F6 / F8 hold norms only.  (a) every parameter gradient of the data loss (F6's run), sampled
    with stride f13_stride(numel); (b) one optimiser step of F8's loss (data + PDE(inter) + PDE(margin), clip 2.5e7, Adam(1e-4, wd 1e-4)): the sampled
    entries of (post - pre) for every parameter, and of the gradient that produced them (so that a test can tell a rounding-level gradient, whose Adam
    step has an arbitrary sign, from a real one)."""
    out = {}
    net.zero_grad()
    crit_d = builder_loss(name='WeightSmoothL1Loss', beta=0.1)
    pe = m.encoding_coord(margin['x'], margin['y'], margin['t'], m.pred_t_span)
    fn = net(margin['field_data'], pe, margin['coord_data'], margin['forecast_h'])
    dl = crit_d(torch.cat(fn, dim=1), margin['labels']).float() * 1e6
    dl.backward()
    names = [k for k, _ in net.named_parameters()]
    out['names'] = np.array(names)
    out['dl.loss'] = np.array(float(dl), np.float64)
    for k, p in net.named_parameters():
        out['dl.g.' + k] = p.grad.detach().flatten()[::f13_stride(p.numel())].numpy().copy()
        out['dl.gmax.' + k] = np.array(float(p.grad.detach().abs().max()), np.float64)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-4)
    net.zero_grad()
    pe = m.encoding_coord(margin['x'], margin['y'], margin['t'], m.pred_t_span)
    fn = net(margin['field_data'], pe, margin['coord_data'], margin['forecast_h'])
    loss = crit_d(torch.cat(fn, dim=1), margin['labels']).float() * 1e6
    crit = builder_loss(name='MSELoss')
    lf = cfg.config['train_cfg']['losses']['loss_factor']
    m.with_clip = True
    for inp_, pre in ((inter, 'inter'), (margin, 'margin')):
        x = inp_['x'].clone().requires_grad_(True)
        y = inp_['y'].clone().requires_grad_(True)
        t = inp_['t'].clone().requires_grad_(True)
        loss = loss + m.place_one_batch(x, y, t, inp_['f'], inp_['field_data'], inp_['coord_data'], inp_['forecast_h'],
                                        crit, lf, global_step=2, local_rank=0, device='cpu', summary=None, prefix=pre)
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=2.5e7)
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    opt.step()
    out['step.loss'], out['step.gnorm'] = np.array(float(loss), np.float64), np.array(float(gnorm), np.float64)
    for k, p in net.named_parameters():
        st = f13_stride(p.numel())
        out['step.delta.' + k] = (p.detach() - sd0[k]).flatten()[::st].numpy().copy()
        out['step.g.' + k] = grads[k].flatten()[::st].numpy().copy()
        out['step.gmax.' + k] = np.array(float(grads[k].abs().max()), np.float64)
    net.load_state_dict(sd0)
    np.savez_compressed(os.path.join(HERE, 'f13_elementwise_data_loss_and_step.npz'), **out)


def write_f11():
    """F11: get_coriolis (dataset/physics_dataset.py:521-526) called on the reference class itself (it never touches `self`): the
    latitude forms its two callers build -- margin points `begin_lat + y_rand * 0.25` with integer node indices (:336-337, :418) and
    interior points with continuous draws (:444-445, :496) -- 1-D (expanded to [n,1]) and already 2-D; pins the f column of the
    on-device collocation sampler (the interpolation itself stays scipy-pinned: xarray is absent)."""
    _stub_third_party()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from DeepPhysiNet.dataset.physics_dataset import PhysicsDataset
    from oracle.fill import unit_uniform
    y_nodes = np.arange(145, dtype=np.int64)
    lat_nodes = 18.0 + y_nodes * 0.25
    y_cont = (unit_uniform('f11.y', 512).astype(np.float64) + 1.0) * 0.5 * 144.0
    lat_cont = 18.0 + y_cont * 0.25
    f_nodes = PhysicsDataset.get_coriolis(None, lat_nodes)
    f_cont = PhysicsDataset.get_coriolis(None, lat_cont)
    f_2d = PhysicsDataset.get_coriolis(None, lat_cont[:7].reshape(7, 1))
    assert f_nodes.shape == (145, 1) and f_cont.shape == (512, 1) and f_2d.shape == (7, 1)
    np.savez_compressed(os.path.join(HERE, 'f11_coriolis.npz'), y_nodes=y_nodes, lat_nodes=lat_nodes, f_nodes=f_nodes,
                        f_nodes_f32=torch.from_numpy(f_nodes).float().numpy(), y_cont=y_cont, lat_cont=lat_cont, f_cont=f_cont,
                        f_cont_f32=torch.from_numpy(f_cont).float().numpy(), f_2d=f_2d)


def main():
    from oracle.fill import fill_state_dict_, synthetic_inputs
    if '--only-f11' in sys.argv:
        write_f11()
        return
    only_f10 = '--only-f10' in sys.argv
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m, cfg = load_reference()
    from DeepPhysiNet.losses.builder import builder_loss  # load_reference() put REF on sys.path
    m.eval()
    net = m.physics_net
    sd = net.state_dict()
    fill_state_dict_(sd)
    net.load_state_dict(sd, strict=True)
    names = [(k, tuple(v.shape)) for k, v in sd.items()]
    out = {}

    # ---- F0: state-dict contract
    np.savez_compressed(os.path.join(HERE, 'f0_state_names.npz'),
                        names=np.array([k for k, _ in names]), shapes=np.array([str(s) for _, s in names]))

    # ---- F1: positional encodings (a1, a2)
    from DeepPhysiNet.utils.position_encoding import SineCosPE
    inp = synthetic_inputs(16, tag='f1')
    x3 = torch.cat([inp['x'] / 27000.0 / 256, inp['y'] / 27000.0 / 144, inp['t'] / 86400.0], dim=1)
    f1 = dict(in3=x3.numpy(), in6=inp['coord_data'].numpy(), in1=np.array([[24.0 / 360.0]], np.float32),
              pe3=SineCosPE(3, N_freqs=32, include_input=False)(x3).numpy(),
              pe6=SineCosPE(6, N_freqs=16, include_input=False)(inp['coord_data']).numpy(),
              pe1_96=SineCosPE(1, N_freqs=96, include_input=False)(torch.tensor([[24.0 / 360.0]])).numpy(),
              pe1_128=SineCosPE(1, N_freqs=128, include_input=False)(torch.tensor([[[24.0 / 360.0]]])).numpy(),
              enc=m.encoding_coord(inp['x'], inp['y'], inp['t'], m.pred_t_span).numpy(),
              x=inp['x'].numpy(), y=inp['y'].numpy(), t=inp['t'].numpy())
    np.savez_compressed(os.path.join(HERE, 'f1_pe.npz'), **f1)

    # ---- F2: encoder (a3-a5)
    N = 256
    inter = synthetic_inputs(N, tag='inter')
    f2 = {}
    with torch.no_grad():
        for h in (0.0, 24.0, 336.0):
            fh = torch.full((1, 1, 1), h / 360.0)
            f2['meta_out_h%d' % int(h)] = net.meta_net(inter['field_data'], fh).numpy()
    np.savez_compressed(os.path.join(HERE, 'f2_encoder.npz'), **f2)

    def run_pde(model, inputs, with_clip, dtype, crit=None, norm_cfg=None):
        model.with_clip = with_clip
        if norm_cfg is not None:
            model.obs_norm_cfg = norm_cfg
        cast = lambda v: v.to(dtype)
        x = cast(inputs['x']).clone().requires_grad_(True)
        y = cast(inputs['y']).clone().requires_grad_(True)
        t = cast(inputs['t']).clone().requires_grad_(True)
        f = cast(inputs['f'])
        crit = crit if crit is not None else builder_loss(name='MSELoss')
        lf = cfg.config['train_cfg']['losses']['loss_factor']
        rec = {}
        pe = model.encoding_coord(x, y, t, model.pred_t_span)
        fn = model.physics_net(cast(inputs['field_data']), pe, cast(inputs['coord_data']), cast(inputs['forecast_h']))
        ph = model.inverse_norm(*fn, obs_norm_cfg=model.obs_norm_cfg)
        jac = torch.stack([torch.cat([model.gradient(v, x), model.gradient(v, y), model.gradient(v, t)], dim=1) for v in ph], dim=1)
        rec['fields_norm'] = torch.cat(fn, dim=1).detach().numpy()
        rec['fields_phys'] = torch.cat(ph, dim=1).detach().numpy()
        rec['jac'] = jac.detach().numpy()
        u, v, P, T, q, rio = ph
        parts = [
            model.montion_equation_u(x, y, t, u, v, P, rio, f, crit, factor=lf['motion_u_factor']),
            model.montion_equation_v(x, y, t, u, v, P, rio, f, crit, factor=lf['motion_v_factor']),
            model.continuous_equation(x, y, t, u, v, rio, crit, factor=lf['continuous_factor']),
            model.energy_equation(x, y, t, u, v, P, T, rio, q, crit, factor=lf['energy_factor']),
            model.vapor_equation(x, y, t, u, v, P, T, q, crit, factor=lf['vapor_factor']),
            model.gas_equation(P, T, rio, q, crit, factor=lf['gas_factor'])]
        rec['parts'] = np.array([float(p_) for p_ in parts], dtype=np.float64)
        # the reference's own entry point (a16)
        x2 = cast(inputs['x']).clone().requires_grad_(True)
        y2 = cast(inputs['y']).clone().requires_grad_(True)
        t2 = cast(inputs['t']).clone().requires_grad_(True)
        total = model.place_one_batch(x2, y2, t2, f, cast(inputs['field_data']), cast(inputs['coord_data']),
                                      cast(inputs['forecast_h']), crit, lf, global_step=2, local_rank=0, device='cpu',
                                      summary=None, prefix='inter', log_step=100)
        rec['total'] = np.array(float(total), dtype=np.float64)
        return rec, total

    if only_f10:                               # add the one fixture without rewriting (re-zipping) the others
        write_f10(m, run_pde)
        print('f10 written')
        return
    if '--only-f12' in sys.argv:
        write_f12(m, run_pde, builder_loss, inter)
        print('f12 written')
        return
    if '--only-f13' in sys.argv:
        write_f13(m, net, cfg, builder_loss, inter, synthetic_inputs(N, tag='margin', margin=True))
        print('f13 written')
        return

    # ---- F3/F4/F5: VariableNet outputs, Jacobian, residuals (a6-a16), fp32, clip on/off
    for wc in (True, False):
        rec, _ = run_pde(m, inter, wc, torch.float32)
        np.savez_compressed(os.path.join(HERE, 'f345_pde_clip%d_fp32.npz' % int(wc)), **rec)

    # ---- F7: parameter gradients of the PDE loss (second-order correctness) + data loss F6
    net.zero_grad()
    rec, total = run_pde(m, inter, True, torch.float32)
    total.backward()
    g = {k: p.grad.detach() for k, p in net.named_parameters()}
    f7 = dict(names=np.array(list(g.keys())), norms=np.array([float(v.double().norm()) for v in g.values()]),
              total=rec['total'])
    for k in g:
        if k.endswith('out_fc.weight') or k.endswith('coord_input_fc.bias') or k.endswith('out_fc.bias') or k == 'meta_net.model.projection.bias' \
                or k.endswith('cat_fc1.fc.2.bias') or k.endswith('cat_fc1.fc.0.bias'):
            f7['g.' + k] = g[k].numpy()
    # a few sampled entries of the big matrices
    for k in ('U_net.cat_fc1.fc.0.weight', 'P_net.cat_fc1.fc.2.weight', 'q_net.data_input_fc.weight', 'T_net.coord_hidden_fc.weight',
              'rio_net.coord_input_fc.weight', 'V_net.fore_h_fc.weight'):
        f7['s.' + k] = g[k].flatten()[::97].numpy()
    np.savez_compressed(os.path.join(HERE, 'f7_grads_fp32.npz'), **f7)

    # ---- F6: data loss (a17) on margin (grid-node) points + its gradients' norms
    margin = synthetic_inputs(N, tag='margin', margin=True)
    net.zero_grad()
    crit_d = builder_loss(name='WeightSmoothL1Loss', beta=0.1)
    pe = m.encoding_coord(margin['x'], margin['y'], margin['t'], m.pred_t_span)
    fn = net(margin['field_data'], pe, margin['coord_data'], margin['forecast_h'])
    dl = crit_d(torch.cat(fn, dim=1), margin['labels']).float() * 1e6
    dl.backward()
    g = {k: p.grad.detach() for k, p in net.named_parameters()}
    np.savez_compressed(os.path.join(HERE, 'f6_data_loss.npz'), loss=np.array(float(dl), np.float64),
                        fields_norm=torch.cat(fn, dim=1).detach().numpy(),
                        names=np.array(list(g.keys())), norms=np.array([float(v.double().norm()) for v in g.values()]))

    # ---- F8: one full optimiser step (a18): data loss + PDE(inter) + PDE(margin), clip 2.5e7, Adam(1e-4, wd 1e-4)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-4)
    net.zero_grad()
    pe = m.encoding_coord(margin['x'], margin['y'], margin['t'], m.pred_t_span)
    fn = net(margin['field_data'], pe, margin['coord_data'], margin['forecast_h'])
    loss = crit_d(torch.cat(fn, dim=1), margin['labels']).float() * 1e6
    crit = builder_loss(name='MSELoss')
    lf = cfg.config['train_cfg']['losses']['loss_factor']
    m.with_clip = True
    for inp_, pre in ((inter, 'inter'), (margin, 'margin')):
        x = inp_['x'].clone().requires_grad_(True)
        y = inp_['y'].clone().requires_grad_(True)
        t = inp_['t'].clone().requires_grad_(True)
        loss = loss + m.place_one_batch(x, y, t, inp_['f'], inp_['field_data'], inp_['coord_data'], inp_['forecast_h'],
                                        crit, lf, global_step=2, local_rank=0, device='cpu', summary=None, prefix=pre)
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=2.5e7)
    opt.step()
    post = {k: v for k, v in net.named_parameters()}
    np.savez_compressed(os.path.join(HERE, 'f8_step.npz'), loss=np.array(float(loss), np.float64), gnorm=np.array(float(gnorm), np.float64),
                        names=np.array(list(post.keys())),
                        post_norms=np.array([float(v.detach().double().norm()) for v in post.values()]),
                        delta_norms=np.array([float((v.detach() - sd0[k]).double().norm()) for k, v in post.items()]))
    net.load_state_dict(sd0)
    write_f13(m, net, cfg, builder_loss, inter, margin)

    # ---- F5b: fp64 run of the reference (tolerance calibration)
    m64 = m.double()
    rec, _ = run_pde(m64, inter, True, torch.float64)
    np.savez_compressed(os.path.join(HERE, 'f5_pde_clip1_fp64.npz'), parts=rec['parts'], total=rec['total'],
                        jac=rec['jac'].astype(np.float64)[:32], fields_norm=rec['fields_norm'][:32])
    m.float()

    # ---- F9: wide outputs (out_fc gain 5: raw outputs have std ~10, so many P/T/q/rho points sit on a clip bound)
    sd9 = net.state_dict()
    fill_state_dict_(sd9, gain=5.0)
    net.load_state_dict(sd9, strict=True)
    for wc in (True, False):
        rec, _ = run_pde(m, synthetic_inputs(128, tag='f9'), wc, torch.float32)
        np.savez_compressed(os.path.join(HERE, 'f9_wide_clip%d_fp32.npz' % int(wc)), **rec)
    net.load_state_dict(sd0)
    write_f10(m, run_pde)
    write_f11()
    print('golden vectors written to', HERE)
    for fn_ in sorted(os.listdir(HERE)):
        if fn_.endswith('.npz'):
            print('  %-34s %8.1f KB' % (fn_, os.path.getsize(os.path.join(HERE, fn_)) / 1024))


if __name__ == '__main__':
    main()
