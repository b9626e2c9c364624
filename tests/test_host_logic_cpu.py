"""CPU: host-side mirror of the reference interface (module tree, state_dict contract, encoder math, factories, generic
residual expressions, checkpoints).  The per-point HIP path is not exercised here (it has no CPU form)."""
import os

import numpy as np
import pytest
import torch

from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
from deepphysinet_amd.losses import builder_loss
from deepphysinet_amd.model import PhysicsNet
from deepphysinet_amd.utils.position_encoding import SineCosPE
from oracle import dpn_oracle as O
from oracle.fill import fill_state_dict_, synthetic_inputs


@pytest.fixture(scope='module', autouse=True)
def _reference_math():
    """These tests evaluate the per-field torch expressions on host tensors (module tree / encoder math against the golden vectors);
    the product refuses host tensors unless this is switched on (deepphysinet_amd._lib.host_math_or_raise)."""
    from deepphysinet_amd import _lib
    _lib.enable_cpu_reference_math(True)
    yield
    _lib.enable_cpu_reference_math(False)


def test_host_tensors_are_refused_without_reference_math():
    from deepphysinet_amd import _lib
    from deepphysinet_amd.linear import linear
    _lib.enable_cpu_reference_math(False)
    try:
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            linear(torch.zeros(2, 4), torch.zeros(3, 4), torch.zeros(3))
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            builder_loss(name='WeightSmoothL1Loss', beta=0.1)(torch.zeros(4, 6), torch.zeros(4, 6))
    finally:
        _lib.enable_cpu_reference_math(True)


@pytest.fixture(scope='module')
def model():
    m = builder_models(**ncep_config())
    sd = m.physics_net.state_dict()
    fill_state_dict_(sd)
    m.physics_net.load_state_dict(sd)
    return m


def test_state_dict_names_shapes_and_order(golden_dir, model):
    d = np.load(os.path.join(golden_dir, 'f0_state_names.npz'))
    ref = [(str(k), str(s)) for k, s in zip(d['names'], d['shapes'])]
    mine = [(k, str(tuple(v.shape))) for k, v in model.physics_net.state_dict().items()]
    assert mine == ref                      # a reference checkpoint loads with strict=True


def test_sine_cos_pe_matches_reference_vectors(golden_dir):
    d = np.load(os.path.join(golden_dir, 'f1_pe.npz'))
    t = torch.from_numpy
    assert np.array_equal(SineCosPE(3, N_freqs=32, include_input=False)(t(d['in3'])).numpy(), d['pe3'])
    assert np.array_equal(SineCosPE(6, N_freqs=16, include_input=False)(t(d['in6'])).numpy(), d['pe6'])
    assert np.array_equal(SineCosPE(1, N_freqs=96, include_input=False)(t(d['in1'])).numpy(), d['pe1_96'])
    pe = SineCosPE(3, N_freqs=4)
    assert pe.out_dim == 27 and pe(torch.zeros(5, 3)).shape == (5, 27)
    assert 'freq_bands' not in pe.state_dict()          # non-persistent buffer, like the reference


def test_encoding_coord(golden_dir, model):
    d = np.load(os.path.join(golden_dir, 'f1_pe.npz'))
    t = torch.from_numpy
    enc = model.encoding_coord(t(d['x']), t(d['y']), t(d['t']), model.pred_t_span).numpy()
    assert np.array_equal(enc, d['enc'])
    assert model.pred_t_span == 86400.0 and model.dx == 27000.0 and (model.lat_size, model.lon_size) == (145, 257)


def test_encoder_matches_reference_vectors(golden_dir, model):
    d = np.load(os.path.join(golden_dir, 'f2_encoder.npz'))
    inp = synthetic_inputs(4)
    with torch.no_grad():
        for h in (0, 24, 336):
            mo = model.physics_net.meta_net(inp['field_data'], torch.full((1, 1, 1), h / 360.0)).numpy()
            ref = d['meta_out_h%d' % h]
            assert np.abs(mo - ref).max() <= 3e-6 * np.abs(ref).max()


def test_hyper_heads_batched_gemm_equals_per_net_linears(model):
    """PhysicsNet.field_weights packs the 12 head GEMMs into one; compare with the reference formulation net by net."""
    inp = synthetic_inputs(4)
    net = model.physics_net
    with torch.no_grad():
        heads, evec, statics = net.field_weights(inp['field_data'], inp['forecast_h'])
        meta_out = net.meta_net(inp['field_data'], inp['forecast_h'])
        for k, vn in enumerate(net.nets_in_output_order()):
            w1b1, w2b2, e = vn.hyper_weights(meta_out, inp['forecast_h'])
            assert torch.allclose(heads[:, k * 193:(k + 1) * 193], w1b1, rtol=1e-5, atol=1e-6)
            assert torch.allclose(heads[:, 1158 + k * 257:1158 + (k + 1) * 257], w2b2, rtol=1e-5, atol=1e-6)
            assert torch.allclose(evec[k], e, rtol=1e-5, atol=1e-6)
    assert len(statics) == 48 and statics[6].shape == (1, 256)


def test_loss_factory_surface():
    assert isinstance(builder_loss('MSELoss'), torch.nn.MSELoss)
    crit = builder_loss(name='WeightSmoothL1Loss', beta=0.1)
    a, b = torch.randn(32, 6), torch.randn(32, 6)
    assert torch.allclose(crit(a, b), torch.nn.functional.smooth_l1_loss(a, b, beta=0.1))
    with pytest.raises(NotImplementedError):
        builder_loss('nope')
    with pytest.raises(NotImplementedError):
        builder_models(name='nope')


def test_generic_equation_methods_equal_reference_terms(golden_dir, model):
    """The six *_equation methods + inverse_norm, fed with autograd-connected oracle fields, reproduce the golden scalars."""
    d = np.load(os.path.join(golden_dir, 'f345_pde_clip1_fp32.npz'))
    inp = synthetic_inputs(256, tag='inter')
    st = O.make_state()
    x, y, t = (inp[k].clone().requires_grad_(True) for k in ('x', 'y', 't'))
    pe = model.encoding_coord(x, y, t, model.pred_t_span)
    fn = O.physics_net_forward(st, inp['field_data'], pe, inp['coord_data'], inp['forecast_h'])
    model.with_clip = True
    u, v, P, T, q, rio = model.inverse_norm(*fn, obs_norm_cfg=model.obs_norm_cfg)
    crit = builder_loss('MSELoss')
    lf = model.train_cfg['losses']['loss_factor']
    f = inp['f']
    mine = [model.montion_equation_u(x, y, t, u, v, P, rio, f, crit, factor=lf['motion_u_factor']),
            model.montion_equation_v(x, y, t, u, v, P, rio, f, crit, factor=lf['motion_v_factor']),
            model.continuous_equation(x, y, t, u, v, rio, crit, factor=lf['continuous_factor']),
            model.energy_equation(x, y, t, u, v, P, T, rio, q, crit, factor=lf['energy_factor']),
            model.vapor_equation(x, y, t, u, v, P, T, q, crit, factor=lf['vapor_factor']),
            model.gas_equation(P, T, rio, q, crit, factor=lf['gas_factor'])]
    mine = np.array([float(m_.detach()) for m_ in mine])
    assert np.all(np.abs(mine - d['parts']) <= 2e-5 * np.abs(d['parts']))


def test_checkpoint_roundtrip_and_ddp_prefix(tmp_path, model):
    model.save_model(str(tmp_path), epoch=3, global_step=77, dx=27000.0)
    assert os.path.exists(tmp_path / 'physics_3.pth') and os.path.exists(tmp_path / 'physics_latest.pth')
    sd, epoch, step = model.load_model(str(tmp_path), prefix='physics')
    assert epoch == 4 and step == 77 and sd['dx'] == 27000.0
    model.physics_net.load_state_dict(sd['model'], strict=True)
    # a checkpoint written under DistributedDataParallel carries 'module.' prefixes
    torch.save({'model': {'module.' + k: v for k, v in model.physics_net.state_dict().items()}, 'epoch': 0, 'gobal_step': 1},
               tmp_path / 'ddp.pth')
    sd2, _, _ = model.load_model(str(tmp_path / 'ddp.pth'))
    model.physics_net.load_state_dict(sd2['model'], strict=True)


def test_point_config_follows_the_config_file(model):
    cfg = model.point_config()
    assert cfg.factors == (1e3, 1e3, 1e10, 1e1, 1e14, 1e-7)
    ph = cfg.physics()
    assert list(ph.clip_on) == [0, 0, 1, 1, 1, 1]               # u, v never clipped
    assert abs(ph.std[2] - 13296.749084125422) < 1e-2 and ph.clip_hi[2] == 500000.0
    model.with_clip = False
    assert list(model.point_config().physics().clip_on) == [0] * 6
    model.with_clip = True
    # the PDE criterion: the three the reference's loss builder offers (module or config dict) map onto the kernel's (kind, beta); anything else raises
    from deepphysinet_amd import _lib as L
    from deepphysinet_amd.losses import builder_loss
    assert (ph.criterion, ph.beta) == (L.CRIT_MSE, 0.0)
    assert model._check_pde_criterion(torch.nn.L1Loss()) == (L.CRIT_L1, 0.0, False)
    assert model._check_pde_criterion(builder_loss('WeightSmoothL1Loss', beta=0.25)) == (L.CRIT_SMOOTH_L1, 0.25, False)
    assert model._check_pde_criterion(dict(name='WeightSmoothL1Loss', beta=0.5)) == (L.CRIT_SMOOTH_L1, 0.5, False)
    assert model._check_pde_criterion(dict(name='MSELoss', reduction='sum')) == (L.CRIT_MSE, 0.0, True)
    assert model.point_config(criterion=torch.nn.MSELoss(reduction='sum')).physics().reduce_sum == 1
    assert model.point_config(criterion=torch.nn.L1Loss()).physics().criterion == L.CRIT_L1
    for bad in (torch.nn.MSELoss(reduction='none'), torch.nn.HuberLoss(), dict(name='CrossEntropyLoss'), dict(name='MSELoss', reduction='none')):
        with pytest.raises(NotImplementedError):
            model._check_pde_criterion(bad)
    # inverse_norm's other branches as the kernel's affine map (:238-243): use_norm False = identity without clip, two-factor min_max
    import copy
    keep = copy.deepcopy(model.obs_norm_cfg)
    try:
        model.obs_norm_cfg['u10']['use_norm'] = False
        model.obs_norm_cfg['q2']['use_norm'] = False
        model.obs_norm_cfg['pres'].update(norm_type='min_max', norm_factor=[80000.0, 100000.0])
        ph2 = model.point_config().physics()
        assert (ph2.mean[0], ph2.std[0], ph2.mean[2], ph2.std[2]) == (0.0, 1.0, 80000.0, 20000.0)
        assert list(ph2.clip_on) == [0, 0, 1, 1, 0, 1]
        assert ph2.clip_lo[0] < -3e38 and ph2.clip_hi[4] > 3e38          # un-normalised variables: bounds that never bind (full-grid maps too)
        model.obs_norm_cfg['pres']['norm_factor'] = [80000.0, 100000.0, 3.0]      # the squared three-factor form (:244-247)
        ph3 = model.point_config().physics()
        assert (ph3.sq_on[2], ph3.sq_add[2], ph3.std[2]) == (1, 3.0, 20000.0) and list(ph3.sq_on)[:2] == [0, 0]
        # a changed clip BOUND alone rebuilds the cached configuration (the bounds were missing from the cache key)
        model.obs_norm_cfg['t2']['bound'] = [150.0, 400.0]
        ph4 = model.point_config().physics()
        assert (ph4.clip_lo[3], ph4.clip_hi[3]) == (150.0, 400.0)
    finally:
        model.obs_norm_cfg.clear()
        model.obs_norm_cfg.update(keep)
        model.point_config()


# ------------------------------------------------------------------------------------------------ round 2: loops, launcher, arena
def test_training_loop_surface_resolves_like_the_reference(model):
    """train.py:47 calls `run_train_interface(checkpoint_path=..., log_path=...)`; the _dist twin exists too (:848).  Without a sample
    source the loop says what it needs instead of reaching for GeoTIFF files."""
    import inspect
    for name in ('run_train_interface', 'run_train_interface_dist'):
        fn = getattr(model, name)
        assert list(inspect.signature(fn).parameters) == ['kwargs']
    m = builder_models(**ncep_config())
    m.train_cfg['num_epoch'] = 1
    with pytest.raises((RuntimeError, AssertionError), match='samples|HIP|CUDA|cuda'):
        m.run_train_interface(checkpoint_path=None, log_path=None, device='cpu')


def test_lr_schedule_of_the_config_matches_the_reference_sequence(model):
    """cfg:160-165 CosineAnnealingLR(T_max=5, eta_min=5e-6) stepped once per epoch (:831-833); the `verbose=True` the config passes
    (a TypeError under torch >= 2.7) is dropped.  Sequence = SURVEY appendix A probe of the reference."""
    w = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adam([{'params': [w], 'initial_lr': 1e-4}], lr=1e-4)
    model.train_cfg['lr_schedule'] = dict(name='CosineAnnealingLR', T_max=5, eta_min=5e-6, verbose=True)
    sched = model._build_lr_schedule(opt, current_epoch=0)
    seq = []
    for _ in range(6):
        opt.step()
        sched.step()
        seq.append(opt.param_groups[0]['lr'])
    assert np.allclose(seq, [9.09e-5, 6.72e-5, 3.78e-5, 1.41e-5, 5e-6, 1.41e-5], rtol=5e-3)


def test_bench_refuses_more_ranks_than_gpus_and_mismatched_launchers():
    """bench.py --gpus N must start N ranks or fail loudly -- never report one rank's number as N.  This container has no
    GPU: the self-launcher exits non-zero before touching a device; a WORLD_SIZE that contradicts --gpus is refused as well."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'DPN_BENCH_ONE_DEVICE')}
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and 'GPU(s) visible' in r.stderr and '{' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=dict(env, WORLD_SIZE='4', RANK='0'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'disagree' in r.stderr


def test_grad_arena_leases_and_fallbacks():
    """grad_arena on CPU tensors: a slot is handed out once per accumulation window, views alias the owner's flat buffer, foreign tensors
    and dead owners fall back to ordinary allocations."""
    from deepphysinet_amd import grad_arena

    class Owner:
        pass
    o = Owner()
    a, b = torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5))
    o._g_flat, o._leased = torch.zeros(4096), set()
    grad_arena.register(o, [a, b], [0, 2048])
    g1 = grad_arena.new_grad(a)
    assert g1.shape == (3, 4) and g1.data_ptr() == o._g_flat.data_ptr()
    g2 = grad_arena.new_grad(a)                                      # second request in the same window: not the slot
    assert g2.data_ptr() != g1.data_ptr()
    assert grad_arena.new_grad(b.detach()).data_ptr() == o._g_flat[2048:].data_ptr()       # a detached alias of the parameter finds the slot
    assert grad_arena.new_grad(torch.zeros(5)).data_ptr() not in (g1.data_ptr(), o._g_flat[2048:].data_ptr())
    o._leased.clear()                                                # zero_grad(set_to_none=True)
    assert grad_arena.new_grad(a, (12,)).data_ptr() == o._g_flat.data_ptr()
    del o
    import gc
    gc.collect()
    assert grad_arena.slot_of(a) is None
    grad_arena.unregister(None)


def test_resmlp_and_min_max_inverse_norm_follow_the_reference(model):
    """VERDICT r1 surface holes: ResMLP.forward (variable_net.py:22-24: fc(x) + x) is callable, and inverse_norm handles the min_max
    branch (:242-249), both as plain expressions."""
    from deepphysinet_amd.model.variable_net import ResMLP
    torch.manual_seed(0)
    r = ResMLP(16)
    x = torch.randn(5, 16)
    want = r.fc[2](torch.relu(r.fc[0](x))) + x
    assert torch.allclose(r(x), want, atol=1e-6)
    cfg = {k: dict(v) for k, v in model.obs_norm_cfg.items()}
    cfg['t2'] = dict(name='t2', norm_factor=[200.0, 320.0], norm_type='min_max', bound=[50, 500], use_norm=True)
    cfg['q2'] = dict(name='q2', norm_factor=[0.0, 0.2, 1e-5], norm_type='min_max', bound=[1e-6, 10], use_norm=True)
    v = [torch.full((3, 1), 0.5) for _ in range(6)]
    model.with_clip = True
    out = model.inverse_norm(*v, cfg)
    assert torch.allclose(out[3], torch.full((3, 1), 260.0)) and torch.allclose(out[4], torch.full((3, 1), 0.1 ** 2 + 1e-5))
    assert torch.allclose(out[0], torch.full((3, 1), 0.5 * 3.0050219075895894 + 0.14507186950562942))


# ------------------------------------------------------------------------------------------------ round 4: ADVICE r3
def test_training_samples_are_never_synthetic_by_default_and_the_per_rank_protocol_is_explicit(model):
    """ADVICE r3 (medium / low): (a) a loop without a `samples` source raises like the reference does without its data files -- synthetic
    data needs samples='synthetic' (train.py --synthetic); (b) a `samples` callable is called as samples(epoch) and sharded here unless the
    per-rank protocol is selected explicitly (keyword samples_per_rank=True or attribute .per_rank), whatever its signature."""
    with pytest.raises(RuntimeError, match="samples='synthetic'"):
        model._train_samples({}, 0)
    with pytest.raises(ValueError, match='synthetic'):
        model._train_samples({'samples': 'random'}, 0)
    calls = []

    def all_samples(epoch, shuffle=True, seed=0):                     # three positional parameters, but NOT the per-rank protocol
        calls.append((epoch, shuffle, seed))
        return list(range(5))
    assert list(model._epoch_samples({'samples': all_samples}, 7, 1, 2)) == [1, 3, 0]       # sharded here: 1, 3, then the wrapped tail
    assert calls == [(7, True, 0)]

    def mine(epoch, rank, world):
        return ['e%d r%d/%d' % (epoch, rank, world)]
    assert list(model._epoch_samples({'samples': mine, 'samples_per_rank': True}, 2, 1, 4)) == ['e2 r1/4']
    mine.per_rank = True
    assert list(model._epoch_samples({'samples': mine}, 3, 0, 2)) == ['e3 r0/2']


def test_staged_data_parallel_step_is_only_taken_with_the_matching_optimiser_layout():
    """FusedClipAdam records which parameters each bucket of its flat buffers holds (`layout_ids`), so that
    training_step can tell an optimiser laid out like PhysicsNet.gradient_buckets() (staged backward + per-bucket all-reduce) from one
    built without a layout (plain backward + grad_sync(parameters))."""
    import inspect
    from deepphysinet_amd.interface import interface_physics
    src = inspect.getsource(interface_physics.InterfacePhysics.training_step)
    assert "getattr(grad_sync, 'opt', None) is optimizer" in src and 'layout_ids' in src and 'requires_grad' in src
    from deepphysinet_amd import optim
    assert 'self.layout_ids' in inspect.getsource(optim.FusedClipAdam.__init__)


def test_bench_power_sampling_degrades_to_none_without_rocm_smi(monkeypatch, tmp_path):
    """bench.py's clock / socket power samples (rocm-smi) are an extra: without the tool, or when it prints nothing usable, the fields are
    null and the bench line is printed all the same."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import shutil
    calls = []
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    monkeypatch.setattr(shutil, 'which', lambda name: None)
    real_exists = os.path.exists
    monkeypatch.setattr(os.path, 'exists', lambda p: False if str(p).endswith('rocm-smi') else real_exists(p))
    assert bench.sample_power(lambda: calls.append(1), seconds=0.05) is None and not calls      # no tool: the work is not even started
    fake = tmp_path / 'rocm-smi'
    fake.write_text('#!/bin/sh\necho "not json"\n')
    fake.chmod(0o755)
    monkeypatch.setattr(shutil, 'which', lambda name: str(fake))
    monkeypatch.setattr(os.path, 'exists', real_exists)
    assert bench.sample_power(lambda: calls.append(1), seconds=0.2) is None and calls            # garbage output: no samples, no exception
    fake.write_text('#!/bin/sh\necho \'{"card0": {"sclk clock speed:": "(1977Mhz)", "Current Socket Graphics Package Power (W)": "1395.0"}}\'\n')
    out = bench.sample_power(lambda: calls.append(1), seconds=0.5)
    assert out is not None and out['sclk_mhz'] == 1977.0 and out['socket_w'] == 1395.0 and out['samples'] >= 1


def test_scaled_init_is_deterministic_keeps_buffers_and_conditions_the_outputs():
    """deepphysinet_amd.utils.init.scaled_init_ (round 6: what `bench.py --leads N` trains from, section 6a of DESIGN.md): the same seed gives the same weights,
    another seed others; every parameter is redrawn with its scale (LayerNorm gains around 1, the hyper-network heads 8 x smaller than a 1/sqrt(fan_in) layer),
    buffers (the sinusoid table) are untouched -- and the encoder's output, through the heads, gives raw VariableNet weights of the order of 1/sqrt(fan_in): the
    point of it (PyTorch's default initialisation: raw outputs with a standard deviation of 7-14, half of the physical values on a clip bound)."""
    from deepphysinet_amd.utils.init import scaled_init_
    m = builder_models(**ncep_config())
    net = m.physics_net
    pe_before = net.meta_net.model.enc_embedding.position_embedding.pe.clone()
    scaled_init_(net, seed=1)
    a = {k: v.detach().clone() for k, v in net.named_parameters()}
    scaled_init_(net, seed=1)
    assert all(torch.equal(a[k], v.detach()) for k, v in net.named_parameters())
    scaled_init_(net, seed=2)
    assert not torch.equal(a['U_net.out_fc.weight'], dict(net.named_parameters())['U_net.out_fc.weight'].detach())
    assert torch.equal(pe_before, net.meta_net.model.enc_embedding.position_embedding.pe)
    p = dict(net.named_parameters())
    g = p['meta_net.model.encoder.norm.weight'].detach()
    assert 0.9 <= float(g.min()) and float(g.max()) <= 1.1
    w_fc = p['U_net.cat_fc1.fc.0.weight'].detach()                      # [256, 256]: |w| <= 1.7 / 16
    w_head = p['U_net.coord_hidden_fc.weight'].detach()                 # [257, 256]: another factor 8 smaller
    assert float(w_fc.abs().max()) <= 1.7 / 16 + 1e-6 and float(w_fc.abs().max()) > 0.09
    assert float(w_head.abs().max()) <= 1.7 / 16 / 8 + 1e-6
    assert float(p['U_net.out_fc.bias'].detach().abs().max()) <= 0.05 + 1e-6
