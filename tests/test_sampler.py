"""Collocation sampler + full-grid gather.

CPU part: the oracle restatement against closed forms (a tri-linear function is reproduced exactly by linear interpolation).
GPU part: `dpn_sample_points` / `dpn_grid_maps` through the C ABI against the oracle on the kernel's own draws, plus
distributional properties of the draws at the reference's batch sizes (4 096 interior, 20 480 margin; physics_dataset.py:30).
Tolerance: interpolation happens in fp64 on both sides and is cast to fp32 once -> 1 ulp (2e-7 relative); coordinates bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import sampler_oracle as SO

IN_LON = 72.0 + np.arange(65) * 1.0
IN_LAT = 18.0 + np.arange(37) * 1.0


def _cube(seed=3):
    g = np.random.default_rng(seed)
    return g.standard_normal((6, 37, 65, 5)).astype(np.float32)


def test_oracle_reproduces_a_trilinear_function_and_nan_outside():
    lat, lon, tt = np.meshgrid(IN_LAT, IN_LON, np.arange(5) * 6.0, indexing='ij')
    cube = np.stack([(0.3 * lat - 0.2 * lon + 0.05 * tt + 0.001 * lat * lon - 0.002 * lon * tt + k) for k in range(6)]).astype(np.float64)
    g = np.random.default_rng(0)
    xr, yr, tr = g.random(500) * 256, g.random(500) * 144, g.integers(0, 25, 500)
    x, y, t, data, f = SO.points_from_draws(cube, xr, yr, tr, 72.0, 18.0, IN_LON, IN_LAT, 6, 27000.0, 27000.0)
    la, lo = 18.0 + yr * 0.25, 72.0 + xr * 0.25
    want = np.stack([(0.3 * la - 0.2 * lo + 0.05 * tr + 0.001 * la * lo - 0.002 * lo * tr + k) for k in range(6)], axis=1)
    np.testing.assert_allclose(data, want.astype(np.float32), rtol=2e-6)
    np.testing.assert_allclose(f[:, 0], (2 * 7.29e-5 * np.sin(la / 180 * np.pi)).astype(np.float32), rtol=0, atol=0)
    assert x.dtype == np.float32 and np.all(t == tr * 3600)
    _, _, _, outside, _ = SO.points_from_draws(cube, [260.0], [3.0], [2], 72.0, 18.0, IN_LON, IN_LAT, 6, 27000.0, 27000.0)
    assert np.all(np.isnan(outside))


def test_oracle_grid_maps_order():
    lon, lat = 5, 3
    out = np.arange(lon * lat * 6, dtype=np.float32).reshape(lon * lat, 6)
    maps = SO.grid_maps(out, lon, lat, [0] * 6, [1] * 6, [-1e9] * 6, [1e9] * 6, True)
    for xx in range(lon):
        for yy in range(lat):
            assert maps[2, yy, xx] == out[xx * lat + yy, 2]          # result_P[y, x] = inter_P[id], id = x * lat + y


def test_oracle_coriolis_matches_reference_vectors(golden_dir):
    """Fixture F11 = PhysicsDataset.get_coriolis of the reference itself (dataset/physics_dataset.py:521-526) on the latitude forms its
    callers build: pins the oracle's restatement bit-for-bit (fp64) and through the pipeline's float32 cast."""
    import os
    d = np.load(os.path.join(golden_dir, 'f11_coriolis.npz'))
    assert np.array_equal(SO.coriolis(d['lat_nodes']), d['f_nodes']) and SO.coriolis(d['lat_nodes']).shape == (145, 1)
    assert np.array_equal(SO.coriolis(d['lat_cont']), d['f_cont'])
    assert np.array_equal(SO.coriolis(d['lat_cont'][:7].reshape(7, 1)), d['f_2d'])
    assert np.array_equal(SO.coriolis(d['lat_cont']).astype(np.float32), d['f_cont_f32'])
    # the point generators: lat = begin_lat + y_rand * 0.25 (:336-337, :444-445)
    _, _, _, _, f = SO.points_from_draws(_cube().astype(np.float64), np.zeros(512), d['y_cont'], np.zeros(512), 72.0, 18.0, IN_LON, IN_LAT, 6,
                                         27000.0, 27000.0)
    assert np.array_equal(f, d['f_cont_f32'])


# ------------------------------------------------------------------------------------------------ GPU
def _sampler(seed=11, with_labels=True):
    from deepphysinet_amd.sampler import CollocationSampler, SamplerConfig
    dev = torch.device('cuda:0')
    cube = _cube()
    labels = np.random.default_rng(5).standard_normal((25, 6, 145, 257)).astype(np.float32) if with_labels else None
    s = CollocationSampler(SamplerConfig(), torch.from_numpy(cube).to(dev), None if labels is None else torch.from_numpy(labels).to(dev), seed=seed)
    return s, cube, labels


def _check_against_oracle(cube, x, y, t, cd, f, raw):
    raw = raw.cpu().numpy()
    ox, oy, ot, od, of = SO.points_from_draws(cube, raw[:, 0], raw[:, 1], raw[:, 2], 72.0, 18.0, IN_LON, IN_LAT, 6, 27000.0, 27000.0)
    np.testing.assert_array_equal(x.cpu().numpy(), ox)                # metres: one fp64 product cast to fp32 on both sides
    np.testing.assert_array_equal(y.cpu().numpy(), oy)
    np.testing.assert_array_equal(t.cpu().numpy(), ot)
    np.testing.assert_allclose(cd.cpu().numpy(), od, rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(f.cpu().numpy(), of, rtol=2e-7, atol=0)


@pytest.mark.gpu
def test_interior_points_match_oracle_and_are_uniform():
    s, cube, _ = _sampler()
    n = 4096
    x, y, t, cd, f, raw = s.get_inter_data(n, with_raw=True)
    assert cd.shape == (n, 6) and f.shape == (n, 1)
    _check_against_oracle(cube, x, y, t, cd, f, raw)
    r = raw.cpu().numpy()
    assert r[:, 0].min() >= 0 and r[:, 0].max() < 256 and r[:, 1].min() >= 0 and r[:, 1].max() < 144
    assert set(np.unique(r[:, 2])) <= set(range(25)) and len(np.unique(r[:, 2])) == 25
    for col, hi in ((0, 256.0), (1, 144.0)):                           # uniform: mean and variance of U[0, hi)
        assert abs(r[:, col].mean() - hi / 2) < 4 * hi / np.sqrt(12 * n)
        assert abs(r[:, col].var() - hi * hi / 12) < 0.1 * hi * hi / 12
    assert np.abs(np.corrcoef(r.T)[np.triu_indices(3, 1)]).max() < 0.06
    x2, *_ = s.get_inter_data(n)                                       # the counter advances: a fresh batch
    assert not torch.equal(x, x2)
    s2, _, _ = _sampler()                                              # same seed -> same stream
    assert torch.equal(s2.get_inter_data(n)[0], x)


@pytest.mark.gpu
def test_device_coriolis_matches_reference_vectors(golden_dir):
    """The f column of the device sampler at every latitude row of the grid against fixture F11 (the reference's get_coriolis output
    cast to float32 as __getitem__ does, physics_dataset.py:517-519): one ulp (the kernel evaluates sin in fp64 and casts once)."""
    import os
    d = np.load(os.path.join(golden_dir, 'f11_coriolis.npz'))
    s, _, _ = _sampler(with_labels=False)
    yi = d['y_nodes'].astype(np.int32)
    _, _, _, _, f = s.get_margin_grid(np.zeros_like(yi), yi, np.zeros_like(yi))
    np.testing.assert_allclose(f.cpu().numpy(), d['f_nodes_f32'], rtol=1.2e-7, atol=0)


@pytest.mark.gpu
def test_margin_points_are_grid_nodes_with_labels():
    s, cube, labels = _sampler(seed=2)
    n = 20480
    x, y, t, lab, f, cd, raw = s.get_item_label_data(n, with_raw=True)
    _check_against_oracle(cube, x, y, t, cd, f, raw)
    r = raw.cpu().numpy()
    assert np.all(r == np.floor(r)) and r[:, 0].max() == 256 and r[:, 1].max() == 144 and r[:, 2].max() == 24 and r.min() == 0
    np.testing.assert_array_equal(lab.cpu().numpy(), SO.labels_at(labels, r[:, 0], r[:, 1], r[:, 2]))
    counts = np.bincount(r[:, 2].astype(int), minlength=25)           # 25 equiprobable hours: chi-square, 24 dof, p ~ 1e-4 bound
    assert ((counts - n / 25) ** 2 / (n / 25)).sum() < 60.0


@pytest.mark.gpu
def test_explicit_grid_and_full_grid_maps():
    from deepphysinet_amd.configs import ncep_config
    from deepphysinet_amd.interface import builder_models
    from oracle.fill import fill_state_dict_, synthetic_inputs
    s, cube, _ = _sampler(with_labels=False)
    x, y, t, cd, f = s.full_grid(7)
    n = 257 * 145
    assert x.shape == (n,) and float(t[0]) == 7 * 3600.0
    xi, yi = np.repeat(np.arange(257), 145), np.tile(np.arange(145), 257)
    ox, oy, ot, od, of = SO.points_from_draws(cube, xi, yi, np.full(n, 7), 72.0, 18.0, IN_LON, IN_LAT, 6, 27000.0, 27000.0)
    np.testing.assert_array_equal(x.cpu().numpy(), ox)
    np.testing.assert_array_equal(y.cpu().numpy(), oy)
    np.testing.assert_allclose(cd.cpu().numpy(), od, rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(f.cpu().numpy(), of, rtol=2e-7)
    with pytest.raises(IndexError):
        s.get_margin_grid([257], [0], [0])
    # full-grid prediction: maps == de-normalised point fields scattered as the reference's loop does (:583-591)
    m = builder_models(**ncep_config(), precision='bf16x2')
    sd = m.physics_net.state_dict()
    fill_state_dict_(sd, gain=1.0)
    m.physics_net.load_state_dict(sd)
    m = m.to(x.device)
    inp = synthetic_inputs(8, 257, 145, 27000.0, 27000.0)
    field, fh = inp['field_data'].to(x.device), inp['forecast_h'].to(x.device)
    for clip in (False, True):
        maps = m.predict_grid(field, x, y, t, cd, fh, with_clip=clip)
        assert maps.shape == (6, 145, 257)
        with torch.no_grad():
            fields = torch.cat(m.physics_net.forward_xyt(field, x, y, t, cd, fh), dim=1).cpu().numpy()
        cfg = m.point_config()
        want = SO.grid_maps(fields, 257, 145, cfg.mean, cfg.std, cfg.clip_lo, cfg.clip_hi, clip)
        np.testing.assert_array_equal(maps.cpu().numpy(), want)     # same two fp32 roundings on both sides: bit-exact


@pytest.mark.gpu
def test_sampler_feeds_the_training_step():
    """Dataset -> step entirely on the device: a batch drawn by the sampler (reference sizes: 20 480 margin + 4 096 interior points) through
    InterfacePhysics.training_step (data loss + both PDE losses + clip + Adam): finite losses, parameters move, fresh points every call."""
    from deepphysinet_amd.configs import ncep_config
    from deepphysinet_amd.interface import builder_models
    from deepphysinet_amd.optim import FusedClipAdam
    from oracle.fill import fill_state_dict_, synthetic_inputs
    s, _, _ = _sampler(seed=4)
    dev = s.cube.device
    m = builder_models(**ncep_config(), precision='bf16')
    sd = m.physics_net.state_dict()
    fill_state_dict_(sd, gain=1.0)
    m.physics_net.load_state_dict(sd)
    m = m.to(dev)
    opt = FusedClipAdam(m.physics_net.parameters(), lr=1e-4, weight_decay=1e-4)
    inp = synthetic_inputs(8, 257, 145, 27000.0, 27000.0)
    field, fh = inp['field_data'].to(dev), inp['forecast_h'].to(dev)
    before = m.physics_net.U_net.out_fc.weight.detach().clone()
    b1 = s.training_batch(field, fh)
    assert b1['margin_x'].shape == (20480, 1) and b1['inter_data'].shape == (4096, 6) and b1['margin_data'].shape == (20480, 6)
    loss, parts, gnorm = m.training_step(b1, opt, with_pde=True)
    assert torch.isfinite(loss) and all(torch.isfinite(v) for v in parts.values()) and torch.isfinite(gnorm).all()
    assert set(parts) == {'margin_loss', 'inter_pde_loss', 'margin_pde_loss'}
    assert not torch.equal(before, m.physics_net.U_net.out_fc.weight.detach())
    b2 = s.training_batch(field, fh)
    assert not torch.equal(b1['inter_x'], b2['inter_x'])


@pytest.mark.gpu
def test_captured_sampler_draws_fresh_points_on_every_replay():
    """The sampler inside a hipGraph: bound to the optimiser's device-side step counter, a captured training_batch() draws the points of
    step k on replay k -- the same points an unbound sampler draws with the host-side offset k * points_per_step."""
    from deepphysinet_amd.optim import FusedClipAdam
    from oracle.fill import synthetic_inputs
    n_m, n_i = 2048, 512
    s, _, _ = _sampler(seed=11)
    dev = s.cube.device
    p = torch.nn.Parameter(torch.zeros(8, device=dev))
    p.grad = torch.ones_like(p)
    opt = FusedClipAdam([p], lr=1e-3, weight_decay=0.0)
    s.bind_step_counter(opt.step_count, n_m + n_i)
    inp = synthetic_inputs(8, 257, 145, 27000.0, 27000.0)
    field, fh = inp['field_data'].to(dev), inp['forecast_h'].to(dev)

    def step():
        b = s.training_batch(field, fh, n_margin=n_m, n_inter=n_i)
        opt.step()
        return b

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                               # eager step 0 (warm-up on the capture stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        b = step()
    seen = []
    for _ in range(3):                                       # replays = steps 1, 2, 3
        g.replay()
        torch.cuda.synchronize()
        seen.append({k: b[k].clone() for k in ('margin_x', 'margin_t', 'margin_data', 'inter_x', 'inter_y', 'inter_data')})
    assert int(opt.step_count) == 4
    assert not torch.equal(seen[0]['inter_x'], seen[1]['inter_x']) and not torch.equal(seen[1]['margin_x'], seen[2]['margin_x'])
    ref, _, _ = _sampler(seed=11)                            # host-side offsets: step k starts at k * (n_m + n_i)
    for k, got in enumerate(seen, start=1):
        ref.offset = k * (n_m + n_i)
        want = ref.training_batch(field, fh, n_margin=n_m, n_inter=n_i)
        for key, v in got.items():
            assert torch.equal(v, want[key]), (k, key)
    with pytest.raises(RuntimeError):                        # more points than the stride reserved
        s.begin_step()
        s.get_inter_data(n_m + n_i + 1)
