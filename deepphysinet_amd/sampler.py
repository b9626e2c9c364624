"""On-device collocation sampler.

Mirrors the three point generators of the reference's dataset (dataset/physics_dataset.py): `get_inter_data` (:431-499),
`get_item_label_data` (:323-429) and `get_margin_grid` (:528-587) -- same names, same return tuples -- but the coarse
forecast cube and the labels are resident in HBM and one HIP kernel (`dpn_sample_points`) draws the points, interpolates
the cube tri-linearly (what `xarray.DataArray.interp` does there, six times per call) and evaluates the Coriolis parameter.
The host is out of the per-step loop.  The draws come from Philox-4x32-10 instead of numpy's global Mersenne state (the
reference's sequence depends on DataLoader worker seeding and is not reproducible either); everything downstream of the
draws is parity-tested against `oracle/sampler_oracle.py`.
"""
import ctypes
from dataclasses import dataclass

import torch

from . import _lib as L
from .point_path import _ptr, _require_gpu, _stream


@dataclass
class SamplerConfig:
    """Grid constants of PhysicsDataset (physics_dataset.py:83-126; configs/DeepPhysiNet_NCEP_cfg.py:93-111)."""
    lon_size: int = 257            # label_lon_size
    lat_size: int = 145            # label_lat_size
    in_lon_size: int = 65          # len(in_lon)
    in_lat_size: int = 37          # len(in_lat)
    input_time_step: int = 6
    input_time_step_nums: int = 4
    out_res_deg: float = 0.25      # spacing of the fine grid (the literal 0.25 of :336-337, :444-445)
    in_res_deg: float = 1.0        # spacing of in_lon / in_lat
    begin_lat: float = 18.0        # out_lat[0]
    dx: float = 27000.0
    dy: float = 27000.0

    def c_struct(self) -> L.DpnSampler:
        ratio = self.out_res_deg / self.in_res_deg
        return L.DpnSampler(self.lon_size, self.lat_size, self.in_lon_size, self.in_lat_size, self.input_time_step_nums + 1,
                            self.input_time_step * self.input_time_step_nums, ratio, ratio, float(self.input_time_step),
                            float(self.begin_lat), float(self.out_res_deg), float(self.dx), float(self.dy))


class CollocationSampler:
    """cube: [6, in_lat, in_lon, t_in] fp32 normalised coarse forecast (obs_name_order u10,v10,pres,t2,q2,rio; the per-variable
    [y, x, t] arrays of physics_dataset.py:400 stacked); labels (optional): [t_hours + 1, 6, lat, lon] fp32 normalised ERA5."""

    def __init__(self, cfg: SamplerConfig, cube: torch.Tensor, labels: torch.Tensor = None, seed: int = 0):
        _require_gpu(cube, 'cube')
        c = cfg
        if tuple(cube.shape) != (6, c.in_lat_size, c.in_lon_size, c.input_time_step_nums + 1):
            raise ValueError('cube must be [6, %d, %d, %d], got %s' % (c.in_lat_size, c.in_lon_size, c.input_time_step_nums + 1, tuple(cube.shape)))
        if (c.lon_size - 1) * c.out_res_deg > (c.in_lon_size - 1) * c.in_res_deg + 1e-9 or \
           (c.lat_size - 1) * c.out_res_deg > (c.in_lat_size - 1) * c.in_res_deg + 1e-9:
            raise ValueError('the fine grid must lie inside the coarse cube')
        self.cfg, self._s = cfg, cfg.c_struct()
        self.cube = cube.detach().float().contiguous()
        self.labels = None
        if labels is not None:
            _require_gpu(labels, 'labels')
            hours = c.input_time_step * c.input_time_step_nums + 1
            if tuple(labels.shape) != (hours, 6, c.lat_size, c.lon_size):
                raise ValueError('labels must be [%d, 6, %d, %d]' % (hours, c.lat_size, c.lon_size))
            self.labels = labels.detach().float().contiguous()
        self.seed, self.offset = int(seed), 0
        self._step_dev, self._stride = None, 0

    def bind_step_counter(self, step_count: torch.Tensor, points_per_step: int):
        """Draw from the DEVICE-side step counter: after this, every draw uses Philox counter = (offset inside the step) +
        step_count * points_per_step, step_count being read by the kernel (FusedClipAdam.step_count: int32[1] on the device, bumped
        once per optimiser step).  A sampler launch captured in a hipGraph then draws fresh points on every replay; host-side state no
        longer advances between steps (begin_step() rewinds the offset inside the step; training_batch() calls it)."""
        _require_gpu(step_count, 'step_count')
        if step_count.dtype != torch.int32 or step_count.numel() != 1:
            raise ValueError('step_count must be one int32 on the device')
        self._step_dev, self._stride, self.offset = step_count, int(points_per_step), 0

    def begin_step(self):
        if self._step_dev is not None:
            self.offset = 0

    def _run(self, mode, n, xi=None, yi=None, ti=None, want_labels=False, want_raw=False):
        dev = self.cube.device
        x, y, t, f = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
        cd = torch.empty((n, 6), dtype=torch.float32, device=dev)
        lab = torch.empty((n, 6), dtype=torch.float32, device=dev) if want_labels else None
        raw = torch.empty((n, 3), dtype=torch.float64, device=dev) if want_raw else None
        if want_labels and self.labels is None:
            raise RuntimeError('this sampler was built without labels')
        lib = L.load()
        if self._step_dev is not None and mode != L.SAMPLE_EXPLICIT and self.offset + n > self._stride:
            raise RuntimeError('more points drawn in one step (%d) than bind_step_counter() reserved (%d)' % (self.offset + n, self._stride))
        L.check(lib.dpn_sample_points_replay(ctypes.byref(self._s), _ptr(self.cube), _ptr(self.labels) if want_labels else None, mode,
                                             _ptr(xi), _ptr(yi), _ptr(ti), n, self.seed, self.offset, _ptr(self._step_dev), self._stride,
                                             _ptr(x), _ptr(y), _ptr(t), _ptr(f), _ptr(cd), _ptr(lab), _ptr(raw), _stream()),
                'dpn_sample_points_replay')
        if mode != L.SAMPLE_EXPLICIT:
            self.offset += n                                   # the next call continues the Philox counter
        return x, y, t, f, cd, lab, raw

    def get_inter_data(self, n: int = 4096, with_raw: bool = False):
        """-> inter_x, inter_y, inter_t, inter_data, inter_f  (physics_dataset.py:499); shapes [n], [n], [n], [n,6], [n,1]."""
        x, y, t, f, cd, _, raw = self._run(L.SAMPLE_INTERIOR, n, want_raw=with_raw)
        out = (x, y, t, cd, f.unsqueeze(1))
        return out + (raw,) if with_raw else out

    def get_item_label_data(self, n: int = 20480, with_raw: bool = False):
        """-> margin_x, margin_y, margin_t, margin_data, margin_f, margin_input_data  (physics_dataset.py:429)."""
        x, y, t, f, cd, lab, raw = self._run(L.SAMPLE_MARGIN, n, want_labels=self.labels is not None, want_raw=with_raw)
        out = (x, y, t, lab, f.unsqueeze(1), cd)
        return out + (raw,) if with_raw else out

    def get_margin_grid(self, margin_x_list, margin_y_list, margin_t_list):
        """-> inter_x, inter_y, inter_t, inter_data, inter_f for caller-given node indices and hours (physics_dataset.py:587)."""
        dev = self.cube.device
        xi, yi, ti = (torch.as_tensor(v, dtype=torch.int32, device=dev).contiguous() for v in (margin_x_list, margin_y_list, margin_t_list))
        c = self.cfg
        if xi.numel() and (int(xi.min()) < 0 or int(xi.max()) >= c.lon_size or int(yi.min()) < 0 or int(yi.max()) >= c.lat_size
                           or int(ti.min()) < 0 or int(ti.max()) > c.input_time_step * c.input_time_step_nums):
            raise IndexError('grid node / hour outside the domain')
        x, y, t, f, cd, _, _ = self._run(L.SAMPLE_EXPLICIT, xi.numel(), xi, yi, ti)
        return x, y, t, cd, f.unsqueeze(1)

    def full_grid(self, time_id: int):
        """All lon*lat nodes at one hour in the reference's visualisation order (x outer, y inner; interface_physics.py:538-543)."""
        c = self.cfg
        dev = self.cube.device
        xs = torch.arange(c.lon_size, dtype=torch.int32, device=dev).repeat_interleave(c.lat_size)
        ys = torch.arange(c.lat_size, dtype=torch.int32, device=dev).repeat(c.lon_size)
        ts = torch.full_like(xs, int(time_id))
        return self.get_margin_grid(xs, ys, ts)

    def training_batch(self, field_data, forecast_h, n_margin: int = 20480, n_inter: int = 4096):
        """One sample of PhysicsDataset.__getitem__ (physics_dataset.py:501-519) as the dict InterfacePhysics.training_step takes: the
        field sample and lead time given by the caller, 20 480 margin (grid-node, labelled) and 4 096 interior collocation points drawn,
        interpolated and labelled on the device (batch sizes: cfg:111, physics_dataset.py:30)."""
        self.begin_step()
        mx, my, mt, mlab, mf, mcd = self.get_item_label_data(n_margin)
        ix, iy, it, icd, if_ = self.get_inter_data(n_inter)
        col = lambda v: v.reshape(-1, 1)
        return {'field_data': field_data, 'forecast_h': forecast_h,
                'margin_x': col(mx), 'margin_y': col(my), 'margin_t': col(mt), 'margin_f': mf, 'margin_data': mlab, 'margin_input_data': mcd,
                'inter_x': col(ix), 'inter_y': col(iy), 'inter_t': col(it), 'inter_f': if_, 'inter_data': icd}



class SyntheticSamples:
    """Default `samples` source of run_train_interface when the configuration names none: one epoch = the 61 six-hourly lead times of the
    reference's file map (tools/generate_input_map.py:41), each a synthetic field sample [1,159,2405] (normalised forecasts ~ N(0,1), the
    four constant rows ~ U[0,1]: SURVEY 8d) whose collocation batch is drawn by the on-device CollocationSampler from synthetic coarse /
    label cubes of the configured shapes.  It stands in for PhysicsDataset's GeoTIFF / xarray reader (dataset/physics_dataset.py: file I/O
    outside this build, no data offline) so that the reference's two-keyword call `run_train_interface(checkpoint_path=, log_path=)`
    (train.py:47) runs end to end.  A sequence: data-parallel ranks index only their own samples."""

    def __init__(self, device, n_margin=20480, n_inter=4096, leads=61, seed=0, lat=145, lon=257):
        g = torch.Generator().manual_seed(seed)
        self.device = torch.device(device)
        cube = torch.randn(6, 37, 65, 5, generator=g).to(self.device)
        labels = torch.randn(25, 6, lat, lon, generator=g).to(self.device)
        self.sampler = CollocationSampler(SamplerConfig(), cube, labels, seed=seed + 1)
        self.n_margin, self.n_inter, self.leads, self.seed = int(n_margin), int(n_inter), int(leads), int(seed)

    def __len__(self):
        return self.leads

    def __getitem__(self, i):
        if not 0 <= i < self.leads:
            raise IndexError(i)
        g = torch.Generator().manual_seed(self.seed * 1000 + 17 + i)
        field = torch.randn(1, 159, 2405, generator=g)
        field[:, 155:, :] = torch.rand(1, 4, 2405, generator=g)
        fh = torch.full((1, 1, 1), 6.0 * i / 360.0)
        return self.sampler.training_batch(field.to(self.device), fh.to(self.device), n_margin=self.n_margin, n_inter=self.n_inter)
