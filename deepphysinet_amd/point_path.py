"""Host side of the HIP point path: buffers, pointer tables and the two autograd entry points.

`point_fields`  : six VariableNets evaluated at N collocation points (PhysicsNet.forward's hot part,
                  reference model/physics_net.py:49-54), differentiable w.r.t. all weights.
`pde_losses`    : place_one_batch's hot part (reference interface/interface_physics.py:278-301): fields,
                  6x3 coordinate Jacobian, inverse_norm (+clip), six residual losses; backward gives every
                  weight gradient without building an autograd graph over the points.

PyTorch is used for device memory, the current HIP stream and autograd bookkeeping only; all arithmetic on
points happens in libdpn_hip.so (deepphysinet_amd/csrc/dpn_kernels.hip).  There is no CPU path.
"""
import ctypes
import os
from dataclasses import dataclass, field
from typing import Optional, Sequence

import torch

from . import _lib as L
from . import branch, config
from .grad_arena import new_grad, slot_of

# configs/DeepPhysiNet_NCEP_cfg.py:64-76 -- order u10, v10, pres, t2, q2, rio (network output order)
OBS_ORDER = ('u10', 'v10', 'pres', 't2', 'q2', 'rio')
LOSS_ORDER = ('motion_u_factor', 'motion_v_factor', 'continuous_factor', 'energy_factor', 'vapor_factor', 'gas_factor')
STATIC_NAMES = ('Wd', 'bd', 'W1', 'bf1', 'W2', 'bf2', 'wo', 'bo')      # per-net parameter tensors, in this order
STATIC_SHAPES = ((256, 192), (256,), (256, 256), (256,), (256, 256), (256,), (1, 256), (1,))


@dataclass
class PointConfig:
    """Geometry + physics constants of one InterfacePhysics instance."""
    dx: float = 27000.0
    dy: float = 27000.0
    lon_size: int = 257
    lat_size: int = 145
    pred_t_span: float = 86400.0
    mean: Sequence[float] = (0.14507186950562942, -0.17325370241478535, 89741.36105771353, 283.58054561520305,
                             0.007909478276582905, 1.0966503643401704)
    std: Sequence[float] = (3.0050219075895894, 3.006602165591562, 13296.749084125422, 15.583177935722373,
                            0.006304067969976075, 0.15166081218127583)
    clip_lo: Sequence[float] = (-500.0, -500.0, 10000.0, 50.0, 1e-6, 1e-6)
    clip_hi: Sequence[float] = (500.0, 500.0, 500000.0, 500.0, 10.0, 10.0)
    with_clip: bool = True
    clip_vars: Sequence[bool] = (False, False, True, True, True, True)      # which variables torch.clip applies to when with_clip (never u, v; not a variable
                                                                            # whose obs_norm_cfg says use_norm: False -- interface_physics.py:236-257)
    factors: Sequence[float] = (1.e3, 1.e3, 1.e10, 1e1, 1.e14, 1.e-7)
    prec: int = L.PREC_BF16X2
    sq_add: Sequence[Optional[float]] = (None,) * 6      # three-factor min_max (:244-247): (out * std + mean) ** 2 + sq_add; None: the affine forms
    criterion: int = L.CRIT_MSE            # the PDE criterion (train_cfg.losses.pde_loss): MSELoss | L1Loss | WeightSmoothL1Loss(beta)
    beta: float = 0.0
    reduce_sum: bool = False               # the criterion's reduction: "mean" (shipped) or "sum"

    def geometry(self) -> L.DpnGeometry:
        return L.DpnGeometry(float(self.dx), float(self.dy), float(self.lon_size - 1), float(self.lat_size - 1), float(self.pred_t_span))

    def physics(self) -> L.DpnPhysics:
        ph = L.DpnPhysics()
        for k in range(L.NETS):
            ph.mean[k], ph.std[k] = float(self.mean[k]), float(self.std[k])
            ph.clip_lo[k], ph.clip_hi[k] = float(self.clip_lo[k]), float(self.clip_hi[k])
            ph.clip_on[k] = int(bool(self.with_clip) and k >= 2 and bool(self.clip_vars[k]))     # u, v are never clipped (interface_physics.py:256-257)
            ph.factor[k] = float(self.factors[k])
        ph.criterion, ph.beta, ph.reduce_sum = int(self.criterion), float(self.beta), int(bool(self.reduce_sum))
        for k in range(L.NETS):
            ph.sq_on[k], ph.sq_add[k] = (0, 0.0) if self.sq_add[k] is None else (1, float(self.sq_add[k]))
        return ph


_freq_cache = {}


def _freqs(device):
    key = str(device)
    if key not in _freq_cache:
        f32 = 2.0 ** torch.linspace(0.0, 4.0, steps=32)       # fp32, exactly utils/position_encoding.py:27
        f16 = 2.0 ** torch.linspace(0.0, 4.0, steps=16)
        _freq_cache[key] = torch.cat([f32, f16]).to(device=device, dtype=torch.float32).contiguous()
    return _freq_cache[key]


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError('deepphysinet_amd point path needs HIP device tensors (%s is on %s); there is no CPU fallback' % (name, t.device))


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    """Device pointer for the C ABI.  Every tensor handed to the library goes through here: a host tensor raises instead of reaching a kernel."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('deepphysinet_amd: a %s tensor on %s was passed to a HIP kernel; there is no CPU fallback' % (tuple(t.shape), t.device))
    return ctypes.c_void_p(t.data_ptr())


_ones = {}


def _one(device):
    """A persistent device scalar 1.0 (the unit cotangent of a total loss)."""
    k = (device.type, device.index)
    if k not in _ones:
        _ones[k] = torch.ones(1, dtype=torch.float32, device=device)
    return _ones[k]


HEADS_COLS = 6 * 193 + 6 * 257          # one GEMM output row: [w1b1 of net 0..5 | w2b2 of net 0..5]
HEADS_W2_OFF = 6 * 193


def _net_ptrs(heads, evec, statics, cls=None):
    """Pointer table over the hyper-network output `heads` [256, 2700] (row stride 2700), evec [6,256] and the 48 static tensors."""
    arr = ((cls or L.DpnNetPtrs) * L.NETS)()
    for k in range(L.NETS):
        arr[k].w1b1 = heads.data_ptr() + k * 193 * 4
        arr[k].w2b2 = heads.data_ptr() + (HEADS_W2_OFF + k * 257) * 4
        arr[k].ld_w1b1 = HEADS_COLS
        arr[k].ld_w2b2 = HEADS_COLS
        arr[k].evec = evec.data_ptr() + k * 256 * 4
        for j, nm in enumerate(STATIC_NAMES):
            setattr(arr[k], nm, statics[k * 8 + j].data_ptr())
    return arr


class _Workspace:
    """Buffers of one forward call that its backward needs again."""

    def __init__(self, n, prec, device, packed=None):
        lib = L.load()
        self.sizes = L.DpnSizes()
        L.check(lib.dpn_sizes(n, prec, ctypes.byref(self.sizes)), 'dpn_sizes')
        self.n, self.prec, self.device = n, prec, device
        # packed: this field's block of a batch packed in ONE launch (_pack_batch); the forward call then has no packing launch of its own
        self.prepacked = packed is not None
        self.packed = packed if packed is not None else torch.empty(self.sizes.packed, dtype=torch.uint8, device=device)
        self.saved = None

    def alloc_saved(self):
        self.saved = torch.empty(self.sizes.saved, dtype=torch.uint8, device=self.device)


class KernelClock:
    """Measurement aid (bench.py): device-clock stamps around the three point kernels INSIDE a captured step.  While `point_path.clock` holds one,
    every launch of dpn_fwd / dpn_bwd_points / dpn_wgrad through this module is bracketed by two dpn_clock_stamp nodes (one-thread kernels appending
    wall_clock64 to a ring); `durations()` turns the ring into per-kernel microseconds.  The product path runs with `clock = None`."""
    NAMES = ('pair', 'fwd', 'bwd', 'wgrad')          # 'pair': two stamps back to back -- the stamp's own cost, subtracted from the others

    def __init__(self, device, cap=4096):
        self.cap = cap
        self.ring = torch.zeros(cap, dtype=torch.int64, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=device)
        khz = ctypes.c_int(0)
        L.check(L.load().dpn_clock_rate_khz(ctypes.byref(khz)), 'dpn_clock_rate_khz')
        self.khz = khz.value
        self.order = []                               # names in stamp order within one replay (recorded while capturing)
        self.recording = True

    def stamp(self, name):
        if self.recording:
            self.order.append(name)
        L.check(L.load().dpn_clock_stamp(_ptr(self.ring), _ptr(self.cursor), self.cap, _stream()), 'dpn_clock_stamp')

    def reset(self):
        self.cursor.zero_()
        self.recording = False

    def durations(self):
        """{name: [us per replay]} from the stamps written since reset(); the 'pair' interval of the same replay is subtracted from the kernels'."""
        n = int(self.cursor.item())
        per = len(self.order)
        if per == 0 or n == 0 or n > self.cap or n % per:
            return {}
        t = self.ring[:n].cpu().view(-1, per).double() * (1e3 / self.khz)          # us
        out = {}
        for name in self.NAMES:
            idx = [i for i, nm in enumerate(self.order) if nm == name]
            if len(idx) == 2:
                out[name] = (t[:, idx[1]] - t[:, idx[0]]).tolist()
        pair = out.get('pair')
        if pair:
            for name in ('fwd', 'bwd', 'wgrad'):
                if name in out:
                    out[name] = [v - p_ for v, p_ in zip(out[name], pair)]
        return out


clock = None            # a KernelClock while bench.py captures its instrumented copy of the step


def _forward_points(cfg: PointConfig, ws: _Workspace, nets, x, y, t, pe_in, coord_data, want_jac, want_saved, ref6=None):
    lib = L.load()
    n = coord_data.shape[0]
    dev = coord_data.device
    # the packed form the forward launch reads: fused (A = W1 w2, B = W1 Wd: five GEMMs per point and net) for the hi+lo mode's tile-split kernel,
    # the plain matrices for the ring kernels (plain bf16, caller-encoded coordinates)
    form = lib.dpn_fwd_form(cfg.prec, 0 if pe_in is None else 1)
    if not ws.prepacked:
        L.check(lib.dpn_pack_weights_form(nets, cfg.prec, form, _ptr(ws.packed), _stream()), 'dpn_pack_weights')
    out_n = torch.empty((n, 6), dtype=torch.float32, device=dev)
    # with caller-encoded coordinates the kernel hands back d out / d pe_in [n, 6, 192] in place of the (x, y, t) Jacobian
    jac_n = torch.empty((n, 6, 3 if pe_in is None else 192), dtype=torch.float32, device=dev) if want_jac else None
    if want_saved:
        ws.alloc_saved()
    geo = cfg.geometry()
    # ref6 [N,6]: the reference's separate ref_data argument (VariableNet.forward standalone); None: coord_data's columns (PhysicsNet.forward)
    n_nets = int(getattr(cfg, 'n_nets', 6) or 6)
    if n_nets < 6 and not want_saved and not want_jac:             # VariableNet.forward standalone, inference: only the nets that are asked for
        out_n.zero_()
        L.check(lib.dpn_fwd_ref_nets(_ptr(x), _ptr(y), _ptr(t), _ptr(pe_in), _ptr(coord_data), _ptr(ref6), n, _ptr(_freqs(dev)), ctypes.byref(geo),
                                     _ptr(ws.packed), cfg.prec, n_nets, _ptr(out_n), None, _stream()), 'dpn_fwd_ref_nets')
        return out_n, jac_n
    if clock is not None:
        clock.stamp('pair'), clock.stamp('pair'), clock.stamp('fwd')
    L.check(lib.dpn_fwd_ref(_ptr(x), _ptr(y), _ptr(t), _ptr(pe_in), _ptr(coord_data), _ptr(ref6), n, _ptr(_freqs(dev)), ctypes.byref(geo),
                            _ptr(ws.packed), cfg.prec, _ptr(out_n), _ptr(jac_n), _ptr(ws.saved), _stream()), 'dpn_fwd')
    if clock is not None:
        clock.stamp('fwd')
    return out_n, jac_n


def _backward_points(cfg: PointConfig, ws: _Workspace, nets, x, y, t, pe_in, coord_data, g_out, g_jxi, statics, into=None, fork=False, keep=(), g_scale=None):
    """Weight gradients from per-point cotangents.  Returns (g_heads [256,2700], g_evec [6,256], [48 static grads]); `into` = the same
    triple preallocated by the caller (a batch of fields writes each field's gradients side by side).

    fork (callers inside an autograd backward pass): the finish stage's second half -- G = S1 w2^T + S2 Wd^T + ..., the gradients of
    cat_fc1.fc.0 / fc.2 and out_fc, static tensors that nothing in the backward pass reads -- goes to the side branch (branch.py) and runs
    beside the hyper-network's and the encoder's backward; `keep` = what that branch reads through raw pointers besides the buffers here.

    (Round 3 had tried a two-stream form with half of dpn_wgrad itself on the side stream: no gain, profiles/round3_two_stream_wgrad.txt.)"""
    lib = L.load()
    n = coord_data.shape[0]
    dev = coord_data.device
    operands = torch.empty(ws.sizes.operands, dtype=torch.uint8, device=dev)
    partials = torch.empty(ws.sizes.partials, dtype=torch.uint8, device=dev)
    geo = cfg.geometry()
    # g_scale: a device scalar multiplied into both cotangent streams as stage 1 reads them (they were formed for a unit cotangent of the total)
    if clock is not None:
        clock.stamp('bwd')
    L.check(lib.dpn_bwd_points_scaled(_ptr(x), _ptr(y), _ptr(t), _ptr(pe_in), _ptr(coord_data), n, _ptr(_freqs(dev)), ctypes.byref(geo),
                                      _ptr(ws.packed), cfg.prec, _ptr(g_out), _ptr(g_jxi), _ptr(g_scale), _ptr(ws.saved), _ptr(operands), _stream()),
            'dpn_bwd_points')
    if clock is not None:
        clock.stamp('bwd')
    arena = False
    if into is None:
        g_heads = torch.empty((256, HEADS_COLS), dtype=torch.float32, device=dev)
        g_evec = torch.empty((6, 256), dtype=torch.float32, device=dev)
        # the 48 static parameters' gradients go straight into the optimiser's flat gradient buffer when one is registered (grad_arena)
        # (not when one tensor fills several slots -- VariableNet.forward standalone -- whose gradients must stay separate tensors)
        arena = len({s.data_ptr() for s in statics}) == len(statics)
        slots = [slot_of(statics[k * 8 + j]) if arena else None for k in range(6) for j in range(8)]
        g_stat = [s_.view(STATIC_SHAPES[i % 8]) if s_ is not None else torch.empty(STATIC_SHAPES[i % 8], dtype=torch.float32, device=dev)
                  for i, s_ in enumerate(slots)]
        arena = arena and all(s_ is not None for s_ in slots)
    else:
        g_heads, g_evec, g_stat = into
    garr = _net_ptrs(g_heads, g_evec, g_stat, cls=L.DpnNetGradPtrs)
    if clock is not None:
        clock.stamp('wgrad')
    L.check(lib.dpn_wgrad(n, cfg.prec, _ptr(g_out), _ptr(ws.saved), _ptr(operands), _ptr(partials), _stream()), 'dpn_wgrad')
    if clock is not None:
        clock.stamp('wgrad')
    # (only when every static gradient is a freshly leased slot of the optimiser's flat buffer: autograd then keeps the view as param.grad without
    # touching it; a gradient it would have to ADD to an existing one on the main stream must be complete when the node returns)
    if fork and arena and branch.enabled('finish'):
        L.check(lib.dpn_wgrad_finish_parts(nets, _ptr(ws.packed), n, cfg.prec, _ptr(partials), garr, 1, _stream()), 'dpn_wgrad_finish')
        # keep: what the branch READS (not the 48 gradient slots: they are persistent, and a second reference to a returned gradient makes
        # autograd copy it instead of keeping the view as param.grad)
        with branch.side(keep=(partials, ws.packed, statics) + tuple(keep)):
            L.check(lib.dpn_wgrad_finish_parts(nets, _ptr(ws.packed), n, cfg.prec, _ptr(partials), garr, 2, _stream()), 'dpn_wgrad_finish')
    else:
        L.check(lib.dpn_wgrad_finish(nets, _ptr(ws.packed), n, cfg.prec, _ptr(partials), garr, _stream()), 'dpn_wgrad_finish')
    return g_heads, g_evec, g_stat


def _stamp(tensors):
    """Version counters of the tensors a backward pass will read again through raw pointers (the point path keeps them in ctx.keep, not in
    save_for_backward: most are detached fp32 views).  _check_stamp raises when one was modified in place between forward and backward
    (e.g. an optimiser step before a delayed backward), which autograd's own check would catch for saved tensors."""
    from . import grad_arena
    owners = {}
    for t in tensors:                       # fused optimisers that own one of these tensors (they rewrite it through a raw pointer)
        e = grad_arena._slots.get(t.data_ptr()) if t is not None else None
        o = e[0]() if e is not None else None
        if o is not None:
            owners[id(o)] = (e[0], o.steps_done)
    return (('owners', tuple(owners.values())),) + tuple((t, t._version) for t in tensors if t is not None)


def _check_stamp(stamp, what):
    if stamp and stamp[0][0] == 'owners':
        # the fused optimiser rewrites parameters through raw pointers (no version bump): its own step count says whether it ran between
        # this forward and its backward (steps replayed from a hipGraph are invisible to the host: not covered)
        for ref, steps in stamp[0][1]:
            o = ref()
            if o is not None and o.steps_done != steps:
                raise RuntimeError('deepphysinet_amd %s: a fused optimiser step ran between the forward pass and its backward pass' % what)
        stamp = stamp[1:]
    for t, v in stamp:
        if t._version != v:
            raise RuntimeError('deepphysinet_amd %s: a tensor of shape %s needed by the backward pass was modified in place after the '
                               'forward pass (version %d -> %d)' % (what, tuple(t.shape), v, t._version))


class _NoSecondOrder(torch.autograd.Function):
    """Identity whose backward raises: marks a first derivative that cannot be differentiated again."""

    @staticmethod
    def forward(ctx, v):
        return v.view_as(v)

    @staticmethod
    def backward(ctx, g):
        raise RuntimeError('deepphysinet_amd: d(fields)/d(coordinates) of PhysicsNet.forward is first-order only; training through the '
                           'standalone *_equation methods needs its parameter derivative -- use InterfacePhysics.place_one_batch (fused '
                           'residual kernels) for that')


class _PointFieldsFn(torch.autograd.Function):
    """out_n [N,6] = six VariableNets at N points.  First-order differentiable w.r.t. the weights and -- when the coordinates come in
    already encoded (pe_in, the reference's PhysicsNet.forward surface) -- w.r.t. pe_in, so that the reference's `gradient(u, x)`
    (interface_physics.py:90-95) evaluates on the outputs of the HIP model.  Differentiating such a derivative again (training through the
    standalone equation methods) raises (_NoSecondOrder); place_one_batch is the fused path for that."""

    @staticmethod
    def forward(ctx, cfg, x, y, t, pe_in, coord_data, heads, evec, *statics):
        for nm, v in (('coord_data', coord_data), ('heads', heads)):
            _require_gpu(v, nm)
        tens = [None if v is None else _f32c(v) for v in (x, y, t, pe_in, coord_data, heads, evec)]
        x_, y_, t_, pe_, cd_, hd_, ev_ = tens
        st = [_f32c(s) for s in statics]
        need_grad = any(v.requires_grad for v in (heads, evec) + tuple(statics)) and int(getattr(cfg, 'n_nets', 6) or 6) == 6
        want_gpe = pe_in is not None and pe_in.requires_grad        # (n_nets < 6: the caller has established that nothing is differentiated)
        ws = _Workspace(cd_.shape[0], cfg.prec, cd_.device)
        nets = _net_ptrs(hd_, ev_, st)
        ref6 = getattr(cfg, 'ref6', None)                 # VariableNet.forward's own ref_data (a constant: no gradient flows to it here)
        out_n, gpe = _forward_points(cfg, ws, nets, x_, y_, t_, pe_, cd_, want_jac=want_gpe, want_saved=need_grad,
                                     ref6=None if ref6 is None else _f32c(ref6.detach()))
        ctx.cfg, ctx.ws, ctx.gpe = cfg, ws, gpe
        ctx.keep = (x_, y_, t_, pe_, cd_, hd_, ev_, st)
        ctx.stamp = _stamp((heads, evec) + tuple(statics))
        return out_n

    @staticmethod
    def backward(ctx, g_out):
        _check_stamp(ctx.stamp, 'point_fields')
        x_, y_, t_, pe_, cd_, hd_, ev_, st = ctx.keep
        g = _f32c(g_out)
        g_pe = None
        if ctx.needs_input_grad[4] and ctx.gpe is not None:
            g_pe = torch.empty((g.shape[0], 192), dtype=torch.float32, device=g.device)
            L.check(L.load().dpn_contract_gpe(_ptr(g), _ptr(ctx.gpe), g.shape[0], _ptr(g_pe), _stream()), 'dpn_contract_gpe')
            if torch.is_grad_enabled():                # create_graph=True (the reference's gradient()): a later backward through this
                g_pe = _NoSecondOrder.apply(g_pe.requires_grad_(True))      # derivative must fail loudly, not drop its parameter part
        if not any(ctx.needs_input_grad[6:]):
            return (None, None, None, None, g_pe) + (None,) * (3 + len(st))
        nets = _net_ptrs(hd_, ev_, st)
        ghd, gev, gst = _backward_points(ctx.cfg, ctx.ws, nets, x_, y_, t_, pe_, cd_, g, None, st, fork=True, keep=(hd_, ev_))
        return (None, None, None, None, g_pe, None, ghd, gev, *gst)


class _PdeLossFn(torch.autograd.Function):
    """losses [6] = (motion_u, motion_v, continuous, energy, vapor, gas), each already scaled by its factor."""

    @staticmethod
    def forward(ctx, cfg, x, y, t, f, coord_data, heads, evec, *statics):
        for nm, v in (('x', x), ('coord_data', coord_data), ('heads', heads)):
            _require_gpu(v, nm)
        lib = L.load()
        x_, y_, t_, f_, cd_, hd_, ev_ = [_f32c(v).reshape(-1) if i < 4 else _f32c(v) for i, v in
                                         enumerate((x, y, t, f, coord_data, heads, evec))]
        st = [_f32c(s) for s in statics]
        n = cd_.shape[0]
        dev = cd_.device
        need_grad = any(v.requires_grad for v in (heads, evec) + tuple(statics))
        ws = _Workspace(n, cfg.prec, dev)
        nets = _net_ptrs(hd_, ev_, st)
        out_n, jac_n = _forward_points(cfg, ws, nets, x_, y_, t_, None, cd_, want_jac=True, want_saved=need_grad)
        sums = torch.empty(((n + 255) // 256) * 6, dtype=torch.float64, device=dev)      # per-block rows, reduced by dpn_residual_finish
        losses7 = torch.empty(7, dtype=torch.float32, device=dev)
        geo, ph = cfg.geometry(), cfg.physics()
        # With gradients wanted, the SAME pass over the points also writes d total / d (out, Jacobian) for a unit cotangent of the total (the usual
        # backward: loss.backward(seed) on the sum of the six terms): the backward pass then has no residual launch of its own, stage 1 multiplies
        # the cotangent that arrives into the streams as it reads them (dpn_bwd_points_scaled).  A cotangent on the individual terms takes the
        # separate pass, as before.
        ctx.unit = None
        if need_grad:
            g_out = torch.empty((n, 6), dtype=torch.float32, device=dev)
            g_jxi = torch.empty((n, 6, 3), dtype=torch.float32, device=dev)
            ctx.unit = (g_out, g_jxi)
        L.check(lib.dpn_residual(_ptr(out_n), _ptr(jac_n), _ptr(f_), n, ctypes.byref(geo), ctypes.byref(ph), None, _ptr(_one(dev)) if need_grad else None,
                                 _ptr(sums), _ptr(g_out) if need_grad else None, _ptr(g_jxi) if need_grad else None, _stream()), 'dpn_residual')
        L.check(lib.dpn_residual_finish(_ptr(sums), n, ctypes.byref(ph), _ptr(losses7), _stream()), 'dpn_residual_finish')
        ctx.cfg, ctx.ws = cfg, ws
        ctx.keep = (x_, y_, t_, f_, cd_, hd_, ev_, st, out_n, jac_n)
        ctx.stamp = _stamp((heads, evec) + tuple(statics))
        ctx.set_materialize_grads(False)
        return losses7[:6], losses7[6]

    @staticmethod
    def backward(ctx, g_losses, g_total):
        lib = L.load()
        cfg = ctx.cfg
        _check_stamp(ctx.stamp, 'pde_losses')
        x_, y_, t_, f_, cd_, hd_, ev_, st, out_n, jac_n = ctx.keep
        n = cd_.shape[0]
        dev = cd_.device
        gl = None if g_losses is None else _f32c(g_losses)
        gt = None if g_total is None else _f32c(g_total).reshape(1)
        if gl is None and gt is None:
            return (None,) * (8 + len(st))
        nets = _net_ptrs(hd_, ev_, st)
        if gl is None and ctx.unit is not None:                  # cotangent on the total only: the unit-cotangent streams of the forward pass, scaled on load
            g_out, g_jxi = ctx.unit
            ghd, gev, gst = _backward_points(cfg, ctx.ws, nets, x_, y_, t_, None, cd_, g_out, g_jxi, st, fork=True, keep=(hd_, ev_), g_scale=gt)
            return (None, None, None, None, None, None, ghd, gev, *gst)
        g_out = torch.empty((n, 6), dtype=torch.float32, device=dev)
        g_jxi = torch.empty((n, 6, 3), dtype=torch.float32, device=dev)
        geo, ph = cfg.geometry(), cfg.physics()
        L.check(lib.dpn_residual(_ptr(out_n), _ptr(jac_n), _ptr(f_), n, ctypes.byref(geo), ctypes.byref(ph), _ptr(gl), _ptr(gt), None,
                                 _ptr(g_out), _ptr(g_jxi), _stream()), 'dpn_residual(grad)')
        ghd, gev, gst = _backward_points(cfg, ctx.ws, nets, x_, y_, t_, None, cd_, g_out, g_jxi, st, fork=True, keep=(hd_, ev_))
        return (None, None, None, None, None, None, ghd, gev, *gst)


def _pack_batch(cfg, heads, evec, statics, n):
    """The packed weight blocks of B fields (heads [B, 256, 2700], evec [B, 6, 256], shared statics) in ONE launch -> uint8 [B, stride]; row b is
    field b's `_Workspace.packed`.  (Per field the launch is 17 us of latency-bound tiles, 61 of them 1 ms of a 51-ms step: DESIGN.md 6a.)"""
    lib = L.load()
    sizes = L.DpnSizes()
    L.check(lib.dpn_sizes(n, cfg.prec, ctypes.byref(sizes)), 'dpn_sizes')
    B = heads.shape[0]
    stride = (int(sizes.packed) + 255) // 256 * 256
    packed = torch.empty((B, stride), dtype=torch.uint8, device=heads.device)
    nets = _net_ptrs(heads[0], evec[0], statics)
    form = lib.dpn_fwd_form(cfg.prec, 0)
    L.check(lib.dpn_pack_weights_batch(nets, B, heads.stride(0), evec.stride(0), cfg.prec, form, _ptr(packed), stride, _stream()), 'dpn_pack_weights_batch')
    return packed


class _PdeLossBatchFn(torch.autograd.Function):
    """BASELINE configs[2]: B field samples (distinct field / lead time => distinct hyper-network weights) with N collocation points each
    in ONE step.  losses [B, 6] and totals [B]; heads [B, 256, 2700], evec [B, 6, 256], point tensors [B, N(,6)].  The point kernels run
    field after field (each launch already fills the chip); the static-parameter gradients of the fields are written side by side and
    added in a fixed order by one dpn_sum_parts launch, so nothing is accumulated through autograd.

    When gradients are wanted, each field's point BACKWARD runs right behind its forward, for a unit cotangent of that field's total
    (the gradients are linear in it; the backward pass scales them by the cotangent that arrives): the field's saved state (0.35 GB) is
    consumed while it is still warm in the memory-side cache and freed at once, instead of 61 of them (21 GB) waiting for the backward
    pass -- measured on the stage-1 backward kernel: 252 us per field at 2 fields, 273 at 8, 293 at 24, 295 at 61 with the state parked, 249
    at 61 this way; the 61-field step 80.5 -> 78.6 ms on the same box.  A
    cotangent on the individual loss terms (not only on the totals) takes the general path: forward again, field by field."""


    @staticmethod
    def _static_layout():
        numels = [int(torch.Size(STATIC_SHAPES[j]).numel()) for _ in range(6) for j in range(8)]
        starts = [0]
        for m_ in numels:
            starts.append(starts[-1] + m_)
        return starts

    @staticmethod
    def forward(ctx, cfg, grad_enabled, x, y, t, f, coord_data, heads, evec, *statics):
        for nm, v in (('x', x), ('coord_data', coord_data), ('heads', heads)):
            _require_gpu(v, nm)
        lib = L.load()
        B, n = coord_data.shape[0], coord_data.shape[1]
        x_, y_, t_, f_ = (_f32c(v).reshape(B, n) for v in (x, y, t, f))
        cd_, hd_, ev_ = _f32c(coord_data), _f32c(heads), _f32c(evec)
        st = [_f32c(s) for s in statics]
        dev = cd_.device
        # Function.forward always runs with grad mode off: the caller's mode arrives as an argument (pde_losses_batch), so that a no-grad
        # evaluation (validation, place_lead_batch scoring) neither saves state nor runs each field's point backward
        need_grad = bool(grad_enabled) and any(v.requires_grad for v in (heads, evec) + tuple(statics))
        eager = need_grad and config.FROZEN.batch_eager_backward
        losses7 = torch.empty((B, 7), dtype=torch.float32, device=dev)
        sums = torch.empty((B, ((n + 255) // 256) * 6), dtype=torch.float64, device=dev)      # per field: block rows, all reduced by ONE launch behind the loop
        geo, ph = cfg.geometry(), cfg.physics()
        fields = []
        one_launch = config.FROZEN.batch_pack                     # every field's weight block: one launch in front of the loop
        packed = _pack_batch(cfg, hd_, ev_, st, n) if one_launch else None
        if eager:
            starts = _PdeLossBatchFn._static_layout()
            one = torch.ones(1, dtype=torch.float32, device=dev)
            g_out = torch.empty((n, 6), dtype=torch.float32, device=dev)
            g_jxi = torch.empty((n, 6, 3), dtype=torch.float32, device=dev)
            g_heads = torch.empty((B, 256, HEADS_COLS), dtype=torch.float32, device=dev)
            g_evec = torch.empty((B, 6, 256), dtype=torch.float32, device=dev)
            flat = torch.empty((B, starts[-1]), dtype=torch.float32, device=dev)
        for b in range(B):
            ws = _Workspace(n, cfg.prec, dev, packed=packed[b] if one_launch else None)
            nets = _net_ptrs(hd_[b], ev_[b], st)
            out_n, jac_n = _forward_points(cfg, ws, nets, x_[b], y_[b], t_[b], None, cd_[b], want_jac=True, want_saved=need_grad)
            # eager: the block sums of the losses AND d total_b / d (out, Jacobian) for a unit cotangent in ONE pass over the points
            L.check(lib.dpn_residual(_ptr(out_n), _ptr(jac_n), _ptr(f_[b]), n, ctypes.byref(geo), ctypes.byref(ph), None,
                                     _ptr(one) if eager else None, _ptr(sums[b]), _ptr(g_out) if eager else None, _ptr(g_jxi) if eager else None,
                                     _stream()), 'dpn_residual')
            if eager:                                             # d total_b / d (this field's weights)
                g_stat = [flat[b, starts[i]:starts[i + 1]].view(STATIC_SHAPES[i % 8]) for i in range(48)]
                _backward_points(cfg, ws, nets, x_[b], y_[b], t_[b], None, cd_[b], g_out, g_jxi, st, into=(g_heads[b], g_evec[b], g_stat))
                del ws, out_n, jac_n
            elif need_grad:
                fields.append((ws, out_n, jac_n))
            if not one_launch:
                L.check(lib.dpn_residual_finish(_ptr(sums[b]), n, ctypes.byref(ph), _ptr(losses7[b]), _stream()), 'dpn_residual_finish')
        if one_launch:
            L.check(lib.dpn_residual_finish_batch(_ptr(sums), n, B, ctypes.byref(ph), _ptr(losses7), _stream()), 'dpn_residual_finish')
        ctx.cfg, ctx.fields = cfg, fields
        ctx.eager = (g_heads, g_evec, flat) if eager else None
        ctx.keep = (x_, y_, t_, f_, cd_, hd_, ev_, st)
        ctx.stamp = _stamp((heads, evec) + tuple(statics))
        ctx.set_materialize_grads(False)
        return losses7[:, :6], losses7[:, 6]

    @staticmethod
    def backward(ctx, g_losses, g_total):
        lib = L.load()
        cfg = ctx.cfg
        _check_stamp(ctx.stamp, 'pde_losses_batch')
        x_, y_, t_, f_, cd_, hd_, ev_, st = ctx.keep
        B, n = cd_.shape[0], cd_.shape[1]
        dev = cd_.device
        if g_losses is None and g_total is None:
            return (None,) * (9 + len(st))
        starts = _PdeLossBatchFn._static_layout()
        if ctx.eager is not None and g_losses is None:
            # the gradients are there for unit cotangents of the B totals: scale them by the cotangents that arrived
            g_heads, g_evec, flat = ctx.eager
            ctx.eager = None
            gt = _f32c(g_total).reshape(B)
            g_heads.mul_(gt.view(B, 1, 1))
            g_evec.mul_(gt.view(B, 1, 1))
            flat.mul_(gt.view(B, 1))
        else:
            gl = None if g_losses is None else _f32c(g_losses)
            gt = None if g_total is None else _f32c(g_total)
            g_out = torch.empty((n, 6), dtype=torch.float32, device=dev)
            g_jxi = torch.empty((n, 6, 3), dtype=torch.float32, device=dev)
            g_heads = torch.empty((B, 256, HEADS_COLS), dtype=torch.float32, device=dev)
            g_evec = torch.empty((B, 6, 256), dtype=torch.float32, device=dev)
            flat = torch.empty((B, starts[-1]), dtype=torch.float32, device=dev)
            geo, ph = cfg.geometry(), cfg.physics()
            recompute = ctx.eager is not None                      # cotangents on the individual terms: the general path, forward again
            ctx.eager = None
            if not recompute and (len(ctx.fields) != B or any(fld is None for fld in ctx.fields)):
                raise RuntimeError('deepphysinet_amd pde_losses_batch: the per-field state saved by the forward pass is released as the backward '
                                   'pass consumes it (21 GB at 61 fields); run the forward pass again instead of a second backward')
            for b in range(B):
                nets = _net_ptrs(hd_[b], ev_[b], st)
                if recompute:
                    ws = _Workspace(n, cfg.prec, dev)
                    out_n, jac_n = _forward_points(cfg, ws, nets, x_[b], y_[b], t_[b], None, cd_[b], want_jac=True, want_saved=True)
                else:
                    ws, out_n, jac_n = ctx.fields[b]
                L.check(lib.dpn_residual(_ptr(out_n), _ptr(jac_n), _ptr(f_[b]), n, ctypes.byref(geo), ctypes.byref(ph),
                                         None if gl is None else _ptr(gl[b]), None if gt is None else _ptr(gt[b:b + 1]), None,
                                         _ptr(g_out), _ptr(g_jxi), _stream()), 'dpn_residual(grad)')
                g_stat = [flat[b, starts[i]:starts[i + 1]].view(STATIC_SHAPES[i % 8]) for i in range(48)]
                _backward_points(cfg, ws, nets, x_[b], y_[b], t_[b], None, cd_[b], g_out, g_jxi, st, into=(g_heads[b], g_evec[b], g_stat))
                if not recompute:
                    ctx.fields[b] = None                          # this field's saved state is no longer needed
        total = torch.empty(starts[-1], dtype=torch.float32, device=dev)
        L.check(lib.dpn_sum_parts(_ptr(flat), B, starts[-1], 0, _ptr(total), _stream()), 'dpn_sum_parts')
        gst = [total[starts[i]:starts[i + 1]].view(STATIC_SHAPES[i % 8]) for i in range(48)]
        return (None, None, None, None, None, None, None, g_heads, g_evec, *gst)


class _StepLossFn(torch.autograd.Function):
    """The loss of the reference's step body (interface_physics.py:464-501) in ONE point pass: the first n_inter points are the interior
    collocation points, the rest the margin (grid-node, labelled) points.  Returns (inter_terms [6], inter_total, margin_terms [6],
    margin_total, data_loss): the PDE means are taken per group, the SmoothL1 data loss over the margin points; the margin points'
    forward serves both of their losses, and all 24 576 points share one backward / weight-gradient / finish sequence."""

    @staticmethod
    def forward(ctx, cfg, n_inter, beta, margin_factor, x, y, t, f, coord_data, labels, heads, evec, *statics):
        for nm, v in (('x', x), ('coord_data', coord_data), ('heads', heads), ('labels', labels)):
            _require_gpu(v, nm)
        lib = L.load()
        x_, y_, t_, f_ = (_f32c(v).reshape(-1) for v in (x, y, t, f))
        cd_, hd_, ev_, lab_ = _f32c(coord_data), _f32c(heads), _f32c(evec), _f32c(labels)
        st = [_f32c(s) for s in statics]
        n = cd_.shape[0]
        n_m = n - n_inter
        assert 0 < n_inter < n and lab_.shape[0] == n_m
        dev = cd_.device
        need_grad = any(v.requires_grad for v in (heads, evec) + tuple(statics))
        ws = _Workspace(n, cfg.prec, dev)
        nets = _net_ptrs(hd_, ev_, st)
        out_n, jac_n = _forward_points(cfg, ws, nets, x_, y_, t_, None, cd_, want_jac=True, want_saved=need_grad)
        geo, ph = cfg.geometry(), cfg.physics()
        losses = torch.empty((2, 7), dtype=torch.float32, device=dev)
        for gi, (a0, a1) in enumerate(((0, n_inter), (n_inter, n))):
            sums = torch.empty(((a1 - a0 + 255) // 256) * 6, dtype=torch.float64, device=dev)
            L.check(lib.dpn_residual(_ptr(out_n[a0:]), _ptr(jac_n[a0:]), _ptr(f_[a0:]), a1 - a0, ctypes.byref(geo), ctypes.byref(ph), None, None,
                                     _ptr(sums), None, None, _stream()), 'dpn_residual')
            L.check(lib.dpn_residual_finish(_ptr(sums), a1 - a0, ctypes.byref(ph), _ptr(losses[gi]), _stream()), 'dpn_residual_finish')
        dsum = torch.empty((n_m * 6 + 255) // 256, dtype=torch.float64, device=dev)
        L.check(lib.dpn_smooth_l1(_ptr(out_n[n_inter:]), _ptr(lab_), n_m, beta, 1.0, _ptr(dsum), None, 0, None, _stream()), 'dpn_smooth_l1')
        data = (dsum.sum() / (6.0 * n_m)).float() * margin_factor
        ctx.cfg, ctx.ws, ctx.n_inter, ctx.beta, ctx.margin_factor = cfg, ws, n_inter, beta, margin_factor
        ctx.keep = (x_, y_, t_, f_, cd_, lab_, hd_, ev_, st, out_n, jac_n)
        ctx.stamp = _stamp((heads, evec) + tuple(statics))
        ctx.set_materialize_grads(False)
        return losses[0, :6], losses[0, 6], losses[1, :6], losses[1, 6], data

    @staticmethod
    def backward(ctx, g_la, g_ta, g_lb, g_tb, g_data):
        lib = L.load()
        cfg, n_inter = ctx.cfg, ctx.n_inter
        _check_stamp(ctx.stamp, 'step_losses')
        x_, y_, t_, f_, cd_, lab_, hd_, ev_, st, out_n, jac_n = ctx.keep
        n = cd_.shape[0]
        n_m = n - n_inter
        dev = cd_.device
        if all(v is None for v in (g_la, g_ta, g_lb, g_tb, g_data)):
            return (None,) * (12 + len(st))
        g_out = torch.empty((n, 6), dtype=torch.float32, device=dev)
        g_jxi = torch.empty((n, 6, 3), dtype=torch.float32, device=dev)
        geo, ph = cfg.geometry(), cfg.physics()
        zero6 = None
        for (a0, a1), gl, gt in (((0, n_inter), g_la, g_ta), ((n_inter, n), g_lb, g_tb)):
            if gl is None and gt is None:                     # this group's PDE losses are unused: zero cotangent
                zero6 = torch.zeros(6, dtype=torch.float32, device=dev) if zero6 is None else zero6
                gl = zero6
            L.check(lib.dpn_residual(_ptr(out_n[a0:]), _ptr(jac_n[a0:]), _ptr(f_[a0:]), a1 - a0, ctypes.byref(geo), ctypes.byref(ph),
                                     None if gl is None else _ptr(_f32c(gl)), None if gt is None else _ptr(_f32c(gt).reshape(1)), None,
                                     _ptr(g_out[a0:]), _ptr(g_jxi[a0:]), _stream()), 'dpn_residual(grad)')
        if g_data is not None:                                # + d(data loss)/d out on the margin rows
            L.check(lib.dpn_smooth_l1(_ptr(out_n[n_inter:]), _ptr(lab_), n_m, ctx.beta, ctx.margin_factor / (6.0 * n_m), None,
                                      _ptr(g_out[n_inter:]), 1, _ptr(_f32c(g_data).reshape(1)), _stream()), 'dpn_smooth_l1(grad)')
        nets = _net_ptrs(hd_, ev_, st)
        ghd, gev, gst = _backward_points(cfg, ctx.ws, nets, x_, y_, t_, None, cd_, g_out, g_jxi, st, fork=True, keep=(hd_, ev_))
        return (None,) * 10 + (ghd, gev, *gst)


def step_losses(cfg: PointConfig, n_inter, x, y, t, f, coord_data, labels, heads, evec, statics, beta=0.1, margin_factor=1.0):
    """(inter_terms [6], inter_total, margin_terms [6], margin_total, data_loss) of the reference's step body in one point pass; the first
    n_inter rows of x, y, t, f, coord_data are the interior points, the rest the margin points whose labels are `labels` (_StepLossFn)."""
    return _StepLossFn.apply(cfg, int(n_inter), float(beta), float(margin_factor), x, y, t, f, coord_data, labels, heads, evec, *statics)


def pde_losses_batch(cfg: PointConfig, x, y, t, f, coord_data, heads, evec, statics):
    """(losses [B,6], totals [B]) for B field samples with N points each; tensors carry a leading B (see _PdeLossBatchFn)."""
    return _PdeLossBatchFn.apply(cfg, torch.is_grad_enabled(), x, y, t, f, coord_data, heads, evec, *statics)


def point_fields(cfg: PointConfig, coord_data, heads, evec, statics, x=None, y=None, t=None, pe_in=None):
    """Normalised fields [N,6].  Give either raw (x,y,t) [N] or caller-encoded coordinates pe_in [N,192].
    heads [256, 2700]: one row per hidden channel, columns [w1b1 of nets 0..5 (193 each) | w2b2 of nets 0..5 (257 each)]."""
    if (pe_in is None) == (x is None):
        raise ValueError('give exactly one of (x,y,t) or pe_in')
    if x is not None:
        x, y, t = (v.reshape(-1) for v in (x, y, t))
    return _PointFieldsFn.apply(cfg, x, y, t, pe_in, coord_data, heads, evec, *statics)


def pde_losses(cfg: PointConfig, x, y, t, f, coord_data, heads, evec, statics, with_total=False):
    """The six scaled residual losses [6] of place_one_batch (and, with_total, their sum in the reference's order as a 0-dim tensor)."""
    terms, total = _PdeLossFn.apply(cfg, x, y, t, f, coord_data, heads, evec, *statics)
    return (terms, total) if with_total else terms


def pde_fields_and_jacobian(cfg: PointConfig, x, y, t, coord_data, heads, evec, statics):
    """No-grad helper: normalised fields [N,6] and d(fields_n)/d(x,y,t) [N,6,3] straight from the forward kernel."""
    with torch.no_grad():
        x_, y_, t_ = (_f32c(v).reshape(-1) for v in (x, y, t))
        cd_, hd_, ev_ = (_f32c(v) for v in (coord_data, heads, evec))
        st = [_f32c(s) for s in statics]
        ws = _Workspace(cd_.shape[0], cfg.prec, cd_.device)
        return _forward_points(cfg, ws, _net_ptrs(hd_, ev_, st), x_, y_, t_, None, cd_, want_jac=True, want_saved=False)


def relu_masks(cfg: PointConfig, x, y, t, coord_data, heads, evec, statics):
    """Diagnostic export of the two ReLU masks of every VariableNet at every point, decoded from the state dpn_fwd saves for the backward
    pass: (m1, m2), bool [6, N, 256] in natural channel order -- m1 = (w1 . pe + b1 > 0) (variable_net.py:67-68), m2 = (cat_fc1.fc.0 pre-
    activation > 0) (ResMLP, :13-24).  The Jacobian and every gradient are piecewise constant / linear in these bits, so a point whose
    pre-activation lies within rounding distance of zero may carry a different bit than another arithmetic's (the parity tests list those
    points and hold everything else to the tight bounds).  Layouts: csrc/dpn_kernels.hip SavedView, dpn_layout.h."""
    with torch.no_grad():
        x_, y_, t_ = (_f32c(v).reshape(-1) for v in (x, y, t))
        cd_, hd_, ev_ = (_f32c(v) for v in (coord_data, heads, evec))
        st = [_f32c(s) for s in statics]
        n = cd_.shape[0]
        dev = cd_.device
        ws = _Workspace(n, cfg.prec, dev)
        _forward_points(cfg, ws, _net_ptrs(hd_, ev_, st), x_, y_, t_, None, cd_, want_jac=True, want_saved=True)
        n_pad, ns = int(ws.sizes.n_pad), int(cfg.prec)
        tiles = n_pad // 32
        mat = 6 * ns * n_pad * 512                                                      # T1 (hi, lo planes) | M2 | m1: SavedView
        raw = ws.saved
        # M2: [6][tiles][kk 2][ct 8][lane 64][8 bf16 of 0/1]: register r = 8 kk + e of lane (col jj = lane & 31, h = lane >> 5) is the mask of
        # channel chain_ch(2 ct + (jj >> 4), (jj >> 3) & 1, jj & 7) at point drow32(r, h) of the tile
        m2_raw = raw[mat:mat + 6 * n_pad * 512].view(torch.int16).view(6, tiles, 2, 8, 64, 8) != 0
        lane = torch.arange(64, device=dev)
        jj, hh = lane & 31, lane >> 5
        ct = torch.arange(8, device=dev)
        ks = 2 * ct[:, None] + (jj >> 4)[None, :]                                       # [ct, lane]
        chan = 16 * ks + 8 * ((jj & 7) >> 2)[None, :] + 4 * ((jj >> 3) & 1)[None, :] + (jj & 3)[None, :]
        r = torch.arange(16, device=dev)
        prow = (r & 3)[:, None] + 8 * (r >> 2)[:, None] + 4 * hh[None, :]              # [r, lane] point row inside the tile
        m2 = torch.zeros((6, tiles, 32, 256), dtype=torch.bool, device=dev)
        src = m2_raw.permute(0, 1, 3, 4, 2, 5).reshape(6, tiles, 8, 64, 16)            # [net, tile, ct, lane, r]
        pi = prow.t()[None, :, :].expand(8, 64, 16)                                    # [ct, lane, r]
        ci = chan[:, :, None].expand(8, 64, 16)
        m2[:, :, pi, ci] = src
        # m1: uint4 per (net, tile, lane): bit 16 (T & 1) + r of word T >> 1 = mask of channel 32 T + drow32(r, h) at point lane & 31
        m1_raw = raw[mat + 6 * n_pad * 512:mat + 6 * n_pad * 512 + 6 * n_pad * 32].view(torch.int32).view(6, tiles, 64, 4)
        T = torch.arange(8, device=dev)
        words = m1_raw[:, :, :, T >> 1]                                                # [net, tile, lane, T]
        bits = (words[..., None] >> (16 * (T & 1)[:, None] + r[None, :])) & 1           # [net, tile, lane, T, r]
        ch1 = 32 * T[None, :, None] + (r & 3)[None, None, :] + 8 * (r >> 2)[None, None, :] + 4 * hh[:, None, None]       # [lane, T, r]
        m1 = torch.zeros((6, tiles, 32, 256), dtype=torch.bool, device=dev)
        p1 = jj[:, None, None].expand(64, 8, 16)
        m1[:, :, p1, ch1] = bits.bool()
        return m1.reshape(6, n_pad, 256)[:, :n], m2.reshape(6, n_pad, 256)[:, :n]


def smooth_l1_data_loss(out_n, labels, beta=0.1, factor=1.0):
    """mean(SmoothL1(beta)) * factor as a differentiable torch scalar (losses/weights_loss.py:17-20); HIP kernel for both passes."""
    return _SmoothL1Fn.apply(out_n, labels, float(beta), float(factor))


class _SmoothL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out_n, labels, beta, factor):
        _require_gpu(out_n, 'out_n')
        lib = L.load()
        o, l = _f32c(out_n), _f32c(labels)
        n = o.shape[0]
        s = torch.empty((n * 6 + 255) // 256, dtype=torch.float64, device=o.device)      # per-block partial sums
        g = torch.empty_like(o)
        L.check(lib.dpn_smooth_l1(_ptr(o), _ptr(l), n, beta, factor / (6.0 * n), _ptr(s), _ptr(g), 0, None, _stream()), 'dpn_smooth_l1')
        ctx.save_for_backward(g)
        return ((s.sum() / (6.0 * n)).float() * factor)

    @staticmethod
    def backward(ctx, gl):
        (g,) = ctx.saved_tensors
        return g * gl, None, None, None
