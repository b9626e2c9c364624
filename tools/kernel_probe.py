"""Times the four point kernels in isolation (HIP events, eager) for the bench workload.
DPN_PROBE_SOAK=<kernel>,<seconds>: first keep that kernel running back to back for so long (tools/clock_watch.py samples the clocks and the
socket power it settles at) -- only that kernel is then timed."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L, point_path as PP
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 257 * 145
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision=prec).to(dev)
b = synth_batch(n, dev, seed=1)
cfg = m.point_config()
lib = L.load()
with torch.no_grad():
    heads, evec, statics = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
    x_, y_, t_, f_ = (PP._f32c(b[k]).reshape(-1) for k in ('x', 'y', 't', 'f'))
    cd_ = PP._f32c(b['coord_data'])
    st = [PP._f32c(s) for s in statics]
    ws = PP._Workspace(n, cfg.prec, dev)
    nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
    out_n, jac_n = PP._forward_points(cfg, ws, nets, x_, y_, t_, None, cd_, True, True)
    geo, ph = cfg.geometry(), cfg.physics()
    g_out = torch.empty((n, 6), device=dev); g_jxi = torch.empty((n, 6, 3), device=dev)
    s = PP._stream()
    L.check(lib.dpn_residual(PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(f_), n, ctypes.byref(geo), ctypes.byref(ph), None, None, None, PP._ptr(g_out), PP._ptr(g_jxi), s), 'res')
    operands = torch.empty(ws.sizes.operands, dtype=torch.uint8, device=dev)
    partials = torch.empty(ws.sizes.partials, dtype=torch.uint8, device=dev)
    fr = PP._freqs(dev)
    def k_fwd(): L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec, PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(ws.saved), s), 'fwd')
    def k_bwd(): L.check(lib.dpn_bwd_points(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec, PP._ptr(g_out), PP._ptr(g_jxi), PP._ptr(ws.saved), PP._ptr(operands), s), 'bwd')
    def k_wgrad(): L.check(lib.dpn_wgrad(n, cfg.prec, PP._ptr(g_out), PP._ptr(ws.saved), PP._ptr(operands), PP._ptr(partials), s), 'wgrad')
    def k_pack(): L.check(lib.dpn_pack_weights(nets, cfg.prec, PP._ptr(ws.packed), s), 'pack')
    def k_fwd_nosave(): L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec, PP._ptr(out_n), PP._ptr(jac_n), None, s), 'fwd')
    def k_fwd_fields(): L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec, PP._ptr(out_n), None, None, s), 'fwd')
    extra = (('fwd, nothing saved', k_fwd_nosave), ('fwd, fields only', k_fwd_fields)) if os.environ.get('DPN_PROBE_VARIANTS') else ()
    soak = os.environ.get('DPN_PROBE_SOAK', '').split(',')
    for name, fn in (('pack', k_pack), ('fwd', k_fwd), ('bwd', k_bwd), ('wgrad', k_wgrad)) + extra:
        if soak[0]:
            if name != soak[0]:
                continue
            import time
            t0 = time.time()
            while time.time() - t0 < float(soak[1]):
                for _ in range(50): fn()
                torch.cuda.synchronize()
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print('%s %-6s %8.1f us  (lib=%s, k_splits=%d)' % (prec, name, e0.elapsed_time(e1) * 100, os.path.basename(L.LIB_PATH), ws.sizes.k_splits))
