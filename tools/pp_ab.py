"""A/B of the tile-split forward + Jacobian kernel and its ping-pong form (dpn_fwd_pp.h, DPN_FWD_PP=0|1): bitwise comparison of everything
they write and interleaved HIP-event timing (back-to-back launches).  usage: pp_ab.py [n ...]       (DPN_LIB selects an experiment library)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L, point_path as PP
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models

sizes = [int(v) for v in sys.argv[1:]] or [257 * 145, 5197, 1037, 129, 1]
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
cfg = m.point_config()
lib = L.load()
os.environ['DPN_FWD_KERNEL'] = 'tiles'
for n in sizes:
    b = synth_batch(n, dev, seed=1)
    with torch.no_grad():
        heads, evec, statics = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
        x_, y_, t_ = (PP._f32c(b[k]).reshape(-1) for k in ('x', 'y', 't'))
        cd_ = PP._f32c(b['coord_data'])
        st = [PP._f32c(s) for s in statics]
        ws = PP._Workspace(n, cfg.prec, dev)
        nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
        s = PP._stream()
        L.check(lib.dpn_pack_weights(nets, cfg.prec, PP._ptr(ws.packed), s), 'pack')
        geo = cfg.geometry()
        fr = PP._freqs(dev)

        def run(pp, out_n, jac_n, saved):
            os.environ['DPN_FWD_PP'] = '1' if pp else '0'
            L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed),
                                cfg.prec, PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(saved), s), 'fwd')
        res = []
        for pp in (0, 1):
            out_n = torch.full((n, 6), 7.0, device=dev); jac_n = torch.full((n, 6, 3), 7.0, device=dev)
            saved = torch.full((ws.sizes.saved,), 0x5a, dtype=torch.uint8, device=dev)
            run(pp, out_n, jac_n, saved)
            torch.cuda.synchronize()
            res.append((out_n, jac_n, saved))
        (o0, j0, s0), (o1, j1, s1) = res
        print('n = %d: fields bitwise %s (max|d| %.3e)  jac bitwise %s (max|d| %.3e)  saved bitwise %s (%d of %d bytes differ)' % (
            n, bool(torch.equal(o0, o1)), float((o0 - o1).abs().max()), bool(torch.equal(j0, j1)), float((j0 - j1).abs().max()),
            bool(torch.equal(s0, s1)), int((s0 != s1).sum()), s0.numel()), flush=True)
        if n >= 1000:
            ts = {0: [], 1: []}
            for rep in range(8):
                for pp in (0, 1):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    run(pp, o0, j0, s0)
                    e0.record()
                    for _ in range(10): run(pp, o0, j0, s0)
                    e1.record(); torch.cuda.synchronize()
                    ts[pp].append(e0.elapsed_time(e1) * 100)
            for pp in (0, 1):
                v = sorted(ts[pp])
                print('   %-9s fwd+jac+save: min %.1f us  median %.1f us' % ('ping-pong' if pp else 'tiles', v[0], v[len(v) // 2]), flush=True)
os.environ.pop('DPN_FWD_PP', None)
os.environ.pop('DPN_FWD_KERNEL', None)
