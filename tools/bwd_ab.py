"""A/B of the two backward stage-1 kernels (ring = dpn_bwd_kernel, tiles = dpn_bwd_tiles_kernel): bitwise comparison of the operand buffer
they write (Z1, Z, Z0, G6, gnet) and interleaved HIP-event timing.  usage: bwd_ab.py [bf16x2|bf16] [n ...]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L, point_path as PP
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x2'
sizes = [int(v) for v in sys.argv[2:]] or [257 * 145, 1037, 1]
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision=prec).to(dev)
cfg = m.point_config()
lib = L.load()
for n in sizes:
    b = synth_batch(n, dev, seed=1)
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
        fr = PP._freqs(dev)
        res = {}

        def run(kind, operands):
            os.environ['DPN_BWD_KERNEL'] = kind
            L.check(lib.dpn_bwd_points(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec,
                                       PP._ptr(g_out), PP._ptr(g_jxi), PP._ptr(ws.saved), PP._ptr(operands), s), 'bwd')
        for kind in ('ring', 'tiles'):
            operands = torch.zeros(ws.sizes.operands, dtype=torch.uint8, device=dev)
            run(kind, operands)
            torch.cuda.synchronize()
            res[kind] = operands
        n_pad = ws.sizes.n_pad
        m256, m192 = 6 * cfg.prec * n_pad * 512, 6 * cfg.prec * n_pad * 384
        names = (('Z1', 0, m256), ('Z', m256, 2 * m256), ('Z0', 2 * m256, 2 * m256 + m192), ('G6', 2 * m256 + m192, 2 * m256 + 2 * m192),
                 ('gnet', 2 * m256 + 2 * m192, 2 * m256 + 2 * m192 + 6 * n_pad * 4))
        print('n = %d (%s)' % (n, prec))
        for nm, lo, hi in names:
            d = (res['ring'][lo:hi] != res['tiles'][lo:hi])
            print('   operand %-4s bitwise %s (%d of %d bytes differ)' % (nm, not bool(d.any()), int(d.sum()), hi - lo))
        if n >= 1000:
            ts = {'ring': [], 'tiles': []}
            for rep in range(6):
                for kind in ('ring', 'tiles'):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    run(kind, res[kind])
                    e0.record()
                    for _ in range(5): run(kind, res[kind])
                    e1.record(); torch.cuda.synchronize()
                    ts[kind].append(e0.elapsed_time(e1) * 200)
            for kind in ('ring', 'tiles'):
                v = sorted(ts[kind])
                print('   %-5s bwd stage 1: min %.1f us  median %.1f us' % (kind, v[0], v[len(v) // 2]))
os.environ.pop('DPN_BWD_KERNEL', None)
