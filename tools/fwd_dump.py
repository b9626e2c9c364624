"""SHA-256 of everything the forward + Jacobian launch and the backward stage-1 launch write (fields, Jacobian, saved state, operand rows), for the library in DPN_LIB
(default: the product build) -- run it for two builds and compare the lines: a change that must not change a bit.  usage: fwd_dump.py [n ...]   (DPN_FWD_PP / DPN_FWD_KERNEL apply)"""
import os, sys, ctypes, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L, point_path as PP
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models

sizes = [int(v) for v in sys.argv[1:]] or [257 * 145, 1037, 129]
dev = torch.device('cuda:0')
for prec in ('bf16x2', 'bf16'):
    torch.manual_seed(1)
    m = builder_models(**ncep_config(), precision=prec).to(dev)
    cfg = m.point_config()
    lib = L.load()
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
            out_n = torch.full((n, 6), 7.0, device=dev); jac_n = torch.full((n, 6, 3), 7.0, device=dev)
            saved = torch.full((ws.sizes.saved,), 0x5a, dtype=torch.uint8, device=dev)
            L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed),
                                cfg.prec, PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(saved), s), 'fwd')
            torch.manual_seed(3)
            g_out = torch.randn((n, 6), device=dev); g_j = torch.randn((n, 6, 3), device=dev)
            ops = torch.full((ws.sizes.operands,), 0x5a, dtype=torch.uint8, device=dev)
            L.check(lib.dpn_bwd_points(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed),
                                       cfg.prec, PP._ptr(g_out), PP._ptr(g_j), PP._ptr(saved), PP._ptr(ops), s), 'bwd')
            torch.cuda.synchronize()
            hs = [hashlib.sha256(t_.cpu().numpy().tobytes()).hexdigest()[:16] for t_ in (out_n, jac_n, saved, ops)]
            print('%s n=%d fields %s jac %s saved %s operands %s' % (prec, n, *hs), flush=True)
