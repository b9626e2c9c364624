#!/usr/bin/env python
"""Phase timeline of dpn_fwd_tiles_kernel (experiment build: python tools/variant_build.py tl -DDPN_TIMELINE -DTS_TIMELINE).

Every wave stamps the shader clock at 21 phase boundaries; this prints, per phase, the mean / median cycles over all waves of the launch,
split into multiply phases (the five GEMM loops of the fused form), epilogue / feature phases and barrier waits.  usage: tiles_timeline.py [n] [variant name, default tl]
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DPN_LIB'] = os.path.join(ROOT, 'tools', '_variants', 'libdpn_hip_%s.so' % (sys.argv[2] if len(sys.argv) > 2 else 'tl'))
os.environ['DPN_FWD_KERNEL'] = 'tiles'
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L, point_path as PP
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models

n = int(sys.argv[1]) if len(sys.argv) > 1 else 257 * 145
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
cfg = m.point_config()
lib = L.load()
lib.dpn_debug_set_timeline.argtypes = [ctypes.c_void_p]
b = synth_batch(n, dev, seed=1)
NAMES = ['prologue: vectors, pe3 features', 'barrier', 'L1 multiply (w1.pe3)', 'L1 epilogue ((w2^T wo).h1, relu, mask, pack)', 'barrier A + store', 'barrier B',
         'A multiply (A.h1)', 'pe6 features (+ (Wd^T wo).pe6)', 'barrier A + store', 'barrier B', 'B multiply (B.pe6)', 'pre2 epilogue (mask, t2, field share)',
         'barrier A + store', 'barrier B + field', 'y multiply (A^T.t2, M2 save inside)', 'y epilogue (mask, T1 save)', 'barrier A + store', 'barrier B',
         'gpe multiply (w1^T.t1, T1 save inside)', 'Jacobian contraction']
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
    out_n = torch.zeros((n, 6), device=dev); jac_n = torch.zeros((n, 6, 3), device=dev)
    saved = torch.zeros(ws.sizes.saved, dtype=torch.uint8, device=dev)
    nblk = ws.sizes.n_pad // 64
    tl = torch.zeros((6, nblk, 4, 48), dtype=torch.int32, device=dev)
    lib.dpn_debug_set_timeline(ctypes.c_void_p(tl.data_ptr()))
    for _ in range(3):
        L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec,
                            PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(saved), s), 'fwd')
    torch.cuda.synchronize()
    t = tl.cpu().numpy().astype('int64') & 0xFFFFFFFF
    import numpy as np
    t = t[:, :, :3, :21]                                    # waves 0..2 run every phase
    d = (t[..., 1:] - t[..., :-1]) & 0xFFFFFFFF
    d = d.reshape(-1, 20)
    total = ((t[..., 20] - t[..., 0]) & 0xFFFFFFFF).reshape(-1)
    print('dpn_fwd_tiles_kernel<2>, %d points: %d waves sampled, wave lifetime mean %.0f / median %.0f cycles' % (n, d.shape[0], total.mean(), np.median(total)))
    groups = {'multiply': 0.0, 'epilogue / features': 0.0, 'barrier + store': 0.0}
    for i, nm in enumerate(NAMES):
        mean, med = d[:, i].mean(), np.median(d[:, i])
        g = 'multiply' if 'multiply' in nm else ('barrier + store' if 'barrier' in nm else 'epilogue / features')
        groups[g] += mean
        print('  %2d %-36s mean %8.0f  median %8.0f  (%4.1f %%)' % (i, nm, mean, med, 100.0 * mean / total.mean()))
    for g, v in groups.items():
        print('  %-22s %8.0f cycles  %4.1f %%' % (g, v, 100.0 * v / total.mean()))
    print('  MFMA issue alone: %d instructions x 32 cycles = %d cycles per wave' % (72 * 12, 72 * 12 * 32))
