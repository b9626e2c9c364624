#!/usr/bin/env python
"""Interval timeline of dpn_fwd_pp_kernel (experiment build: python tools/variant_build.py pptl -DDPN_TIMELINE -DPP_TIMELINE [-D...]).

Every wave stamps the shader clock at the boundaries of the ten intervals of its workgroup's SECOND item; this prints, per interval, the work time and the
wait at the barrier that ends it, for the two groups (group 1 runs one interval behind group 0).  usage: pp_timeline.py [n] [variant name, default pptl]
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DPN_LIB'] = os.path.join(ROOT, 'tools', '_variants', 'libdpn_hip_%s.so' % (sys.argv[2] if len(sys.argv) > 2 else 'pptl'))
os.environ['DPN_FWD_KERNEL'] = 'tiles'
os.environ['DPN_FWD_PP'] = '1'
import numpy as np
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
NAMES = ['E0 contraction(prev)', 'E0 vectors + pe3 features', 'M1 w1.pe3', 'E1 relu, mask, pack, store', 'MA A.h1', 'P6 pe6 features, store', 'MB B.pe6',
         'E2 mask, t2, field share, store', 'My A^T.t2 (+M2 save)', 'Ey mask, store', 'Mg w1^T.t1 (+T1 save)']
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
    nwg = torch.cuda.get_device_properties(dev).multi_processor_count
    tl = torch.zeros((nwg, 8, 32), dtype=torch.int32, device=dev)
    lib.dpn_debug_set_timeline(ctypes.c_void_p(tl.data_ptr()))
    for _ in range(3):
        L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec,
                            PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(saved), s), 'fwd')
    torch.cuda.synchronize()
    t = tl.cpu().numpy().astype('int64') & 0xFFFFFFFF
    ok = (t[:, 0, 20] != 0)                                  # workgroups that ran at least three items
    t = t[ok]
    D = lambda a, b_: ((t[..., a] - t[..., b_]) & 0xFFFFFFFF).astype('float64')
    # work of interval i: E0 = [0 -> 21 -> 1], M1 = [2 -> 3], E1 = [4 -> 5], ...; wait behind interval i: [end_i -> start_{i+1}]
    starts = [0, 21, 2, 4, 6, 8, 10, 12, 14, 16, 18]
    ends = [21, 1, 3, 5, 7, 9, 11, 13, 15, 17, 19]
    nexts = [None, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20]
    item = D(20, 0)
    print('dpn_fwd_pp_kernel<2>, %d points, %d workgroups sampled: one item (128 points, ten intervals) takes mean %.0f / median %.0f cycles per group' % (
        n, t.shape[0], item.mean(), np.median(item)))
    for gname, ws_ in (('group 0, waves 0-2', slice(0, 3)), ('group 0, wave 3', slice(3, 4)), ('group 1, waves 0-2', slice(4, 7)), ('group 1, wave 3', slice(7, 8))):
        print(' %s' % gname)
        tot_w = tot_b = 0.0
        for i, nm in enumerate(NAMES):
            wk = D(ends[i], starts[i])[:, ws_].mean()
            bw = D(nexts[i], ends[i])[:, ws_].mean() if nexts[i] is not None else 0.0
            tot_w += wk; tot_b += bw
            print('   %-36s work %8.0f   barrier wait behind it %8.0f' % (nm, wk, bw))
        print('   %-36s work %8.0f   barrier waits %8.0f   (item %.0f)' % ('sum', tot_w, tot_b, item[:, ws_].mean()))
    print(' MFMA issue alone: 864 instructions x 32 cycles = 27648 cycles per wave and item; two groups: 55296 per item pair')
