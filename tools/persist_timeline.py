#!/usr/bin/env python
"""Phase timeline of dpn_fwd_tiles_persist_kernel (experiment build: python tools/variant_build.py tsptl -DDPN_TIMELINE -DTSP_TIMELINE): cycles of the third item of
every workgroup -- item start -> first GEMM, the layers up to the y epilogue, the tail (waves 0..2: gpe multiply, contraction; wave 3: half of the next item's
features, the wait for the multiplying waves' flags, the other half), the wait at the barrier that ends the item.  usage: persist_timeline.py [n] [variant]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DPN_LIB'] = os.path.join(ROOT, 'tools', '_variants', 'libdpn_hip_%s.so' % (sys.argv[2] if len(sys.argv) > 2 else 'tsptl'))
os.environ['DPN_FWD_KERNEL'] = 'tiles'
os.environ['DPN_FWD_PERSIST'] = '1'
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
    nwg = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
    tl = torch.zeros((nwg, 4, 16), dtype=torch.int32, device=dev)
    lib.dpn_debug_set_timeline(ctypes.c_void_p(tl.data_ptr()))
    for _ in range(3):
        L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec,
                            PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(saved), s), 'fwd')
    torch.cuda.synchronize()
    t = tl.cpu().numpy().astype('int64') & 0xFFFFFFFF
    t = t[t[:, 0, 6] != 0]
    D = lambda a, b_, ws_: (((t[:, ws_, a] - t[:, ws_, b_]) & 0xFFFFFFFF).astype('float64')).mean()
    print('dpn_fwd_tiles_persist_kernel<2>, %d points, %d workgroups sampled (third item of each)' % (n, t.shape[0]))
    for nm, ws_ in (('waves 0-2', slice(0, 3)), ('wave 3', slice(3, 4))):
        print(' %s: item start -> first GEMM %6.0f | layers L1 .. y epilogue %6.0f | two barriers + t1 store %6.0f | tail part 1 (gpe multiply / features 0-5) %6.0f | '
              '(flag set -> contraction / wait for the flags) %6.0f | (- / store + features 6-11) %6.0f | item %6.0f' % (
                  nm, D(1, 0, ws_), D(2, 1, ws_), D(3, 2, ws_), D(4, 3, ws_), D(5, 4, ws_) if nm == 'wave 3' else D(6, 4, ws_), D(6, 5, ws_) if nm == 'wave 3' else 0.0, D(6, 0, ws_)))
