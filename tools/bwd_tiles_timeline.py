#!/usr/bin/env python
"""Phase timeline of dpn_bwd_tiles_kernel (experiment build: python tools/variant_build.py tl -DDPN_TIMELINE -DTS_TIMELINE [ablation flags
-DTS_ABL_NOSTORE | -DTS_ABL_NOMFMA | -DTS_ABL_NOALOAD]).  Every wave stamps the shader clock at 5 phase boundaries (round 5: one GEMM is left in this kernel); this runs the bench
workload's forward + stage-1 backward and prints the mean cycles of every phase over all waves, plus the kernel's time from HIP events.
usage: bwd_tiles_timeline.py [n] [variant name, default tl]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DPN_LIB'] = os.path.join(ROOT, 'tools', '_variants', 'libdpn_hip_%s.so' % (sys.argv[2] if len(sys.argv) > 2 else 'tl'))
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
NAMES = ['prologue: b1, cotangents, Z0 features -> X and K-layout rows', 'barrier', 'GEMM Z1 = w1 Z0', 'Z1 epilogue (mask, pack, K-layout rows)']
with torch.no_grad():
    heads, evec, statics = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
    x_, y_, t_ = (PP._f32c(b[k]).reshape(-1) for k in ('x', 'y', 't'))
    cd_ = PP._f32c(b['coord_data'])
    st = [PP._f32c(s) for s in statics]
    ws = PP._Workspace(n, cfg.prec, dev)
    ws.alloc_saved()
    nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
    s = PP._stream()
    L.check(lib.dpn_pack_weights(nets, cfg.prec, PP._ptr(ws.packed), s), 'pack')
    geo = cfg.geometry()
    fr = PP._freqs(dev)
    out_n = torch.zeros((n, 6), device=dev); jac_n = torch.zeros((n, 6, 3), device=dev)
    lib.dpn_debug_set_timeline(None)
    L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec,
                        PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(ws.saved), s), 'fwd')
    g_out = torch.randn((n, 6), device=dev); g_jxi = torch.randn((n, 6, 3), device=dev)
    operands = torch.empty(ws.sizes.operands, dtype=torch.uint8, device=dev)
    nblk = ws.sizes.n_pad // 64
    tl = torch.zeros((6, nblk, 4, 48), dtype=torch.int32, device=dev)
    lib.dpn_debug_set_timeline(ctypes.c_void_p(tl.data_ptr()))

    def bwd():
        L.check(lib.dpn_bwd_points(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec,
                                   PP._ptr(g_out), PP._ptr(g_jxi), PP._ptr(ws.saved), PP._ptr(operands), s), 'bwd')
    for _ in range(3):
        bwd()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for e0, e1 in ev:
        e0.record(); bwd(); e1.record()
    torch.cuda.synchronize()
    us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    t = tl.cpu().numpy().astype('int64') & 0xFFFFFFFF
    t = t[1:, :, :, :5]                                      # nets 1..5 (the net-0 workgroups also write the pe6 table)
    d = ((t[..., 1:] - t[..., :-1]) & 0xFFFFFFFF).reshape(-1, 4)
    total = ((t[..., 4] - t[..., 0]) & 0xFFFFFFFF).reshape(-1)
    print('dpn_bwd_tiles_kernel<2>, %d points, library %s: kernel median %.1f us (min %.1f); %d waves sampled, wave lifetime mean %.0f / median %.0f cycles'
          % (n, os.path.basename(os.environ['DPN_LIB']), us[len(us) // 2], us[0], d.shape[0], total.mean(), np.median(total)))
    groups = {'multiply loops (with the saved-operand hand-over inside)': 0.0, 'features / epilogues / stores': 0.0, 'barrier waits': 0.0}
    for i, nm in enumerate(NAMES):
        mean, med = d[:, i].mean(), np.median(d[:, i])
        gname = ('multiply loops (with the saved-operand hand-over inside)' if nm.startswith('GEMM') else 'barrier waits' if nm == 'barrier'
                 else 'features / epilogues / stores')
        groups[gname] += mean
        print('  %2d %-52s mean %8.0f  median %8.0f  (%4.1f %%)' % (i, nm, mean, med, 100.0 * mean / total.mean()))
    for gname, v in groups.items():
        print('  %-58s %8.0f cycles  %4.1f %%' % (gname, v, 100.0 * v / total.mean()))
    print('  MFMA issue alone (multiply loop): %d instructions x 32 cycles = %d cycles per wave (+ 2 transposing MFMAs per saved plane)' % (12 * 12, 12 * 12 * 32))
