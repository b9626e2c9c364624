#!/usr/bin/env python
"""Where a dpn_wgrad_kernel workgroup spends its cycles: per 32-point tile, the counted vmcnt wait, the barrier, the LDS-DMA issue of the
next tile and the multiply (experiment build, s_memtime around each phase; the stamps serialise the LDS queue, so read the split, not the sum).

    python tools/timeline_probe.py --build -DDPN_WGRAD_PHASES -DDPN_EXPERIMENT_SPLITS     # here
    python tools/wgrad_phase_probe.py [bf16|bf16x2]                                        # on the GPU box
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DPN_LIB'] = os.path.join(ROOT, 'tools', '_variants', 'libdpn_hip_timeline.so')


def main():
    import numpy as np
    import torch
    from bench import synth_batch
    from deepphysinet_amd import _lib as L
    from deepphysinet_amd import point_path as PP
    from deepphysinet_amd.configs import ncep_config
    from deepphysinet_amd.interface import builder_models
    prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x2'
    n = 257 * 145
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    m = builder_models(**ncep_config(), precision=prec).to(dev)
    b = synth_batch(n, dev, seed=1)
    lib = L.load()
    lib.dpn_debug_set_wgrad_phases.argtypes = [ctypes.c_void_p]
    cfg = m.point_config()
    with torch.no_grad():
        heads, evec, statics = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
    x_, y_, t_ = (PP._f32c(b[k]).reshape(-1) for k in ('x', 'y', 't'))
    cd_ = PP._f32c(b['coord_data'])
    st = [PP._f32c(s) for s in statics]
    nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
    g_out = torch.randn(n, 6, device=dev) * 1e-3
    g_jxi = torch.randn(n, 6, 3, device=dev) * 1e-3
    geo = cfg.geometry()
    ws = PP._Workspace(n, cfg.prec, dev)
    PP._forward_points(cfg, ws, nets, x_, y_, t_, None, cd_, True, True)
    operands = torch.empty(ws.sizes.operands, dtype=torch.uint8, device=dev)
    partials = torch.empty(ws.sizes.partials, dtype=torch.uint8, device=dev)
    L.check(lib.dpn_bwd_points(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(PP._freqs(dev)), ctypes.byref(geo),
                               PP._ptr(ws.packed), cfg.prec, PP._ptr(g_out), PP._ptr(g_jxi), PP._ptr(ws.saved), PP._ptr(operands), PP._stream()), 'bwd')
    ph = torch.zeros((6, 64, 8, 8), dtype=torch.int32, device=dev)
    for it in range(3):
        lib.dpn_debug_set_wgrad_phases(ctypes.c_void_p(ph.data_ptr()) if it == 2 else None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.dpn_wgrad(n, cfg.prec, PP._ptr(g_out), PP._ptr(ws.saved), PP._ptr(operands), PP._ptr(partials), PP._stream()), 'wgrad')
        e1.record()
        torch.cuda.synchronize()
    lib.dpn_debug_set_wgrad_phases(None)
    print('%s: kernel (instrumented) %.1f us' % (prec, e0.elapsed_time(e1) * 1e3))
    P = ph.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    P = P.reshape(-1, 8, 8)
    P = P[P[:, 0, 4] > 0]                                   # workgroups that ran
    print('workgroups %d; cycles per tile (100 MHz s_memtime ticks x 24 = shader cycles at 2.4 GHz is NOT applied: raw ticks)' % len(P))
    names = ('wait vmcnt', 'barrier', 'issue DMA', 'multiply')
    for prod in range(4):
        Q = P[P[:, 0, 5] == prod]
        if not len(Q):
            continue
        tiles = Q[:, :, 4].astype(np.float64)
        per = [Q[:, :, k] / tiles for k in range(4)]
        tot = sum(per)
        print('  product %d: %3d workgroups, %5.1f tiles each; ticks per tile: %s | sum %.1f' %
              (prod, len(Q), tiles.mean(), '  '.join('%s %.1f' % (nm, v.mean()) for nm, v in zip(names, per)), tot.mean()))
        for w in range(8):
            print('      wave %d: %s' % (w, '  '.join('%s %.1f' % (nm, v[:, w].mean()) for nm, v in zip(names, per))))


if __name__ == '__main__':
    main()
