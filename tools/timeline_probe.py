#!/usr/bin/env python
"""Per-step shader-clock timeline of dpn_fwd_kernel (experiment build with -DDPN_TIMELINE).

    python tools/timeline_probe.py --build          # here (hipcc cross-compiles): tools/_variants/libdpn_hip_timeline.so
    python tools/timeline_probe.py [bf16|bf16x2]    # on the GPU box: cycles per pipeline step, by stage, for a sample of workgroups

Stamp i of a wave = s_memtime at the start of pipeline step i - 2 (0 = kernel entry, 1 = prologue done, 62 = exit); the steps are the
54 weight chunks of the forward kernel: L1 8 x 12 k-steps | L2 8 x 16 | Wd 8 x 12 | fc1 8 x 16 | v 8 x 16 | y 8 x 16 | gpe 6 x 16.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', '_variants', 'libdpn_hip_timeline.so')


def build(extra=()):
    from deepphysinet_amd import build as B
    obj = os.path.join(B.HERE, 'csrc', '_obj')
    os.makedirs(obj, exist_ok=True)
    objs = []
    for src, flags, name in B.UNITS:
        o = os.path.join(obj, 'tl_' + name)
        cmd = ['hipcc', *B.COMMON, *flags, '-DDPN_TIMELINE', *extra, '-I' + os.path.join(ROOT, 'include'), '-c', src, '-o', o]
        subprocess.run(cmd, check=True)
        objs.append(o)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', LIB], check=True)
    print(LIB)


STAGES = [('L1', 0, 8, 12), ('L2', 8, 16, 16), ('Wd', 16, 24, 12), ('fc1', 24, 32, 16), ('v', 32, 40, 16), ('y', 40, 48, 16), ('gpe', 48, 54, 16)]


def main():
    if '--build' in sys.argv:
        return build([a for a in sys.argv[1:] if a.startswith('-D')])       # e.g. --build -DDPN_DMA_INTERLEAVE
    os.environ['DPN_LIB'] = LIB
    import numpy as np
    import torch
    from bench import synth_batch
    from deepphysinet_amd import _lib as L
    from deepphysinet_amd import point_path as PP
    from deepphysinet_amd.configs import ncep_config
    from deepphysinet_amd.interface import builder_models
    prec = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'bf16x2'
    n = 257 * 145
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    m = builder_models(**ncep_config(), precision=prec).to(dev)
    b = synth_batch(n, dev, seed=1)
    lib = L.load()
    lib.dpn_debug_set_timeline.argtypes = [ctypes.c_void_p]
    cfg = m.point_config()
    nblk = (n + 127) // 128
    nw = 8 if os.environ.get('DPN_FWD2') else 4                  # waves per workgroup (8: the shelved eight-wave kernel, experiment builds with -DDPN_EXPERIMENT_FWD2)
    tl = torch.zeros((nblk, 6, 8, 64), dtype=torch.int32, device=dev)
    with torch.no_grad():
        heads, evec, statics = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
        x_, y_, t_ = (PP._f32c(b[k]).reshape(-1) for k in ('x', 'y', 't'))
        cd_ = PP._f32c(b['coord_data'])
        st = [PP._f32c(s) for s in statics]
        ws = PP._Workspace(n, cfg.prec, dev)
        nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
        for it in range(3):
            lib.dpn_debug_set_timeline(ctypes.c_void_p(tl.data_ptr()) if it == 2 else None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            PP._forward_points(cfg, ws, nets, x_, y_, t_, None, cd_, True, True)
            e1.record()
            torch.cuda.synchronize()
        print('kernel + pack (instrumented): %.1f us' % (e0.elapsed_time(e1) * 1e3))
    lib.dpn_debug_set_timeline(None)
    T = tl.cpu().numpy().astype(np.int64)[:, :, :nw] & 0xFFFFFFFF
    if os.environ.get('DPN_FWD_PHASES'):                   # build with -DDPN_FWD_PHASES: slots 56..61 hold phase sums over the 54 steps
        P = T[..., 56:62].astype(np.float64)
        names = ('vmcnt wait', 'barrier', 'DMA issue', 'reads + MFMAs', '(unused)', 'epilogue')
        print('%s: cycles per wave over the 54 pipeline steps, by phase: %s | sum %.0f' %
              (prec, '  '.join('%s %.0f' % (nm, P[..., k].mean()) for k, nm in enumerate(names) if k != 4), P.sum(-1).mean()))
        print('   per step: %s' % '  '.join('%s %.0f' % (nm, P[..., k].mean() / 54) for k, nm in enumerate(names) if k != 4))
    d = (T[..., 1:] - T[..., :-1]) & 0xFFFFFFFF            # wrap-safe deltas, [blk, net, wave, 63]
    # a stamp that was never written (early exit) or a wave whose 32-bit clock word wrapped twice gives a delta of ~2^32: those samples are
    # DROPPED from every statistic below (round 2's stage table averaged them in: "fc1 1 235 626 cycles per chunk")
    d = np.where(d < (1 << 26), d, np.nan)
    total = (T[..., 62] - T[..., 0]) & 0xFFFFFFFF
    print('%s: waves %d, cycles per wave entry -> exit: mean %.0f  min %d  max %d' % (prec, total.size, total.mean(), total.min(), total.max()))
    print('prologue (entry -> ring primed): %.0f   primed -> first step: %.0f' % (np.nanmean(d[..., 0]), np.nanmean(d[..., 1])))
    nsplit = (3 if prec == 'bf16x2' else 1) * (4.0 / nw)         # the eight-wave kernel's waves multiply half of K each
    print('%-5s %6s %10s %10s %10s %12s' % ('stage', 'chunks', 'cyc/chunk', 'min', 'max', 'MFMA-bound'))
    for name, c0, c1, nk in STAGES:
        seg = d[..., 2 + c0:2 + c1]                        # step C spans stamp 2+C -> 3+C; the last step of the kernel ends at stamp 62
        if c1 == 54:
            seg = np.concatenate([d[..., 2 + c0:2 + c1 - 1], ((T[..., 62] - T[..., 2 + 53]) & 0xFFFFFFFF)[..., None]], axis=-1)
        print('%-5s %6d %10.0f %10d %10d %12d' % (name, c1 - c0, np.nanmean(seg), np.nanmin(seg), np.nanmax(seg), int(nk * 32 * nsplit)))
    # per-chunk profile of one workgroup in the middle of the grid, wave 0..3
    blk = nblk // 2
    for w in range(nw):
        print('blk %d net 2 wave %d:' % (blk, w), ' '.join(('%d' % v) if v == v else 'dropped' for v in d[blk, 2, w, :56]))
    starts = T[blk, 2, :, 2:56]
    print('inter-wave skew at step starts (max - min over the waves), blk %d net 2:' % blk, ' '.join('%d' % v for v in ((starts.max(0) - starts.min(0)) & 0xFFFFFFFF)))


if __name__ == '__main__':
    main()
