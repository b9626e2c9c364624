"""Would the backward stage-1 kernel (power / HBM-write bound) and the weight-gradient kernel (HBM-read bound) finish sooner side by side?
They are dependent inside one step, but a step could pipeline them by point ranges.  This probe answers the hardware question only: the two
kernels on two streams whose CU masks split the chip (hipExtStreamCreateWithCUMask), the weight-gradient kernel reading one operand buffer while
the stage-1 kernel writes another, against the same two launches back to back on the whole chip.
usage: overlap_probe.py [bf16x2|bf16] [CUs for the stage-1 kernel, default 160]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L, point_path as PP
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x2'
cus_a = int(sys.argv[2]) if len(sys.argv) > 2 else 160
n = 257 * 145
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision=prec).to(dev)
b = synth_batch(n, dev, seed=1)
cfg = m.point_config()
lib = L.load()
hip = ctypes.CDLL('libamdhip64.so')
n_cu = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(lo, hi):
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in range(lo, hi):
        mask[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, rc
    return s


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
    s0 = PP._stream()
    L.check(lib.dpn_residual(PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(f_), n, ctypes.byref(geo), ctypes.byref(ph), None, None, None, PP._ptr(g_out), PP._ptr(g_jxi), s0), 'res')
    ops = [torch.empty(ws.sizes.operands, dtype=torch.uint8, device=dev) for _ in range(2)]
    partials = torch.empty(ws.sizes.partials, dtype=torch.uint8, device=dev)
    fr = PP._freqs(dev)

    def k_bwd(op, s):
        L.check(lib.dpn_bwd_points(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec,
                                   PP._ptr(g_out), PP._ptr(g_jxi), PP._ptr(ws.saved), PP._ptr(op), s), 'bwd')

    def k_wgrad(op, s):
        L.check(lib.dpn_wgrad(n, cfg.prec, PP._ptr(g_out), PP._ptr(ws.saved), PP._ptr(op), PP._ptr(partials), s), 'wgrad')
    for op in ops:
        k_bwd(op, s0)
    torch.cuda.synchronize()
    import time

    def timed(fn, reps=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    serial = timed(lambda: (k_bwd(ops[1], s0), k_wgrad(ops[0], s0)))
    only_b = timed(lambda: k_bwd(ops[1], s0))
    only_w = timed(lambda: k_wgrad(ops[0], s0))
    print('%s, %d points, whole chip (%d CUs): stage 1 %.1f us, weight gradients %.1f us, one behind the other %.1f us' % (prec, n, n_cu, only_b, only_w, serial))
    for ca in (cus_a, 128, 192):
        sa, sb = masked_stream(0, ca), masked_stream(ca, n_cu)
        a_only = timed(lambda: k_bwd(ops[1], sa))
        b_only = timed(lambda: k_wgrad(ops[0], sb))

        def both():
            k_bwd(ops[1], sa)
            k_wgrad(ops[0], sb)
        side = timed(both)
        print('   stage 1 on CUs 0-%d alone %.1f us, weight gradients on CUs %d-%d alone %.1f us, side by side %.1f us per pair' % (ca - 1, a_only, ca, n_cu - 1, b_only, side))
        hip.hipStreamDestroy(sa); hip.hipStreamDestroy(sb)
