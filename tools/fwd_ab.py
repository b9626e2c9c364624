"""A/B of the two forward + Jacobian kernels (ring = dpn_fwd_kernel, tiles = dpn_fwd_tiles_kernel): bitwise comparison of everything
they write and interleaved HIP-event timing.  usage: fwd_ab.py [bf16x2|bf16] [n ...]"""
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
        x_, y_, t_ = (PP._f32c(b[k]).reshape(-1) for k in ('x', 'y', 't'))
        cd_ = PP._f32c(b['coord_data'])
        st = [PP._f32c(s) for s in statics]
        ws = PP._Workspace(n, cfg.prec, dev)
        nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
        s = PP._stream()
        L.check(lib.dpn_pack_weights(nets, cfg.prec, PP._ptr(ws.packed), s), 'pack')
        geo = cfg.geometry()
        fr = PP._freqs(dev)
        res = {}

        def run(kind, out_n, jac_n, saved):
            os.environ['DPN_FWD_KERNEL'] = kind
            L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed),
                                cfg.prec, PP._ptr(out_n), PP._ptr(jac_n) if jac_n is not None else None, PP._ptr(saved) if saved is not None else None, s), 'fwd')

        for kind in ('ring', 'tiles'):
            out_n = torch.zeros((n, 6), device=dev); jac_n = torch.zeros((n, 6, 3), device=dev)
            saved = torch.zeros(ws.sizes.saved, dtype=torch.uint8, device=dev)
            run(kind, out_n, jac_n, saved)
            torch.cuda.synchronize()
            res[kind] = (out_n, jac_n, saved)
            o2 = torch.zeros((n, 6), device=dev)
            run(kind, o2, None, None)                      # fields-only early exit
            torch.cuda.synchronize()
            res[kind + '_fields'] = o2
        (o0, j0, s0), (o1, j1, s1) = res['ring'], res['tiles']
        n_pad = ws.sizes.n_pad
        mat = 6 * cfg.prec * n_pad * 512
        names = (('T1', 0, mat), ('M2', mat, mat + 6 * n_pad * 512), ('m1', mat + 6 * n_pad * 512, mat + 6 * n_pad * 512 + 6 * n_pad * 32))
        print('n = %d (%s): out max|d| %.3e (rel %.3e)  fields-only vs full: ring %.1e tiles %.1e   jac bitwise %s (max|d| %.3e)' % (
            n, prec, float((o0 - o1).abs().max()), float((o0 - o1).abs().max() / o0.abs().max()),
            float((res['ring_fields'] - o0).abs().max()), float((res['tiles_fields'] - o1).abs().max()),
            bool(torch.equal(j0, j1)), float((j0 - j1).abs().max())))
        for nm, lo, hi in names:
            d = (s0[lo:hi] != s1[lo:hi])
            print('   saved %-3s bitwise %s (%d of %d bytes differ)' % (nm, not bool(d.any()), int(d.sum()), hi - lo))
        if n >= 1000:
            saved = res['tiles'][2]
            ts = {'ring': [], 'tiles': []}
            for rep in range(6):
                for kind in ('ring', 'tiles'):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    run(kind, o0, j0, saved)
                    e0.record()
                    for _ in range(5): run(kind, o0, j0, saved)
                    e1.record(); torch.cuda.synchronize()
                    ts[kind].append(e0.elapsed_time(e1) * 200)
            for kind in ('ring', 'tiles'):
                v = sorted(ts[kind])
                print('   %-5s fwd+jac+save: min %.1f us  median %.1f us' % (kind, v[0], v[len(v) // 2]))
os.environ.pop('DPN_FWD_KERNEL', None)
