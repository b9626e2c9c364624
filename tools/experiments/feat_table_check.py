"""The per-point (sin, cos) feature table against in-kernel evaluation: the tile-split forward kernel with a saved-state buffer (table) and
without one (angles evaluated in the kernel) must write the same Jacobian bit for bit.  Prints where they differ.  usage: feat_table_check.py [n]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    saved = torch.zeros(ws.sizes.saved, dtype=torch.uint8, device=dev)

    def run(with_saved):
        out_n = torch.zeros((n, 6), device=dev); jac_n = torch.zeros((n, 6, 3), device=dev)
        L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(fr), ctypes.byref(geo), PP._ptr(ws.packed), cfg.prec,
                            PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(saved) if with_saved else None, s), 'fwd')
        torch.cuda.synchronize()
        return out_n, jac_n
    o0, j0 = run(False)
    for rep in range(4):
        o1, j1 = run(True)
        bad = (j0.view(torch.int32) != j1.view(torch.int32)).nonzero()
        print('rep %d: out bitwise %s, jac differs at %d of %d entries, max|d| %.3e' % (rep, torch.equal(o0, o1), bad.shape[0], j0.numel(), float((j0 - j1).abs().max())))
        if bad.shape[0]:
            pts = bad[:, 0]
            print('   points %s ...' % pts[:12].tolist())
            print('   tiles (64) %s' % sorted(set((pts // 64).tolist()))[:20])
            print('   lane in tile %s' % sorted(set((pts % 64).tolist()))[:64])
            print('   nets %s coords %s' % (sorted(set(bad[:, 1].tolist())), sorted(set(bad[:, 2].tolist()))))
            k = bad[0]
            print('   first: point %d net %d coord %d: %r vs %r' % (k[0], k[1], k[2], float(j0[k[0], k[1], k[2]]), float(j1[k[0], k[1], k[2]])))
    if os.environ.get('DPN_LIB', '').find('feat') >= 0:
        lib.dpn_debug_set_timeline.argtypes = [ctypes.c_void_p]
        import numpy as np
        for rep in range(2):
            dbg = torch.zeros(16 + 8 * 4000 + 6 * 600 * 4 * 48, dtype=torch.int32, device=dev)
            lib.dpn_debug_set_timeline(ctypes.c_void_p(dbg.data_ptr()))
            run(True)
            d = dbg.cpu().numpy().view(np.uint32)
            cnt = int(d[0])
            print('in-kernel compare: %d mismatching table values' % cnt)
            rec = d[16:16 + 8 * min(cnt, 4000)].reshape(-1, 8)
            for r_ in rec[:24]:
                print('   tile %4d  net/wave/t/p %04d  lane %2d  r %2d  table %r  kernel %r  fetched again (sin or cos) %r' % (r_[0], r_[1], r_[2], r_[3], float(np.array(r_[4], dtype=np.uint32).view(np.float32)),
                                                                                               float(np.array(r_[5], dtype=np.uint32).view(np.float32)), float(np.array(r_[6], dtype=np.uint32).view(np.float32))))
            if cnt:
                print('   lanes %s' % sorted(set(rec[:, 2].tolist())))
                print('   r %s' % sorted(set(rec[:, 3].tolist())))
                print('   net/wave/t/p %s' % sorted(set(rec[:, 1].tolist()))[:40])
        lib.dpn_debug_set_timeline(None)
    # the table as it sits in memory after a run: zeros where a sine / cosine pair cannot be zero?  (sin^2 + cos^2 = 1 for every pair)
    import numpy as np

    n_pad = (n + 127) // 128 * 128
    state = 6 * 2 * n_pad * 512 + 6 * n_pad * 512 + 6 * n_pad * 32
    for rep in range(3):
        run(True)
        tab = saved[state:state + (n_pad // 64) * 98304].cpu().numpy().view(np.float32).reshape(n_pad // 64, 2, 12, 2, 64, 4, 2)
        norm = tab[..., 0] ** 2 + tab[..., 1] ** 2
        bad = np.argwhere(np.abs(norm - 1.0) > 1e-3)
        print('table in memory after run %d: %d of %d pairs with sin^2 + cos^2 != 1' % (rep, bad.shape[0], norm.size))
        for b_ in bad[:10]:
            print('   tile %d table %d ks %d p %d lane %d q %d: %r' % (*b_, tab[tuple(b_)].tolist()))
