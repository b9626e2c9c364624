"""Does the forward point kernel's time depend on WHERE its buffers lie?  (Found by accident: the same kernel on the same inputs measured
413 us in one process and 515 us in another on the same box.)  Times dpn_fwd with the saved-state buffer / the packed weights placed at
different offsets inside one big allocation."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L, point_path as PP
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
dev = torch.device('cuda:0')
n = 257 * 145
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
nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
sz = L.DpnSizes()
lib.dpn_sizes(n, cfg.prec, ctypes.byref(sz))
geo = cfg.geometry()
s = PP._stream()
big = torch.empty(sz.saved + sz.packed + (64 << 20), dtype=torch.uint8, device=dev)
out_n = torch.empty((n, 6), dtype=torch.float32, device=dev)
jac_n = torch.empty((n, 6, 3), dtype=torch.float32, device=dev)
print('big base %#x  saved %d B packed %d B' % (big.data_ptr(), sz.saved, sz.packed))


def timed(packed_ptr, saved_ptr, reps=12):
    L.check(lib.dpn_pack_weights(nets, cfg.prec, ctypes.c_void_p(packed_ptr), s), 'pack')
    ts = []
    for r in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200000)
        e0.record()
        L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(PP._freqs(dev)), ctypes.byref(geo),
                            ctypes.c_void_p(packed_ptr), cfg.prec, PP._ptr(out_n), PP._ptr(jac_n), ctypes.c_void_p(saved_ptr), s), 'fwd')
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


base = big.data_ptr()
base = (base + (2 << 20) - 1) & ~((2 << 20) - 1)                      # 2 MB aligned
pk0 = base
sv0 = base + (16 << 20)
for name, pk_off, sv_off in (('aligned 2 MB / 2 MB', 0, 0), ('saved + 4 KB', 0, 4096), ('saved + 64 KB', 0, 65536), ('saved + 256 B', 0, 256),
                             ('saved + 1 MB', 0, 1 << 20), ('saved + 1 MB + 4 KB', 0, (1 << 20) + 4096), ('packed + 4 KB', 4096, 0),
                             ('packed + 64 KB', 65536, 0), ('packed + 256 B', 256, 0), ('both + 33 KB', 33 << 10, 33 << 10),
                             ('aligned again', 0, 0)):
    med, mn = timed(pk0 + pk_off, sv0 + sv_off)
    print('%-24s median %7.1f us  min %7.1f us' % (name, med, mn))
# fresh torch allocations (what the product does)
for k in range(3):
    ws = PP._Workspace(n, cfg.prec, dev)
    ws.alloc_saved()
    med, mn = timed(ws.packed.data_ptr(), ws.saved.data_ptr())
    print('torch.empty #%d: packed %#x saved %#x  median %7.1f us  min %7.1f us' % (k, ws.packed.data_ptr(), ws.saved.data_ptr(), med, mn))
    keep = torch.empty(37 << 20, dtype=torch.uint8, device=dev)     # shift what the allocator hands out next
