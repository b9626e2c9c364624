"""Which kernel puts the chip into the slow state?  300 rounds of [filler, forward point kernel]; the forward kernel's median time of the
last 200 rounds, per filler.  Fillers: nothing; one encoder layer on the row-local fused nodes (dpn_enc_fwd / dpn_enc_bwd) for B fields;
the same layer on the per-GEMM nodes of rounds 1-3 (DPN_ENCODER_UNFUSED=1).  usage: clock_state_probe.py [B ...]"""
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
ws = PP._Workspace(n, cfg.prec, dev)
ws.alloc_saved()
geo = cfg.geometry()
s = PP._stream()
out_n = torch.empty((n, 6), dtype=torch.float32, device=dev)
jac_n = torch.empty((n, 6, 3), dtype=torch.float32, device=dev)
L.check(lib.dpn_pack_weights(nets, cfg.prec, PP._ptr(ws.packed), s), 'pack')
layer = m.physics_net.meta_net.model.encoder.attn_layers[0]


def fwd():
    L.check(lib.dpn_fwd(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(PP._freqs(dev)), ctypes.byref(geo),
                        PP._ptr(ws.packed), cfg.prec, PP._ptr(out_n), PP._ptr(jac_n), PP._ptr(ws.saved), s), 'fwd')


def layer_filler(B, unfused, reps):
    x = torch.randn(B, 287, 256, device=dev)

    def run():
        __import__('deepphysinet_amd.config').config.set_switches(encoder_unfused=bool(unfused))
        with torch.no_grad():
            for _ in range(reps):
                layer(x)
    return run


fillers = {'nothing (forward kernels back to back)': lambda: None}
for B in [int(v) for v in sys.argv[1:]] or [1, 61]:
    fillers['fused layer forward, %d fields' % B] = layer_filler(B, False, 1)
    fillers['per-GEMM layer forward, %d fields' % B] = layer_filler(B, True, 1)
fillers['nothing again'] = lambda: None
for name, fill in fillers.items():
    ts = []
    for r in range(300):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fill()
        e0.record()
        fwd()
        e1.record()
        if r % 50 == 49:
            torch.cuda.synchronize()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in ts[100:])
    print('%-44s forward kernel median %6.1f us  (10 %% %6.1f, 90 %% %6.1f)' % (name, v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10]))
