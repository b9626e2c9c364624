"""Time dpn_pack_weights (the fused form: dpn_pack_fused_kernel) alone, captured 20 launches per graph -- also with role-ablation builds (DPN_LIB).  usage: pack_probe.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L, point_path as PP
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
cfg = m.point_config()
lib = L.load()
b = synth_batch(4096, dev, seed=1)
with torch.no_grad():
    heads, evec, statics = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
    st = [PP._f32c(s) for s in statics]
    ws = PP._Workspace(4096, cfg.prec, dev)
    nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
    def run():
        for _ in range(20):
            L.check(lib.dpn_pack_weights(nets, cfg.prec, PP._ptr(ws.packed), PP._stream()), 'pack')
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print('%s dpn_pack_weights (fused form) %.2f us/launch' % (os.path.basename(L.LIB_PATH), e0.elapsed_time(e1) / 400 * 1e3))
