"""Time the finish stage of the point backward (dpn_wgrad_finish_parts) alone: parts = 1 (rows + the W1^T diag(u) side), 2 (the G side + fc2), 3 (all: the product's
three launches), captured 10 calls per graph.  usage: finish_probe.py"""
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
n = 257 * 145
b = synth_batch(n, dev, seed=1)
with torch.no_grad():
    heads, evec, statics = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
    st = [PP._f32c(s) for s in statics]
    ws = PP._Workspace(n, cfg.prec, dev)
    nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
    L.check(lib.dpn_pack_weights(nets, cfg.prec, PP._ptr(ws.packed), PP._stream()), 'pack')
    partials = torch.randn(ws.sizes.partials // 4, device=dev) * 1e-3
    g_heads, g_evec, g_stat = torch.empty_like(heads), torch.empty_like(evec), [torch.empty_like(s) for s in st]
    garr = PP._net_ptrs(g_heads, g_evec, g_stat, cls=L.DpnNetGradPtrs)
    for parts in (3, 1, 2, 3):
        def run():
            for _ in range(10):
                L.check(lib.dpn_wgrad_finish_parts(nets, PP._ptr(ws.packed), n, cfg.prec, PP._ptr(partials), garr, parts, PP._stream()), 'finish')
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
        print('%s finish parts = %d: %.2f us per call' % (os.path.basename(L.LIB_PATH), parts, e0.elapsed_time(e1) / 200 * 1e3))
