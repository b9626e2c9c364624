"""Time dpn_attn_fwd / dpn_attn_bwd (graphs of 40 launches) -- also used with stage-exit experiment builds (DPN_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepphysinet_amd import _lib as L
from deepphysinet_amd.encoder_ops import _p, _s
dev = torch.device('cuda:0')
q, k, v, go = (torch.randn(287, 256, device=dev) for _ in range(4))
o, P = torch.empty_like(q), torch.empty(8, 288, 288, device=dev)
dq, dk, dv = (torch.empty_like(q) for _ in range(3))
lib = L.load()
L.check(lib.dpn_attn_fwd(_p(q), _p(k), _p(v), 287, 1, _p(o), _p(P), _s()), 'attn')


def timed(fn, name):
    def run():
        for _ in range(40):
            fn()
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
    print('%s %s %.2f us/launch' % (os.path.basename(L.LIB_PATH), name, e0.elapsed_time(e1) / 800 * 1e3))


which = sys.argv[1] if len(sys.argv) > 1 else 'both'
if which in ('fwd', 'both'):
    timed(lambda: L.check(lib.dpn_attn_fwd(_p(q), _p(k), _p(v), 287, 1, _p(o), _p(P), _s()), 'attn'), 'attn_fwd')
if which in ('bwd', 'both'):
    timed(lambda: L.check(lib.dpn_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(P), _p(go), 287, 1, _p(dq), _p(dk), _p(dv), _s()), 'attn'), 'attn_bwd')
if which in ('fwd16', 'both'):
    timed(lambda: L.check(lib.dpn_attn16_fwd(_p(q), _p(k), _p(v), 287, 1, _p(o), _p(P), _s()), 'attn16'), 'attn16_fwd')
