"""dpn_wgrad16 / dpn_gemm16 in isolation: time per launch for n problems of [rows x 256]^T [rows x 256] (events over 200 launches; inputs rewritten
by a dummy kernel in between so that they are L2-cold like in the step).  usage: wgrad16_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepphysinet_amd.encoder_ops import wgrad16
dev = torch.device('cuda:0')


def run(n, rows, cold):
    G = [torch.randn(rows, 256, device=dev) for _ in range(n)]
    X = [torch.randn(rows, 256, device=dev) for _ in range(n)]
    dW = [torch.empty(256, 256, device=dev) for _ in range(n)]
    db = [torch.empty(256, device=dev) for _ in range(n)]
    flush = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    probs = list(zip(G, X, dW, db))
    for _ in range(5):
        wgrad16(probs)
    torch.cuda.synchronize()
    ts = []
    for _ in range(50):
        if cold:
            flush.add_(1.0)                                  # 256 MB read + write: evicts L2 and most of the MALL
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); wgrad16(probs); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
    ref = G[0].double().T @ X[0].double()
    err = float((dW[0].double() - ref).abs().max() / ref.abs().max())
    print('n = %2d problems, rows = %5d, %s: median %6.1f us (min %6.1f)   max rel err %.1e' % (n, rows, 'cold' if cold else 'warm', v[len(v) // 2], v[0], err))


for n, rows in ((1, 32), (1, 287), (25, 32), (25, 287), (25, 1024)):
    for cold in (False, True):
        run(n, rows, cold)
