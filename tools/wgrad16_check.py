"""dpn_wgrad16 against fp64 for mixed problem lists (equal tiles, 512-wide ones, the token convolution's 7 215 columns; one and two row slices): prints the
maximum relative error of dW / maximum absolute error of db per problem.  usage: wgrad16_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepphysinet_amd.encoder_ops import wgrad16
dev = torch.device('cuda:0')
torch.manual_seed(0)
for shapes in ([(256, 256)] * 6, [(256, 256)] * 4 + [(512, 256), (256, 512)], [(256, 256)] * 25 + [(256, 7215)], [(256, 7215)]):
    for rows in (287, 2 * 287):
        G = [torch.randn(rows, m, device=dev) for m, n in shapes]
        X = [torch.randn(rows, n, device=dev) for m, n in shapes]
        dW = [torch.zeros(m, n, device=dev) for m, n in shapes]
        db = [torch.zeros(m, device=dev) for m, n in shapes]
        wgrad16(list(zip(G, X, dW, db)))
        torch.cuda.synchronize()
        errs = []
        for g, x, w, b in zip(G, X, dW, db):
            ref = g.double().T @ x.double()
            errs.append('%.0e/%.0e' % (float((w.double() - ref).abs().max() / ref.abs().max()), float((b.double() - g.double().sum(0)).abs().max())))
        print(len(shapes), rows, ' '.join(errs))
