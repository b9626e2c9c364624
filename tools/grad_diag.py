"""Per-parameter gradient error of the HIP step against the fp32 CPU oracle at n points (worst ten), and the loss terms.
usage: grad_diag.py [n=5197] [prec=bf16x2]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import test_gpu_parity as T
from oracle.fill import synthetic_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5197
prec = sys.argv[2] if len(sys.argv) > 2 else 'bf16x2'
inp = synthetic_inputs(n, tag='inter')
ref = T._oracle(inp)
m = T._model(prec); g = T._gpu(inp)
m.physics_net.zero_grad()
terms = m.pde_loss_terms(g['x'], g['y'], g['t'], g['f'], g['field_data'], g['coord_data'], g['forecast_h'])
terms.sum().backward()
print('loss rel err', np.abs(terms.detach().cpu().numpy() - ref['parts']) / np.abs(ref['parts']))
rows = []
for name, p in m.physics_net.named_parameters():
    r = ref['grads'][name]
    d = (p.grad.cpu() - r).abs()
    rows.append((float(d.max() / (r.abs().max() + 1e-30)), float(d.pow(2).mean().sqrt() / (r.pow(2).mean().sqrt() + 1e-30)), name, float(r.abs().max())))
rows.sort(reverse=True)
for e, l2, name, mx in rows[:12]:
    print('%.2e  relL2 %.2e  max|g| %.2e  %s' % (e, l2, mx, name))
