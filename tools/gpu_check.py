"""Diagnostic run on a GPU box: HIP point path vs the CPU oracle, printed term by term."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch

import deepphysinet_amd as dpn
from deepphysinet_amd import _lib as L
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
from oracle import dpn_oracle as O
from oracle.fill import fill_state_dict_, synthetic_inputs

dev = torch.device('cuda:0')
lib = L.load()
scratch = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
print('selftest (MFMA layout):', lib.dpn_selftest(ctypes.c_void_p(scratch.data_ptr()), torch.cuda.current_stream().cuda_stream))

N = int(os.environ.get('N', '256'))
GEO = O.Geometry()
inp = synthetic_inputs(N, tag='inter')
st = O.make_state(requires_grad=True)
x, y, t = (inp[k].clone().requires_grad_(True) for k in ('x', 'y', 't'))
total, parts, fn, ph = O.place_one_batch(st, x, y, t, inp['f'], inp['field_data'], inp['coord_data'], inp['forecast_h'], GEO, return_parts=True)
jac = O.jacobian_fields(x, y, t, fn)          # normalised-field Jacobian
names = O.param_names(st)
ref_g = dict(zip(names, torch.autograd.grad(total, [st[n] for n in names])))
ref_parts = np.array([float(p.detach()) for p in parts])
fn = torch.cat(fn, 1).detach()

for prec in ('bf16x2', 'bf16'):
    m = builder_models(**ncep_config(), precision=prec)
    sd = m.physics_net.state_dict(); fill_state_dict_(sd); m.physics_net.load_state_dict(sd)
    m = m.to(dev)
    g = {k: v.to(dev) for k, v in inp.items()}
    cfg = m.point_config()
    with torch.no_grad():
        heads, evec, statics = m.physics_net.field_weights(g['field_data'], g['forecast_h'])
        out_n, jac_n = dpn.pde_fields_and_jacobian(cfg, g['x'], g['y'], g['t'], g['coord_data'], heads, evec, statics)
    torch.cuda.synchronize()
    e_f = (out_n.cpu() - fn).abs().max() / fn.abs().max()
    print('[%s] fields rel err %.3e' % (prec, e_f))
    jr = jac.detach()
    for k in range(6):
        print('   jac[%d] rel err %.3e  (max ref %.3e)' % (k, float((jac_n.cpu()[:, k] - jr[:, k]).abs().max() / jr[:, k].abs().max()), float(jr[:, k].abs().max())))
    m.physics_net.zero_grad()
    terms = m.pde_loss_terms(g['x'], g['y'], g['t'], g['f'], g['field_data'], g['coord_data'], g['forecast_h'])
    tot = terms.sum()
    tot.backward()
    torch.cuda.synchronize()
    mine = terms.detach().cpu().numpy()
    print('   losses mine', mine)
    print('   losses ref ', ref_parts)
    print('   rel err    ', np.abs(mine - ref_parts) / np.abs(ref_parts))
    worst = []
    for n_, p_ in m.physics_net.named_parameters():
        r = ref_g[n_]
        err = float((p_.grad.cpu() - r).abs().max() / (r.abs().max() + 1e-30))
        worst.append((err, n_))
    worst.sort(reverse=True)
    print('   worst param-grad rel errs:')
    for e, n_ in worst[:14]:
        print('      %.3e  %s' % (e, n_))
    print('   median %.3e' % np.median([e for e, _ in worst]))
