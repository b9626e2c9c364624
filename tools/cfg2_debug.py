"""configs[2] numerics: the 61-lead step, eager, a few optimiser steps; per step the loss, the worst field's six terms and whether
gradients / parameters are finite.  usage: cfg2_debug.py [leads] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
from deepphysinet_amd.optim import FusedClipAdam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 61
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
many = [synth_batch(257 * 145, dev, seed=1000 + b) for b in range(B)]
lead = {k: torch.stack([mb[k].reshape(mb[k].shape[0], -1).squeeze(-1) if k in ('x', 'y', 't', 'f') else mb[k] for mb in many]) for k in ('x', 'y', 't', 'f', 'coord_data')}
lead['field_data'] = torch.cat([mb['field_data'] for mb in many])
lead['forecast_h'] = torch.arange(B, device=dev, dtype=torch.float32).mul_(24.0 / 360.0).view(-1, 1, 1)
lf = m.train_cfg['losses']['loss_factor']
opt = FusedClipAdam(m.physics_net.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=2.5e7)
crit = torch.nn.MSELoss()
for it in range(steps):
    opt.zero_grad(set_to_none=True)
    loss, terms = m.place_lead_batch(lead['x'], lead['y'], lead['t'], lead['f'], lead['field_data'], lead['coord_data'], lead['forecast_h'], crit, lf)
    loss.backward()
    bad = [k for k, p in m.physics_net.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.physics_net.parameters() if p.grad is not None)))
    tt = terms.detach().float()
    tot = tt.sum(dim=1)
    w = int(tot.argmax())
    print('step %d: loss %.6g  grad norm %.4g  non-finite grads %d %s' % (it, float(loss), gn, len(bad), bad[:3]))
    print('   median field total %.4g, worst field %d total %.4g terms %s' % (float(tot.median()), w, float(tot[w]), ['%.3g' % v for v in tt[w].tolist()]))
    opt.step()
    nf = [k for k, p in m.physics_net.named_parameters() if not bool(torch.isfinite(p).all())]
    print('   after the step: non-finite parameters %d %s' % (len(nf), nf[:3]))
