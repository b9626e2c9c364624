"""configs[2] sanity: the hyper-network heads of B = 61 fields from the one-launch path against the per-field launches (values and all
parameter gradients of a full lead-batch step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
dev = torch.device('cuda:0')
B, n = 61, 4096
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
bs = [synth_batch(n, dev, seed=100 + k) for k in range(B)]
field = torch.cat([b['field_data'] for b in bs], dim=0)
fh = torch.arange(B, device=dev, dtype=torch.float32).mul_(6.0 / 360.0).view(-1, 1, 1)
x, y, t, f = (torch.stack([b[k].reshape(-1) for b in bs]) for k in ('x', 'y', 't', 'f'))
cd = torch.stack([b['coord_data'] for b in bs])
crit = torch.nn.MSELoss()
lf = m.train_cfg['losses']['loss_factor']
res = {}
for mode in ('1', '0'):
    __import__('deepphysinet_amd.config').config.set_switches(heads_per_field=mode)
    m.physics_net.zero_grad(set_to_none=True)
    with torch.no_grad():
        hw = m.physics_net.field_weights(field, fh)
    loss, terms = m.place_lead_batch(x, y, t, f, field, cd, fh, crit, lf)
    loss.backward()
    res[mode] = (hw[0].clone(), hw[1].clone(), float(loss.detach()), terms.detach().clone(), {k: p.grad.detach().clone() for k, p in m.physics_net.named_parameters()})
a, b = res['1'], res['0']
rel = lambda u, v: float((u - v).abs().max() / u.abs().max())
print('heads rel diff %.3e  evec rel diff %.3e  |heads| max %.3e  nonzero fraction %.4f' % (rel(a[0], b[0]), rel(a[1], b[1]), float(b[0].abs().max()), float((b[0] != 0).float().mean())))
print('loss per-field path %.9g  one-launch path %.9g' % (a[2], b[2]))
print('terms rel diff %.3e' % rel(a[3], b[3]))
worst = max((rel(a[4][k], b[4][k]), k) for k in a[4] if not k.endswith('key_projection.bias'))
print('worst gradient rel diff %.3e (%s)' % worst)
