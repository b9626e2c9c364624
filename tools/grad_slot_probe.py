"""Which parameters' gradients are NOT produced in their slot of the fused optimiser's flat buffer (each costs a copy kernel per step)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
dev = torch.device('cuda:0')
n = 4096
b = synth_batch(n, dev, seed=1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
opt = m.build_optimizer()
crit = torch.nn.MSELoss()
lf = m.train_cfg['losses']['loss_factor']
opt.zero_grad(set_to_none=True)
loss = m.place_one_batch(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h'], crit, lf, 0, 0, dev)
loss.backward()
names = {id(p): k for k, p in m.named_parameters()}
bad = 0
for p, s in zip(opt.params, opt._slots):
    g = p.grad
    if g is None or g.data_ptr() != s.data_ptr() or not g.is_contiguous():
        bad += 1
        print('%-70s grad %s' % (names.get(id(p), '?'), 'None' if g is None else 'elsewhere, shape %s stride %s' % (tuple(g.shape), g.stride())))
print('%d of %d parameters need a copy' % (bad, len(opt.params)))
