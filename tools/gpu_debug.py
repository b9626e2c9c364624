import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import deepphysinet_amd as dpn
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
from oracle.fill import fill_state_dict_
from bench import synth_batch
dev = torch.device('cuda:0')
for prec in ('bf16x2', 'bf16'):
    m = builder_models(**ncep_config(), precision=prec)
    sd = m.physics_net.state_dict(); fill_state_dict_(sd); m.physics_net.load_state_dict(sd); m = m.to(dev)
    n = 257 * 145
    b = synth_batch(n, dev, seed=3)
    cfg = m.point_config()
    with torch.no_grad():
        heads, evec, statics = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
        def run(sel):
            return dpn.pde_fields_and_jacobian(cfg, b['x'][sel], b['y'][sel], b['t'][sel], b['coord_data'][sel], heads, evec, statics)
        o1, j1 = run(slice(None)); o2, j2 = run(slice(None))
        print(prec, 'repeat determinism: fields', float((o1 - o2).abs().max()), 'jac', float((j1 - j2).abs().max()))
        o3, j3 = run(slice(1, n))
        print('   shift-by-1: fields max abs diff', float((o1[1:] - o3).abs().max()), 'rel', float((o1[1:] - o3).abs().max() / o1.abs().max()),
              ' jac rel', float((j1[1:] - j3).abs().max() / j1.abs().max()))
        o4, j4 = run(slice(0, 20001))
        print('   prefix: fields diff', float((o1[:20001] - o4).abs().max()), 'jac rel', float((j1[:20001] - j4).abs().max() / j1.abs().max()))
        t1 = m.pde_loss_terms(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h']).cpu().numpy()
        t2 = m.pde_loss_terms(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h']).cpu().numpy()
        print('   loss repeat', t1, t2)
