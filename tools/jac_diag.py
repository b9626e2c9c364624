import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, numpy as np
import test_gpu_parity as T
from oracle.fill import synthetic_inputs
import deepphysinet_amd as dpn
for n in (1037, 256):
    inp = synthetic_inputs(n, tag='inter')
    ref = T._oracle(inp, want_grads=False)
    for prec in ('bf16x2', 'bf16'):
        m = T._model(prec); g = T._gpu(inp); cfg = m.point_config()
        with torch.no_grad():
            heads, evec, statics = m.physics_net.field_weights(g['field_data'], g['forecast_h'])
            out_n, jac_n = dpn.pde_fields_and_jacobian(cfg, g['x'], g['y'], g['t'], g['coord_data'], heads, evec, statics)
        for k in range(6):
            r = ref['jac_n'][:, k]; e = (jac_n.cpu()[:, k] - r).abs()
            col = r.abs().amax(0)
            rel = e / col
            idx = int(rel.amax(1).argmax())
            print(n, prec, 'net', k, 'max/globalmax %.2e' % float(e.max() / r.abs().max()), 'relL2 %.2e' % float(e.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()),
                  'percol max', ['%.1e' % v for v in rel.amax(0).tolist()], 'worst pt', idx, 'n>tol', int((e > 2e-4 * r.abs().max()).sum()))
