"""Race hunt: the same full-grid step (fixed parameters, fixed inputs) repeated N times, eager and as a replayed hipGraph; the loss and
every parameter gradient must be BITWISE equal to the first run every time.  The point kernels synchronise their LDS rings by hand
(counted vmcnt / lgkmcnt waits, raw s_barrier): a missing wait shows up here as an occasional differing bit, not as a crash.
usage: soak.py [bf16|bf16x2] [iterations=300]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device('cuda:0')
torch.manual_seed(3)
m = builder_models(**ncep_config(), precision=prec).to(dev)
b = synth_batch(257 * 145, dev, seed=3)
lf = m.train_cfg['losses']['loss_factor']
crit = torch.nn.MSELoss()
params = list(m.physics_net.parameters())


def step():
    m.physics_net.zero_grad(set_to_none=True)
    loss = m.place_one_batch(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h'], crit, lf, 0, 0, dev)
    loss.backward()
    return loss.detach()          # never keep the autograd graph alive across iterations (its AccumulateGrad nodes pin a stream)


ref_loss = step().detach().clone()
ref = [p.grad.detach().clone() for p in params]
bad = 0
for i in range(iters):                                      # eager
    loss = step()
    if not torch.equal(loss.detach(), ref_loss) or any(not torch.equal(p.grad, r) for p, r in zip(params, ref)):
        bad += 1
print('%s eager : %d / %d runs differ' % (prec, bad, iters), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
gbad = 0
for i in range(iters):
    g.replay()
    torch.cuda.synchronize()
    if not torch.equal(static_loss.detach(), ref_loss) or any(not torch.equal(p.grad, r) for p, r in zip(params, ref)):
        gbad += 1
print('%s graph : %d / %d replays differ' % (prec, gbad, iters), flush=True)
sys.exit(1 if bad or gbad else 0)
