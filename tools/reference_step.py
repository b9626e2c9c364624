"""The reference-shaped training step (interface_physics.py:443-515): data loss on 20 480 margin points, PDE losses on 4 096 interior +
20 480 margin points, backward, clip, Adam -- the batch is drawn on the device by the sampler INSIDE the captured step (Philox counter
advanced by the optimiser's device-side step counter: fresh points on every replay), the whole thing one hipGraph.  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_batch
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
from deepphysinet_amd.optim import FusedClipAdam
from deepphysinet_amd.sampler import CollocationSampler, SamplerConfig

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision=prec).to(dev)
opt = FusedClipAdam(m.physics_net.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=2.5e7)
g = np.random.default_rng(0)
smp = CollocationSampler(SamplerConfig(), torch.from_numpy(g.standard_normal((6, 37, 65, 5)).astype(np.float32)).to(dev),
                         torch.from_numpy(g.standard_normal((25, 6, 145, 257)).astype(np.float32)).to(dev), seed=1)
b0 = synth_batch(8, dev, seed=1)
smp.bind_step_counter(opt.step_count, 20480 + 4096)
drawn = {}


def step():
    batch = smp.training_batch(b0['field_data'], b0['forecast_h'])       # 2 sampler launches, captured with the step
    drawn['inter_x'] = batch['inter_x']
    m.training_step(batch, opt, with_pde=True)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
first = None
for i in range(10):
    graph.replay()
    if i == 0:
        torch.cuda.synchronize()
        first = drawn['inter_x'].clone()
torch.cuda.synchronize()
assert not torch.equal(first, drawn['inter_x']), 'the captured sampler must draw fresh points on every replay'
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    graph.replay()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 200
print(json.dumps({'workload': 'reference-shaped step: data loss 20480 pts + PDE 4096 interior + PDE 20480 margin, bwd, clip, Adam', 'precision': prec, 'sampler': 'inside the captured step (device-side Philox offset)',
                  'ms_per_step': ms, 'pde_points_per_s': 24576 / ms * 1e3}))
