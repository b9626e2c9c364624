"""Two-stream weight-gradient path against the one-stream path: all 155 gradients of one full-grid backward pass (eager), then timing of the
step as one hipGraph in both forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import point_path as PP
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
dev = torch.device('cuda:0')
n = 257 * 145
b = synth_batch(n, dev, seed=1)
crit = torch.nn.MSELoss()
res = {}
for mode in (False, True):
    PP.TWO_STREAM_WGRAD = mode
    torch.manual_seed(1)
    m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
    opt = m.build_optimizer()
    lf = m.train_cfg['losses']['loss_factor']
    for it in range(2):
        opt.zero_grad(set_to_none=True)
        loss = m.place_one_batch(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h'], crit, lf, 0, 0, dev)
        loss.backward()
        torch.cuda.synchronize()
    res[mode] = {k: p.grad.detach().clone() for k, p in m.physics_net.named_parameters()}
worst = 0.0
for k, a in res[False].items():
    d = float((a - res[True][k]).abs().max() / (a.abs().max() + 1e-30))
    worst = max(worst, d)
    if d > 1e-5:
        print('   %-60s rel diff %.3e' % (k, d))
print('worst relative difference over 155 gradients: %.3e' % worst)

# ---- the step as ONE hipGraph (bench.py's shape), both forms: parameters after the same number of steps, and time per replay
out = {}
for mode in (False, True):
    PP.TWO_STREAM_WGRAD = mode
    torch.manual_seed(1)
    m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
    opt = m.build_optimizer(max_norm=2.5e7)
    lf = m.train_cfg['losses']['loss_factor']
    one = torch.ones((), device=dev)

    def whole():
        opt.zero_grad(set_to_none=True)
        loss = m.place_one_batch(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h'], crit, lf, 0, 0, dev)
        loss.backward(one)
        opt.step()
        return loss
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        whole(); whole()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        whole()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    out[mode] = ({k: p.detach().clone() for k, p in m.physics_net.named_parameters()}, e0.elapsed_time(e1) / 50)
    print('two streams %s: %.3f ms per replayed step' % (mode, out[mode][1]))
worst = max(float((a - out[True][0][k]).abs().max() / (a.abs().max() + 1e-30)) for k, a in out[False][0].items())
print('parameters after 56 steps, worst relative difference: %.3e' % worst)

# ---- eager timing of the same step, both forms
for mode in (False, True):
    PP.TWO_STREAM_WGRAD = mode
    torch.manual_seed(1)
    m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
    opt = m.build_optimizer(max_norm=2.5e7)
    lf = m.train_cfg['losses']['loss_factor']
    one = torch.ones((), device=dev)

    def whole():
        opt.zero_grad(set_to_none=True)
        loss = m.place_one_batch(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h'], crit, lf, 0, 0, dev)
        loss.backward(one)
        opt.step()
        return loss
    for _ in range(5):
        whole()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        whole()
    e1.record(); torch.cuda.synchronize()
    print('eager, two streams %s: %.3f ms per step' % (mode, e0.elapsed_time(e1) / 30))
