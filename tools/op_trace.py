"""List the torch (non-library) ops of one eager training step with the autograd node / module call that issued them.
Usage on the GPU box: python tools/op_trace.py [bf16|bf16x2]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    from bench import synth_batch
    from deepphysinet_amd.configs import ncep_config
    from deepphysinet_amd.interface import builder_models
    from deepphysinet_amd.optim import FusedClipAdam
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    m = builder_models(**ncep_config(), precision=prec).to(dev)
    b = synth_batch(257 * 145, dev, seed=1)
    lf = m.train_cfg['losses']['loss_factor']
    opt = FusedClipAdam(m.physics_net.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=2.5e7)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = m.place_one_batch(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h'], torch.nn.MSELoss(), lf, 0, 0, dev)
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    interesting = ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::cat', 'aten::mul', 'aten::gelu',
                   'aten::gelu_backward', 'aten::roll', 'aten::sin', 'aten::cos', 'aten::sum', 'aten::div', 'aten::_foreach')
    counts = collections.Counter()
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CPU:
            continue
        if not any(e.name == n or e.name.startswith(n) for n in interesting):
            continue
        if not e.kernels:
            continue
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 6:
            chain.append(p.name)
            p = p.cpu_parent
        shapes = ''
        counts[(e.name, ' <- '.join(chain))] += 1
    for (name, chain), c in sorted(counts.items(), key=lambda kv: -kv[1]):
        print('%3d  %-22s %s' % (c, name, chain))
    print('total torch-op launches listed:', sum(counts.values()))


if __name__ == '__main__':
    main()
