#!/usr/bin/env python
"""Two measurements behind the "weight-gradient products of the static tensors under the encoder backward" design:

  (i)  dpn_wgrad_kernel time against the number of point splits (workgroups = 24 x splits), experiment build -DDPN_EXPERIMENT_SPLITS;
  (ii) the encoder + heads backward chain (one hipGraph) alone, dpn_wgrad alone, and both at once on two streams.

    python tools/timeline_probe.py --build -DDPN_EXPERIMENT_SPLITS     # here
    python tools/wgrad_overlap_probe.py [bf16|bf16x2]                  # on the GPU box
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DPN_LIB'] = os.path.join(ROOT, 'tools', '_variants', 'libdpn_hip_timeline.so')


def main():
    import torch
    from bench import synth_batch
    from deepphysinet_amd import _lib as L
    from deepphysinet_amd import point_path as PP
    from deepphysinet_amd.configs import ncep_config
    from deepphysinet_amd.interface import builder_models
    prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x2'
    sys.argv = [a for a in sys.argv if a != '--overlap'] + (['--overlap'] if '--overlap' in sys.argv else [])
    n = 257 * 145
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    m = builder_models(**ncep_config(), precision=prec).to(dev)
    b = synth_batch(n, dev, seed=1)
    lib = L.load()
    cfg = m.point_config()

    def timed(fn, reps=60):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    with torch.no_grad():
        heads, evec, statics = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
    x_, y_, t_ = (PP._f32c(b[k]).reshape(-1) for k in ('x', 'y', 't'))
    cd_ = PP._f32c(b['coord_data'])
    st = [PP._f32c(s) for s in statics]
    nets = PP._net_ptrs(PP._f32c(heads), PP._f32c(evec), st)
    g_out = torch.randn(n, 6, device=dev) * 1e-3
    g_jxi = torch.randn(n, 6, 3, device=dev) * 1e-3
    geo = cfg.geometry()
    print('(i) dpn_wgrad time against the number of splits, %s' % prec)
    plans = ['10,10,10,10', '9,12,10,11', '10,12,10,10', '10,11,10,11', '11,11,10,10', '9,12,11,10', '8,12,11,11', '10,11,11,10']
    if len(sys.argv) > 2:
        plans = [a for a in sys.argv[2:] if a != '--overlap'] or plans
    for plan in plans:
        os.environ['DPN_WGRAD_PLAN'] = plan
        splits = sum(int(v) for v in plan.split(',')) / 4.0
        ws = PP._Workspace(n, cfg.prec, dev)
        PP._forward_points(cfg, ws, nets, x_, y_, t_, None, cd_, True, True)
        operands = torch.empty(ws.sizes.operands, dtype=torch.uint8, device=dev)
        partials = torch.empty(ws.sizes.partials, dtype=torch.uint8, device=dev)
        L.check(lib.dpn_bwd_points(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(PP._freqs(dev)), ctypes.byref(geo),
                                   PP._ptr(ws.packed), cfg.prec, PP._ptr(g_out), PP._ptr(g_jxi), PP._ptr(ws.saved), PP._ptr(operands), PP._stream()), 'bwd')
        us = timed(lambda: L.check(lib.dpn_wgrad(n, cfg.prec, PP._ptr(g_out), PP._ptr(ws.saved), PP._ptr(operands), PP._ptr(partials), PP._stream()), 'wgrad'))
        print('   plan %-12s workgroups %3d  %7.1f us' % (plan, int(24 * splits), us))
        del ws, operands, partials
    os.environ.pop('DPN_WGRAD_PLAN')
    if '--overlap' not in sys.argv:
        return
    if os.environ.get('DPN_OVERLAP_PLAN'):            # round 3: leave CUs free for the chain (the round-2 probe ran wgrad on all 252)
        os.environ['DPN_WGRAD_PLAN'] = os.environ['DPN_OVERLAP_PLAN']
        print('overlap section with wgrad plan', os.environ['DPN_WGRAD_PLAN'])

    # (ii) the encoder + heads chain (forward + backward) as one graph; wgrad as another; alone and together
    ws = PP._Workspace(n, cfg.prec, dev)
    PP._forward_points(cfg, ws, nets, x_, y_, t_, None, cd_, True, True)
    operands = torch.empty(ws.sizes.operands, dtype=torch.uint8, device=dev)
    partials = torch.empty(ws.sizes.partials, dtype=torch.uint8, device=dev)
    L.check(lib.dpn_bwd_points(PP._ptr(x_), PP._ptr(y_), PP._ptr(t_), None, PP._ptr(cd_), n, PP._ptr(PP._freqs(dev)), ctypes.byref(geo),
                               PP._ptr(ws.packed), cfg.prec, PP._ptr(g_out), PP._ptr(g_jxi), PP._ptr(ws.saved), PP._ptr(operands), PP._stream()), 'bwd')
    gh, ge = torch.randn_like(heads), torch.randn_like(evec)
    params = [p for p in m.physics_net.parameters() if p.requires_grad]

    def enc_fwd():
        h, e, _ = m.physics_net.field_weights(b['field_data'], b['forecast_h'])
        return h, e

    def enc_both():
        h, e = enc_fwd()
        torch.autograd.grad([h, e], params, [gh, ge], allow_unused=True)

    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    graphs = {}
    for name, fn, s in (('enc_fwd', lambda: enc_fwd(), sA), ('enc_fwd_bwd', enc_both, sA),
                        ('wgrad', lambda: L.check(lib.dpn_wgrad(n, cfg.prec, PP._ptr(g_out), PP._ptr(ws.saved), PP._ptr(operands), PP._ptr(partials),
                                                                  torch.cuda.current_stream().cuda_stream), 'wgrad'), sB)):
        with torch.cuda.stream(s):
            for _ in range(2):
                fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        graphs[name] = g
    t_f = timed(graphs['enc_fwd'].replay)
    t_fb = timed(graphs['enc_fwd_bwd'].replay)
    t_w = timed(graphs['wgrad'].replay)
    print('(ii) alone: encoder forward %.1f us, forward + backward %.1f us (backward chain ~ %.1f), wgrad %.1f us' % (t_f, t_fb, t_fb - t_f, t_w))

    def together():
        sA.wait_stream(torch.cuda.current_stream())
        sB.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sB):
            graphs['wgrad'].replay()
        with torch.cuda.stream(sA):
            graphs['enc_fwd_bwd'].replay()
        torch.cuda.current_stream().wait_stream(sA)
        torch.cuda.current_stream().wait_stream(sB)
    t_both = timed(together)
    print('     both at once on two streams: %.1f us  (sum %.1f, longer one %.1f)' % (t_both, t_fb + t_w, max(t_fb, t_w)))

    # one graph, two branches (fork / join inside the capture)
    def forked():
        sB.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sB):
            L.check(lib.dpn_wgrad(n, cfg.prec, PP._ptr(g_out), PP._ptr(ws.saved), PP._ptr(operands), PP._ptr(partials), sB.cuda_stream), 'wgrad')
        enc_both()
        torch.cuda.current_stream().wait_stream(sB)
    with torch.cuda.stream(sA):
        for _ in range(2):
            forked()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=sA):
        forked()
    print('     one graph with the two as parallel branches: %.1f us' % timed(g.replay))
    # and the backward chain only beside wgrad: what the step would do
    with torch.cuda.stream(sA):                       # the backward pass runs on the stream of its forward
        h, e = enc_fwd()
    torch.cuda.synchronize()

    def forked_bwd():
        sB.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sB):
            L.check(lib.dpn_wgrad(n, cfg.prec, PP._ptr(g_out), PP._ptr(ws.saved), PP._ptr(operands), PP._ptr(partials), sB.cuda_stream), 'wgrad')
        torch.autograd.grad([h, e], params, [gh, ge], allow_unused=True, retain_graph=True)
        torch.cuda.current_stream().wait_stream(sB)
    with torch.cuda.stream(sA):
        for _ in range(2):
            forked_bwd()
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=sA):
        forked_bwd()
    print('     one graph: encoder backward chain beside wgrad: %.1f us  (alone: ~%.1f and %.1f)' % (timed(g2.replay), t_fb - t_f, t_w))


if __name__ == '__main__':
    main()
