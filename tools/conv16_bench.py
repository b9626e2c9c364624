"""dpn_conv16 (the token convolution on pre-split f16 planes) in isolation: time against the number of K-slices, result against fp64.
usage: conv16_bench.py [batch]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L, encoder_ops as E
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
net = m.physics_net.meta_net.model
b = synth_batch(1024, dev, seed=1)
field = b['field_data'].repeat(B, 1, 1) * (1.0 + 0.1 * torch.arange(B, device=dev).view(B, 1, 1))
h = b['forecast_h'].repeat(B, 1, 1)
layers = list(net.encoder.attn_layers)
lib = L.load_experiments()
with torch.no_grad():
    __import__('deepphysinet_amd.config').config.set_switches(conv16=True)
    prep = E.encoder_prep(field, h, net.enc_embedding, None, layers, net.encoder.norm, net.projection)
    xs, xe, ws, we, Kp, cw = prep.conv16
    T, D = field.shape[1], cw.shape[0]
    ref = (prep.xu.double() @ cw.view(D, -1).double().t())
    for parts in (15, 16, 24, 32, 57):
        out = torch.zeros((parts, B * T, D), dtype=torch.float32, device=dev)
        call = lambda: L.check(lib.dpn_conv16(E._p(xs), E._p(xe), E._p(ws), E._p(we), B * T, D, Kp, parts, E._p(out), E._s()), 'dpn_conv16')
        try:
            call()
        except RuntimeError as e:
            print('slices %3d: refused (%s)' % (parts, str(e)[:60]))
            continue
        torch.cuda.synchronize()
        err = float((out.double().sum(0) - ref).abs().max() / ref.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5): call()
        e0.record()
        for _ in range(50): call()
        e1.record(); torch.cuda.synchronize()
        print('B = %d  slices %3d (%2d blocks each): %6.1f us per launch (back to back), max err / max |ref| %.2e' % (B, parts, -(-(Kp // 32) // parts), e0.elapsed_time(e1) * 20, err))
