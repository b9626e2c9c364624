"""The data embedding's forward (token convolution as a split-K exact-fp32 MFMA GEMM + assemble) captured in a hipGraph and replayed: microseconds per replay for the
number of K-slices in DPN_EMBED_PARTS (run once per value: the switch is frozen at import).  usage: DPN_EMBED_PARTS=N python tools/embed_parts_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd import encoder_ops as EO, config
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
b = synth_batch(4096, dev, seed=1)
net = m.physics_net.meta_net.model
with torch.no_grad():
    layers = list(net.encoder.attn_layers)
    prep = EO.encoder_prep(b['field_data'], b['forecast_h'], net.enc_embedding, None, layers, net.encoder.norm, net.projection)
    def op():
        return EO.data_embedding_fused(b['field_data'], net.enc_embedding, net.learnable_token, b['forecast_h'], prep=prep)
    ref = op()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): op()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(10): out = op()
    ts = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 200)
    ts.sort()
    print('DPN_EMBED_PARTS=%s: GEMM + assemble %.2f us per forward (median of 7 x 200; min %.2f)   max|out - first| %.1e' % (
        os.environ.get('DPN_EMBED_PARTS', '16 (default)'), ts[3], ts[0], float((out - ref).abs().max())))
