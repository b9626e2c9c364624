"""Fused row-local encoder nodes and the per-GEMM nodes of rounds 1-3 against torch autograd in fp64 (oracle.meta_net_forward, field by
field) for a batch of B fields (B >= 8 takes the 32-row workgroups): encoder output and all encoder parameter gradients, for a cotangent
whose magnitude varies by `spread` decades from field to field (the PDE step's does: one field's loss can be 100 x the median's).
usage: enc_batch_check.py [B] [spread]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
from oracle import dpn_oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spread = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
field = torch.randn(B, 159, 2405, device=dev)
h = (torch.arange(B, device=dev, dtype=torch.float32).reshape(B, 1, 1) * 24.0) / 360.0
gy = torch.randn(B, 287, 256, device=dev) * (10.0 ** (spread * (torch.rand(B, 1, 1, device=dev) - 0.5)))
res = {}
for mode in ('1', '0'):
    __import__('deepphysinet_amd.config').config.set_switches(encoder_unfused=(mode == '1'))
    m.physics_net.zero_grad(set_to_none=True)
    y = m.physics_net.meta_net(field, h)
    (y * gy).sum().backward()
    torch.cuda.synchronize()
    res[mode] = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.physics_net.meta_net.named_parameters()})
# fp64 reference, field by field
st = {'meta_net.' + k: v.detach().double().cpu().requires_grad_(v.is_floating_point()) for k, v in m.physics_net.meta_net.state_dict().items()}   # the oracle runs on the host
names = [k for k, v in st.items() if v.requires_grad and not k.endswith('position_embedding.pe')]
tot = 0.0
ys = []
for b in range(B):
    yb = O.meta_net_forward(st, field[b:b + 1].double().cpu(), h[b:b + 1].double().cpu())
    ys.append(yb.detach())
    tot = tot + (yb * gy[b:b + 1].double().cpu()).sum()
gref = dict(zip(names, torch.autograd.grad(tot, [st[k] for k in names])))
yref = torch.cat(ys)
for mode, tag in (('1', 'per-GEMM nodes (rounds 1-3)'), ('0', 'row-local fused nodes')):
    y, g = res[mode]
    worst = ('', 0.0)
    for k in g:
        if k.endswith('key_projection.bias'):
            continue
        r_ = gref['meta_net.' + k]
        r = float((g[k].double().cpu() - r_).abs().max() / r_.abs().max().clamp_min(1e-300))
        if r > worst[1]:
            worst = (k, r)
    print('%-30s B = %d spread %.0f decades: output vs fp64 %.3e, worst gradient vs fp64 %.3e (%s)' % (
        tag, B, spread, float((y.double().cpu() - yref).abs().max() / yref.abs().max()), worst[1], worst[0]))
