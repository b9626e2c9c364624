#!/usr/bin/env python
"""VERDICT r2 item 5's one measurement: the encoder's forward GEMMs on the block-scaled (MX) fp8 instruction of gfx950
(v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 per instruction, twice the rate of the non-scaled fp8 MFMA) at B = 61 field samples -- time and
encoder-output error against the exact-fp32 product path and the non-scaled fp8 kernel.  Prints one JSON object
(profiles/round3_fp8_mx_encoder.json)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
dev = torch.device('cuda:0')


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
b = synth_batch(257 * 145, dev, seed=1)
leads = 61
many = torch.cat([synth_batch(8, dev, seed=100 + k)['field_data'] for k in range(leads)], dim=0)
fh = torch.arange(leads, device=dev, dtype=torch.float32).mul_(6.0 / 360.0).view(-1, 1, 1)
res = {'note': 'encoder forward (four layers x six GEMMs on 61 x 287 = 17 507 rows, K = 256 / 512) with the GEMMs on: exact-fp32 MFMA (product), '
               'non-scaled fp8 MFMA with per-row scales (DPN_ENCODER_FP8=1), block-scaled MX fp8 MFMA K=64 (DPN_ENCODER_FP8=mx); operands are '
               'quantised from fp32 inside the GEMM kernels in both fp8 forms'}
out = {}
with torch.no_grad():
    for mode in ('0', '1', 'mx'):
        __import__('deepphysinet_amd.config').config.set_switches(encoder_fp8=mode if mode in ('1', 'mx') else '')
        meta1 = m.physics_net.meta_net(b['field_data'], b['forecast_h']).clone()
        losses = m.pde_loss_terms(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h']).cpu().double()
        out[mode] = {'meta': meta1, 'losses': losses,
                     'ms_1_field': timed(lambda: m.physics_net.meta_net(b['field_data'], b['forecast_h']), 20),
                     'ms_61_fields': timed(lambda: m.physics_net.meta_net(many, fh), 5)}
__import__('deepphysinet_amd.config').config.set_switches(encoder_fp8='')
names = {'0': 'fp32_mfma', '1': 'fp8_mfma_per_row_scales', 'mx': 'fp8_mx_mfma_k64'}
for mode in ('0', '1', 'mx'):
    o = out[mode]
    res[names[mode]] = {'encoder_fwd_ms_1_field': o['ms_1_field'], 'encoder_fwd_ms_61_fields': o['ms_61_fields'],
                        'encoder_output_error_max_rel': float((o['meta'] - out['0']['meta']).abs().max() / out['0']['meta'].abs().max()),
                        'pde_loss_rel_error': ((o['losses'] - out['0']['losses']).abs() / out['0']['losses'].abs()).tolist()}
print(json.dumps(res, indent=1))
