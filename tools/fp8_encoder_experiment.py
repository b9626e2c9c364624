#!/usr/bin/env python
"""BASELINE configs[4] as a measured experiment: the encoder layers' forward GEMMs on the fp8 matrix cores (DPN_ENCODER_FP8=1,
csrc/dpn_fp8.hip) against the product's exact-fp32 MFMA GEMMs -- parity error and time, one field (configs[1]) and 61 fields
(configs[2]).  Prints one JSON object (committed as profiles/round2_fp8_encoder_experiment.json).

    python tools/fp8_encoder_experiment.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import synth_batch
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models

dev = torch.device('cuda:0')
n = 257 * 145


def timed(fn, reps=20):
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


def run(fp8, prec, leads):
    __import__('deepphysinet_amd.config').config.set_switches(encoder_fp8='1' if fp8 else '')
    torch.manual_seed(1)
    m = builder_models(**ncep_config(), precision=prec).to(dev)
    b = synth_batch(n, dev, seed=1)
    out = {}
    with torch.no_grad():
        meta = m.physics_net.meta_net(b['field_data'], b['forecast_h'])
        out['meta_out'] = meta.detach().clone()
        out['losses'] = m.pde_loss_terms(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h']).cpu().double().numpy()
        out['encoder_fwd_ms_1_field'] = timed(lambda: m.physics_net.meta_net(b['field_data'], b['forecast_h']))
        many = torch.cat([synth_batch(8, dev, seed=100 + k)['field_data'] for k in range(leads)], dim=0)
        fh = torch.arange(leads, device=dev, dtype=torch.float32).mul_(6.0 / 360.0).view(-1, 1, 1)
        out['encoder_fwd_ms_%d_fields' % leads] = timed(lambda: m.physics_net.meta_net(many, fh), reps=5)
    opt = m.build_optimizer()
    lf = m.train_cfg['losses']['loss_factor']
    crit = torch.nn.MSELoss()

    def step():
        opt.zero_grad(set_to_none=True)
        m.place_one_batch(b['x'], b['y'], b['t'], b['f'], b['field_data'], b['coord_data'], b['forecast_h'], crit, lf, 0, 0, dev).backward()
        opt.step()
    out['step_ms_configs1_eager'] = timed(step, reps=20)
    return out


res = {'note': 'encoder forward GEMMs (q/k/v/out projections, conv1, conv2 of the four layers): exact-fp32 MFMA (product) vs fp8 e4m3 MFMA with '
               'per-row scales (experiment, DPN_ENCODER_FP8=1); the heads, the point kernels and every backward GEMM are unchanged',
       'points': n}
for prec in ('bf16x2',):
    ref = run(False, prec, 61)
    f8 = run(True, prec, 61)
    rel_meta = float((f8['meta_out'] - ref['meta_out']).abs().max() / ref['meta_out'].abs().max())
    rel_l2 = float((f8['meta_out'] - ref['meta_out']).pow(2).mean().sqrt() / ref['meta_out'].pow(2).mean().sqrt())
    res[prec] = {
        'encoder_output_error_max_rel': rel_meta, 'encoder_output_error_l2_rel': rel_l2,
        'pde_losses_fp32_encoder': ref['losses'].tolist(), 'pde_losses_fp8_encoder': f8['losses'].tolist(),
        'pde_loss_rel_error': (abs(f8['losses'] - ref['losses']) / abs(ref['losses'])).tolist(),
        'encoder_fwd_ms_1_field': {'fp32_mfma': ref['encoder_fwd_ms_1_field'], 'fp8_mfma': f8['encoder_fwd_ms_1_field']},
        'encoder_fwd_ms_61_fields': {'fp32_mfma': ref['encoder_fwd_ms_61_fields'], 'fp8_mfma': f8['encoder_fwd_ms_61_fields']},
        'step_ms_configs1_eager': {'fp32_mfma': ref['step_ms_configs1_eager'], 'fp8_mfma': f8['step_ms_configs1_eager']},
    }
__import__('deepphysinet_amd.config').config.set_switches(encoder_fp8='')
print(json.dumps(res, indent=1))
