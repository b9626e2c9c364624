#!/usr/bin/env python
"""Phase timeline of the row-local encoder kernels (experiment build: python tools/variant_build.py enctl --unit=5 -DDPN_ENC_TIMELINE).

Lane 0 of every wave stamps the shader clock at the phase boundaries of dpn_enc_fwd_kernel / dpn_enc_bwd_kernel; this runs the encoder of
the bench step forward + backward once and prints, per launch, the mean cycles of every phase over the launch's waves.
usage: enc_timeline.py [variant name, default enctl]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DPN_LIB'] = os.path.join(ROOT, 'tools', '_variants', 'libdpn_hip_%s.so' % (sys.argv[1] if len(sys.argv) > 1 else 'enctl'))
import numpy as np
import torch
from bench import synth_batch
from deepphysinet_amd import _lib as L
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models

FWD = ['start -> input rows split', 'barrier', 'GEMM out-proj', 'staging + barrier', 'row pass: +res, LN1, split', 'barrier', 'GEMM conv1',
       'staging + barrier', 'row pass: GELU, split', 'barrier', 'GEMM conv2', 'staging + barrier', 'row pass: +res, LN2 (LNf), split', 'barrier',
       'GEMM q (proj) + store', 'GEMM k + store', 'GEMM v + store']
BWD = ['start -> head rows split', 'barrier', 'GEMM head (dq Wq + dk Wk + dv Wv | dmeta Wp)', 'staging + barrier', 'row pass: LN2 bwd, split',
       'barrier', 'GEMM conv2^T', 'staging + barrier', 'row pass: gelu grad, split', 'barrier', 'GEMM conv1^T', 'staging + barrier',
       'row pass: LN1 bwd, split', 'barrier', 'GEMM out-proj^T + store']
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
b = synth_batch(257 * 145, dev, seed=1)
lib = L.load()
lib.dpn_enc_debug_set_timeline.argtypes = [ctypes.c_void_p]
records = []
for name in ('dpn_enc_fwd', 'dpn_enc_bwd'):
    orig = getattr(lib, name)

    def wrapped(p, stream, orig=orig, name=name):
        s = p._obj
        nwg = (s.rows + 16 * s.row_tiles - 1) // (16 * s.row_tiles)
        buf = torch.zeros((nwg, 8, 32), dtype=torch.int32, device=dev)
        lib.dpn_enc_debug_set_timeline(ctypes.c_void_p(buf.data_ptr()))
        tag = ('fwd tail=%d next=%d' % (s.tail, s.next)) if name == 'dpn_enc_fwd' else ('bwd head=%d body=%d' % (s.head, s.body))
        records.append((tag, buf))
        if os.environ.get('ENC_TL_WARM') == '1':            # the same launch once before: its weight images are then L2-resident
            orig(p, stream)
        return orig(p, stream)
    setattr(lib, name, wrapped)
for it in range(3):
    records.clear()
    meta = m.physics_net.meta_net(b['field_data'], b['forecast_h'])
    meta.sum().backward()
    torch.cuda.synchronize()
for tag, buf in records:
    t = buf.cpu().numpy().astype('int64') & 0xFFFFFFFF
    names = FWD if tag.startswith('fwd') else BWD
    valid = [i for i in range(32) if (t[:, :, i] != 0).all()]
    first, last = valid[0], valid[-1]
    life = ((t[:, :, last] - t[:, :, first]) & 0xFFFFFFFF)
    print('%s: %d workgroups, wave lifetime mean %.0f max %.0f cycles (stamps %s)' % (tag, t.shape[0], life.mean(), life.max(), valid))
    for i, j in zip(valid[:-1], valid[1:]):
        d = (t[:, :, j] - t[:, :, i]) & 0xFFFFFFFF
        print('   %2d -> %2d  %-52s mean %7.0f  min %7.0f  max %7.0f   per wave %s' % (i, j, names[j - 1] if j - 1 < len(names) else '', d.mean(), d.min(), d.max(),
                                                                                       ' '.join('%5.0f' % x for x in d.mean(axis=0))))
    arr = (t[:, :, valid] - t[:, :1, valid[:1]]) & 0xFFFFFFFF            # arrival of every wave at every stamp, relative to wave 0's start
    print('   arrival at the stamps, workgroup 0, per wave (cycles after wave 0 started):')
    for w in range(8):
        print('      wave %d: %s' % (w, ' '.join('%6d' % x for x in arr[0, w])))
