"""GPU (MI355X): the step machinery around the point path -- flat gradient buffer, fused optimiser as a torch.optim.Optimizer, staged
backward with bucketed gradient all-reduce (two ranks on the one GPU of the test box), training loops, bench.py's N > 1 launch.
Reference: interface/interface_physics.py:334-515 (single-GPU loop and step body), :848-1065 (DDP loop), cfg:151-165 (optimiser, schedule).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dpn_oracle as O
from oracle.fill import fill_state_dict_, synthetic_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEO = O.Geometry()


def _dev():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    return torch.device('cuda:0')


def _model(prec='bf16x2', seed=None):
    from deepphysinet_amd.configs import ncep_config
    from deepphysinet_amd.interface import builder_models
    if seed is not None:
        torch.manual_seed(seed)
    m = builder_models(**ncep_config(), precision=prec)
    if seed is None:
        sd = m.physics_net.state_dict()
        fill_state_dict_(sd)
        m.physics_net.load_state_dict(sd)
    return m.to(_dev())


def _gpu(batch):
    return {k: v.to(_dev()) for k, v in batch.items()}


def _loss(m, g):
    lf = m.train_cfg['losses']['loss_factor']
    return m.place_one_batch(g['x'], g['y'], g['t'], g['f'], g['field_data'], g['coord_data'], g['forecast_h'], torch.nn.MSELoss(), lf, 0, 0, _dev())


def test_gradients_land_in_the_flat_buffer_and_equal_the_plain_backward():
    """With an optimiser registered, backward writes every parameter gradient into the optimiser's flat buffer (p.grad aliases its
    slot, nothing is copied) and the values are bit-for-bit those of a backward pass without arena."""
    inp = synthetic_inputs(300, tag='inter')
    g = _gpu(inp)
    m = _model()
    _loss(m, g).backward()
    plain = {k: p.grad.clone() for k, p in m.physics_net.named_parameters()}
    m.physics_net.zero_grad(set_to_none=True)
    opt = m.build_optimizer()
    opt.zero_grad(set_to_none=True)
    _loss(m, g).backward()
    flat = opt.flat_gradients()
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    outside = [k for k, p in m.physics_net.named_parameters() if not (lo <= p.grad.data_ptr() < hi)]
    assert outside in ([], ['meta_net.model.learnable_token']), outside      # the token gradient is a view of the incoming cotangent
    for k, p in m.physics_net.named_parameters():
        assert torch.equal(p.grad, plain[k]), k
    # a second backward without zero_grad accumulates (the leased slots are not overwritten)
    _loss(m, g).backward()
    for k, p in m.physics_net.named_parameters():
        assert torch.allclose(p.grad, 2 * plain[k], rtol=1e-6, atol=0), k
    # bucket layout: 48 static tensors | 36 head tensors | encoder layers, encoder.norm, projection | data embedding (token, token convolution)
    sizes = [sum(p.numel() for p in b) for b in m.physics_net.gradient_buckets()]
    assert [len(b) for b in m.physics_net.gradient_buckets()] == [48, 36, 68, 3] and sum(sizes) == 5607314
    assert sizes[3] == 128 * 256 + 256 * 2405 * 3 + 256
    assert opt.bucket_bounds[0][0] == 0 and opt.bucket_bounds[-1][1] == flat.numel()
    assert all(a[1] == b[0] for a, b in zip(opt.bucket_bounds, opt.bucket_bounds[1:]))


def test_parameter_shared_by_two_nodes_accumulates():
    """Reference-shaped step without the fused one-pass path: data loss + two place_one_batch calls read the same parameters through
    three point-path nodes; the gradients must add up (only the first node may write the slot)."""
    inp = synthetic_inputs(128, tag='inter')
    g = _gpu(inp)
    m = _model()
    lab = g['labels']
    ref = {}
    for which in range(3):
        m.physics_net.zero_grad(set_to_none=True)
        (m.data_loss(g['x'], g['y'], g['t'], g['field_data'], g['coord_data'], lab, g['forecast_h']) if which == 0 else _loss(m, g)).backward()
        for k, p in m.physics_net.named_parameters():
            ref[k] = ref.get(k, 0) + p.grad.double()
    m.physics_net.zero_grad(set_to_none=True)
    opt = m.build_optimizer()
    opt.zero_grad(set_to_none=True)
    total = m.data_loss(g['x'], g['y'], g['t'], g['field_data'], g['coord_data'], lab, g['forecast_h']) + _loss(m, g) + _loss(m, g)
    total.backward()
    for k, p in m.physics_net.named_parameters():
        d = float((p.grad.double() - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30))
        assert d < 1e-5, (k, d)


def test_fused_optimizer_is_a_torch_optimizer_with_device_side_lr():
    """FusedClipAdam subclasses torch.optim.Optimizer: CosineAnnealingLR attaches (cfg:160-165) and produces the reference's sequence,
    state_dict round-trips into torch.optim.Adam, and a step captured in a hipGraph follows the schedule on replay."""
    from deepphysinet_amd.optim import FusedClipAdam
    dev = _dev()
    torch.manual_seed(3)
    w = [torch.randn(300, 7, device=dev).requires_grad_(True), torch.randn(5000, device=dev).requires_grad_(True)]
    ref_w = [t.detach().clone().requires_grad_(True) for t in w]
    grads = [torch.randn_like(t) for t in w]
    mine = FusedClipAdam(w, lr=1e-4, weight_decay=1e-4, max_norm=1e9)
    assert isinstance(mine, torch.optim.Optimizer)
    mine.param_groups[0]['initial_lr'] = 1e-4
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(mine, T_max=5, eta_min=5e-6, last_epoch=-1)
    ref = torch.optim.Adam(ref_w, lr=1e-4, weight_decay=1e-4)
    ref_sched = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=5, eta_min=5e-6)
    for t, g_ in zip(w, grads):
        t.grad = g_.clone()
    # capture ONE step; every replay must use the learning rate of its epoch
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        mine.step()
    torch.cuda.current_stream().wait_stream(s)
    for t, g_ in zip(ref_w, grads):
        t.grad = g_.clone()
    ref.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        mine.step()
    lrs = []
    for epoch in range(6):
        graph.replay()
        ref.step()
        sched.step()
        ref_sched.step()
        mine.sync_hyper()
        lrs.append(mine.param_groups[0]['lr'])
    assert np.allclose(lrs[:5], [9.09e-5, 6.72e-5, 3.78e-5, 1.41e-5, 5e-6], rtol=5e-3)       # SURVEY appendix A probe sequence
    torch.cuda.synchronize()
    for a, b in zip(w, ref_w):
        assert float((a - b).detach().abs().max()) < 2e-6 * float(b.detach().abs().max()), 'graph replays did not follow the schedule'
    assert int(mine.step_count) == 7                             # one eager step + six replays (the capture itself executes nothing)
    sd = mine.state_dict()
    other = torch.optim.Adam([t.detach().clone().requires_grad_(True) for t in w], lr=1.0)
    other.load_state_dict(sd)                                   # same keys as torch.optim.Adam
    again = FusedClipAdam([t.detach().clone().requires_grad_(True) for t in w], lr=1.0)
    again.load_state_dict(sd)
    assert int(again.step_count) == 7 and abs(again.param_groups[0]['lr'] - mine.param_groups[0]['lr']) < 1e-12
    assert all(torch.equal(a, b) for a, b in zip(again.exp_avg, mine.exp_avg))


def test_staged_step_equals_the_plain_step():
    """StagedPdeStep (backward segments; the layout buckets stage_buckets[i] complete after segment i) gives bit-for-bit the gradients of
    loss.backward().  One field on the fused encoder: TWO segments, the encoder's and the embedding's buckets as one all-reduce (round 6: the
    token convolution's gradient is written by the stack's own weight-gradient launch, the two buckets complete together)."""
    from deepphysinet_amd.interface.interface_physics import StagedPdeStep
    g = _gpu(synthetic_inputs(700, tag='inter'))
    m = _model()
    opt = m.build_optimizer()
    opt.zero_grad(set_to_none=True)
    loss = _loss(m, g)
    loss.backward()
    plain = {k: p.grad.clone() for k, p in m.physics_net.named_parameters()}
    st = StagedPdeStep(m, opt, g)
    assert len(st.stages) == 2 and st.stage_buckets == ((0, 2), (2, 4))
    done = []
    for i, stage in enumerate(st.stages):
        stage()
        for k in range(*st.stage_buckets[i]):
            assert all(p.grad is not None for p in m.physics_net.gradient_buckets()[k])
        done.append(i)
    assert float(st.loss) == float(loss.detach())
    for k, p in m.physics_net.named_parameters():
        assert torch.equal(p.grad, plain[k]), k
    opt.step()
    assert np.isfinite(float(opt.grad_norm))


def test_staged_lead_batch_step_equals_place_lead_batch():
    """the staged step for a batch of field samples (configs[2] sharded over ranks = configs[3] as SURVEY 8d defines it):
    StagedPdeStep(lead_batch=True) gives bit for bit the gradients of place_lead_batch(...).backward(), every layout bucket complete after its
    stage."""
    from deepphysinet_amd.interface.interface_physics import StagedPdeStep
    B, n = 3, 1500
    many = [_gpu(synthetic_inputs(n, tag='lead%d' % k, forecast_h=24.0 * k / 360.0)) for k in range(B)]
    lead = {k: torch.stack([b_[k].reshape(-1) for b_ in many]) for k in ('x', 'y', 't', 'f')}
    lead['coord_data'] = torch.stack([b_['coord_data'] for b_ in many])
    lead['field_data'] = torch.cat([b_['field_data'] * (1.0 + 0.1 * k) for k, b_ in enumerate(many)], dim=0)
    lead['forecast_h'] = torch.cat([b_['forecast_h'] for b_ in many], dim=0)
    m = _model()
    opt = m.build_optimizer()
    opt.zero_grad(set_to_none=True)
    lf = m.train_cfg['losses']['loss_factor']
    loss, _ = m.place_lead_batch(lead['x'], lead['y'], lead['t'], lead['f'], lead['field_data'], lead['coord_data'], lead['forecast_h'],
                                 torch.nn.MSELoss(), lf)
    loss.backward()
    plain = {k: p.grad.clone() for k, p in m.physics_net.named_parameters()}
    st = StagedPdeStep(m, opt, lead, lead_batch=True)
    assert len(st.stages) == 3 and st.stage_buckets == ((0, 2), (2, 3), (3, 4))      # (a batch's embedding backward is a launch of its own)
    for i, stage in enumerate(st.stages):
        stage()
        for k in range(*st.stage_buckets[i]):
            assert all(p.grad is not None for p in m.physics_net.gradient_buckets()[k])
    assert float(st.loss) == float(loss.detach())
    for k, p in m.physics_net.named_parameters():
        assert torch.equal(p.grad, plain[k]), k


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


_RANK_SCRIPT = r'''
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, {root!r})
from deepphysinet_amd import distributed as D
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
from deepphysinet_amd.interface.interface_physics import StagedPdeStep
from oracle.fill import fill_state_dict_, synthetic_inputs
rank, world, _ = D.init_from_env('gloo')          # two ranks share the one GPU of the test box: gloo carries the buckets through the host
dev = torch.device('cuda:0')
m = builder_models(**ncep_config(), precision='bf16x2')
sd = m.physics_net.state_dict(); fill_state_dict_(sd); m.physics_net.load_state_dict(sd)
m = m.to(dev)
D.broadcast_parameters(m.physics_net)
opt = m.build_optimizer()
sync = D.GradientAllReduce(opt)
inp = synthetic_inputs(384, tag='rank%d' % rank, forecast_h=(24.0 + 48.0 * rank) / 360.0)     # each rank: its own field sample + points
g = {{k: v.to(dev) for k, v in inp.items()}}
g['field_data'] = g['field_data'] * (1.0 + 0.25 * rank)
st = StagedPdeStep(m, opt, g)
for i, stage in enumerate(st.stages):
    stage()
    sync.reduce_bucket(*st.stage_buckets[i])
sync.wait()
torch.cuda.synchronize()
out = {{k: p.grad.detach().cpu().numpy() for k, p in m.physics_net.named_parameters()}}
np.savez({out!r} % rank, **out)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


def test_two_ranks_on_one_device_average_the_hip_models_gradients(tmp_path):
    """configs[3] in miniature: two processes (gloo; the test box has one GPU), each running the HIP model's staged step on its OWN field
    sample with the bucket all-reduces of distributed.GradientAllReduce; all 155 averaged gradients must equal the mean of the two
    single-process gradients (what DistributedDataParallel gives the reference, interface_physics.py:903-907, :1056)."""
    script = tmp_path / 'rank.py'
    pattern = str(tmp_path / 'grads_rank%d.npz')
    script.write_text(_RANK_SCRIPT.format(root=ROOT, out=pattern))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got = [np.load(pattern % k) for k in range(2)]
    # single process: the two samples one after the other
    m = _model()
    ref = {}
    for rank in range(2):
        inp = synthetic_inputs(384, tag='rank%d' % rank, forecast_h=(24.0 + 48.0 * rank) / 360.0)
        g = _gpu(inp)
        g['field_data'] = g['field_data'] * (1.0 + 0.25 * rank)
        m.physics_net.zero_grad(set_to_none=True)
        _loss(m, g).backward()
        for k, p in m.physics_net.named_parameters():
            ref[k] = ref.get(k, 0) + 0.5 * p.grad.double().cpu().numpy()
    assert len(ref) == 155
    for k, r_ in ref.items():
        assert np.array_equal(got[0][k], got[1][k]), k                  # both ranks hold the same averaged gradient
        err = np.abs(got[0][k] - r_).max() / (np.abs(r_).max() + 1e-30)
        assert err < 2e-6, (k, err)                                      # fp32 sum of two fp32 gradients


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a torchrun environment launches two ranks itself and reports n_gpus = 2 (both ranks on the
    one GPU of the test box: DPN_BENCH_ONE_DEVICE=1, gloo)."""
    env = dict(os.environ, DPN_BENCH_ONE_DEVICE='1', DPN_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--points', '4096',
                        '--no-cpu-baseline', '--no-alt'], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['config']['step_segments'] == 3
    assert out['value'] > 0 and np.isfinite(out['ms_per_step'])
    assert out['config']['workload'].startswith('configs[3]') and out['collective']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_bench_two_ranks_with_a_lead_batch_run_the_staged_captured_step():
    """`bench.py --gpus 2 --leads 3` (configs[2] per rank, sharded: configs[3] as SURVEY 8d defines it) runs the staged step
    -- three captured backward segments with a bucket all-reduce behind each, then the optimiser -- not an eager step with one collective."""
    env = dict(os.environ, DPN_BENCH_ONE_DEVICE='1', DPN_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--leads', '3', '--steps', '2', '--warmup', '1', '--points', '4096',
                        '--no-cpu-baseline', '--no-alt'], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['config']['leads'] == 3 and out['config']['step_segments'] == 4 and out['config']['hip_graph'] is True
    assert out['config']['workload'].startswith('configs[3] (configs[2] per GPU')
    assert out['value'] > 0 and np.isfinite(out['ms_per_step']) and out['parameters_finite'] is True


def test_bench_eight_ranks_on_one_device_run_the_scaling_benchmarks_code_path():
    """the driver's 8-GPU run must not be the first time `bench.py --gpus 8` executes.  Eight processes on the one GPU of the test
    box (DPN_BENCH_ONE_DEVICE=1, gloo through the host -- RCCL refuses several ranks on one device), a tiny workload: every rank builds the same bucket
    layout (the fingerprint all-gather would raise), runs the same number of steps and all-reduces, the per-rank record covers all eight, and the
    launcher exits cleanly."""
    env = dict(os.environ, DPN_BENCH_ONE_DEVICE='1', DPN_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--points', '512', '--blocks', '2',
                        '--no-prewarm', '--no-cpu-baseline', '--no-alt', '--no-power', '--no-lead-probe'], env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['n_gpus'] == 8 and out['config']['parallelism'] == 'dp8' and out['config']['step_segments'] == 3 and out['config']['hip_graph'] is True
    coll = out['collective']
    assert coll['world'] == 8 and len(coll['devices']) == 8 and sorted(d['rank'] for d in coll['devices']) == list(range(8))
    assert len({d['pid'] for d in coll['devices']}) == 8                       # eight processes
    pr = coll['per_rank']
    assert len(pr['ms_per_step']) == 8 and pr['min'] <= pr['max'] and 0 <= pr['slowest_rank'] < 8
    assert out['value'] > 0 and np.isfinite(out['ms_per_step']) and out['timed_blocks'] == 2


def test_bench_step_with_rccl_collectives_on_one_rank():
    """The N > 1 step shape (three segment graphs, an in-place bucket all-reduce behind each, the optimiser graph) with the REAL backend
    ('nccl' = RCCL) in a one-rank group: the collective calls, ReduceOp.AVG, and their interplay with hipGraph capture / replay and the
    process group's watchdog thread -- everything of `bench.py --gpus N` a one-GPU box can run.  The loss equals the one-graph step's."""
    env = dict(os.environ, DPN_BENCH_RCCL_ONE_RANK='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'DPN_BENCH_BACKEND', 'DPN_BENCH_ONE_DEVICE'):
        env.pop(k, None)
    outs = []
    for e in (env, {k: v for k, v in env.items() if k != 'DPN_BENCH_RCCL_ONE_RANK'}):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2', '--points', '4096',
                            '--no-cpu-baseline', '--no-alt', '--no-prewarm', '--blocks', '1', '--no-lead-probe'], env=e, capture_output=True, text=True,
                           timeout=1200)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]))
    assert outs[0]['config']['step_segments'] == 3 and outs[1]['config']['step_segments'] == 1
    for k, b in outs[1]['pde_losses'].items():                            # after the same number of optimiser steps (--no-prewarm: a fixed count)
        a = outs[0]['pde_losses'][k]
        assert np.isfinite(a) and abs(a - b) <= 1e-5 * abs(b), (k, a, b)
    # roofline durations come from INSIDE the replayed step (device-clock stamps around the launches, point_path.KernelClock) in both step forms,
    # and agree with the event pair around an eager launch of the same kernel to the extent two clock / cache states can
    for o in outs:
        r = o['roofline']
        ins = r['in_step']
        assert ins and 'error' not in ins, ins
        assert ins['replays'] == 40 and ins['clock_khz'] > 0 and 0.0 < ins['stamp_pair_us'] < 20.0, ins
        assert abs(r['kernel_ms'] * 1e3 - ins['fwd_us']) < 1e-6 and r['kernel_ms_source'].startswith('device-clock stamps')
        assert 0.7 < ins['fwd_us'] / (r['kernel_ms_eager_pair'] * 1e3) < 1.3, (ins, r['kernel_ms_eager_pair'])
        assert ins['bwd_us'] > 0 and ins['wgrad_us'] > 0


def test_bench_trial_of_the_one_graph_form_keeps_the_segment_forms_line_when_it_stalls():
    """`bench.py --gpus N` over RCCL tries the one-graph form (all-reduces captured) at the END of the run, with the finished line of the segment form in
    hand.  A stall of the captured collectives -- simulated here: DPN_BENCH_TRIAL_TEST_STALL=replay parks the rank in front of the first replay -- must
    end with the segment form's line, saying where it stalled, and with exit code 14 (bench.TRIAL_STALL_EXIT: NON-zero since round 6 -- a launcher must be able
    to tell a wedged collective from a clean run; the line is still the valid measurement); without the stall the trial reports both forms' times."""
    env = dict(os.environ, DPN_BENCH_RCCL_ONE_RANK='1', DPN_BENCH_TRY_FORMS='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'DPN_BENCH_BACKEND', 'DPN_BENCH_ONE_DEVICE', 'DPN_BENCH_CAPTURE_COLLECTIVES'):
        env.pop(k, None)
    args = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2', '--points', '4096', '--no-cpu-baseline', '--no-alt',
            '--no-prewarm', '--blocks', '2', '--no-lead-probe', '--no-power']
    r = subprocess.run(args, env=dict(env, MASTER_PORT=str(_free_port()), DPN_BENCH_TRIAL_TEST_STALL='replay', DPN_BENCH_TRIAL_WATCHDOG_S='5'),
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 14, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out['config']['step_segments'] == 3 and out['value'] > 0
    assert out['collective']['step_form_trial']['error'].startswith('stalled in: first replays of the one-graph form'), out['collective']['step_form_trial']
    assert 'the one-graph trial stalled' in r.stderr
    r = subprocess.run(args, env=dict(env, MASTER_PORT=str(_free_port())), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    t = out['collective']['step_form_trial']
    assert t['error'] is None and t['segments_ms'] > 0 and t['one_graph_ms'] > 0, t
    assert out['config']['step_segments'] == (1 if t['one_graph_ms'] < 0.99 * t['segments_ms'] else 3)
    if out['config']['step_segments'] == 1:
        assert t['segment_form_result']['ms_per_step'] > 0 and out['config']['collectives_in_graph']


def test_run_train_interface_drives_steps_and_resumes(tmp_path):
    """The reference's train.py calls run_train_interface(checkpoint_path=..., log_path=...) (train.py:47): three steps through the loop
    (data loss only, then PDE losses on), an epoch-end checkpoint + schedule step, and a resume from physics_latest.pth."""
    from deepphysinet_amd.sampler import CollocationSampler, SamplerConfig
    dev = _dev()
    m = _model(seed=5)
    m.train_cfg['num_epoch'] = 2
    m.train_cfg['checkpoints'] = dict(checkpoints_path=str(tmp_path), save_step=1)
    g = torch.Generator().manual_seed(0)
    cube = torch.randn(6, 37, 65, 5, generator=g).to(dev)
    labels = torch.randn(25, 6, 145, 257, generator=g).to(dev)
    smp = CollocationSampler(SamplerConfig(), cube, labels, seed=11)
    field = synthetic_inputs(1)['field_data'].to(dev)
    fh = torch.full((1, 1, 1), 24.0 / 360.0, device=dev)
    samples = lambda epoch: (smp.training_batch(field, fh, n_margin=2048, n_inter=512) for _ in range(2))
    before = {k: v.clone() for k, v in m.physics_net.state_dict().items()}
    out = m.run_train_interface(checkpoint_path=str(tmp_path), log_path=str(tmp_path), samples=samples, pde_start_step=1, device='cuda:0')
    assert out['global_step'] == 4 and os.path.exists(os.path.join(str(tmp_path), 'physics_latest.pth'))
    assert set(out['last']['parts']) == {'margin_loss', 'inter_pde_loss', 'margin_pde_loss'}          # PDE losses were on (step >= 1)
    assert np.isfinite(float(out['last']['loss'])) and abs(out['lr'] - 6.72e-5) < 1e-6                # two scheduler steps from 1e-4
    changed = sum(int(not torch.equal(v, before[k])) for k, v in m.physics_net.state_dict().items())
    assert changed >= 155
    m2 = _model(seed=6)
    m2.train_cfg['num_epoch'] = 3
    m2.train_cfg['checkpoints'] = dict(checkpoints_path=str(tmp_path), save_step=1)
    out2 = m2.run_train_interface(checkpoint_path=str(tmp_path), samples=samples, pde_start_step=1, max_steps=5, device='cuda:0')
    assert out2['global_step'] == 5                                     # resumed at epoch 2, global step 4


def test_encoder_cache_is_refreshed_after_an_optimizer_step():
    """the encode_field cache must not return the previous encoder output once the parameters have moved."""
    g = _gpu(synthetic_inputs(64, tag='inter'))
    m = _model()
    opt = m.build_optimizer(lr=1e-2)
    with torch.no_grad():
        a = m.physics_net.encode_field(g['field_data'], g['forecast_h'], use_cache=True).clone()
    opt.zero_grad()
    _loss(m, g).backward()
    opt.step()
    with torch.no_grad():
        b = m.physics_net.encode_field(g['field_data'], g['forecast_h'], use_cache=True)
    assert not torch.equal(a, b)


def test_point_path_detects_weights_modified_between_forward_and_backward():
    g = _gpu(synthetic_inputs(64, tag='inter'))
    m = _model()
    loss = _loss(m, g)
    with torch.no_grad():
        m.physics_net.U_net.out_fc.weight.mul_(1.5)
    with pytest.raises(RuntimeError, match='modified in place'):
        loss.backward()


def test_fused_optimizer_loads_a_torch_adam_state_dict():
    """the state dicts load BOTH ways.  A torch.optim.Adam state dict has no 'max_norm' (Optimizer.load_state_dict replaces the
    parameter group wholesale); the fused optimiser refills its own extras from its defaults and continues with Adam's moments."""
    m = _model(seed=3)
    g = _gpu(synthetic_inputs(64, tag='inter'))
    ref = torch.optim.Adam(m.physics_net.parameters(), lr=1e-4, weight_decay=1e-4)
    ref.zero_grad()
    _loss(m, g).backward()
    ref.step()
    sd = ref.state_dict()
    opt = m.build_optimizer()
    opt.load_state_dict(sd)
    assert opt.max_norm == 2.5e7 and int(opt.step_count.item()) == 1
    p0 = opt.param_groups[0]['params'][0]
    assert torch.equal(opt.state[p0]['exp_avg'], ref.state[p0]['exp_avg'])
    opt.zero_grad()
    _loss(m, g).backward()
    opt.step()                                              # no KeyError('max_norm'), second Adam step
    assert int(opt.step_count.item()) == 2


def test_fused_optimizer_refuses_reallocated_parameters_and_delayed_backward():
    """the kernels write through raw pointers taken at construction -- a parameter re-allocated afterwards raises instead of
    being updated in freed memory; and a fused step between a forward pass and its backward is detected (the tensors' version counters do
    not move, the optimiser's own step count does)."""
    m = _model(seed=4)
    g = _gpu(synthetic_inputs(64, tag='inter'))
    opt = m.build_optimizer()
    loss = _loss(m, g)
    opt.zero_grad()
    _loss(m, g).backward()
    opt.step()
    with pytest.raises(RuntimeError, match='fused optimiser step ran between'):
        loss.backward()
    p = m.physics_net.U_net.out_fc.weight
    p.data = p.data.clone()
    opt.zero_grad()
    _loss(m, g).backward()
    with pytest.raises(RuntimeError, match='re-allocated'):
        opt.step()


def test_plain_backward_without_an_optimiser_captures_into_a_graph_after_eager_steps_on_another_stream():
    """Round 4 (found by tools/soak.py): the fused encoder used to leave the data embedding's output -- and with it the whole autograd graph of
    the call -- on the module after EVERY forward.  The next forward then reused the parameters' AccumulateGrad nodes of that graph, which were
    pinned to the stream of an earlier iteration; capturing the step on another stream pulled that stream into the capture and the runtime
    died in hipStreamEndCapture.  reference shape: the plain loop of interface_physics.py:1016-1060 (forward, backward, no fused optimiser)."""
    m = _model(seed=11)
    g = _gpu(synthetic_inputs(256, tag='inter'))
    params = list(m.physics_net.parameters())

    def step():
        m.physics_net.zero_grad(set_to_none=True)
        loss = _loss(m, g)
        loss.backward()
        return loss.detach()
    ref_loss = step().clone()
    ref = [p.grad.detach().clone() for p in params]
    assert getattr(m.physics_net.meta_net.model, 'last_embedding', None) is None      # nothing of the step's graph stays on the module
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(); step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = step()
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(static_loss, ref_loss)
        assert all(torch.equal(p.grad, r) for p, r in zip(params, ref))


def test_encoder_cache_is_not_trusted_once_a_fused_step_lives_in_a_graph():
    """replays of a captured optimiser step rewrite the parameters without touching any host-side counter; from the capture on,
    encode_field(use_cache=True) recomputes outside a capture instead of serving a value from before the replays."""
    from deepphysinet_amd import grad_arena
    m = _model(seed=7)
    g = _gpu(synthetic_inputs(64, tag='inter'))
    opt = m.build_optimizer(lr=1e-2)

    def step():
        opt.zero_grad(set_to_none=True)
        _loss(m, g).backward()
        opt.step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(); step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    was = grad_arena.captured_step[0]
    try:
        with torch.cuda.graph(graph):
            step()
        assert grad_arena.captured_step[0]
        with torch.no_grad():
            a = m.physics_net.encode_field(g['field_data'], g['forecast_h'], use_cache=True).clone()
        graph.replay(); graph.replay()
        torch.cuda.synchronize()
        with torch.no_grad():
            b = m.physics_net.encode_field(g['field_data'], g['forecast_h'], use_cache=True)
        assert not torch.equal(a, b)
    finally:
        grad_arena.captured_step[0] = was


_DIST_STEP_SCRIPT = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
import numpy as np, torch
from deepphysinet_amd import distributed as D
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
from deepphysinet_amd.sampler import CollocationSampler, SamplerConfig
from oracle.fill import synthetic_inputs
rank, world, local = D.init_from_env('nccl', force=True)
dev = torch.device('cuda:0')
res = {{}}
for mode in ('plain', 'staged'):
    torch.manual_seed(5)
    m = builder_models(**ncep_config(), precision='bf16x2').to(dev)
    opt = m.build_optimizer()
    gen = torch.Generator().manual_seed(0)
    cube = torch.randn(6, 37, 65, 5, generator=gen).to(dev)
    labels = torch.randn(25, 6, 145, 257, generator=gen).to(dev)
    smp = CollocationSampler(SamplerConfig(), cube, labels, seed=11)
    field = synthetic_inputs(1)['field_data'].to(dev)
    fh = torch.full((1, 1, 1), 24.0 / 360.0, device=dev)
    batch = smp.training_batch(field, fh, n_margin=2048, n_inter=512)
    sync = D.GradientAllReduce(opt, single_rank_too=True) if mode == 'staged' else None
    loss, parts, gnorm = m.training_step(batch, opt, with_pde=True, grad_sync=sync)
    torch.cuda.synchronize()
    res[mode] = (float(loss), float(gnorm), {{k: v.detach().cpu().numpy() for k, v in m.physics_net.state_dict().items()}})
assert res['plain'][0] == res['staged'][0], (res['plain'][0], res['staged'][0])
assert abs(res['plain'][1] - res['staged'][1]) <= 1e-6 * abs(res['plain'][1])
worst = max(float(np.abs(a - res['staged'][2][k]).max()) for k, a in res['plain'][2].items())
print('DIST_STEP_OK worst parameter difference %.3e' % worst)
assert worst <= 1e-7, worst
torch.distributed.destroy_process_group()
'''


def test_data_parallel_training_step_overlaps_buckets_and_matches_the_plain_step(tmp_path):
    """training_step with a gradient reducer cuts its backward at the optimiser's three layout buckets and queues each
    bucket's all-reduce at once (what bench.py's staged step does); with the real backend (RCCL, a one-rank group: AVG over one rank is the
    identity) the step must leave the same parameters as the plain backward + optimiser step."""
    script = tmp_path / 'dist_step.py'
    script.write_text(_DIST_STEP_SCRIPT.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'DIST_STEP_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_train_py_two_keyword_call_runs_end_to_end(tmp_path):
    """SURVEY section 2 row 18 / VERDICT r2 J3: `python train.py --checkpoint_path D --log_path L` (the reference's launcher flags,
    train.py:17-19) builds the interface from the NCEP configuration and calls run_train_interface(checkpoint_path=, log_path=) exactly as
    the reference does (train.py:47); --synthetic opts in to random field samples and on-device collocation batches (without a `samples`
    source the loop raises, test_host_logic_cpu).  Two steps, then a resume."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    ck = str(tmp_path / 'ck')
    for want in ('global_step 2', 'global_step 3'):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '--checkpoint_path', ck, '--log_path', str(tmp_path / 'log'),
                            '--max_steps', want.split()[-1], '--synthetic'], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        assert want in r.stdout, r.stdout[-2000:]
        # the checkpoint is written at epoch end only; a run cut by --max_steps inside epoch 0 writes it when the epoch loop leaves
    assert os.path.exists(os.path.join(ck, 'physics_latest.pth'))


_DIST_LOOP_SCRIPT = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
import numpy as np, torch
from deepphysinet_amd.configs import ncep_config
from deepphysinet_amd.interface import builder_models
from deepphysinet_amd.sampler import CollocationSampler, SamplerConfig
from oracle.fill import synthetic_inputs
rank = int(os.environ['RANK'])
dev = torch.device('cuda:0')
torch.manual_seed(5 + rank)                      # different initial weights per rank: the wrap-time broadcast must align them
m = builder_models(**ncep_config(), precision='bf16x2')
m.train_cfg['num_epoch'] = 1
gen = torch.Generator().manual_seed(0)
cube = torch.randn(6, 37, 65, 5, generator=gen).to(dev)
labels = torch.randn(25, 6, 145, 257, generator=gen).to(dev)
smp = CollocationSampler(SamplerConfig(), cube, labels, seed=11)
field = synthetic_inputs(1)['field_data'].to(dev)
drawn = []

class Samples:                                   # THREE samples for two ranks, in DistributedSampler's seed-0 permutation: padded with its first sample
    def __len__(self): return 3
    def __getitem__(self, i):
        if not 0 <= i < 3: raise IndexError(i)
        drawn.append(i)
        fh = torch.full((1, 1, 1), (24.0 + 24.0 * i) / 360.0, device=dev)
        return smp.training_batch(field * (1.0 + 0.1 * i), fh, n_margin=1024, n_inter=256)
out = m.run_train_interface_dist(samples=Samples(), pde_start_step=0, backend='gloo', device=0)
assert out['global_step'] == 2, out['global_step']
from torch.utils.data.distributed import DistributedSampler
want = list(DistributedSampler(list(range(3)), num_replicas=2, rank=rank))     # the reference's sampler (:936): shuffle=True, seed 0, no set_epoch
assert drawn == want and len(want) == 2, (drawn, want)
np.savez({out!r} % rank, **{{k: v.detach().cpu().numpy() for k, v in m.physics_net.state_dict().items()}})
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


def test_distributed_training_loop_with_an_odd_sample_count(tmp_path):
    """run_train_interface_dist with a sample count that is not a multiple of the world size.  Two processes (gloo; both on
    the one GPU of the test box), three samples: every rank runs TWO steps (DistributedSampler padding: the tail wraps to the epoch's first
    sample), draws only its own samples, and both ranks end with bit-identical parameters (same averaged gradients, same optimiser steps)."""
    script = tmp_path / 'loop.py'
    pattern = str(tmp_path / 'params_rank%d.npz')
    script.write_text(_DIST_LOOP_SCRIPT.format(root=ROOT, out=pattern))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(pattern % 0), np.load(pattern % 1)
    assert len(a.files) == 156
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_plain_bf16_trains_like_the_parity_grade_mode():
    """the plain-bf16 operand mode is quoted as `other_precision_mode` only, and nothing showed that it trains.  Thirty
    optimiser steps from the same initial weights on the same batch in both modes, in the regime the reference trains in for its first 2000
    steps (data loss only, interface_physics.py:436-441; with the PDE terms switched on from step 0 and Adam at 1e-4 the loss of THIS
    initialisation explodes within three steps in either mode -- factors up to 1e14 -- and a trajectory comparison means nothing): the bf16
    loss falls in both modes by a factor > 2, and the mean over the last ten steps agrees within 25 % (measured: 3.63e6 -> 1.14e6 parity-grade,
    -> 1.33e6 plain bf16; step by step the two runs differ by up to 40 % -- Adam at 1e-4 through the hyper-network makes single steps jump by
    factors of two in either mode, so a step-wise bound would only measure that).  Plain bf16 stays out of every headline regardless."""
    g = _gpu(synthetic_inputs(4096, tag='margin', margin=True))
    traj = {}
    for prec in ('bf16x2', 'bf16'):
        m = _model(prec)
        opt = m.build_optimizer()
        losses = []
        for it in range(30):
            opt.zero_grad(set_to_none=True)
            loss = m.data_loss(g['x'], g['y'], g['t'], g['field_data'], g['coord_data'], g['labels'], g['forecast_h'])
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        traj[prec] = np.array(losses)
    a, b = traj['bf16x2'], traj['bf16']
    print('data loss, parity-grade mode:', a[[0, 9, 19, 29]], ' plain bf16:', b[[0, 9, 19, 29]], ' max step-wise deviation %.2e' % np.max(np.abs(b - a) / np.abs(a)))
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
    assert abs(b[0] - a[0]) <= 5e-3 * a[0]                           # same initial weights: the first loss differs by the operand rounding only
    assert a[-10:].mean() < 0.5 * a[0] and b[-10:].mean() < 0.5 * b[0], (a, b)
    assert abs(b[-10:].mean() - a[-10:].mean()) <= 0.25 * a[-10:].mean(), (a[-10:].mean(), b[-10:].mean())
