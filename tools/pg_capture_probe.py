"""Probe: does ProcessGroupNCCL's watchdog thread crash the process when a stream capture that INCLUDES a collective starts while completed eager
collectives are still in its work list?  (profiles/round5_pg_watchdog_capture_race.txt)
usage: pg_capture_probe.py <seconds to wait between the last eager collective and the capture> [seconds spent inside the capture]"""
import os, sys, time
import torch
import torch.distributed as dist
drain = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
inside = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29631')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dist.init_process_group('nccl')
x = torch.ones(1 << 20, device='cuda')
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(6):
        dist.all_reduce(x, async_op=True).wait()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
time.sleep(drain)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    w = dist.all_reduce(x, async_op=True)
    time.sleep(inside)                       # host time inside the capture: the watchdog thread polls its list every ~100 ms
    w.wait()
    x.mul_(0.5)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print('ok: drain %.2f s, %.2f s inside the capture, x[0] = %g' % (drain, inside, float(x[0])), flush=True)
dist.destroy_process_group()
