import torch, time
dev = torch.device('cuda:0')
a = torch.randn(4096, 4096, device=dev)
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True); e1 = torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print('no external kw:', ex); raise SystemExit
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    b = a @ a
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    c = a * 2.0
    e0.record()
    b = a @ a
    e1.record()
    d = b + 1.0
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print('captured-event elapsed ms:', e0.elapsed_time(e1))
x0 = torch.cuda.Event(enable_timing=True); x1 = torch.cuda.Event(enable_timing=True)
x0.record(); b = a @ a; x1.record(); torch.cuda.synchronize(); print('eager matmul ms:', x0.elapsed_time(x1))
