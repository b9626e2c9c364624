import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepphysinet_amd.linear import linear
dev='cuda'
x=torch.randn(287,256,device=dev); ws=[torch.randn(256,256,device=dev) for _ in range(50)]; b=torch.randn(256,device=dev)
def graph_time(fn, reps=50):
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
def chain_lin():
    y=x
    for w in ws: y=linear(y,w,b)*0.01      # dependent chain of 50 GEMMs (+50 muls), cold weights each
    return y
def chain_mul():
    y=x
    for w in ws: y=y*0.01
    return y
def chain_torch():
    y=x
    for w in ws: y=torch.nn.functional.linear(y,w,b)*0.01
    return y
with torch.no_grad():
    tl=graph_time(chain_lin); tm=graph_time(chain_mul); tt=graph_time(chain_torch)
print('50 x (dpn linear + mul): %.1f us -> %.2f us per pair; 50 x mul: %.1f us -> %.2f us each; => dpn linear ~ %.2f us; torch linear ~ %.2f us'%(tl,tl/50,tm,tm/50,(tl-tm)/50,(tt-tm)/50))
