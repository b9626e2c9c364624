import torch, time
dev='cuda'
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
x=torch.randn(287,256,device=dev); w=torch.randn(256,256,device=dev); b=torch.randn(256,device=dev)
m=torch.randn(256,256,device=dev); wh=torch.randn(193,256,device=dev); bh=torch.randn(193,device=dev)
wc=torch.randn(2700,256,device=dev); bc=torch.randn(2700,device=dev)
for lib in ('default','hipblaslt','rocblas'):
    if lib!='default':
        try: torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e: print('cannot set',lib,e); continue
    print(lib, 'linear 287x256x256: %.1f us'%bench(lambda: torch.nn.functional.linear(x,w,b)),
          ' head 256x256->193: %.1f us'%bench(lambda: torch.nn.functional.linear(m,wh,bh)),
          ' heads cat 256x256->2700: %.1f us'%bench(lambda: torch.nn.functional.linear(m,wc,bc)),
          ' matmul no bias: %.1f us'%bench(lambda: x@w))
xb=x.bfloat16(); wb=w.bfloat16()
print('bf16 linear: %.1f us'%bench(lambda: torch.nn.functional.linear(xb,wb)))
print('empty kernel-ish add: %.1f us'%bench(lambda: x.add_(1.0)))
