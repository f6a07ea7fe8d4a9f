import torch, time, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyg_lib_amd import ops
torch.manual_seed(0)
dev='cuda:0'
def err(out, x, ptr, w, bias=None):
    num=0; den=0
    for b in range(w.shape[0]):
        s,e=int(ptr[b]),int(ptr[b+1])
        ref = x[s:e].double() @ w[b].double()
        if bias is not None: ref += bias[b].double()
        num += (out[s:e].double()-ref).pow(2).sum().item(); den += ref.pow(2).sum().item()
    return (num/max(den,1e-300))**0.5
for M in (128, 256):
  for trial in range(3):
    B=13
    sizes=torch.randint(0, 3000, (B,)); sizes[2]=0; sizes[5]=1
    ptr=torch.cat([torch.zeros(1,dtype=torch.long), sizes.cumsum(0)])
    n=int(ptr[-1])
    x=torch.randn(n,128,device=dev) * (1 if trial!=1 else 1000.0)
    if trial==2: x = x.abs()
    w=torch.randn(B,128,M,device=dev)/11
    if trial==2: w = w.abs()
    bias=torch.randn(B,M,device=dev) if trial==0 else None
    for mode in (True, False):
        torch.set_float32_matmul_precision('high' if mode else 'highest')
        out=ops.segment_matmul(x,ptr,w,bias)
        torch.cuda.synchronize()
        print(M, trial, mode, ops.matmul_last_variant(), 'relerr %.3e' % err(out,x,ptr,w,bias), 'maxabs %.3e' % (out.double() - torch.cat([x[int(ptr[b]):int(ptr[b+1])].double()@w[b].double() + (bias[b].double() if bias is not None else 0) for b in range(B)])).abs().max().item())
# grouped with transposed weight views
xs=[torch.randn(r,128,device=dev) for r in (500, 0, 3333)]
ws=[torch.randn(128,128,device=dev).t() for _ in xs]
for mode in (True, False):
    torch.set_float32_matmul_precision('high' if mode else 'highest')
    outs=ops.grouped_matmul(xs,ws)
    print('grouped trans', mode, ops.matmul_last_variant(), [((o.double()-a.double()@b.double()).norm()/max((a.double()@b.double()).norm().item(),1e-30)).item() for o,a,b in zip(outs,xs,ws)])
# timing on C2 fp32
n=1<<22; B=64
x=torch.randn(n,128,device=dev); w=torch.randn(B,128,128,device=dev)/11
ptr=torch.arange(0,n+1,n//B)
for mode in (True, False):
    torch.set_float32_matmul_precision('high' if mode else 'highest')
    for _ in range(3): out=ops.segment_matmul(x,ptr,w)
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): out=ops.segment_matmul(x,ptr,w)
    b.record(); torch.cuda.synchronize()
    ms=a.elapsed_time(b)/20
    print('C2 fp32', mode, ops.matmul_last_variant(), '%.3f ms' % ms, '%.2f TB/s' % (n*1024/ms/1e9), 'relerr %.3e' % err(out[:n//B*2],x,ptr[:3],w[:2]))
