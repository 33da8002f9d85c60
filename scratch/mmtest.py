import torch, time
d='cuda'
a=torch.randn(10080,1600,device=d,dtype=torch.bfloat16); b=torch.randn(10080,6400,device=d,dtype=torch.bfloat16)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
try:
    r=torch.mm(b.t(),a,out_dtype=torch.float32); print('out_dtype ok',r.dtype,r.shape)
    ref=(b.t().float()@a.float()); print('err',(r-ref).abs().max().item()/ref.abs().max().item())
    print('us f32out',t(lambda: torch.mm(b.t(),a,out_dtype=torch.float32)), 'bf16out', t(lambda: torch.mm(b.t(),a)))
    o=torch.empty(6400,1600,device=d)
    try:
        torch.mm(b.t(),a,out_dtype=torch.float32,out=o); print('out= ok', (o-ref).abs().max().item())
        print('us out=',t(lambda: torch.mm(b.t(),a,out_dtype=torch.float32,out=o)))
    except Exception as ex: print('out= fail',ex)
except Exception as ex: print('out_dtype fail',ex)
# addmm with bias
w=torch.randn(6400,1600,device=d,dtype=torch.bfloat16); bias=torch.randn(6400,device=d,dtype=torch.bfloat16)
print('addmm us',t(lambda: torch.addmm(bias,a,w.t())),'mm us',t(lambda: torch.mm(a,w.t())))
# addmm with residual
x=torch.randn(10080,1600,device=d,dtype=torch.bfloat16); w2=torch.randn(1600,6400,device=d,dtype=torch.bfloat16)
print('addmm res us',t(lambda: torch.addmm(x,b,w2.t())),'mm us',t(lambda: torch.mm(b,w2.t())))
