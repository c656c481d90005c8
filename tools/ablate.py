import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myimagecaptioningmodel_amd import _lib
dev='cuda:0'; bf=torch.bfloat16; code=_lib.BF16
def p(t): return None if t is None else t.data_ptr()
def timeit(fn, iters=20):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): fn(st)
    a,b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn(st)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/iters*1e3
for (h,cin,cout,k) in [(56,64,256,1),(56,256,64,1),(28,128,512,1),(14,256,256,3)]:
    B=64; M=B*h*h; K=k*k*cin
    x=torch.randn((B,h,h,cin),device=dev).to(bf); w=(torch.randn((cout,K),device=dev)/K**.5).to(bf)
    y=torch.zeros((B,h,h,cout),device=dev,dtype=bf)
    g=_lib.ConvGeom(B,h,h,cin,h,h,k,k,1,1,k//2,cin)
    res=[]
    for act in (0,100,102,103):
        f=lambda st: _lib.lib().capmi_igemm_nt(p(x),p(w),p(y),g,cout,K,cout,None,None,0,None,0,None,act,0,0,code,st)
        res.append(timeit(f))
    print('%dx%d %d->%d k%d: full %.1f us | no global stores %.1f | no epilogue %.1f | setup only %.1f' % (h,h,cin,cout,k,*res))
