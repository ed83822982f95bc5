import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
CFGS = [(144,32,32,64,0,64,3,1,1,1,True), (144,16,16,128,0,128,3,1,1,1,True), (144,64,64,64,0,64,3,1,1,1,False), (144,8,8,256,0,256,3,1,1,1,True), (144,128,128,32,0,32,3,1,1,1,False)]
if os.environ.get('ONLY'): CFGS=[CFGS[int(os.environ['ONLY'])]]
for cfg in CFGS:
    M,H,W,C1,C2,Cout,k,s,pad,act,use_res = cfg
    x1 = torch.randn(M,H,W,C1,device='cuda'); w = torch.randn(Cout,(C1+C2)*k*k,device='cuda')/np.sqrt((C1+C2)*k*k); b = torch.randn(Cout,device='cuda')
    Ho,Wo = (H+2*pad-k)//s+1,(W+2*pad-k)//s+1
    res = torch.randn(M,Ho,Wo,Cout,device='cuda') if use_res else None
    out = torch.empty(M,Ho,Wo,Cout,device='cuda')
    def run():
        rc = lib.omni_conv2d_nhwc_f32(P(x1),None,P(w),P(b),P(res),P(out),M,H,W,C1,C2,Cout,k,k,s,pad,act,S()); assert rc==0, lib.omni_last_error()
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    tt = e0.elapsed_time(e1)/20*1e-3
    fl = 2*M*Ho*Wo*Cout*(C1+C2)*k*k
    print(cfg, '%.1f us %.1f TF/s' % (tt*1e6, fl/tt/1e12))
