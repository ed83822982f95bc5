import sys, ctypes, time, numpy as np, torch, torch.nn.functional as F
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def split_w(w):   # [Cout][K] fp32 -> [Cout][K/32][2][32] half
    hi = w.half(); hi = torch.where(w.abs() < 6.103515625e-05, torch.zeros_like(hi), hi)
    lo = ((w - hi.float()) * 2048.0).half()
    Co, K = w.shape
    return torch.stack([hi.reshape(Co, K // 32, 32), lo.reshape(Co, K // 32, 32)], 2).contiguous()

def to_sh(x):
    out = torch.empty(x.numel() * 2, dtype=torch.float16, device=x.device)
    assert lib.omni_f32_to_sh(P(x), P(out), ctypes.c_size_t(x.numel()), S()) == 0
    return out

import os
CFGS = [(3,16,16,64,0,64,3,1,1,1,True), (5,32,32,64,0,128,3,2,1,1,False), (4,8,8,256,256,128,3,1,1,1,False),
            (36,4,4,512,0,512,3,1,1,1,True), (40,64,64,32,0,32,3,1,1,1,False), (144,32,32,64,0,64,3,1,1,1,True), (144,16,16,128,0,128,3,1,1,1,True)]
if os.environ.get('ONLY'): CFGS=[CFGS[int(os.environ['ONLY'])]]
for cfg in CFGS:
    M,H,W,C1,C2,Cout,k,s,pad,act,use_res = cfg
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(M,H,W,C1,generator=g); x2 = torch.randn(M,H,W,C2,generator=g) if C2 else None
    w = torch.randn(Cout,C1+C2,k,k,generator=g)/np.sqrt((C1+C2)*k*k); b = torch.randn(Cout,generator=g)
    Ho,Wo = (H+2*pad-k)//s+1,(W+2*pad-k)//s+1
    res = torch.randn(M,Ho,Wo,Cout,generator=g) if use_res else None
    ref = None
    if M*H*W*C1 < 3e6:
        xin = torch.cat([x1,x2],-1) if C2 else x1
        ref = F.conv2d(xin.permute(0,3,1,2).double(), w.double(), b.double(), stride=s, padding=pad).permute(0,2,3,1)
        if use_res: ref = ref + res.double()
        if act==1: ref = F.relu(ref)
    wt = split_w(w.permute(0,2,3,1).reshape(Cout,-1).contiguous()).cuda()
    X1 = to_sh(x1.cuda()); X2 = to_sh(x2.cuda()) if C2 else None; R = to_sh(res.cuda()) if use_res else None
    B = b.cuda()
    out = torch.empty(M*Ho*Wo*Cout*2, dtype=torch.float16, device='cuda')
    def run():
        rc = lib.omni_conv2d_sh_f16x3(P(X1),P(X2),P(wt),P(B),P(R),P(out),M,H,W,C1,C2,Cout,k,k,s,pad,act,0,S()); assert rc==0, lib.omni_last_error()
    run(); torch.cuda.synchronize()
    o32 = torch.empty(M,Ho,Wo,Cout,device='cuda'); lib.omni_sh_to_f32(P(out),P(o32),ctypes.c_size_t(o32.numel()),S())
    err = (o32.cpu().double()-ref).abs().max().item() if ref is not None else float('nan')
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    tt = e0.elapsed_time(e1)/20*1e-3
    fl = 2*M*Ho*Wo*Cout*(C1+C2)*k*k
    print(cfg, 'err %.2e' % err, '%.1f us %.1f TF/s' % (tt*1e6, fl/tt/1e12))
