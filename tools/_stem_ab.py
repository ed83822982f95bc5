import sys, ctypes, torch, time
sys.path.insert(0, '.')
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
new = _lib.load(); old = ctypes.CDLL("./gpurun_in_old.so")
P_ = lambda t: ctypes.c_void_p(t.data_ptr())
g = torch.Generator().manual_seed(3)
for (M, P) in ((3, 32), (18, 128), (144, 128)):
    src = (torch.rand(M, 3, P, P, generator=g) * 2 - 1).cuda()
    wt = (torch.randn(147, 64, generator=g) / 12); b = torch.randn(64, generator=g).cuda()
    wk = torch.zeros(64, 3, 7, 8); wk[..., :7] = wt.reshape(7, 7, 3, 64).permute(3, 2, 0, 1)
    w16 = split_weights_f16x3(torch.cat([wk.reshape(64, 168), torch.zeros(64, 24)], 1)).cuda()
    outs = []
    for lib in (new, old):
        o = torch.empty(M, P // 2, P // 2, 64, device="cuda")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.omni_stem_sh_f16x3(P_(src), P_(w16), P_(b), P_(o), M, P, st) == 0
        for _ in range(3): lib.omni_stem_sh_f16x3(P_(src), P_(w16), P_(b), P_(o), M, P, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.omni_stem_sh_f16x3(P_(src), P_(w16), P_(b), P_(o), M, P, st)
        e1.record(); torch.cuda.synchronize()
        outs.append((o.clone(), e0.elapsed_time(e1) / 20 * 1e3))
    print("M=%d P=%d: new %.1f us  old %.1f us  same bits: %s" % (M, P, outs[0][1], outs[1][1], torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32))), flush=True)
