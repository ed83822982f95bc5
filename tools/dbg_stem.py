import sys, ctypes, torch
sys.path.insert(0, '.')
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
lib = _lib.load()
P_ = lambda t: ctypes.c_void_p(t.data_ptr())
g = torch.Generator().manual_seed(3)
for (M, P) in ((3, 32), (2, 64), (2, 128)):
    src = torch.rand(M, 3, P, P, generator=g).cuda()
    wt = (torch.randn(147, 64, generator=g) / 12); b = torch.randn(64, generator=g).cuda()
    o32 = torch.empty(M, P // 2, P // 2, 64, device="cuda"); osh = torch.empty_like(o32); o2 = torch.empty_like(o32)
    assert lib.omni_stem_f32(P_(src), P_(wt.cuda()), P_(b), P_(o32), M, P, None) == 0
    wk = torch.zeros(64, 3, 7, 8); wk[..., :7] = wt.reshape(7, 7, 3, 64).permute(3, 2, 0, 1)
    w16 = split_weights_f16x3(torch.cat([wk.reshape(64, 168), torch.zeros(64, 24)], 1)).cuda()
    assert lib.omni_stem_sh_f16x3(P_(src), P_(w16), P_(b), P_(osh), M, P, None) == 0
    lib.omni_sh_to_f32(P_(osh), P_(o2), ctypes.c_size_t(o2.numel()), None)
    torch.cuda.synchronize()
    d = (o2 - o32).abs()
    print(M, P, "max err", d.max().item(), "bad px", (d.amax(-1) > 1e-4).nonzero()[:8].tolist())
for C in (32, 64, 512):
    x = torch.randn(3, 10, 12, C, generator=g).cuda(); xs = torch.empty_like(x)
    lib.omni_sh_from_f32(P_(x), P_(xs), ctypes.c_size_t(x.numel()), None)
    u32 = torch.empty(3, 20, 24, C, device="cuda"); ush = torch.empty_like(u32); u2 = torch.empty_like(u32)
    lib.omni_upsample_bilinear_f32(P_(x), P_(u32), 3, 10, 12, C, 20, 24, None)
    lib.omni_upsample_bilinear_sh(P_(xs), P_(ush), 3, 10, 12, C, 20, 24, None)
    lib.omni_sh_to_f32(P_(ush), P_(u2), ctypes.c_size_t(u2.numel()), None)
    torch.cuda.synchronize()
    print("up C", C, (u2 - u32).abs().max().item())
