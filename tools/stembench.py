"""The stem (conv 7x7 s2 + BN + ReLU on the matrix cores) alone at M patches of P x P, per value of OMNI_CONV_EPI_LDS, with a bit comparison."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
lib = _lib.load()
M, P = int(os.environ.get("M", "144")), 128
net = spherical_fusion(4, 18, (P, P), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
net(torch.rand((1, 3, 512, 1024), device="cuda"))                     # packs the weights
w = net._eng.w
x = torch.rand((M, 3, P, P), device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ref = None
for v in (0, 1, 0, 1):
    _lib.set_option("conv_epi_lds", v)
    out = torch.empty((M, P // 2, P // 2, 64), device="cuda")
    run = lambda: _lib.check(lib.omni_stem_sh_f16x3(p(x), p(w["stem.w16"]), p(w["stem.b"]), p(out), M, P, s()), "stem")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    print(f"stem M={M} conv_epi_lds={v}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us  {'same bits' if torch.equal(out, ref) else 'DIFFERENT'}")
