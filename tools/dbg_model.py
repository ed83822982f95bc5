import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from _util import golden
from omnifusion_amd import _lib
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
lib = _lib.load()
g = golden("G6_model_single")
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda()
net.load_state_dict(make_state_dict(42, 18, False))
rgb = torch.from_numpy(g["rgb"]).cuda()
def err(): return float(np.abs(net(rgb, confidence=True).cpu().numpy() - g["depth_conf"]).max())
print("default", err())
orig_stem = lib.omni_stem_sh_f16x3
def stem_valu(src, w16, b, dst, M, P, s):
    return lib.omni_stem_sh(src, ctypes.c_void_p(net._eng.w["stem.w"].data_ptr()), b, dst, M, P, s)
lib.omni_stem_sh_f16x3 = stem_valu
print("valu stem", err())
lib.omni_stem_sh_f16x3 = orig_stem
