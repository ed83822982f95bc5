"""TEST INFRASTRUCTURE ONLY — never imported by the product path.

Imports the upstream OmniFusion Python from /root/reference *in this container*
so that golden vectors can be generated from the reference itself
(oracle/gen_golden.py) and so that the restatements in oracle/ can be pinned
against it.  /root/reference does not exist on the GPU box: nothing in
`-m gpu` tests, smoke() or bench.py may call into this module.

The reference imports a few packages the image lacks.  They are only touched at
import time or under `__main__`, so empty shims are injected into sys.modules
(the reference files themselves are never modified or copied):

  cv2                       equi2pers_v3.py:6, pers2equi_v3.py:7 (used under __main__ only)
  timm.models.layers        model/blocks.py:6   (DropPath -> unused at drop_path=0, blocks.py:80)
  timm.models.registry      model/blocks.py:7
  timm.models.vision_transformer  model/blocks.py:8
  mmseg.utils               model/blocks.py:10
  mmcv.runner               model/blocks.py:11
  torchvision.models.resnet34     model/spherical_model.py:197 (pretrained weights cannot
                            be downloaded; BASELINE config 1 says "random weights").  The
                            stub below restates the *public topology* of torchvision's
                            ResNet-34 (module names conv1,bn1,relu,maxpool,layer1-4,
                            BasicBlock{conv1,bn1,relu,conv2,bn2,downsample}) which is what
                            the reference's state_dict schema depends on.
"""
import os
import sys
import types
import contextlib
import tempfile

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("OMNI_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "equi_pers", "equi2pers_v3.py"))


# --------------------------------------------------------------------------- stubs
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class _ResNet34(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, 3, 1)
        self.layer2 = self._make(128, 4, 2)
        self.layer3 = self._make(256, 6, 2)
        self.layer4 = self._make(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)

    def _make(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes))
        layers = [_BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(_BasicBlock(planes, planes))
        return nn.Sequential(*layers)


def _install_stubs():
    def mod(name):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        return m

    mod("cv2")
    timm = mod("timm"); tm = mod("timm.models"); tl = mod("timm.models.layers")
    tr = mod("timm.models.registry"); tv = mod("timm.models.vision_transformer")
    timm.models = tm; tm.layers = tl; tm.registry = tr; tm.vision_transformer = tv
    tl.DropPath = lambda p=0.0: nn.Identity()
    tl.to_2tuple = lambda t: t if isinstance(t, tuple) else (t, t)
    tl.trunc_normal_ = nn.init.trunc_normal_
    tr.register_model = lambda f: f
    tv._cfg = lambda **kw: dict(kw)
    mmseg = mod("mmseg"); mu = mod("mmseg.utils"); mmseg.utils = mu
    mu.get_root_logger = lambda *a, **k: None
    mmcv = mod("mmcv"); mr = mod("mmcv.runner"); mmcv.runner = mr
    mr.load_checkpoint = lambda *a, **k: None
    tvn = mod("torchvision"); tvm = mod("torchvision.models"); tvn.models = tvm
    tvm.resnet34 = lambda pretrained=False, **kw: _ResNet34()


_loaded = {}


def load_reference():
    """Return a namespace with the reference's hot-path callables."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    e2p = importlib.import_module("equi_pers.equi2pers_v3")
    p2e = importlib.import_module("equi_pers.pers2equi_v3")
    sm = importlib.import_module("model.spherical_model")
    smi = importlib.import_module("model.spherical_model_iterative")
    _loaded.update(equi2pers=e2p.equi2pers, pers2equi=p2e.pers2equi,
                   spherical_fusion=sm.spherical_fusion,
                   spherical_fusion_iterative=smi.spherical_fusion)
    return types.SimpleNamespace(**_loaded)


@contextlib.contextmanager
def scratch_cwd():
    """The reference's pers2equi writes ./grid/<layer_name>.pth relative to cwd
    (pers2equi_v3.py:24-29,156) keyed by name only.  Run every oracle call in a
    fresh scratch directory so no stale table is ever reused."""
    old = os.getcwd()
    with tempfile.TemporaryDirectory(prefix="omni_ref_") as d:
        os.chdir(d)
        try:
            yield d
        finally:
            os.chdir(old)


def ref_equi2pers(erp, fov, nrows, patch_size):
    ref = load_reference()
    with torch.no_grad():
        return ref.equi2pers(erp, fov, nrows, patch_size)


def ref_pers2equi(pers, fov, nrows, patch_size, erp_size):
    ref = load_reference()
    with scratch_cwd(), torch.no_grad():
        return ref.pers2equi(pers, fov, nrows, patch_size, erp_size, "oracle")
