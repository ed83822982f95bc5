"""spherical_fusion (iterative) — host-side mirror of /root/reference/model/spherical_model_iterative.py:253-456.

    net = spherical_fusion()                       # reference default patch_size (256,256) cannot run;
    net = spherical_fusion(patch_size=(128, 128))  # the only size at which the reference network exists
    outs = net(rgb, iter=2, confidence=False)      # list of `iter` [B,1,H,W] depth maps (test.py:198-199)

Iteration k >= 2 re-projects the previous ERP depth into P/4 patches (equi2pers on a 1-channel map,
:385), scales the unit rays by it and feeds mlp_points2 (:387-393); the RGB patches of :384 are the
same tensor as :315 and are sampled once.
"""
import torch

from .spherical_model import spherical_fusion as _single
from ..equi_pers.equi2pers_v3 import equi2pers_patches, equi2pers_aux
from .. import _lib


class spherical_fusion(_single):
    _ITERATIVE = True

    def __init__(self, nrows=4, npatches=18, patch_size=(256, 256), fov=(80, 80)):
        super().__init__(nrows, npatches, patch_size, fov)

    @torch.no_grad()
    def forward(self, high_res, iter, confidence=False):
        with self._execution(high_res):
            return self._forward(high_res, iter, confidence)

    def _forward(self, high_res, iter, confidence):
        self._check(high_res)
        e = self._eng
        bs, _, H, W = high_res.shape
        P = e.patch_size[0]
        p4 = (P // 4, P // 4)
        with torch.cuda.device(high_res.device):
            patches = equi2pers_patches(high_res, self.fov, self.nrows, self.patch_size, layout=_lib.LAYOUT_BNCHW)   # :315 (= :384)
            self._input_read = torch.cuda.current_stream(high_res.device).record_event() if self._want_input_event else None
            xyz, _, _ = equi2pers_aux(high_res.device, self.fov, self.nrows, p4, want_xyz=True, want_uv=False)       # :316
            pf = e.mlp_points("mlp_points1", xyz, None, self.npatches)                                               # :319
            a, c = self.network(patches, bs, confidence, point_feat=pf)
            outs = [e.blend(a, c, (H, W))]                                                                           # :371-380
            for i in range(iter - 1):                                                                                # :383
                depth = equi2pers_patches(outs[i], self.fov, self.nrows, p4, layout=_lib.LAYOUT_BNCHW)               # :385 [B,N,1,p,p]
                pf = e.mlp_points("mlp_points2", xyz, depth, bs * self.npatches)                                     # :387-393
                a, c = self.network(patches, bs, confidence, point_feat=pf)
                outs.append(e.blend(a, c, (H, W)))                                                                   # :444-454
        return outs
