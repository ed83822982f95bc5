"""Point-cloud export on the device — host-side mirror of /root/reference/test.py:210-240 (coords2uv / uv2xyz x depth,
util.py:159-174) and of the binary PLY that ply.write_ply (ply.py:217-330) produces for it.

    pts, col = depth_to_pointcloud(depth, rgb)                  # [B,H*W,3] float32 xyz = rays * depth, [B,H*W,3] uint8 colours
    write_ply_pointcloud("pred_0", depth[0:1], rgb[0:1])        # = write_ply(name, [predxyz_np, rgb_img], ['x','y','z','blue','green','red'])

The rays, the scaling and the 15-byte vertex records (3 x float32 + 3 x uint8, exactly the numpy structured array of
ply.py:303-314) are produced by ONE kernel (csrc/omni_io.hip `pointcloud_kernel`); the host only copies the finished records
and prepends the text header, instead of the reference's per-batch `.detach().cpu().numpy()` of four full-size tensors.
"""
import ctypes
import sys

import numpy as np
import torch

from . import _lib

FIELDS = ("x", "y", "z", "blue", "green", "red")        # test.py:237-238: the loader's channel order is BGR (quirk q10)


def _records(depth, rgb):
    if not depth.is_cuda or not rgb.is_cuda:
        raise ValueError("depth and rgb must live on an MI355X device; there is no CPU path")
    if depth.dim() != 4 or depth.shape[1] != 1 or rgb.dim() != 4 or rgb.shape[1] != 3 or depth.shape[0] != rgb.shape[0] or depth.shape[2:] != rgb.shape[2:]:
        raise ValueError("expected depth [B,1,H,W] and rgb [B,3,H,W]")
    d = depth.contiguous().to(torch.float32); c = rgb.contiguous().to(torch.float32)
    B, _, H, W = d.shape
    rec = torch.empty((B, H * W, 15), dtype=torch.uint8, device=d.device)
    with torch.cuda.device(d.device):
        _lib.check(_lib.load().omni_pointcloud_ply_f32(ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(c.data_ptr()), ctypes.c_void_p(rec.data_ptr()),
                                                      B, H, W, _lib.stream_of(d)), "pointcloud")
    return rec


def depth_to_pointcloud(depth, rgb):
    rec = _records(depth, rgb)
    pts = rec[..., :12].contiguous().view(torch.float32)                      # [B, H*W, 3]
    return pts, rec[..., 12:].contiguous()


def write_ply_pointcloud(filename, depth, rgb):
    """One file per call, first batch item (test.py:233-238 writes item 0 of every 20th batch)."""
    rec = _records(depth[:1], rgb[:1])[0].cpu().numpy()
    if not filename.endswith(".ply"):
        filename += ".ply"                                                    # ply.py:272-273
    header = ["ply", "format binary_" + sys.byteorder + "_endian 1.0", "element vertex %d" % rec.shape[0]]
    header += ["property float32 %s" % n for n in FIELDS[:3]] + ["property uint8 %s" % n for n in FIELDS[3:]]
    header.append("end_header")
    with open(filename, "w") as fh:
        for line in header:
            fh.write("%s\n" % line)
    with open(filename, "ab") as fh:
        np.ascontiguousarray(rec).tofile(fh)
    return filename
