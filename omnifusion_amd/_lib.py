"""ctypes loader of libomnifusion_hip.so — the ONLY compute path of this package.

There is no CPU / PyTorch fallback: if the HIP library is missing or fails to load,
importing the operators raises (the product path must fail loudly, never silently run
something else).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OMNI_LIB_VARIANT") or os.path.join(_HERE, "csrc", "libomnifusion_hip.so")    # (OMNI_LIB_VARIANT: tools/ only — a variant build of the library for same-box A/B timings)
LIB_PATH_DEBUG = os.path.join(_HERE, "csrc", "libomnifusion_hip_dbg.so")     # tools/ only: ablation switches + micro-benchmarks

OMNI_OK, OMNI_ERR_INVALID, OMNI_ERR_HIP, OMNI_ERR_UNSUPPORTED = 0, 1, 2, 3
LAYOUT_BCHWN, LAYOUT_BNCHW, LAYOUT_BNHWC = 0, 1, 2
F32, F16 = 0, 1

_lib = None


class OmniLibraryMissing(ImportError):
    pass


# every symbol include/omnifusion.h declares (tests/test_boundary.py parses the header and
# checks this list against it and against the built library)
EXPORTS = [
    "omni_version", "omni_last_error", "omni_set_option", "omni_get_option", "omni_num_patches", "omni_patch_centers",
    "omni_geometry_create", "omni_geometry_destroy", "omni_geometry_cache_clear", "omni_geometry_cache_size",
    "omni_equi2pers", "omni_equi2pers_aux", "omni_pers2equi", "omni_pers2equi_conf", "omni_patches_to_planar",
    "omni_equi2pers_g", "omni_pers2equi_g", "omni_equi2pers_bwd", "omni_pers2equi_bwd",
    "omni_conv2d_nhwc_f32", "omni_stem_f32", "omni_maxpool3x3s2_f32", "omni_upsample_bilinear_f32",
    "omni_add_hw_f32", "omni_add_period_f32", "omni_token_pack_f32", "omni_layernorm512_f32",
    "omni_attention_f32", "omni_heads_f32", "omni_mlp_points_f32",
    "omni_conv2d_nhwc_f16x3_ws", "omni_conv2d_sh_f16x3_ws", "omni_conv2d_sh_f16x3_post_ws", "omni_gemm_rows_sh_f16x3", "omni_gemm_rows_ln_sh_f16x3", "omni_gemm_rows_pack", "omni_conv3x3_up2_sh_f16x3", "omni_conv3x3_up2_heads_sh_f16x3", "omni_up2_heads_scratch_bytes", "omni_heads_pack_f16x3", "omni_sh_from_f32", "omni_sh_to_f32", "omni_sh_overflow",
    "omni_stem_sh", "omni_stem_sh_f16x3", "omni_maxpool3x3s2_sh", "omni_upsample_bilinear_sh", "omni_add_hw_sh", "omni_add_period_sh", "omni_layernorm512_sh", "omni_gemm_sh_f16x3_ln512_ws", "omni_gemm_rows_slices_sh_f16x3", "omni_gemm_rows_ln_parts_sh_f16x3", "omni_splitk_reduce_ln512", "omni_wino_input_sh", "omni_conv3x3_wino_sh_f16x3", "omni_attention_qkv_sh",
    "omni_conv2d_splitk_plan", "omni_conv2d_nhwc_f32_ws",
    "omni_masked_median_f32", "omni_depth_metrics_f32",
    "omni_png_info", "omni_png_decode", "omni_png_decode_batch", "omni_zlib_inflate", "omni_png_checksums",
    "omni_preprocess_rgb_u8", "omni_preprocess_depth_u16", "omni_berhu_workspace_bytes", "omni_berhu_loss_f32", "omni_berhu_grad_f32",
    "omni_pointcloud_ply_f32",
]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OmniLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m omnifusion_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    # The library and PyTorch must share ONE HIP runtime (device pointers and streams cross the
    # boundary).  Both name libamdhip64.so.7; importing torch first makes the dynamic loader bind
    # our DT_NEEDED entry to the copy torch already mapped (loading /opt/rocm's copy first leaves
    # two runtimes in the process and the second one sees no device).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    lib.omni_last_error.restype = ctypes.c_char_p
    lib.omni_version.restype = ctypes.c_int
    lib.omni_up2_heads_scratch_bytes.restype = ctypes.c_size_t
    for name in EXPORTS:
        getattr(lib, name)          # AttributeError here = header/library mismatch
    _lib = lib
    return lib


def load_debug():
    """The DEBUG build of the library (python -m omnifusion_amd.build --debug): same entry points plus the
    result-changing ablation switches (OMNI_*_DBG) and the omni_debug_* micro-benchmarks.  tools/ only — nothing under
    omnifusion_amd/ or tests/ loads it."""
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH_DEBUG):
        raise OmniLibraryMissing(f"{LIB_PATH_DEBUG} not found: python -m omnifusion_amd.build --debug")
    lib = ctypes.CDLL(LIB_PATH_DEBUG)
    lib.omni_last_error.restype = ctypes.c_char_p
    return lib


def set_option(name, value):
    check(load().omni_set_option(name.encode(), int(value)), "set_option")


def get_option(name):
    v = ctypes.c_int(0)
    check(load().omni_get_option(name.encode(), ctypes.byref(v)), "get_option")
    return v.value


CALL_LOG = None          # tools/fwd.py --labels: a list that receives the label of every library call (profiles/ label their kernel rows with it)


def check(status, what=""):
    """Map C-ABI status codes to the exceptions the Python boundary promises
    (SURVEY.md §8b 'Errors'): ValueError for bad arguments, RuntimeError for HIP errors."""
    if CALL_LOG is not None:
        CALL_LOG.append(what)
    if status == OMNI_OK:
        return
    msg = load().omni_last_error().decode(errors="replace")
    if status == OMNI_ERR_INVALID:
        raise ValueError(f"{what}: {msg}")
    if status == OMNI_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg}")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise ValueError(f"unsupported dtype {t.dtype}: the HIP path stores float32 or float16")
