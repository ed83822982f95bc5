"""Host-side engine of the network: weight packing (BN folding, NHWC-GEMM layouts) and the launch
sequence of one forward over the C-ABI operators of libomnifusion_hip.so.

Everything numeric runs in the HIP library; PyTorch only owns the device buffers (caching
allocator), the stream and — once per `load_state_dict` — the constant folding of the weights
(BatchNorm -> conv, the input-independent `mlp_points` branch of the single-pass model).

Reference: model/spherical_model.py:238-314 (single pass), model/spherical_model_iterative.py:308-456.
"""
import ctypes
import os

import torch

from .. import _lib
from ..equi_pers.equi2pers_v3 import equi2pers_patches, equi2pers_aux, pair, _NPATCH
from ..equi_pers.pers2equi_v3 import pers2equi, pers2equi_conf

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
_LAYERS = [("layer1", 3, 1), ("layer2", 4, 2), ("layer3", 6, 2), ("layer4", 3, 2)]
_BN_EPS = 1e-5


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def split_weights_f16x3(w):
    """[Cout][K] fp32 (K % 32 == 0) -> halfs [Cout][K/32][2][32]: hi = fp16(w) (0 below the fp16 normal range),
    lo = fp16((w - hi) * 2048) — the operand format of omni_conv2d_nhwc_f16x3_ws."""
    w = w.to(torch.float32)
    wmax = float(w.abs().max()) if w.numel() else 0.0
    if not wmax < 65504.0:                      # also catches NaN / inf
        raise ValueError(f"folded weight magnitude {wmax:g} is outside the fp16 range of the f16x3 operand format: "
                         "run this checkpoint with OMNI_NET_PRECISION=fp32")
    hi = w.half()
    hi = torch.where(w.abs() < 6.103515625e-05, torch.zeros_like(hi), hi)
    lo = ((w - hi.float()) * 2048.0).half()
    co, k = w.shape
    return torch.stack([hi.reshape(co, k // 32, 32), lo.reshape(co, k // 32, 32)], 2).contiguous()


def strip_module_prefix(sd):
    """checkpoints saved through nn.DataParallel carry a 'module.' prefix (train_erp_depth.py:307)"""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


class Engine:
    def __init__(self, nrows, npatches, patch_size, fov, iterative):
        self.nrows, self.npatches, self.iterative = nrows, npatches, iterative
        self.patch_size = pair(patch_size)
        self.fov = pair(fov)
        if self.patch_size[0] != self.patch_size[1]:
            raise ValueError("square patches only")
        self.w = None          # packed weights (device tensors)
        self.device = None
        # "f16x3": conv products on the fp16 matrix cores with split operands (fp32-class accuracy, 3/16 of the MFMA time),
        # activations between the convolutions in the split-half layout; "fp32": the exact fp32 MFMA, fp32 NHWC activations.
        # OMNI_NET_PRECISION overrides.
        import os
        self.precision = os.environ.get("OMNI_NET_PRECISION", "f16x3")

    @property
    def sh(self):
        return self.precision == "f16x3"

    def lane(self):
        """A second execution context over the SAME packed weights (own split-K workspace and per-call state): the model
        runs the two halves of a batch on two streams so that one half's kernel tails overlap the other half's work."""
        e = Engine(self.nrows, self.npatches, self.patch_size, self.fov, self.iterative)
        e.precision, e.w, e.device, e.head_bias = self.precision, self.w, self.device, getattr(self, "head_bias", None)
        return e

    # ------------------------------------------------------------------ packing
    @staticmethod
    def _fold(sd, conv, bn):
        w = sd[conv + ".weight"].double()
        if w.dim() == 5:
            w = w[..., 0]
        g = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + _BN_EPS)
        b = sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * g
        return w * g[:, None, None, None], b

    @staticmethod
    def _gemm_layout(w):
        # [O, I, kh, kw] -> [O][kh*kw*I] with k = (ky, kx, c)
        return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)

    def pack(self, state_dict, device):
        sd = {k: v.detach().cpu() for k, v in strip_module_prefix(state_dict).items()}
        dev = torch.device(device)
        f = lambda t: t.to(torch.float32).contiguous().to(dev)
        W = {}
        w, b = self._fold(sd, "conv1", "bn1")                                    # stem: [147][64], k = (ky*7+kx)*3 + c
        W["stem.w"] = f(w.permute(2, 3, 1, 0).reshape(147, 64)); W["stem.b"] = f(b)
        wk = torch.zeros((64, 3, 7, 8), dtype=torch.float64); wk[..., :7] = w                 # k = (c*7 + ky)*8 + kx, kx = 7 zero
        W["stem.w16"] = split_weights_f16x3(torch.cat([wk.reshape(64, 168), torch.zeros((64, 24), dtype=torch.float64)], 1)).to(dev)

        def convbn(key, conv, bn):
            w, b = self._fold(sd, conv, bn)
            wg = self._gemm_layout(w).to(torch.float32)
            W[key + ".w"] = f(wg); W[key + ".b"] = f(b)
            W[key + ".w16"] = split_weights_f16x3(wg).to(dev)
        for lname, nblk, _ in _LAYERS:
            for i in range(nblk):
                p = f"{lname}.{i}"
                convbn(p + ".c1", p + ".conv1", p + ".bn1")
                convbn(p + ".c2", p + ".conv2", p + ".bn2")
                if (p + ".downsample.0.weight") in sd:
                    convbn(p + ".ds", p + ".downsample.0", p + ".downsample.1")
        for name in ("de_conv0_0", "de_conv0_1", "de_conv1_0", "de_conv1_1", "de_conv2_0", "de_conv2_1",
                     "de_conv3_0", "de_conv3_1", "de_conv4_0"):
            convbn(name, name + ".conv", name + ".bn")
        down = "down1" if self.iterative else "down"
        W["down.w"] = f(sd[down + ".weight"].reshape(32, 512)); W["down.b"] = f(sd[down + ".bias"])
        W["down.w16"] = split_weights_f16x3(sd[down + ".weight"].reshape(32, 512)).to(dev)
        W["pos"] = f(sd["transformer.pos_emb"].reshape(self.npatches, 512))
        for i in range(6):
            p = f"transformer.layer.{i}"
            for k in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "attn.q.weight", "attn.kv.weight",
                      "attn.proj.weight", "attn.proj.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"):
                W[f"t{i}.{k}"] = f(sd[f"{p}.{k}"])
        for i in range(6):                                                       # f16x3 operands of the transformer GEMMs
            t = f"t{i}."
            W[t + "attn.qkv.w16"] = split_weights_f16x3(torch.cat([W[t + "attn.q.weight"], W[t + "attn.kv.weight"]], 0).cpu()).to(dev)
            for k in ("attn.proj", "mlp.fc1", "mlp.fc2"):
                W[t + k + ".w16"] = split_weights_f16x3(W[t + k + ".weight"].cpu()).to(dev)
        W["enc_norm.w"] = f(sd["transformer.encoder_norm.weight"]); W["enc_norm.b"] = f(sd["transformer.encoder_norm.bias"])
        hw = torch.stack([sd["pred.weight"][0, :, :, :, 0], sd["weight_pred.weight"][0, :, :, :, 0]])    # [2,32,3,3]
        W["heads.w"] = f(hw.permute(0, 2, 3, 1).reshape(2, 9, 32))
        import numpy as np                                                       # the same weights as the f16x3 operand of the fused de_conv4_0 + heads kernel
        hw_host = np.ascontiguousarray(hw.permute(0, 2, 3, 1).reshape(2, 9, 32).to(torch.float32).numpy())
        frag = np.zeros(4 * 64 * 8, np.float16)
        _lib.check(_lib.load().omni_heads_pack_f16x3(hw_host.ctypes.data_as(ctypes.c_void_p), frag.ctypes.data_as(ctypes.c_void_p)), "heads_pack")
        W["heads.w16f"] = torch.from_numpy(frag).to(dev)
        self.head_bias = (float(sd["pred.bias"][0]), float(sd["weight_pred.bias"][0]))

        def mlp(name):
            w1, b1 = self._fold(sd, name + ".0", name + ".1")
            w2, b2 = self._fold(sd, name + ".3", name + ".4")
            return w1[:, :, 0, 0], b1, w2[:, :, 0, 0], b2
        P4 = self.patch_size[0] // 4
        if self.iterative:
            for name in ("mlp_points1", "mlp_points2"):
                w1, b1, w2, b2 = mlp(name)
                W[name + ".w1"], W[name + ".b1"], W[name + ".w2"], W[name + ".b2"] = f(w1), f(b1), f(w2), f(b2)
        else:
            # single pass: the input of mlp_points is [cx, cy, 1, cx, cy] broadcast over the patch
            # (spherical_model.py:245-251) — a constant of the weights: fold it to one vector per patch.
            cp = (ctypes.c_float * (2 * self.npatches))()
            _lib.check(_lib.load().omni_patch_centers(int(self.nrows), 0, cp), "patch_centers")
            c = torch.tensor(list(cp), dtype=torch.float64).reshape(self.npatches, 2)
            x = torch.cat([c, torch.ones(self.npatches, 1, dtype=torch.float64), c], 1)          # [N,5]
            w1, b1, w2, b2 = mlp("mlp_points")
            h = torch.relu(x @ w1.T + b1)
            pf = torch.relu(h @ w2.T + b2)                                                         # [N,64]
            W["point_feat"] = f(pf[:, None, None, :].expand(self.npatches, P4, P4, 64))
        self.w, self.device = W, dev

    def read_overflow_flag(self, reset=True):
        """sticky range flag of the split-half format on this engine's device (omni_sh_overflow); synchronises"""
        if self.device is None:
            return False
        flag = ctypes.c_int(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().omni_sh_overflow(ctypes.byref(flag), 1 if reset else 0), "sh_overflow")
        return bool(flag.value)

    # ------------------------------------------------------------------ operator shims
    def _conv(self, x, key, M, H, Wd, C1, Cout, k, stride, pad, act, x2=None, C2=0, res=None, bias=True, out_f32=False, out=None, post=None):
        """One convolution (+ folded BN, bias, residual, activation).  In the f16x3 mode the activations x, x2, res and
        the result are split-half (SH) tensors (csrc/omni_sh.h) unless out_f32 asks for a plain fp32 NHWC result."""
        lib = _lib.load()
        Ho = (H + 2 * pad - k) // stride + 1
        Wo = (Wd + 2 * pad - k) // stride + 1
        if out is None:
            out = torch.empty((M, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
        S, ws, nb = self._splitk(M * Ho * Wo, Cout, k * k * (C1 + C2) // 32, x.device, shape=(k, k, stride, pad, H, Wd))
        b = _p(self.w[key + ".b"]) if bias else None
        late_post = None
        if post is not None and (S > 1 or not self.sh or out_f32):
            # the fused `+ post` epilogue exists for un-split SH launches only (omni_conv2d_sh_f16x3_post_ws rejects splitk > 1): a shape
            # whose plan splits K (few rows: small patches, nrows = 3, a lone panorama) adds it as a pass of its own instead
            late_post, post = post, None
        if self.sh and post is not None:                            # `+ post` after the activation, inside the epilogue (SH mode, never split)
            rc = lib.omni_conv2d_sh_f16x3_post_ws(_p(x), _p(x2), _p(self.w[key + ".w16"]), b, _p(res), _p(out), 0 if out_f32 else 1,
                                                  M, H, Wd, C1, C2, Cout, k, k, stride, pad, act, S, _p(ws), ctypes.c_size_t(nb),
                                                  _p(post), ctypes.c_size_t(post.numel()), self._s)
        elif self.sh:
            lat = 4 if (self._bs == 1 and self.latency_plan) else 0     # fmt bit 2: a lone panorama keeps the im2col tiles for 16-wide images
            rc = lib.omni_conv2d_sh_f16x3_ws(_p(x), _p(x2), _p(self.w[key + ".w16"]), b, _p(res), _p(out), (0 if out_f32 else 1) | lat,
                                             M, H, Wd, C1, C2, Cout, k, k, stride, pad, act, S, _p(ws), ctypes.c_size_t(nb), self._s)
        else:
            rc = lib.omni_conv2d_nhwc_f32_ws(_p(x), _p(x2), _p(self.w[key + ".w"]), b, _p(res), _p(out), M, H, Wd, C1, C2, Cout,
                                             k, k, stride, pad, act, S, _p(ws), ctypes.c_size_t(nb), self._s)
        _lib.check(rc, "conv2d " + key)
        if late_post is not None:
            _lib.check((lib.omni_add_period_f32 if (out_f32 or not self.sh) else lib.omni_add_period_sh)(   # an fp32 NHWC output is never split-half
                _p(out), _p(late_post), ctypes.c_size_t(out.numel()), ctypes.c_size_t(late_post.numel()), self._s), "add post " + key)
        return out

    NOMINAL_BATCH = 8
    SINGLE_BATCH = 4           # the batch a lone panorama plans its split-K factors for (latency_plan)
    latency_plan = True
    rows_gemm = True           # ... and its transformer GEMMs (18 rows) take the register-streaming kernel

    def _splitk(self, rows, Cout, ksteps, device, shape=None):
        """Split factor planned for a NOMINAL batch (not the actual one), so that the K summation order — and with it
        every output bit — is the same whether a panorama is processed inside a batch of 2 or of 64 (image-sharded
        multi-GPU runs reproduce single-GPU results bit for bit).  A LONE panorama (BASELINE cfg 2) is latency-bound —
        every layer is one round of at most one block per CU and costs the length of its K loop — and plans for
        SINGLE_BATCH instead: deeper splits on layer3/layer4/decoder, 1.28 -> 1.13 ms per forward, results equal to the
        batched ones to 2e-5 abs instead of bit for bit (`Engine.latency_plan = False` restores the one plan for all)."""
        plan_batch = self.SINGLE_BATCH if (self._bs == 1 and self.latency_plan) else self.NOMINAL_BATCH
        rows_plan = rows // self._bs * plan_batch
        lib = _lib.load()
        S = int(lib.omni_conv2d_splitk_plan(ctypes.c_longlong(rows_plan), Cout, ksteps))
        if S <= 1:
            return 1, None, 0
        ws, nb = self._workspace(S * rows * Cout * 4, device)
        return S, ws, nb

    def _workspace(self, nbytes, device):
        """split-K scratch: one buffer reused by every launch of the stream (launches are stream-ordered)"""
        nbytes = int(nbytes)
        if nbytes == 0:
            return None, 0
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() * 4 < nbytes or ws.device != device:
            if ws is not None:                   # a hipGraph captured earlier (graphed()) still replays into the old buffer:
                self._ws_retired = getattr(self, "_ws_retired", []) + [ws]       # keep it alive for the engine's lifetime
            self._ws = ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        return ws, ws.numel() * 4

    def _gemm(self, x, wkey, bkey, rows, K, Nout, act=ACT_NONE, res=None):
        lib = _lib.load()
        out = torch.empty((rows, Nout), dtype=torch.float32, device=x.device)
        S, ws, nb = self._splitk(rows, Nout, K // 32, x.device)
        rc = lib.omni_conv2d_nhwc_f32_ws(_p(x), None, _p(self.w[wkey]), _p(self.w[bkey]) if bkey else None, _p(res), _p(out),
                                         rows, 1, 1, K, 0, Nout, 1, 1, 1, 0, act, S, _p(ws), ctypes.c_size_t(nb), self._s)
        _lib.check(rc, "gemm " + wkey)
        return out

    def _rows_weights(self, w16key, Nout, K):
        """the transformer matrix `w16key` in the fragment order of omni_gemm_rows_sh_f16x3 (made on first use: only single-panorama
        forwards need the second copy, 75 MB for the six layers)"""
        key = w16key + ".rows"
        w = self.w.get(key)
        if w is None:
            src = self.w[w16key]
            w = torch.empty_like(src)
            _lib.check(_lib.load().omni_gemm_rows_pack(_p(src), _p(w), Nout, K, self._s), "gemm_rows_pack")
            if not torch.cuda.is_current_stream_capturing():     # engines on other streams share the dict: hand over a finished tensor
                torch.cuda.current_stream(src.device).synchronize()
            self.w[key] = w
        return w

    def _gemm_sh(self, x, w16key, bkey, rows, K, Nout, act=ACT_NONE, res=None, out_sh=False):
        """f16x3 GEMM on an SH activation matrix x [rows, K]; res (fp32) and bias optional; result fp32 or SH"""
        lib = _lib.load()
        out = torch.empty((rows, Nout), dtype=torch.float32, device=x.device)
        if self._bs == 1 and self.latency_plan and self.rows_gemm and rows <= 32 and K in (512, 2048):      # a lone panorama: the register-streaming form
            _lib.check(lib.omni_gemm_rows_sh_f16x3(_p(x), _p(self._rows_weights(w16key, Nout, K)), _p(self.w[bkey]) if bkey else None, _p(res), _p(out),
                                                   1 if out_sh else 0, rows, K, Nout, act, self._s), "gemm " + w16key)
            return out
        S, ws, nb = (1, None, 0) if K <= 512 else self._splitk(rows, Nout, K // 32, x.device)
        rc = lib.omni_conv2d_sh_f16x3_ws(_p(x), None, _p(self.w[w16key]), _p(self.w[bkey]) if bkey else None, _p(res), _p(out),
                                         (1 if out_sh else 0) | 2, rows, 1, 1, K, 0, Nout, 1, 1, 1, 0, act, S, _p(ws),
                                         ctypes.c_size_t(nb), self._s)
        _lib.check(rc, "gemm " + w16key)
        return out

    fuse_fc2_ln = os.environ.get("OMNI_FUSE_FC2_LN", "1") != "0"   # fc2's split-K second pass also applies the LayerNorm that follows (one launch fewer per layer, same bits)

    fc2_slices = int(os.environ.get("OMNI_FC2_SLICES", "4"))       # a lone panorama: fc2 in this many K slices (1: off) — 16 blocks stream its 4 MB of weights otherwise

    def _fc2_ln(self, h, w16key, bkey, rows, tok, nxt):
        """tok + fc2(h).  Batches: where fc2 runs split-K, the LayerNorm `nxt` = (weight key, bias key, eps, SH result?) of the result in the same second
        pass (omni_gemm_sh_f16x3_ln512_ws).  A lone panorama: fc2 as K slices whose sum the NEXT kernel forms (omni_gemm_rows_slices_sh_f16x3).
        Returns (new tok or None, LayerNorm(new tok) or None, pending K slices or None)."""
        lone = self._bs == 1 and self.latency_plan and self.rows_gemm and rows <= 32
        if lone and self.fc2_slices > 1 and self.fuse_ln:
            S = self.fc2_slices
            parts = torch.empty((S, rows, 512), dtype=torch.float32, device=h.device)
            _lib.check(_lib.load().omni_gemm_rows_slices_sh_f16x3(_p(h), _p(self._rows_weights(w16key, 512, 2048)), _p(parts), rows, 2048, 512, S, self._s), "gemm slices " + w16key)
            return None, None, (parts, S, bkey, tok)
        S, ws, nb = (1, None, 0) if lone else self._splitk(rows, 512, 2048 // 32, h.device)
        if lone or S <= 1 or not self.fuse_fc2_ln:
            return self._gemm_sh(h, w16key, bkey, rows, 2048, 512, res=tok), None, None
        wk, bk, eps, out_sh = nxt
        out, y = torch.empty((rows, 512), dtype=torch.float32, device=h.device), torch.empty((rows, 512), dtype=torch.float32, device=h.device)
        rc = _lib.load().omni_gemm_sh_f16x3_ln512_ws(_p(h), _p(self.w[w16key]), _p(self.w[bkey]), _p(tok), _p(out), _p(self.w[wk]), _p(self.w[bk]), ctypes.c_float(eps),
                                                     _p(y), 1 if out_sh else 0, rows, 2048, S, _p(ws), ctypes.c_size_t(nb), self._s)
        _lib.check(rc, "gemm+ln " + w16key)
        return out, y, None

    def _ln_gemm_parts(self, pending, lnw, lnb, eps, w16key, rows, Nout):
        """the next block's norm1 + qkv on a token matrix that still is fc2's K slices: (tok, qkv)"""
        parts, S, bkey, res_tok = pending
        tok = torch.empty((rows, 512), dtype=torch.float32, device=parts.device)
        out = torch.empty((rows, Nout), dtype=torch.float32, device=parts.device)
        rc = _lib.load().omni_gemm_rows_ln_parts_sh_f16x3(_p(parts), S, _p(self.w[bkey]), _p(res_tok), _p(tok), _p(self.w[lnw]), _p(self.w[lnb]), ctypes.c_float(eps),
                                                          _p(self._rows_weights(w16key, Nout, 512)), None, _p(out), 0, rows, Nout, ACT_NONE, self._s)
        _lib.check(rc, "ln+gemm (slices) " + w16key)
        return tok, out

    fuse_ln = os.environ.get("OMNI_FUSE_LN", "1") != "0"   # a lone panorama: LayerNorm inside the following rows GEMM (one launch instead of two, same bits)

    def _ln_gemm_sh(self, x, lnw, lnb, eps, w16key, bkey, rows, Nout, act=ACT_NONE, out_sh=False):
        """LayerNorm(512) + GEMM (K = 512) on the fp32 token matrix x [rows, 512]: one launch for a lone panorama, else the two kernels"""
        if self._bs == 1 and self.latency_plan and self.rows_gemm and self.fuse_ln and rows <= 32:
            lib = _lib.load()
            out = torch.empty((rows, Nout), dtype=torch.float32, device=x.device)
            _lib.check(lib.omni_gemm_rows_ln_sh_f16x3(_p(x), _p(self.w[lnw]), _p(self.w[lnb]), ctypes.c_float(eps), _p(self._rows_weights(w16key, Nout, 512)),
                                                      _p(self.w[bkey]) if bkey else None, None, _p(out), 1 if out_sh else 0, rows, Nout, act, self._s), "ln+gemm " + w16key)
            return out
        return self._gemm_sh(self._ln_sh(x, lnw, lnb, rows, eps), w16key, bkey, rows, 512, Nout, act=act, out_sh=out_sh)

    def _ln_sh(self, x, wk, bk, rows, eps):
        y = torch.empty_like(x)
        _lib.check(_lib.load().omni_layernorm512_sh(_p(x), _p(self.w[wk]), _p(self.w[bk]), _p(y), rows, ctypes.c_float(eps), self._s), "layernorm")
        return y

    def _ln(self, x, wk, bk, rows, eps):
        y = torch.empty_like(x)
        _lib.check(_lib.load().omni_layernorm512_f32(_p(x), _p(self.w[wk]), _p(self.w[bk]), _p(y), rows, ctypes.c_float(eps), self._s), "layernorm")
        return y

    def _up(self, x, M, H, Wd, C, Ho, Wo):
        y = torch.empty((M, Ho, Wo, C), dtype=torch.float32, device=x.device)
        fn = _lib.load().omni_upsample_bilinear_sh if self.sh else _lib.load().omni_upsample_bilinear_f32
        _lib.check(fn(_p(x), _p(y), M, H, Wd, C, Ho, Wo, self._s), "upsample")
        return y

    fold_point_feat = True     # `layer1 + point_feat` inside layer1's last convolution (one rounding to the SH format instead of two: results move by ~1e-7)
    fuse_up = True             # decoder: up-sampling computed inside the following convolution's halo fill where the shape allows

    def _up_conv(self, x, key, M, H, Wd, C, Cout, act, out_f32=False, out=None):
        """conv3x3(upsample2x(x)) — `F.interpolate` + the stage's first ConvBnReLU (:279-301); one kernel when the output is a
        multiple of 4 x 32 pixels (same bits as the two-kernel form)."""
        Ho, Wo = 2 * H, 2 * Wd
        if self.sh and self.fuse_up and Wo % 32 == 0 and Ho % 4 == 0:
            if out is None:
                out = torch.empty((M, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
            _lib.check(_lib.load().omni_conv3x3_up2_sh_f16x3(_p(x), _p(self.w[key + ".w16"]), _p(self.w[key + ".b"]), _p(out), (0 if out_f32 else 1) | (4 if (self._bs == 1 and self.latency_plan) else 0),
                                                             M, H, Wd, C, Cout, act, self._s), "up+conv " + key)
            return out
        return self._conv(self._up(x, M, H, Wd, C, Ho, Wo), key, M, Ho, Wo, C, Cout, 3, 1, 1, act, out_f32=out_f32, out=out)

    # ------------------------------------------------------------------ network over the patch batch
    def network(self, patches, point_feat, bs, confidence, out=None):
        """patches: planar [bs, N, 3, P, P]; point_feat: NHWC [N or bs*N, P/4, P/4, 64].
        Returns (a, c) planar [bs, N, 1, P, P]: a = relu(pred) (* conf), c = sigmoid(weight) or None."""
        lib = _lib.load()
        dev = patches.device
        self._s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        N, P = self.npatches, self.patch_size[0]
        M = bs * N
        self._bs = bs
        P2, P4, P8, P16, P32 = P // 2, P // 4, P // 8, P // 16, P // 32
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        conv1 = new(M, P2, P2, 64)
        sh = self.sh                                                # activations between the convolutions: SH or fp32 NHWC
        x = new(M, P4, P4, 64)
        pat = patches.view(M, 3, P, P)
        for m0, m1 in self._chunks(bs, N, self.front_chunk):        # stem -> maxpool per chunk: conv1 is re-read while still cached
            Mc = m1 - m0
            if sh and P % 32 == 0:
                _lib.check(lib.omni_stem_sh_f16x3(_p(pat[m0:m1]), _p(self.w["stem.w16"]), _p(self.w["stem.b"]), _p(conv1[m0:m1]), Mc, P, self._s), "stem")
            else:
                _lib.check((lib.omni_stem_sh if sh else lib.omni_stem_f32)(_p(pat[m0:m1]), _p(self.w["stem.w"]), _p(self.w["stem.b"]),
                                                                            _p(conv1[m0:m1]), Mc, P, self._s), "stem")
            _lib.check((lib.omni_maxpool3x3s2_sh if sh else lib.omni_maxpool3x3s2_f32)(_p(conv1[m0:m1]), _p(x[m0:m1]), Mc, P2, P2, 64, self._s), "maxpool")
        feats = {}
        cin, size = 64, P4
        for lname, nblk, stride in _LAYERS:
            cout = {"layer1": 64, "layer2": 128, "layer3": 256, "layer4": 512}[lname]
            for i in range(nblk):
                p = f"{lname}.{i}"
                s = stride if i == 0 else 1
                ident = x
                if (p + ".ds.w") in self.w:
                    ident = self._conv(x, p + ".ds", M, size, size, cin, cout, 1, s, 0, ACT_NONE)
                y = self._conv(x, p + ".c1", M, size, size, cin, cout, 3, s, 1, ACT_RELU)
                size = (size + 2 - 3) // s + 1
                fold = sh and self.fold_point_feat and lname == "layer1" and i == nblk - 1     # layer1 + point_feat (:258) in the block's last epilogue
                x = self._conv(y, p + ".c2", M, size, size, cout, cout, 3, 1, 1, ACT_RELU, res=ident, post=point_feat if fold else None)
                cin = cout
            if lname == "layer1" and not (sh and self.fold_point_feat):                          # ... or as a pass of its own
                _lib.check((lib.omni_add_period_sh if sh else lib.omni_add_period_f32)(
                    _p(x), _p(point_feat), ctypes.c_size_t(x.numel()), ctypes.c_size_t(point_feat.numel()), self._s), "add point_feat")
            feats[lname] = x
        layer1, layer2, layer3, layer4 = (feats[k] for k in ("layer1", "layer2", "layer3", "layer4"))
        # ---- transformer over the N tokens of each panorama (:263-268)
        if 32 * P32 * P32 != 512:
            raise RuntimeError(f"patch size {P}: token dim {32 * P32 * P32} != 512 — the reference network only exists at "
                               "patch size 128 (SURVEY.md finding 0.1)")
        d = self._conv(layer4, "down", M, P32, P32, 512, 32, 1, 1, 0, ACT_NONE, out_f32=True)
        tok = new(M, 512)
        _lib.check(lib.omni_token_pack_f32(_p(d), _p(self.w["pos"]), _p(tok), M, N, P32 * P32, 32, self._s), "token_pack")
        normed = None                                                # LayerNorm(tok) for the NEXT consumer, when fc2's second pass has already made it
        pending = None                                               # a lone panorama: fc2's K slices (parts, slices, bias key, residual) — the next consumer adds them up
        for i in range(6):
            t = f"t{i}."
            if sh:                                                   # LN / attention emit SH, the GEMMs run f16x3 from it
                if pending is not None:
                    tok, qkv = self._ln_gemm_parts(pending, t + "norm1.weight", t + "norm1.bias", 1e-5, t + "attn.qkv.w16", M, 1536)
                    pending = None
                elif normed is not None:
                    qkv = self._gemm_sh(normed, t + "attn.qkv.w16", None, M, 512, 1536)
                else:
                    qkv = self._ln_gemm_sh(tok, t + "norm1.weight", t + "norm1.bias", 1e-5, t + "attn.qkv.w16", None, M, 1536)
                att = new(M, 512)
                _lib.check(lib.omni_attention_qkv_sh(_p(qkv), _p(att), bs, N, self._s), "attention")
                tok = self._gemm_sh(att, t + "attn.proj.w16", t + "attn.proj.bias", M, 512, 512, res=tok)
                h = self._ln_gemm_sh(tok, t + "norm2.weight", t + "norm2.bias", 1e-5, t + "mlp.fc1.w16", t + "mlp.fc1.bias", M, 2048, act=ACT_GELU, out_sh=True)
                nxt = (f"t{i + 1}.norm1.weight", f"t{i + 1}.norm1.bias", 1e-5, True) if i < 5 else ("enc_norm.w", "enc_norm.b", 1e-6, False)
                tok, normed, pending = self._fc2_ln(h, t + "mlp.fc2.w16", t + "mlp.fc2.bias", M, tok, nxt)
                continue
            y = self._ln(tok, t + "norm1.weight", t + "norm1.bias", M, 1e-5)
            q = self._gemm(y, t + "attn.q.weight", None, M, 512, 512)
            kv = self._gemm(y, t + "attn.kv.weight", None, M, 512, 1024)
            att = new(M, 512)
            _lib.check(lib.omni_attention_f32(_p(q), _p(kv), _p(att), bs, N, self._s), "attention")
            tok = self._gemm(att, t + "attn.proj.weight", t + "attn.proj.bias", M, 512, 512, res=tok)
            y = self._ln(tok, t + "norm2.weight", t + "norm2.bias", M, 1e-5)
            h = self._gemm(y, t + "mlp.fc1.weight", t + "mlp.fc1.bias", M, 512, 2048, act=ACT_GELU)
            tok = self._gemm(h, t + "mlp.fc2.weight", t + "mlp.fc2.bias", M, 2048, 512, res=tok)
        if sh and pending is not None:                                 # the last fc2's K slices: sum + bias + residual + encoder_norm in one kernel
            parts, S, bkey, res_tok = pending
            tok_sum, tok = new(M, 512), new(M, 512)
            _lib.check(lib.omni_splitk_reduce_ln512(_p(parts), S, _p(self.w[bkey]), _p(res_tok), _p(tok_sum), _p(self.w["enc_norm.w"]), _p(self.w["enc_norm.b"]),
                                                    ctypes.c_float(1e-6), _p(tok), 0, M, self._s), "reduce+ln")
        else:
            tok = normed if (sh and normed is not None) else self._ln(tok, "enc_norm.w", "enc_norm.b", M, 1e-6)
        _lib.check((lib.omni_add_hw_sh if sh else lib.omni_add_hw_f32)(_p(layer4), _p(tok), M, P32 * P32, 512, self._s), "token bias")
        # ---- decoder (:270-302); torch.cat is the two-source form of the conv
        up = self._up(layer4, M, P32, P32, 512, P16, P16)
        x = self._conv(up, "de_conv0_0", M, P16, P16, 512, 256, 3, 1, 1, ACT_RELU)
        x = self._conv(x, "de_conv0_1", M, P16, P16, 256, 128, 3, 1, 1, ACT_RELU, x2=layer3, C2=256)
        up = self._up(x, M, P16, P16, 128, P8, P8)
        x = self._conv(up, "de_conv1_0", M, P8, P8, 128, 128, 3, 1, 1, ACT_RELU)
        x = self._conv(x, "de_conv1_1", M, P8, P8, 128, 64, 3, 1, 1, ACT_RELU, x2=layer2, C2=128)
        x = self._up_conv(x, "de_conv2_0", M, P8, P8, 64, 64, ACT_RELU)
        x = self._conv(x, "de_conv2_1", M, P4, P4, 64, 64, 3, 1, 1, ACT_RELU, x2=layer1, C2=64)
        # the two widest stages, optionally a few panoramas at a time (Engine.tail_chunk; no operator here mixes patches)
        a, c = out if out is not None else (new(bs, N, 1, P, P), new(bs, N, 1, P, P) if confidence else None)
        av, cv = a.view(M, P, P), c.view(M, P, P) if c is not None else None
        fused = sh and self.fuse_heads and P % 32 == 0              # (the fused kernel up-samples inside whatever `fuse_up` says)
        x_in, de4 = x, (None if fused else new(M, P, P, 32))
        for m0, m1 in self._chunks(bs, N, self.tail_chunk):
            Mc = m1 - m0
            x = self._up_conv(x_in[m0:m1], "de_conv3_0", Mc, P4, P4, 64, 64, ACT_RELU)
            x = self._conv(x, "de_conv3_1", Mc, P2, P2, 64, 32, 3, 1, 1, ACT_RELU, x2=conv1[m0:m1], C2=64)
            if fused:                                                # de_conv4_0 + pred / weight_pred in one pass: its 302-MB output (8 panoramas) never exists
                nb = int(lib.omni_up2_heads_scratch_bytes(Mc, P))
                hs = new((nb + 3) // 4)
                _lib.check(lib.omni_conv3x3_up2_heads_sh_f16x3(_p(x), _p(self.w["de_conv4_0.w16"]), _p(self.w["de_conv4_0.b"]), _p(self.w["heads.w16f"]),
                                                               ctypes.c_float(self.head_bias[0]), ctypes.c_float(self.head_bias[1]), _p(hs), ctypes.c_size_t(nb),
                                                               _p(av[m0:m1]), _p(cv[m0:m1]) if cv is not None else None, Mc, P, 1 if confidence else 0, self._s), "up+conv+heads")
                continue
            self._up_conv(x, "de_conv4_0", Mc, P2, P2, 32, 32, ACT_RELU, out_f32=True, out=de4[m0:m1])
            _lib.check(lib.omni_heads_f32(_p(de4[m0:m1]), _p(self.w["heads.w"]), ctypes.c_float(self.head_bias[0]), ctypes.c_float(self.head_bias[1]),
                                          _p(av[m0:m1]), _p(cv[m0:m1]) if cv is not None else None, Mc, P, 1 if confidence else 0, self._s), "heads")
        self.last = {"de_conv4_0": de4, "layer4": layer4}
        return a, c

    # de_conv4_0 and the two heads in one pass (omni_conv3x3_up2_heads_sh_f16x3): the decoder's last feature map is never written.  False: the two
    # kernels (fp32 heads; `Engine.last["de_conv4_0"]` then holds that map — the G6 check-point test reads it).  The outputs agree to ~1e-6 relative.
    fuse_heads = os.environ.get("OMNI_FUSE_HEADS", "1") != "0"

    # Passes of a few panoramas through the widest stages (same kernels, same bits) were worth +1.3 % with three forwards in flight while the
    # up-sampled tensors still went through HBM; since the up-sampling is computed inside the convolution (fuse_up) they change nothing
    # (tools/chunk_ab.py: 3515 vs 3517 panoramas/s), so the default is the whole batch at once.
    tail_chunk = 0             # panoramas per pass through the decoder's two widest stages (0: the whole batch at once)
    front_chunk = 0            # ... and through stem -> max-pool

    @staticmethod
    def _chunks(bs, N, per):
        per = bs if per <= 0 else min(per, bs)
        return [(b0 * N, min(bs, b0 + per) * N) for b0 in range(0, bs, per)]

    def blend(self, a, c, erp_hw):
        P = self.patch_size
        if c is not None:
            return pers2equi_conf(a, c, self.fov, self.nrows, P, erp_hw, layout=_lib.LAYOUT_BNCHW)
        return pers2equi(a, self.fov, self.nrows, P, erp_hw, None, layout=_lib.LAYOUT_BNCHW)

    def mlp_points(self, name, xyz, depth, Mo):
        lib = _lib.load()
        P4 = self.patch_size[0] // 4
        out = torch.empty((Mo, P4, P4, 64), dtype=torch.float32, device=xyz.device)
        s = ctypes.c_void_p(torch.cuda.current_stream(xyz.device).cuda_stream)
        _lib.check(lib.omni_mlp_points_f32(_p(xyz), _p(depth), _p(self.w[name + ".w1"]), _p(self.w[name + ".b1"]),
                                           _p(self.w[name + ".w2"]), _p(self.w[name + ".b2"]), _p(out), Mo, self.npatches,
                                           P4 * P4, s), "mlp_points")
        return out

    def check_input(self, rgb):
        if not isinstance(rgb, torch.Tensor) or rgb.dim() != 4 or rgb.shape[1] != 3:
            raise ValueError("expected an RGB panorama batch [B,3,H,W]")
        if not rgb.is_cuda:
            raise ValueError("the model runs on an MI355X only (got a CPU tensor); there is no CPU path")
        if rgb.dtype != torch.float32:
            raise ValueError("float32 input expected")
        if _NPATCH[self.nrows] != self.npatches:
            raise ValueError("npatches does not match nrows")
