"""Full-model CPU oracle — TEST INFRASTRUCTURE ONLY (never imported by omnifusion_amd/).

A plain PyTorch fp32 restatement of the reference's two forwards, written functionally over a
state_dict with the reference schema:

  single pass   /root/reference/model/spherical_model.py:238-314
  iterative     /root/reference/model/spherical_model_iterative.py:308-456
  transformer   /root/reference/model/blocks.py:33-89, model/spherical_model.py:169-187

The reference runs a ResNet-34 as Conv3d with (k,k,1) kernels over a [B,C,P,P,N] tensor
(spherical_model.py:122-167); mathematically that is the 2-D network applied independently to the
B*N patches, which is how it is written here ([B*N,C,P,P]).  equi2pers / pers2equi come from the C
oracle (oracle/c_oracle.py).  Pinned against the reference itself by oracle/gen_golden_model.py
(fixtures G6/G7 under tests/golden/) and tests/test_model_oracle.py.

`fp16=True` mimics the storage precision of the HIP path (weights and every stored activation
rounded to fp16, fp32 accumulation) — used only to size the fp16-vs-fp32 error budget.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import c_oracle as co


def _strip(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


class _Q:
    """optional fp16 storage emulation"""
    def __init__(self, fp16):
        self.fp16 = fp16

    def __call__(self, x):
        return x.half().float() if self.fp16 else x


def _conv(sd, name, x, stride, pad, q):
    w = sd[name + ".weight"]
    if w.dim() == 5:
        w = w[..., 0]
    b = sd.get(name + ".bias")
    return F.conv2d(x, q(w), b, stride=stride, padding=pad)


def _bn(sd, name, x, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], False, 0.0, eps)


def _conv_bn(sd, conv, bn, x, stride, pad, q, relu):
    """conv + eval-mode BN (+ReLU).  With fp16 emulation BN is folded into the weights first,
    like the HIP path does at load time."""
    if q.fp16:
        w = sd[conv + ".weight"][..., 0]
        g = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-5)
        y = F.conv2d(x, q(w * g[:, None, None, None]), sd[bn + ".bias"] - sd[bn + ".running_mean"] * g,
                     stride=stride, padding=pad)
    else:
        y = _bn(sd, bn, _conv(sd, conv, x, stride, pad, q))
    return F.relu(y) if relu else y


def _basic_block(sd, p, x, stride, q):
    # torchvision BasicBlock as rewritten to Conv3d/BatchNorm3d by convert_conv/convert_bn (:122-167)
    out = q(_conv_bn(sd, p + ".conv1", p + ".bn1", x, stride, 1, q, True))
    out = _conv_bn(sd, p + ".conv2", p + ".bn2", out, 1, 1, q, False)
    if (p + ".downsample.0.weight") in sd:
        x = _conv_bn(sd, p + ".downsample.0", p + ".downsample.1", x, stride, 0, q, False)
    return q(F.relu(out + x))


def _layer(sd, name, x, nblk, stride, q):
    for b in range(nblk):
        x = _basic_block(sd, f"{name}.{b}", x, stride if b == 0 else 1, q)
    return x


def _attention(sd, p, x):
    # blocks.py:50-66 — q and kv Linear without bias, 4 heads x 128, softmax over the N tokens
    B, N, C = x.shape
    h = 4
    qv = F.linear(x, sd[p + ".q.weight"]).reshape(B, N, h, C // h).permute(0, 2, 1, 3)
    kv = F.linear(x, sd[p + ".kv.weight"]).reshape(B, N, 2, h, C // h).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (qv @ k.transpose(-2, -1)) * ((C // h) ** -0.5)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def _transformer(sd, x):
    # spherical_model.py:180-187, blocks.py:85-88 (pre-LN, eps 1e-5 inside blocks, 1e-6 for encoder_norm)
    x = x + sd["transformer.pos_emb"]
    for i in range(6):
        p = f"transformer.layer.{i}"
        y = F.layer_norm(x, (512,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
        x = x + _attention(sd, p + ".attn", y)
        y = F.layer_norm(x, (512,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
        y = F.linear(F.gelu(F.linear(y, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])),
                     sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
        x = x + y
    return F.layer_norm(x, (512,), sd["transformer.encoder_norm.weight"], sd["transformer.encoder_norm.bias"], 1e-6)


def _mlp_points(sd, name, x):
    # spherical_model.py:228-235: 1x1 conv -> BN -> ReLU, twice (plain BatchNorm2d, eps 1e-5)
    x = F.relu(_bn(sd, name + ".1", F.conv2d(x, sd[name + ".0.weight"])))
    return F.relu(_bn(sd, name + ".4", F.conv2d(x, sd[name + ".3.weight"])))


def _up(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)   # :271,279,286,293,300 (q9)


def _network(sd, patches, point_feat, bs, n_patch, P, q, down, taps=None):
    """patches [B*N,3,P,P], point_feat [B*N or N,64,P/4,P/4] -> (pred [B*N,1,P,P] after ReLU,
    weight [B*N,1,P,P] after sigmoid).  spherical_model.py:254-306."""
    conv1 = q(_conv_bn(sd, "conv1", "bn1", q(patches), 2, 3, q, True))                 # :254
    pool = F.max_pool2d(conv1, 3, 2, 1)                                                 # :255
    layer1 = _layer(sd, "layer1", pool, 3, 1, q)                                        # :257
    if point_feat.shape[0] == n_patch and bs > 1:
        point_feat = point_feat.repeat(bs, 1, 1, 1)
    layer1 = q(layer1 + point_feat)                                                     # :258
    layer2 = _layer(sd, "layer2", layer1, 4, 2, q)
    layer3 = _layer(sd, "layer3", layer2, 6, 2, q)
    layer4 = _layer(sd, "layer4", layer3, 3, 2, q)
    s4 = layer4.shape[-1]
    # :263-268  down -> [bs, 32*s4*s4, N] -> tokens [bs, N, 512] -> transformer -> per-channel bias
    d = _conv(sd, down, layer4, 1, 0, q)                                                # [B*N,32,s4,s4]
    tok = d.reshape(bs, n_patch, -1)                                                    # token dim = c*s4*s4 + h*s4 + w
    if tok.shape[-1] != 512:
        raise RuntimeError(f"token dim {tok.shape[-1]} != 512: the reference only runs at patch size 128 (SURVEY 0.1)")
    tok = _transformer(sd, tok)
    layer4 = q(layer4 + tok.reshape(bs * n_patch, 512, 1, 1))
    if taps is not None:
        taps["layer4"] = layer4
    up = _up(layer4, layer3.shape[-2:])
    x = q(_conv_bn(sd, "de_conv0_0.conv", "de_conv0_0.bn", q(up), 1, 1, q, True))
    x = q(_conv_bn(sd, "de_conv0_1.conv", "de_conv0_1.bn", torch.cat([x, layer3], 1), 1, 1, q, True))
    up = _up(x, layer2.shape[-2:])
    x = q(_conv_bn(sd, "de_conv1_0.conv", "de_conv1_0.bn", q(up), 1, 1, q, True))
    x = q(_conv_bn(sd, "de_conv1_1.conv", "de_conv1_1.bn", torch.cat([x, layer2], 1), 1, 1, q, True))
    up = _up(x, layer1.shape[-2:])
    x = q(_conv_bn(sd, "de_conv2_0.conv", "de_conv2_0.bn", q(up), 1, 1, q, True))
    x = q(_conv_bn(sd, "de_conv2_1.conv", "de_conv2_1.bn", torch.cat([x, layer1], 1), 1, 1, q, True))
    up = _up(x, conv1.shape[-2:])
    x = q(_conv_bn(sd, "de_conv3_0.conv", "de_conv3_0.bn", q(up), 1, 1, q, True))
    x = q(_conv_bn(sd, "de_conv3_1.conv", "de_conv3_1.bn", torch.cat([x, conv1], 1), 1, 1, q, True))
    up = _up(x, (P, P))
    x = q(_conv_bn(sd, "de_conv4_0.conv", "de_conv4_0.bn", q(up), 1, 1, q, True))
    if taps is not None:
        taps["de_conv4_0"] = x
    pred = F.relu(_conv(sd, "pred", x, 1, 1, q))                                        # :304
    weight = torch.sigmoid(_conv(sd, "weight_pred", x, 1, 1, q))                        # :306
    return pred, weight


def _to_planar(pers):
    """reference layout [B,C,P,P,N] (numpy) -> [B*N,C,P,P] torch"""
    B, C, P, _, N = pers.shape
    return torch.from_numpy(np.ascontiguousarray(pers.transpose(0, 4, 1, 2, 3))).reshape(B * N, C, P, P)


def _to_ref_layout(x, bs, n_patch):
    """[B*N,1,P,P] torch -> [B,1,P,P,N] numpy"""
    P = x.shape[-1]
    return np.ascontiguousarray(x.reshape(bs, n_patch, 1, P, P).permute(0, 2, 3, 4, 1).numpy())


def _blend(pred, weight, confidence, bs, n_patch, fov, nrows, P, erp_hw):
    if confidence:                                                                      # :307-311
        out = co.pers2equi_conf(_to_ref_layout(pred * weight, bs, n_patch), _to_ref_layout(weight, bs, n_patch),
                                fov, nrows, (P, P), erp_hw)
    else:                                                                               # :313
        out = co.pers2equi(_to_ref_layout(pred, bs, n_patch), fov, nrows, (P, P), erp_hw)
    return torch.from_numpy(out)


@torch.no_grad()
def spherical_fusion_forward(state_dict, rgb, nrows=4, patch_size=128, fov=(80, 80), confidence=True,
                             fp16=False, taps=None):
    """model/spherical_model.py:238-314.  rgb: torch float32 [B,3,H,W] (CPU) -> [B,1,H,W]."""
    sd = _strip(state_dict); q = _Q(fp16)
    bs, _, H, W = rgb.shape
    P = patch_size
    pers, _, _, _ = co.equi2pers(rgb.numpy(), fov, nrows, (P, P))                       # :243
    _, _, uv, center_p = co.equi2pers(rgb.numpy()[:, :, :2, :2], fov, nrows, (P // 4, P // 4), want_pers=False)   # :244
    n_patch = pers.shape[-1]
    cp = torch.from_numpy(center_p).reshape(-1, 2, 1, 1).repeat(1, 1, P // 4, P // 4)   # :247-248
    rho = torch.ones((n_patch, 1, P // 4, P // 4))
    point_feat = _mlp_points(sd, "mlp_points", torch.cat([cp, rho, cp], 1))             # :250-251
    pred, weight = _network(sd, _to_planar(pers), point_feat, bs, n_patch, P, q, "down", taps)
    if taps is not None:
        taps["pred"] = pred; taps["weight"] = weight
    return _blend(pred, weight, confidence, bs, n_patch, fov, nrows, P, (H, W))


@torch.no_grad()
def spherical_fusion_iterative_forward(state_dict, rgb, iters, nrows=4, patch_size=128, fov=(80, 80),
                                       confidence=False, fp16=False):
    """model/spherical_model_iterative.py:308-456.  Returns the list of `iters` ERP depth maps."""
    sd = _strip(state_dict); q = _Q(fp16)
    bs, _, H, W = rgb.shape
    P = patch_size
    pers, _, _, _ = co.equi2pers(rgb.numpy(), fov, nrows, (P, P))                       # :315 (and :384, identical)
    _, xyz, _, _ = co.equi2pers(rgb.numpy()[:, :, :2, :2], fov, nrows, (P // 4, P // 4), want_pers=False)   # :316
    n_patch = pers.shape[-1]
    patches = _to_planar(pers)
    xyz_t = torch.from_numpy(xyz)                                                       # [N,3,P/4,P/4]
    point_feat = _mlp_points(sd, "mlp_points1", xyz_t)                                  # :319
    pred, weight = _network(sd, patches, point_feat, bs, n_patch, P, q, "down1")
    outs = [_blend(pred, weight, confidence, bs, n_patch, fov, nrows, P, (H, W))]       # :371-380
    for i in range(iters - 1):                                                          # :383
        dpers, _, _, _ = co.equi2pers(outs[i].numpy(), fov, nrows, (P // 4, P // 4))    # :385  [B,1,P/4,P/4,N]
        depth = _to_planar(dpers)                                                       # [B*N,1,P/4,P/4]
        hx = xyz_t.repeat(bs, 1, 1, 1) * depth                                          # :387-390
        point_feat = _mlp_points(sd, "mlp_points2", hx)                                 # :392
        pred, weight = _network(sd, patches, point_feat, bs, n_patch, P, q, "down1")
        outs.append(_blend(pred, weight, confidence, bs, n_patch, fov, nrows, P, (H, W)))
    return outs
