"""State-dict schema of the reference models and a deterministic random-weight generator.

The reference ships no checkpoints (`*.pth` git-ignored) and its `torchvision.models.resnet34(
pretrained=True)` (model/spherical_model.py:197) cannot download here, so tests and bench.py use
random-init weights of the reference architecture: `make_state_dict(seed, ...)` returns tensors
named and shaped exactly like `spherical_fusion().state_dict()` of the reference
(model/spherical_model.py:190-235 / model/spherical_model_iterative.py:253-305; 363 / 375
tensors, conv weights 5-D `[O,I,k,k,1]`) — tests/test_model_oracle.py and tests/test_module_api.py check the schema (names, shapes,
ORDER) against a listing taken from the reference itself (tests/golden/state_dict_schema_*.json).

Values come from numpy's PCG64 seeded per tensor name (stable across numpy / torch versions).
Scales are He-style so activations neither vanish nor explode through the 50-odd layers, BN
running statistics are perturbed away from (0, 1) so that BN folding is exercised, and the depth
head gets a positive bias so that ReLU(pred) is not identically zero.
"""
import zlib

import numpy as np
import torch

_LAYERS = [("layer1", 64, 64, 3, 1), ("layer2", 64, 128, 4, 2), ("layer3", 128, 256, 6, 2), ("layer4", 256, 512, 3, 2)]
_DECODER = [("de_conv0_0", 512, 256), ("de_conv0_1", 512, 128), ("de_conv1_0", 128, 128), ("de_conv1_1", 256, 64),
            ("de_conv2_0", 64, 64), ("de_conv2_1", 128, 64), ("de_conv3_0", 64, 64), ("de_conv3_1", 128, 32),
            ("de_conv4_0", 32, 32)]


def _bn(schema, name, c):
    for s in ("weight", "bias", "running_mean", "running_var"):
        schema[f"{name}.{s}"] = ((c,), "float32")
    schema[f"{name}.num_batches_tracked"] = ((), "int64")


def schema(npatches=18, iterative=False):
    """Ordered {name: (shape, dtype)} of the reference state_dict."""
    s = {}
    s["conv1.weight"] = ((64, 3, 7, 7, 1), "float32")
    _bn(s, "bn1", 64)
    for lname, cin, cout, nblk, stride in _LAYERS:
        for b in range(nblk):
            ci = cin if b == 0 else cout
            s[f"{lname}.{b}.conv1.weight"] = ((cout, ci, 3, 3, 1), "float32")
            _bn(s, f"{lname}.{b}.bn1", cout)
            s[f"{lname}.{b}.conv2.weight"] = ((cout, cout, 3, 3, 1), "float32")
            _bn(s, f"{lname}.{b}.bn2", cout)
            if b == 0 and (stride != 1 or cin != cout):
                s[f"{lname}.{b}.downsample.0.weight"] = ((cout, ci, 1, 1, 1), "float32")
                _bn(s, f"{lname}.{b}.downsample.1", cout)
    down = "down1" if iterative else "down"
    s[f"{down}.weight"] = ((32, 512, 1, 1, 1), "float32")
    s[f"{down}.bias"] = ((32,), "float32")
    s["transformer.pos_emb"] = ((1, npatches, 512), "float32")
    for i in range(6):
        p = f"transformer.layer.{i}"
        s[f"{p}.norm1.weight"] = ((512,), "float32"); s[f"{p}.norm1.bias"] = ((512,), "float32")
        s[f"{p}.attn.q.weight"] = ((512, 512), "float32")
        s[f"{p}.attn.kv.weight"] = ((1024, 512), "float32")
        s[f"{p}.attn.proj.weight"] = ((512, 512), "float32"); s[f"{p}.attn.proj.bias"] = ((512,), "float32")
        s[f"{p}.norm2.weight"] = ((512,), "float32"); s[f"{p}.norm2.bias"] = ((512,), "float32")
        s[f"{p}.mlp.fc1.weight"] = ((2048, 512), "float32"); s[f"{p}.mlp.fc1.bias"] = ((2048,), "float32")
        s[f"{p}.mlp.fc2.weight"] = ((512, 2048), "float32"); s[f"{p}.mlp.fc2.bias"] = ((512,), "float32")
    s["transformer.encoder_norm.weight"] = ((512,), "float32")
    s["transformer.encoder_norm.bias"] = ((512,), "float32")
    for name, cin, cout in _DECODER:
        s[f"{name}.conv.weight"] = ((cout, cin, 3, 3, 1), "float32")
        _bn(s, f"{name}.bn", cout)
    for head in ("pred", "weight_pred"):
        s[f"{head}.weight"] = ((1, 32, 3, 3, 1), "float32")
        s[f"{head}.bias"] = ((1,), "float32")
    mlps = [("mlp_points1", 3), ("mlp_points2", 3)] if iterative else [("mlp_points", 5)]
    for name, cin in mlps:
        s[f"{name}.0.weight"] = ((16, cin, 1, 1), "float32")
        _bn(s, f"{name}.1", 16)
        s[f"{name}.3.weight"] = ((64, 16, 1, 1), "float32")
        _bn(s, f"{name}.4", 64)
    return s


def _rng(seed, name):
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def make_state_dict(seed=42, npatches=18, iterative=False):
    sd = {}
    for name, (shape, dtype) in schema(npatches, iterative).items():
        r = _rng(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        if dtype == "int64":
            v = np.zeros(shape, np.int64)
        elif leaf == "running_var":
            v = r.uniform(0.6, 1.4, shape)
        elif leaf == "running_mean":
            v = r.uniform(-0.2, 0.2, shape)
        elif name == "transformer.pos_emb":
            v = r.normal(0.0, 0.02, shape)
        elif ".norm" in name or "encoder_norm" in name or ".bn" in name or name.startswith("bn1") \
                or "downsample.1" in name or (name.startswith("mlp_points") and name.split(".")[1] in ("1", "4")):
            v = r.uniform(0.8, 1.2, shape) if leaf == "weight" else r.uniform(-0.1, 0.1, shape)
        elif leaf == "bias":
            v = r.uniform(-0.05, 0.05, shape)
            if name == "pred.bias":
                v = v + 1.0                       # keeps ReLU(pred) away from the all-zero map
        else:                                     # conv / linear weights: He-style, fan_in = prod(shape[1:])
            fan_in = int(np.prod(shape[1:]))
            gain = 1.0 if (name.startswith("transformer") or name.startswith("pred") or name.startswith("weight_pred")) else 2.0
            v = r.normal(0.0, np.sqrt(gain / fan_in), shape)
            if ".conv2." in name:                 # residual branches: keep the sum from growing block after block
                v = v * 0.5
        sd[name] = torch.from_numpy(np.asarray(v)).to(torch.int64 if dtype == "int64" else torch.float32)
    return sd
