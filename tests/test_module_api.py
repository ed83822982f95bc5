"""CPU: spherical_fusion is an nn.Module with the reference's state_dict schema (model/spherical_model.py:190-235,
model/spherical_model_iterative.py:253-305) — what test.py:104-111 (`convert_model`, `nn.DataParallel`,
`load_state_dict`, `.cuda()`, `.eval()`) needs from it.  The forward itself needs the GPU (tests/test_model_gpu.py)."""
import json
import os

import pytest
import torch
from torch import nn

from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.model.spherical_model_iterative import spherical_fusion as spherical_fusion_it
from omnifusion_amd.weights import make_state_dict, schema

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("cls,it,name", [(spherical_fusion, False, "single"), (spherical_fusion_it, True, "iterative")])
def test_state_dict_round_trips_reference_schema(cls, it, name):
    net = cls(4, 18, (128, 128), (80, 80))
    assert isinstance(net, nn.Module) and not net.training
    ref = json.load(open(os.path.join(GOLDEN, f"state_dict_schema_{name}.json")))     # listing taken from the reference itself
    keys = list(net.state_dict().keys())
    assert keys == list(schema(18, it).keys())
    ref_keys = [k for k, _ in ref] if isinstance(ref, list) else list(ref.keys())
    assert keys == ref_keys                                                           # same names, same ORDER
    sd = make_state_dict(7, 18, it)
    out = net.load_state_dict(sd)
    assert not out.missing_keys and not out.unexpected_keys and net._loaded and net._dirty
    back = net.state_dict()
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    n_par = sum(1 for _ in net.parameters()); n_buf = sum(1 for _ in net.buffers())
    assert n_par + n_buf == len(sd) and n_buf == 3 * sum(k.endswith("running_mean") for k in sd)
    assert all(not p.requires_grad for p in net.parameters())


def test_load_through_dataparallel_wrapper_marks_dirty():
    """test.py:107-110: `network = nn.DataParallel(network); network.load_state_dict(ckpt)` — the wrapper loads
    'module.'-prefixed keys through the submodules' _load_from_state_dict, never calling our load_state_dict override;
    the post-hook must still see it."""
    net = spherical_fusion(4, 18, (128, 128), (80, 80))
    net._dirty = False
    wrapped = nn.DataParallel(net) if torch.cuda.is_available() else None
    sd = make_state_dict(3, 18, False)
    if wrapped is None:                       # DataParallel without GPUs degenerates to the bare module: emulate its load path
        class Wrap(nn.Module):
            def __init__(self, m):
                super().__init__(); self.module = m
        wrapped = Wrap(net)
    wrapped.load_state_dict({"module." + k: v for k, v in sd.items()})
    assert net._loaded and net._dirty
    assert torch.equal(net.layer3[2].conv1.weight if hasattr(net.layer3, "__getitem__") else net.state_dict()["layer3.2.conv1.weight"],
                       sd["layer3.2.conv1.weight"])
    # checkpoint saved through DataParallel loaded into the bare module (train_erp_depth.py:307)
    net2 = spherical_fusion(4, 18, (128, 128), (80, 80))
    net2.load_state_dict({"module." + k: v for k, v in sd.items()})
    assert torch.equal(net2.state_dict()["pred.bias"], sd["pred.bias"])


def test_strictness_and_modes():
    net = spherical_fusion(4, 18, (128, 128), (80, 80))
    sd = make_state_dict(3, 18, False)
    bad = dict(sd); bad.pop("layer3.2.conv1.weight")
    with pytest.raises(RuntimeError, match="layer3.2.conv1.weight"):
        net.load_state_dict(bad)
    assert not net._loaded
    r = net.load_state_dict(bad, strict=False)
    assert r.missing_keys == ["layer3.2.conv1.weight"] and not net._loaded          # an incomplete model is not runnable
    wrong = dict(sd); wrong["pred.weight"] = torch.zeros(1, 32, 3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        net.load_state_dict(wrong)
    assert net.eval() is net and not net.training
    net.train()
    assert net.training
    with pytest.raises(NotImplementedError, match="inference-only"):
        net(torch.zeros(1, 3, 32, 64))
    net.eval()
    with pytest.raises(RuntimeError, match="no weights loaded"):
        spherical_fusion()._sync_packed(torch.device("cpu"))


def test_convert_model_like_traversal_is_a_no_op():
    """sync_batchnorm.convert_model (test.py:105) walks named_children() replacing BatchNorm modules: there are none."""
    net = spherical_fusion_it(6, 46, (128, 128), (80, 80))
    assert not any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in net.modules())
    assert len(list(net.named_children())) > 10


def test_dataparallel_replica_borrows_a_device_context_and_never_the_wrapped_engine():
    """test.py:105-111 on a multi-GPU node: nn.DataParallel replicates the module per forward (`_replicate_for_data_parallel`: a shallow
    __dict__ copy with NO parameters, one thread each).  A replica must not share the wrapped module's Engine nor pack from its own
    (empty) state_dict: it points back at the wrapped module and takes its device's context from it."""
    for cls, it in ((spherical_fusion, False), (spherical_fusion_it, True)):
        net = cls(4, 18, (128, 128), (80, 80))
        net.load_state_dict(make_state_dict(3, 18, it))
        rep = net._replicate_for_data_parallel()
        assert rep.__dict__["_origin"] is net and rep._is_replica and net.__dict__["_origin"] is None
        assert "_origin" not in rep._modules and len(rep._parameters) == 0
        rep2 = rep._replicate_for_data_parallel()                  # a replica of a replica still points at the real module
        assert rep2.__dict__["_origin"] is net
        assert rep._contexts is net._contexts                      # ONE table of per-device contexts, owned by the wrapped module
        # the replica goes through _execution(): on a CPU tensor it fails like the module itself, before touching any engine
        with pytest.raises(ValueError, match="no CPU path"):
            rep(torch.zeros(1, 3, 32, 64), *((1,) if it else ()))
        eng = net._eng
        v = net._master_version
        net.load_state_dict(make_state_dict(4, 18, it))
        assert net._master_version == v + 1 and net._eng is eng    # contexts are re-packed lazily, by version
    fresh = spherical_fusion(4, 18, (128, 128), (80, 80))
    with pytest.raises(RuntimeError, match="no weights loaded"):
        fresh._device_context("cuda:0")
