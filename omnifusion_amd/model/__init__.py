"""Mirror of the reference's `model` package (spherical_model.py / spherical_model_iterative.py)."""


def smoke_model(dev="cuda:0"):
    """One tiny forward of the flagship model on `dev`, checked against the CPU oracle (smoke())."""
    import numpy as np
    import torch
    from .spherical_model import spherical_fusion
    from ..weights import make_state_dict
    from oracle import model_ref
    sd = make_state_dict(42, 18, False)
    net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(torch.device(dev).index)
    net.load_state_dict(sd)
    rng = np.random.default_rng(0)
    rgb = torch.from_numpy(rng.random((1, 3, 32, 64), dtype=np.float32))
    got = net(rgb.to(dev), confidence=True).cpu()
    want = model_ref.spherical_fusion_forward(sd, rgb, confidence=True)
    d = (got - want).abs().max().item()
    assert d < 1e-3, f"model smoke: max |d| = {d}"
