"""Host-side layer logic that needs no GPU (the training layers are a torch composition when the tensors are not on a
device; the HIP kernels behind them are covered by the -m gpu tests)."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def pn2():
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    return importlib.import_module("pn2_amd")


def test_training_conv2d_pool_argument_is_max_over_the_grouped_axis(pn2):
    """conv2d(..., is_training=True, pool=K) == conv2d(...) followed by the max over the K axis
    (pointnet_util.py:167-170), values, moving averages and gradients; same variables (same scope, same seed)."""
    tfu = pn2.util.tf_util
    x = torch.randn(2, 5, 8, 7)
    outs = []
    for pool in (8, 0):
        store = tfu.set_default_store(tfu.VariableStore(device=torch.device("cpu"), seed=3))
        xx = x.clone().requires_grad_(True)
        y = tfu.conv2d(xx, 16, [1, 1], padding="VALID", stride=[1, 1], bn=True, is_training=True, scope="c", bn_decay=0.7,
                       pool=pool)
        if not pool:
            y = y.amax(dim=2, keepdim=True)
        assert tuple(y.shape) == (2, 5, 1, 16)
        (y * torch.arange(y.numel(), dtype=torch.float32).reshape(y.shape)).sum().backward()
        outs.append((y.detach(), xx.grad, {k: v.grad.clone() for k, v in store.params.items()},
                     {k: v.clone() for k, v in store.buffers.items()}))
    (ya, ga, pa, ba), (yb, gb, pb, bb) = outs
    assert torch.equal(ya, yb) and torch.allclose(ga, gb)
    assert pa.keys() == pb.keys() and all(torch.allclose(pa[k], pb[k]) for k in pa)
    assert ba.keys() == bb.keys() and all(torch.equal(ba[k], bb[k]) for k in ba)
    assert any("moving_mean" in k for k in ba) and not torch.equal(ba[[k for k in ba if "moving_mean" in k][0]], torch.zeros(16))


def test_training_layer_rejects_a_pool_that_does_not_divide(pn2):
    tfu = pn2.util.tf_util
    with pytest.raises(ValueError):
        tfu._train_layer(torch.randn(2, 5, 8, 7), torch.randn(7, 16), torch.zeros(16), None, None, True, pool=3)


def test_layer_api_refuses_cpu_tensors(pn2):
    """The SA / FP modules are HIP-only: a CPU tensor is an error, never a silent torch fallback."""
    with pytest.raises((ValueError, RuntimeError, TypeError)):
        pn2.util.pointnet_util.pointnet_sa_module(torch.randn(1, 64, 3), None, 16, 0.5, 8, [16], None, False, False, None, "s")
