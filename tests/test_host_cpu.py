"""Host-side layer logic that needs no GPU: argument validation and the refusal to run anything off the device."""
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


def test_training_layers_have_no_torch_fallback(pn2):
    """the training path's dense layers run on the HIP library only -- a CPU tensor (or a configuration
    the kernels do not cover) is an error, not a library GEMM / F.batch_norm composition.  (The plain-torch reference of
    the layer lives in tests/torch_layers.py; the pool == max-over-K equivalence is checked on the GPU,
    tests/test_layers_gpu.py::test_train_layer_hip_equals_torch.)"""
    tfu = pn2.util.tf_util
    tfu.set_default_store(tfu.VariableStore(device=torch.device("cpu"), seed=3))
    with pytest.raises((ValueError, RuntimeError, TypeError)):
        tfu.conv2d(torch.randn(2, 5, 8, 7), 16, [1, 1], padding="VALID", stride=[1, 1], bn=True, is_training=True, scope="c",
                   bn_decay=0.7, pool=8)
    with pytest.raises((ValueError, RuntimeError, TypeError)):
        tfu.conv1d(torch.randn(2, 5, 7), 9, 1, padding="VALID", activation_fn=None, is_training=True, scope="d")
    import ast
    tree = ast.parse(open(tfu.__file__).read())  # code only (docstrings describe the maths with `x2d @ w`)
    assert not any(isinstance(n, ast.MatMult) for n in ast.walk(tree)), "a torch matmul lives in util/tf_util.py"
    names = {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute)} | {n.id for n in ast.walk(tree) if isinstance(n, ast.Name)}
    assert "batch_norm" not in names and not any(nm.startswith("USE_HIP_") for nm in names)


def test_training_layer_rejects_a_pool_that_does_not_divide(pn2):
    tfu = pn2.util.tf_util
    with pytest.raises(ValueError):
        tfu._train_layer(torch.randn(2, 5, 8, 7), torch.randn(7, 16), torch.zeros(16), None, None, True, pool=3)


def test_layer_api_refuses_cpu_tensors(pn2):
    """The SA / FP modules are HIP-only: a CPU tensor is an error, never a silent torch fallback."""
    with pytest.raises((ValueError, RuntimeError, TypeError)):
        pn2.util.pointnet_util.pointnet_sa_module(torch.randn(1, 64, 3), None, 16, 0.5, 8, [16], None, False, False, None, "s")
