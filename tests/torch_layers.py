"""TEST INFRASTRUCTURE: plain-PyTorch (library GEMM / F.batch_norm) versions of the training path's dense layer, used by
the GPU tests as the fp32 reference of the HIP layer (`tf_util._train_layer`).  Lived behind `USE_HIP_* = False` switches
inside the product package until round 3: the package itself has no torch / hipBLASLt compute path."""
import torch
import torch.nn.functional as F

BN_EPSILON = 1e-3  # util/tf_util.py (tf_util.py:571-581 of the reference)


def batch_norm_eval(x, bnv):
    beta, gamma, mean, var = bnv
    return (x - mean) / torch.sqrt(var + BN_EPSILON) * gamma + beta


def batch_norm_train(x, bnv, bn_decay):
    beta, gamma, mean, var = bnv
    decay = 0.9 if bn_decay is None else float(bn_decay)  # tf_util.py:571
    c = x.shape[-1]
    y = F.batch_norm(x.reshape(-1, c), mean, var, gamma, beta, training=True, momentum=1.0 - decay, eps=BN_EPSILON)
    return y.reshape(x.shape)


def train_layer_torch(inputs, w2d, b, bnv, bn_decay, relu, pool=0, defer=False):  # defer: an optimisation hint of the HIP layer, nothing to do here
    """same signature and semantics as tf_util._train_layer, all on torch ops (autograd gives the gradients)"""
    cout = w2d.shape[1]
    pool = int(pool) if pool and pool > 1 else 0
    lead = list(inputs.shape[:-1])
    if pool:
        lead[-1] //= pool
    y = inputs @ w2d + b
    if bnv is not None:
        y = batch_norm_train(y, bnv, bn_decay)
    if relu:
        y = torch.relu(y)
    if pool:
        y = y.reshape(lead + [pool, cout]).amax(dim=-2)
    return y
