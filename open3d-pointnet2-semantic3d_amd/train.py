"""One data-parallel training step of the SSG network (BASELINE config[3]).

Restates what train.py of the reference does per batch (train.py:80-119 schedules, :381-388 Adam +
minimize, model.py:152-161 loss) on the MI355X layer API: forward with batch-statistic BatchNorm
(HIP index/gather kernels + differentiable torch layers), weighted sparse softmax cross-entropy,
backward through the HIP gradient kernels (group_point_grad, gather_point_grad,
three_interpolate_grad), ONE flat RCCL all-reduce, Adam.
"""
import torch

from . import dist as pdist
from . import model
from .util import tf_util


def learning_rate(step, batch_size, base_lr=1e-3, decay_step=200000, decay_rate=0.7, floor=1e-5):
    """tf.train.exponential_decay(staircase=True) clipped at 1e-5 (train.py:80-98)."""
    return max(base_lr * decay_rate ** ((step * batch_size) // decay_step), floor)


def bn_decay(step, batch_size, init=0.5, decay_step=200000, decay_rate=0.5, clip=0.99):
    """bn_decay = min(clip, 1 - init * rate^floor(step*B/decay_step)) (train.py:101-119)."""
    return min(clip, 1.0 - init * decay_rate ** ((step * batch_size) // decay_step))


class Trainer:
    def __init__(self, hyperparams, num_class, store=None, device="cuda"):
        self.hp = dict(hyperparams)
        self.num_class = num_class
        self.store = store or tf_util.set_default_store(tf_util.VariableStore(device=device, seed=0))
        self.step_count = 0
        self.opt = None
        self.bucket = None

    def _lazy_init(self, pc):
        # variables are created by the first forward (TF-style get_variable semantics).  That forward must leave no
        # trace: no autograd graph, and the moving averages restored (the reference applies ONE update per step,
        # train.py:381-388; the HIP BN kernel updates them through raw pointers even under no_grad).
        with torch.no_grad():
            before = {k: v.clone() for k, v in self.store.buffers.items()}  # a pre-loaded store keeps its statistics
            model.get_model(pc[:1], True, self.num_class, self.hp, bn_decay=bn_decay(0, pc.shape[0]))
            for k, v in self.store.buffers.items():
                if k in before:
                    v.copy_(before[k])
                else:
                    v.fill_(0.0 if k.endswith("moving_mean") else 1.0)  # tf_util.py:571-581 initial moving averages
        params = self.store.parameters()
        pdist.broadcast_parameters(params)
        pdist.broadcast_parameters(list(self.store.buffers.values()))
        self.opt = torch.optim.Adam(params, lr=learning_rate(0, pc.shape[0]))  # TF Adam defaults == torch defaults except eps
        for g in self.opt.param_groups:
            g["eps"] = 1e-8
        self.bucket = pdist.FlatGradAllReduce(params)

    def train_step(self, pc, labels, smpw):
        """pc (B,N,6) float32, labels (B,N) int, smpw (B,N) float32 -> loss (python float)."""
        tf_util.set_default_store(self.store)
        if self.opt is None:
            self._lazy_init(pc)
        b = pc.shape[0]
        for g in self.opt.param_groups:
            g["lr"] = learning_rate(self.step_count, b)
        self.opt.zero_grad(set_to_none=True)
        logits, _ = model.get_model(pc, True, self.num_class, self.hp, bn_decay=bn_decay(self.step_count, b))
        loss = model.get_loss(logits, labels, smpw)
        loss.backward()
        self.bucket.allreduce_()  # one 3.87 MB all-reduce (sum / world)
        self.opt.step()
        self.step_count += 1
        return float(loss.detach())
