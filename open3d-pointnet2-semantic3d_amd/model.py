"""SSG PointNet++ semantic-segmentation network of the reference (model.py:22-148)
on the MI355X layer API: 4 x pointnet_sa_module -> 4 x pointnet_fp_module ->
conv1d(128)+BN+ReLU -> dropout(0.5) -> conv1d(num_class).

`get_sa_fp_features` is the SA+FP stack alone -- the path BASELINE.json's metric
("points/sec through SA+FP stack") is quoted on; `get_model` adds the head.
Hyper-parameter keys are the reference's semantic.json keys verbatim.
"""
import json

import torch

from .util import tf_util
from .util import pointnet_util as pu
from .util.pointnet_util import pointnet_fp_module, pointnet_sa_module
from .tf_ops.tf_interpolate import three_nn

# semantic.json:8-9,11-15,19-21,23-37 of the reference
SEMANTIC_HYPERPARAMS = {
    "num_point": 8192, "batch_size": 16, "use_color": 1,
    "optimizer": "adam", "momentum": 0.9, "learning_rate": 0.001, "decay_step": 200000, "learning_rate_decay_rate": 0.7,
    "bn_init_decay": 0.5, "bn_decay_decay_rate": 0.5, "bn_decay_clip": 0.99,
    "l1_radius": 0.5, "l1_nsample": 32, "l1_npoint": 1024,
    "l2_radius": 1.0, "l2_nsample": 32, "l2_npoint": 256,
    "l3_radius": 2.0, "l3_nsample": 32, "l3_npoint": 64,
    "l4_radius": 4.0, "l4_nsample": 32, "l4_npoint": 16,
}

SA_MLPS = ([32, 32, 64], [64, 64, 128], [128, 128, 256], [256, 256, 512])  # model.py:36-87
FP_MLPS = ([256, 256], [256, 256], [256, 128], [128, 128, 128])            # model.py:90-129


def load_hyperparams(path):
    with open(path) as f:
        return json.load(f)




def coarse_levels(l1_xyz, hyperparams):
    """SA levels 2-4 (sampling, ball query) and the 3-NN tables of FP1-FP3 from the level-1 samples in ONE launch
    (pu.coarse_geometry) -> [level 2, level 3, level 4] or None when the configuration does not fit that kernel."""
    keys = ["l%d_" % i for i in (2, 3, 4)]
    npoints = [int(hyperparams[k + "npoint"]) for k in keys]
    if not pu.coarse_geometry_fits(l1_xyz.shape[1], npoints):
        return None
    return pu.coarse_geometry(l1_xyz, npoints, [hyperparams[k + "radius"] for k in keys],
                              [int(hyperparams[k + "nsample"]) for k in keys])


def compute_geometry(l0_xyz, hyperparams, plans=False):
    """The weight-independent half of the whole stack for one batch: FPS + gather + ball query of the four SA levels and
    three_nn of the four FP levels (coordinates only, HIP index kernels, no autograd).  A trainer runs it for batch k+1
    on a side stream while batch k trains -- the 0.7 ms chain of dependent FPS rounds (16 of 256 CUs) then sits beside
    the dense work instead of in front of it.  plans=True (training) also builds the scatter lists of the backward pass
    (pu.scatter_plan: group_point's and three_interpolate's gradients as gathers; they depend on idx / dist only).
    -> {"xyzs": [5], "idxs": [4], "nn": [4 x (dist, idx)], "gplans": [4], "iplans": [4]}, coarse to fine; a plan is None
    where the gradient kernel cannot use one (feature width not a multiple of 4) or plans=False."""
    with torch.no_grad():
        xyzs, idxs = [l0_xyz.contiguous()], []
        new_xyz, idx = pu.sa_geometry(xyzs[0], hyperparams["l1_npoint"], hyperparams["l1_radius"], hyperparams["l1_nsample"])
        xyzs.append(new_xyz)
        idxs.append(idx)
        coarse = coarse_levels(new_xyz, hyperparams)  # levels 2-4 and the 3-NN tables of FP1-FP3 in one launch
        if coarse is not None:
            xyzs += [lv["new_xyz"] for lv in coarse]
            idxs += [lv["idx"] for lv in coarse]
            nn = [coarse[2 - fi]["nn"] for fi in range(3)] + [three_nn(xyzs[0], xyzs[1])]
        else:
            for li in range(1, 4):
                k = "l%d_" % (li + 1)
                new_xyz, idx = pu.sa_geometry(xyzs[-1], hyperparams[k + "npoint"], hyperparams[k + "radius"],
                                              hyperparams[k + "nsample"])
                xyzs.append(new_xyz)
                idxs.append(idx)
            nn = [three_nn(xyzs[3 - fi], xyzs[4 - fi]) for fi in range(4)]
        gplans, iplans = [None] * 4, [None] * 4
        if plans:
            specs = [None] * 8  # all of them in one memset + three launches (pu.scatter_plans)
            for li in range(4):  # level li groups the features of level li: colour (3 wide) for li = 0, else the MLP output
                width = 3 * int(hyperparams["use_color"]) if li == 0 else SA_MLPS[li - 1][-1]
                if width > 0 and width % 4 == 0:
                    specs[li] = (idxs[li], xyzs[li].shape[1], None, None)
            for fi in range(4):
                width = SA_MLPS[3][-1] if fi == 0 else FP_MLPS[fi - 1][-1]
                if width % 4 == 0:
                    specs[4 + fi] = (nn[fi][1], xyzs[4 - fi].shape[1], nn[fi][0], 2)
            built = pu.scatter_plans(specs)
            gplans, iplans = built[:4], built[4:]
    return {"xyzs": xyzs, "idxs": idxs, "nn": nn, "gplans": gplans, "iplans": iplans}


def geometry_tensors(geo):
    """flat list of the tensors of a geometry dict (fixed order): for copies between static graph buffers"""
    plans = [t for t in geo.get("gplans", []) + geo.get("iplans", []) if t is not None]
    return geo["xyzs"][1:] + geo["idxs"] + [t for pair in geo["nn"] for t in pair] + plans


def clone_geometry(geo, l0_xyz):
    """a copy of a geometry dict in fresh buffers (the static buffers a captured graph reads), level 0 = l0_xyz"""
    cl = lambda t: None if t is None else t.clone()
    return {"xyzs": [l0_xyz] + [t.clone() for t in geo["xyzs"][1:]], "idxs": [t.clone() for t in geo["idxs"]],
            "nn": [(d.clone(), i.clone()) for d, i in geo["nn"]],
            "gplans": [cl(t) for t in geo.get("gplans", [None] * 4)], "iplans": [cl(t) for t in geo.get("iplans", [None] * 4)]}


def sa1_samples(point_cloud, hyperparams, bins=True):
    """The first dependent chain of a batch, alone: farthest point sampling + gather of SA level 1 (pointnet_util.py:36-37 as
    model.py:36-47 calls it) -> (l0_xyz (B,N,3): a VIEW of the batch when it can be read in place, new_xyz (B, l1_npoint, 3) carrying
    the run's tie record, bins).  bins (bins=True and a level the LDS-grid ball query takes): the cloud sorted into the grid of
    SA1's radius ONCE (tf_grouping.ball_query_bin: one workgroup per cloud, like the sampler, and it depends on the input only),
    so that the ball query of the dense half copies the cell-sorted cloud instead of re-binning it in every workgroup
    (20.3 -> 15.6 us of chip-filling time; the 10.6 us of binning ride in the sampler half, 16 CUs).  None otherwise.
    get_sa_fp_features(..., sa1=) takes it from there; runtime.StaggeredPipeline captures the two halves of a batch as two graphs."""
    from .tf_ops import tf_grouping
    from .tf_ops.tf_sampling import farthest_point_sample_and_gather
    with torch.no_grad():
        l0_xyz = point_cloud[:, :, 0:3]  # a view: pn2_fps_nested_ld reads the xyz columns of the batch in place
        if not (point_cloud.dtype == torch.float32 and point_cloud.is_contiguous()):
            l0_xyz = l0_xyz.contiguous()
        new_xyz = farthest_point_sample_and_gather(int(hyperparams["l1_npoint"]), l0_xyz)[1]
        made = None
        if (bins and l0_xyz.dtype == torch.float32 and tf_grouping.BIN_MIN_N <= l0_xyz.shape[1] <= tf_grouping.BIN_MAX_N
                and int(hyperparams["l1_npoint"]) >= tf_grouping.BIN_MIN_M
                and int(hyperparams["l1_nsample"]) <= tf_grouping.BIN_MAX_NSAMPLE):
            made = tf_grouping.ball_query_bin(float(hyperparams["l1_radius"]), l0_xyz)
        return l0_xyz, new_xyz, made


def get_sa_fp_features(point_cloud, is_training, hyperparams, bn_decay=None, geometry=None,
                       head_width=0, cut_sa_fp=False, sa1=None):
    """point_cloud (B,N,3 or 6) -> l0_points (B,N,128) and end_points.
    geometry (extension): compute_geometry(l0_xyz) of this very batch, computed ahead.
    sa1 (extension, inference): sa1_samples(point_cloud) of this very batch, computed ahead (everything else is done here).
    head_width (extension, training; get_model passes 128): the result goes to ONE batch-normalised layer of that width and
    nowhere else, so the last FP layer may hand over its un-normalised output (pointnet_fp_module defer_last_bn).
    cut_sa_fp (extension, training): the FP modules read DETACHED copies of the SA outputs (end_points["sa_features_cut"],
    requiring grad), so the backward pass can be run in two pieces -- loss -> head -> FP -> the cut, then the cut -> SA --
    with the gradient all-reduce of the first piece's parameters travelling during the second (train.Trainer)."""
    end_points = {}
    if sa1 is not None and (is_training or geometry is not None):
        # sa1[0] may be a strided view of the (b,n,6) batch: only the inference path reads rows where they lie
        raise ValueError("sa1= is an inference-only hand-over: not with is_training=True or geometry=")
    if is_training:
        tf_util.reset_bn_links()  # producer records of the previous forward pass (tf_util._TrainDenseBnRelu)
    # model.py:26-29 slices the batch into coordinates and colours.  Inference: the slices stay VIEWS of point_cloud -- the
    # sampler, SA1's ball query and fused MLP, FP4's three_nn and chain read their column block where it lies (the *_ld entry
    # points), no copy kernel runs.  Training: dense copies (the training kernels read dense rows).
    in_place = not is_training and point_cloud.dtype == torch.float32 and point_cloud.is_contiguous() and geometry is None
    dense = (lambda t: t) if in_place else (lambda t: t.contiguous())
    if hyperparams["use_color"]:
        feature_size = 3 * int(hyperparams["use_color"])
        l0_xyz = sa1[0] if sa1 is not None else dense(point_cloud[:, :, 0:3])
        l0_points = dense(point_cloud[:, :, 3:3 + feature_size])
    else:
        l0_xyz = sa1[0] if sa1 is not None else point_cloud.contiguous()
        l0_points = None
    end_points["l0_xyz"] = l0_xyz
    xyzs, feats = [l0_xyz], [l0_points]
    coarse = None  # geometry not computed ahead: levels 2-4 + the 3-NN tables of FP1-FP3 in one launch after SA1's sampling
    for li in range(4):
        k = "l%d_" % (li + 1)
        if geometry is not None:
            geo = (geometry["xyzs"][li + 1], geometry["idxs"][li], geometry.get("gplans", [None] * 4)[li])
        elif li >= 1 and coarse is not None:
            geo = (coarse[li - 1]["new_xyz"], coarse[li - 1]["idx"])
        elif li == 0 and sa1 is not None and not is_training:
            geo = (sa1[1], None, sa1[2] if len(sa1) > 2 else None)  # (samples, ball query still to do, the cloud's bins or None)
        else:
            geo = None
        new_xyz, new_points, _ = pointnet_sa_module(
            xyzs[-1], feats[-1], npoint=hyperparams[k + "npoint"], radius=hyperparams[k + "radius"],
            nsample=hyperparams[k + "nsample"], mlp=list(SA_MLPS[li]), mlp2=None, group_all=False,
            is_training=is_training, bn_decay=bn_decay, scope="layer%d" % (li + 1), geometry=geo)
        if li == 0 and geometry is None:
            coarse = coarse_levels(new_xyz, hyperparams)
        xyzs.append(new_xyz)
        feats.append(new_points)

    # feature propagation, coarse to fine (model.py:90-129)
    fp_in = feats
    if cut_sa_fp and is_training:
        fp_in = [feats[0]] + [f.detach().requires_grad_(True) for f in feats[1:]]
        end_points["sa_features_cut"] = fp_in[1:]
    up = fp_in[4]
    for fi in range(4):
        lvl = 3 - fi  # target level: 3,2,1,0
        up = pointnet_fp_module(xyzs[lvl], xyzs[lvl + 1], fp_in[lvl], up, list(FP_MLPS[fi]), is_training, bn_decay,
                                scope="fa_layer%d" % (fi + 1),
                                nn=(tuple(geometry["nn"][fi]) + (geometry.get("iplans", [None] * 4)[fi],) if geometry is not None
                                    else (tuple(coarse[2 - fi]["nn"]) + (None,) if (coarse is not None and fi < 3) else None)),
                                defer_last_bn=head_width if (fi == 3 and is_training) else 0)
    end_points["xyzs"] = xyzs
    end_points["sa_features"] = feats  # [l0 .. l4] point features of the SA levels (extension: hooks of the trainer)
    return up, end_points


def get_head(l0_points, is_training, num_class, bn_decay=None, end_points=None):
    """The classification head on per-point features (B,N,128) (model.py:131-146): conv1d(128)+BN+ReLU -> dropout(0.5)
    -> conv1d(num_class), scopes fc1 / dp1 / fc2.  -> logits (B,N,num_class)."""
    net = tf_util.conv1d(l0_points, 128, 1, padding="VALID", bn=True, is_training=is_training, scope="fc1",
                         bn_decay=bn_decay)
    if end_points is not None:
        end_points["feats"] = net
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope="dp1")
    return tf_util.conv1d(net, num_class, 1, padding="VALID", activation_fn=None, scope="fc2", is_training=is_training)


def get_model(point_cloud, is_training, num_class, hyperparams, bn_decay=None, geometry=None, cut_sa_fp=False):
    """-> logits (B,N,num_class), end_points (model.py:22-148).  cut_sa_fp: see get_sa_fp_features."""
    l0_points, end_points = get_sa_fp_features(point_cloud, is_training, hyperparams, bn_decay, geometry=geometry,
                                               head_width=128 if is_training else 0, cut_sa_fp=cut_sa_fp)
    return get_head(l0_points, is_training, num_class, bn_decay, end_points), end_points


class _WeightedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred2d, label, w):
        from ._lib import check, lib, ptr, stream_ptr
        rows, c = pred2d.shape
        lse = torch.empty((rows,), dtype=torch.float32, device=pred2d.device)
        acc = torch.empty((2,), dtype=torch.float64, device=pred2d.device)
        loss = torch.empty((), dtype=torch.float32, device=pred2d.device)
        with torch.cuda.device(pred2d.device):
            check(lib.pn2_weighted_ce_forward(rows, c, ptr(pred2d), ptr(label), int(label.dtype == torch.int64), ptr(w),
                                              ptr(lse), ptr(acc), ptr(loss), stream_ptr()), "pn2_weighted_ce_forward")
        ctx.save_for_backward(pred2d, label, w, lse, acc)
        return loss

    @staticmethod
    def backward(ctx, gout):
        from ._lib import check, lib, ptr, stream_ptr
        pred2d, label, w, lse, acc = ctx.saved_tensors
        rows, c = pred2d.shape
        d = torch.empty_like(pred2d)
        g = gout.contiguous().float()
        with torch.cuda.device(pred2d.device):
            check(lib.pn2_weighted_ce_backward(rows, c, ptr(pred2d), ptr(label), int(label.dtype == torch.int64), ptr(w),
                                               ptr(lse), ptr(acc), ptr(g), ptr(d), stream_ptr()), "pn2_weighted_ce_backward")
        return d, None, None


def get_loss(pred, label, smpw, end_points=None):
    """Weighted sparse softmax cross-entropy, tf.losses reduction SUM_BY_NONZERO_WEIGHTS
    (model.py:152-161): sum(w * ce) / count(w != 0), on pn2_weighted_ce_forward / _backward (one pass each, no host
    synchronisation).  `end_points` is accepted and unused, as in the reference."""
    from ._lib import require_cuda
    require_cuda(pred, label, smpw)
    if label.dtype not in (torch.int32, torch.int64):
        label = label.long()
    pred2d = pred.reshape(-1, pred.shape[-1]).contiguous().float()
    return _WeightedCE.apply(pred2d, label.reshape(-1).contiguous(), smpw.reshape(-1).contiguous().float())
