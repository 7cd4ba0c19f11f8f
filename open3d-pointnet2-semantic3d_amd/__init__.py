"""MI355X-native PointNet++ SA/FP stack (drop-in for the tf_ops + util/pointnet_util
path of isl-org/Open3D-PointNet2-Semantic3D).  Importing this package loads
libpn2_hip.so; there is no CPU fallback."""
from . import config  # noqa: F401
from . import _lib  # noqa: F401  (fails loudly if the HIP extension is missing)
from . import tf_ops, util  # noqa: F401
from . import model, runtime, dist, train, dataset, downsample  # noqa: F401
from .tf_ops.tf_sampling import farthest_point_sample, gather_point, prob_sample  # noqa: F401
from .tf_ops.tf_grouping import query_ball_point, group_point, knn_point, select_top_k  # noqa: F401
from .tf_ops.tf_interpolate import three_nn, three_interpolate, interpolate_label_with_color  # noqa: F401
from .util.pointnet_util import (sample_and_group, sample_and_group_all, pointnet_sa_module,  # noqa: F401
                                 pointnet_sa_module_msg, pointnet_fp_module)

__all__ = ["farthest_point_sample", "gather_point", "query_ball_point", "group_point", "three_nn",
           "three_interpolate", "interpolate_label_with_color", "sample_and_group", "sample_and_group_all", "pointnet_sa_module",
           "pointnet_sa_module_msg", "pointnet_fp_module", "model", "config", "tf_ops", "util"]
