from . import tf_util, pointnet_util  # noqa: F401
