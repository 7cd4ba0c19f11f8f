from . import tf_sampling, tf_grouping, tf_interpolate  # noqa: F401
