"""numpy restatement of the reference's scene sampler and voxel down-sampling.

TEST INFRASTRUCTURE ONLY (tests/, tests/golden/make_dataset_golden.py).

  FileDataOracle     dataset/semantic_dataset.py:57-186 (SemanticFileData: x-sort, _get_fix_sized_sample_mask,
                     _center_box, _extract_z_box, sample) -- same numpy calls in the same order, so a seeded
                     np.random stream gives the reference's draws.  PINNED: tests/golden/make_dataset_golden.py lifts
                     the reference's own method bodies out of its source file (the module itself imports open3d) and
                     checks this restatement against them bit for bit; the outputs are frozen in
                     tests/golden/dataset_sampler.npz.
  voxel_down_sample  downsample.py:46-67.  The voxel grid itself is Open3D's voxel_down_sample_and_trace
                     (IntelVCL/Open3D @33e46f7, tf_ops/open3d_builder.cmake:8-9 -- fetched at build time, absent from
                     /root/reference, cannot be imported here): its published algorithm is restated -- voxel index =
                     floor((p - min_bound) / voxel_size) per axis, voxel point / colour = float64 sum of the members
                     in input order / count -- with the voxels emitted sorted by index (Open3D's order is that of a
                     std::unordered_map: unspecified).  Parity of this half is UNPINNED beyond the restatement; the
                     label rule (np.bincount(...).argmax()) is the reference's own line (:60).
"""
import numpy as np


class FileDataOracle:
    def __init__(self, points, labels, colors, box_size_x, box_size_y):
        self.box_size_x, self.box_size_y = box_size_x, box_size_y
        self.points = np.asarray(points, dtype=np.float64)
        self.labels = np.asarray(labels)
        self.colors = np.asarray(colors, dtype=np.float64)
        sort_idx = np.argsort(self.points[:, 0])  # :85
        self.points = self.points[sort_idx]
        self.labels = self.labels[sort_idx]
        self.colors = self.colors[sort_idx]

    def _get_fix_sized_sample_mask(self, points, num_points_per_sample):  # :90-107
        if len(points) - num_points_per_sample > 0:
            true_array = np.ones(num_points_per_sample, dtype=bool)
            false_array = np.zeros(len(points) - num_points_per_sample, dtype=bool)
            sample_mask = np.concatenate((true_array, false_array), axis=0)
            np.random.shuffle(sample_mask)
        else:
            sample_mask = np.arange(len(points))
            while len(sample_mask) < num_points_per_sample:
                sample_mask = np.concatenate((sample_mask, sample_mask), axis=0)
            sample_mask = sample_mask[:num_points_per_sample]
        return sample_mask

    def _center_box(self, points):  # :109-121
        box_min = np.min(points, axis=0)
        shift = np.array([box_min[0] + self.box_size_x / 2, box_min[1] + self.box_size_y / 2, box_min[2]])
        return points - shift

    def _extract_z_box(self, center_point):  # :123-163
        scene_z_size = np.max(self.points, axis=0)[2] - np.min(self.points, axis=0)[2]
        box_min = center_point - [self.box_size_x / 2, self.box_size_y / 2, scene_z_size]
        box_max = center_point + [self.box_size_x / 2, self.box_size_y / 2, scene_z_size]
        i_min = np.searchsorted(self.points[:, 0], box_min[0])
        i_max = np.searchsorted(self.points[:, 0], box_max[0])
        mask = np.sum((self.points[i_min:i_max, :] >= box_min) * (self.points[i_min:i_max, :] <= box_max), axis=1) == 3
        mask = np.hstack((np.zeros(i_min, dtype=bool), mask, np.zeros(len(self.points) - i_max, dtype=bool)))
        assert np.sum(mask) != 0
        return mask

    def sample(self, num_points_per_sample, draws=None):  # :165-186
        """draws (optional dict) records what entered from np.random: 'center' index, 'mask' boolean/None, 'count'."""
        points = self.points
        ci = np.random.randint(0, len(points))
        center_point = points[ci]
        scene_extract_mask = self._extract_z_box(center_point)
        points = points[scene_extract_mask]
        labels = self.labels[scene_extract_mask]
        colors = self.colors[scene_extract_mask]
        sample_mask = self._get_fix_sized_sample_mask(points, num_points_per_sample)
        if draws is not None:
            draws.update(center=ci, count=len(points), mask=sample_mask.copy() if sample_mask.dtype == bool else None)
        points = points[sample_mask]
        labels = labels[sample_mask]
        colors = colors[sample_mask]
        return self._center_box(points), points, labels, colors


def voxel_down_sample(points, colors, labels, voxel_size):
    """-> (sparse_points f64, sparse_colors f64, sparse_labels int) sorted by voxel index (ix, iy, iz)."""
    points = np.asarray(points, dtype=np.float64)
    colors = None if colors is None else np.asarray(colors, dtype=np.float64)
    min_bound = points.min(axis=0) - voxel_size * 0.5  # downsample.py:47
    vox = np.floor((points - min_bound) / voxel_size).astype(np.int64)
    key = (vox[:, 0] << 42) | (vox[:, 1] << 21) | vox[:, 2]
    order = np.argsort(key, kind="stable")  # members stay in input order
    ks = key[order]
    heads = np.flatnonzero(np.concatenate(([True], ks[1:] != ks[:-1])))
    ends = np.concatenate((heads[1:], [len(ks)]))
    sp = np.empty((len(heads), 3)); sc = np.zeros((len(heads), 3)); sl = np.zeros(len(heads), dtype=np.int64)
    for v, (a, b) in enumerate(zip(heads, ends)):
        ids = order[a:b]
        acc = np.zeros(3)
        for i in ids:  # sequential float64 accumulation, like Open3D's AccumulatedPoint
            acc = acc + points[i]
        sp[v] = acc / float(len(ids))
        if colors is not None:
            acc = np.zeros(3)
            for i in ids:
                acc = acc + colors[i]
            sc[v] = acc / float(len(ids))
        if labels is not None:
            sl[v] = np.bincount(labels[ids]).argmax()  # downsample.py:60
    return sp, sc, (sl if labels is not None else None)
