"""On-GPU scene sampler: SemanticFileData of the reference (dataset/semantic_dataset.py:57-214) with the scene resident
in HBM.  The reference crops, samples and centres every column in numpy inside an mp.Pool and ships batches over PCIe;
here `sample_batch` is two launches (pn2_scene_extract_z_box, pn2_scene_sample) and the batch never leaves the device.

Same method names and return values as the reference (`sample`, `sample_batch` -> points_centered, points_raw, labels,
colors).  The reference's np.random draws are explicit inputs, so a caller that replays the reference's RNG stream gets
bit-identical batches (tests/test_dataset_gpu.py); by default the draws come from a torch generator on the device.
Coordinates and colours are float64 in the reference (Open3D arrays) and stay float64 here; `points_centered` is cast
to float32, the dtype the network is fed with.
"""
import torch

from .._lib import check, lib, ptr, require_cuda, stream_ptr
from ..util.point_cloud_util import load_labels, read_point_cloud_pcd


class SemanticFileData:
    def __init__(self, file_path_without_ext=None, has_label=True, use_color=True, box_size_x=10, box_size_y=10,
                 points=None, labels=None, colors=None, device="cuda", strict=False):
        """Loads <prefix>.pcd / <prefix>.labels (semantic_dataset.py:60-82), or takes the arrays directly; sorts by x
        (:84-88) and uploads the scene once."""
        import numpy as np
        self.file_path_without_ext = file_path_without_ext
        self.box_size_x, self.box_size_y = box_size_x, box_size_y
        # strict: sample_batch() itself raises on a bad sample (one host synchronisation per batch).  Otherwise a rejected
        # sample (status != 0) comes back ZERO-FILLED, never as uninitialised memory, and check_last() reports it.
        self.strict = bool(strict)
        if points is None:
            points, file_colors = read_point_cloud_pcd(file_path_without_ext + ".pcd")
            labels = load_labels(file_path_without_ext + ".labels") if has_label else np.zeros(len(points), dtype=bool)
            colors = file_colors if use_color else np.zeros_like(points)
        points = np.asarray(points, dtype=np.float64)
        if labels is None:
            labels = np.zeros(len(points), dtype=bool)
        if colors is None or not use_color:
            colors = np.zeros_like(points)
        sort_idx = np.argsort(points[:, 0])  # :85 (same numpy call, same permutation as the reference)
        dev = torch.device(device)
        self.points = torch.from_numpy(np.ascontiguousarray(points[sort_idx])).to(dev)
        self.labels = torch.from_numpy(np.ascontiguousarray(np.asarray(labels)[sort_idx].astype(np.int32))).to(dev)
        self.colors = torch.from_numpy(np.ascontiguousarray(np.asarray(colors, dtype=np.float64)[sort_idx])).to(dev)
        # scene_z_size = max z - min z (:132; the reference recomputes it per sample)
        self.scene_z_size = float(self.points[:, 2].max() - self.points[:, 2].min())
        self.generator = torch.Generator(device=dev)

    def __len__(self):
        return self.points.shape[0]

    def extract_z_box(self, center_points, capacity=None):
        """_extract_z_box (:123-163) for (b,3) float64 centre points -> idx (b,cap) int32 scene indices in scene order,
        cnt (b) int32."""
        require_cuda(self.points, center_points)
        c = center_points.to(torch.float64).contiguous()
        b = c.shape[0]
        cap = int(capacity or len(self))
        idx = torch.empty((b, cap), dtype=torch.int32, device=self.points.device)
        cnt = torch.empty((b,), dtype=torch.int32, device=self.points.device)
        with torch.cuda.device(self.points.device):
            check(lib.pn2_scene_extract_z_box(len(self), ptr(self.points), b, ptr(c), self.box_size_x / 2, self.box_size_y / 2,
                                              self.scene_z_size, cap, ptr(idx), ptr(cnt), stream_ptr()),
                  "pn2_scene_extract_z_box")
        return idx, cnt

    def sample_batch(self, batch_size, num_points_per_sample, center_indices=None, sample_masks=None, capacity=None):
        """sample() x batch_size (:165-214) in two launches.
        center_indices (b) int: the reference's np.random.randint(0, len(points)) draws (default: device RNG);
        sample_masks   (b,cap) uint8/bool: for columns larger than num_points_per_sample the reference's shuffled
                       boolean mask in the first cnt entries (default: a uniform random subset from the device RNG).
        -> points_centered (b,n,3) f32, points_raw (b,n,3) f64, labels (b,n) i32, colors (b,n,3) f32."""
        dev = self.points.device
        n, npts = len(self), int(num_points_per_sample)
        if center_indices is None:
            center_indices = torch.randint(0, n, (batch_size,), device=dev, generator=self.generator)
        centers = self.points[center_indices.to(dev).long()]
        idx, cnt = self.extract_z_box(centers, capacity)
        cap = idx.shape[1]
        if sample_masks is None:
            # uniform random subset of exactly npts of the first cnt entries: the npts smallest of cnt random keys
            keys = torch.rand((batch_size, cap), device=dev, generator=self.generator)
            keys.masked_fill_(torch.arange(cap, device=dev)[None, :] >= cnt[:, None], 2.0)
            # (the INDICES of the k smallest keys, not `keys <= k-th key`: two equal fp32 keys -- a few percent of the batches
            # at 50k-100k points per column -- would select npts + 1 entries and the kernel rejects the sample, status 3)
            pick = torch.topk(keys, min(npts, cap), dim=1, largest=False, sorted=False).indices
            sample_masks = torch.zeros((batch_size, cap), dtype=torch.uint8, device=dev).scatter_(1, pick, 1)
        mask = sample_masks.to(dev).to(torch.uint8).contiguous()
        if mask.shape != (batch_size, cap):
            raise ValueError("sample_masks must be (batch_size, capacity)")
        sel = torch.zeros((batch_size, npts), dtype=torch.int32, device=dev)
        centered = torch.zeros((batch_size, npts, 3), dtype=torch.float32, device=dev)
        raw = torch.zeros((batch_size, npts, 3), dtype=torch.float64, device=dev)
        labels = torch.zeros((batch_size, npts), dtype=torch.int32, device=dev)
        colors = torch.zeros((batch_size, npts, 3), dtype=torch.float32, device=dev)
        status = torch.zeros((batch_size,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(lib.pn2_scene_sample(batch_size, npts, cap, ptr(self.points), ptr(self.labels), ptr(self.colors), ptr(idx),
                                       ptr(cnt), ptr(mask), self.box_size_x / 2, self.box_size_y / 2, ptr(sel), ptr(centered),
                                       ptr(raw), ptr(labels), ptr(colors), ptr(status), stream_ptr()), "pn2_scene_sample")
        self.last_status, self.last_sel, self.last_cnt = status, sel, cnt
        if self.strict:
            self.check_last()  # one host synchronisation per batch; strict=False leaves the check to the caller
        return centered, raw, labels, colors

    def sample(self, num_points_per_sample, **kw):
        c, r, l, col = self.sample_batch(1, num_points_per_sample, **kw)
        return c[0], r[0], l[0], col[0]

    def check_last(self):
        """raise if the last batch had an empty column / overflowed the capacity / got a bad mask (one host sync)."""
        st = self.last_status.cpu().tolist()
        if any(st):
            raise RuntimeError("scene sampler status per sample (1 empty column, 2 capacity, 3 mask): %s" % st)
