"""Voxel down-sampling with a majority label on the device (reference downsample.py:46-67: Open3D
voxel_down_sample_and_trace + np.bincount(cubic_labels).argmax() per voxel, a host loop over every voxel).

    sparse_points, sparse_colors, sparse_labels = down_sample_arrays(points, colors, labels, voxel_size=0.05)

Points / colours float64 like the reference's Open3D arrays; output voxels sorted by voxel index (Open3D's own order is
the iteration order of a std::unordered_map, i.e. unspecified)."""
import torch

from ._lib import check, lib, ptr, require_cuda, stream_ptr


def down_sample_arrays(points, colors=None, labels=None, voxel_size=0.05, skip_label_zero=True):
    """points (n,3) f64, colors (n,3) f64 or None, labels (n) int32 or None (device tensors).  `skip_label_zero` drops the
    unlabeled points first, as downsample.py:29-43 does when labels exist."""
    require_cuda(points, colors, labels)
    points = points.to(torch.float64)
    if labels is not None and skip_label_zero:
        keep = labels != 0
        points = points[keep]
        colors = None if colors is None else colors[keep]
        labels = labels[keep]
    points = points.contiguous()
    colors = None if colors is None else colors.to(torch.float64).contiguous()
    labels = None if labels is None else labels.to(torch.int32).contiguous()
    n, dev = points.shape[0], points.device
    out_p = torch.empty((n, 3), dtype=torch.float64, device=dev)
    out_c = torch.empty((n, 3), dtype=torch.float64, device=dev)
    out_l = torch.empty((n,), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((lib.pn2_voxel_downsample_workspace_bytes(n) + 255,), dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(dev):
        check(lib.pn2_voxel_downsample(n, ptr(points), ptr(colors), ptr(labels), float(voxel_size), ptr(out_p), ptr(out_c),
                                       ptr(out_l), ptr(count), ptr(status), ws.data_ptr() + off, ws.numel() - off,
                                       stream_ptr()), "pn2_voxel_downsample")
    st, m = int(status.item()), int(count.item())
    if st:
        raise ValueError("pn2_voxel_downsample: %s" % {1: "voxel index exceeds 21 bits per axis", 2: "label outside [0,64)"}[st])
    return out_p[:m], out_c[:m], (out_l[:m] if labels is not None else None)
