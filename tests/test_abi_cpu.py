"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/pn2_abi.h declares, and validates arguments before touching the device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pn2_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pn2_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_nine_reference_ops():
    names = _declared()
    for n in ["pn2_farthest_point_sample", "pn2_gather_point", "pn2_gather_point_grad", "pn2_query_ball_point",
              "pn2_group_point", "pn2_group_point_grad", "pn2_three_nn", "pn2_three_interpolate",
              "pn2_three_interpolate_grad", "pn2_sa_mlp_max_fused", "pn2_linear", "pn2_fp_interp_concat"]:
        assert n in names


def test_library_exports_every_declared_symbol(pn2):
    lib = ctypes.CDLL(pn2._lib.LIB_PATH)
    for n in _declared():
        assert hasattr(lib, n), n
    assert pn2._lib.lib.pn2_abi_version() == 1
    assert b"gfx950" in pn2._lib.lib.pn2_build_info()
    # the shipped library has no tuning hooks: no process-global kernel-selection state behind the ABI
    assert not hasattr(lib, "pn2_debug_set") and not hasattr(lib, "pn2_debug_set_grouping")


def test_python_signatures_cover_the_header(pn2):
    decl = set(_declared()) - {"pn2_abi_version", "pn2_build_info", "pn2_strerror",
                               "pn2_interpolate_label_workspace_bytes", "pn2_fps_large_workspace_bytes",
                               "pn2_bn_workspace_bytes", "pn2_three_interpolate_grad_workspace_bytes",
                               "pn2_group_point_grad_workspace_bytes", "pn2_voxel_downsample_workspace_bytes",
                               "pn2_scatter_plan_bytes", "pn2_ball_query_bin_bytes"}  # bound separately: size_t
    assert decl <= set(pn2._lib.SIGNATURES)
    # ... and nothing is bound that the header does not declare (no undocumented entry points in the product)
    assert set(pn2._lib.SIGNATURES) <= decl
    assert pn2._lib.lib.pn2_interpolate_label_workspace_bytes(1000) > 2 * (1 << 21) * 4  # two cell tables + lists


def test_argument_validation_needs_no_gpu(pn2):
    L = pn2._lib.lib
    nul = None
    assert L.pn2_farthest_point_sample(0, 8, 4, nul, nul, nul, 1, nul) == -1      # PN2_EINVAL
    assert L.pn2_farthest_point_sample(1, 8, 4, nul, nul, nul, 1, nul) == -2      # PN2_ENULL
    assert L.pn2_query_ball_point(1, 8, 4, 0.0, 4, nul, nul, nul, nul, 1, nul) == -1  # radius must be > 0
    assert L.pn2_query_ball_point(1, 8, 4, 0.5, 0, nul, nul, nul, nul, 1, nul) == -1  # nsample must be > 0
    assert L.pn2_three_nn(1, 8, 2, nul, nul, nul, nul, nul) == -1                  # needs >= 3 known points
    assert L.pn2_group_point(1, 8, 0, 4, 4, nul, nul, nul, nul) == -1
    assert b"PN2_ENULL" in L.pn2_strerror(-2)
    assert L.pn2_interpolate_label_with_color(10, 10, nul, nul, nul, nul, nul, 0, nul, 0, nul) == -1   # knn > 0
    assert L.pn2_interpolate_label_with_color(10, 10, nul, nul, nul, nul, nul, 3, nul, 0, nul) == -2
    assert L.pn2_fp_mlp_fused(1, 8, 4, 0, 8, nul, nul, nul, nul, 1, nul, nul, nul, nul, nul) == -2
    assert L.pn2_mlp_chain(0, 8, nul, 1, nul, nul, nul, 0, nul, nul) == -1
    assert L.pn2_scene_extract_z_box(0, nul, 1, nul, 5.0, 5.0, 1.0, 8, nul, nul, nul) == -1
    assert L.pn2_scene_extract_z_box(10, nul, 1, nul, 5.0, 5.0, 1.0, 8, nul, nul, nul) == -2
    assert L.pn2_scene_sample(1, 0, 8, nul, nul, nul, nul, nul, nul, 5.0, 5.0, nul, nul, nul, nul, nul, nul, nul) == -1
    assert L.pn2_voxel_downsample(10, nul, nul, nul, 0.0, nul, nul, nul, nul, nul, nul, 0, nul) == -1   # voxel_size > 0
    assert L.pn2_voxel_downsample(10, nul, nul, nul, 0.05, nul, nul, nul, nul, nul, nul, 0, nul) == -2
    # pn2_coarse_geometry: host arrays of per-level sizes and of device pointers
    import ctypes
    one_i = lambda v: (ctypes.c_int * 1)(v)        # noqa: E731
    one_f = lambda v: (ctypes.c_float * 1)(v)      # noqa: E731
    one_p = lambda v: (ctypes.c_void_p * 1)(v)     # noqa: E731
    fake = ctypes.c_void_p(4096)                   # never dereferenced: every call below is refused before a launch
    assert L.pn2_coarse_geometry(1, 64, 1, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, 2, 1, nul) == -2
    args = lambda b, n0, m, nn: (b, n0, 1, one_i(m), one_f(0.5), one_i(8), fake, nul, one_p(4096), one_p(4096), one_p(4096), nul,   # noqa: E731
                                 one_p(4096) if nn else nul, one_p(4096) if nn else nul, nul, 2, 1, nul)
    assert L.pn2_coarse_geometry(*args(0, 64, 16, False)) == -1       # b > 0
    assert L.pn2_coarse_geometry(*args(1, 2048, 16, False)) == -4     # PN2_EUNSUP: source cloud above 1024 points
    assert L.pn2_coarse_geometry(*args(1, 64, 65, False)) == -4       # more samples than points
    assert L.pn2_coarse_geometry(*args(1, 1024, 512, True)) == -4     # 3-NN table over more than 256 samples
    assert L.pn2_coarse_geometry(*args(1, 64, 2, True)) == -1         # 3-NN needs >= 3 known points
    a = list(args(1, 64, 16, False)); a[-3] = 7
    assert L.pn2_coarse_geometry(*a) == -1                            # unknown arithmetic mode
    # r05: the *_ld entry points (row stride in floats) and pn2_relu_grad: refused before any launch
    assert L.pn2_fps_nested_ld(1, 64, 8, fake, 2, fake, nul, nul, nul, 2, nul) == -1             # ld >= 3
    assert L.pn2_fps_nested_ld(1, 64, 8, nul, 6, fake, nul, nul, nul, 2, nul) == -2              # inp
    assert L.pn2_fps_nested_ld(1, 20000, 8, fake, 6, fake, nul, nul, nul, 2, nul) == -4          # streaming kernel: dense rows only
    assert L.pn2_fps_nested_ld(1, 64, 8, fake, 6, fake, nul, nul, nul, 9, nul) == -1             # arithmetic mode
    assert L.pn2_query_ball_point_ld(1, 8192, 1024, 0.5, 32, fake, 2, fake, fake, fake, 1, nul) == -1
    assert L.pn2_query_ball_point_ld(1, 8192, 1024, -1.0, 32, fake, 6, fake, fake, fake, 1, nul) == -1
    assert L.pn2_query_ball_point_ld(1, 8192, 1024, 0.5, 32, fake, 6, nul, fake, fake, 1, nul) == -2
    assert L.pn2_query_ball_point_ld(1, 1000, 100, 0.5, 32, fake, 6, fake, fake, fake, 1, nul) == -4   # outside the LDS-grid kernel
    assert L.pn2_three_nn_ld(1, 64, 2, fake, 6, fake, fake, fake, nul) == -1                     # >= 3 known points
    assert L.pn2_ball_query_bin_ld(1, 8192, 0.5, fake, 2, fake, 1 << 20, nul) == -1              # r06: ld >= 3
    assert L.pn2_ball_query_bin_ld(1, 8192, 0.5, nul, 6, fake, 1 << 20, nul) == -2
    assert L.pn2_ball_query_bin_ld(1, 20000, 0.5, fake, 6, fake, 1 << 20, nul) == -4             # outside the LDS-grid kernel
    assert L.pn2_three_nn_ld(1, 64, 16, fake, 1, fake, fake, fake, nul) == -1                    # ld >= 3
    assert L.pn2_three_nn_ld(1, 64, 16, fake, 6, nul, fake, fake, nul) == -2
    w1 = one_i(32)
    assert L.pn2_sa_mlp_max_fused_ld(1, 64, 8, 32, 3, fake, 2, fake, fake, 6, fake, 1, w1, one_p(4096), one_p(4096), fake, nul) == -1
    assert L.pn2_sa_mlp_max_fused_ld(1, 64, 8, 32, 3, fake, 6, fake, fake, 2, fake, 1, w1, one_p(4096), one_p(4096), fake, nul) == -1
    assert L.pn2_sa_mlp_max_fused_ld(1, 64, 8, 32, 8, fake, 6, fake, fake, 16, fake, 1, w1, one_p(4096), one_p(4096), fake, nul) == -4  # 16-byte gathers: dense rows
    assert L.pn2_fp_mlp_fused_pre_ld(1, 64, 16, 3, fake, fake, fake, 2, fake, 2, w1, one_p(4096), one_p(4096), fake, nul) == -1    # ld < c1
    assert L.pn2_relu_grad(0, fake, fake, fake, nul) == -1
    assert L.pn2_relu_grad(16, nul, fake, fake, nul) == -2


def test_rows_in_place_recognises_column_blocks():
    """_lib.rows_in_place: a dense (b,n,c) tensor or a column block of a wider dense one is read where it lies (row stride in
    floats); anything else is copied.  Host logic only."""
    import torch
    from pn2_amd._lib import rows_in_place
    pc = torch.arange(2 * 5 * 6, dtype=torch.float32).reshape(2, 5, 6)
    t, ld = rows_in_place(pc[:, :, 0:3])
    assert ld == 6 and t.data_ptr() == pc.data_ptr() and not t.is_contiguous()
    t, ld = rows_in_place(pc[:, :, 3:6])
    assert ld == 6 and t.data_ptr() == pc.data_ptr() + 12
    t, ld = rows_in_place(pc)
    assert ld == 6 and t.data_ptr() == pc.data_ptr()
    t, ld = rows_in_place(pc[:, ::2, 0:3])          # every second row: clouds are no longer n * ld apart
    assert ld == 3 and t.is_contiguous() and torch.equal(t, pc[:, ::2, 0:3])
    t, ld = rows_in_place(pc.transpose(1, 2))        # not row-major at all
    assert ld == 5 and t.is_contiguous()
    t, ld = rows_in_place(pc[:1, :, 0:3])            # a single cloud
    assert torch.equal(t, pc[:1, :, 0:3]) and ld in (3, 6)


def test_ops_refuse_cpu_tensors_loudly(pn2):
    import torch
    x = torch.zeros(1, 16, 3)
    with pytest.raises(ValueError, match="MI355X only"):
        pn2.farthest_point_sample(4, x)
    with pytest.raises(ValueError, match="MI355X only"):
        pn2.three_nn(x, x)
    with pytest.raises(ValueError, match="positive npoint"):
        pn2.farthest_point_sample(0, x)
    with pytest.raises(ValueError, match="positive radius"):
        pn2.query_ball_point(-1.0, 4, x, x)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "open3d-pointnet2-semantic3d_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "pn2_oracle" not in txt, f
