// pn2_label_interp.hip -- InterpolateLabelWithColor for gfx950: the dense-label kNN vote that follows the
// SA/FP stack in the reference's inference pipeline (tf_ops/tf_interpolate.cpp:71-115, callers
// interpolate.py:35-44, predict.py:40-42, kitti_predict.py:61-63).  The reference builds an Open3D/FLANN
// KD-tree over the sparse points on the CPU and queries it per dense point under OpenMP; here the sparse
// points are binned into a uniform grid on the device (cell keys -> radix sort -> cell ranges) and every
// dense point searches growing cubic shells of cells until its knn-th neighbour is provably final.
// Exact semantics (same as oracle_interpolate_label_with_color): squared L2 in float64,
// ((0+dx*dx)+dy*dy)+dz*dz, ascending, ties -> lowest index; vote of :96-107; colours of :45-47.
#include <hipcub/hipcub.hpp>

#include <math.h>

#include "pn2_common.h"

namespace {

constexpr int kLiMaxCells = 1 << 21;  // dense cell table (2 x 8 MiB of int32)
constexpr int kLiCellBits = 21;

struct LiParams {
    float lo[3], hi[3];
    float h, inv_h, slack;
    int n[3];
};

struct LiLayout {  // byte offsets into the caller's workspace (256-byte aligned)
    size_t params, bbox, keys_in, keys_out, vals_in, vals_out, cell_start, cell_end, pts, cub, total;
    size_t cub_bytes;
};

inline size_t li_align(size_t x) { return (x + 255) & ~(size_t)255; }

LiLayout li_layout(int ns) {
    LiLayout L{};
    size_t o = 0;
    const size_t n = ns > 0 ? (size_t)ns : 1;
    L.params = o; o = li_align(o + sizeof(LiParams));
    L.bbox = o; o = li_align(o + 6 * sizeof(unsigned));
    L.keys_in = o; o = li_align(o + n * 4);
    L.keys_out = o; o = li_align(o + n * 4);
    L.vals_in = o; o = li_align(o + n * 4);
    L.vals_out = o; o = li_align(o + n * 4);
    L.cell_start = o; o = li_align(o + (size_t)kLiMaxCells * 4);
    L.cell_end = o; o = li_align(o + (size_t)kLiMaxCells * 4);
    L.pts = o; o = li_align(o + n * 16);
    size_t cub = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, cub, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                       (int*)nullptr, (int)n, 0, kLiCellBits);
    L.cub_bytes = cub;
    L.cub = o; o = li_align(o + cub);
    L.total = o;
    return L;
}

// float <-> unsigned with the same ordering (for atomicMin / atomicMax on coordinates)
__device__ __forceinline__ unsigned li_ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float li_unord(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void li_init_kernel(unsigned* bbox) {
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0xffffffffu;       // running minima
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u;           // running maxima
}

__global__ void __launch_bounds__(256)
li_bbox_kernel(int ns, const float* __restrict__ pts, unsigned* __restrict__ bbox) {
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const unsigned o = li_ord(pts[(size_t)i * 3 + a]);
            mn[a] = o < mn[a] ? o : mn[a];
            mx[a] = o > mx[a] ? o : mx[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const unsigned wmn = ~pn2_wave_umax(~mn[a]);
        const unsigned wmx = pn2_wave_umax(mx[a]);
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&bbox[a], wmn);
            atomicMax(&bbox[3 + a], wmx);
        }
    }
}

// one thread: cell size so that the grid has about ns/2 cells (<= kLiMaxCells)
__global__ void li_params_kernel(int ns, const unsigned* __restrict__ bbox, LiParams* __restrict__ prm) {
    float lo[3], hi[3], ext[3], maxext = 0.f;
    for (int a = 0; a < 3; ++a) {
        lo[a] = li_unord(bbox[a]);
        hi[a] = li_unord(bbox[3 + a]);
        ext[a] = hi[a] - lo[a];
        maxext = fmaxf(maxext, ext[a]);
    }
    float h = 1.0f;
    int n[3] = {1, 1, 1};
    if (maxext > 0.f) {
        float target = (float)ns * 0.5f;
        target = target < 1.f ? 1.f : (target > (float)(kLiMaxCells / 2) ? (float)(kLiMaxCells / 2) : target);
        float vol = 1.f;
        for (int a = 0; a < 3; ++a) vol *= fmaxf(ext[a], maxext * 1e-3f);  // flat clouds: no zero-thickness axis
        h = cbrtf(vol / target);
        h = fmaxf(h, maxext * (1.0f / 2048.0f));
        for (int it = 0; it < 64; ++it) {
            long long tot = 1;
            for (int a = 0; a < 3; ++a) {
                n[a] = (int)floorf(ext[a] / h) + 1;
                tot *= n[a];
            }
            if (tot <= (long long)kLiMaxCells) break;
            h *= 1.26f;
        }
    }
    for (int a = 0; a < 3; ++a) { prm->lo[a] = lo[a]; prm->hi[a] = hi[a]; prm->n[a] = n[a]; }
    prm->h = h;
    prm->inv_h = 1.0f / h;
    prm->slack = 1e-6f * (maxext + h);  // >> the rounding of (x - lo) * inv_h at a cell boundary
}

// the ONE cell-coordinate function (points and queries): monotone in x
__device__ __forceinline__ int li_cell(float x, float lo, float inv_h, int n) {
    int c = (int)floorf((x - lo) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__global__ void __launch_bounds__(256)
li_keys_kernel(int ns, const float* __restrict__ pts, const LiParams* __restrict__ prm, unsigned* __restrict__ keys,
               int* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const int cx = li_cell(pts[(size_t)i * 3 + 0], prm->lo[0], prm->inv_h, prm->n[0]);
    const int cy = li_cell(pts[(size_t)i * 3 + 1], prm->lo[1], prm->inv_h, prm->n[1]);
    const int cz = li_cell(pts[(size_t)i * 3 + 2], prm->lo[2], prm->inv_h, prm->n[2]);
    keys[i] = (unsigned)((cz * prm->n[1] + cy) * prm->n[0] + cx);
    vals[i] = i;
}

__global__ void __launch_bounds__(256)
li_bounds_kernel(int ns, const float* __restrict__ pts, const unsigned* __restrict__ keys, const int* __restrict__ vals,
                 int* __restrict__ cell_start, int* __restrict__ cell_end, float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const unsigned k = keys[i];
    if (i == 0 || keys[i - 1] != k) cell_start[k] = i;
    if (i == ns - 1 || keys[i + 1] != k) cell_end[k] = i + 1;
    const int src = vals[i];  // stable sort: ascending original index inside a cell
    sorted[i] = make_float4(pts[(size_t)src * 3 + 0], pts[(size_t)src * 3 + 1], pts[(size_t)src * 3 + 2],
                            __int_as_float(src));
}

__device__ __forceinline__ void li_color(int label, uint8_t* __restrict__ c) {
    // tf_interpolate.cpp:45-47
    constexpr unsigned lut[9] = {0xffffffu, 0xff0000u, 0x000080u, 0xff00ffu, 0x008000u,
                                 0x0000ffu, 0x800080u, 0x800000u, 0x008080u};  // 0xBBGGRR
    const unsigned v = (label >= 0 && label < 9) ? lut[label] : 0u;
    c[0] = (uint8_t)(v & 0xff); c[1] = (uint8_t)((v >> 8) & 0xff); c[2] = (uint8_t)((v >> 16) & 0xff);
}

// one thread per dense point
template <int KMAX>
__global__ void __launch_bounds__(256)
li_query_kernel(int nd, int kf, const LiParams* __restrict__ prm, const int* __restrict__ cell_start,
                const int* __restrict__ cell_end, const float4* __restrict__ sorted,
                const int* __restrict__ labels, const float* __restrict__ dense, int* __restrict__ out_labels,
                uint8_t* __restrict__ out_colors) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nd) return;
    const float qxf = dense[(size_t)j * 3 + 0], qyf = dense[(size_t)j * 3 + 1], qzf = dense[(size_t)j * 3 + 2];
    const double qx = qxf, qy = qyf, qz = qzf;
    const int nx = prm->n[0], ny = prm->n[1], nz = prm->n[2];
    // cell of the query CLAMPED into the box: the shell bound below is about that point
    const int cx = li_cell(fminf(fmaxf(qxf, prm->lo[0]), prm->hi[0]), prm->lo[0], prm->inv_h, nx);
    const int cy = li_cell(fminf(fmaxf(qyf, prm->lo[1]), prm->hi[1]), prm->lo[1], prm->inv_h, ny);
    const int cz = li_cell(fminf(fmaxf(qzf, prm->lo[2]), prm->hi[2]), prm->lo[2], prm->inv_h, nz);
    // squared distance from the query to the bounding box of the sparse points (0 inside)
    double o2 = 0.0;
    {
        const float q[3] = {qxf, qyf, qzf};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float c = fminf(fmaxf(q[a], prm->lo[a]), prm->hi[a]);
            const double o = (double)q[a] - (double)c;
            o2 += o * o;
        }
    }
    const double h = prm->h, slack = prm->slack;
    double bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int p = 0; p < KMAX; ++p) { bd[p] = INFINITY; bi[p] = 0x7fffffff; }
    int cnt = 0;
    auto visit = [&](int x, int y, int z) {
        const int cell = (z * ny + y) * nx + x;
        const int e = cell_end[cell];
        for (int p = cell_start[cell]; p < e; ++p) {
            const float4 pt = sorted[p];
            const double dx = qx - (double)pt.x, dy = qy - (double)pt.y, dz = qz - (double)pt.z;
            const double d = (dx * dx + dy * dy) + dz * dz;  // contraction is off
            const int id = __float_as_int(pt.w);
            // keep the kf smallest by (distance, index)
            const bool room = cnt < kf;
            bool better = false;
#pragma unroll
            for (int s = 0; s < KMAX; ++s)
                if (s == kf - 1) better = d < bd[s] || (d == bd[s] && id < bi[s]);
            if (!room && !better) continue;
            const int pos = room ? cnt : kf - 1;
#pragma unroll
            for (int s = 0; s < KMAX; ++s)
                if (s == pos) { bd[s] = d; bi[s] = id; }
            cnt += room ? 1 : 0;
#pragma unroll
            for (int s = KMAX - 1; s > 0; --s) {
                if (s < cnt && (bd[s] < bd[s - 1] || (bd[s] == bd[s - 1] && bi[s] < bi[s - 1]))) {
                    const double td = bd[s]; bd[s] = bd[s - 1]; bd[s - 1] = td;
                    const int ti = bi[s]; bi[s] = bi[s - 1]; bi[s - 1] = ti;
                }
            }
        }
    };
    int maxr = cx;
    maxr = max(maxr, nx - 1 - cx); maxr = max(maxr, cy); maxr = max(maxr, ny - 1 - cy);
    maxr = max(maxr, cz); maxr = max(maxr, nz - 1 - cz);
    for (int r = 0; r <= maxr; ++r) {
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= nz) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= ny) continue;
                const bool face = (dz == -r || dz == r || dy == -r || dy == r);
                if (face) {
                    const int x0 = max(cx - r, 0), x1 = min(cx + r, nx - 1);
                    for (int x = x0; x <= x1; ++x) visit(x, y, z);
                } else {
                    if (cx - r >= 0) visit(cx - r, y, z);
                    if (cx + r < nx) visit(cx + r, y, z);  // r > 0 here
                }
            }
        }
        if (cnt == kf) {
            // every unvisited cell is >= r+1 cells away: its points are farther than r*h (minus the
            // boundary rounding) from the clamped query, plus the query's own offset from the box
            const double rb = (double)r * h - slack;
            double worst = 0.0;
#pragma unroll
            for (int s = 0; s < KMAX; ++s)
                if (s == kf - 1) worst = bd[s];
            if (rb > 0.0 && worst < o2 + rb * rb) break;
        }
    }
    // vote (tf_interpolate.cpp:96-107): a label wins when its running count exceeds the running maximum
    int lab[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) lab[s] = s < cnt ? labels[bi[s]] : -1;
    int best = -1, max_count = 0;
#pragma unroll
    for (int a = 0; a < KMAX; ++a) {
        if (a < cnt) {
            int c = 0;
#pragma unroll
            for (int e = 0; e < KMAX; ++e) c += (e <= a && lab[e] == lab[a]) ? 1 : 0;
            if (c > max_count) { best = lab[a]; max_count = c; }
        }
    }
    out_labels[j] = best;
    li_color(best, out_colors + (size_t)j * 3);
}

__global__ void __launch_bounds__(256)
li_fill_kernel(int nd, int* __restrict__ out_labels, uint8_t* __restrict__ out_colors) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nd) return;
    out_labels[j] = -1;
    out_colors[(size_t)j * 3 + 0] = 0; out_colors[(size_t)j * 3 + 1] = 0; out_colors[(size_t)j * 3 + 2] = 0;
}

}  // namespace

extern "C" size_t pn2_interpolate_label_workspace_bytes(int num_sparse_points) {
    if (num_sparse_points < 0) return 0;
    return li_layout(num_sparse_points).total;
}

extern "C" int pn2_interpolate_label_with_color(int num_sparse_points, int num_dense_points,
                                                const float* sparse_points, const int* sparse_labels,
                                                const float* dense_points, int* dense_labels,
                                                uint8_t* dense_colors, int knn, void* workspace,
                                                size_t workspace_bytes, void* stream) {
    if (num_sparse_points < 0 || num_dense_points < 0 || knn <= 0) return PN2_EINVAL;
    if (num_dense_points == 0) return PN2_OK;
    if (!dense_points || !dense_labels || !dense_colors) return PN2_ENULL;
    if (knn > 16) return PN2_EUNSUP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nd = num_dense_points, ns = num_sparse_points;
    const int qgrid = (nd + 255) / 256;
    if (ns == 0) {
        li_fill_kernel<<<qgrid, 256, 0, st>>>(nd, dense_labels, dense_colors);
        PN2_RETURN_IF_LAUNCH_FAILED();
        return PN2_OK;
    }
    if (!sparse_points || !sparse_labels || !workspace) return PN2_ENULL;
    const LiLayout L = li_layout(ns);
    if (workspace_bytes < L.total || ((uintptr_t)workspace & 255) != 0) return PN2_EINVAL;
    char* ws = static_cast<char*>(workspace);
    LiParams* prm = reinterpret_cast<LiParams*>(ws + L.params);
    unsigned* bbox = reinterpret_cast<unsigned*>(ws + L.bbox);
    unsigned* keys_in = reinterpret_cast<unsigned*>(ws + L.keys_in);
    unsigned* keys_out = reinterpret_cast<unsigned*>(ws + L.keys_out);
    int* vals_in = reinterpret_cast<int*>(ws + L.vals_in);
    int* vals_out = reinterpret_cast<int*>(ws + L.vals_out);
    int* cell_start = reinterpret_cast<int*>(ws + L.cell_start);
    int* cell_end = reinterpret_cast<int*>(ws + L.cell_end);
    float4* sorted = reinterpret_cast<float4*>(ws + L.pts);
    const int sgrid = (ns + 255) / 256;

    li_init_kernel<<<1, 64, 0, st>>>(bbox);
    li_bbox_kernel<<<sgrid < 1024 ? sgrid : 1024, 256, 0, st>>>(ns, sparse_points, bbox);
    li_params_kernel<<<1, 1, 0, st>>>(ns, bbox, prm);
    li_keys_kernel<<<sgrid, 256, 0, st>>>(ns, sparse_points, prm, keys_in, vals_in);
    PN2_RETURN_IF_LAUNCH_FAILED();
    size_t cub = L.cub_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub, keys_in, keys_out, vals_in, vals_out, ns, 0,
                                                      kLiCellBits, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(cell_start, 0, (size_t)kLiMaxCells * 4, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(cell_end, 0, (size_t)kLiMaxCells * 4, st);
    if (e != hipSuccess) return (int)e;
    li_bounds_kernel<<<sgrid, 256, 0, st>>>(ns, sparse_points, keys_out, vals_out, cell_start, cell_end, sorted);
    const int kf = knn < ns ? knn : ns;
    if (knn <= 4)
        li_query_kernel<4><<<qgrid, 256, 0, st>>>(nd, kf, prm, cell_start, cell_end, sorted, sparse_labels, dense_points,
                                                  dense_labels, dense_colors);
    else
        li_query_kernel<16><<<qgrid, 256, 0, st>>>(nd, kf, prm, cell_start, cell_end, sorted, sparse_labels, dense_points,
                                                   dense_labels, dense_colors);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
