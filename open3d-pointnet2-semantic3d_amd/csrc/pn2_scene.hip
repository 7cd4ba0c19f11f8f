// pn2_scene.hip -- the step BEFORE the SA/FP stack, on the device:
//   * the scene sampler of dataset/semantic_dataset.py:90-186 (column crop on the x-sorted scene, fixed-size
//     sampling, centring) -- the reference runs it in numpy inside an mp.Pool and ships every batch over PCIe;
//   * voxel down-sampling with a majority label, downsample.py:46-67 (Open3D voxel_down_sample_and_trace +
//     np.bincount(...).argmax()).
// Everything the reference computes in float64 (Open3D point arrays are float64) is computed in float64 here with
// the same operations in the same order, so results are bit-identical; random draws are INPUTS (the caller owns the
// RNG), never generated inside a kernel.
#include <hipcub/hipcub.hpp>

#include "pn2_common.h"

namespace {

constexpr int kScT = 1024;  // threads per sample

// exclusive prefix of per-thread flags inside a 1024-thread block; returns this thread's offset and the block total
__device__ __forceinline__ int block_excl_scan(int flag, int* wsum, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < kScT / 64; ++w) {
        const int c = wsum[w];
        if (w < wave) base += c;
        tot += c;
    }
    __syncthreads();  // wsum reusable
    total = tot;
    return base + before;
}

// first index i in [0, n) with x[3*i] >= v  (np.searchsorted(points[:, 0], v), side='left'); points sorted by x
__device__ __forceinline__ int lower_bound_x(const double* __restrict__ pts, int n, double v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (pts[(size_t)mid * 3] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// _extract_z_box (semantic_dataset.py:123-163): one workgroup per sample; writes the scene indices of the column in
// scene order (= the order of points[scene_extract_mask]) and their count.
__global__ void __launch_bounds__(kScT)
scene_extract_kernel(int n, const double* __restrict__ pts, const double* __restrict__ centers, double hx, double hy,
                     double zsize, int cap, int* __restrict__ idx_all, int* __restrict__ cnt_all) {
    __shared__ int wsum[kScT / 64];
    const int s = blockIdx.x, tid = threadIdx.x;
    const double cx = centers[s * 3 + 0], cy = centers[s * 3 + 1], cz = centers[s * 3 + 2];
    // box_min = center_point - [bx/2, by/2, scene_z_size], box_max = center_point + [...]  (:133-142), float64
    const double lo0 = cx - hx, lo1 = cy - hy, lo2 = cz - zsize;
    const double hi0 = cx + hx, hi1 = cy + hy, hi2 = cz + zsize;
    const int i_min = lower_bound_x(pts, n, lo0);  // :144
    const int i_max = lower_bound_x(pts, n, hi0);  // :145 (side='left': points with x == box_max[0] stay outside)
    int* __restrict__ out = idx_all + (size_t)s * cap;
    int count = 0;
    for (int base = i_min; base < i_max; base += kScT) {
        const int i = base + tid;
        int in = 0;
        if (i < i_max) {
            const double x = pts[(size_t)i * 3], y = pts[(size_t)i * 3 + 1], z = pts[(size_t)i * 3 + 2];
            in = (x >= lo0) & (x <= hi0) & (y >= lo1) & (y <= hi1) & (z >= lo2) & (z <= hi2);  // :146-153
        }
        int tot;
        const int off = block_excl_scan(in, wsum, tot);
        if (in && count + off < cap) out[count + off] = i;
        count += tot;
    }
    if (tid == 0) cnt_all[s] = count;  // may exceed cap: the caller must check (nothing is silently truncated)
}

// _get_fix_sized_sample_mask + the gathers + _center_box (semantic_dataset.py:90-121,165-186).
//   cnt > npts : `mask` (cnt bytes, exactly npts non-zero: the reference's shuffled boolean array) selects, order kept;
//   cnt <= npts: the index list is repeated until npts entries exist (arange doubled and cut == i mod cnt).
// Outputs per sample: sel (npts) scene indices, centered (npts,3) f32, raw (npts,3) f64, labels (npts), colors (npts,3) f32.
__global__ void __launch_bounds__(kScT)
scene_sample_kernel(int npts, int cap, const double* __restrict__ pts, const int* __restrict__ labels,
                    const double* __restrict__ colors, const int* __restrict__ idx_all, const int* __restrict__ cnt_all,
                    const unsigned char* __restrict__ mask_all, double hx, double hy, int* __restrict__ sel_all,
                    float* __restrict__ centered_all, double* __restrict__ raw_all, int* __restrict__ lab_all,
                    float* __restrict__ col_all, int* __restrict__ status) {
    __shared__ int wsum[kScT / 64];
    __shared__ double smin[3][kScT / 64];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int cnt = cnt_all[s];
    const int* __restrict__ idx = idx_all + (size_t)s * cap;
    int* __restrict__ sel = sel_all + (size_t)s * npts;
    if (cnt <= 0 || cnt > cap) {  // empty column (the reference asserts, :162) or capacity overflow: flag it
        if (tid == 0) status[s] = cnt <= 0 ? 1 : 2;
        return;
    }
    if (cnt > npts) {
        const unsigned char* __restrict__ mask = mask_all + (size_t)s * cap;
        int count = 0;
        for (int base = 0; base < cnt; base += kScT) {
            const int i = base + tid;
            const int in = (i < cnt) && mask[i] != 0;
            int tot;
            const int off = block_excl_scan(in, wsum, tot);
            if (in && count + off < npts) sel[count + off] = idx[i];
            count += tot;
        }
        if (count != npts) {  // the mask must hold exactly npts selections (:96-100)
            if (tid == 0) status[s] = 3;
            return;
        }
    } else {
        for (int j = tid; j < npts; j += kScT) sel[j] = idx[j % cnt];  // :102-106
    }
    __syncthreads();
    // box_min = np.min(points, axis=0) over the SAMPLED points (:111)
    double m0 = 1.0e300, m1 = 1.0e300, m2 = 1.0e300;
    for (int j = tid; j < npts; j += kScT) {
        const int k = sel[j];
        m0 = fmin(m0, pts[(size_t)k * 3]); m1 = fmin(m1, pts[(size_t)k * 3 + 1]); m2 = fmin(m2, pts[(size_t)k * 3 + 2]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        m0 = fmin(m0, __shfl_xor(m0, o)); m1 = fmin(m1, __shfl_xor(m1, o)); m2 = fmin(m2, __shfl_xor(m2, o));
    }
    if ((tid & 63) == 0) { smin[0][tid >> 6] = m0; smin[1][tid >> 6] = m1; smin[2][tid >> 6] = m2; }
    __syncthreads();
    for (int w = 0; w < kScT / 64; ++w) { m0 = fmin(m0, smin[0][w]); m1 = fmin(m1, smin[1][w]); m2 = fmin(m2, smin[2][w]); }
    const double sh0 = m0 + hx, sh1 = m1 + hy, sh2 = m2;  // shift (:112-118), float64
    if (tid == 0) status[s] = 0;
    float* __restrict__ cen = centered_all + (size_t)s * npts * 3;
    double* __restrict__ raw = raw_all ? raw_all + (size_t)s * npts * 3 : nullptr;
    int* __restrict__ lab = lab_all ? lab_all + (size_t)s * npts : nullptr;
    float* __restrict__ col = col_all ? col_all + (size_t)s * npts * 3 : nullptr;
    for (int j = tid; j < npts; j += kScT) {
        const int k = sel[j];
        const double x = pts[(size_t)k * 3], y = pts[(size_t)k * 3 + 1], z = pts[(size_t)k * 3 + 2];
        cen[j * 3 + 0] = (float)(x - sh0); cen[j * 3 + 1] = (float)(y - sh1); cen[j * 3 + 2] = (float)(z - sh2);  // :119, then astype(float32)
        if (raw) { raw[j * 3 + 0] = x; raw[j * 3 + 1] = y; raw[j * 3 + 2] = z; }
        if (lab) lab[j] = labels ? labels[k] : 0;
        if (col) {
#pragma unroll
            for (int c = 0; c < 3; ++c) col[j * 3 + c] = colors ? (float)colors[(size_t)k * 3 + c] : 0.f;
        }
    }
}

// ---- voxel down-sampling ------------------------------------------------------------------------------------------
// Open3D (IntelVCL/Open3D @33e46f7, not vendored by the reference) VoxelDownSampleAndTrace: voxel of a point =
// floor((p - voxel_min_bound) / voxel_size) per axis, the output point / colour of a voxel = the float64 sum of its
// members IN INPUT ORDER divided by their count.  Its output ORDER is the iteration order of a std::unordered_map and
// therefore unspecified: here voxels come out sorted by (ix, iy, iz).  Label of a voxel = np.bincount(labels).argmax()
// (downsample.py:57-62): the most frequent label, ties -> the smallest label.
__global__ void voxel_minmax_kernel(int n, const double* __restrict__ pts, double* __restrict__ mm /* 6: min xyz, max xyz, as ordered u64 */) {
    __shared__ double sm[6][4];
    double lo[3] = {1.0e300, 1.0e300, 1.0e300}, hi[3] = {-1.0e300, -1.0e300, -1.0e300};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const double v = pts[i * 3 + a]; lo[a] = fmin(lo[a], v); hi[a] = fmax(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o)); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { sm[a][wave] = lo[a]; sm[3 + a][wave] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        double v = sm[a][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = a < 3 ? fmin(v, sm[a][w]) : fmax(v, sm[a][w]);
        // float64 min/max through 64-bit integer atomics on an order-preserving key
        unsigned long long k = (unsigned long long)__double_as_longlong(v);
        k = (k >> 63) ? ~k : (k | 0x8000000000000000ull);
        if (a < 3) atomicMin(reinterpret_cast<unsigned long long*>(mm) + a, k);
        else atomicMax(reinterpret_cast<unsigned long long*>(mm) + a, k);
    }
}
__device__ __forceinline__ double ordered_to_double(unsigned long long k) {
    k = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)k);
}

__global__ void voxel_key_kernel(int n, const double* __restrict__ pts, const double* __restrict__ mm, double vs,
                                 unsigned long long* __restrict__ keys, int* __restrict__ vals, int* __restrict__ err) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double mb = ordered_to_double(reinterpret_cast<const unsigned long long*>(mm)[a]) - vs * 0.5;  // downsample.py:47
        const double r = (pts[i * 3 + a] - mb) / vs;  // a true division, as Open3D does
        const long long c = (long long)floor(r);
        if (c < 0 || c >= (1ll << 21)) { atomicExch(err, 1); }
        key = (key << 21) | (unsigned long long)(c & 0x1FFFFF);
    }
    keys[i] = key;
    vals[i] = (int)i;
}

__global__ void voxel_heads_kernel(int n, const unsigned long long* __restrict__ keys, int* __restrict__ flags) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// heads[v] = first sorted position of voxel v (from the inclusive scan of the flags)
__global__ void voxel_starts_kernel(int n, const int* __restrict__ flags, const int* __restrict__ scan, int* __restrict__ starts,
                                    int* __restrict__ nvox) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) starts[scan[i] - 1] = (int)i;
    if (i == n - 1) { *nvox = scan[i]; starts[scan[i]] = n; }
}

constexpr int kVoxMaxLabel = 64;

__global__ void voxel_reduce_kernel(const int* __restrict__ nvox_p, const int* __restrict__ starts, const int* __restrict__ order,
                                    const double* __restrict__ pts, const double* __restrict__ colors,
                                    const int* __restrict__ labels, double* __restrict__ out_pts, double* __restrict__ out_col,
                                    int* __restrict__ out_lab, int* __restrict__ err) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= *nvox_p) return;
    const int s0 = starts[v], s1 = starts[v + 1];
    double p[3] = {0, 0, 0}, c[3] = {0, 0, 0};
    for (int s = s0; s < s1; ++s) {  // the stable sort kept the members in input order: same float64 sums as Open3D
        const int k = order[s];
#pragma unroll
        for (int a = 0; a < 3; ++a) { p[a] += pts[(size_t)k * 3 + a]; if (colors) c[a] += colors[(size_t)k * 3 + a]; }
    }
    const double cntd = (double)(s1 - s0);
#pragma unroll
    for (int a = 0; a < 3; ++a) { out_pts[(size_t)v * 3 + a] = p[a] / cntd; if (out_col) out_col[(size_t)v * 3 + a] = colors ? c[a] / cntd : 0.0; }
    if (labels && out_lab) {
        // majority label, ties -> smallest (np.bincount(...).argmax()): labels are small non-negative ints
        int best = 0, bestc = -1;
        for (int L = 0; L < kVoxMaxLabel; ++L) {
            int cl = 0;
            for (int s = s0; s < s1; ++s) cl += (labels[order[s]] == L);
            if (cl > bestc) { bestc = cl; best = L; }
        }
        int covered = 0;
        for (int s = s0; s < s1; ++s) { const int L = labels[order[s]]; covered += (L >= 0 && L < kVoxMaxLabel); }
        if (covered != s1 - s0) atomicExch(err, 2);
        out_lab[v] = best;
    }
}

struct VoxLayout {
    size_t keys_in, keys_out, vals_in, vals_out, flags, scan, starts, mm, misc, cub, total;
    size_t cub_bytes;
};
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline VoxLayout vox_layout(int n) {
    VoxLayout L;
    size_t o = 0;
    L.keys_in = o; o += al256((size_t)n * 8);
    L.keys_out = o; o += al256((size_t)n * 8);
    L.vals_in = o; o += al256((size_t)n * 4);
    L.vals_out = o; o += al256((size_t)n * 4);
    L.flags = o; o += al256((size_t)n * 4);
    L.scan = o; o += al256((size_t)n * 4);
    L.starts = o; o += al256((size_t)(n + 1) * 4);
    L.mm = o; o += 256;
    L.misc = o; o += 256;
    size_t c1 = 0, c2 = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, c1, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                       (const int*)nullptr, (int*)nullptr, n, 0, 63);
    hipcub::DeviceScan::InclusiveSum(nullptr, c2, (const int*)nullptr, (int*)nullptr, n);
    L.cub_bytes = c1 > c2 ? c1 : c2;
    L.cub = o; o += al256(L.cub_bytes);
    L.total = o;
    return L;
}

}  // namespace

// semantic_dataset.py:123-163 (_extract_z_box) for `b` centre points at once.  points (n,3) float64 sorted by x
// (as SemanticFileData.__init__ leaves them, :84-88); centers (b,3) float64 = the drawn centre points;
// half_x / half_y = box_size/2; scene_z_size = max z - min z of the scene (:132).  out_idx (b,cap), out_cnt (b).
extern "C" int pn2_scene_extract_z_box(int n, const double* points, int b, const double* centers, double half_x,
                                       double half_y, double scene_z_size, int cap, int* out_idx, int* out_cnt,
                                       void* stream) {
    if (n <= 0 || b <= 0 || cap <= 0) return PN2_EINVAL;
    if (!points || !centers || !out_idx || !out_cnt) return PN2_ENULL;
    scene_extract_kernel<<<b, kScT, 0, static_cast<hipStream_t>(stream)>>>(n, points, centers, half_x, half_y, scene_z_size,
                                                                            cap, out_idx, out_cnt);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// semantic_dataset.py:90-121,165-186 (fixed-size sampling, gathers, centring) for the b columns found above.
// mask (b,cap) bytes: for a column with cnt > npts its first cnt bytes are the reference's shuffled boolean sample
// mask (exactly npts non-zero); ignored for columns with cnt <= npts (may be NULL if there are none).
// status (b): 0 ok, 1 empty column, 2 cnt > cap, 3 mask does not select exactly npts points.
extern "C" int pn2_scene_sample(int b, int npts, int cap, const double* points, const int* labels, const double* colors,
                                const int* idx, const int* cnt, const unsigned char* mask, double half_x, double half_y,
                                int* out_sel, float* out_centered, double* out_raw, int* out_labels, float* out_colors,
                                int* status, void* stream) {
    if (b <= 0 || npts <= 0 || cap <= 0) return PN2_EINVAL;
    if (!points || !idx || !cnt || !out_sel || !out_centered || !status) return PN2_ENULL;
    scene_sample_kernel<<<b, kScT, 0, static_cast<hipStream_t>(stream)>>>(npts, cap, points, labels, colors, idx, cnt, mask,
                                                                           half_x, half_y, out_sel, out_centered, out_raw,
                                                                           out_labels, out_colors, status);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" size_t pn2_voxel_downsample_workspace_bytes(int n) { return n > 0 ? vox_layout(n).total : 0; }

// downsample.py:46-67.  points/colors (n,3) float64, labels (n) int32 in [0,64) or NULL.  Outputs sized for n voxels:
// out_points / out_colors (n,3) float64, out_labels (n); *out_count (device int) = number of voxels; voxels sorted by
// (ix,iy,iz).  status (device int): 0 ok, 1 voxel index outside 21 bits per axis, 2 label outside [0,64).
extern "C" int pn2_voxel_downsample(int n, const double* points, const double* colors, const int* labels, double voxel_size,
                                    double* out_points, double* out_colors, int* out_labels, int* out_count, int* status,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (n <= 0 || !(voxel_size > 0)) return PN2_EINVAL;
    if (!points || !out_points || !out_count || !status || !workspace) return PN2_ENULL;
    const VoxLayout L = vox_layout(n);
    if (workspace_bytes < L.total || ((uintptr_t)workspace & 255) != 0) return PN2_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(ws + L.keys_in);
    unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(ws + L.keys_out);
    int* vals_in = reinterpret_cast<int*>(ws + L.vals_in);
    int* vals_out = reinterpret_cast<int*>(ws + L.vals_out);
    int* flags = reinterpret_cast<int*>(ws + L.flags);
    int* scan = reinterpret_cast<int*>(ws + L.scan);
    int* starts = reinterpret_cast<int*>(ws + L.starts);
    double* mm = reinterpret_cast<double*>(ws + L.mm);
    hipError_t e = hipMemsetAsync(mm, 0xFF, 24, st);  // min keys: all ones
    if (e == hipSuccess) e = hipMemsetAsync(mm + 3, 0, 24, st);  // max keys: zero
    if (e == hipSuccess) e = hipMemsetAsync(status, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    const int blocks = (n + 255) / 256;
    voxel_minmax_kernel<<<blocks < 1024 ? blocks : 1024, 256, 0, st>>>(n, points, mm);
    voxel_key_kernel<<<blocks, 256, 0, st>>>(n, points, mm, voxel_size, keys_in, vals_in, status);
    size_t cub = L.cub_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub, keys_in, keys_out, vals_in, vals_out, n, 0, 63, st);
    if (e != hipSuccess) return (int)e;
    voxel_heads_kernel<<<blocks, 256, 0, st>>>(n, keys_out, flags);
    cub = L.cub_bytes;
    e = hipcub::DeviceScan::InclusiveSum(ws + L.cub, cub, flags, scan, n, st);
    if (e != hipSuccess) return (int)e;
    voxel_starts_kernel<<<blocks, 256, 0, st>>>(n, flags, scan, starts, out_count);
    voxel_reduce_kernel<<<blocks, 256, 0, st>>>(out_count, starts, vals_out, points, colors, labels, out_points, out_colors,
                                                out_labels, status);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
