// pn2_interpolate.hip -- three_nn, three_interpolate (+grad) and the fused FP
// front end for gfx950.  MI355X-native replacements for the CPU-only ops of
// tf_ops/tf_interpolate.cpp:213-243,307-330,397-421 (which bounce every FP layer
// GPU->host->GPU in the reference) and of util/pointnet_util.py:300-311.
//
// three_nn is exact float64 like the reference's KD-tree (tf_interpolate.cpp:20-28:
// points widened to Eigen::Vector3d; FLANN L2<double> accumulates
// ((0+dx*dx)+dy*dy)+dz*dz), brute force with a rigorous fp32 ranking prefilter
// (3 FMAs per pair) and float64 evaluation of the few survivors.  Bound: VALU
// issue, not HBM.
#include <math.h>

#include "pn2_common.h"
#include "pn2_three_nn.h"

namespace {

using namespace pn2nn;

PN2_TUNABLE(int, g_nn_blocks, 2048)  // target workgroups per launch (sweep: 16384 = one group per wave 42.1 us, 2048 39.6, 1024 43.0, 512 45.8); tuning hook 12

// PERSISTENT waves (r03): a wave keeps the candidates in registers (m <= 64 * kNnChunks: the common case) and walks
// query groups grp, grp + stride, ... (pn2nn::three_nn_wave); grid.y = batch.
template <int kNnChunks>
__global__ void __launch_bounds__(kNnThreads)
three_nn_kernel(int n, int m, const float* __restrict__ xyz1_all, int ld1,
                const float* __restrict__ xyz2_all, float* __restrict__ dist_all,
                int* __restrict__ idx_all) {
    __shared__ __attribute__((aligned(16))) unsigned char slds[kNnWaves * kNnWaveLdsBytes];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    three_nn_wave<kNnChunks>(n, m, xyz1_all + (size_t)bi * n * ld1, xyz2_all + (size_t)bi * m * 3, dist_all + (size_t)bi * n * 3,
                             idx_all + (size_t)bi * n * 3, blockIdx.x * kNnWaves + wave, gridDim.x * kNnWaves,
                             nn_wave_lds(slds, wave), ld1);
}

// out[row, :] = (p1*w1 + p2*w2) + p3*w3, unfused fp32 (tf_interpolate.cpp:322-324).
// grid.y = batch, e indexes the n*c/VEC vector elements of one batch element.
template <typename VT, int VEC>
__global__ void __launch_bounds__(256)
three_interpolate_kernel(int m, int c, int n, const float* __restrict__ points_all,
                         const int* __restrict__ idx_all, const float* __restrict__ weight_all,
                         float* __restrict__ out_all) {
    const unsigned cv = (unsigned)c / VEC;
    const unsigned total = (unsigned)n * cv;
    const int bi = blockIdx.y;
    const VT* __restrict__ pts = reinterpret_cast<const VT*>(points_all + (size_t)bi * m * c);
    const int* __restrict__ idx = idx_all + (size_t)bi * n * 3;
    const float* __restrict__ w = weight_all + (size_t)bi * n * 3;
    VT* __restrict__ out = reinterpret_cast<VT*>(out_all + (size_t)bi * n * c);
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned row = e / cv, col = e - row * cv;
        const int i1 = idx[row * 3 + 0], i2 = idx[row * 3 + 1], i3 = idx[row * 3 + 2];
        const float w1 = w[row * 3 + 0], w2 = w[row * 3 + 1], w3 = w[row * 3 + 2];
        const VT p1 = pts[(size_t)i1 * cv + col], p2 = pts[(size_t)i2 * cv + col], p3 = pts[(size_t)i3 * cv + col];
        out[e] = (p1 * w1 + p2 * w2) + p3 * w3;
    }
}

template <typename VT, int VEC>
__global__ void __launch_bounds__(256)
three_interpolate_grad_kernel(int n, int c, int m, const float* __restrict__ grad_out_all,
                              const int* __restrict__ idx_all, const float* __restrict__ weight_all,
                              float* __restrict__ grad_points_all) {
    const unsigned cv = (unsigned)c / VEC;
    const unsigned total = (unsigned)n * cv;
    const int bi = blockIdx.y;
    const VT* __restrict__ go = reinterpret_cast<const VT*>(grad_out_all + (size_t)bi * n * c);
    const int* __restrict__ idx = idx_all + (size_t)bi * n * 3;
    const float* __restrict__ w = weight_all + (size_t)bi * n * 3;
    float* __restrict__ gp = grad_points_all + (size_t)bi * m * c;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned row = e / cv, col = e - row * cv;
        const VT g = go[e];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int ii = idx[row * 3 + t];
            const float wt = w[row * 3 + t];
            float* dst = gp + (size_t)ii * c + col * VEC;
            if constexpr (VEC == 4) {
                atomicAdd(dst + 0, g.x * wt); atomicAdd(dst + 1, g.y * wt);
                atomicAdd(dst + 2, g.z * wt); atomicAdd(dst + 3, g.w * wt);
            } else {
                atomicAdd(dst, g * wt);  // tf_interpolate.cpp:411-415
            }
        }
    }
}

// ---- three_interpolate gradient without float atomics (training path) -------------------------------------
// The scatter above issues 3*c float atomics per query row (50 M for the l1 -> l0 level of the training step: 690 us,
// all of it atomic throughput).  Inverted: a CSR list of (query, weight) per SOURCE point is built with 3*n integer
// atomics per cloud, then each source row is produced by one group of lanes that walks its list and reads the grad_out
// rows with 16-byte loads -- 3x the grad_out bytes, mostly from L2, and one plain store per output element.
__global__ void __launch_bounds__(256)
ti_csr_count_kernel(int n3, int m, const int* __restrict__ idx_all, int* __restrict__ cnt_all) {
    const int bi = blockIdx.y;
    const int* __restrict__ idx = idx_all + (size_t)bi * n3;
    int* __restrict__ cnt = cnt_all + (size_t)bi * m;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n3; e += gridDim.x * blockDim.x) atomicAdd(&cnt[idx[e]], 1);
}

// one block per cloud: off = exclusive scan of cnt; cnt is cleared to serve as the fill cursor
__global__ void __launch_bounds__(256)
ti_csr_scan_kernel(int m, int* __restrict__ cnt_all, int* __restrict__ off_all) {
    __shared__ int part[256];
    const int bi = blockIdx.x, t = threadIdx.x;
    int* __restrict__ cnt = cnt_all + (size_t)bi * m;
    int* __restrict__ off = off_all + (size_t)bi * m;
    const int per = (m + 255) / 256;
    const int lo = t * per < m ? t * per : m, hi = lo + per < m ? lo + per : m;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan of the 256 partial sums
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = lo; i < hi; ++i) {
        const int v = cnt[i];
        off[i] = run;
        cnt[i] = 0;
        run += v;
    }
}

__global__ void __launch_bounds__(256)
ti_csr_fill_kernel(int n3, int m, int div, const int* __restrict__ idx_all, const float* __restrict__ weight_all,
                   int weight_kind, const int* __restrict__ off_all, int* __restrict__ cur_all, int* __restrict__ ent_q_all,
                   float* __restrict__ ent_w_all) {
    const int bi = blockIdx.y;
    const int* __restrict__ idx = idx_all + (size_t)bi * n3;
    const float* __restrict__ w = weight_all ? weight_all + (size_t)bi * n3 : nullptr;
    const int* __restrict__ off = off_all + (size_t)bi * m;
    int* __restrict__ cur = cur_all + (size_t)bi * m;
    int* __restrict__ eq = ent_q_all + (size_t)bi * n3;
    float* __restrict__ ew = ent_w_all + (size_t)bi * n3;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n3; e += gridDim.x * blockDim.x) {
        const int s = idx[e];
        const int p = off[s] + atomicAdd(&cur[s], 1);
        eq[p] = e / div;   // div entries per input row (3 for three_interpolate, 1 for group_point)
        float wt = 1.f;
        if (w && weight_kind == 2) {
            // w holds three_nn's squared distances (n, 3): the inverse-distance weights of pointnet_util.py:300-303 with the
            // float expressions of the forward kernels (fp_interp_concat_*): IEEE divisions, (r1 + r2) + r3
            const int r = e / 3, j = e - 3 * r;
            const float r1 = 1.0f / fmaxf(w[r * 3 + 0], 1e-10f), r2 = 1.0f / fmaxf(w[r * 3 + 1], 1e-10f);
            const float r3 = 1.0f / fmaxf(w[r * 3 + 2], 1e-10f);
            const float norm = (r1 + r2) + r3;
            wt = (j == 0 ? r1 : (j == 1 ? r2 : r3)) / norm;
        } else if (w) {
            wt = w[e];
        }
        ew[p] = wt;
    }
}

// thread -> (source slot, float4 column); each slot walks the list of its source point, four entries in flight.
// Input rows are `stride` floats apart (>= c: the caller may hand over a column slice of a wider gradient in place);
// ALIGNED = every row starts on a 16-byte boundary, else the 16-byte loads are issued as 4-byte-aligned ones.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
template <bool ALIGNED>
__global__ void __launch_bounds__(256)
ti_csr_gather_kernel(int n, int n3, int c, int stride, int m, const float* __restrict__ grad_out_all, const int* __restrict__ off_all,
                     const int* __restrict__ ent_q_all, const float* __restrict__ ent_w_all,
                     float* __restrict__ grad_points_all) {
    const int cv = c >> 2;
    const int spb = 256 / cv;  // sources per block (host guarantees cv <= 256)
    const int slot = (int)threadIdx.x / cv, col = (int)threadIdx.x - slot * cv;
    const int bi = blockIdx.y;
    const int s = blockIdx.x * spb + slot;
    if (slot >= spb || s >= m) return;
    const float* __restrict__ gbase = grad_out_all + (size_t)bi * n * stride + 4 * col;
    auto row4 = [&](int q) -> f32x4 {
        if constexpr (ALIGNED) return *reinterpret_cast<const f32x4*>(gbase + (size_t)q * stride);
        else return *reinterpret_cast<const f32x4_u*>(gbase + (size_t)q * stride);
    };
    const int* __restrict__ off = off_all + (size_t)bi * m;
    const int* __restrict__ eq = ent_q_all + (size_t)bi * n3;
    const float* __restrict__ ew = ent_w_all + (size_t)bi * n3;
    const int lo = off[s], hi = s + 1 < m ? off[s + 1] : n3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int e = lo;
    for (; e + 3 < hi; e += 4) {
        int q[4]; float w[4]; f32x4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { q[u] = eq[e + u]; w[u] = ew[e + u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = row4(q[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc.x = __builtin_fmaf(g[u].x, w[u], acc.x); acc.y = __builtin_fmaf(g[u].y, w[u], acc.y);
            acc.z = __builtin_fmaf(g[u].z, w[u], acc.z); acc.w = __builtin_fmaf(g[u].w, w[u], acc.w);
        }
    }
    for (; e < hi; ++e) {
        const f32x4 g = row4(eq[e]);
        const float w = ew[e];
        acc.x = __builtin_fmaf(g.x, w, acc.x); acc.y = __builtin_fmaf(g.y, w, acc.y);
        acc.z = __builtin_fmaf(g.z, w, acc.z); acc.w = __builtin_fmaf(g.w, w, acc.w);
    }
    reinterpret_cast<f32x4*>(grad_points_all + ((size_t)bi * m + s) * c)[col] = acc;
}

// Fused FP front end: inverse-distance weights (pointnet_util.py:300-303) +
// three_interpolate + concat([interp, points1]) (pointnet_util.py:304-311).
// One wave owns 64 consecutive rows: lane l computes the three weights of row l once (IEEE
// divisions), then the rows are produced one after another with that row's (idx, weight) broadcast
// through SGPRs (v_readlane) and the lanes striding over the channels -- every global access is a
// contiguous 256-byte segment even for odd row widths such as 131.
__global__ void __launch_bounds__(256)
fp_interp_concat_kernel(int n, int m, int c1, int c2, int ostride, int rpw, const float* __restrict__ dist_all,
                        const int* __restrict__ idx_all, const float* __restrict__ points1_all,
                        const float* __restrict__ points2_all, float* __restrict__ out_all) {
    // rpw = rows per wave (1..64, chosen by the host so that small levels still fill the chip)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    const int cw = ostride;  // row stride of the output (>= c1 + c2; the pad columns are zero-filled)
    const float* __restrict__ dist = dist_all + (size_t)bi * n * 3;
    const int* __restrict__ idx = idx_all + (size_t)bi * n * 3;
    const float* __restrict__ p2 = points2_all + (size_t)bi * m * c2;
    const float* __restrict__ p1 = points1_all ? points1_all + (size_t)bi * n * c1 : nullptr;
    float* __restrict__ out = out_all + (size_t)bi * n * cw;
    for (int row0 = (blockIdx.x * 4 + wave) * rpw; row0 < n; row0 += gridDim.x * 4 * rpw) {
        int r = row0 + (lane < rpw ? lane : rpw - 1);
        r = r < n ? r : n - 1;
        const float d1 = fmaxf(dist[r * 3 + 0], 1e-10f);
        const float d2 = fmaxf(dist[r * 3 + 1], 1e-10f);
        const float d3 = fmaxf(dist[r * 3 + 2], 1e-10f);
        const float r1 = 1.0f / d1, r2 = 1.0f / d2, r3 = 1.0f / d3;  // IEEE division
        const float norm = (r1 + r2) + r3;
        const float w1 = r1 / norm, w2 = r2 / norm, w3 = r3 / norm;
        const int i1 = idx[r * 3 + 0], i2 = idx[r * 3 + 1], i3 = idx[r * 3 + 2];
        const int nrows = n - row0 < rpw ? n - row0 : rpw;
        auto one_row = [&](int rr) {
            const float a1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w1), rr));
            const float a2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w2), rr));
            const float a3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w3), rr));
            const float* __restrict__ s1 = p2 + (size_t)__builtin_amdgcn_readlane(i1, rr) * c2;
            const float* __restrict__ s2 = p2 + (size_t)__builtin_amdgcn_readlane(i2, rr) * c2;
            const float* __restrict__ s3 = p2 + (size_t)__builtin_amdgcn_readlane(i3, rr) * c2;
            float* __restrict__ o = out + (size_t)(row0 + rr) * cw;
#pragma unroll 4
            for (int ch = lane; ch < c2; ch += 64) o[ch] = (s1[ch] * a1 + s2[ch] * a2) + s3[ch] * a3;
            if (p1) {
                const float* __restrict__ q = p1 + (size_t)(row0 + rr) * c1;
                for (int ch = lane; ch < c1; ch += 64) o[c2 + ch] = q[ch];
            }
            if (lane < cw - c1 - c2) o[c1 + c2 + lane] = 0.f;  // zero the (< 8) pad columns
        };
        int rr = 0;
        for (; rr + 1 < nrows; rr += 2) { one_row(rr); one_row(rr + 1); }
        if (rr < nrows) one_row(rr);
    }
}

// Same operation with R rows in flight per wave: all gathers of R rows are issued before the first
// store (the row-at-a-time kernel above serialises load -> store -> load because the stores may alias
// the loads), every load is unconditional (clamped column, result selected afterwards).  CHW =
// ceil(ostride / 64) column chunks per row.
template <int CHW, int R>
__global__ void __launch_bounds__(256)
fp_interp_concat_rows_kernel(int n, int m, int c1, int c2, int ostride, int rpw, const float* __restrict__ dist_all,
                             const int* __restrict__ idx_all, const float* __restrict__ points1_all,
                             const float* __restrict__ points2_all, float* __restrict__ out_all) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware block remap (speed only): the dispatcher places workgroup L on XCD L % 8; giving every
    // XCD a contiguous range of logical blocks = whole batch elements keeps each private 4 MiB L2
    // gathering from its own 1/8 of points2 instead of from all of it.
    int bx = blockIdx.x, bi = blockIdx.y;
    {
        const unsigned nwg = gridDim.x * gridDim.y;
        if ((nwg & 7u) == 0u) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned swz = (lin & 7u) * (nwg >> 3) + (lin >> 3);
            bx = (int)(swz % gridDim.x);
            bi = (int)(swz / gridDim.x);
        }
    }
    const int cw = ostride;
    const float* __restrict__ dist = dist_all + (size_t)bi * n * 3;
    const int* __restrict__ idx = idx_all + (size_t)bi * n * 3;
    const float* __restrict__ p2 = points2_all + (size_t)bi * m * c2;
    const float* __restrict__ p1 = points1_all ? points1_all + (size_t)bi * n * c1 : p2;  // dummy when c1 == 0
    float* __restrict__ out = out_all + (size_t)bi * n * cw;
    const int c1m = c1 > 0 ? c1 - 1 : 0;
    for (int row0 = (bx * 4 + wave) * rpw; row0 < n; row0 += gridDim.x * 4 * rpw) {
        int r = row0 + (lane < rpw ? lane : rpw - 1);
        r = r < n ? r : n - 1;
        const float d1 = fmaxf(dist[r * 3 + 0], 1e-10f);
        const float d2 = fmaxf(dist[r * 3 + 1], 1e-10f);
        const float d3 = fmaxf(dist[r * 3 + 2], 1e-10f);
        const float r1 = 1.0f / d1, r2 = 1.0f / d2, r3 = 1.0f / d3;  // IEEE division
        const float norm = (r1 + r2) + r3;
        const float w1 = r1 / norm, w2 = r2 / norm, w3 = r3 / norm;
        const int i1 = idx[r * 3 + 0], i2 = idx[r * 3 + 1], i3 = idx[r * 3 + 2];
        const int nrows = n - row0 < rpw ? n - row0 : rpw;
        for (int rr = 0; rr < nrows; rr += R) {
            float v[R][CHW];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int rx = rr + q < nrows ? rr + q : nrows - 1;
                const float a1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w1), rx));
                const float a2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w2), rx));
                const float a3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w3), rx));
                const float* __restrict__ s1 = p2 + (size_t)__builtin_amdgcn_readlane(i1, rx) * c2;
                const float* __restrict__ s2 = p2 + (size_t)__builtin_amdgcn_readlane(i2, rx) * c2;
                const float* __restrict__ s3 = p2 + (size_t)__builtin_amdgcn_readlane(i3, rx) * c2;
                const float* __restrict__ qp = p1 + (size_t)(row0 + rx) * c1;
#pragma unroll
                for (int j = 0; j < CHW; ++j) {
                    const int col = lane + 64 * j;
                    const int cc = col < c2 ? col : c2 - 1;
                    int ci = col - c2;
                    ci = ci < 0 ? 0 : (ci > c1m ? c1m : ci);
                    const float x1 = s1[cc], x2 = s2[cc], x3 = s3[cc], xp = qp[ci];
                    const float val = (x1 * a1 + x2 * a2) + x3 * a3;
                    v[q][j] = col < c2 ? val : (col < c2 + c1 ? xp : 0.f);
                }
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if (rr + q < nrows) {
                    float* __restrict__ o = out + (size_t)(row0 + rr + q) * cw;
#pragma unroll
                    for (int j = 0; j < CHW; ++j) {
                        const int col = lane + 64 * j;
                        if (col < cw) o[col] = v[q][j];
                    }
                }
            }
        }
    }
}

// 16-byte-lane version for the common power-of-two layouts (c2/4 and (ostride-c2)/4 powers of two):
// a wave owns up to 64 consecutive rows; lane l computes the weights of row l once and parks
// (w1,w2,w3,i1,i2,i3) in LDS; then the lanes sweep the (row, float4 column) elements of the interpolated
// part -- three 16-byte gathers, the unfused (p1*w1 + p2*w2) + p3*w3 per component, one 16-byte store
// -- four elements in flight per lane, followed by the [points1 | zero pad] tail columns.
// P1V: points1 rows are 16-byte loadable (c1 % 4 == 0, aligned).
template <bool P1V>
__global__ void __launch_bounds__(256)
fp_interp_concat_v4_kernel(int n, int m, int c1, int c2, int ostride, int rpw, int sh2, int sht,
                           const float* __restrict__ dist_all, const int* __restrict__ idx_all,
                           const float* __restrict__ points1_all, const float* __restrict__ points2_all,
                           float* __restrict__ out_all) {
    __shared__ __attribute__((aligned(16))) float swt[4 * 64 * 8];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx = blockIdx.x, bi = blockIdx.y;
    {   // XCD-aware block remap, see fp_interp_concat_rows_kernel
        const unsigned nwg = gridDim.x * gridDim.y;
        if ((nwg & 7u) == 0u) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned swz = (lin & 7u) * (nwg >> 3) + (lin >> 3);
            bx = (int)(swz % gridDim.x);
            bi = (int)(swz / gridDim.x);
        }
    }
    const int cv = ostride >> 2, c2v = c2 >> 2;
    const float* __restrict__ dist = dist_all + (size_t)bi * n * 3;
    const int* __restrict__ idx = idx_all + (size_t)bi * n * 3;
    const f32x4* __restrict__ p2 = reinterpret_cast<const f32x4*>(points2_all + (size_t)bi * m * c2);
    const float* __restrict__ p1 = points1_all ? points1_all + (size_t)bi * n * c1 : nullptr;
    f32x4* __restrict__ out = reinterpret_cast<f32x4*>(out_all + (size_t)bi * n * ostride);
    float* wv = swt + wave * 64 * 8;
    const unsigned m2 = (1u << sh2) - 1u, mt = (1u << sht) - 1u;
    for (int row0 = (bx * 4 + wave) * rpw; row0 < n; row0 += gridDim.x * 4 * rpw) {
        {
            int r = row0 + (lane < rpw ? lane : rpw - 1);
            r = r < n ? r : n - 1;
            const float d1 = fmaxf(dist[r * 3 + 0], 1e-10f);
            const float d2 = fmaxf(dist[r * 3 + 1], 1e-10f);
            const float d3 = fmaxf(dist[r * 3 + 2], 1e-10f);
            const float r1 = 1.0f / d1, r2 = 1.0f / d2, r3 = 1.0f / d3;  // IEEE division
            const float norm = (r1 + r2) + r3;
            const f32x4 wq = {r1 / norm, r2 / norm, r3 / norm, 0.f};
            const f32x4 iq = {__int_as_float(idx[r * 3 + 0]), __int_as_float(idx[r * 3 + 1]),
                              __int_as_float(idx[r * 3 + 2]), 0.f};
            *reinterpret_cast<f32x4*>(wv + lane * 8) = wq;
            *reinterpret_cast<f32x4*>(wv + lane * 8 + 4) = iq;
        }
        const int nrows = n - row0 < rpw ? n - row0 : rpw;
        // ---- interpolated columns ----
        const unsigned E = (unsigned)nrows << sh2;
        for (unsigned e0 = lane; e0 < E; e0 += 64 * 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned e = e0 + 64 * u;
                e = e < E ? e : E - 1;  // clamped: unconditional loads
                const unsigned row = e >> sh2, col = e & m2;
                const f32x4 wq = *reinterpret_cast<const f32x4*>(wv + row * 8);
                const f32x4 iq = *reinterpret_cast<const f32x4*>(wv + row * 8 + 4);
                const f32x4 x1 = p2[(size_t)__float_as_int(iq[0]) * c2v + col];
                const f32x4 x2 = p2[(size_t)__float_as_int(iq[1]) * c2v + col];
                const f32x4 x3 = p2[(size_t)__float_as_int(iq[2]) * c2v + col];
                v[u] = (x1 * wq[0] + x2 * wq[1]) + x3 * wq[2];  // tf_interpolate.cpp:322-324, unfused
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned e = e0 + 64 * u;
                if (e < E) out[(size_t)(row0 + (e >> sh2)) * cv + (e & m2)] = v[u];
            }
        }
        // ---- [points1 | zero pad] columns ----
        const unsigned ET = (unsigned)nrows << sht;
        for (unsigned e = lane; e < ET; e += 64) {
            const unsigned row = e >> sht, ct = e & mt;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if constexpr (P1V) {
                if ((int)ct * 4 < c1) t = *reinterpret_cast<const f32x4*>(p1 + (size_t)(row0 + row) * c1 + ct * 4);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = (int)ct * 4 + i;
                    if (ch < c1) t[i] = p1[(size_t)(row0 + row) * c1 + ch];
                }
            }
            out[(size_t)(row0 + row) * cv + c2v + ct] = t;
        }
    }
}

inline int grid_x_for(unsigned long long total, int block, int batches) {
    unsigned long long g = (total + block - 1) / block;
    unsigned long long cap = (256ull * 8 + batches - 1) / batches;
    if (cap < 1) cap = 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#ifdef PN2_TUNING_HOOKS
extern "C" int pn2_debug_set_interp(int what, int value) { if (what == 12) { g_nn_blocks = value; return 0; } return -1; }
#endif

// pn2_three_nn with query rows ld1 floats apart (ld1 >= 3: the xyz columns of a (b,n,6) xyz+rgb batch read in place,
// model.py:26-29); bit-identical to pn2_three_nn on a dense copy.
extern "C" int pn2_three_nn_ld(int b, int n, int m, const float* xyz1, int ld1, const float* xyz2, float* dist,
                               int* idx, void* stream) {
    if (b <= 0 || n <= 0 || m < 3 || ld1 < 3) return PN2_EINVAL;
    if (!xyz1 || !xyz2 || !dist || !idx) return PN2_ENULL;
    if ((long long)n * ld1 > 0x7fffffffLL || (long long)m * 3 > 0x7fffffffLL || b > 65535) return PN2_ERANGE;
    // persistent waves: ~4 workgroups per CU over the whole batch, every wave takes the same number of query groups
    const int blocks_all = (n + kNnWaves * kNnQ - 1) / (kNnWaves * kNnQ);
    int per = (b * blocks_all + g_nn_blocks - 1) / g_nn_blocks;        // groups per wave
    if (per < 1) per = 1;
    dim3 grid((blocks_all + per - 1) / per, b);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (m <= 64) three_nn_kernel<1><<<grid, kNnThreads, 0, st>>>(n, m, xyz1, ld1, xyz2, dist, idx);
    else if (m <= 256) three_nn_kernel<4><<<grid, kNnThreads, 0, st>>>(n, m, xyz1, ld1, xyz2, dist, idx);
    else three_nn_kernel<16><<<grid, kNnThreads, 0, st>>>(n, m, xyz1, ld1, xyz2, dist, idx);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_three_nn(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist,
                            int* idx, void* stream) {
    return pn2_three_nn_ld(b, n, m, xyz1, 3, xyz2, dist, idx, stream);
}

extern "C" int pn2_three_interpolate(int b, int m, int c, int n, const float* points, const int* idx,
                                     const float* weight, float* out, void* stream) {
    if (b <= 0 || m <= 0 || c <= 0 || n <= 0) return PN2_EINVAL;
    if (!points || !idx || !weight || !out) return PN2_ENULL;
    if ((unsigned long long)n * c > 0xffffffffull || b > 65535) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec4 = (c % 4 == 0) && (((uintptr_t)points | (uintptr_t)out) % 16 == 0);
    if (vec4) {
        dim3 grid(grid_x_for((unsigned long long)n * (c / 4), 256, b), b);
        three_interpolate_kernel<f32x4, 4><<<grid, 256, 0, st>>>(m, c, n, points, idx, weight, out);
    } else {
        dim3 grid(grid_x_for((unsigned long long)n * c, 256, b), b);
        three_interpolate_kernel<float, 1><<<grid, 256, 0, st>>>(m, c, n, points, idx, weight, out);
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_three_interpolate_grad(int b, int n, int c, int m, const float* grad_out,
                                          const int* idx, const float* weight, float* grad_points,
                                          void* stream) {
    if (b <= 0 || m <= 0 || c <= 0 || n <= 0) return PN2_EINVAL;
    if (!grad_out || !idx || !weight || !grad_points) return PN2_ENULL;
    if ((unsigned long long)n * c > 0xffffffffull || b > 65535) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * m * c, st);
    if (e != hipSuccess) return (int)e;
    const bool vec4 = (c % 4 == 0) && ((uintptr_t)grad_out % 16 == 0);
    if (vec4) {
        dim3 grid(grid_x_for((unsigned long long)n * (c / 4), 256, b), b);
        three_interpolate_grad_kernel<f32x4, 4><<<grid, 256, 0, st>>>(n, c, m, grad_out, idx, weight, grad_points);
    } else {
        dim3 grid(grid_x_for((unsigned long long)n * c, 256, b), b);
        three_interpolate_grad_kernel<float, 1><<<grid, 256, 0, st>>>(n, c, m, grad_out, idx, weight, grad_points);
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// out[b, idx[b,e], :] += weight[b,e] * rows_in[b, e/div, :] for e < nent, as a list build + gather (see ti_csr_*).
// rows_in (b, nent/div, c), out (b, nsrc, c) overwritten.  The list depends on (idx, weight) only -- geometry -- so a
// training step builds it ahead (pn2_scatter_plan_build, beside the previous step's dense work) and its backward pass
// only gathers (pn2_scatter_plan_apply).  Plan layout (ints): cursor[b][nsrc] | offset[b][nsrc] | entry row[b][nent] |
// entry weight[b][nent].
extern "C" size_t pn2_scatter_plan_bytes(int b, int nent, int nsrc) {
    if (b <= 0 || nent <= 0 || nsrc <= 0) return 0;
    return sizeof(int) * ((size_t)2 * b * nsrc + (size_t)2 * b * nent);
}

extern "C" int pn2_scatter_plan_build(int b, int nent, int div, int nsrc, const int* idx, const float* weight, int weight_kind,
                                      void* plan, size_t plan_bytes, void* stream) {
    if (b <= 0 || nent <= 0 || nsrc <= 0 || div <= 0 || nent % div != 0) return PN2_EINVAL;
    if (weight_kind < 0 || weight_kind > 2 || (weight_kind == 2 && div != 3)) return PN2_EINVAL;
    if (!idx || !plan || (weight_kind != 0 && !weight)) return PN2_ENULL;
    if (b > 65535) return PN2_ERANGE;
    if (plan_bytes < pn2_scatter_plan_bytes(b, nent, nsrc) || ((uintptr_t)plan % 4) != 0) return PN2_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int* cnt = static_cast<int*>(plan);
    int* off = cnt + (size_t)b * nsrc;
    int* ent_q = off + (size_t)b * nsrc;
    float* ent_w = reinterpret_cast<float*>(ent_q + (size_t)b * nent);
    hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)b * nsrc, st);
    if (e != hipSuccess) return (int)e;
    dim3 ge(grid_x_for((unsigned long long)nent, 256, b), b);
    ti_csr_count_kernel<<<ge, 256, 0, st>>>(nent, nsrc, idx, cnt);
    ti_csr_scan_kernel<<<b, 256, 0, st>>>(nsrc, cnt, off);
    ti_csr_fill_kernel<<<ge, 256, 0, st>>>(nent, nsrc, div, idx, weight_kind ? weight : nullptr, weight_kind, off, cnt, ent_q, ent_w);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// ---- several plans in one set of launches ------------------------------------------------------------------------------------
// A training step builds SEVEN plans per batch (three grouping levels, four interpolation levels) on the geometry stream beside
// the previous step's dense work: 7 x (memset, count, scan, fill) = 28 small launches whose only cost is being launches next to a
// busy graph (measured: +0.09 ms per step).  Here the plans are slices of ONE buffer and every kernel takes the plan index from
// blockIdx.z: one memset + three launches.
constexpr int kMaxPlans = 8;
struct PlanDesc {
    int nent, nsrc, div, kind;
    const int* idx;
    const float* w;
    int* cnt;
    int* off;
    int* eq;
    float* ew;
};
struct PlanBatch {
    PlanDesc d[kMaxPlans];
};

__global__ void __launch_bounds__(256)
ti_csr_count_multi_kernel(PlanBatch pb) {
    const PlanDesc& p = pb.d[blockIdx.z];
    const int bi = blockIdx.y;
    const int* __restrict__ idx = p.idx + (size_t)bi * p.nent;
    int* __restrict__ cnt = p.cnt + (size_t)bi * p.nsrc;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < p.nent; e += gridDim.x * blockDim.x) atomicAdd(&cnt[idx[e]], 1);
}

__global__ void __launch_bounds__(256)
ti_csr_scan_multi_kernel(PlanBatch pb) {
    __shared__ int part[256];
    const PlanDesc& p = pb.d[blockIdx.y];
    const int bi = blockIdx.x, t = threadIdx.x, m = p.nsrc;
    int* __restrict__ cnt = p.cnt + (size_t)bi * m;
    int* __restrict__ off = p.off + (size_t)bi * m;
    const int per = (m + 255) / 256;
    const int lo = t * per < m ? t * per : m, hi = lo + per < m ? lo + per : m;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // as ti_csr_scan_kernel
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = lo; i < hi; ++i) {
        const int v = cnt[i];
        off[i] = run;
        cnt[i] = 0;
        run += v;
    }
}

__global__ void __launch_bounds__(256)
ti_csr_fill_multi_kernel(PlanBatch pb) {
    const PlanDesc& p = pb.d[blockIdx.z];
    const int bi = blockIdx.y, n3 = p.nent, m = p.nsrc;
    const int* __restrict__ idx = p.idx + (size_t)bi * n3;
    const float* __restrict__ w = p.kind ? p.w + (size_t)bi * n3 : nullptr;
    const int* __restrict__ off = p.off + (size_t)bi * m;
    int* __restrict__ cur = p.cnt + (size_t)bi * m;
    int* __restrict__ eq = p.eq + (size_t)bi * n3;
    float* __restrict__ ew = p.ew + (size_t)bi * n3;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n3; e += gridDim.x * blockDim.x) {
        const int s_ = idx[e];
        const int q = off[s_] + atomicAdd(&cur[s_], 1);
        eq[q] = e / p.div;
        float wt = 1.f;
        if (w && p.kind == 2) {  // three_nn's squared distances -> inverse-distance weights, as ti_csr_fill_kernel
            const int r = e / 3, j = e - 3 * r;
            const float r1 = 1.0f / fmaxf(w[r * 3 + 0], 1e-10f), r2 = 1.0f / fmaxf(w[r * 3 + 1], 1e-10f);
            const float r3 = 1.0f / fmaxf(w[r * 3 + 2], 1e-10f);
            const float norm = (r1 + r2) + r3;
            wt = (j == 0 ? r1 : (j == 1 ? r2 : r3)) / norm;
        } else if (w) {
            wt = w[e];
        }
        ew[q] = wt;
    }
}

// The three kernels above as ONE, for plans whose source counts fit LDS: a workgroup owns one (plan, cloud) pair -- counts with LDS
// atomics, an exclusive scan over the <= kPlanLdsMaxSrc counters in LDS, then the fill with LDS cursors.  The global atomics of the
// three-kernel form (1.1 M of them for the seven plans of a configs[1] training step, executed at the memory side: 43 + 62 us on
// the geometry stream, and they slow the training step beside them by almost their whole duration) become LDS atomics; cnt / off
// are written once, so the range needs no memset.  The order inside a source's list is as arbitrary as before.
constexpr int kPlanLdsMaxSrc = 16384;
constexpr int kPlanLdsThreads = 1024;
__global__ void __launch_bounds__(kPlanLdsThreads)
ti_csr_build_lds_kernel(PlanBatch pb) {
    extern __shared__ int plan_lds[];
    __shared__ int wsum[kPlanLdsThreads / 64];
    const PlanDesc& p = pb.d[blockIdx.y];
    const int bi = blockIdx.x, t = threadIdx.x, m = p.nsrc, n3 = p.nent;
    int* __restrict__ cnt = plan_lds;       // [m] counters, then cursors
    int* __restrict__ off = plan_lds + m;   // [m] list starts
    const int* __restrict__ idx = p.idx + (size_t)bi * n3;
    for (int i = t; i < m; i += kPlanLdsThreads) cnt[i] = 0;
    __syncthreads();
    for (int e = t; e < n3; e += kPlanLdsThreads) atomicAdd(&cnt[idx[e]], 1);
    __syncthreads();
    // exclusive scan: thread t owns the counters [t * per, (t + 1) * per)
    const int per = (m + kPlanLdsThreads - 1) / kPlanLdsThreads;
    const int lo = t * per < m ? t * per : m, hi = lo + per < m ? lo + per : m;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    int inc = sum;  // inclusive scan inside the wave, then over the 16 wave totals
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w_ = 0; w_ < wave; ++w_) base += wsum[w_];
    int run = base + inc - sum;
    int* __restrict__ gcnt = p.cnt + (size_t)bi * m;
    int* __restrict__ goff = p.off + (size_t)bi * m;
    for (int i = lo; i < hi; ++i) {
        const int v = cnt[i];
        off[i] = run;
        goff[i] = run;
        gcnt[i] = v;   // (the three-kernel form leaves the list lengths here too)
        cnt[i] = 0;
        run += v;
    }
    __syncthreads();
    const float* __restrict__ w = p.kind ? p.w + (size_t)bi * n3 : nullptr;
    int* __restrict__ eq = p.eq + (size_t)bi * n3;
    float* __restrict__ ew = p.ew + (size_t)bi * n3;
    for (int e = t; e < n3; e += kPlanLdsThreads) {
        const int s_ = idx[e];
        const int q = off[s_] + atomicAdd(&cnt[s_], 1);
        eq[q] = e / p.div;
        float wt = 1.f;
        if (w && p.kind == 2) {  // three_nn's squared distances -> inverse-distance weights, as ti_csr_fill_kernel
            const int r = e / 3, j = e - 3 * r;
            const float r1 = 1.0f / fmaxf(w[r * 3 + 0], 1e-10f), r2 = 1.0f / fmaxf(w[r * 3 + 1], 1e-10f);
            const float r3 = 1.0f / fmaxf(w[r * 3 + 2], 1e-10f);
            const float norm = (r1 + r2) + r3;
            wt = (j == 0 ? r1 : (j == 1 ? r2 : r3)) / norm;
        } else if (w) {
            wt = w[e];
        }
        ew[q] = wt;
    }
}

// pn2_scatter_plan_build for nplans <= 8 plans of the same batch size at once: plan i is the pn2_scatter_plan_bytes(b, nent[i],
// nsrc[i]) bytes at `buffer + offset[i]` (offsets ascending, 16-byte aligned, the whole range inside buffer_bytes) -- each of
// them a plan pn2_scatter_plan_apply takes.  One memset of the range + three launches.
extern "C" int pn2_scatter_plan_build_multi(int nplans, int b, const int* nent, const int* div, const int* nsrc,
                                            const int* const* idx, const float* const* weight, const int* weight_kind,
                                            void* buffer, const size_t* offset, size_t buffer_bytes, void* stream) {
    if (nplans <= 0 || nplans > kMaxPlans || b <= 0) return PN2_EINVAL;
    if (!nent || !div || !nsrc || !idx || !weight || !weight_kind || !buffer || !offset) return PN2_ENULL;
    if (b > 65535) return PN2_ERANGE;
    PlanBatch pb{};
    int max_ent = 0;
    size_t end = 0;
    for (int i = 0; i < nplans; ++i) {
        if (nent[i] <= 0 || nsrc[i] <= 0 || div[i] <= 0 || nent[i] % div[i] != 0) return PN2_EINVAL;
        if (weight_kind[i] < 0 || weight_kind[i] > 2 || (weight_kind[i] == 2 && div[i] != 3)) return PN2_EINVAL;
        if (!idx[i] || (weight_kind[i] != 0 && !weight[i])) return PN2_ENULL;
        const size_t bytes = pn2_scatter_plan_bytes(b, nent[i], nsrc[i]);
        if (offset[i] % 16 != 0 || offset[i] < end || offset[i] + bytes > buffer_bytes) return PN2_EINVAL;
        end = offset[i] + bytes;
        int* cnt = reinterpret_cast<int*>(static_cast<char*>(buffer) + offset[i]);
        PlanDesc& d = pb.d[i];
        d.nent = nent[i]; d.nsrc = nsrc[i]; d.div = div[i]; d.kind = weight_kind[i];
        d.idx = idx[i]; d.w = weight_kind[i] ? weight[i] : nullptr;
        d.cnt = cnt; d.off = cnt + (size_t)b * nsrc[i]; d.eq = d.off + (size_t)b * nsrc[i];
        d.ew = reinterpret_cast<float*>(d.eq + (size_t)b * nent[i]);
        max_ent = nent[i] > max_ent ? nent[i] : max_ent;
    }
    if (((uintptr_t)buffer % 16) != 0) return PN2_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int max_src = 0;
    for (int i = 0; i < nplans; ++i) max_src = nsrc[i] > max_src ? nsrc[i] : max_src;
    if (max_src <= kPlanLdsMaxSrc) {  // one launch, LDS atomics, no memset (cnt / off / eq / ew are all written)
        const size_t lds = sizeof(int) * 2 * (size_t)max_src;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(ti_csr_build_lds_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(int) * 2 * kPlanLdsMaxSrc);
            if (ea != hipSuccess) return (int)ea;
            attr_set = true;
        }
        ti_csr_build_lds_kernel<<<dim3(b, nplans), kPlanLdsThreads, lds, st>>>(pb);
        PN2_RETURN_IF_LAUNCH_FAILED();
        return PN2_OK;
    }
    hipError_t e = hipMemsetAsync(static_cast<char*>(buffer) + offset[0], 0, end - offset[0], st);
    if (e != hipSuccess) return (int)e;
    dim3 ge(grid_x_for((unsigned long long)max_ent, 256, b * nplans), b, nplans);
    ti_csr_count_multi_kernel<<<ge, 256, 0, st>>>(pb);
    ti_csr_scan_multi_kernel<<<dim3(b, nplans), 256, 0, st>>>(pb);
    ti_csr_fill_multi_kernel<<<ge, 256, 0, st>>>(pb);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_scatter_plan_apply(int b, int nent, int div, int c, int nsrc, const float* rows_in, int in_stride,
                                      const void* plan, size_t plan_bytes, float* out, void* stream) {
    if (b <= 0 || nent <= 0 || nsrc <= 0 || div <= 0 || nent % div != 0 || c <= 0 || in_stride < c) return PN2_EINVAL;
    if (!rows_in || !plan || !out) return PN2_ENULL;
    if (c % 4 != 0 || c > 1024 || ((uintptr_t)out % 16) != 0 || ((uintptr_t)rows_in % 4) != 0) return PN2_EUNSUP;
    if (b > 65535) return PN2_ERANGE;
    if (plan_bytes < pn2_scatter_plan_bytes(b, nent, nsrc) || ((uintptr_t)plan % 4) != 0) return PN2_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int* off = static_cast<const int*>(plan) + (size_t)b * nsrc;
    const int* ent_q = off + (size_t)b * nsrc;
    const float* ent_w = reinterpret_cast<const float*>(ent_q + (size_t)b * nent);
    const int spb = 256 / (c / 4);
    dim3 gg((nsrc + spb - 1) / spb, b);
    const bool aligned = ((uintptr_t)rows_in % 16) == 0 && in_stride % 4 == 0;
    if (aligned) ti_csr_gather_kernel<true><<<gg, 256, 0, st>>>(nent / div, nent, c, in_stride, nsrc, rows_in, off, ent_q, ent_w, out);
    else ti_csr_gather_kernel<false><<<gg, 256, 0, st>>>(nent / div, nent, c, in_stride, nsrc, rows_in, off, ent_q, ent_w, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// Internal: list build + gather in one call, shared by the two *_grad_ws entry points.
extern "C" size_t pn2_scatter_rows_workspace_bytes(int b, int nent, int nsrc) { return pn2_scatter_plan_bytes(b, nent, nsrc); }

extern "C" int pn2_scatter_rows_gather(int b, int nent, int div, int c, int nsrc, const float* rows_in, const int* idx,
                                       const float* weight, float* out, void* workspace, void* stream) {
    const size_t bytes = pn2_scatter_plan_bytes(b, nent, nsrc);
    const int rc = pn2_scatter_plan_build(b, nent, div, nsrc, idx, weight, weight ? 1 : 0, workspace, bytes, stream);
    if (rc != PN2_OK) return rc;
    return pn2_scatter_plan_apply(b, nent, div, c, nsrc, rows_in, c, workspace, bytes, out, stream);
}

extern "C" size_t pn2_three_interpolate_grad_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return pn2_scatter_rows_workspace_bytes(b, 3 * n, m);
}

extern "C" int pn2_three_interpolate_grad_ws(int b, int n, int c, int m, const float* grad_out, const int* idx,
                                             const float* weight, float* grad_points, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    if (b <= 0 || m <= 0 || c <= 0 || n <= 0) return PN2_EINVAL;
    if (!grad_out || !idx || !weight || !grad_points) return PN2_ENULL;
    const bool gather_ok = workspace && c % 4 == 0 && c <= 1024 && (((uintptr_t)grad_out | (uintptr_t)grad_points) % 16) == 0 &&
                           (unsigned long long)n * 3 <= 0x7fffffffull && b <= 65535;
    // the list build costs ~4 small launches: worth it from ~1 M float atomics
    if (!gather_ok || (unsigned long long)b * n * 3 * c < (1ull << 20))
        return pn2_three_interpolate_grad(b, n, c, m, grad_out, idx, weight, grad_points, stream);
    if (workspace_bytes < pn2_three_interpolate_grad_workspace_bytes(b, n, m) || ((uintptr_t)workspace % 4) != 0) return PN2_EINVAL;
    return pn2_scatter_rows_gather(b, 3 * n, 3, c, m, grad_out, idx, weight, grad_points, workspace, stream);
}

extern "C" int pn2_fp_interp_concat(int b, int n, int m, int c1, int c2, const float* dist,
                                    const int* idx, const float* points1, const float* points2,
                                    float* out, int out_stride, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c2 <= 0 || c1 < 0) return PN2_EINVAL;
    if (out_stride == 0) out_stride = c1 + c2;
    if (out_stride < c1 + c2 || out_stride > c1 + c2 + 64) return PN2_EINVAL;
    if (!dist || !idx || !points2 || !out || (c1 > 0 && !points1)) return PN2_ENULL;
    if ((unsigned long long)n * out_stride > 0x7fffffffull || b > 65535) return PN2_ERANGE;
    // rows per wave: as many as possible (amortises the per-row weight maths) while keeping
    // >= ~4096 waves in flight; 4 waves per block, grid-stride beyond ~8 blocks per CU
    int rpw = (int)(((long long)b * n) / 4096);
    if (rpw < 1) rpw = 1;
    if (rpw > 64) rpw = 64;
    int gx = (n + 4 * rpw - 1) / (4 * rpw);
    const int cap = (256 * 8 + b - 1) / b;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    dim3 grid(gx, b);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* q1 = c1 > 0 ? points1 : nullptr;
    {   // 16-byte-lane kernel when the layout allows it
        const int tailv = (out_stride - c2) / 4, c2v = c2 / 4;
        const bool pow2 = c2 % 4 == 0 && out_stride % 4 == 0 && (c2v & (c2v - 1)) == 0 && tailv > 0 &&
                          (tailv & (tailv - 1)) == 0 && c2v <= 1024;
        const bool al = (((uintptr_t)points2 | (uintptr_t)out) % 16) == 0;
        if (pow2 && al) {
            const int sh2 = 31 - __builtin_clz((unsigned)c2v), sht = 31 - __builtin_clz((unsigned)tailv);
            const bool p1v = c1 > 0 && c1 % 4 == 0 && ((uintptr_t)points1 % 16) == 0;
            if (p1v || c1 == 0)
                fp_interp_concat_v4_kernel<true><<<grid, 256, 0, st>>>(n, m, c1, c2, out_stride, rpw, sh2, sht, dist, idx,
                                                                     q1, points2, out);
            else
                fp_interp_concat_v4_kernel<false><<<grid, 256, 0, st>>>(n, m, c1, c2, out_stride, rpw, sh2, sht, dist,
                                                                      idx, q1, points2, out);
            PN2_RETURN_IF_LAUNCH_FAILED();
            return PN2_OK;
        }
    }
    const int chw = (out_stride + 63) / 64;
#define PN2_FIC(CHW_, R_) fp_interp_concat_rows_kernel<CHW_, R_><<<grid, 256, 0, st>>>( \
        n, m, c1, c2, out_stride, rpw, dist, idx, q1, points2, out)
    if (chw <= 1) PN2_FIC(1, 8);
    else if (chw == 2) PN2_FIC(2, 4);
    else if (chw == 3) PN2_FIC(3, 4);
    else if (chw == 4) PN2_FIC(4, 2);
    else if (chw <= 6) PN2_FIC(6, 2);
    else if (chw <= 12) PN2_FIC(12, 1);
    else fp_interp_concat_kernel<<<grid, 256, 0, st>>>(n, m, c1, c2, out_stride, rpw, dist, idx, q1, points2, out);
#undef PN2_FIC
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
