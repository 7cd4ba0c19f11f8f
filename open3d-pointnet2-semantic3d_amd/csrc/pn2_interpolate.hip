// pn2_interpolate.hip -- three_nn, three_interpolate (+grad) and the fused FP
// front end for gfx950.  MI355X-native replacements for the CPU-only ops of
// tf_ops/tf_interpolate.cpp:213-243,307-330,397-421 (which bounce every FP layer
// GPU->host->GPU in the reference) and of util/pointnet_util.py:300-311.
//
// three_nn is exact float64 like the reference's KD-tree (tf_interpolate.cpp:20-28:
// points widened to Eigen::Vector3d; FLANN L2<double> accumulates
// ((0+dx*dx)+dy*dy)+dz*dz), brute force with an fp32 prefilter: the known points
// of a batch element are staged in LDS and broadcast-read by a thread-per-query
// scan.  Bound: VALU issue, not HBM.
#include <math.h>

#include "pn2_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNnThreads = 256;
constexpr int kNnWaves = kNnThreads / 64;
constexpr int kNnQ = 8;        // queries per wave
constexpr int kNnList = 32;    // per-query candidate list (LDS)

struct NnPoint { float x, y, z; };

__device__ __forceinline__ float nn_d32(float qx, float qy, float qz, const NnPoint& c) {
    const float fx = qx - c.x, fy = qy - c.y, fz = qz - c.z;
    return __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));  // filter only: any rounding within the bound
}

// wave64 minimum of non-negative floats / +inf (as int bits), uniform result
__device__ __forceinline__ int nn_wave_imin(int v) {
    asm volatile(
        "s_nop 1\n"
        "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
        "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// Exact float64 3-NN (reference semantics: tf_interpolate.cpp:20-28 -> FLANN L2<double>,
// ((0+dx*dx)+dy*dy)+dz*dz), organised like the ball query: one wave owns kNnQ queries, the known
// points sit one per lane in registers (16 chunks = 1024 points at a time) and are shared by the
// wave's queries.
//   pass 1  fp32, branch-free: per query, every lane keeps the minimum d32 over its own candidates;
//           the 3rd smallest of the 64 lane minima (three distinct candidates) bounds the 3rd-NN
//           distance from above:  thr = t3 * (1 + 2e-6);
//   pass 2  fp32: candidates with d32 <= thr (a handful) are appended in index order (ballot +
//           mbcnt) to the query's list in LDS;
//   refine  lane q walks the list of query q in float64 with a strict '<' insertion, so ties keep
//           the lowest index exactly like a full ascending scan.
// Filter safety: inputs are exact and all terms non-negative, so |d32 - d| <= 5*2^-24 * d; the three
// bounding candidates have exact distances <= t3*(1+4e-7), hence every true top-3 candidate has
// d32 <= t3*(1+8e-7) < thr.  A list overflow (> 32 candidates inside thr: heavy duplication) falls
// back to a full float64 scan of that query by one lane.
// kNnChunks = candidate chunks (of 64) held in registers at a time (16 -> 1024 points; small known
// sets instantiate 4 or 1 so that the unrolled chunk loops do no dead work).
template <int kNnChunks>
__global__ void __launch_bounds__(kNnThreads)
three_nn_kernel(int n, int m, const float* __restrict__ xyz1_all,
                const float* __restrict__ xyz2_all, float* __restrict__ dist_all,
                int* __restrict__ idx_all) {
    __shared__ int slist[kNnWaves * kNnQ * kNnList];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    const int q0 = (blockIdx.x * kNnWaves + wave) * kNnQ;
    if (q0 >= n) return;  // wave-uniform, no barriers in this kernel
    const float* __restrict__ xyz1 = xyz1_all + (size_t)bi * n * 3;
    const float* __restrict__ xyz2 = xyz2_all + (size_t)bi * m * 3;
    const NnPoint* __restrict__ cand = reinterpret_cast<const NnPoint*>(xyz2);
    int* wl = slist + wave * kNnQ * kNnList;

    float qx[kNnQ], qy[kNnQ], qz[kNnQ];
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) {
        const int jq = q0 + q < n ? q0 + q : n - 1;
        qx[q] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(xyz1[jq * 3 + 0])));
        qy[q] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(xyz1[jq * 3 + 1])));
        qz[q] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(xyz1[jq * 3 + 2])));
    }
    const int last = m - 1;
    const int nblk = (m + 64 * kNnChunks - 1) / (64 * kNnChunks);
    NnPoint c[kNnChunks];
    auto load_block = [&](int blk) {
#pragma unroll
        for (int t = 0; t < kNnChunks; ++t) {
            const int k = (blk * kNnChunks + t) * 64 + lane;
            c[t] = cand[k < last ? k : last];  // clamped; out-of-range lanes are masked by index below
        }
    };

    // ---- pass 1 -------------------------------------------------------------------------------
    int mn[kNnQ];  // per-lane minimum d32 as int bits (distances are >= +0: int order == float order)
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) mn[q] = 0x7F800000;
    for (int blk = 0; blk < nblk; ++blk) {
        load_block(blk);
#pragma unroll
        for (int t = 0; t < kNnChunks; ++t) {
            const int k = (blk * kNnChunks + t) * 64 + lane;
            const int cbase = (blk * kNnChunks + t) * 64;
            if (cbase + 64 <= m) {  // full chunk (wave-uniform): no masking, integer min on the bits
#pragma unroll
                for (int q = 0; q < kNnQ; ++q) {
                    const int di = __float_as_int(nn_d32(qx[q], qy[q], qz[q], c[t]));
                    mn[q] = di < mn[q] ? di : mn[q];
                }
            } else if (cbase < m) {  // partial last chunk: lanes past m re-read point m-1 and must not count
                const bool valid = k < m;
#pragma unroll
                for (int q = 0; q < kNnQ; ++q) {
                    const int di = valid ? __float_as_int(nn_d32(qx[q], qy[q], qz[q], c[t])) : 0x7F800000;
                    mn[q] = di < mn[q] ? di : mn[q];
                }
            }
        }
    }
    float thr[kNnQ];
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) {
        int v = mn[q];
        int t3 = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            t3 = nn_wave_imin(v);
            const unsigned long long eq = __ballot(v == t3);
            const int first = __ffsll((long long)eq) - 1;
            if (lane == first) v = 0x7F800000;  // drop exactly one holder of the current minimum
        }
        thr[q] = __int_as_float(t3) * (1.0f + 2e-6f);  // +inf stays +inf
    }

    // ---- pass 2: collect ----------------------------------------------------------------------
    int cnt[kNnQ];
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) cnt[q] = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        if (nblk > 1) load_block(blk);  // single block: still resident from pass 1
#pragma unroll
        for (int t = 0; t < kNnChunks; ++t) {
            const int k = (blk * kNnChunks + t) * 64 + lane;
            const int cbase = (blk * kNnChunks + t) * 64;
            if (cbase < m) {
                const bool tail = cbase + 64 > m;  // wave-uniform
                unsigned long long mk[kNnQ];
                unsigned long long any = 0ull;
#pragma unroll
                for (int q = 0; q < kNnQ; ++q) {
                    bool hit = nn_d32(qx[q], qy[q], qz[q], c[t]) <= thr[q];
                    if (tail) hit = hit && (k < m);
                    mk[q] = __ballot(hit);
                    any |= mk[q];
                }
                if (any != 0ull) {  // one branch per chunk (scalar ops are expensive, see pn2_grouping.hip)
#pragma unroll
                    for (int q = 0; q < kNnQ; ++q) {
                        const unsigned long long mask = mk[q];
                        if (mask != 0ull) {
                            const int pos = cnt[q] + (int)__builtin_amdgcn_mbcnt_hi(
                                                         (unsigned)(mask >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                            if (((mask >> lane) & 1ull) && pos < kNnList) wl[q * kNnList + pos] = k;
                            cnt[q] += __popcll(mask);  // may exceed kNnList: overflow marker
                        }
                    }
                }
            }
        }
    }

    // ---- refine: lane q owns query q ----------------------------------------------------------
    int myc = 0;
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) myc = lane == q ? cnt[q] : myc;
    if (lane < kNnQ && q0 + lane < n) {
        const int jq = q0 + lane;
        const double dqx = xyz1[jq * 3 + 0], dqy = xyz1[jq * 3 + 1], dqz = xyz1[jq * 3 + 2];
        double b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
        int i1 = 0, i2 = 0, i3 = 0;
        const bool overflow = myc > kNnList;
        const int ne = overflow ? m : myc;
        for (int e = 0; e < ne; ++e) {
            const int kk = overflow ? e : wl[lane * kNnList + e];
            const double dx = dqx - (double)xyz2[kk * 3 + 0];
            const double dy = dqy - (double)xyz2[kk * 3 + 1];
            const double dz = dqz - (double)xyz2[kk * 3 + 2];
            const double d = (dx * dx + dy * dy) + dz * dz;  // contraction is off
            if (d < b3) {  // strict: ascending index order keeps the lowest index on ties
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = kk; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = kk; }
                else { b3 = d; i3 = kk; }
            }
        }
        const size_t o = ((size_t)bi * n + jq) * 3;
        dist_all[o + 0] = (float)b1; dist_all[o + 1] = (float)b2; dist_all[o + 2] = (float)b3;
        idx_all[o + 0] = i1; idx_all[o + 1] = i2; idx_all[o + 2] = i3;
    }
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

inline int grid_x_for(unsigned long long total, int block, int batches) {
    unsigned long long g = (total + block - 1) / block;
    unsigned long long cap = (256ull * 8 + batches - 1) / batches;
    if (cap < 1) cap = 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int pn2_three_nn(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist,
                            int* idx, void* stream) {
    if (b <= 0 || n <= 0 || m < 3) return PN2_EINVAL;
    if (!xyz1 || !xyz2 || !dist || !idx) return PN2_ENULL;
    if ((long long)n * 3 > 0x7fffffffLL || (long long)m * 3 > 0x7fffffffLL || b > 65535) return PN2_ERANGE;
    dim3 grid((n + kNnWaves * kNnQ - 1) / (kNnWaves * kNnQ), b);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (m <= 64) three_nn_kernel<1><<<grid, kNnThreads, 0, st>>>(n, m, xyz1, xyz2, dist, idx);
    else if (m <= 256) three_nn_kernel<4><<<grid, kNnThreads, 0, st>>>(n, m, xyz1, xyz2, dist, idx);
    else three_nn_kernel<16><<<grid, kNnThreads, 0, st>>>(n, m, xyz1, xyz2, dist, idx);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
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
    fp_interp_concat_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
        n, m, c1, c2, out_stride, rpw, dist, idx, c1 > 0 ? points1 : nullptr, points2, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
