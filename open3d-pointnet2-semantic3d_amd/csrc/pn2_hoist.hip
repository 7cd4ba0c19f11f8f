// pn2_hoist.hip -- first layer of an SA / FP module of the TRAINING path with its feature half applied to the source rows.
//
// The first 1x1 conv of pointnet_sa_module acts on concat[grouped_xyz - new_xyz | group_point(points)]
// (util/pointnet_util.py:39-54,150-156), that of pointnet_fp_module on concat[three_interpolate(points2) | points1]
// (:300-312).  Gathering and interpolating are linear and act row-wise, so they commute with the conv:
//     SA:  y[b,j,k,:] = (xyz[b,idx] - new_xyz[b,j]) @ W[:3]  +  (points @ W[3:])[b, idx[b,j,k], :]
//     FP:  y[b,i,:]   = sum_k w_k (points2 @ W[:c2])[b, idx[b,i,k], :]  +  points1[b,i,:] @ W[c2:]
// The product z = points @ W[3:] resp. points2 @ W[:c2] is one GEMM over the SOURCE rows (8x fewer than the grouped /
// interpolated rows for K = 32 neighbours over 4x fewer centres, resp. n / m = 4..8), and so are its data and weight
// gradients; the kernels here do what is left per output row: gather 1 (SA) or 3 weighted (FP) rows of z and add the
// 3-channel (xyz) resp. c1 <= 8 channel (points1) product.  HBM-bound: one write of y, the gathers are served by L2
// (workgroup L runs on XCD L % 8: whole batch elements per XCD).  Inference uses the fused *_pre kernels instead
// (pn2_sa_fused.hip, pn2_mlp_wide.hip); batch norm needs y in memory, so training keeps it.
#include "pn2_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHoistMaxCa = 8;
#ifndef PN2_HOIST_STAT_WGS
#define PN2_HOIST_STAT_WGS 512
#endif

struct HoistParams {
    int rows;      // output rows per batch element (SA: m * nsample, FP: n)
    int nsrc;      // rows of z per batch element
    int cout;      // width of z and y, % 4 == 0, <= 1024
    int ca;        // channels of the direct operand (SA: 3, FP: c1 <= 8)
    int nsample;   // SA: neighbours per centre
    int m;         // SA: centres per batch element
    const int* idx;        // SA (b, m, nsample), FP (b, n, 3)
    const float* dist;     // FP (b, n, 3) squared distances of three_nn
    const float* z;        // (b, nsrc, cout)
    const float* xyz;      // SA (b, nsrc, 3)
    const float* new_xyz;  // SA (b, m, 3)
    const float* points1;  // FP (b, n, ca)
    const float* wa;       // (ca, cout)
    float* y;              // (b, rows, cout)
    float* gxyz;           // SA (b, rows, 3) centred coordinates (operand of the weight gradient), may be null
    int accumulate;        // FP with ca == 0: y already holds points1 @ W[c2:] (a GEMM of the caller), the blended rows are added
    // STATS (pn2_*_hoist_rows_bn): the column sums of y and y^2 the batch norm behind this layer needs, taken on the way out
    // (fp64 from the first term on, as bn_stats_kernel forms them) into slot copies of the ZEROED workspace; the workgroup that
    // draws the last ticket folds them and derives the deferred batch norm's constants (pn2_common.h pn2_bn_finish)
    double* stats_ws;
    int nslots;
    Pn2BnFinish fin;
};

template <bool SA, int CA, bool STATS = false>
__global__ void __launch_bounds__(256)
hoist_rows_kernel(HoistParams p) {
    int bx = blockIdx.x, bi = blockIdx.y;
    {   // XCD-aware block remap (speed only), as in fp_interp_concat_rows_kernel
        const unsigned nwg = gridDim.x * gridDim.y;
        if ((nwg & 7u) == 0u) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned swz = (lin & 7u) * (nwg >> 3) + (lin >> 3);
            bx = (int)(swz % gridDim.x);
            bi = (int)(swz / gridDim.x);
        }
    }
    const int cv = p.cout >> 2;                 // float4 columns
    const int rp = 256 / cv;                    // rows per pass of the block (cv <= 256)
    const int rr = (int)threadIdx.x / cv, cc = (int)threadIdx.x - rr * cv;
    const bool active = rr < rp;
    if (!STATS && !active) return;
    double part[2][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) part[0][v] = part[1][v] = 0.0;
    const f32x4* __restrict__ z = reinterpret_cast<const f32x4*>(p.z + (size_t)bi * p.nsrc * p.cout);
    f32x4* __restrict__ y = reinterpret_cast<f32x4*>(p.y + (size_t)bi * p.rows * p.cout);
    f32x4 wa[CA > 0 ? CA : 1];
#pragma unroll
    for (int a = 0; a < CA; ++a) wa[a] = *reinterpret_cast<const f32x4*>(p.wa + (size_t)a * p.cout + cc * 4);
    constexpr int U = 4;  // rows in flight per thread
    for (int r0 = bx * rp * U + rr; active && r0 < p.rows; r0 += gridDim.x * rp * U) {
        f32x4 acc[U];
        float av[U][CA > 0 ? CA : 1];
        int rows_[U];
        if constexpr (SA) {
            const int* __restrict__ idx = p.idx + (size_t)bi * p.rows;
            const float* __restrict__ xyz = p.xyz + (size_t)bi * p.nsrc * 3;
            const float* __restrict__ nx = p.new_xyz + (size_t)bi * p.m * 3;
            int src[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * rp;
                rows_[u] = r;
                const int rc = r < p.rows ? r : p.rows - 1;
                src[u] = idx[rc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rc = rows_[u] < p.rows ? rows_[u] : p.rows - 1;
                const int j = rc / p.nsample;
                acc[u] = z[(size_t)src[u] * cv + cc];
#pragma unroll
                for (int a = 0; a < 3; ++a) av[u][a] = xyz[(size_t)src[u] * 3 + a] - nx[(size_t)j * 3 + a];
            }
            if (p.gxyz && cc == 0) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (rows_[u] < p.rows) {
                        float* __restrict__ g = p.gxyz + ((size_t)bi * p.rows + rows_[u]) * 3;
                        g[0] = av[u][0]; g[1] = av[u][1]; g[2] = av[u][2];
                    }
            }
        } else {
            const int* __restrict__ idx = p.idx + (size_t)bi * p.rows * 3;
            const float* __restrict__ dist = p.dist + (size_t)bi * p.rows * 3;
            const float* __restrict__ p1 = CA > 0 ? p.points1 + (size_t)bi * p.rows * CA : nullptr;
            int i1[U], i2[U], i3[U];
            float w1[U], w2[U], w3[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * rp;
                rows_[u] = r;
                const int rc = r < p.rows ? r : p.rows - 1;
                i1[u] = idx[rc * 3 + 0]; i2[u] = idx[rc * 3 + 1]; i3[u] = idx[rc * 3 + 2];
                // the weights of pointnet_util.py:300-303 with fp_interp_concat's float expressions
                const float d1 = fmaxf(dist[rc * 3 + 0], 1e-10f), d2 = fmaxf(dist[rc * 3 + 1], 1e-10f);
                const float d3 = fmaxf(dist[rc * 3 + 2], 1e-10f);
                const float q1 = 1.0f / d1, q2 = 1.0f / d2, q3 = 1.0f / d3;
                const float norm = (q1 + q2) + q3;
                w1[u] = q1 / norm; w2[u] = q2 / norm; w3[u] = q3 / norm;
#pragma unroll
                for (int a = 0; a < CA; ++a) av[u][a] = p1[(size_t)rc * CA + a];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4 a = z[(size_t)i1[u] * cv + cc], b = z[(size_t)i2[u] * cv + cc], c = z[(size_t)i3[u] * cv + cc];
                acc[u] = (a * w1[u] + b * w2[u]) + c * w3[u];
                if (CA == 0 && p.accumulate) acc[u] += y[(size_t)(rows_[u] < p.rows ? rows_[u] : p.rows - 1) * cv + cc];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 v = acc[u];
#pragma unroll
            for (int a = 0; a < CA; ++a) {
                v[0] = __builtin_fmaf(av[u][a], wa[a][0], v[0]);
                v[1] = __builtin_fmaf(av[u][a], wa[a][1], v[1]);
                v[2] = __builtin_fmaf(av[u][a], wa[a][2], v[2]);
                v[3] = __builtin_fmaf(av[u][a], wa[a][3], v[3]);
            }
            if (rows_[u] < p.rows) {
                y[(size_t)rows_[u] * cv + cc] = v;
                if constexpr (STATS) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double d = (double)v[q];
                        part[0][q] += d;
                        part[1][q] = __builtin_fma(d, d, part[1][q]);
                    }
                }
            }
        }
    }
    if constexpr (STATS) {
        // the rp row slots of the workgroup meet in LDS (tree; rp need not be a power of two), one atomic pair per column
        __shared__ double red[256 * 2 * 4];
        if (active) {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[((s_ * 4 + v) * rp + rr) * cv + cc] = part[s_][v];
        }
        __syncthreads();
        for (int sz = rp; sz > 1;) {
            const int h = (sz + 1) >> 1;
            if (active && rr + h < sz) {
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                    for (int v = 0; v < 4; ++v) red[((s_ * 4 + v) * rp + rr) * cv + cc] += red[((s_ * 4 + v) * rp + rr + h) * cv + cc];
            }
            __syncthreads();
            sz = h;
        }
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, nwg = gridDim.x * gridDim.y;
        double* __restrict__ slot = p.stats_ws + kPn2BnHead + (size_t)2 * p.cout * (1 + lin % (unsigned)p.nslots);
        if (active && rr == 0) {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int v = 0; v < 4; ++v) atomicAdd(&slot[(size_t)s_ * p.cout + cc * 4 + v], red[((s_ * 4 + v) * rp) * cv + cc]);
        }
        pn2_bn_finish(p.fin, nwg, lin);
    }
}

template <bool SA, int CA>
int launch_hoist(int b, const HoistParams& p, hipStream_t st) {
    const int cv = p.cout / 4, rp = 256 / cv;
    long long blocks = ((long long)p.rows + rp * 4 - 1) / (rp * 4);
    // ~8 workgroups per CU over the batch; with the statistics epilogue every workgroup ends in 2 * cout fp64 atomics and a ticket:
    // PN2_HOIST_STAT_WGS of them (tools/hoist_stats_ab.py)
    const int total = p.stats_ws ? PN2_HOIST_STAT_WGS : 2048;
    const long long cap = total / (b < 1 ? 1 : b) < 8 ? 8 : total / b;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if ((blocks * b) % 8 != 0 && blocks > 8) blocks -= blocks % 8;     // keeps the XCD remap active
    if (p.stats_ws) hoist_rows_kernel<SA, CA, true><<<dim3((unsigned)blocks, b), 256, 0, st>>>(p);
    else hoist_rows_kernel<SA, CA><<<dim3((unsigned)blocks, b), 256, 0, st>>>(p);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

}  // namespace

// SA: y (b, m, nsample, cout) = (group_point(xyz, idx) - new_xyz) @ w_xyz (3, cout) + z[b, idx] with z (b, n, cout) =
// points @ W[3:] computed by the caller (pn2_linear); gxyz (b, m, nsample, 3) (optional) receives the centred coordinates
// (x operand of w_xyz's gradient).  cout % 4 == 0, cout <= 1024.
// the statistics arguments of the *_bn entry points -> (stats_ws, nslots, fin) of HoistParams; finish: 0 sums only (slot copies
// left in the workspace), 1 folded, 2 folded + the deferred batch norm's constants (as pn2_linear_bn_stats_fin)
static int hoist_stats_args(HoistParams& p, long long rows_total, int cout, void* bn_workspace, size_t workspace_bytes, int finish,
                            const float* gamma, const float* beta, const float* bias, float eps, float decay, float* running_mean,
                            float* running_var, float* save_mean, float* save_invstd, float* scale, float* shift) {
    if (!bn_workspace) return PN2_ENULL;
    if (finish < 0 || finish > 2) return PN2_EINVAL;
    if (workspace_bytes < sizeof(double) * pn2_bn_ws_doubles(cout, kPn2BnSlots) || ((uintptr_t)bn_workspace % 8) != 0) return PN2_EINVAL;
    p.stats_ws = static_cast<double*>(bn_workspace);
    p.nslots = kPn2BnSlots;
    Pn2BnFinish f{};
    f.kind = finish; f.c = cout; f.nslots = kPn2BnSlots; f.rows = rows_total; f.ws = p.stats_ws;
    if (finish == 2) {
        if (!gamma || !beta || !save_mean || !save_invstd || (scale == nullptr) != (shift == nullptr)) return PN2_ENULL;
        if ((running_mean == nullptr) != (running_var == nullptr)) return PN2_ENULL;
        f.gamma = gamma; f.beta = beta; f.bias = bias; f.eps = eps; f.decay = decay; f.running_mean = running_mean;
        f.running_var = running_var; f.save_mean = save_mean; f.save_invstd = save_invstd; f.scale = scale; f.shift = shift;
    }
    p.fin = f;
    return PN2_OK;
}

static int sa_hoist_impl(int b, int n, int m, int nsample, int cout, const float* xyz, const float* new_xyz, const int* idx,
                         const float* z, const float* w_xyz, float* y, float* gxyz, void* stream, const HoistParams* st_args) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || cout <= 0) return PN2_EINVAL;
    if (!xyz || !new_xyz || !idx || !z || !w_xyz || !y) return PN2_ENULL;
    if (cout % 4 != 0 || cout > 1024) return PN2_EUNSUP;
    if ((((uintptr_t)z | (uintptr_t)y | (uintptr_t)w_xyz) % 16) != 0) return PN2_EINVAL;
    if ((long long)m * nsample > 0x7fffffffLL || b > 65535) return PN2_ERANGE;
    HoistParams p = {};
    p.rows = m * nsample; p.nsrc = n; p.cout = cout; p.ca = 3; p.nsample = nsample; p.m = m;
    p.idx = idx; p.z = z; p.xyz = xyz; p.new_xyz = new_xyz; p.wa = w_xyz; p.y = y; p.gxyz = gxyz;
    if (st_args) { p.stats_ws = st_args->stats_ws; p.nslots = st_args->nslots; p.fin = st_args->fin; }
    return launch_hoist<true, 3>(b, p, static_cast<hipStream_t>(stream));
}

extern "C" int pn2_sa_hoist_rows(int b, int n, int m, int nsample, int cout, const float* xyz, const float* new_xyz,
                                 const int* idx, const float* z, const float* w_xyz, float* y, float* gxyz, void* stream) {
    return sa_hoist_impl(b, n, m, nsample, cout, xyz, new_xyz, idx, z, w_xyz, y, gxyz, stream, nullptr);
}

// pn2_sa_hoist_rows that also leaves the batch statistics of y (tf_util.py:186-204: conv2d -> batch_norm_template) in the ZEROED
// batch-norm workspace (pn2_bn_workspace_bytes(cout)) -- no statistics pass over y -- and, finish = 1 / 2, folds them / publishes
// what pn2_bn_relu_forward_deferred publishes, as pn2_linear_bn_stats_fin does for a GEMM.
extern "C" int pn2_sa_hoist_rows_bn(int b, int n, int m, int nsample, int cout, const float* xyz, const float* new_xyz,
                                    const int* idx, const float* z, const float* w_xyz, float* y, float* gxyz, void* bn_workspace,
                                    size_t workspace_bytes, int finish, const float* gamma, const float* beta, const float* bias,
                                    float eps, float decay, float* running_mean, float* running_var, float* save_mean,
                                    float* save_invstd, float* scale, float* shift, void* stream) {
    if (b <= 0 || m <= 0 || nsample <= 0 || cout <= 0) return PN2_EINVAL;
    HoistParams sp = {};
    const int rc = hoist_stats_args(sp, (long long)b * m * nsample, cout, bn_workspace, workspace_bytes, finish, gamma, beta, bias, eps,
                                    decay, running_mean, running_var, save_mean, save_invstd, scale, shift);
    if (rc != PN2_OK) return rc;
    return sa_hoist_impl(b, n, m, nsample, cout, xyz, new_xyz, idx, z, w_xyz, y, gxyz, stream, &sp);
}

// FP: y (b, n, cout) = three_interpolate(z, idx, w(dist)) + points1 (b, n, c1) @ w1 (c1, cout) with z (b, m, cout) =
// points2 @ W[:c2] computed by the caller; the weights are formed from three_nn's squared distances as in
// pn2_fp_interp_concat.  1 <= c1 <= 8: the product with points1 is formed here (the level-0 module: colours).
// c1 == 0 with points1 == w1 == NULL: y += three_interpolate(...), y holding the caller's GEMM points1 @ W[c2:] (wider skip
// links: that product is a real GEMM).
static int fp_hoist_impl(int b, int n, int m, int c1, int cout, const float* dist, const int* idx, const float* points1,
                         const float* z, const float* w1, float* y, void* stream, const HoistParams* st_args) {
    if (b <= 0 || n <= 0 || m <= 0 || cout <= 0) return PN2_EINVAL;
    if (!dist || !idx || !z || !y) return PN2_ENULL;
    if (c1 > 0 && (!points1 || !w1)) return PN2_ENULL;
    if (c1 < 0 || c1 > kHoistMaxCa || cout % 4 != 0 || cout > 1024) return PN2_EUNSUP;
    if ((((uintptr_t)z | (uintptr_t)y | (uintptr_t)w1) % 16) != 0) return PN2_EINVAL;
    if ((long long)n * 3 > 0x7fffffffLL || b > 65535) return PN2_ERANGE;
    HoistParams p = {};
    p.rows = n; p.nsrc = m; p.cout = cout; p.ca = c1;
    p.idx = idx; p.dist = dist; p.z = z; p.points1 = points1; p.wa = w1; p.y = y;
    p.accumulate = c1 == 0;
    if (st_args) { p.stats_ws = st_args->stats_ws; p.nslots = st_args->nslots; p.fin = st_args->fin; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (c1) {
        case 0: return launch_hoist<false, 0>(b, p, st);
        case 1: return launch_hoist<false, 1>(b, p, st);
        case 2: return launch_hoist<false, 2>(b, p, st);
        case 3: return launch_hoist<false, 3>(b, p, st);
        case 4: return launch_hoist<false, 4>(b, p, st);
        case 5: return launch_hoist<false, 5>(b, p, st);
        case 6: return launch_hoist<false, 6>(b, p, st);
        case 7: return launch_hoist<false, 7>(b, p, st);
        default: return launch_hoist<false, 8>(b, p, st);
    }
}

extern "C" int pn2_fp_hoist_rows(int b, int n, int m, int c1, int cout, const float* dist, const int* idx,
                                 const float* points1, const float* z, const float* w1, float* y, void* stream) {
    return fp_hoist_impl(b, n, m, c1, cout, dist, idx, points1, z, w1, y, stream, nullptr);
}

// pn2_fp_hoist_rows with the statistics epilogue of pn2_sa_hoist_rows_bn.
extern "C" int pn2_fp_hoist_rows_bn(int b, int n, int m, int c1, int cout, const float* dist, const int* idx,
                                    const float* points1, const float* z, const float* w1, float* y, void* bn_workspace,
                                    size_t workspace_bytes, int finish, const float* gamma, const float* beta, const float* bias,
                                    float eps, float decay, float* running_mean, float* running_var, float* save_mean,
                                    float* save_invstd, float* scale, float* shift, void* stream) {
    if (b <= 0 || n <= 0 || cout <= 0) return PN2_EINVAL;
    HoistParams sp = {};
    const int rc = hoist_stats_args(sp, (long long)b * n, cout, bn_workspace, workspace_bytes, finish, gamma, beta, bias, eps, decay,
                                    running_mean, running_var, save_mean, save_invstd, scale, shift);
    if (rc != PN2_OK) return rc;
    return fp_hoist_impl(b, n, m, c1, cout, dist, idx, points1, z, w1, y, stream, &sp);
}
