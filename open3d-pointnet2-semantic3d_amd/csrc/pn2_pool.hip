// pn2_pool.hip -- the pooling variants of pointnet_sa_module over the K neighbours of a group
// (util/pointnet_util.py:165-191 of the reference): max, avg, weighted_avg, max_and_avg, forward and gradient.
// The model only instantiates "max" (fused into the SA kernels / the last layer's batch-norm kernels); these cover the
// other values of the layer API's `pooling=` argument and `group_all` without falling back to library reductions.
// HBM-bound: one read of x (rows, k, c) -- thread = (row, 4 channels), k strided 16-byte loads, coalesced along c.
#include "pn2_common.h"

namespace {

enum { kPoolMax = 0, kPoolAvg = 1, kPoolWeightedAvg = 2, kPoolMaxAndAvg = 3 };

// weights of "weighted_avg": w_j = exp(-5 |g_j|) / sum_j exp(-5 |g_j|), g = grouped_xyz (rows, k, 3)   (:175-181)
__device__ __forceinline__ float pool_expdist(const float* __restrict__ g) {
    return expf(-(sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2])) * 5.0f);
}

template <int V>  // V = channels per thread (4: 16-byte loads when c % 4 == 0; 1 otherwise)
__global__ void __launch_bounds__(256)
group_pool_kernel(long long rows, int k, int c, int mode, const float* __restrict__ x, const float* __restrict__ gxyz,
                  float* __restrict__ out) {
    const int cv = c / V;
    const long long total = rows * cv;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / cv;
        const int c0 = (int)(e - r * cv) * V;
        const float* __restrict__ xr = x + (r * k) * c + c0;
        float mx[V], sm[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { mx[v] = -INFINITY; sm[v] = 0.f; }
        float wsum = 0.f;
        if (mode == kPoolWeightedAvg)
            for (int j = 0; j < k; ++j) wsum += pool_expdist(gxyz + (r * k + j) * 3);
        for (int j = 0; j < k; ++j) {
            float xv[V];
            if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(xr + (long long)j * c); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; }
            else xv[0] = xr[(long long)j * c];
            const float w = mode == kPoolWeightedAvg ? pool_expdist(gxyz + (r * k + j) * 3) / wsum : 1.0f;
#pragma unroll
            for (int v = 0; v < V; ++v) { mx[v] = fmaxf(mx[v], xv[v]); sm[v] += xv[v] * w; }
        }
        const int oc = mode == kPoolMaxAndAvg ? 2 * c : c;
        float* __restrict__ o = out + r * oc + c0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (mode == kPoolMax) o[v] = mx[v];
            else if (mode == kPoolAvg) o[v] = sm[v] / (float)k;
            else if (mode == kPoolWeightedAvg) o[v] = sm[v];
            else { o[v] = sm[v] / (float)k; o[c + v] = mx[v]; }  // concat [avg, max]  (:189)
        }
    }
}

// gradient w.r.t. x only (grouped_xyz comes from index ops: no gradient, like the reference's NoGradient ops).
// max: the gradient goes to the FIRST neighbour that attains the maximum (tf.reduce_max splits ties evenly in TF; the
// model's own max pooling lives in the batch-norm kernels with TF's even split -- this entry point serves the
// inference-style poolings of the API and documents the difference).
template <int V>
__global__ void __launch_bounds__(256)
group_pool_grad_kernel(long long rows, int k, int c, int mode, const float* __restrict__ x, const float* __restrict__ gxyz,
                       const float* __restrict__ dout, float* __restrict__ dx) {
    const int cv = c / V;
    const long long total = rows * cv;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / cv;
        const int c0 = (int)(e - r * cv) * V;
        const float* __restrict__ xr = x + (r * k) * c + c0;
        float* __restrict__ dr = dx + (r * k) * c + c0;
        const int oc = mode == kPoolMaxAndAvg ? 2 * c : c;
        const float* __restrict__ g = dout + r * oc + c0;
        float mx[V];
        int ties[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { mx[v] = -INFINITY; ties[v] = 0; }
        float wsum = 0.f;
        if (mode == kPoolWeightedAvg)
            for (int j = 0; j < k; ++j) wsum += pool_expdist(gxyz + (r * k + j) * 3);
        if (mode == kPoolMax || mode == kPoolMaxAndAvg) {
            for (int j = 0; j < k; ++j)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float t = xr[(long long)j * c + v];
                    if (t > mx[v]) { mx[v] = t; ties[v] = 1; } else if (t == mx[v]) ++ties[v];
                }
        }
        for (int j = 0; j < k; ++j) {
            const float w = mode == kPoolWeightedAvg ? pool_expdist(gxyz + (r * k + j) * 3) / wsum : 1.0f / (float)k;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float t = xr[(long long)j * c + v];
                float d = 0.f;
                if (mode == kPoolAvg || mode == kPoolWeightedAvg) d = g[v] * w;
                else if (mode == kPoolMax) d = t == mx[v] ? g[v] / (float)ties[v] : 0.f;  // even split among ties (tf.reduce_max)
                else d = g[v] * w + (t == mx[v] ? g[c + v] / (float)ties[v] : 0.f);
                dr[(long long)j * c + v] = d;
            }
        }
    }
}

inline int pool_grid(long long total) {
    long long g = (total + 255) / 256;
    const long long cap = 256LL * 16;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// x (rows, k, c), gxyz (rows, k, 3) (weighted_avg only, else may be NULL) -> out (rows, c) [max_and_avg: (rows, 2c) = [avg | max]]
extern "C" int pn2_group_pool(long long rows, int k, int c, int mode, const float* x, const float* gxyz, float* out, void* stream) {
    if (rows <= 0 || k <= 0 || c <= 0 || mode < 0 || mode > 3) return PN2_EINVAL;
    if (!x || !out || (mode == kPoolWeightedAvg && !gxyz)) return PN2_ENULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c % 4 == 0 && ((uintptr_t)x & 15) == 0) group_pool_kernel<4><<<pool_grid(rows * (c / 4)), 256, 0, st>>>(rows, k, c, mode, x, gxyz, out);
    else group_pool_kernel<1><<<pool_grid(rows * c), 256, 0, st>>>(rows, k, c, mode, x, gxyz, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_group_pool_grad(long long rows, int k, int c, int mode, const float* x, const float* gxyz, const float* dout,
                                   float* dx, void* stream) {
    if (rows <= 0 || k <= 0 || c <= 0 || mode < 0 || mode > 3) return PN2_EINVAL;
    if (!x || !dout || !dx || (mode == kPoolWeightedAvg && !gxyz)) return PN2_ENULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    group_pool_grad_kernel<1><<<pool_grid(rows * c), 256, 0, st>>>(rows, k, c, mode, x, gxyz, dout, dx);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
