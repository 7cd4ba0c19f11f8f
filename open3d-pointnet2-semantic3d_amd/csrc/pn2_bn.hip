// pn2_bn.hip -- training-mode batch normalisation (+ ReLU) of a dense layer's output, forward and backward, for the
// data-parallel training path (reference: util/tf_util.py:555-581 batch_norm_template -> tf.contrib.layers.batch_norm,
// applied after every conv2d / conv1d of the SA / FP stack, tf_util.py:186-204, followed by tf.nn.relu).
//
// Layout: y (rows, c) row-major ("channels last"), the same buffer pn2_linear / the GEMM wrote.  All kernels are
// HBM streams: the forward reads y twice (statistics, then normalise) and writes z once; the backward reads (dz, y)
// twice and writes dy once.  With pool > 1 the SA layer's max over the K rows of a neighbourhood
// (pointnet_util.py:167-170) rides along: the forward writes only the pooled maxima + tie counts, the backward reads
// the pooled gradient.  Per-channel sums are carried in fp64 (full-rate on CDNA4, the kernels are memory-bound
// anyway) so that var = E[y^2] - E[y]^2 has no cancellation problem at fp32 accuracy.  Partial sums of a block meet in
// LDS and leave with one fp64 atomic per channel -- but atomics on ONE address retire serially (~0.1 us each, measured:
// 2048 blocks on 2*c addresses cost 190 us), so the blocks spread over up to kBnSlots copies of the accumulators (<= 32
// atomics per address) and a one-block launch folds the copies into the final 2*c sums.
#include "pn2_common.h"

namespace {

constexpr int kBnMaxC = 1024;   // channels held in LDS by the apply kernels
constexpr int kBnThreads = 256;
constexpr int kBnSlots = kPn2BnSlots;  // most copies of the per-channel accumulators the reduction blocks spread their atomics over
constexpr int kBnBlocks = 512;   // most reduction blocks (2 per CU, 8 x 16-byte loads in flight per thread); the training step is
                                 // flat from 256 to 1024 and slower at 2048 (pn2_debug_set(10, v) sweep: every block ends in 2*c atomics)

constexpr int kBnHead = kPn2BnHead;

// workspace layout: pn2_common.h
__host__ __device__ inline size_t bn_ws_doubles(int c, int nslots) { return pn2_bn_ws_doubles(c, nslots); }

// thread -> (row slot rr, float4 column cc): cv = c/VEC columns, rp = 256/cv rows per pass
template <int VEC>
struct BnMap {
    int cv, rp, rr, cc;
    bool active;
    __device__ BnMap(int c) {
        cv = c / VEC;
        rp = kBnThreads / cv;
        rr = (int)threadIdx.x / cv;
        cc = (int)threadIdx.x - rr * cv;
        active = rr < rp;
    }
};

template <int VEC>
__device__ __forceinline__ void bn_load(const float* __restrict__ p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}

// add the per-thread partial sums (NS doubles per channel of the thread's VEC channels) over the rp row slots of
// the block and push them to this block's slot copy of the accumulators (bn_fold_kernel adds the copies up into
// ws[kBnHead .. kBnHead+NS*c), the sums every consumer reads)
template <int VEC, int NS>
__device__ __forceinline__ void bn_block_sums(const BnMap<VEC>& mp, int c, int nslots, double (&part)[NS][VEC],
                                              double* __restrict__ ws, const Pn2BnFinish& fin) {
    __shared__ double red[kBnThreads * NS * VEC];
    if (mp.active) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int v = 0; v < VEC; ++v) red[((s * VEC + v) * mp.rp + mp.rr) * mp.cv + mp.cc] = part[s][v];
    }
    __syncthreads();
    // tree over the row slots (sz need not be a power of two)
    for (int sz = mp.rp; sz > 1;) {
        const int h = (sz + 1) >> 1;
        if (mp.active && mp.rr + h < sz) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    red[((s * VEC + v) * mp.rp + mp.rr) * mp.cv + mp.cc] += red[((s * VEC + v) * mp.rp + mp.rr + h) * mp.cv + mp.cc];
        }
        __syncthreads();
        sz = h;
    }
    const unsigned my = blockIdx.x % (unsigned)nslots;
    double* __restrict__ slot = ws + kBnHead + (size_t)NS * c * (1 + my);
    if (mp.active && mp.rr == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                atomicAdd(&slot[(size_t)s * c + mp.cc * VEC + v], red[((s * VEC + v) * mp.rp) * mp.cv + mp.cc]);
    }
    pn2_bn_finish(fin, gridDim.x, blockIdx.x);  // the last workgroup folds the copies (and derives the constants)
}

// final[col] = sum over the slot copies.  A launch of its own rather than a "last block folds" epilogue: that variant
// needs a ticket counter and device-scope fences in every reduction block and measured the same step time (8.2-8.7 ms
// for 128-512 blocks) -- the kernel boundary gives the ordering for free.
// (Eight threads per column with all their copies in flight at once measured the same 4.9 us per launch: the duration of
// this kernel is a fixed cost, not its eight dependent L2 round trips.)
__global__ void __launch_bounds__(kBnThreads)
bn_fold_kernel(int cols, int nslots, double* __restrict__ ws) {
    const int col = blockIdx.x * kBnThreads + threadIdx.x;
    if (col >= cols) return;
    double t = 0.0;
#pragma unroll 8
    for (int k = 0; k < nslots; ++k) t += ws[kBnHead + (size_t)cols * (1 + k) + col];
    ws[kBnHead + col] = t;
}

// acc[0][ch] = sum_r y[r][ch], acc[1][ch] = sum_r y[r][ch]^2 over the block's slab of rows
template <int VEC>
__global__ void __launch_bounds__(kBnThreads)
bn_stats_kernel(long long rows, int c, long long slab, int nslots, const float* __restrict__ y, double* __restrict__ ws,
                Pn2BnFinish fin) {
    const BnMap<VEC> mp(c);
    const long long rb = (long long)blockIdx.x * slab;
    const long long re = rb + slab < rows ? rb + slab : rows;
    double part[2][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) part[0][v] = part[1][v] = 0.0;
    if (mp.active) {
        const float* __restrict__ p = y + (size_t)mp.cc * VEC;
        long long r = rb + mp.rr;
        const long long step = mp.rp;
        for (; r + 7 * step < re; r += 8 * step) {  // eight independent loads in flight
            float a[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; ++u) bn_load<VEC>(p + (size_t)(r + u * step) * c, a[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double d = (double)a[u][v];
                    part[0][v] += d;
                    part[1][v] = __builtin_fma(d, d, part[1][v]);
                }
        }
        for (; r < re; r += step) {
            float a[VEC];
            bn_load<VEC>(p + (size_t)r * c, a);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double d = (double)a[v];
                part[0][v] += d;
                part[1][v] = __builtin_fma(d, d, part[1][v]);
            }
        }
    }
    bn_block_sums<VEC, 2>(mp, c, nslots, part, ws, fin);
}

// The first layer of an SA module whose points carry FEW channels (the level-0 module: xyz + rgb, util/pointnet_util.py:39-54 then
// tf_util.py:181-186) computed where its input is gathered: row r = (cloud, centre j, neighbour k),
//     in = [xyz[idx[r]] - new_xyz[j] | points[idx[r], 0:c]]   (CIN = 3 + c <= 8 values),   y[r, :] = in @ W (CIN, cout),
// one thread per (row, 4 output channels) -- W's CIN rows of its four columns in registers, an fma chain in k order -- with the
// batch statistics of y taken on the way out (the mapping and the block reduction of bn_stats_kernel) and the grouped input xg
// (rows, CIN) kept for the weight gradient.  An MFMA tile would pad K = 6 to 32 behind scalar operand loads (measured 50 us for
// 524288 x 6 -> 32); this is bound by the write of y (67 MB).  cout % 4 == 0, cout <= 1024.
template <int CIN>
__global__ void __launch_bounds__(kBnThreads)
sa_first_layer_stats_kernel(long long rows, int n, int m, int nsample, int cout, long long slab, int nslots,
                            const float* __restrict__ xyz, const float* __restrict__ new_xyz, const float* __restrict__ points,
                            const int* __restrict__ idx, const float* __restrict__ w, float* __restrict__ y,
                            float* __restrict__ xg, double* __restrict__ ws, Pn2BnFinish fin) {
    constexpr int C = CIN - 3;
    const BnMap<4> mp(cout);
    const long long rb = (long long)blockIdx.x * slab;
    const long long re = rb + slab < rows ? rb + slab : rows;
    double part[2][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) part[0][v] = part[1][v] = 0.0;
    if (mp.active) {
        float4 wr[CIN];
#pragma unroll
        for (int k = 0; k < CIN; ++k) wr[k] = *reinterpret_cast<const float4*>(w + (size_t)k * cout + (size_t)mp.cc * 4);
        const long long per_cloud = (long long)m * nsample;
        constexpr int U = 4;  // rows in flight per thread: idx -> gather is a dependent chain of two round trips
        for (long long r0 = rb + mp.rr; r0 < re; r0 += (long long)U * mp.rp) {
            int ii[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long r = r0 + (long long)u * mp.rp;
                ii[u] = idx[r < re ? r : re - 1];
            }
            float in[U][CIN];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long r = r0 + (long long)u * mp.rp;
                const long long rc = r < re ? r : re - 1;
                const unsigned bi = (unsigned)rc / (unsigned)per_cloud;  // rows < 2^31 (checked by the entry point)
                const unsigned grp = (unsigned)rc / (unsigned)nsample;   // = bi * m + j
                const size_t src = (size_t)bi * n + (size_t)ii[u];
#pragma unroll
                for (int a = 0; a < 3; ++a) in[u][a] = xyz[src * 3 + a] - new_xyz[(size_t)grp * 3 + a];
#pragma unroll
                for (int a = 0; a < C; ++a) in[u][3 + a] = points[src * C + a];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long r = r0 + (long long)u * mp.rp;
                if (r >= re) break;
                float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < CIN; ++k) {
                    o[0] = __builtin_fmaf(in[u][k], wr[k].x, o[0]);
                    o[1] = __builtin_fmaf(in[u][k], wr[k].y, o[1]);
                    o[2] = __builtin_fmaf(in[u][k], wr[k].z, o[2]);
                    o[3] = __builtin_fmaf(in[u][k], wr[k].w, o[3]);
                }
                *reinterpret_cast<float4*>(y + (size_t)r * cout + (size_t)mp.cc * 4) = make_float4(o[0], o[1], o[2], o[3]);
                if (xg && mp.cc == 0) {
#pragma unroll
                    for (int k = 0; k < CIN; ++k) xg[(size_t)r * CIN + k] = in[u][k];
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double d = (double)o[v];
                    part[0][v] += d;
                    part[1][v] = __builtin_fma(d, d, part[1][v]);
                }
            }
        }
    }
    bn_block_sums<4, 2>(mp, cout, nslots, part, ws, fin);
}

// bn_scale_shift (the per-channel constants of the normalisation): pn2_common.h

// Per-channel (scale, shift) of the forward into LDS.  Every block derives (mean, invstd) of all channels from the fp64
// sums (c rsqrt's: noise next to its share of the stream); block 0 also publishes them for the backward and moves
// the running averages.
__device__ __forceinline__ void bn_forward_constants(long long rows, int c, const double* __restrict__ acc,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ bias, float eps, float decay,
                                                     float* __restrict__ running_mean, float* __restrict__ running_var,
                                                     float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                     float* sc, float* sh) {
    const double inv_n = 1.0 / (double)rows;
    for (int ch = threadIdx.x; ch < c; ch += kBnThreads) {
        const double mean_d = acc[ch] * inv_n;
        double var_d = acc[(size_t)c + ch] * inv_n - mean_d * mean_d;
        var_d = var_d > 0.0 ? var_d : 0.0;
        const float mean = (float)mean_d;
        const float invstd = (float)(1.0 / __builtin_sqrt(var_d + (double)eps));
        bn_scale_shift(gamma[ch], beta[ch], mean, invstd, sc[ch], sh[ch]);
        if (blockIdx.x == 0) {
            save_mean[ch] = mean;
            save_invstd[ch] = invstd;
            if (running_mean) {
                // moving averages as tf's fused batch norm keeps them: the mean of the layer output INCLUDING the
                // bias the caller folded away (a per-channel constant before BN only moves the mean), and the
                // unbiased batch variance
                const double m_out = mean_d + (bias ? (double)bias[ch] : 0.0);
                const double var_unb = rows > 1 ? var_d * ((double)rows / (double)(rows - 1)) : var_d;
                running_mean[ch] = (float)((double)decay * running_mean[ch] + (1.0 - (double)decay) * m_out);
                running_var[ch] = (float)((double)decay * running_var[ch] + (1.0 - (double)decay) * var_unb);
            }
        }
    }
    __syncthreads();
}

// The deferred form of the forward (pn2_bn_relu_forward_deferred): fold the slot copies and publish the per-channel constants
// -- saved moments, moving averages, and (scale, shift) of z = relu?(fma(y, scale, shift)) -- without writing z: the NEXT
// layer's GEMM and weight gradient apply them while loading y (pn2_linear_bn_stats_xf, pn2_linear_wgrad_accumulate_xf).
__global__ void __launch_bounds__(kBnThreads)
bn_constants_kernel(long long rows, int c, int nslots, double* __restrict__ ws, const float* __restrict__ gamma,
                    const float* __restrict__ beta, const float* __restrict__ bias, float eps, float decay,
                    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ save_mean,
                    float* __restrict__ save_invstd, float* __restrict__ scale, float* __restrict__ shift) {
    const int ch = blockIdx.x * kBnThreads + threadIdx.x;
    if (ch >= c) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
    for (int k = 0; k < nslots; ++k) {
        s1 += ws[kBnHead + (size_t)2 * c * (1 + k) + ch];
        s2 += ws[kBnHead + (size_t)2 * c * (1 + k) + c + ch];
    }
    ws[kBnHead + ch] = s1;
    ws[kBnHead + c + ch] = s2;
    const double inv_n = 1.0 / (double)rows;
    const double mean_d = s1 * inv_n;
    double var_d = s2 * inv_n - mean_d * mean_d;
    var_d = var_d > 0.0 ? var_d : 0.0;
    const float mean = (float)mean_d;
    const float invstd = (float)(1.0 / __builtin_sqrt(var_d + (double)eps));
    float sc, sh;
    bn_scale_shift(gamma[ch], beta[ch], mean, invstd, sc, sh);
    scale[ch] = sc;
    shift[ch] = sh;
    save_mean[ch] = mean;
    save_invstd[ch] = invstd;
    if (running_mean) {  // as bn_forward_constants
        const double m_out = mean_d + (bias ? (double)bias[ch] : 0.0);
        const double var_unb = rows > 1 ? var_d * ((double)rows / (double)(rows - 1)) : var_d;
        running_mean[ch] = (float)((double)decay * running_mean[ch] + (1.0 - (double)decay) * m_out);
        running_var[ch] = (float)((double)decay * running_var[ch] + (1.0 - (double)decay) * var_unb);
    }
}

// The backward twin of bn_constants_kernel (pn2_bn_grad_constants): fold the slot copies of (sum g, sum g * xhat) and publish
// dgamma / dbeta and the six per-channel constants of dy = sc * fma(-xhat, k2, g - k1) -- the float values bn_grad_apply_kernel
// derives in every block -- for the gradient GEMMs that form dy while they load (y, dz) (Pn2GradOnLoad).
__global__ void __launch_bounds__(kBnThreads)
bn_grad_constants_kernel(long long rows, int c, int nslots, double* __restrict__ ws, const float* __restrict__ gamma,
                         const float* __restrict__ beta, const float* __restrict__ save_mean,
                         const float* __restrict__ save_invstd, float* __restrict__ coef, float* __restrict__ dgamma,
                         float* __restrict__ dbeta) {
    const int ch = blockIdx.x * kBnThreads + threadIdx.x;
    if (ch >= c) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
    for (int k = 0; k < nslots; ++k) {
        s1 += ws[kBnHead + (size_t)2 * c * (1 + k) + ch];
        s2 += ws[kBnHead + (size_t)2 * c * (1 + k) + c + ch];
    }
    ws[kBnHead + ch] = s1;
    ws[kBnHead + c + ch] = s2;
    const double inv_n = 1.0 / (double)rows;
    const float mu = save_mean[ch], is = save_invstd[ch];
    float sc, sh;
    bn_scale_shift(gamma[ch], beta[ch], mu, is, sc, sh);
    coef[ch] = sc;
    coef[(size_t)c + ch] = sh;
    coef[(size_t)2 * c + ch] = mu;
    coef[(size_t)3 * c + ch] = is;
    coef[(size_t)4 * c + ch] = (float)(s1 * inv_n);
    coef[(size_t)5 * c + ch] = (float)(s2 * inv_n);
    dbeta[ch] = (float)s1;
    dgamma[ch] = (float)s2;
}

// z = relu?(fma(y, sc[ch], sh[ch]))
template <int VEC>
__global__ void __launch_bounds__(kBnThreads)
bn_apply_kernel(long long rows, int c, const float* __restrict__ y, const double* __restrict__ acc,
                const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ bias,
                float eps, float decay, int relu, float* __restrict__ running_mean, float* __restrict__ running_var,
                float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ z) {
    __shared__ float sc[kBnMaxC], sh[kBnMaxC];
    bn_forward_constants(rows, c, acc, gamma, beta, bias, eps, decay, running_mean, running_var, save_mean, save_invstd, sc, sh);
    const size_t total = (size_t)rows * c / VEC;
    const size_t stride = (size_t)gridDim.x * kBnThreads;
    const bool pow2 = (c & (c - 1)) == 0;
    for (size_t e = (size_t)blockIdx.x * kBnThreads + threadIdx.x; e < total; e += stride) {
        const size_t f = e * VEC;
        const int ch = pow2 ? (int)(f & (size_t)(c - 1)) : (int)(f % (size_t)c);
        float v[VEC];
        bn_load<VEC>(y + f, v);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float t = __builtin_fmaf(v[k], sc[ch + k], sh[ch + k]);
            if (relu) t = t > 0.f ? t : 0.f;
            v[k] = t;
        }
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(z + f) = make_float4(v[0], v[1], v[2], v[3]);
        else z[f] = v[0];
    }
}

// The same followed by the max over each group of `pool` consecutive rows (the SA layer's tf.reduce_max over the K
// neighbours, pointnet_util.py:167-170): zmax (rows/pool, c) and the number of rows that attain it (tf / torch give
// every tied row an equal share of the gradient; duplicated neighbours of a sparse ball tie exactly).  The (rows, c)
// activation is never written.  thread -> (group slot, 16-byte column); 8 rows in flight.
template <int VEC>
__global__ void __launch_bounds__(kBnThreads)
bn_apply_pool_kernel(long long rows, int c, int pool, const float* __restrict__ y, const double* __restrict__ acc,
                     const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ bias,
                     float eps, float decay, int relu, float* __restrict__ running_mean, float* __restrict__ running_var,
                     float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ zmax,
                     float* __restrict__ ties, float* __restrict__ ysel) {
    __shared__ float sc[kBnMaxC], sh[kBnMaxC];
    bn_forward_constants(rows, c, acc, gamma, beta, bias, eps, decay, running_mean, running_var, save_mean, save_invstd, sc, sh);
    const BnMap<VEC> mp(c);
    if (!mp.active) return;
    const long long groups = rows / pool;
    float s4[VEC], h4[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s4[v] = sc[mp.cc * VEC + v]; h4[v] = sh[mp.cc * VEC + v]; }
    for (long long g = (long long)blockIdx.x * mp.rp + mp.rr; g < groups; g += (long long)gridDim.x * mp.rp) {
        const float* __restrict__ p = y + (size_t)g * pool * c + (size_t)mp.cc * VEC;
        float best[VEC], cnt[VEC], ysl[VEC];  // ysl: the pre-normalisation value of the FIRST row attaining the maximum
#pragma unroll
        for (int v = 0; v < VEC; ++v) { best[v] = -__builtin_inff(); cnt[v] = 0.f; ysl[v] = 0.f; }
        auto take = [&](const float (&a)[VEC]) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float t = __builtin_fmaf(a[v], s4[v], h4[v]);
                if (relu) t = t > 0.f ? t : 0.f;
                cnt[v] = t > best[v] ? 1.f : (t == best[v] ? cnt[v] + 1.f : cnt[v]);
                ysl[v] = t > best[v] ? a[v] : ysl[v];
                best[v] = t > best[v] ? t : best[v];
            }
        };
        int r = 0;
        for (; r + 7 < pool; r += 8) {
            float a[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; ++u) bn_load<VEC>(p + (size_t)(r + u) * c, a[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) take(a[u]);
        }
        for (; r < pool; ++r) {
            float a[VEC];
            bn_load<VEC>(p + (size_t)r * c, a);
            take(a);
        }
        const size_t o = (size_t)g * c + (size_t)mp.cc * VEC;
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(zmax + o) = make_float4(best[0], best[1], best[2], best[3]);
            *reinterpret_cast<float4*>(ties + o) = make_float4(cnt[0], cnt[1], cnt[2], cnt[3]);
            if (ysel) *reinterpret_cast<float4*>(ysel + o) = make_float4(ysl[0], ysl[1], ysl[2], ysl[3]);
        } else {
            zmax[o] = best[0];
            ties[o] = cnt[0];
            if (ysel) ysel[o] = ysl[0];
        }
    }
}

// gradient reaching the (un-pooled) activation of row `row`: dz itself, or -- behind the fused max pool -- the pooled
// gradient shared equally among the rows of the group that attain the maximum.  t = the forward value of the element.
template <int VEC>
__device__ __forceinline__ void bn_incoming_grad(const float* __restrict__ dz, int pool, const float* __restrict__ zmax,
                                                 const float* __restrict__ ties, long long row, int c, size_t col,
                                                 const float (&t)[VEC], float (&g)[VEC]) {
    if (pool <= 1) {
        bn_load<VEC>(dz + (size_t)row * c + col, g);
        return;
    }
    const size_t o = (size_t)(row / pool) * c + col;
    float d[VEC], m[VEC], n[VEC];
    bn_load<VEC>(dz + o, d);
    bn_load<VEC>(zmax + o, m);
    bn_load<VEC>(ties + o, n);
#pragma unroll
    for (int v = 0; v < VEC; ++v) g[v] = t[v] == m[v] ? d[v] / n[v] : 0.f;
}

// backward pass 1: acc[0][ch] = sum_r g, acc[1][ch] = sum_r g * xhat, with g = dz * [z > 0] (relu; dz through the
// fused max pool when pool > 1) and xhat = (y - mean) * invstd
template <int VEC>
__global__ void __launch_bounds__(kBnThreads)
bn_grad_reduce_kernel(long long rows, int c, long long slab, int nslots, const float* __restrict__ dz, const float* __restrict__ y,
                      const float* __restrict__ gamma, const float* __restrict__ beta,
                      const float* __restrict__ save_mean, const float* __restrict__ save_invstd, int relu, int pool,
                      const float* __restrict__ zmax, const float* __restrict__ ties, double* __restrict__ ws,
                      Pn2BnFinish fin) {
    const BnMap<VEC> mp(c);
    const long long rb = (long long)blockIdx.x * slab;
    const long long re = rb + slab < rows ? rb + slab : rows;
    double part[2][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) part[0][v] = part[1][v] = 0.0;
    if (mp.active) {
        float sc[VEC], sh[VEC], mean[VEC], invstd[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int ch = mp.cc * VEC + v;
            mean[v] = save_mean[ch];
            invstd[v] = save_invstd[ch];
            bn_scale_shift(gamma[ch], beta[ch], mean[v], invstd[v], sc[v], sh[v]);
        }
        const size_t col = (size_t)mp.cc * VEC;
        auto take = [&](long long row, const float (&a)[VEC]) {
            float t[VEC], g[VEC];
            bool on[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float lin = __builtin_fmaf(a[v], sc[v], sh[v]);
                on[v] = !relu || lin > 0.f;
                t[v] = on[v] ? lin : 0.f;
            }
            bn_incoming_grad<VEC>(dz, pool, zmax, ties, row, c, col, t, g);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double gd = on[v] ? (double)g[v] : 0.0;
                const double xh = (double)((a[v] - mean[v]) * invstd[v]);
                part[0][v] += gd;
                part[1][v] = __builtin_fma(gd, xh, part[1][v]);
            }
        };
        long long r = rb + mp.rr;
        const long long step = mp.rp;
        for (; r + 3 * step < re; r += 4 * step) {  // four rows in flight
            float a[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) bn_load<VEC>(y + (size_t)(r + u * step) * c + col, a[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) take(r + u * step, a[u]);
        }
        for (; r < re; r += step) {
            float a0[VEC];
            bn_load<VEC>(y + (size_t)r * c + col, a0);
            take(r, a0);
        }
    }
    bn_block_sums<VEC, 2>(mp, c, nslots, part, ws, fin);
}

// backward pass 1 behind the fused max pool, from the POOLED tensors alone (groups = rows / pool entries per channel instead of
// rows): the gradient reaching the activation is non-zero only on the rows that attain a group's maximum, where it is dzp / n on
// each of the n tied rows -- all of them have the forward value zmax -- so sum g = sum over groups of dzp and
// sum g * xhat = sum over groups of dzp * xhat(ysel), ysel = the pre-normalisation value of the first such row (kept by the
// forward, bn_apply_pool_kernel); a group whose maximum is the ReLU's 0 passes nothing.  Same sums as bn_grad_reduce_kernel up to
// rounding (n * fl(dzp / n) vs dzp; tied rows whose y differ inside one rounding of the normalisation), without its pass over y.
template <int VEC>
__global__ void __launch_bounds__(kBnThreads)
bn_grad_reduce_pooled_kernel(long long groups, int c, long long slab, int nslots, const float* __restrict__ dzp,
                             const float* __restrict__ zmax, const float* __restrict__ ysel, const float* __restrict__ save_mean,
                             const float* __restrict__ save_invstd, int relu, double* __restrict__ ws, Pn2BnFinish fin) {
    const BnMap<VEC> mp(c);
    const long long gb = (long long)blockIdx.x * slab;
    const long long ge = gb + slab < groups ? gb + slab : groups;
    double part[2][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) part[0][v] = part[1][v] = 0.0;
    if (mp.active) {
        float mean[VEC], invstd[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { mean[v] = save_mean[mp.cc * VEC + v]; invstd[v] = save_invstd[mp.cc * VEC + v]; }
        const size_t col = (size_t)mp.cc * VEC;
        for (long long g = gb + mp.rr; g < ge; g += mp.rp) {
            float d[VEC], m[VEC], ys[VEC];
            bn_load<VEC>(dzp + (size_t)g * c + col, d);
            bn_load<VEC>(zmax + (size_t)g * c + col, m);
            bn_load<VEC>(ysel + (size_t)g * c + col, ys);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double gd = (!relu || m[v] > 0.f) ? (double)d[v] : 0.0;
                const double xh = (double)((ys[v] - mean[v]) * invstd[v]);
                part[0][v] += gd;
                part[1][v] = __builtin_fma(gd, xh, part[1][v]);
            }
        }
    }
    bn_block_sums<VEC, 2>(mp, c, nslots, part, ws, fin);
}

// backward pass 2: dy = sc * (g - mean(g) - xhat * mean(g * xhat)); block 0 publishes dgamma = sum g*xhat, dbeta = sum g
template <int VEC>
__global__ void __launch_bounds__(kBnThreads)
bn_grad_apply_kernel(long long rows, int c, const float* __restrict__ dz, const float* __restrict__ y,
                     const double* __restrict__ acc, const float* __restrict__ gamma, const float* __restrict__ beta,
                     const float* __restrict__ save_mean, const float* __restrict__ save_invstd, int relu, int pool,
                     const float* __restrict__ zmax, const float* __restrict__ ties,
                     float* __restrict__ dy, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sc[kBnMaxC], sh[kBnMaxC], mu[kBnMaxC], is[kBnMaxC], k1[kBnMaxC], k2[kBnMaxC];
    const double inv_n = 1.0 / (double)rows;
    for (int ch = threadIdx.x; ch < c; ch += kBnThreads) {
        mu[ch] = save_mean[ch];
        is[ch] = save_invstd[ch];
        bn_scale_shift(gamma[ch], beta[ch], mu[ch], is[ch], sc[ch], sh[ch]);
        k1[ch] = (float)(acc[ch] * inv_n);
        k2[ch] = (float)(acc[(size_t)c + ch] * inv_n);
        if (blockIdx.x == 0) {
            dbeta[ch] = (float)acc[ch];
            dgamma[ch] = (float)acc[(size_t)c + ch];
        }
    }
    __syncthreads();
    const size_t total = (size_t)rows * c / VEC;
    const size_t stride = (size_t)gridDim.x * kBnThreads;
    const bool pow2 = (c & (c - 1)) == 0;
    for (size_t e = (size_t)blockIdx.x * kBnThreads + threadIdx.x; e < total; e += stride) {
        const size_t f = e * VEC;
        const int ch = pow2 ? (int)(f & (size_t)(c - 1)) : (int)(f % (size_t)c);
        float g[VEC], a[VEC], t[VEC];
        bool on[VEC];
        bn_load<VEC>(y + f, a);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float lin = __builtin_fmaf(a[k], sc[ch + k], sh[ch + k]);
            on[k] = !relu || lin > 0.f;
            t[k] = on[k] ? lin : 0.f;
        }
        if (pool <= 1) bn_load<VEC>(dz + f, g);
        else bn_incoming_grad<VEC>(dz, pool, zmax, ties, (long long)((f - (size_t)ch) / (size_t)c), c, (size_t)ch, t, g);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int cc = ch + k;
            const float gk = on[k] ? g[k] : 0.f;
            const float xh = (a[k] - mu[cc]) * is[cc];
            g[k] = sc[cc] * __builtin_fmaf(-xh, k2[cc], gk - k1[cc]);
        }
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(dy + f) = make_float4(g[0], g[1], g[2], g[3]);
        else dy[f] = g[0];
    }
}

PN2_TUNABLE(int, g_bn_blocks, 0)  // tuning hook (pn2_debug_set(10, v)): reduction blocks, 0 = kBnBlocks

struct BnPlan {
    int vec, nslots;
    int stat_blocks, apply_blocks;
    long long slab;
};

int bn_plan(long long rows, int c, const void* a, const void* b, const void* o, BnPlan& p) {
    if (rows <= 0 || c <= 0) return PN2_EINVAL;
    const bool al = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)o) % 16) == 0;
    p.vec = (c % 4 == 0 && al) ? 4 : 1;
    if (c / p.vec > kBnThreads || c > kBnMaxC) return PN2_EUNSUP;
    const int rp = kBnThreads / (c / p.vec);
    // reductions: kBnBlocks blocks, each with at least 8 passes of rows
    long long blocks = g_bn_blocks > 0 ? g_bn_blocks : kBnBlocks;
    long long slab = (rows + blocks - 1) / blocks;
    const long long min_slab = (long long)rp * 8;
    if (slab < min_slab) slab = min_slab;
    slab = (slab + rp - 1) / rp * rp;
    p.slab = slab;
    p.stat_blocks = (int)((rows + slab - 1) / slab);
    p.nslots = (p.stat_blocks + 31) / 32;  // <= 32 same-address atomics (they retire one after another)
    if (p.nslots > kBnSlots) p.nslots = kBnSlots;
    const long long total = rows * (long long)c / p.vec;
    long long ab = (total + kBnThreads * 4 - 1) / (kBnThreads * 4);  // >= 4 elements per thread
    if (ab > 4096) ab = 4096;
    if (ab < 1) ab = 1;
    p.apply_blocks = (int)ab;
    return PN2_OK;
}

Pn2BnFinish bn_finish_fold(long long rows, int c, int nslots, double* ws) {
    Pn2BnFinish f{};
    f.kind = 1; f.c = c; f.nslots = nslots; f.rows = rows; f.ws = ws;
    return f;
}

}  // namespace

#ifdef PN2_TUNING_HOOKS
extern "C" int pn2_debug_set_bn(int what, int value) {
    if (what == 10) { g_bn_blocks = value; return 0; }
    return PN2_EINVAL;
}
#endif  // PN2_TUNING_HOOKS

extern "C" size_t pn2_bn_workspace_bytes(int c) { return c > 0 ? sizeof(double) * bn_ws_doubles(c, kBnSlots) : 0; }

static int bn_relu_forward_impl(long long rows, int c, const float* y, const float* gamma, const float* beta,
                                const float* bias, float eps, float decay, int relu, int pool, float* running_mean,
                                float* running_var, void* workspace, size_t workspace_bytes, float* save_mean,
                                float* save_invstd, float* z, float* ties, void* stream, int mode, float* ysel = nullptr) {
    // mode 0: zero the workspace here; 1: the caller zeroed it; 2: the caller zeroed it AND pn2_linear_bn_stats has already
    // added the column sums of y to all kBnSlots slot copies (no statistics pass); 3: ... and the GEMM's last workgroup has folded
    // them too (pn2_linear_bn_stats_fin): only the normalisation is left
    if (!y || !gamma || !beta || !workspace || !save_mean || !save_invstd || !z) return PN2_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return PN2_ENULL;
    if (pool > 1 && !ties) return PN2_ENULL;
    if (pool > 1 && rows % pool != 0) return PN2_EINVAL;
    BnPlan p;
    const int rc = bn_plan(rows, c, y, z, ties, p);
    if (rc != PN2_OK) return rc;
    if (workspace_bytes < pn2_bn_workspace_bytes(c) || ((uintptr_t)workspace % 8) != 0) return PN2_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    const double* acc = ws + kBnHead;
    if (mode == 0) {
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(double) * bn_ws_doubles(c, p.nslots), st);
        if (e != hipSuccess) return (int)e;
    }
    if (mode >= 2) p.nslots = kBnSlots;
    const int fold_blocks = (2 * c + kBnThreads - 1) / kBnThreads;
    const Pn2BnFinish fin = bn_finish_fold(rows, c, p.nslots, ws);  // the statistics kernel's last workgroup folds the copies
    long long pb = 1;  // pooled apply: one group per (thread row slot), grid-stride beyond 8 blocks per CU
    if (pool > 1) {
        const int rp = kBnThreads / (c / p.vec);
        pb = (rows / pool + rp - 1) / rp;
        if (pb > 2048) pb = 2048;
    }
#define PN2_BN_FWD(V_)                                                                                                   \
    do {                                                                                                                 \
        if (mode < 2) bn_stats_kernel<V_><<<p.stat_blocks, kBnThreads, 0, st>>>(rows, c, p.slab, p.nslots, y, ws, fin);   \
        if (mode == 2) bn_fold_kernel<<<fold_blocks, kBnThreads, 0, st>>>(2 * c, p.nslots, ws);                           \
        if (pool > 1)                                                                                                    \
            bn_apply_pool_kernel<V_><<<(int)pb, kBnThreads, 0, st>>>(rows, c, pool, y, acc, gamma, beta, bias, eps, decay, \
                                                                   relu, running_mean, running_var, save_mean,           \
                                                                   save_invstd, z, ties, ysel);                          \
        else                                                                                                             \
            bn_apply_kernel<V_><<<p.apply_blocks, kBnThreads, 0, st>>>(rows, c, y, acc, gamma, beta, bias, eps, decay,    \
                                                                     relu, running_mean, running_var, save_mean,         \
                                                                     save_invstd, z);                                    \
    } while (0)
    if (p.vec == 4) PN2_BN_FWD(4);
    else PN2_BN_FWD(1);
#undef PN2_BN_FWD
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_bn_relu_forward(long long rows, int c, const float* y, const float* gamma, const float* beta,
                                   const float* bias, float eps, float decay, int relu, int pool, float* running_mean,
                                   float* running_var, void* workspace, size_t workspace_bytes, float* save_mean,
                                   float* save_invstd, float* z, float* ties, void* stream) {
    return bn_relu_forward_impl(rows, c, y, gamma, beta, bias, eps, decay, relu, pool, running_mean, running_var, workspace,
                                workspace_bytes, save_mean, save_invstd, z, ties, stream, 0);
}
// the same with a workspace the CALLER has already zero-filled (one fill of an arena that holds the scratch of every layer
// of a training step replaces one memset per call)
extern "C" int pn2_bn_relu_forward_ws0(long long rows, int c, const float* y, const float* gamma, const float* beta,
                                       const float* bias, float eps, float decay, int relu, int pool, float* running_mean,
                                       float* running_var, void* workspace, size_t workspace_bytes, float* save_mean,
                                       float* save_invstd, float* z, float* ties, void* stream) {
    return bn_relu_forward_impl(rows, c, y, gamma, beta, bias, eps, decay, relu, pool, running_mean, running_var, workspace,
                                workspace_bytes, save_mean, save_invstd, z, ties, stream, 1);
}
// the same for a y produced by pn2_linear_bn_stats with this workspace: the column sums are already there
extern "C" int pn2_bn_relu_forward_stats(long long rows, int c, const float* y, const float* gamma, const float* beta,
                                         const float* bias, float eps, float decay, int relu, int pool, float* running_mean,
                                         float* running_var, void* workspace, size_t workspace_bytes, float* save_mean,
                                         float* save_invstd, float* z, float* ties, void* stream) {
    return bn_relu_forward_impl(rows, c, y, gamma, beta, bias, eps, decay, relu, pool, running_mean, running_var, workspace,
                                workspace_bytes, save_mean, save_invstd, z, ties, stream, 2);
}

// pn2_bn_relu_forward with pool > 1 that also keeps ysel (rows / pool, c): the pre-normalisation value of the first row attaining
// each pooled maximum -- what lets the backward take its reduction from the pooled tensors alone (pn2_bn_grad_constants).
// stats_mode 0: zero the workspace here; 1: the caller zeroed it; 2: pn2_linear_bn_stats already left the column sums in it;
// 3: pn2_linear_bn_stats_fin left them there AND folded.
extern "C" int pn2_bn_relu_forward_pool(long long rows, int c, const float* y, const float* gamma, const float* beta,
                                        const float* bias, float eps, float decay, int relu, int pool, float* running_mean,
                                        float* running_var, void* workspace, size_t workspace_bytes, int stats_mode,
                                        float* save_mean, float* save_invstd, float* zmax, float* ties, float* ysel,
                                        void* stream) {
    if (pool <= 1 || stats_mode < 0 || stats_mode > 3) return PN2_EINVAL;
    if (!ysel) return PN2_ENULL;
    if ((c % 4 == 0) && ((uintptr_t)ysel % 16) != 0) return PN2_EINVAL;
    return bn_relu_forward_impl(rows, c, y, gamma, beta, bias, eps, decay, relu, pool, running_mean, running_var, workspace,
                                workspace_bytes, save_mean, save_invstd, zmax, ties, stream, stats_mode, ysel);
}

// Batch norm of the training path WITHOUT writing the normalised activation: the statistics (stats_done = 1: already left in
// the zeroed workspace by pn2_linear_bn_stats; 0: taken here with one pass over y, workspace zeroed by the caller) are folded
// and turned into save_mean / save_invstd, the moving averages and the per-channel (scale, shift) of
// z = relu?(fma(y, scale, shift)); the consumer applies them while it loads y (pn2_linear_bn_stats_xf /
// pn2_linear_wgrad_accumulate_xf), so the write and the re-read of z disappear.  tf_util.py:555-581 followed by :186-204 of the
// next layer.
extern "C" int pn2_bn_relu_forward_deferred(long long rows, int c, const float* y, const float* gamma, const float* beta,
                                            const float* bias, float eps, float decay, int stats_done, float* running_mean,
                                            float* running_var, void* workspace, size_t workspace_bytes, float* save_mean,
                                            float* save_invstd, float* scale, float* shift, void* stream) {
    if (!y || !gamma || !beta || !workspace || !save_mean || !save_invstd || !scale || !shift) return PN2_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return PN2_ENULL;
    BnPlan p;
    const int rc = bn_plan(rows, c, y, y, y, p);
    if (rc != PN2_OK) return rc;
    if (workspace_bytes < pn2_bn_workspace_bytes(c) || ((uintptr_t)workspace % 8) != 0) return PN2_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    if (stats_done) {
        p.nslots = kBnSlots;
        bn_constants_kernel<<<(c + kBnThreads - 1) / kBnThreads, kBnThreads, 0, st>>>(rows, c, p.nslots, ws, gamma, beta, bias, eps,
                                                                                     decay, running_mean, running_var, save_mean,
                                                                                     save_invstd, scale, shift);
    } else {  // one launch: the statistics kernel's last workgroup folds the copies and publishes the constants
        Pn2BnFinish f{};
        f.kind = 2; f.c = c; f.nslots = p.nslots; f.rows = rows; f.ws = ws;
        f.gamma = gamma; f.beta = beta; f.bias = bias; f.eps = eps; f.decay = decay;
        f.running_mean = running_mean; f.running_var = running_var; f.save_mean = save_mean; f.save_invstd = save_invstd;
        f.scale = scale; f.shift = shift;
        if (p.vec == 4) bn_stats_kernel<4><<<p.stat_blocks, kBnThreads, 0, st>>>(rows, c, p.slab, p.nslots, y, ws, f);
        else bn_stats_kernel<1><<<p.stat_blocks, kBnThreads, 0, st>>>(rows, c, p.slab, p.nslots, y, ws, f);
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

static int bn_relu_backward_impl(long long rows, int c, const float* dz, const float* y, const float* gamma,
                                 const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                 int pool, const float* zmax, const float* ties, void* workspace,
                                 size_t workspace_bytes, float* dy, float* dgamma, float* dbeta, void* stream, int mode) {
    // mode 0: zero the workspace here; 1: the caller zeroed it; 2: the caller zeroed it AND pn2_linear_dgrad_bn_grad_stats has
    // already added (sum g, sum g * xhat) to all kBnSlots slot copies while it produced dz (no reduction pass; pool <= 1);
    // 3: ... and that GEMM's last workgroup has folded them (pn2_linear_dgrad_fin, finish kind 1)
    if (!dz || !y || !gamma || !beta || !save_mean || !save_invstd || !workspace || !dy || !dgamma || !dbeta) return PN2_ENULL;
    if (pool > 1 && (!zmax || !ties)) return PN2_ENULL;
    if (pool > 1 && (rows % pool != 0 || dy == dz)) return PN2_EINVAL;
    if (mode >= 2 && pool > 1) return PN2_EINVAL;
    BnPlan p;
    int rc = bn_plan(rows, c, dz, y, dy, p);
    if (rc != PN2_OK) return rc;
    if (pool > 1 && p.vec == 4 && (((uintptr_t)zmax | (uintptr_t)ties) % 16) != 0) return PN2_EINVAL;
    if (workspace_bytes < pn2_bn_workspace_bytes(c) || ((uintptr_t)workspace % 8) != 0) return PN2_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    const double* acc = ws + kBnHead;
    if (mode == 0) {
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(double) * bn_ws_doubles(c, p.nslots), st);
        if (e != hipSuccess) return (int)e;
    }
    if (mode >= 2) p.nslots = kBnSlots;
    const int fold_blocks = (2 * c + kBnThreads - 1) / kBnThreads;
    const Pn2BnFinish fin = bn_finish_fold(rows, c, p.nslots, ws);  // the reduction's last workgroup folds the copies
#define PN2_BN_BWD(V_)                                                                                                  \
    do {                                                                                                                \
        if (mode < 2)                                                                                                   \
            bn_grad_reduce_kernel<V_><<<p.stat_blocks, kBnThreads, 0, st>>>(rows, c, p.slab, p.nslots, dz, y, gamma, beta, \
                                                                          save_mean, save_invstd, relu, pool, zmax, ties, ws, fin); \
        if (mode == 2) bn_fold_kernel<<<fold_blocks, kBnThreads, 0, st>>>(2 * c, p.nslots, ws);                          \
        bn_grad_apply_kernel<V_><<<p.apply_blocks, kBnThreads, 0, st>>>(rows, c, dz, y, acc, gamma, beta, save_mean,     \
                                                                      save_invstd, relu, pool, zmax, ties, dy, dgamma,  \
                                                                      dbeta);                                           \
    } while (0)
    if (p.vec == 4) PN2_BN_BWD(4);
    else PN2_BN_BWD(1);
#undef PN2_BN_BWD
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_bn_relu_backward(long long rows, int c, const float* dz, const float* y, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                    int pool, const float* zmax, const float* ties, void* workspace,
                                    size_t workspace_bytes, float* dy, float* dgamma, float* dbeta, void* stream) {
    return bn_relu_backward_impl(rows, c, dz, y, gamma, beta, save_mean, save_invstd, relu, pool, zmax, ties, workspace,
                                 workspace_bytes, dy, dgamma, dbeta, stream, 0);
}
extern "C" int pn2_bn_relu_backward_ws0(long long rows, int c, const float* dz, const float* y, const float* gamma,
                                        const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                        int pool, const float* zmax, const float* ties, void* workspace,
                                        size_t workspace_bytes, float* dy, float* dgamma, float* dbeta, void* stream) {
    return bn_relu_backward_impl(rows, c, dz, y, gamma, beta, save_mean, save_invstd, relu, pool, zmax, ties, workspace,
                                 workspace_bytes, dy, dgamma, dbeta, stream, 1);
}
// the same for a dz produced by pn2_linear_dgrad_bn_grad_stats with this workspace: the two sums are already there
extern "C" int pn2_bn_relu_backward_stats(long long rows, int c, const float* dz, const float* y, const float* gamma,
                                          const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                          int pool, const float* zmax, const float* ties, void* workspace,
                                          size_t workspace_bytes, float* dy, float* dgamma, float* dbeta, void* stream) {
    return bn_relu_backward_impl(rows, c, dz, y, gamma, beta, save_mean, save_invstd, relu, pool, zmax, ties, workspace,
                                 workspace_bytes, dy, dgamma, dbeta, stream, 2);
}

// The batch-norm gradient WITHOUT writing dy: the two per-channel sums (stats_done = 1: already left in the zeroed workspace by
// pn2_linear_dgrad_bn_grad_stats; 0: taken here with one pass over (dz, y), workspace zeroed by the caller) are folded into dgamma,
// dbeta and coef (6, c) = sc, sh, mean, invstd, k1, k2; the layer's data and weight gradient GEMMs apply them while they load
// (y, dz) (pn2_linear_dgrad_gx / pn2_linear_wgrad_gx), so the write of dy and its two re-reads disappear.  pool as in
// pn2_bn_relu_backward.  util/tf_util.py:555-581 via tf.gradients.
extern "C" int pn2_bn_grad_constants(long long rows, int c, const float* dz, const float* y, const float* gamma, const float* beta,
                                     const float* save_mean, const float* save_invstd, int relu, int pool, const float* zmax,
                                     const float* ties, const float* ysel, int stats_done, void* workspace,
                                     size_t workspace_bytes, float* coef, float* dgamma, float* dbeta, void* stream) {
    if (!dz || !y || !gamma || !beta || !save_mean || !save_invstd || !workspace || !coef || !dgamma || !dbeta) return PN2_ENULL;
    if (pool > 1 && (!zmax || !ties)) return PN2_ENULL;
    if (pool > 1 && rows % pool != 0) return PN2_EINVAL;
    if (stats_done && pool > 1) return PN2_EINVAL;
    BnPlan p;
    const int rc = bn_plan(rows, c, dz, y, y, p);
    if (rc != PN2_OK) return rc;
    if (pool > 1 && p.vec == 4 && (((uintptr_t)zmax | (uintptr_t)ties) % 16) != 0) return PN2_EINVAL;
    if (workspace_bytes < pn2_bn_workspace_bytes(c) || ((uintptr_t)workspace % 8) != 0) return PN2_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    if (stats_done) {
        p.nslots = kBnSlots;
        bn_grad_constants_kernel<<<(c + kBnThreads - 1) / kBnThreads, kBnThreads, 0, st>>>(rows, c, p.nslots, ws, gamma, beta,
                                                                                          save_mean, save_invstd, coef, dgamma, dbeta);
        PN2_RETURN_IF_LAUNCH_FAILED();
        return PN2_OK;
    }
    // one launch: the reduction's last workgroup folds the copies and publishes the constants
    Pn2BnFinish f{};
    f.kind = 3; f.c = c; f.rows = rows; f.ws = ws;
    f.gamma = gamma; f.beta = beta; f.mean_in = save_mean; f.invstd_in = save_invstd;
    f.coef = coef; f.dgamma = dgamma; f.dbeta = dbeta;
    if (pool > 1 && ysel) {  // the reduction from the pooled tensors alone: rows / pool entries per channel
        BnPlan q;
        const int rq = bn_plan(rows / pool, c, dz, zmax, ysel, q);
        if (rq != PN2_OK) return rq;
        f.nslots = q.nslots;
        if (q.vec == 4)
            bn_grad_reduce_pooled_kernel<4><<<q.stat_blocks, kBnThreads, 0, st>>>(rows / pool, c, q.slab, q.nslots, dz, zmax, ysel,
                                                                                save_mean, save_invstd, relu, ws, f);
        else
            bn_grad_reduce_pooled_kernel<1><<<q.stat_blocks, kBnThreads, 0, st>>>(rows / pool, c, q.slab, q.nslots, dz, zmax, ysel,
                                                                                save_mean, save_invstd, relu, ws, f);
    } else {
        f.nslots = p.nslots;
        if (p.vec == 4)
            bn_grad_reduce_kernel<4><<<p.stat_blocks, kBnThreads, 0, st>>>(rows, c, p.slab, p.nslots, dz, y, gamma, beta, save_mean,
                                                                         save_invstd, relu, pool, zmax, ties, ws, f);
        else
            bn_grad_reduce_kernel<1><<<p.stat_blocks, kBnThreads, 0, st>>>(rows, c, p.slab, p.nslots, dz, y, gamma, beta, save_mean,
                                                                         save_invstd, relu, pool, zmax, ties, ws, f);
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// pn2_bn_relu_forward / pn2_bn_relu_backward with the state of the workspace given explicitly (the numbered modes of the two
// implementations above; 3 = the sums are there AND folded, left by pn2_linear_bn_stats_fin / pn2_linear_dgrad_fin with finish
// kind 1): the one-block fold launch between a GEMM and the normalisation pass is gone.
extern "C" int pn2_bn_relu_forward_mode(long long rows, int c, const float* y, const float* gamma, const float* beta,
                                        const float* bias, float eps, float decay, int relu, float* running_mean,
                                        float* running_var, void* workspace, size_t workspace_bytes, int stats_mode,
                                        float* save_mean, float* save_invstd, float* z, void* stream) {
    if (stats_mode < 0 || stats_mode > 3) return PN2_EINVAL;
    return bn_relu_forward_impl(rows, c, y, gamma, beta, bias, eps, decay, relu, 0, running_mean, running_var, workspace,
                                workspace_bytes, save_mean, save_invstd, z, nullptr, stream, stats_mode);
}
extern "C" int pn2_bn_relu_backward_mode(long long rows, int c, const float* dz, const float* y, const float* gamma,
                                         const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                         int pool, const float* zmax, const float* ties, void* workspace,
                                         size_t workspace_bytes, int stats_mode, float* dy, float* dgamma, float* dbeta,
                                         void* stream) {
    if (stats_mode < 0 || stats_mode > 3) return PN2_EINVAL;
    return bn_relu_backward_impl(rows, c, dz, y, gamma, beta, save_mean, save_invstd, relu, pool, zmax, ties, workspace,
                                 workspace_bytes, dy, dgamma, dbeta, stream, stats_mode);
}

// First layer of an SA module with few point channels (c <= 5: the level-0 module's colours), training path, in ONE launch:
// gather + centre + concat (pn2_sa_group_concat) + the (3 + c) -> cout product (tf_util.py:181-186) + the batch statistics of y
// (workspace ZEROED by the caller) + their fold by the launch's last workgroup (finish 1; pn2_bn_relu_forward_mode /
// pn2_bn_relu_forward_pool with stats_mode 3 normalise) or fold + the constants pn2_bn_relu_forward_deferred publishes
// (finish 2).  y (b, m, nsample, cout) un-normalised; xg (b, m, nsample, 3 + c), optional: the grouped input, operand of the
// weight gradient.  cout % 4 == 0, cout <= 1024, 16-byte aligned w / y.
extern "C" int pn2_sa_first_layer_bn(int b, int n, int m, int nsample, int c, int cout, const float* xyz, const float* new_xyz,
                                     const float* points, const int* idx, const float* w, float* y, float* xg,
                                     void* workspace, size_t workspace_bytes, int finish, const float* gamma, const float* beta,
                                     const float* bias, float eps, float decay, float* running_mean, float* running_var,
                                     float* save_mean, float* save_invstd, float* scale, float* shift, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || c < 0 || cout <= 0) return PN2_EINVAL;
    if (!xyz || !new_xyz || !idx || !w || !y || !workspace || (c > 0 && !points)) return PN2_ENULL;
    if (finish != 1 && finish != 2) return PN2_EINVAL;
    if (c > 5 || cout % 4 != 0 || cout > 1024 || (((uintptr_t)w | (uintptr_t)y) % 16) != 0) return PN2_EUNSUP;
    const long long rows = (long long)b * m * nsample;
    if (rows + 128 > 0x7fffffffLL) return PN2_ERANGE;
    BnPlan p;
    const int rc = bn_plan(rows, cout, y, y, y, p);
    if (rc != PN2_OK) return rc;
    if (p.vec != 4) return PN2_EUNSUP;
    if (workspace_bytes < pn2_bn_workspace_bytes(cout) || ((uintptr_t)workspace % 8) != 0) return PN2_EINVAL;
    double* ws = static_cast<double*>(workspace);
    Pn2BnFinish f{};
    f.kind = finish; f.c = cout; f.nslots = p.nslots; f.rows = rows; f.ws = ws;
    if (finish == 2) {
        if (!gamma || !beta || !save_mean || !save_invstd || (scale == nullptr) != (shift == nullptr)) return PN2_ENULL;
        if ((running_mean == nullptr) != (running_var == nullptr)) return PN2_ENULL;
        f.gamma = gamma; f.beta = beta; f.bias = bias; f.eps = eps; f.decay = decay; f.running_mean = running_mean;
        f.running_var = running_var; f.save_mean = save_mean; f.save_invstd = save_invstd; f.scale = scale; f.shift = shift;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
#define PN2_SAF(CIN_)                                                                                                          \
    case CIN_:                                                                                                                 \
        sa_first_layer_stats_kernel<CIN_><<<p.stat_blocks, kBnThreads, 0, st>>>(rows, n, m, nsample, cout, p.slab, p.nslots, xyz, \
                                                                               new_xyz, points, idx, w, y, xg, ws, f);          \
        break;
    switch (3 + c) {
        PN2_SAF(3) PN2_SAF(4) PN2_SAF(5) PN2_SAF(6) PN2_SAF(7) PN2_SAF(8)
    }
#undef PN2_SAF
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
