// pn2_train.hip -- the training step's non-GEMM tail on the device:
//   * weighted sparse softmax cross-entropy, reduction SUM_BY_NONZERO_WEIGHTS (reference model.py:152-161), forward + backward;
//   * dropout (util/tf_util.py:646-665 -> tf.nn.dropout: keep with probability keep_prob, scale by 1/keep_prob);
//   * Adam exactly as tf.train.AdamOptimizer applies it (reference train.py:381-388), ONE launch over the flat parameter buffer.
// Step-dependent scalars (learning rate with bias correction, dropout step counter, upstream loss gradient) are read from
// DEVICE memory, so a captured hipGraph of the whole step can be replayed while they change.
#include "pn2_common.h"

namespace {

constexpr int kCeMaxClasses = 64;

// one thread per point: logits row (C floats) -> lse, w * (lse - logit[label]); block partials -> two fp64 atomics
template <typename LabelT>
__global__ void __launch_bounds__(256)
ce_forward_kernel(int rows, int C, const float* __restrict__ logits, const LabelT* __restrict__ labels,
                  const float* __restrict__ w, float* __restrict__ lse_out, double* __restrict__ acc /* [sum, nonzero] */) {
    __shared__ double s_sum[4], s_cnt[4];
    const int r = blockIdx.x * 256 + threadIdx.x;
    double loss = 0.0, nz = 0.0;
    if (r < rows) {
        const float* __restrict__ z = logits + (size_t)r * C;
        float mx = z[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
        const float lse = mx + logf(se);
        lse_out[r] = lse;
        const int lab = (int)labels[r];
        const float wr = w[r];
        const float ce = lse - z[lab >= 0 && lab < C ? lab : 0];
        loss = (double)(wr * ce);
        nz = wr != 0.f ? 1.0 : 0.0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { loss += __shfl_xor(loss, o); nz += __shfl_xor(nz, o); }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = loss; s_cnt[threadIdx.x >> 6] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&acc[0], s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
        atomicAdd(&acc[1], s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
    }
}

__global__ void ce_finalize_kernel(const double* __restrict__ acc, float* __restrict__ loss) {
    const double nz = acc[1] > 0.0 ? acc[1] : 1.0;  // tf.losses: safe division, 0 when no weight is non-zero
    *loss = (float)(acc[0] / nz);
}

template <typename LabelT>
__global__ void __launch_bounds__(256)
ce_backward_kernel(long long total, int C, const float* __restrict__ logits, const LabelT* __restrict__ labels,
                   const float* __restrict__ w, const float* __restrict__ lse, const double* __restrict__ acc,
                   const float* __restrict__ gout, float* __restrict__ dlogits) {
    const double nz = acc[1] > 0.0 ? acc[1] : 1.0;
    const float scale = (gout ? *gout : 1.0f) / (float)nz;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / C;
        const int c = (int)(e - r * C);
        const float p = expf(logits[e] - lse[r]);
        dlogits[e] = scale * w[r] * (p - ((int)labels[r] == c ? 1.f : 0.f));
    }
}

// counter-based RNG: one 64-bit mix per element of (seed, step, index) -- stateless, replayable, graph-safe
__device__ __forceinline__ unsigned mix_u32(unsigned long long seed, unsigned long long step, unsigned long long i) {
    unsigned long long x = seed ^ (step * 0x9E3779B97F4A7C15ull) ^ (i * 0xD1B54A32D192ED03ull);
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return (unsigned)x;
}

// VEC = 4: 16-byte lanes (x, y) and one 32-bit word of four mask bytes per thread; the draw of element i does not depend on VEC
template <int VEC>
__global__ void __launch_bounds__(256)
dropout_kernel(long long n, const float* __restrict__ x, float keep, const long long* __restrict__ state /* [seed, step] */,
               float* __restrict__ y, unsigned char* __restrict__ mask) {
    const unsigned long long seed = (unsigned long long)state[0], step = (unsigned long long)state[1];
    const float inv = 1.0f / keep;
    const unsigned thr = keep >= 1.f ? 0xFFFFFFFFu : (unsigned)((double)keep * 4294967296.0);
    const long long nv = n / VEC;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nv; e += (long long)gridDim.x * 256) {
        if constexpr (VEC == 4) {
            const float4 v = reinterpret_cast<const float4*>(x)[e];
            const float in[4] = {v.x, v.y, v.z, v.w};
            float out[4];
            unsigned mw = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool k = mix_u32(seed, step, (unsigned long long)(e * 4 + u)) < thr;
                mw |= (k ? 1u : 0u) << (8 * u);
                out[u] = k ? in[u] * inv : 0.f;
            }
            reinterpret_cast<unsigned*>(mask)[e] = mw;
            reinterpret_cast<float4*>(y)[e] = make_float4(out[0], out[1], out[2], out[3]);
        } else {
            const bool k = mix_u32(seed, step, (unsigned long long)e) < thr;
            mask[e] = k ? 1 : 0;
            y[e] = k ? x[e] * inv : 0.f;
        }
    }
}

template <int VEC>
__global__ void __launch_bounds__(256)
dropout_grad_kernel(long long n, const float* __restrict__ dy, const unsigned char* __restrict__ mask, float keep,
                    float* __restrict__ dx) {
    const float inv = 1.0f / keep;
    const long long nv = n / VEC;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nv; e += (long long)gridDim.x * 256) {
        if constexpr (VEC == 4) {
            const float4 v = reinterpret_cast<const float4*>(dy)[e];
            const unsigned mw = reinterpret_cast<const unsigned*>(mask)[e];
            reinterpret_cast<float4*>(dx)[e] = make_float4((mw & 0xffu) ? v.x * inv : 0.f, (mw & 0xff00u) ? v.y * inv : 0.f,
                                                           (mw & 0xff0000u) ? v.z * inv : 0.f, (mw & 0xff000000u) ? v.w * inv : 0.f);
        } else {
            dx[e] = mask[e] ? dy[e] * inv : 0.f;
        }
    }
}

// Gradient of the ReLU of an un-normalised layer (tf_util.py:186-204 with bn=False: conv -> bias_add -> relu): TF's ReluGrad
// passes the upstream gradient where the OUTPUT is positive.
template <int VEC>
__global__ void __launch_bounds__(256)
relu_grad_kernel(long long n, const float* __restrict__ z, const float* __restrict__ dz, float* __restrict__ dx) {
    const long long nv = n / VEC;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nv; e += (long long)gridDim.x * 256) {
        if constexpr (VEC == 4) {
            const float4 a = reinterpret_cast<const float4*>(z)[e];
            const float4 g = reinterpret_cast<const float4*>(dz)[e];
            reinterpret_cast<float4*>(dx)[e] = make_float4(a.x > 0.f ? g.x : 0.f, a.y > 0.f ? g.y : 0.f,
                                                           a.z > 0.f ? g.z : 0.f, a.w > 0.f ? g.w : 0.f);
        } else {
            dx[e] = z[e] > 0.f ? dz[e] : 0.f;
        }
    }
}

// tf.train.AdamOptimizer (python/training/adam.py): m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2;
// p <- p - lr_t * m / (sqrt(v) + eps) with lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) supplied by the host in hyper[0].
__global__ void __launch_bounds__(256)
adam_kernel(long long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            const float* __restrict__ hyper /* [lr_t, beta1, beta2, eps, grad_scale] */) {
    const float lr_t = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], gs = hyper[4];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * gs;  // gs = 1/world after a summing all-reduce
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

inline int grid_1d(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}


// up to kMultiCopyMax device-to-device copies in ONE launch (the per-batch geometry tensors -- mixed int32 / float32 / byte
// buffers -- going into the static buffers a captured training step reads: torch._foreach_copy_ falls back to one memcpy
// per tensor for mixed dtypes, ~25 launches on the step's critical path)
constexpr int kMultiCopyMax = 48;
constexpr int kMultiFillMax = 4;
struct MultiCopyArgs {
    const unsigned char* src[kMultiCopyMax];
    unsigned char* dst[kMultiCopyMax];
    unsigned long long bytes[kMultiCopyMax];
    // a few 4- or 8-byte scalars written by the same launch (their values travel in the launch arguments): the per-step scalars of
    // a captured training step -- Adam's lr_t, the dropout step -- which were one fill launch each between two graph replays
    void* fill_dst[kMultiFillMax];
    unsigned long long fill_val[kMultiFillMax];
    int fill_bytes[kMultiFillMax];
    int nfill;
};

__global__ void __launch_bounds__(256)
multi_copy_kernel(MultiCopyArgs a) {
    const int t = blockIdx.y;
    if (t == 0 && blockIdx.x == 0 && (int)threadIdx.x < a.nfill) {
        const int f = threadIdx.x;
        if (a.fill_bytes[f] == 8) *static_cast<unsigned long long*>(a.fill_dst[f]) = a.fill_val[f];
        else *static_cast<unsigned*>(a.fill_dst[f]) = (unsigned)a.fill_val[f];
    }
    const unsigned char* __restrict__ s = a.src[t];
    unsigned char* __restrict__ d = a.dst[t];
    const unsigned long long nb = a.bytes[t];
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        const unsigned long long nv = nb >> 4;
        const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(s);
        uint4* __restrict__ d4 = reinterpret_cast<uint4*>(d);
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) d4[i] = s4[i];
        for (unsigned long long i = (nv << 4) + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) d[i] = s[i];
    } else {
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) d[i] = s[i];
    }
}

}  // namespace

// model.get_loss (reference model.py:152-161): logits (rows,C) f32, labels (rows) int32 (label64 = 0) or int64 (label64 = 1),
// weights (rows) f32 -> *loss = sum_r w_r * ce_r / max(1, #{w_r != 0}).  lse (rows) and acc (2 doubles) are kept for the
// backward; acc is zeroed here.  No host synchronisation.
extern "C" int pn2_weighted_ce_forward(int rows, int num_class, const float* logits, const void* labels, int label64,
                                       const float* weights, float* lse, double* acc, float* loss, void* stream) {
    if (rows <= 0 || num_class <= 0) return PN2_EINVAL;
    if (num_class > kCeMaxClasses) return PN2_EUNSUP;
    if (!logits || !labels || !weights || !lse || !acc || !loss) return PN2_ENULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(acc, 0, 2 * sizeof(double), st);
    if (e != hipSuccess) return (int)e;
    const int blocks = (rows + 255) / 256;
    if (label64) ce_forward_kernel<long long><<<blocks, 256, 0, st>>>(rows, num_class, logits, static_cast<const long long*>(labels), weights, lse, acc);
    else ce_forward_kernel<int><<<blocks, 256, 0, st>>>(rows, num_class, logits, static_cast<const int*>(labels), weights, lse, acc);
    ce_finalize_kernel<<<1, 1, 0, st>>>(acc, loss);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// d loss / d logits = gout * w_r / nz * (softmax_r - onehot_r); gout (device scalar) may be NULL (= 1).
extern "C" int pn2_weighted_ce_backward(int rows, int num_class, const float* logits, const void* labels, int label64,
                                        const float* weights, const float* lse, const double* acc, const float* gout,
                                        float* dlogits, void* stream) {
    if (rows <= 0 || num_class <= 0) return PN2_EINVAL;
    if (!logits || !labels || !weights || !lse || !acc || !dlogits) return PN2_ENULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long total = (long long)rows * num_class;
    if (label64) ce_backward_kernel<long long><<<grid_1d(total), 256, 0, st>>>(total, num_class, logits, static_cast<const long long*>(labels), weights, lse, acc, gout, dlogits);
    else ce_backward_kernel<int><<<grid_1d(total), 256, 0, st>>>(total, num_class, logits, static_cast<const int*>(labels), weights, lse, acc, gout, dlogits);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// tf.nn.dropout (util/tf_util.py:646-665): y = x / keep_prob where kept, 0 elsewhere; mask (n bytes) for the backward.
// state: device int64[2] = {seed, step}: the draw is a pure function of (seed, step, element index).
extern "C" int pn2_dropout(long long n, const float* x, float keep_prob, const long long* state, float* y,
                           unsigned char* mask, void* stream) {
    if (n <= 0 || !(keep_prob > 0.f) || keep_prob > 1.f) return PN2_EINVAL;
    if (!x || !state || !y || !mask) return PN2_ENULL;
    if (n % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)mask) % 16) == 0)
        dropout_kernel<4><<<grid_1d(n / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(n, x, keep_prob, state, y, mask);
    else
        dropout_kernel<1><<<grid_1d(n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, x, keep_prob, state, y, mask);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_dropout_grad(long long n, const float* dy, const unsigned char* mask, float keep_prob, float* dx,
                                void* stream) {
    if (n <= 0 || !(keep_prob > 0.f) || keep_prob > 1.f) return PN2_EINVAL;
    if (!dy || !mask || !dx) return PN2_ENULL;
    if (n % 4 == 0 && (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)mask) % 16) == 0)
        dropout_grad_kernel<4><<<grid_1d(n / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(n, dy, mask, keep_prob, dx);
    else
        dropout_grad_kernel<1><<<grid_1d(n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, dy, mask, keep_prob, dx);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// dx = dz where z > 0, else 0 (n elements; dx may alias dz): backward of the bias + ReLU epilogue of pn2_linear on the training
// path of layers WITHOUT batch norm (util/tf_util.py _TrainDenseRelu; reference tf_util.py:186-204 with bn=False).
extern "C" int pn2_relu_grad(long long n, const float* z, const float* dz, float* dx, void* stream) {
    if (n <= 0) return PN2_EINVAL;
    if (!z || !dz || !dx) return PN2_ENULL;
    if (n % 4 == 0 && (((uintptr_t)z | (uintptr_t)dz | (uintptr_t)dx) % 16) == 0)
        relu_grad_kernel<4><<<grid_1d(n / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(n, z, dz, dx);
    else
        relu_grad_kernel<1><<<grid_1d(n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, z, dz, dx);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// One Adam step over flat fp32 buffers of n elements (parameters, gradients, first / second moments).
// hyper: device float[5] = {lr_t, beta1, beta2, epsilon, grad_scale}, lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t).
extern "C" int pn2_adam_step(long long n, float* params, const float* grads, float* m, float* v, const float* hyper,
                             void* stream) {
    if (n <= 0) return PN2_EINVAL;
    if (!params || !grads || !m || !v || !hyper) return PN2_ENULL;
    adam_kernel<<<grid_1d(n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, params, grads, m, v, hyper);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// n (<= 48) independent device-to-device copies in one launch: dst[i][0 .. bytes[i]) = src[i][0 .. bytes[i]).  srcs / dsts /
// bytes are HOST arrays (read at call time); the regions must not overlap.  Plumbing of the training step (geometry
// tensors of a batch into the static buffers of the captured graph).
static int multi_copy_impl(int n, const void* const* srcs, void* const* dsts, const unsigned long long* bytes, int nfill,
                           void* const* fill_dsts, const unsigned long long* fill_vals, const int* fill_bytes, void* stream) {
    if (n <= 0 || nfill < 0) return PN2_EINVAL;
    if (n > kMultiCopyMax || nfill > kMultiFillMax) return PN2_ERANGE;
    if (!srcs || !dsts || !bytes || (nfill > 0 && (!fill_dsts || !fill_vals || !fill_bytes))) return PN2_ENULL;
    MultiCopyArgs a = {};
    a.nfill = nfill;
    for (int i = 0; i < nfill; ++i) {
        if (!fill_dsts[i]) return PN2_ENULL;
        if ((fill_bytes[i] != 4 && fill_bytes[i] != 8) || ((uintptr_t)fill_dsts[i] % (uintptr_t)fill_bytes[i]) != 0) return PN2_EINVAL;
        a.fill_dst[i] = fill_dsts[i];
        a.fill_val[i] = fill_vals[i];
        a.fill_bytes[i] = fill_bytes[i];
    }
    unsigned long long mx = 0;
    for (int i = 0; i < n; ++i) {
        if (!srcs[i] || !dsts[i]) return PN2_ENULL;
        a.src[i] = static_cast<const unsigned char*>(srcs[i]);
        a.dst[i] = static_cast<unsigned char*>(dsts[i]);
        a.bytes[i] = bytes[i];
        if (bytes[i] > mx) mx = bytes[i];
    }
    unsigned long long gx = (mx / 16 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;  // 64 x n blocks: the largest tensors (scatter plans, ~5 MB) take a few grid strides
    multi_copy_kernel<<<dim3((unsigned)gx, (unsigned)n), 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_multi_copy(int n, const void* const* srcs, void* const* dsts, const unsigned long long* bytes, void* stream) {
    return multi_copy_impl(n, srcs, dsts, bytes, 0, nullptr, nullptr, nullptr, stream);
}

// pn2_multi_copy that also writes nfill (<= 4) scalars of 4 or 8 bytes: *fill_dsts[i] = the low fill_bytes[i] bytes of
// fill_vals[i] (HOST arrays, the values travel in the launch arguments -- no host buffer a run-ahead host could rewrite before
// an asynchronous copy has read it).  The training step's Adam rate and dropout step ride with its input copy: ONE launch
// between two replays of the captured step (train.py:381-388 is the reference's per-step feed of those scalars).
extern "C" int pn2_multi_copy_fill(int n, const void* const* srcs, void* const* dsts, const unsigned long long* bytes, int nfill,
                                   void* const* fill_dsts, const unsigned long long* fill_vals, const int* fill_bytes,
                                   void* stream) {
    return multi_copy_impl(n, srcs, dsts, bytes, nfill, fill_dsts, fill_vals, fill_bytes, stream);
}
