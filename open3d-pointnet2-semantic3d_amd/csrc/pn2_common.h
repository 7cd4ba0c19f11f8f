// pn2_common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels.
// The library is compiled with -ffp-contract=off: every fused multiply-add in
// these kernels is an explicit __builtin_fmaf / MFMA, never a compiler choice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pn2_abi.h"

#define PN2_WAVE 64

// Kernel-selection knobs used by the A/B scripts under tools/.  In the shipped library they are compile-time constants:
// the library keeps NO mutable global state (include/pn2_abi.h).  A tuning build (`build.py --tuning`,
// -DPN2_TUNING_HOOKS) turns them into process-global variables behind pn2_debug_set(what, value) -- experiments only.
#ifdef PN2_TUNING_HOOKS
#define PN2_TUNABLE(type, name, value) type name = value;
#else
#define PN2_TUNABLE(type, name, value) static constexpr type name = value;
#endif

#define PN2_RETURN_IF_LAUNCH_FAILED()                 \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

// Training-mode batch-norm workspace (pn2_bn.hip; pn2_linear_bn_stats in pn2_linear.hip writes into it), in doubles:
//   head[kPn2BnHead] | final[2][c] | slot[nslots][2][c], nslots <= kPn2BnSlots chosen per call
// head (zeroed with the rest of the workspace) holds the ticket counters of pn2_bn_finish: 64 first-level + 1 second-level.
constexpr int kPn2BnHead = 48;   // doubles in front: 384 bytes, keeps the sums 64-byte aligned
constexpr int kPn2BnSlots = 64;  // most copies of the per-channel accumulators the producers spread their atomics over
constexpr int kPn2BnTickets = 64;
#ifndef PN2_STREAM_MIN_ROWS
#define PN2_STREAM_MIN_ROWS 65536   // fewest rows the wave-per-tile streaming kernels of the training GEMMs take (pn2_fwd_narrow.h, pn2_dgrad_wide.h, the
                                    // eight-wave weight gradient); measured: 32768 -> step 3.19 -> 3.21 ms, 16384 -> 3.24 (too few tiles per wave)
#endif
#ifndef PN2_FOLD_DEPTH
#define PN2_FOLD_DEPTH 64
#endif
constexpr int kPn2FoldDepth = PN2_FOLD_DEPTH;  // slot copies a folding thread keeps in flight (a device-scope load is a ~1 us round trip)
__host__ __device__ inline size_t pn2_bn_ws_doubles(int c, int nslots) { return kPn2BnHead + (size_t)(1 + nslots) * 2 * (size_t)c; }

// per-channel constants of the normalisation, identical float expressions in the forward and the backward kernels so
// that the ReLU mask recomputed in the backward is the forward's: z = fma(y, sc, sh), sc = gamma*invstd,
// sh = fma(-mean, sc, beta)
__device__ __forceinline__ void bn_scale_shift(float gamma, float beta, float mean, float invstd, float& sc, float& sh) {
    sc = gamma * invstd;
    sh = __builtin_fmaf(-mean, sc, beta);
}

// ---- "the last workgroup finishes": what follows a batch-norm reduction, inside the kernel that produced the sums ------------
// Every producer of per-channel sums (the GEMM epilogues of pn2_linear.hip, the reduction kernels of pn2_bn.hip) used to be
// followed by a one-block launch that folds the slot copies and derives per-channel constants: 45 launches of ~5 us per training
// step, each on the critical path.  Instead every workgroup takes a ticket once its atomics have been performed, and the one
// that draws the last ticket does that work.  Same-address atomics retire one after another (~0.1 us each), so the ticket is
// two-level: workgroup L counts on counter L % 64, the last arrival of each counter counts on the second level.
//   * a workgroup's sums are device-scope atomics (performed at the coherence point, not in its XCD's L2); `s_waitcnt vmcnt(0)`
//     in every wave + the workgroup barrier make them complete before the ticket is drawn;
//   * the finishing workgroup reads the slot copies with device-scope (sc1) loads, which bypass its XCD's non-coherent L2;
//   * what it writes (folded sums, constants) is read by LATER kernels only.
// kind: 0 nothing, 1 fold only, 2 fold + forward constants (bn_constants_kernel), 3 fold + gradient constants
// (bn_grad_constants_kernel).
struct Pn2BnFinish {
    int kind;
    int c, nslots;
    long long rows;
    double* ws;
    const float* gamma;
    const float* beta;
    const float* bias;          // kind 2 (may be null)
    const float* mean_in;       // kind 3: saved moments of the forward
    const float* invstd_in;
    float eps, decay;           // kind 2
    float* running_mean;        // kind 2 (may be null, both or none)
    float* running_var;
    float* save_mean;           // kind 2
    float* save_invstd;
    float* scale;               // kind 2 (may be null: the consumer derives them from the folded sums)
    float* shift;
    float* coef;                // kind 3: (6, c)
    float* dgamma;
    float* dbeta;
};

__device__ __forceinline__ double pn2_load_device_scope(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}

// Call from EVERY thread of EVERY workgroup of the producing kernel, after the workgroup's last atomic on f.ws (uniform control
// flow: contains workgroup barriers).  nwg = workgroups of the launch, wg = this workgroup's linear index.
__device__ __forceinline__ void pn2_bn_finish(const Pn2BnFinish& f, unsigned nwg, unsigned wg) {
    if (f.kind == 0) return;
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* tick = reinterpret_cast<unsigned*>(f.ws);
        const unsigned g = wg % (unsigned)kPn2BnTickets;
        const unsigned ng = nwg < (unsigned)kPn2BnTickets ? nwg : (unsigned)kPn2BnTickets;  // first-level counters in use
        const unsigned expect = nwg / kPn2BnTickets + (g < nwg % kPn2BnTickets ? 1u : 0u);  // workgroups counting on counter g
        int last = 0;
        if (__hip_atomic_fetch_add(tick + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expect - 1u)
            last = __hip_atomic_fetch_add(tick + kPn2BnTickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1u;
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const int c = f.c;
    // fold: one thread per COLUMN of the 2c sums, 16 copies in flight per thread (a device-scope load is a ~1 us round trip: the
    // chain of batches is this tail's duration), added in slot order like bn_fold_kernel; the two sums of a channel then meet in LDS
    for (int col = threadIdx.x; col < 2 * c; col += blockDim.x) {
        double t = 0.0;
        const double* sl = f.ws + kPn2BnHead + (size_t)2 * c + col;
        int k = 0;
        for (; k + kPn2FoldDepth <= f.nslots; k += kPn2FoldDepth) {
            double v[kPn2FoldDepth];
#pragma unroll
            for (int u = 0; u < kPn2FoldDepth; ++u) v[u] = pn2_load_device_scope(sl + (size_t)2 * c * (k + u));
#pragma unroll
            for (int u = 0; u < kPn2FoldDepth; ++u) t += v[u];
        }
        for (; k < f.nslots; ++k) t += pn2_load_device_scope(sl + (size_t)2 * c * k);
        f.ws[kPn2BnHead + col] = t;
    }
    if (f.kind == 1) return;
    __threadfence_block();  // the folded sums were stored by other threads of THIS workgroup: visible after the barrier
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        const double s1 = __hip_atomic_load(f.ws + kPn2BnHead + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const double s2 = __hip_atomic_load(f.ws + kPn2BnHead + c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const double inv_n = 1.0 / (double)f.rows;
        if (f.kind == 2) {  // bn_constants_kernel
            const double mean_d = s1 * inv_n;
            double var_d = s2 * inv_n - mean_d * mean_d;
            var_d = var_d > 0.0 ? var_d : 0.0;
            const float mean = (float)mean_d;
            const float invstd = (float)(1.0 / __builtin_sqrt(var_d + (double)f.eps));
            if (f.scale) {
                float sc, sh;
                bn_scale_shift(f.gamma[ch], f.beta[ch], mean, invstd, sc, sh);
                f.scale[ch] = sc;
                f.shift[ch] = sh;
            }
            f.save_mean[ch] = mean;
            f.save_invstd[ch] = invstd;
            if (f.running_mean) {
                const double m_out = mean_d + (f.bias ? (double)f.bias[ch] : 0.0);
                const double var_unb = f.rows > 1 ? var_d * ((double)f.rows / (double)(f.rows - 1)) : var_d;
                f.running_mean[ch] = (float)((double)f.decay * f.running_mean[ch] + (1.0 - (double)f.decay) * m_out);
                f.running_var[ch] = (float)((double)f.decay * f.running_var[ch] + (1.0 - (double)f.decay) * var_unb);
            }
        } else if (f.kind == 3) {  // bn_grad_constants_kernel
            const float mu = f.mean_in[ch], is = f.invstd_in[ch];
            float sc, sh;
            bn_scale_shift(f.gamma[ch], f.beta[ch], mu, is, sc, sh);
            f.coef[ch] = sc;
            f.coef[(size_t)c + ch] = sh;
            f.coef[(size_t)2 * c + ch] = mu;
            f.coef[(size_t)3 * c + ch] = is;
            f.coef[(size_t)4 * c + ch] = (float)(s1 * inv_n);
            f.coef[(size_t)5 * c + ch] = (float)(s2 * inv_n);
            f.dbeta[ch] = (float)s1;
            f.dgamma[ch] = (float)s2;
        }
    }
}

// Operand transform of the training GEMMs that read the PRE-normalisation output of the layer below: the batch norm (+ReLU)
// of that layer, a = relu?(fma(x, scale[k], shift[k])) per input channel k (pn2_bn_relu_forward_deferred publishes scale / shift).
struct Pn2LoadTransform {
    const float* scale;
    const float* shift;
    int relu;
};

// Epilogue of a data-gradient GEMM whose output dx IS the gradient dz reaching the batch norm (+ReLU) of the layer below
// (pn2_linear_dgrad_bn_grad_stats): the first backward reduction of that batch norm -- sum g and sum g * xhat per channel,
// g = dz * [relu mask], xhat = (y - mean) * invstd -- is taken from the accumulator tiles, so bn_grad_reduce_kernel's pass
// over (dz, y) is not run.  ws = null: plain data gradient.
struct Pn2BnGradEpilogue {
    const float* y;        // (rows, c) pre-normalisation output of the layer below (c = width of dx)
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* invstd;
    double* ws;            // its ZEROED batch-norm workspace (all kPn2BnSlots slot copies are used)
    int relu;
};

// Operand transform of the two gradient GEMMs of a dense layer that consume the gradient dy LEAVING its batch norm (+ReLU
// [+ max over groups of 32 rows]): dy is formed while the operand is staged and never written (util/tf_util.py:555-581 via
// tf.gradients; pn2_linear_dgrad_gx / pn2_linear_wgrad_gx).  The float expressions are bn_grad_apply_kernel's, so the GEMMs see
// the bits the materialised form would have handed them.  coef (6, c) = sc, sh, mean, invstd, k1, k2 per channel
// (pn2_bn_grad_constants: sc = gamma * invstd, sh = fma(-mean, sc, beta), k1 = mean over rows of g, k2 = of g * xhat).
constexpr int kPn2GxMaxC = 512;  // widest batch norm the on-load form takes (its constants are staged in LDS)
struct Pn2GradOnLoad {
    const float* y;     // (rows, c) pre-normalisation output of this layer
    const float* dz;    // (rows, c) gradient reaching the activation; pool == 32: (rows / 32, c) gradient of the pooled maxima
    const float* coef;  // (6, c)
    const float* zmax;  // pool == 32: (rows / 32, c) the pooled maxima
    const float* ties;  // pool == 32: (rows / 32, c) rows attaining them
    int relu;
    int pool;           // 0 or 32
};

// one element of that gradient; g_in = the gradient reaching the activation of this element (before the ReLU mask)
__device__ __forceinline__ float pn2_bn_grad_element(float y, float g_in, float sc, float sh, float mu, float is, float k1,
                                                     float k2, int relu) {
    const float lin = __builtin_fmaf(y, sc, sh);
    const bool on = !relu || lin > 0.f;
    const float gk = on ? g_in : 0.f;
    const float xh = (y - mu) * is;
    return sc * __builtin_fmaf(-xh, k2, gk - k1);
}
// the same behind the fused max over 32 rows: the pooled gradient d is shared equally by the n rows attaining the maximum m
__device__ __forceinline__ float pn2_bn_grad_element_pooled(float y, float d, float m, float n, float sc, float sh, float mu,
                                                            float is, float k1, float k2, int relu) {
    const float lin = __builtin_fmaf(y, sc, sh);
    const bool on = !relu || lin > 0.f;
    const float t = on ? lin : 0.f;
    const float g = t == m ? d / n : 0.f;
    const float gk = on ? g : 0.f;
    const float xh = (y - mu) * is;
    return sc * __builtin_fmaf(-xh, k2, gk - k1);
}

// Largest float T with  max(sqrtf(T), 1e-20f) < radius  (sqrtf correctly rounded, hence monotone): for every s >= 0,
// (s <= T) <=> the reference's ball-query predicate tf_grouping.cu:28-31.  Returns -1 when nothing can match.  Host side.
inline float pn2_ball_threshold(float radius) {
    if (!(radius > 1e-20f)) return -1.0f;
    auto pred = [radius](float s) { return sqrtf(s) < radius; };
    float t = radius * radius;
    if (!(t <= 3.402823466e38f)) t = 3.402823466e38f;
    while (!pred(t)) t = nextafterf(t, -INFINITY);
    for (;;) {
        const float u = nextafterf(t, INFINITY);
        if (!(u <= 3.402823466e38f) || !pred(u)) break;
        t = u;
    }
    return t;
}

// Squared distance exactly as the reference expression
//   (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)
// (tf_sampling.cu:149-150, tf_grouping.cu:28-30) under the three contraction
// hypotheses of pn2_abi.h.  dx/dy/dz are already-rounded differences.
template <int MODE>
__device__ __forceinline__ float pn2_sqdist(float dx, float dy, float dz) {
    if constexpr (MODE == PN2_ARITH_FMA) {
        return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
    } else if constexpr (MODE == PN2_ARITH_FMA_ALT) {
        return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    } else {
        return (dx * dx + dy * dy) + dz * dz;  // -ffp-contract=off keeps these separate
    }
}

// ---- wave64 DPP reductions (result broadcast through an SGPR) -------------
// row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143 (gfx9 DPP).
// old == src and bound_ctrl=false: lanes without a valid source keep their value.
#define PN2_DPP_U32(v, ctrl, rmask) \
    ((unsigned)__builtin_amdgcn_update_dpp((int)(v), (int)(v), (ctrl), (rmask), 0xF, false))

__device__ __forceinline__ unsigned pn2_wave_umax(unsigned v) {
    unsigned t;
    t = PN2_DPP_U32(v, 0x111, 0xF); v = v > t ? v : t;
    t = PN2_DPP_U32(v, 0x112, 0xF); v = v > t ? v : t;
    t = PN2_DPP_U32(v, 0x114, 0xF); v = v > t ? v : t;
    t = PN2_DPP_U32(v, 0x118, 0xF); v = v > t ? v : t;
    t = PN2_DPP_U32(v, 0x142, 0xA); v = v > t ? v : t;
    t = PN2_DPP_U32(v, 0x143, 0xC); v = v > t ? v : t;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// max of a 64-bit key over the 16 lanes of row 0; result (uniform) from lane 15.
#define PN2_U64MAX_STEP(v, ctrl)                                        \
    do {                                                                \
        unsigned lo__ = (unsigned)(v), hi__ = (unsigned)((v) >> 32);    \
        unsigned tlo__ = PN2_DPP_U32(lo__, (ctrl), 0xF);                \
        unsigned thi__ = PN2_DPP_U32(hi__, (ctrl), 0xF);                \
        unsigned long long t__ = ((unsigned long long)thi__ << 32) | tlo__; \
        (v) = t__ > (v) ? t__ : (v);                                    \
    } while (0)

__device__ __forceinline__ unsigned long long pn2_row0_u64max(unsigned long long v) {
    PN2_U64MAX_STEP(v, 0x111);
    PN2_U64MAX_STEP(v, 0x112);
    PN2_U64MAX_STEP(v, 0x114);
    PN2_U64MAX_STEP(v, 0x118);
    unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 15);
    unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 15);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ int pn2_lane_id() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
