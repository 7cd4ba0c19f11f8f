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
//   reserved[kPn2BnHead] | final[2][c] | slot[nslots][2][c], nslots <= kPn2BnSlots chosen per call
constexpr int kPn2BnHead = 8;    // doubles reserved in front (alignment of the sums to 64 bytes)
constexpr int kPn2BnSlots = 64;  // most copies of the per-channel accumulators the producers spread their atomics over
__host__ __device__ inline size_t pn2_bn_ws_doubles(int c, int nslots) { return kPn2BnHead + (size_t)(1 + nslots) * 2 * (size_t)c; }

// per-channel constants of the normalisation, identical float expressions in the forward and the backward kernels so
// that the ReLU mask recomputed in the backward is the forward's: z = fma(y, sc, sh), sc = gamma*invstd,
// sh = fma(-mean, sc, beta)
__device__ __forceinline__ void bn_scale_shift(float gamma, float beta, float mean, float invstd, float& sc, float& sh) {
    sc = gamma * invstd;
    sh = __builtin_fmaf(-mean, sc, beta);
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
