// pn2_mfma_stats.h -- batch-norm sums taken from 32x32 MFMA accumulator tiles: shared by the GEMM epilogues of pn2_linear.hip and
// the fused backward kernel of pn2_bwd_fused.hip.  (v_mfma_f32_32x32x2_f32 result layout: lane l holds column j = l & 31 and the
// rows i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) of register r = 0..15.)
#pragma once
#include "pn2_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Column sums of one 32x32 accumulator tile for the batch norm that follows (pn2_linear_bn_stats), in fp64 from the first
// term on -- the same moments the two-pass path (bn_stats_kernel) forms, up to summation order: 48 fp64 operations per lane
// and tile, ~23 us over all layers of a training step against the 280 us of statistics passes they replace.  One atomic
// pair per column into slot copy `slot`.  Rows past `rows` hold exact zeros (their A operand was zeroed): they add nothing.
__device__ __forceinline__ void push_column_stats(const f32x16& acc, int half, int col, int cout, unsigned slot,
                                                  double* __restrict__ stats) {
    double d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double d = (double)acc[r];
        d1 += d;
        d2 = __builtin_fma(d, d, d2);
    }
    d1 += __shfl_xor(d1, 32);
    d2 += __shfl_xor(d2, 32);
    if (half == 0 && col < cout) {
        double* __restrict__ sl = stats + kPn2BnHead + (size_t)2 * cout * (1 + slot % (unsigned)kPn2BnSlots);
        atomicAdd(sl + col, d1);
        atomicAdd(sl + cout + col, d2);
    }
}

// The same for the data gradient that reaches a batch norm (+ReLU): column sums of g = dz * [mask] and g * xhat over one
// 32x32 tile of dz (see Pn2BnGradEpilogue); float expressions of bn_grad_reduce_kernel, fp64 accumulation.
__device__ __forceinline__ void push_column_grad_stats(const f32x16& acc, int half, int wrow0, int rows, int col, int c,
                                                       unsigned slot, const Pn2BnGradEpilogue& e) {
    double d1 = 0.0, d2 = 0.0;
    if (col < c) {
        const float mean = e.mean[col], invstd = e.invstd[col];
        float sc, sh;
        bn_scale_shift(e.gamma[col], e.beta[col], mean, invstd, sc, sh);
        float a[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            a[r] = e.y[(size_t)(row < rows ? row : rows - 1) * c + col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool on = row < rows && (!e.relu || __builtin_fmaf(a[r], sc, sh) > 0.f);
            const double gd = on ? (double)acc[r] : 0.0;
            const double xh = (double)((a[r] - mean) * invstd);
            d1 += gd;
            d2 = __builtin_fma(gd, xh, d2);
        }
    }
    d1 += __shfl_xor(d1, 32);
    d2 += __shfl_xor(d2, 32);
    if (half == 0 && col < c) {
        double* __restrict__ sl = e.ws + kPn2BnHead + (size_t)2 * c * (1 + slot % (unsigned)kPn2BnSlots);
        atomicAdd(sl + col, d1);
        atomicAdd(sl + c + col, d2);
    }
}

}  // namespace
