// pn2_dgrad_wide.h -- the data gradient of a 128 -> 128 (or 64 -> 128) dense + batch-norm layer of the training path over >= 65536 rows, with the
// upstream gradient formed on load: dx (rows, 128) = dy . W^T, dy = the gradient LEAVING this layer's batch norm (+ReLU) formed
// from (y, dz) and the six per-channel constants (Pn2GradOnLoad, GX = 1), plus the batch-norm gradient sums of the layer BELOW from
// the dx tiles (Pn2BnGradEpilogue) and their finish -- what pn2_linear_dgrad_fin computes on linear_kernel<2, 2, 2, ..., TB, GX = 1>
// (reference: tf.gradients through conv2d -> batch_norm -> relu, util/tf_util.py:181-204,555-581).
//
// The streaming form of pn2_fwd_narrow.h's fwd_wide_in_kernel: eight waves per workgroup (two per SIMD: one wave's transform,
// loads of y_below, stores and sums run under the other's MFMAs), a wave owns 32-row tiles, the operand tile goes through the wave's
// own LDS tile in K-slices of PN2_DGW_KS = 32 channels (loaded with coalesced 16-byte lanes one slice ahead), W^T sits in LDS in fragment
// order, and the two gradient sums per column stay in registers (fp64) across the wave's tiles: one pair of atomics per column
// and wave.  Same contraction order as linear_kernel, hence the same dx bits.
#pragma once
#include "pn2_common.h"
#include "pn2_mfma_stats.h"

#ifndef PN2_DGW_KS
#define PN2_DGW_KS 32   // channels per K-slice of the operand tile (64: the two operand streams of a slice spill)
#endif

namespace {

// CO = n_in / 32 (2, 4); GX = 1: dz (rows, 128); GX = 2: behind the max over groups of 32 rows -- a wave's tile IS one group, its
// pooled gradient / maximum / tie count rows (gx.dz / zmax / ties, (rows / 32, 128)) travel with the slice
template <int CO, int GX>
__global__ void __launch_bounds__(512, 1)
dgrad_wide_kernel(int rows, const float* __restrict__ w, float* __restrict__ dx, Pn2GradOnLoad gx, Pn2BnGradEpilogue gepi,
                  Pn2BnFinish fin) {
    constexpr int K = 128, N = 32 * CO, KS = PN2_DGW_KS, NS = K / KS, NW = 8;
    constexpr int AS = KS + 4;
    constexpr int NF = 32 * (KS / 4) / 64;  // float4 of a slice per lane (8)
    extern __shared__ __attribute__((aligned(16))) float dgw_lds[];
    float* __restrict__ Wf = dgw_lds;                           // (T, nt, lane, 4): (K / 8) * CO * 256 floats = 64 KB
    float* __restrict__ coef = dgw_lds + (K / 8) * CO * 256;    // (6, K)
    float* __restrict__ Aall = coef + 6 * K;                    // 8 waves x 32 x AS
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* __restrict__ As = Aall + wave * (32 * AS);
    // W (N, K) row-major as the forward holds it (n_in x n_out): B(k, n) = w[n * K + k] -> fragment order
    for (int e = tid; e < N * (K / 4); e += 512) {
        const int n = e / (K / 4), k = (e - n * (K / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)n * K + k);
        *reinterpret_cast<f32x4*>(Wf + (((((k >> 3) * CO + (n >> 5)) * 64) + ((k >> 2) & 1) * 32 + (n & 31)) << 2)) = v;
    }
    for (int e = tid; e < 6 * K; e += 512) coef[e] = gx.coef[e];
    const int k4 = lane % (KS / 4);  // this lane's float4 column inside a slice: the same for every load
    __syncthreads();

    const int ntiles = rows / 32;
    const int tstep = gridDim.x * NW;
    int tile = blockIdx.x * NW + wave;
    f32x4 py[NF], pg[GX == 1 ? NF : 3];  // ONE slice of (y, dz) ahead (GX = 2: the group's three pooled rows instead of dz)
    auto fetch = [&](int t, int s_) __attribute__((always_inline)) {
        const int tc = t < ntiles ? t : ntiles - 1;
        const size_t base = (size_t)tc * 32 * K + s_ * KS + k4 * 4;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const size_t o = base + (size_t)((lane + 64 * i) / (KS / 4)) * K;
            py[i] = *reinterpret_cast<const f32x4*>(gx.y + o);
            if constexpr (GX == 1) pg[i] = *reinterpret_cast<const f32x4*>(gx.dz + o);
        }
        if constexpr (GX == 2) {
            const size_t go = (size_t)tc * K + s_ * KS + k4 * 4;
            pg[0] = *reinterpret_cast<const f32x4*>(gx.dz + go);
            pg[1] = *reinterpret_cast<const f32x4*>(gx.zmax + go);
            pg[2] = *reinterpret_cast<const f32x4*>(gx.ties + go);
        }
    };
    // the layer below: this lane's column of each 32-column block (running sums in fp64, as push_column_grad_stats forms them)
    float b_mean[CO], b_is[CO], b_sc[CO], b_sh[CO];
    double s1[CO], s2[CO];
#pragma unroll
    for (int nt = 0; nt < CO; ++nt) {
        s1[nt] = s2[nt] = 0.0;
        b_mean[nt] = b_is[nt] = b_sc[nt] = b_sh[nt] = 0.f;
        if (gepi.ws) {
            const int col = nt * 32 + l31;
            b_mean[nt] = gepi.mean[col]; b_is[nt] = gepi.invstd[col];
            bn_scale_shift(gepi.gamma[col], gepi.beta[col], b_mean[nt], b_is[nt], b_sc[nt], b_sh[nt]);
        }
    }
    if (tile < ntiles) fetch(tile, 0);
    while (tile < ntiles) {
        f32x16 acc[CO];
#pragma unroll
        for (int nt = 0; nt < CO; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            f32x4 gc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) gc[j] = *reinterpret_cast<const f32x4*>(coef + j * K + s_ * KS + k4 * 4);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int r = (lane + 64 * i) / (KS / 4);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (GX == 1)
                        v[q] = pn2_bn_grad_element(py[i][q], pg[i][q], gc[0][q], gc[1][q], gc[2][q], gc[3][q], gc[4][q], gc[5][q], gx.relu);
                    else
                        v[q] = pn2_bn_grad_element_pooled(py[i][q], pg[0][q], pg[1][q], pg[2][q], gc[0][q], gc[1][q], gc[2][q], gc[3][q],
                                                          gc[4][q], gc[5][q], gx.relu);
                }
                *reinterpret_cast<f32x4*>(As + r * AS + k4 * 4) = v;
            }
            // the next slice: of this tile, or the first one of this wave's next tile (clamped past the end: never used)
            if (s_ + 1 < NS) fetch(tile, s_ + 1);
            else fetch(tile + tstep, 0);
            __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations execute in issue order)
            const float* __restrict__ as = As + l31 * AS + 4 * half;
#pragma unroll
            for (int T = 0; T < KS / 8; ++T) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(as + 8 * T);
                f32x4 b[CO];
#pragma unroll
                for (int nt = 0; nt < CO; ++nt)
                    b[nt] = *reinterpret_cast<const f32x4*>(Wf + (((s_ * (KS / 8) + T) * CO + nt) * 64 + lane) * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int nt = 0; nt < CO; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], b[nt][q], acc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();  // the next slice overwrites As
        }
        const int row0 = tile * 32;
#pragma unroll
        for (int nt = 0; nt < CO; ++nt) {
            const int col = nt * 32 + l31;
            float yb[16];
            if (gepi.ws) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yb[r] = gepi.y[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * half) * N + col];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                dx[(size_t)row * N + col] = acc[nt][r];
            }
            if (gepi.ws) {  // the float expressions of push_column_grad_stats / bn_grad_reduce_kernel, fp64 accumulation
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool on = !gepi.relu || __builtin_fmaf(yb[r], b_sc[nt], b_sh[nt]) > 0.f;
                    const double gd = on ? (double)acc[nt][r] : 0.0;
                    const double xh = (double)((yb[r] - b_mean[nt]) * b_is[nt]);
                    s1[nt] += gd;
                    s2[nt] = __builtin_fma(gd, xh, s2[nt]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // one column block at a time: its 16 loads of y_below are not hoisted over the others'
        }
        tile += tstep;
    }
    if (gepi.ws) {
        const unsigned slot = (blockIdx.x * NW + wave) % (unsigned)kPn2BnSlots;
        double* __restrict__ sl = gepi.ws + kPn2BnHead + (size_t)2 * N * (1 + slot);
#pragma unroll
        for (int nt = 0; nt < CO; ++nt) {
            const double d1 = s1[nt] + __shfl_xor(s1[nt], 32), d2 = s2[nt] + __shfl_xor(s2[nt], 32);
            if (half == 0) {
                atomicAdd(sl + nt * 32 + l31, d1);
                atomicAdd(sl + N + nt * 32 + l31, d2);
            }
        }
    }
    pn2_bn_finish(fin, gridDim.x, blockIdx.x);
}

// n_out = 128, n_in in {64, 128}, rows % 32 == 0, enough rows, dy formed on load (plain or behind the max over 32 rows), 16-byte
// aligned operands
inline bool dgrad_wide_fits(int rows, int n_in, int n_out, const Pn2GradOnLoad* gx, const void* w_) {
    if (!gx || n_out != 128 || (n_in != 128 && n_in != 64)) return false;
    if (gx->pool != 0 && gx->pool != 32) return false;
    if (rows % 32 != 0 || rows < PN2_STREAM_MIN_ROWS) return false;
    uintptr_t a = (uintptr_t)gx->y | (uintptr_t)gx->dz | (uintptr_t)gx->coef | (uintptr_t)w_;
    if (gx->pool) a |= (uintptr_t)gx->zmax | (uintptr_t)gx->ties;
    return (a % 16) == 0;
}

template <int CO, int GX>
int launch_dgrad_wide_one(int rows, const float* w, float* dx, const Pn2GradOnLoad& gx, const Pn2BnGradEpilogue& gepi,
                          const Pn2BnFinish* fin, hipStream_t st) {
    const int ntiles = rows / 32;
    int blocks = (ntiles + 7) / 8;
    if (blocks > 256) blocks = 256;  // one workgroup of eight waves per CU
    constexpr size_t lds = sizeof(float) * ((size_t)16 * CO * 256 + 6 * 128 + 8 * 32 * (PN2_DGW_KS + 4));
    static bool attr_set = false;  // per instantiation; benign race (idempotent call)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_wide_kernel<CO, GX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dgrad_wide_kernel<CO, GX><<<blocks, 512, lds, st>>>(rows, w, dx, gx, gepi, fin ? *fin : Pn2BnFinish{});
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

inline int launch_dgrad_wide(int rows, int n_in, const float* w, float* dx, const Pn2GradOnLoad& gx, const Pn2BnGradEpilogue& gepi,
                             const Pn2BnFinish* fin, hipStream_t st) {
    if (n_in == 128) return gx.pool ? launch_dgrad_wide_one<4, 2>(rows, w, dx, gx, gepi, fin, st)
                                    : launch_dgrad_wide_one<4, 1>(rows, w, dx, gx, gepi, fin, st);
    return gx.pool ? launch_dgrad_wide_one<2, 2>(rows, w, dx, gx, gepi, fin, st)
                   : launch_dgrad_wide_one<2, 1>(rows, w, dx, gx, gepi, fin, st);
}

}  // namespace
