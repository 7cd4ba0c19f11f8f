// pn2_fwd_narrow.h -- the forward of a NARROW dense layer of the training path (32 / 64 input channels, 32 / 64 / 128 output
// channels) over 10^5 .. 10^6 rows: y = a . W with a = x or relu?(fma(x, scale[k], shift[k])) (the batch norm of the layer below
// applied on load), the column sums of y and y^2 for the batch norm behind it, and the last-workgroup finish -- what
// pn2_linear_bn_stats(_xf,_fin) compute (reference: tf.nn.conv2d 1x1 -> batch_norm_template, util/tf_util.py:181-204,555-581).
//
// These layers (SA1: 524288 x 32 -> 32 -> 64, SA2: 131072 x 64 -> 64 -> 128) are HBM streams: 16 .. 32 MFMAs per 32-row tile
// against 4 .. 16 KB of input and 4 .. 16 KB of output.  linear_kernel gives every 128-row tile a workgroup whose ONE k-tile leaves
// nothing to prefetch behind -- load, barrier, 16 MFMAs, epilogue, each phase alone -- and reaches 2.3 .. 2.8 TB/s on them.  Here, as
// in the backward's bwd_narrow_kernel (pn2_bwd_fused.hip):
//   * a WAVE owns 32-row tiles (tile t, t + waves, ...), no workgroup barrier after the weight panel is staged;
//   * a tile of x is 32 x cin contiguous floats: loaded with fully coalesced 16-byte lanes TWO tiles ahead into registers, the
//     operand transform applied on the way into the wave's own LDS tile (row stride cin + 4: the MFMA-layout 16-byte reads are
//     conflict-free), from where lane (row = l & 31, half = l >> 5) reads k = 8T + 4*half + {0..3} -- linear_kernel's visiting
//     order, hence its bits;
//   * the weight panel sits in LDS in fragment order (one conflict-free ds_read_b128 per lane = the B operand of four MFMAs);
//   * the column sums stay in registers (fp64) across all tiles of the wave: one pair of atomics per column and WAVE instead of one
//     per column and tile (4096 waves instead of 16384 tiles at SA1).
#pragma once
#include <type_traits>

#include "pn2_common.h"
#include "pn2_mfma_stats.h"

namespace {

// CI = cin / 32 (1, 2), CO = cout / 32 (1, 2, 4)
template <int CI, int CO, bool XF>
__global__ void __launch_bounds__(256, 2)
fwd_narrow_kernel(int rows, const float* __restrict__ x, Pn2LoadTransform xf, const float* __restrict__ w, float* __restrict__ y,
                  double* __restrict__ stats, Pn2BnFinish fin) {
    constexpr int CIN = 32 * CI, COUT = 32 * CO;
    constexpr int AS = CIN + 4;          // row stride of a wave's operand tile
    constexpr int K4 = CIN / 4;          // float4 columns of an x row
    constexpr int NF = 32 * K4 / 64;     // float4 of a tile per lane (4 / 8)
    constexpr int K8 = CIN / 8;          // fragments along the contraction
    extern __shared__ __attribute__((aligned(16))) float fwn_lds[];   // (64 x 128: 66 KB, beyond the static limit)
    float* __restrict__ Wf = fwn_lds;                    // (T, nt, lane, 4): K8 * CO * 256 floats
    float* __restrict__ Aall = fwn_lds + K8 * CO * 256;  // 4 waves x 32 x AS
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* __restrict__ As = Aall + wave * (32 * AS);

    for (int e = tid; e < CIN * (COUT / 4); e += 256) {  // w (CIN, COUT) row-major -> fragment order
        const int k = e / (COUT / 4), n4 = e - k * (COUT / 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)k * COUT + n4 * 4);
        const int base = ((k >> 3) * CO * 64 + ((k >> 2) & 1) * 32) * 4 + (k & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n4 * 4 + j;
            Wf[base + ((n >> 5) * 64 + (n & 31)) * 4] = v[j];
        }
    }
    // this lane's four input channels are the same for every float4 it loads (64 % K4 == 0): their constants stay in registers
    const int k4 = lane % K4;
    f32x4 xsc = {1.f, 1.f, 1.f, 1.f}, xsh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (XF) {
        xsc = *reinterpret_cast<const f32x4*>(xf.scale + k4 * 4);
        xsh = *reinterpret_cast<const f32x4*>(xf.shift + k4 * 4);
    }
    __syncthreads();  // Wf staged

    const int ntiles = rows / 32;
    const int tstep = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    f32x4 pa[2][NF];  // two tiles in flight, addressed statically (the tile loop is unrolled by two)
    auto fetch = [&](auto uc, int t) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        const int tc = t < ntiles ? t : ntiles - 1;
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(x + (size_t)tc * 32 * CIN);
#pragma unroll
        for (int i = 0; i < NF; ++i) pa[u][i] = src[lane + 64 * i];
    };
    double s1[CO], s2[CO];
#pragma unroll
    for (int nt = 0; nt < CO; ++nt) s1[nt] = s2[nt] = 0.0;
    auto process = [&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        // the operand tile, transformed, into the wave's LDS tile
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int r = (lane + 64 * i) / K4;
            f32x4 v = pa[u][i];
            if constexpr (XF) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = __builtin_fmaf(v[q], xsc[q], xsh[q]);
                    v[q] = xf.relu ? fmaxf(t, 0.f) : t;
                }
            }
            *reinterpret_cast<f32x4*>(As + r * AS + k4 * 4) = v;
        }
        fetch(std::integral_constant<int, u>{}, tile + 2 * tstep);  // (clamped past the end: never used)
        __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations execute in issue order: the reads below see these writes)
        f32x16 acc[CO];
#pragma unroll
        for (int nt = 0; nt < CO; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const float* __restrict__ as = As + l31 * AS + 4 * half;
#pragma unroll
        for (int T = 0; T < K8; ++T) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(as + 8 * T);
            f32x4 b[CO];
#pragma unroll
            for (int nt = 0; nt < CO; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(Wf + ((T * CO + nt) * 64 + lane) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < CO; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], b[nt][q], acc[nt], 0, 0, 0);
        }
        const int row0 = tile * 32;
#pragma unroll
        for (int nt = 0; nt < CO; ++nt) {
            const int col = nt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                y[(size_t)row * COUT + col] = acc[nt][r];
            }
            if (stats) {  // fp64 from the first term on, as push_column_stats; the wave's running sums stay in registers
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double d = (double)acc[nt][r];
                    s1[nt] += d;
                    s2[nt] = __builtin_fma(d, d, s2[nt]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the next tile overwrites As: every read above has been issued
        tile += tstep;
    };
    if (tile < ntiles) {
        fetch(std::integral_constant<int, 0>{}, tile);
        fetch(std::integral_constant<int, 1>{}, tile + tstep);
    }
    while (tile < ntiles) {
        process(std::integral_constant<int, 0>{});
        if (tile < ntiles) process(std::integral_constant<int, 1>{});
    }
    if (stats) {
        const unsigned slot = (blockIdx.x * 4 + wave) % (unsigned)kPn2BnSlots;
        double* __restrict__ sl = stats + kPn2BnHead + (size_t)2 * COUT * (1 + slot);
#pragma unroll
        for (int nt = 0; nt < CO; ++nt) {
            const double d1 = s1[nt] + __shfl_xor(s1[nt], 32), d2 = s2[nt] + __shfl_xor(s2[nt], 32);
            if (half == 0) {
                atomicAdd(sl + nt * 32 + l31, d1);
                atomicAdd(sl + COUT + nt * 32 + l31, d2);
            }
        }
    }
    pn2_bn_finish(fin, gridDim.x, blockIdx.x);
}

// The same for 128 INPUT channels (FP4 / the head: 131072 x 128 -> 128): eight waves per workgroup (two per SIMD, so that one wave's
// transform / stores / sums run under the other's MFMAs), the operand tile staged in two K-SLICES of 64 channels so that eight wave
// tiles (8.7 KB each) fit beside the 64 KB weight panel.  Loads: sub-tile s of a tile = its columns [64 s, 64 s + 64): four rows x 256
// contiguous bytes per load instruction.
template <int CO, bool XF>
__global__ void __launch_bounds__(512, 1)
fwd_wide_in_kernel(int rows, const float* __restrict__ x, Pn2LoadTransform xf, const float* __restrict__ w, float* __restrict__ y,
                   double* __restrict__ stats, Pn2BnFinish fin) {
    constexpr int CIN = 128, COUT = 32 * CO, KS = 64, NS = CIN / KS, NW = 8;
    constexpr int AS = KS + 4;           // row stride of a wave's operand slice
    constexpr int NF = 32 * (KS / 4) / 64;   // float4 of a slice per lane (8)
    constexpr int K8 = CIN / 8;
    extern __shared__ __attribute__((aligned(16))) float fwn_lds[];
    float* __restrict__ Wf = fwn_lds;                    // (T, nt, lane, 4): K8 * CO * 256 floats
    float* __restrict__ Aall = fwn_lds + K8 * CO * 256;  // 8 waves x 32 x AS
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* __restrict__ As = Aall + wave * (32 * AS);
    for (int e = tid; e < CIN * (COUT / 4); e += 512) {  // w (CIN, COUT) row-major -> fragment order
        const int k = e / (COUT / 4), n4 = e - k * (COUT / 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)k * COUT + n4 * 4);
        const int base = ((k >> 3) * CO * 64 + ((k >> 2) & 1) * 32) * 4 + (k & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n4 * 4 + j;
            Wf[base + ((n >> 5) * 64 + (n & 31)) * 4] = v[j];
        }
    }
    const int k4 = lane % (KS / 4);      // this lane's float4 column inside a slice: the same for every load
    f32x4 xsc[NS], xsh[NS];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        xsc[s_] = f32x4{1.f, 1.f, 1.f, 1.f}; xsh[s_] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (XF) {
            xsc[s_] = *reinterpret_cast<const f32x4*>(xf.scale + s_ * KS + k4 * 4);
            xsh[s_] = *reinterpret_cast<const f32x4*>(xf.shift + s_ * KS + k4 * 4);
        }
    }
    __syncthreads();  // Wf staged
    const int ntiles = rows / 32;
    const int tstep = gridDim.x * NW;
    int tile = blockIdx.x * NW + wave;
    f32x4 pa[NS][NF];  // ONE tile ahead (the registers of a tile are free once its last slice sits in LDS; the SIMD partner
                       // wave covers the rest of the round trip)
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const int tc = t < ntiles ? t : ntiles - 1;
        const float* __restrict__ src = x + (size_t)tc * 32 * CIN;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int r = (lane + 64 * i) / (KS / 4);
                pa[s_][i] = *reinterpret_cast<const f32x4*>(src + (size_t)r * CIN + s_ * KS + k4 * 4);
            }
    };
    double s1[CO], s2[CO];
#pragma unroll
    for (int nt = 0; nt < CO; ++nt) s1[nt] = s2[nt] = 0.0;
    auto process = [&]() __attribute__((always_inline)) {
        f32x16 acc[CO];
#pragma unroll
        for (int nt = 0; nt < CO; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int r = (lane + 64 * i) / (KS / 4);
                f32x4 v = pa[s_][i];
                if constexpr (XF) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = __builtin_fmaf(v[q], xsc[s_][q], xsh[s_][q]);
                        v[q] = xf.relu ? fmaxf(t, 0.f) : t;
                    }
                }
                *reinterpret_cast<f32x4*>(As + r * AS + k4 * 4) = v;
            }
            if (s_ == NS - 1) fetch(tile + tstep);  // (clamped past the end: never used)
            __builtin_amdgcn_wave_barrier();
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                y[(size_t)row * COUT + col] = acc[nt][r];
            }
            if (stats) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double d = (double)acc[nt][r];
                    s1[nt] += d;
                    s2[nt] = __builtin_fma(d, d, s2[nt]);
                }
            }
        }
        tile += tstep;
    };
    if (tile < ntiles) fetch(tile);
    while (tile < ntiles) process();
    if (stats) {
        const unsigned slot = (blockIdx.x * NW + wave) % (unsigned)kPn2BnSlots;
        double* __restrict__ sl = stats + kPn2BnHead + (size_t)2 * COUT * (1 + slot);
#pragma unroll
        for (int nt = 0; nt < CO; ++nt) {
            const double d1 = s1[nt] + __shfl_xor(s1[nt], 32), d2 = s2[nt] + __shfl_xor(s2[nt], 32);
            if (half == 0) {
                atomicAdd(sl + nt * 32 + l31, d1);
                atomicAdd(sl + COUT + nt * 32 + l31, d2);
            }
        }
    }
    pn2_bn_finish(fin, gridDim.x, blockIdx.x);
}

template <int CO>
int launch_fwd_wide_in(int rows, const float* x, const Pn2LoadTransform* xf, const float* w, float* y, double* stats,
                       const Pn2BnFinish& f, hipStream_t st) {
    const int ntiles = rows / 32;
    int blocks = (ntiles + 7) / 8;
    if (blocks > 256) blocks = 256;  // one workgroup of eight waves per CU
    constexpr size_t lds = sizeof(float) * ((size_t)16 * CO * 256 + 8 * 32 * (64 + 4));
    static bool attr_set = false;  // per instantiation; benign race (idempotent call)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_wide_in_kernel<CO, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_wide_in_kernel<CO, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (xf) fwd_wide_in_kernel<CO, true><<<blocks, 512, lds, st>>>(rows, x, *xf, w, y, stats, f);
    else fwd_wide_in_kernel<CO, false><<<blocks, 512, lds, st>>>(rows, x, Pn2LoadTransform{}, w, y, stats, f);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// cin in {32, 64}, cout in {32, 64, 128}, rows % 32 == 0 and enough of them that every SIMD gets several tiles; 16-byte aligned
// operands (checked by the callers' own preconditions for the load-transform form).
inline bool fwd_narrow_fits(int rows, int cin, int cout, const void* x, const void* w_) {
    if (cin != 32 && cin != 64 && !(cin == 128 && cout == 128)) return false;
    if (cout != 32 && cout != 64 && cout != 128) return false;
    if (rows % 32 != 0 || rows < PN2_STREAM_MIN_ROWS) return false;
    return (((uintptr_t)x | (uintptr_t)w_) % 16) == 0;
}

template <int CI, int CO>
int launch_fwd_narrow_one(int rows, const float* x, const Pn2LoadTransform* xf, const float* w, float* y, double* stats,
                          const Pn2BnFinish& f, hipStream_t st) {
    const int ntiles = rows / 32;
    int blocks = (ntiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;  // ~4 workgroups per CU: several tiles per wave, one pair of atomics per column and wave
    constexpr size_t lds = sizeof(float) * ((size_t)(CI * 4) * CO * 256 + 4 * 32 * (32 * CI + 4));
    if constexpr (lds > 64 * 1024) {
        static bool attr_set = false;  // per instantiation; benign race (idempotent call)
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_narrow_kernel<CI, CO, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_narrow_kernel<CI, CO, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
    }
    if (xf) fwd_narrow_kernel<CI, CO, true><<<blocks, 256, lds, st>>>(rows, x, *xf, w, y, stats, f);
    else fwd_narrow_kernel<CI, CO, false><<<blocks, 256, lds, st>>>(rows, x, Pn2LoadTransform{}, w, y, stats, f);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

inline int launch_fwd_narrow(int rows, int cin, int cout, const float* x, const Pn2LoadTransform* xf, const float* w, float* y,
                             double* stats, const Pn2BnFinish* fin, hipStream_t st) {
    const Pn2BnFinish f = fin ? *fin : Pn2BnFinish{};
    if (cin == 128) return launch_fwd_wide_in<4>(rows, x, xf, w, y, stats, f, st);
#define PN2_FWN(CI_, CO_) return launch_fwd_narrow_one<CI_, CO_>(rows, x, xf, w, y, stats, f, st)
    if (cin == 32) {
        if (cout == 32) PN2_FWN(1, 1);
        if (cout == 64) PN2_FWN(1, 2);
        PN2_FWN(1, 4);
    }
    if (cout == 32) PN2_FWN(2, 1);
    if (cout == 64) PN2_FWN(2, 2);
    PN2_FWN(2, 4);
#undef PN2_FWN
}

}  // namespace
