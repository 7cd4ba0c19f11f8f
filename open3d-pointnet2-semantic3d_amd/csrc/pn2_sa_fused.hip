// pn2_sa_fused.hip -- fused set-abstraction MLP for gfx950:
//   gather(idx) -> [xyz - centre | features] -> up to 3 x (1x1 conv + bias + ReLU)
//   -> max over the K=32 neighbours, with the grouped (B,M,K,C) tensor and all
//   intermediate activations living only in VGPRs.
// Replaces the TF sub-graph util/pointnet_util.py:43-54 (group/centre/concat) +
// :150-162 (conv2d stack, util/tf_util.py:181-203 with inference BN folded by the
// host) + :167-170 (reduce_max over K).  No reference kernel exists for it.
//
// Mapping onto v_mfma_f32_32x32x2_f32 (exact fp32):
//   * one wave64 owns one (b, j) neighbourhood = 32 rows; lane l carries
//     neighbour (l & 31); the two half-waves carry two different input channels,
//     which are the k = 0 / k = 1 slices of one 32x32x2 MFMA.
//   * hidden layers are computed TRANSPOSED, D^T[channel][neighbour] =
//     W^T x X^T (A operand = weights, B operand = activations).  The accumulator
//     layout (lane = neighbour, register r -> channel (r&3)+8*(r>>2)+4*half) is
//     then directly the B-operand layout of the next layer: no LDS round trip, no
//     cross-lane traffic between layers.  The contraction index is visited in
//     that (permuted) channel order; the weights are stored in LDS pre-permuted
//     to match ([k-step][half][cout], conflict-free ds_read_b32).
//   * the last layer is computed un-transposed (A = activations, B = weights) so
//     the K-max is an in-lane max over the 16 accumulator registers plus one
//     half-wave exchange, and the pooled row is written as 128-byte segments.
// Weights (<= 3 layers, widths <= 128) stay resident in LDS for the whole
// persistent workgroup.
#include "pn2_common.h"
#include <type_traits>
#include <utility>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SaFusedParams {
    int n, m, c, groups;  // groups = 32-row tiles: b*m*K/32 (gather mode) or ceil(rows/32) (dense mode)
    int kshift;           // gather mode: log2(K), K = neighbours per centre (16, 32, 64, 128, ...)
    int rows;             // dense mode: number of input rows
    int w[3];             // layer widths
    const float* xyz;
    const float* new_xyz;
    const float* points;
    const int* idx;
    const float* dist;     // INTERP mode: three_nn distances (rows, 3); idx is then (rows, 3)
    const float* points1;  // INTERP mode: skip-link features (rows, c1) or nullptr
    int c1;
    // row strides in floats, 0 = dense (3 / c / c1): the xyz and rgb columns of a (b,n,6) batch read in place (model.py:26-29).
    // ldp applies to the un-vectorised feature path only (c % 8 != 0); launch_chain fills the zeros in.
    int ldx, ldp, ld1;
    const float* W[3];
    const float* bias[3];
    float* out;
    int prio;  // 1: stagger the two waves of a SIMD (see the kernel)
    int schedule;  // FP chain: 0 = lockstep kernel, 1 = software-pipelined kernel, -1 = the default (pipelined where it applies)
#ifdef PN2_TUNING_HOOKS
    long long* stats;  // tuning builds: cycle stamps of the first workgroups (tools/chain_stage_ab.py)
    long long* trace;  // tuning builds: per-wave (start, end) in 100 MHz ticks of the last 16 launches of every graph (tools/insitu_chain.py)
    int tag;           //   which graph (0..7) this launch belongs to: read from g_chain_tag when the launch is captured
#endif
};

// stage stamp k of tile t (tuning builds only): lane 0 of waves 0 and NW/2 of workgroups 0..3
#ifdef PN2_TUNING_HOOKS
#define PN2_CHAIN_STAMP(k_)                                                                                              \
    do {                                                                                                                 \
        if (p.stats && blockIdx.x < 4 && lane == 0 && (wave == 0 || wave == NW / 2) && tile_no < 4)                      \
            p.stats[((blockIdx.x * 2 + (wave ? 1 : 0)) * 4 + tile_no) * 8 + (k_)] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#else
#define PN2_CHAIN_STAMP(k_) do { } while (0)
#endif

__device__ __forceinline__ int acc_chan(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// number of k-steps of layer 1 for c feature channels
// (c1 = extra un-vectorised channels appended after the c vectorised ones: FP skip link)
__host__ __device__ inline int l1_steps(int c, bool vec8, bool dense = false, int c1 = 0) {
    return (dense ? 0 : 2) + (vec8 ? (c / 8) * 4 : (c + 1) / 2) + (c1 + 1) / 2;
}

// input channel (row of W1) fed by half-wave `h` at k-step `s` of layer 1; -1 = zero pad
__device__ __forceinline__ int l1_chan(int s, int h, int c, bool vec8, bool dense, int c1 = 0) {
    int base = 0;
    if (!dense) {
        if (s == 0) return h;              // x | y
        if (s == 1) return h ? -1 : 2;     // z | 0
        s -= 2;
        base = 3;
    }
    const int nmain = vec8 ? (c / 8) * 4 : (c + 1) / 2;
    if (s >= nmain) {  // appended skip-link channels
        const int ch = 2 * (s - nmain) + h;
        return ch < c1 ? base + c + ch : -1;
    }
    if (vec8) return base + 8 * (s >> 2) + 4 * h + (s & 3);
    const int ch = 2 * s + h;
    return ch < c ? base + ch : -1;
}

// LDS weight layout of sa_fused_kernel (r04): row (k-step s, half-wave h) = w floats at Wp[(2 s + h) w]; inside a row the NT
// column tiles are INTERLEAVED -- column 32 nt + l sits at l NT + nt -- so that a lane fetches its NT operands of a k-step with
// ONE LDS read (ds_read_b128 at NT = 4, b64 at 2) instead of NT reads at stride 32: non-MFMA instructions cost their full
// issue time on a SIMD whose matrix pipe is busy (fp_chain_pipe_kernel's header).  `wrow` = Wp + (2 s + h) w + l31 NT.
typedef float f32x2w __attribute__((ext_vector_type(2)));
template <int NT>
__device__ __forceinline__ void load_w(float (&wv)[NT], const float* __restrict__ wrow) {
    if constexpr (NT == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wrow);
        wv[0] = v[0]; wv[1] = v[1]; wv[2] = v[2]; wv[3] = v[3];
    } else if constexpr (NT == 2) {
        const f32x2w v = *reinterpret_cast<const f32x2w*>(wrow);
        wv[0] = v[0]; wv[1] = v[1];
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wv[nt] = wrow[nt];
    }
}

template <int NT, bool LAST>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[NT], const float* __restrict__ wrow, float act) {
    float wv[NT];
    load_w<NT>(wv, wrow);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (LAST) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(act, wv[nt], acc[nt], 0, 0, 0);
        else acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[nt], act, acc[nt], 0, 0, 0);
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
}

// hidden-layer epilogue: relu(acc + bias[channel]) in the transposed layout
template <int NT>
__device__ __forceinline__ void bias_relu_T(f32x16 (&acc)[NT], const float* __restrict__ sbias, int half) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[nt][r] = fmaxf(acc[nt][r] + sbias[nt * 32 + acc_chan(r, half)], 0.f);
}

// The same epilogue with a quarter of the LDS reads and half of the adds: registers 4q .. 4q+3 of a column tile hold the
// channels 32 nt + 8 q + 4 half + (0..3), i.e. FOUR CONSECUTIVE biases -- one ds_read_b128 -- and the adds go two at a time
// (v_pk_add_f32).  Same operations on the same values: bit-identical to bias_relu_T.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NT>
__device__ __forceinline__ void bias_relu_T_pk(f32x16 (&acc)[NT], const float* __restrict__ sbias_half /* sbias + 4 * half */) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(sbias_half + nt * 32 + 8 * q);
            const f32x2 lo = f32x2{acc[nt][4 * q], acc[nt][4 * q + 1]} + f32x2{b[0], b[1]};
            const f32x2 hi = f32x2{acc[nt][4 * q + 2], acc[nt][4 * q + 3]} + f32x2{b[2], b[3]};
            acc[nt][4 * q] = fmaxf(lo[0], 0.f); acc[nt][4 * q + 1] = fmaxf(lo[1], 0.f);
            acc[nt][4 * q + 2] = fmaxf(hi[0], 0.f); acc[nt][4 * q + 3] = fmaxf(hi[1], 0.f);
        }
}

// dense layer fed from the previous layer's accumulators.  The weight reads of
// step s+1 are issued ahead of the MFMAs of step s and a sched_barrier pins that
// order: without it the scheduler hoists all NTP*16*NT ds_reads of the unrolled
// body to the top and the kernel spills.

template <int NT, bool LAST>
__device__ __forceinline__ void mfma_regs(f32x16 (&acc)[NT], const float (&wv)[NT], float act) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (LAST) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(act, wv[nt], acc[nt], 0, 0, 0);
        else acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[nt], act, acc[nt], 0, 0, 0);
    }
}

// Weight prefetch distance in k-steps: the ds_reads of step s+kPF are issued right after the MFMAs of step s
// (into the register slot that step just consumed).  tools/mfma_lds.hip: 136 / 142 / 149 TFLOP/s at distance
// 1 / 2 / 4 for this loop shape at 2 waves/SIMD (and 100 without the sched_barrier pinning).
constexpr int kPF = 4;

template <int NTP, int NT, bool LAST>
__device__ __forceinline__ void layer_from_regs(const f32x16 (&in)[NTP], f32x16 (&acc)[NT],
                                                const float* __restrict__ wp, int w, int half, int l31) {
    const float* __restrict__ wl = wp + half * w + l31 * NT;
    constexpr int S = NTP * 16;
    float wq[kPF][NT];
#pragma unroll
    for (int pf = 0; pf < kPF; ++pf) load_w<NT>(wq[pf], wl + pf * 2 * w);
#pragma unroll
    for (int ntp = 0; ntp < NTP; ++ntp)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = ntp * 16 + r;
            mfma_regs<NT, LAST>(acc, wq[s % kPF], in[ntp][r]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + kPF < S) {
                load_w<NT>(wq[s % kPF], wl + (s + kPF) * 2 * w);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
}

// last-layer epilogue: max over the 32 rows, + bias, relu, store 32 floats per tile
template <int NT>
__device__ __forceinline__ void pool_store(const f32x16 (&acc)[NT], const float* __restrict__ sbias,
                                           float* __restrict__ orow, int half, int l31) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float v = acc[nt][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[nt][r]);
        v = fmaxf(v, __shfl_xor(v, 32));
        v = fmaxf(v + sbias[nt * 32 + l31], 0.f);
        if (half == 0) orow[nt * 32 + l31] = v;
    }
}

// pooled epilogue for K != 32 (kshift != 5): K = 16 -> the tile holds two centres (rows 0..15 = accumulator
// registers 0..7 of both half-waves, rows 16..31 = registers 8..15); K = 32*t -> t tiles share a centre and merge
// their maxima with an integer atomicMax on the zero-initialised output (values are post-ReLU, >= +0).
template <int NT>
__device__ __forceinline__ void pool_store_k(const f32x16 (&acc)[NT], const float* __restrict__ sbias,
                                             float* __restrict__ out, int wout, int g, int kshift, int half, int l31) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bv = sbias[nt * 32 + l31];
        if (kshift == 4) {
            float v0 = acc[nt][0], v1 = acc[nt][8];
#pragma unroll
            for (int r = 1; r < 8; ++r) { v0 = fmaxf(v0, acc[nt][r]); v1 = fmaxf(v1, acc[nt][8 + r]); }
            v0 = fmaxf(v0, __shfl_xor(v0, 32));
            v1 = fmaxf(v1, __shfl_xor(v1, 32));
            if (half == 0) {
                out[(size_t)(2 * g) * wout + nt * 32 + l31] = fmaxf(v0 + bv, 0.f);
                out[(size_t)(2 * g + 1) * wout + nt * 32 + l31] = fmaxf(v1 + bv, 0.f);
            }
        } else {
            float v = acc[nt][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[nt][r]);
            v = fmaxf(v, __shfl_xor(v, 32));
            v = fmaxf(v + bv, 0.f);
            if (half == 0)
                atomicMax(reinterpret_cast<int*>(out + (size_t)(g >> (kshift - 5)) * wout + nt * 32 + l31), __float_as_int(v));
        }
    }
}

// last-layer epilogue without pooling: relu(acc + bias) for all 32 rows of the tile
template <int NT>
__device__ __forceinline__ void rows_store(const f32x16 (&acc)[NT], const float* __restrict__ sbias,
                                           float* __restrict__ obase, int wout, int row0, int rows,
                                           int half, int l31) {
    if (row0 + 32 <= rows) {  // whole tile in range (wave-uniform): no per-store exec masking, constant offsets
        float* __restrict__ o = obase + (size_t)(row0 + 4 * half) * wout + l31;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float bv = sbias[nt * 32 + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                o[((r & 3) + 8 * (r >> 2)) * wout + nt * 32] = fmaxf(acc[nt][r] + bv, 0.f);
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bv = sbias[nt * 32 + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + acc_chan(r, half);
            if (row < rows) obase[(size_t)row * wout + nt * 32 + l31] = fmaxf(acc[nt][r] + bv, 0.f);
        }
    }
}

// rows_store for whole tiles with the bias added two registers at a time (same values as rows_store)
template <int NT>
__device__ __forceinline__ void rows_store_pk(const f32x16 (&acc)[NT], const float* __restrict__ sbias, float* __restrict__ obase,
                                              int wout, int row0, int rows, int half, int l31) {
    if (row0 + 32 > rows) { rows_store<NT>(acc, sbias, obase, wout, row0, rows, half, l31); return; }
    // four row-group bases (rows 8 q + 4 half + 0..3): every store then addresses base + a constant below 4 KB, instead of a
    // 64-bit add per store
    float* __restrict__ oq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) oq[q] = obase + (size_t)(row0 + 4 * half + 8 * q) * wout + l31;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bv = sbias[nt * 32 + l31];
        const f32x2 b2 = {bv, bv};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 v = f32x2{acc[nt][r], acc[nt][r + 1]} + b2;
            oq[r >> 2][(r & 3) * wout + nt * 32] = fmaxf(v[0], 0.f);
            oq[r >> 2][((r + 1) & 3) * wout + nt * 32] = fmaxf(v[1], 0.f);
        }
    }
}

// DENSE = false: set-abstraction mode (rows gathered by idx, [xyz - centre | features]).
// DENSE = true : plain rows of a (rows, c) matrix (feature-propagation MLPs); POOL selects the
//                max over each 32-row tile or the full (rows, wout) output.
// INTERP (with DENSE, VEC8): the rows are not read but produced on the fly as the feature-propagation
//                front end [ three_interpolate(points2, idx, w(dist)) | points1 ] (pointnet_util.py:300-311),
//                same fp32 operation order as fp_interp_concat_kernel / the reference ops.
// NW waves per workgroup share one LDS copy of the weights.
// PREZ (with INTERP): the first layer's product with the INTERPOLATED channels is hoisted out of the kernel by
//                linearity: interp(points2) @ W1a == interp(points2 @ W1a), and Z = points2 @ W1a has m rows per cloud
//                instead of n (8x fewer at every FP level of semantic.json).  p.points is then Z (b*m, W1): the three
//                gathered rows of Z are blended STRAIGHT INTO the accumulator layout of layer 1 (register r of tile nt
//                <-> channel 32 nt + (r & 3) + 8 (r >> 2) + 4 half: four consecutive channels per 16-byte load), only the
//                c1 skip-link channels still go through the MFMA.  Same gather traffic, (c2 / 2) * NT1 fewer MFMAs per tile.
template <int L, int NT1, int NT2, int NT3, bool VEC8, bool DENSE, bool POOL, int NW, bool INTERP = false, bool PREZ = false>
__global__ void __launch_bounds__(NW * 64, (NW >= 8 ? NW / 4 : 2))  // NW <= 8: 2 waves/SIMD (<= 256 VGPR+AGPR per lane)
sa_fused_kernel(SaFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int W1 = NT1 * 32, W2 = NT2 * 32, W3 = NT3 * 32;
    const int c = PREZ ? 0 : p.c;  // PREZ: layer 1's LDS weights cover the skip-link channels only
    const int c1 = INTERP ? p.c1 : 0;
    const int steps1 = l1_steps(c, VEC8, DENSE, c1);
    constexpr int NTH = NW * 64;

    int tile_no = 0;
    (void)tile_no;
    PN2_CHAIN_STAMP(0);
    // Waves w and w + NW/2 share a SIMD and run in lockstep: all gather (the CU's vector-memory pipe is the limit: 8 waves
    // x 48 KB), then all want the matrix pipe, then all store -- the matrix pipe idles during the gather / store phases,
    // 35-40 % of a tile (stage stamps, tools/chain_stage_ab.py; profiles/r03_fp4_chain_stages.txt).  STAGGER (tuning
    // experiment, off): the second half of the waves holds its first tile back until its SIMD partner has finished the MFMA
    // layers of its own first tile.  Measured: no gain -- ONE wave drives the matrix pipe at half rate (34 k cycles for a
    // 128 -> 128 layer alone, 2 x 16 k for two waves together), so a SIMD needs both of its waves in the MFMA phase, and a
    // third wave per SIMD does not fit the register budget (12-wave workgroups: 128 vs 98 us).
#ifdef PN2_TUNING_HOOKS
    const bool stagger = NW == 8 && L >= 2 && p.prio != 0;
#else
    constexpr bool stagger = false;  // the experiment is compiled out of the shipped library
#endif
    // ---- LDS carve + weight staging (once per persistent workgroup) -----------
    float* wp1 = smem;
    float* wp2 = wp1 + steps1 * 2 * W1;
    float* wp3 = wp2 + (L >= 2 ? W1 * W2 : 0);            // (W1/2 steps) * 2 * W2
    float* sb1 = wp3 + (L >= 3 ? W2 * W3 : 0);
    float* sb2 = sb1 + W1;
    float* sb3 = sb2 + (L >= 2 ? W2 : 0);
    int* stagger_flag = reinterpret_cast<int*>(sb3 + (L >= 3 ? W3 : 0));  // NW ints behind the biases (launch_chain sizes them in)
    if (stagger && tid < NW) stagger_flag[tid] = 0;
    // Weight staging, 8 independent global loads in flight per thread (the permutation index maths is
    // cheap; what must be hidden is the L2 latency -- a workgroup may own only a handful of tiles).
    // 16-byte version: index_of(e4) returns the global FLOAT index of 4 consecutive columns (or -1).
    auto stage4 = [&](float* dst, int count4, const float* __restrict__ src, auto index_of, int ntw) {
        const int W = ntw * 32;
        for (int base = 0; base < count4; base += NTH * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * NTH + tid;
                const int gi = e < count4 ? index_of(e) : -1;
                const f32x4 t = *reinterpret_cast<const f32x4*>(src + (gi >= 0 ? gi : 0));
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                v[u] = gi >= 0 ? t : z;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * NTH + tid;
                if (e < count4) {
                    if (ntw == 1) {
                        *reinterpret_cast<f32x4*>(dst + e * 4) = v[u];
                    } else {  // columns col .. col+3 of row sh share their column tile: position (col & 31) ntw + (col >> 5), stride ntw
                        const int col = (e % (W / 4)) * 4, sh = e / (W / 4);
                        float* __restrict__ d = dst + sh * W + (col & 31) * ntw + (col >> 5);
#pragma unroll
                        for (int j = 0; j < 4; ++j) d[j * ntw] = v[u][j];
                    }
                }
            }
        }
    };
    auto stage1 = [&](float* dst, int count, const float* __restrict__ src) {
        for (int e = tid; e < count; e += NTH) dst[e] = src[e];
    };
    stage4(wp1, steps1 * 2 * W1 / 4, p.W[0], [&](int e4) {
        const int col = (e4 % (W1 / 4)) * 4, sh = e4 / (W1 / 4);
        const int ch = l1_chan(sh >> 1, sh & 1, c, VEC8, DENSE, c1);
        return ch >= 0 ? ch * W1 + col : -1;
    }, NT1);
    stage1(sb1, W1, p.bias[0]);
    if constexpr (L >= 2) {
        stage4(wp2, W1 * W2 / 4, p.W[1], [&](int e4) {
            const int col = (e4 % (W2 / 4)) * 4, sh = e4 / (W2 / 4);
            const int s = sh >> 1, h = sh & 1;
            return ((s >> 4) * 32 + acc_chan(s & 15, h)) * W2 + col;
        }, NT2);
        stage1(sb2, W2, p.bias[1]);
    }
    if constexpr (L >= 3) {
        stage4(wp3, W2 * W3 / 4, p.W[2], [&](int e4) {
            const int col = (e4 % (W3 / 4)) * 4, sh = e4 / (W3 / 4);
            const int s = sh >> 1, h = sh & 1;
            return ((s >> 4) * 32 + acc_chan(s & 15, h)) * W3 + col;
        }, NT3);
        stage1(sb3, W3, p.bias[2]);
    }
    __syncthreads();
    PN2_CHAIN_STAMP(1);

    constexpr int WOUT = L == 1 ? W1 : (L == 2 ? W2 : W3);
    // XCD-aware tile order (speed only): the dispatcher places workgroup w on XCD w % 8.  Giving XCD x the
    // contiguous tile range [x*G/8, (x+1)*G/8) = whole batch elements makes each private 4 MiB L2 gather
    // from 1/8 of the feature table instead of all of it.
    int g_lo = blockIdx.x * NW + wave, g_hi = p.groups, g_step = gridDim.x * NW;
    if ((gridDim.x & 7u) == 0u && (p.groups & 7) == 0) {
        const int per = p.groups >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        g_lo = xcd * per + slot * NW + wave;
        g_hi = (xcd + 1) * per;
        g_step = (gridDim.x >> 3) * NW;
    }
    for (int g = g_lo; g < g_hi; g += g_step, ++tile_no) {
        PN2_CHAIN_STAMP(2);
        size_t prow;  // row of the feature matrix feeding this lane
        f32x16 a1[NT1];
        zero_acc<NT1>(a1);
        constexpr bool LAST1 = (L == 1);
        const float* w1l = wp1 + half * W1 + l31 * NT1;
        constexpr int S0 = DENSE ? 0 : 2;  // first feature k-step
        if constexpr (DENSE) {
            const int row = g * 32 + l31;
            prow = (size_t)(row < p.rows ? row : p.rows - 1);
        } else {
            // tile g = rows [32g, 32g+32) of the flat (b*m*K) neighbour list; K = 32: one centre per tile,
            // K = 16: two centres per tile, K = 64, 128, ...: a centre spans several tiles (max merged at the end)
            const int grp = (g * 32 + l31) >> p.kshift;
            const int bi = grp / p.m;
            const int ii = p.idx[(size_t)g * 32 + l31];
            prow = (size_t)bi * p.n + ii;
            const float cxv = p.new_xyz[(size_t)grp * 3 + 0];
            const float cyv = p.new_xyz[(size_t)grp * 3 + 1];
            const float czv = p.new_xyz[(size_t)grp * 3 + 2];
            const float rx = p.xyz[prow * p.ldx + 0] - cxv;  // grouped_xyz -= tile(new_xyz) :44-46
            const float ry = p.xyz[prow * p.ldx + 1] - cyv;
            const float rz = p.xyz[prow * p.ldx + 2] - czv;
            mfma_step<NT1, LAST1>(a1, w1l + 0 * 2 * W1, half ? ry : rx);
            mfma_step<NT1, LAST1>(a1, w1l + 1 * 2 * W1, half ? 0.f : rz);
        }
        if constexpr (INTERP) {
            // inverse-distance weights exactly as fp_interp_concat_kernel (IEEE divisions)
            const float* __restrict__ dr = p.dist + prow * 3;
            const int* __restrict__ ir = p.idx + prow * 3;
            const float d1 = fmaxf(dr[0], 1e-10f), d2 = fmaxf(dr[1], 1e-10f), d3 = fmaxf(dr[2], 1e-10f);
            const float r1 = 1.0f / d1, r2 = 1.0f / d2, r3 = 1.0f / d3;
            const float norm = (r1 + r2) + r3;
            const float w1 = r1 / norm, w2 = r2 / norm, w3 = r3 / norm;
            const size_t kb = (prow / (size_t)p.n) * (size_t)p.m;  // first known row of this batch element
            if constexpr (PREZ) {
                const f32x4* __restrict__ z1 = reinterpret_cast<const f32x4*>(p.points + (kb + ir[0]) * W1) + half;
                const f32x4* __restrict__ z2 = reinterpret_cast<const f32x4*>(p.points + (kb + ir[1]) * W1) + half;
                const f32x4* __restrict__ z3 = reinterpret_cast<const f32x4*>(p.points + (kb + ir[2]) * W1) + half;
                if (c1 > 0) {  // skip-link channels: the only MFMA work left in layer 1
                    const float* __restrict__ fp = p.points1 + prow * p.ld1;
                    const int ns = (c1 + 1) >> 1;
                    for (int sp = 0; sp < ns; ++sp) {
                        const int ch = 2 * sp + half;
                        const float v = ch < c1 ? fp[ch] : 0.f;
                        mfma_step<NT1, LAST1>(a1, w1l + sp * 2 * W1, v);
                    }
                }
                // (All 3 x 4 NT1 gathers in flight at once instead of per column tile: no change -- the phase is bound by the
                // CU's vector-memory pipe, 8 waves x 48 KB at the same time, not by round trips; tools/chain_stage_ab.py.)
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt) {
                    f32x4 q1[4], q2[4], q3[4];
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {  // float4 index of channel 32 nt + 8 rq + 4 half
                        q1[rq] = z1[nt * 8 + rq * 2]; q2[rq] = z2[nt * 8 + rq * 2]; q3[rq] = z3[nt * 8 + rq * 2];
                    }
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x4 cur = (q1[rq] * w1 + q2[rq] * w2) + q3[rq] * w3;
#pragma unroll
                        for (int e = 0; e < 4; ++e) a1[nt][4 * rq + e] += cur[e];
                    }
                }
            } else {
            const f32x4* __restrict__ f1 = reinterpret_cast<const f32x4*>(p.points + (kb + ir[0]) * c) + half;
            const f32x4* __restrict__ f2 = reinterpret_cast<const f32x4*>(p.points + (kb + ir[1]) * c) + half;
            const f32x4* __restrict__ f3 = reinterpret_cast<const f32x4*>(p.points + (kb + ir[2]) * c) + half;
            const int nt8 = c >> 3;
            f32x4 c1v = f1[0], c2v = f2[0], c3v = f3[0];
            float wq[4][NT1];
#pragma unroll
            for (int q = 0; q < 4; ++q) load_w<NT1>(wq[q], w1l + q * 2 * W1);
            for (int t = 0; t < nt8; ++t) {
                const int tn = (t + 1 < nt8 ? t + 1 : t) * 2;  // unconditional (clamped): counted vmcnt
                const f32x4 n1 = f1[tn], n2 = f2[tn], n3 = f3[tn];
                const f32x4 cur = (c1v * w1 + c2v * w2) + c3v * w3;  // tf_interpolate.cpp:322-324 order, unfused
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // slot q is refilled for the next t right after it is consumed
                    mfma_regs<NT1, LAST1>(a1, wq[q], cur[q]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int sn = 4 * (t + 1) + q;
                    load_w<NT1>(wq[q], w1l + (sn < steps1 ? sn : steps1 - 1) * 2 * W1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                c1v = n1; c2v = n2; c3v = n3;
            }
            if (c1 > 0) {
                const float* __restrict__ fp = p.points1 + prow * p.ld1;
                const int ns = (c1 + 1) >> 1;
                for (int sp = 0; sp < ns; ++sp) {
                    const int ch = 2 * sp + half;
                    const float v = ch < c1 ? fp[ch] : 0.f;
                    mfma_step<NT1, LAST1>(a1, w1l + (4 * nt8 + sp) * 2 * W1, v);
                }
            }
            }  // !PREZ
        } else if constexpr (PREZ) {
            // set abstraction with the FEATURE part of layer 1 hoisted: [x - c | f] @ W1 = (x - c) @ W1[:3] + (f @ W1[3:]), and
            // zf = points @ W1[3:] has one row per SOURCE point (n per cloud) instead of one per grouped neighbour (m K):
            // the neighbour's row of zf is added straight into the accumulator layout.  (The xyz part stays on the MFMA --
            // hoisting it too would subtract two large products, x @ W and c @ W, to get a small one.)
            const f32x4* __restrict__ zr = reinterpret_cast<const f32x4*>(p.points + prow * W1) + half;
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) {
                f32x4 q[4];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) q[rq] = zr[nt * 8 + rq * 2];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a1[nt][4 * rq + e] += q[rq][e];
            }
        } else if constexpr (VEC8) {
            const f32x4* __restrict__ fp =
                reinterpret_cast<const f32x4*>(p.points + prow * c) + half;
            const int nt8 = c >> 3;
            f32x4 cur = fp[0];
            float wq[4][NT1];
#pragma unroll
            for (int q = 0; q < 4; ++q) load_w<NT1>(wq[q], w1l + (S0 + q) * 2 * W1);
            for (int t = 0; t < nt8; ++t) {
                const f32x4 nxt = fp[(t + 1 < nt8 ? t + 1 : t) * 2];  // unconditional (clamped): counted vmcnt
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // slot q is refilled for the next t right after it is consumed
                    mfma_regs<NT1, LAST1>(a1, wq[q], cur[q]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int sn = S0 + 4 * (t + 1) + q;
                    load_w<NT1>(wq[q], w1l + (sn < steps1 ? sn : steps1 - 1) * 2 * W1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                cur = nxt;
            }
        } else {
            const float* __restrict__ fp = p.points + prow * p.ldp;
            const int ns = (c + 1) >> 1;
            for (int sp = 0; sp < ns; ++sp) {
                const int ch = 2 * sp + half;
                const float v = ch < c ? fp[ch] : 0.f;
                mfma_step<NT1, LAST1>(a1, w1l + (S0 + sp) * 2 * W1, v);
            }
        }
        float* __restrict__ orow = p.out + (size_t)g * WOUT;
        PN2_CHAIN_STAMP(3);
        if (stagger && tile_no == 0 && wave >= NW / 2) {
            while (__hip_atomic_load(&stagger_flag[wave - NW / 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
                __builtin_amdgcn_s_sleep(8);
        }
        if constexpr (L == 1) {
            if constexpr (POOL) {
                if (DENSE || p.kshift == 5) pool_store<NT1>(a1, sb1, orow, half, l31);
                else pool_store_k<NT1>(a1, sb1, p.out, WOUT, g, p.kshift, half, l31);
            }
            else rows_store<NT1>(a1, sb1, p.out, WOUT, g * 32, p.rows, half, l31);
        } else {
            bias_relu_T_pk<NT1>(a1, sb1 + 4 * half);
            f32x16 a2[NT2];
            zero_acc<NT2>(a2);
            layer_from_regs<NT1, NT2, L == 2>(a1, a2, wp2, W2, half, l31);
            PN2_CHAIN_STAMP(4);
            if (L == 2 && stagger && tile_no == 0 && wave < NW / 2 && lane == 0)
                __hip_atomic_store(&stagger_flag[wave], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (L == 2) {
                if constexpr (POOL) {
                if (DENSE || p.kshift == 5) pool_store<NT2>(a2, sb2, orow, half, l31);
                else pool_store_k<NT2>(a2, sb2, p.out, WOUT, g, p.kshift, half, l31);
            }
                else rows_store<NT2>(a2, sb2, p.out, WOUT, g * 32, p.rows, half, l31);
            } else {
                bias_relu_T_pk<NT2>(a2, sb2 + 4 * half);
                f32x16 a3[NT3];
                zero_acc<NT3>(a3);
                layer_from_regs<NT2, NT3, true>(a2, a3, wp3, W3, half, l31);
                PN2_CHAIN_STAMP(5);
                if (stagger && tile_no == 0 && wave < NW / 2 && lane == 0)
                    __hip_atomic_store(&stagger_flag[wave], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if constexpr (POOL) {
                if (DENSE || p.kshift == 5) pool_store<NT3>(a3, sb3, orow, half, l31);
                else pool_store_k<NT3>(a3, sb3, p.out, WOUT, g, p.kshift, half, l31);
            }
                else rows_store<NT3>(a3, sb3, p.out, WOUT, g * 32, p.rows, half, l31);
            }
        }
        PN2_CHAIN_STAMP(6);
    }
    // a wave of the first half without a (complete) first tile must not leave its partner waiting
    if (stagger && wave < NW / 2 && lane == 0)
        __hip_atomic_store(&stagger_flag[wave], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- software-pipelined feature-propagation chain (FP4 of semantic.json: 131072 rows, 3 x 128) ----------------------------
// Same arithmetic, bit for bit, as sa_fused_kernel<3, 4, 4, 4, VEC8, DENSE, !POOL, NW, INTERP, PREZ> -- same MFMA order, same
// blend expression -- with another SCHEDULE.  There, the 8 waves of a CU gather in lockstep (3 rows of z x 512 B per output
// row = 48 KB per wave and tile through a vector L1 that turns around ~16 B per cycle: 18-31 k cycles), then all run their
// MFMA layers, then all store: the phases add up (profiles/r03_fp4_chain_stages.txt).  What CAN overlap on this machine is
// memory latency with MFMA issue -- not instruction issue: while a SIMD's matrix pipe is busy, neither the issuing wave nor
// its SIMD partner gets VALU or vector-memory instructions through (tools/mfma_interference_ubench.hip: a partner's VALU chain
// runs at one instruction per ~180 cycles beside a saturating MFMA stream; the MFMA wave itself keeps 95 % of the pipe alone).
// So: ONE wave per SIMD (4-wave workgroups, 512 registers per lane) owns 4 tiles and issues the loads of tile t + 1 between
// the MFMA steps of tile t, a column chunk at a time (12 x 16 B per lane), blending a chunk into the layer-1 accumulator
// ~24 steps (6 k cycles) after its loads went out.  Column tile c of that accumulator is dead once layer 2 has consumed its 16
// steps, and is rebuilt in place (zero, skip-link MFMAs, blend): no second copy.  The first tile's gather overlaps the weight
// staging; exposed are its tail and the last tile's store.
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), unrolled by construction (a `#pragma
// unroll` over a body of this size is declined by the unroller, and a rolled loop indexes the accumulators dynamically)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

constexpr int kPipeSkipSteps = 4;  // skip-link channels c1 <= 8 (two per MFMA step)
// NS: skip-link MFMA steps executed per column tile (2: c1 <= 4, the FP4 layer; 4: c1 <= 8).  Steps beyond (c1 + 1) / 2 multiply
// zero-filled weight rows by zero values: no run-time branch around an accumulator update (a branch there turns every
// 16-register accumulator tuple into a phi of two paths, and the compiler then moves whole tuples around at each merge).

template <int NT2, int NT3, int NS>
__global__ void __launch_bounds__(256, 1)
fp_chain_pipe_kernel(SaFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = 4, NTH = 256, NT1 = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int W1 = NT1 * 32, W2 = NT2 * 32, W3 = NT3 * 32;
    static_assert(NT2 == 4 && NT3 == 4, "side-job schedule written for three 128-wide layers");
    const int c1 = p.c1;
    constexpr int steps1 = NS;  // rows beyond the c1 skip-link channels are staged as zeros (l1_chan -> -1)
    float* wp1 = smem;
    float* wp2 = wp1 + steps1 * 2 * W1;
    float* wp3 = wp2 + W1 * W2;
    float* sb1 = wp3 + W2 * W3;
    float* sb2 = sb1 + W1;
    float* sb3 = sb2 + W2;
    int tile_no = 0;
    (void)tile_no;
    PN2_CHAIN_STAMP(0);
#ifdef PN2_TUNING_HOOKS
    const long long t_begin = (long long)__builtin_readcyclecounter();
    const long long t_real0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // XCD-aware tile order, as sa_fused_kernel
    int g_lo = blockIdx.x * NW + wave, g_hi = p.groups, g_step = gridDim.x * NW;
    if ((gridDim.x & 7u) == 0u && (p.groups & 7) == 0) {
        const int per = p.groups >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        g_lo = xcd * per + slot * NW + wave;
        g_hi = (xcd + 1) * per;
        g_step = (gridDim.x >> 3) * NW;
    }
    const bool any_tile = g_lo < g_hi;
    const float* w1l = wp1 + half * W1 + l31;
    // layers 2 / 3: the four column tiles' weights of a (k-step, half-wave) are stored INTERLEAVED, [l31][nt], so that a lane
    // fetches its four B (resp. A) values with ONE ds_read_b128 instead of two ds_read2_b32 and an address add -- every
    // non-MFMA instruction costs its full issue time here (see the header comment)
    const f32x4* w2l = reinterpret_cast<const f32x4*>(wp2 + half * W2) + l31;
    const f32x4* w3l = reinterpret_cast<const f32x4*>(wp3 + half * W3) + l31;

    // ---- front end of a tile, in slices.  fi / fd / fsk: raw loads; fw*: blend weights; fz*: this lane's three rows of z
    int fi0 = 0, fi1 = 0, fi2 = 0;
    float fd0 = 0.f, fd1 = 0.f, fd2 = 0.f, fw1 = 0.f, fw2 = 0.f, fw3 = 0.f;
    float fsk[NS];
    size_t frow = 0;
    const f32x4* __restrict__ fz1 = nullptr;
    const f32x4* __restrict__ fz2 = nullptr;
    const f32x4* __restrict__ fz3 = nullptr;
    auto front_issue = [&](int g) {  // indices, distances, skip-link values: 6 + NS loads per lane
        const int row = g * 32 + l31;
        frow = (size_t)(row < p.rows ? row : p.rows - 1);
        const float* __restrict__ dr = p.dist + frow * 3;
        const int* __restrict__ ir = p.idx + frow * 3;
        fi0 = ir[0]; fi1 = ir[1]; fi2 = ir[2];
        fd0 = dr[0]; fd1 = dr[1]; fd2 = dr[2];
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
            const int ch = 2 * sp + half;
            fsk[sp] = ch < c1 ? p.points1[frow * p.ld1 + ch] : 0.f;
        }
    };
    auto front_weights = [&]() {  // inverse-distance weights exactly as fp_interp_concat_kernel (IEEE divisions)
        const float d1 = fmaxf(fd0, 1e-10f), d2 = fmaxf(fd1, 1e-10f), d3 = fmaxf(fd2, 1e-10f);
        const float r1 = 1.0f / d1, r2 = 1.0f / d2, r3 = 1.0f / d3;
        const float norm = (r1 + r2) + r3;
        fw1 = r1 / norm; fw2 = r2 / norm; fw3 = r3 / norm;
        const size_t kb = (frow / (size_t)p.n) * (size_t)p.m;
        fz1 = reinterpret_cast<const f32x4*>(p.points + (kb + fi0) * W1) + half;
        fz2 = reinterpret_cast<const f32x4*>(p.points + (kb + fi1) * W1) + half;
        fz3 = reinterpret_cast<const f32x4*>(p.points + (kb + fi2) * W1) + half;
    };
    auto chunk_issue = [&](int nt, f32x4 (&q1)[4], f32x4 (&q2)[4], f32x4 (&q3)[4]) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {  // float4 index of channel 32 nt + 8 rq + 4 half
            q1[rq] = fz1[nt * 8 + rq * 2]; q2[rq] = fz2[nt * 8 + rq * 2]; q3[rq] = fz3[nt * 8 + rq * 2];
        }
    };
    auto chunk_blend = [&](f32x16& ax, const f32x4 (&q1)[4], const f32x4 (&q2)[4], const f32x4 (&q3)[4]) {
        // (z1 w1 + z2 w2) + z3 w3, unfused (tf_interpolate.cpp:322-324 order), two channels per instruction (v_pk_mul_f32 /
        // v_pk_add_f32: the same IEEE operations as the scalar forms)
        const f32x2 W1p = {fw1, fw1}, W2p = {fw2, fw2}, W3p = {fw3, fw3};
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 z1 = {q1[rq][2 * h], q1[rq][2 * h + 1]}, z2 = {q2[rq][2 * h], q2[rq][2 * h + 1]};
                const f32x2 z3 = {q3[rq][2 * h], q3[rq][2 * h + 1]};
                const f32x2 cur = (z1 * W1p + z2 * W2p) + z3 * W3p;
                const f32x2 acc2 = f32x2{ax[4 * rq + 2 * h], ax[4 * rq + 2 * h + 1]} + cur;
                ax[4 * rq + 2 * h] = acc2[0]; ax[4 * rq + 2 * h + 1] = acc2[1];
            }
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- first tile: its gather goes out BEFORE the weight staging and lands while the 128 KB of weights are staged
    f32x4 pq1[NT1][4], pq2[NT1][4], pq3[NT1][4];
    if (any_tile) {
        front_issue(g_lo);
        front_weights();
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) chunk_issue(nt, pq1[nt], pq2[nt], pq3[nt]);
    }
    // ---- weight staging: the layout of sa_fused_kernel (PREZ: layer 1 holds the skip-link rows only)
    auto stage4 = [&](float* dst, int count4, const float* __restrict__ src, auto index_of) {
        for (int base = 0; base < count4; base += NTH * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * NTH + tid;
                const int gi = e < count4 ? index_of(e) : -1;
                const f32x4 t = *reinterpret_cast<const f32x4*>(src + (gi >= 0 ? gi : 0));
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                v[u] = gi >= 0 ? t : z;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * NTH + tid;
                if (e < count4) *reinterpret_cast<f32x4*>(dst + e * 4) = v[u];
            }
        }
    };
    auto stage1 = [&](float* dst, int count, const float* __restrict__ src) {
        for (int e = tid; e < count; e += NTH) dst[e] = src[e];
    };
    stage4(wp1, steps1 * 2 * W1 / 4, p.W[0], [&](int e4) {
        const int col = (e4 % (W1 / 4)) * 4, sh = e4 / (W1 / 4);
        const int ch = l1_chan(sh >> 1, sh & 1, 0, true, true, c1);
        return ch >= 0 ? ch * W1 + col : -1;
    });
    stage1(sb1, W1, p.bias[0]);
    // same (k-step, half) row order as sa_fused_kernel; inside a row column 32 nt + l goes to position 4 l + nt
    auto stage4i = [&](float* dst, int W, int count4, const float* __restrict__ src) {
        for (int base = 0; base < count4; base += NTH * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * NTH + tid;
                const int ec = e < count4 ? e : 0;
                const int col = (ec % (W / 4)) * 4, sh = ec / (W / 4);
                const int s = sh >> 1, h = sh & 1;
                v[u] = *reinterpret_cast<const f32x4*>(src + ((s >> 4) * 32 + acc_chan(s & 15, h)) * W + col);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * NTH + tid;
                if (e < count4) {
                    const int col = (e % (W / 4)) * 4, sh = e / (W / 4);
                    float* __restrict__ d = dst + sh * W + (col & 31) * 4 + (col >> 5);
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[4 * j] = v[u][j];
                }
            }
        }
    };
    stage4i(wp2, W2, W1 * W2 / 4, p.W[1]);
    stage1(sb2, W2, p.bias[1]);
    stage4i(wp3, W3, W2 * W3 / 4, p.W[2]);
    stage1(sb3, W3, p.bias[2]);
    __syncthreads();
    PN2_CHAIN_STAMP(1);
    if (!any_tile) return;

    f32x16 a1[NT1];  // layer-1 accumulators of the CURRENT tile during layer 2; rebuilt column tile by column tile for the NEXT
    PN2_CHAIN_STAMP(2);
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) {
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
            if (sp == 0) a1[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1l[sp * 2 * W1 + nt * 32], fsk[sp], zero16, 0, 0, 0);
            else a1[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1l[sp * 2 * W1 + nt * 32], fsk[sp], a1[nt], 0, 0, 0);
        }
        chunk_blend(a1[nt], pq1[nt], pq2[nt], pq3[nt]);
    }

    for (int g = g_lo; g < g_hi; g += g_step) {
        const int gnx = g + g_step;
        const int gn = gnx < g_hi ? gnx : g;  // the last tile rebuilds a1 for itself again: cached loads, a result nobody reads
        f32x16 a2[NT2], a3[NT3];
        f32x4 q1[4], q2[4], q3[4];  // one column chunk of the next tile's z gather in flight
        if (tile_no > 0) PN2_CHAIN_STAMP(2);
        PN2_CHAIN_STAMP(3);
        // side job of MFMA step s (0..63 layer 2, 64..127 layer 3): the next tile's front end.  Chunk c: loads at LS, a1[c] zeroed
        // by the first skip-link MFMA (C = 0) at ZS >= 16 (c + 1) (layer 2 has consumed a1[c]), blend at BS
        auto side = [&](auto s_c) __attribute__((always_inline)) {
            constexpr int s = decltype(s_c)::value;
            constexpr int LS[4] = {13, 38, 63, 88}, ZS[4] = {25, 40, 64, 90}, BS[4] = {37, 62, 87, 112};
            if constexpr (s == 0) front_issue(gn);
            if constexpr (s == 10) front_weights();
            static_for<4>([&](auto c_c) __attribute__((always_inline)) {
                constexpr int c = decltype(c_c)::value;
                if constexpr (s == LS[c]) chunk_issue(c, q1, q2, q3);
                if constexpr (s == ZS[c]) {
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) {
                        if (sp == 0) a1[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1l[sp * 2 * W1 + c * 32], fsk[sp], zero16, 0, 0, 0);
                        else a1[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1l[sp * 2 * W1 + c * 32], fsk[sp], a1[c], 0, 0, 0);
                    }
                }
                if constexpr (s == BS[c]) chunk_blend(a1[c], q1, q2, q3);
            });
        };
        bias_relu_T_pk<NT1>(a1, sb1 + 4 * half);
        {
            f32x4 wq[kPF];
#pragma unroll
            for (int pf = 0; pf < kPF; ++pf) wq[pf] = w2l[pf * 2 * (W2 / 4)];
            static_for<NT1 * 16>([&](auto s_c) __attribute__((always_inline)) {
                constexpr int s = decltype(s_c)::value;
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) {
                    if constexpr (s == 0) a2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[s % kPF][nt], a1[s >> 4][s & 15], zero16, 0, 0, 0);
                    else a2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[s % kPF][nt], a1[s >> 4][s & 15], a2[nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s + kPF < NT1 * 16) wq[s % kPF] = w2l[(s + kPF) * 2 * (W2 / 4)];
                __builtin_amdgcn_sched_barrier(0);
                side(std::integral_constant<int, s>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        PN2_CHAIN_STAMP(4);
        bias_relu_T_pk<NT2>(a2, sb2 + 4 * half);
        {
            f32x4 wq[kPF];
#pragma unroll
            for (int pf = 0; pf < kPF; ++pf) wq[pf] = w3l[pf * 2 * (W3 / 4)];
            static_for<NT2 * 16>([&](auto s_c) __attribute__((always_inline)) {
                constexpr int s = decltype(s_c)::value;
#pragma unroll
                for (int nt = 0; nt < NT3; ++nt) {
                    if constexpr (s == 0) a3[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s >> 4][s & 15], wq[s % kPF][nt], zero16, 0, 0, 0);
                    else a3[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s >> 4][s & 15], wq[s % kPF][nt], a3[nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s + kPF < NT2 * 16) wq[s % kPF] = w3l[(s + kPF) * 2 * (W3 / 4)];
                __builtin_amdgcn_sched_barrier(0);
                side(std::integral_constant<int, NT1 * 16 + s>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        PN2_CHAIN_STAMP(5);
        rows_store_pk<NT3>(a3, sb3, p.out, W3, g * 32, p.rows, half, l31);
        PN2_CHAIN_STAMP(6);
        ++tile_no;
    }
#ifdef PN2_TUNING_HOOKS
    if (p.stats && lane == 0) {  // every wave's whole-kernel cycles, and its XCC / CU (tools/chain_stage_ab.py)
        p.stats[256 + blockIdx.x * NW + wave] = (long long)__builtin_readcyclecounter() - t_begin;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.stats[256 + 1024 + blockIdx.x * NW + wave] = ((long long)xcc << 32) | hwid;
        p.stats[256 + 2048 + blockIdx.x * NW + wave] = (long long)__builtin_amdgcn_s_memrealtime();  // 100 MHz, chip-wide: end
        p.stats[256 + 3072 + blockIdx.x * NW + wave] = t_real0;
    }
    if (p.trace && lane == 0) {  // in-situ spans WITHOUT a profiler: ring of the wave's last 16 launches under this graph's tag
        const int wv = blockIdx.x * NW + wave;
        unsigned long long* cnt = reinterpret_cast<unsigned long long*>(p.trace) + (size_t)p.tag * 1024 + wv;
        const unsigned long long cidx = atomicAdd(cnt, 1ull);  // the wave's own word: uncontended
        long long* r = p.trace + 8 * 1024 + (((size_t)p.tag * 1024 + wv) * 16 + (cidx & 15ull)) * 2;
        r[0] = t_real0;
        r[1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#endif
}

PN2_TUNABLE(int, g_chain_prio, 0)    // tuning hook (pn2_debug_set(13, v)): 1 = stagger the two waves of a SIMD (experiment, see the kernel)
PN2_TUNABLE(long long*, g_chain_stats, nullptr)  // tuning hook: device buffer of 4 x 2 x 4 x 8 cycle stamps
PN2_TUNABLE(int, g_chain_nw, 0)      // tuning hook (pn2_debug_set(7, v)): 16 = 16-wave workgroups for the single-layer kernels
PN2_TUNABLE(int, g_chain_pipe, 1)    // tuning hook (pn2_debug_set(14, v)): 0 = the lockstep schedule for the FP4 chain (A/B)
PN2_TUNABLE(int, g_chain_grid, 256)  // tuning hook (pn2_debug_set(6, v)): persistent workgroups of the 1-per-CU configuration
PN2_TUNABLE(long long*, g_chain_trace, nullptr)  // tuning hook: (8 x 1024) counters + (8 x 1024 x 16 x 2) stamps, see SaFusedParams::trace
PN2_TUNABLE(int, g_chain_tag, 0)     // tuning hook (pn2_debug_set(16, v)): graph tag of the launches captured from now on

template <int L, int NT1, int NT2, int NT3, bool VEC8, bool DENSE, bool POOL, bool INTERP = false, bool PREZ = false>
int launch_chain(const SaFusedParams& p_in, hipStream_t st) {
    SaFusedParams p = p_in;
    p.prio = g_chain_prio;
    if (!p.ldx) p.ldx = 3;
    if (!p.ldp) p.ldp = p.c;
    if (!p.ld1) p.ld1 = p.c1;
#ifdef PN2_TUNING_HOOKS
    p.stats = g_chain_stats;
    p.trace = g_chain_trace;
    p.tag = g_chain_tag & 7;
#endif
    constexpr int W1 = NT1 * 32, W2 = NT2 * 32, W3 = NT3 * 32;
    const int steps1 = l1_steps(PREZ ? 0 : p.c, VEC8, DENSE, INTERP ? p.c1 : 0);
    size_t floats = (size_t)steps1 * 2 * W1 + W1;
    if (L >= 2) floats += (size_t)W1 * W2 + W2;
    if (L >= 3) floats += (size_t)W2 * W3 + W3;
    const size_t bytes = floats * sizeof(float) + 64;  // + the stagger flags
    if (bytes > 150 * 1024) return PN2_EUNSUP;
    const int need4 = (p.groups + 3) / 4;
    if constexpr (L == 1) {
        if (g_chain_nw == 16 && p.groups >= 4096) {  // one 16-wave workgroup per CU: 4 waves/SIMD share the MFMA pipe
            auto kern = sa_fused_kernel<L, NT1, NT2, NT3, VEC8, DENSE, POOL, 16, INTERP, PREZ>;
            static bool attr_set = false;
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return (int)e;
                attr_set = true;
            }
            kern<<<g_chain_grid, 1024, bytes, st>>>(p);
            PN2_RETURN_IF_LAUNCH_FAILED();
            return PN2_OK;
        }
    }
    if constexpr (L == 3 && INTERP && PREZ && DENSE && !POOL && NT1 == 4) {
        // the software-pipelined schedule (bit-identical values): one wave per SIMD, 4 tiles per wave at the FP4 shape
        const bool want_pipe = p.schedule < 0 ? g_chain_pipe != 0 : p.schedule == 1;
        if (p.schedule == 1 && p.c1 > 2 * kPipeSkipSteps) return PN2_EUNSUP;
        if (want_pipe && p.c1 <= 2 * kPipeSkipSteps && (p.groups >= 2048 || p.schedule == 1)) {
            const bool ns2 = p.c1 <= 4;
            auto kern = ns2 ? fp_chain_pipe_kernel<NT2, NT3, 2> : fp_chain_pipe_kernel<NT2, NT3, 4>;
            static bool attr_set[2] = {false, false};
            if (!attr_set[ns2]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return (int)e;
                attr_set[ns2] = true;
            }
            const size_t pbytes = ((size_t)(ns2 ? 2 : 4) * 2 * W1 + W1 + (size_t)W1 * W2 + W2 + (size_t)W2 * W3 + W3) * sizeof(float);
            int grid = g_chain_grid;
            if (grid > need4) grid = need4;
            kern<<<grid, 256, pbytes, st>>>(p);
            PN2_RETURN_IF_LAUNCH_FAILED();
            return PN2_OK;
        }
    }
    if constexpr (L == 3 && INTERP && PREZ) {
#ifdef PN2_FP4_NW12
        constexpr bool nw12 = true;   // A/B build without the tuning hooks' register cost: build.py extra_flags -DPN2_FP4_NW12
#else
        const bool nw12 = g_chain_nw == 12;
#endif
        // three waves per SIMD (168 registers, 12 B of scratch): 128-131 us against 101 for two -- measured twice, with and
        // without spills; not used
        if (nw12 && bytes > 78 * 1024 && p.groups >= 2048) {
            auto kern = sa_fused_kernel<L, NT1, NT2, NT3, VEC8, DENSE, POOL, 12, INTERP, PREZ>;
            static bool attr_set = false;
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return (int)e;
                attr_set = true;
            }
            int grid = g_chain_grid;
            const int need12 = (p.groups + 11) / 12;
            if (grid > need12) grid = need12;
            kern<<<grid, 768, bytes, st>>>(p);
            PN2_RETURN_IF_LAUNCH_FAILED();
            return PN2_OK;
        }
    }
    if (bytes > 78 * 1024 && p.groups >= 2048 && g_chain_nw != 4) {  // g_chain_nw == 4 (tuning): one wave per SIMD
        // only one workgroup fits per CU: give it 8 waves (2 per SIMD) sharing the LDS weights.
        // (With fewer than 2048 tiles, 4-wave workgroups spread the tiles over twice as many CUs.)
        auto kern = sa_fused_kernel<L, NT1, NT2, NT3, VEC8, DENSE, POOL, 8, INTERP, PREZ>;
        static bool attr_set = false;  // per instantiation; benign race (idempotent call)
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        int grid = g_chain_grid;
        const int need8 = (p.groups + 7) / 8;
        if (grid > need8) grid = need8;
        kern<<<grid, 512, bytes, st>>>(p);
    } else {
        auto kern = sa_fused_kernel<L, NT1, NT2, NT3, VEC8, DENSE, POOL, 4, INTERP, PREZ>;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        int grid = bytes > 78 * 1024 ? g_chain_grid : g_chain_grid * 2;  // 4-wave workgroups: as many as LDS lets co-reside per CU
        if (grid > need4) grid = need4;
        kern<<<grid, 256, bytes, st>>>(p);
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

}  // namespace

#ifdef PN2_TUNING_HOOKS
extern "C" int pn2_debug_set_chain_stats(long long* dev_ptr) { g_chain_stats = dev_ptr; return 0; }
extern "C" int pn2_debug_set_chain_trace(long long* dev_ptr) { g_chain_trace = dev_ptr; return 0; }
extern "C" int pn2_debug_set_fused(int what, int value) {
    if (what == 6) { g_chain_grid = value; return 0; }
    if (what == 7) { g_chain_nw = value; return 0; }
    if (what == 13) { g_chain_prio = value; return 0; }
    if (what == 14) { g_chain_pipe = value; return 0; }
    if (what == 16) { g_chain_tag = value; return 0; }
    return PN2_EINVAL;
}
#endif  // PN2_TUNING_HOOKS

static int sa_fused_impl(int b, int n, int m, int nsample, int c, const float* xyz,
                         const float* new_xyz, const float* points, const int* idx,
                         int nlayers, const int* widths, const float* const* w,
                         const float* const* bias, float* out, bool pool, void* stream, int ld_xyz, int ld_points);

extern "C" int pn2_sa_mlp_max_fused(int b, int n, int m, int nsample, int c, const float* xyz,
                                    const float* new_xyz, const float* points, const int* idx,
                                    int nlayers, const int* widths, const float* const* w,
                                    const float* const* bias, float* out, void* stream) {
    return sa_fused_impl(b, n, m, nsample, c, xyz, new_xyz, points, idx, nlayers, widths, w, bias, out, true, stream, 3, c);
}

// pn2_sa_mlp_max_fused with the rows of xyz / points ld_xyz / ld_points floats apart (>= 3 / >= c): the xyz and rgb columns
// of a (b,n,6) batch gathered in place (model.py:26-29 slices them out of the input tensor; here no copy is made).
// ld_points != c needs the un-vectorised feature path (c % 8 != 0): PN2_EUNSUP otherwise.  Same bits as the dense call.
extern "C" int pn2_sa_mlp_max_fused_ld(int b, int n, int m, int nsample, int c, const float* xyz, int ld_xyz,
                                       const float* new_xyz, const float* points, int ld_points, const int* idx,
                                       int nlayers, const int* widths, const float* const* w,
                                       const float* const* bias, float* out, void* stream) {
    return sa_fused_impl(b, n, m, nsample, c, xyz, new_xyz, points, idx, nlayers, widths, w, bias, out, true, stream, ld_xyz,
                         c > 0 ? ld_points : 0);
}

// Same gather + MLP chain WITHOUT the max over the neighbours: out is (b, m, nsample, widths[last])
// with ReLU applied -- the input of a wider following layer that runs on pn2_linear (+pool).  Used
// when an SA stack starts with <= 128-wide layers and ends with a wider one ([128,128,256]).
extern "C" int pn2_sa_mlp_rows_fused(int b, int n, int m, int nsample, int c, const float* xyz,
                                     const float* new_xyz, const float* points, const int* idx,
                                     int nlayers, const int* widths, const float* const* w,
                                     const float* const* bias, float* out, void* stream) {
    return sa_fused_impl(b, n, m, nsample, c, xyz, new_xyz, points, idx, nlayers, widths, w, bias, out, false, stream, 3, c);
}

static int sa_fused_impl(int b, int n, int m, int nsample, int c, const float* xyz,
                         const float* new_xyz, const float* points, const int* idx,
                         int nlayers, const int* widths, const float* const* w,
                         const float* const* bias, float* out, bool pool, void* stream, int ld_xyz, int ld_points) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || c < 0 || nlayers <= 0) return PN2_EINVAL;
    if (!xyz || !new_xyz || !idx || !widths || !w || !bias || !out || (c > 0 && !points)) return PN2_ENULL;
    if (nlayers > 3) return PN2_EUNSUP;
    // K = 32 is the native tile; 16 (two centres per tile) and 64, 128, 256 (several tiles per centre) reuse it
    int kshift = 0;
    while ((1 << kshift) < nsample) ++kshift;
    if ((1 << kshift) != nsample || kshift < 4 || kshift > 8) return PN2_EUNSUP;
    if (kshift != 5 && !pool) return PN2_EUNSUP;
    if (kshift == 4 && (((long long)b * m) & 1)) return PN2_EUNSUP;
    if ((long long)b * m * nsample > 0x7fffffffLL - 64) return PN2_ERANGE;
    SaFusedParams p{};
    p.schedule = -1;
    p.n = n; p.m = m; p.c = c; p.kshift = kshift;
    p.groups = (int)(((long long)b * m * nsample) / 32); p.rows = p.groups * 32;
    p.xyz = xyz; p.new_xyz = new_xyz; p.points = points; p.idx = idx; p.out = out;
    int nt[3] = {0, 0, 0};
    for (int l = 0; l < nlayers; ++l) {
        if (widths[l] <= 0 || widths[l] % 32 != 0 || widths[l] > 128) return PN2_EUNSUP;
        if (!w[l] || !bias[l]) return PN2_ENULL;
        if ((uintptr_t)w[l] % 16 != 0) return PN2_EUNSUP;  // 16-byte weight staging
        p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l];
        nt[l] = widths[l] / 32;
    }
    const bool vec8 = c > 0 && (c % 8 == 0) && ((uintptr_t)points % 16 == 0);
    if (ld_xyz < 3 || (c > 0 && ld_points < c)) return PN2_EINVAL;
    if (vec8 && ld_points != c) return PN2_EUNSUP;  // the 16-byte feature gathers read dense rows
    p.ldx = ld_xyz; p.ldp = ld_points;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (kshift > 5) {  // tiles of one centre merge through atomicMax: start from +0
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)b * m * widths[nlayers - 1], st);
        if (e != hipSuccess) return (int)e;
    }
    const int key = nlayers * 1000 + nt[0] * 100 + nt[1] * 10 + nt[2];
    if (!pool) {
        // un-pooled variant: only the shapes the model needs ([*,128,128] prefixes)
        if (key == 2440) return vec8 ? launch_chain<2, 4, 4, 1, true, false, false>(p, st)
                                     : launch_chain<2, 4, 4, 1, false, false, false>(p, st);
        if (key == 1400) return vec8 ? launch_chain<1, 4, 1, 1, true, false, false>(p, st)
                                     : launch_chain<1, 4, 1, 1, false, false, false>(p, st);
        return PN2_EUNSUP;
    }
#define PN2_SA_CASE(L_, A_, B_, C_)                                                     \
    case (L_ * 1000 + A_ * 100 + B_ * 10 + C_):                                         \
        return vec8 ? launch_chain<L_, A_, (B_ ? B_ : 1), (C_ ? C_ : 1), true, false, true>(p, st) \
                    : launch_chain<L_, A_, (B_ ? B_ : 1), (C_ ? C_ : 1), false, false, true>(p, st);
    switch (key) {
        PN2_SA_CASE(3, 1, 1, 2)  // SA1 of semantic.json: [32,32,64]
        PN2_SA_CASE(3, 2, 2, 4)  // SA2: [64,64,128]
        PN2_SA_CASE(3, 2, 3, 4)  // MSG scale [64,96,128]
        PN2_SA_CASE(3, 1, 1, 1)
        PN2_SA_CASE(3, 2, 2, 2)
        PN2_SA_CASE(2, 2, 4, 0)
        PN2_SA_CASE(2, 4, 4, 0)
        PN2_SA_CASE(1, 1, 0, 0)
        PN2_SA_CASE(1, 2, 0, 0)
        PN2_SA_CASE(1, 4, 0, 0)  // north-star shape: one 128 -> 128 layer + max
        default: return PN2_EUNSUP;
    }
#undef PN2_SA_CASE
}

// Dense-row MLP chain (feature-propagation layers, pointnet_util.py:312-325 with inference BN
// folded): y = relu(...relu(x @ W0 + b0)... @ W_{L-1} + b_{L-1}), optionally max-pooled over each
// consecutive group of 32 rows.  Same kernel as the fused SA MLP with the gather switched off: the
// weights of all layers stay resident in LDS and activations never leave registers between layers.
//   x (rows, cin) row-major;  y (rows, w_last) or (rows/32, w_last) when pool == 32.
// Constraints: 1 <= nlayers <= 2 here (3 x 128-wide layers do not fit LDS), widths multiples of 32,
// <= 128, all weights must fit ~150 KB of LDS; PN2_EUNSUP otherwise (callers fall back to pn2_linear).
extern "C" int pn2_mlp_chain(int rows, int cin, const float* x, int nlayers, const int* widths,
                             const float* const* w, const float* const* bias, int pool, float* y,
                             void* stream) {
    if (rows <= 0 || cin <= 0 || nlayers <= 0) return PN2_EINVAL;
    if (!x || !widths || !w || !bias || !y) return PN2_ENULL;
    if (nlayers > 2 || (pool != 0 && pool != 32)) return PN2_EUNSUP;
    if (pool == 32 && rows % 32 != 0) return PN2_EINVAL;
    SaFusedParams p{};
    p.schedule = -1;
    p.c = cin; p.rows = rows; p.groups = (rows + 31) / 32;
    p.points = x; p.out = y;
    int nt[2] = {0, 0};
    for (int l = 0; l < nlayers; ++l) {
        if (widths[l] <= 0 || widths[l] % 32 != 0 || widths[l] > 128) return PN2_EUNSUP;
        if (!w[l] || !bias[l]) return PN2_ENULL;
        if ((uintptr_t)w[l] % 16 != 0) return PN2_EUNSUP;  // 16-byte weight staging
        p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l];
        nt[l] = widths[l] / 32;
    }
    const bool vec8 = (cin % 8 == 0) && ((uintptr_t)x % 16 == 0);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int key = nlayers * 100 + nt[0] * 10 + nt[1];
#define PN2_CHAIN_CASE(L_, A_, B_)                                                                   \
    case (L_ * 100 + A_ * 10 + B_):                                                                  \
        if (pool) return vec8 ? launch_chain<L_, A_, (B_ ? B_ : 1), 1, true, true, true>(p, st)      \
                              : launch_chain<L_, A_, (B_ ? B_ : 1), 1, false, true, true>(p, st);    \
        return vec8 ? launch_chain<L_, A_, (B_ ? B_ : 1), 1, true, true, false>(p, st)               \
                    : launch_chain<L_, A_, (B_ ? B_ : 1), 1, false, true, false>(p, st);
    switch (key) {
        PN2_CHAIN_CASE(1, 4, 0)
        PN2_CHAIN_CASE(2, 4, 4)
        PN2_CHAIN_CASE(1, 2, 0)
        PN2_CHAIN_CASE(2, 2, 4)
        default: return PN2_EUNSUP;
    }
#undef PN2_CHAIN_CASE
}

// Fused feature-propagation block (pointnet_util.py:300-325, inference BN folded): the FP front end
// (inverse-distance weights + three_interpolate + concat [interp | points1]) feeds the first MFMA layer
// directly from L2 -- the (b, n, c2+c1) concatenated tensor is never written to HBM -- followed by up to
// two LDS-resident dense layers (+bias, ReLU).
//   dist, idx (b,n,3) from pn2_three_nn;  points2 (b,m,c2) known features;  points1 (b,n,c1) or NULL
//   w[0] ((c2+c1) rows or more, widths[0]) with the interpolated channels first;  y (b*n, widths[last])
// Constraints: c2 % 8 == 0, points2 16-byte aligned, nlayers <= 2, widths multiples of 32 and <= 128;
// PN2_EUNSUP otherwise (callers fall back to pn2_fp_interp_concat + pn2_mlp_chain / pn2_linear).
extern "C" int pn2_fp_mlp_fused(int b, int n, int m, int c1, int c2, const float* dist, const int* idx,
                                const float* points1, const float* points2, int nlayers, const int* widths,
                                const float* const* w, const float* const* bias, float* y, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c2 <= 0 || c1 < 0 || nlayers <= 0) return PN2_EINVAL;
    if (!dist || !idx || !points2 || !widths || !w || !bias || !y || (c1 > 0 && !points1)) return PN2_ENULL;
    if ((long long)b * n > 0x7fffffffLL - 32) return PN2_ERANGE;
    if (nlayers > 2 || c2 % 8 != 0 || (uintptr_t)points2 % 16 != 0) return PN2_EUNSUP;
    SaFusedParams p{};
    p.schedule = -1;
    p.n = n; p.m = m; p.c = c2; p.c1 = c1; p.rows = b * n; p.groups = (b * n + 31) / 32;
    p.points = points2; p.points1 = points1; p.dist = dist; p.idx = idx; p.out = y;
    int nt[2] = {0, 0};
    for (int l = 0; l < nlayers; ++l) {
        if (widths[l] <= 0 || widths[l] % 32 != 0 || widths[l] > 128) return PN2_EUNSUP;
        if (!w[l] || !bias[l]) return PN2_ENULL;
        if ((uintptr_t)w[l] % 16 != 0) return PN2_EUNSUP;
        p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l];
        nt[l] = widths[l] / 32;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (nlayers * 100 + nt[0] * 10 + nt[1]) {
        case 244: return launch_chain<2, 4, 4, 1, true, true, false, true>(p, st);
        case 140: return launch_chain<1, 4, 1, 1, true, true, false, true>(p, st);
        case 224: return launch_chain<2, 2, 4, 1, true, true, false, true>(p, st);
        case 120: return launch_chain<1, 2, 1, 1, true, true, false, true>(p, st);
        default: return PN2_EUNSUP;
    }
}

// pn2_fp_mlp_fused with the first layer's product HOISTED (see PREZ above): z = points2 @ W1[:c2] (b*m rows, widths[0]
// wide; the caller computes it with pn2_linear, no bias, no activation -- 1/8 of the rows of the FP level) replaces
// points2, w[0] holds only the c1 skip-link rows of W1 (c1 x widths[0]; may be NULL when c1 == 0).  nlayers counts ALL
// layers (2 or 3); the chain after the front end is LDS-resident as in pn2_fp_mlp_fused.
// Values: interp(points2) @ W1a == interp(points2 @ W1a) exactly in real arithmetic; in fp32 the two orders differ by
// rounding only (1e-7 of the activation scale; tests/test_layers_gpu.py holds both to the fp64 oracle at 1e-5).
static int fp_mlp_fused_pre_impl(int b, int n, int m, int c1, const float* dist, const int* idx, const float* points1,
                                 const float* z, int nlayers, const int* widths, const float* const* w,
                                 const float* const* bias, float* y, int schedule, void* stream, int ld_points1 = 0);

extern "C" int pn2_fp_mlp_fused_pre(int b, int n, int m, int c1, const float* dist, const int* idx, const float* points1,
                                    const float* z, int nlayers, const int* widths, const float* const* w,
                                    const float* const* bias, float* y, void* stream) {
    return fp_mlp_fused_pre_impl(b, n, m, c1, dist, idx, points1, z, nlayers, widths, w, bias, y, -1, stream);
}

// pn2_fp_mlp_fused_pre with the skip-link rows ld_points1 floats apart (>= c1): the rgb columns of a (b,n,6) batch read in
// place (model.py:26-29, 121-129).  Same bits as the dense call.
extern "C" int pn2_fp_mlp_fused_pre_ld(int b, int n, int m, int c1, const float* dist, const int* idx, const float* points1,
                                       int ld_points1, const float* z, int nlayers, const int* widths, const float* const* w,
                                       const float* const* bias, float* y, void* stream) {
    return fp_mlp_fused_pre_impl(b, n, m, c1, dist, idx, points1, z, nlayers, widths, w, bias, y, -1, stream, ld_points1);
}

// The same call with the kernel SCHEDULE named by the caller (stateless door for the parity tests and A/B timing):
// 0 = the lockstep kernel (8 waves per workgroup: gather, then MFMA layers, then store), 1 = the software-pipelined kernel
// (one wave per SIMD builds the next tile's first-layer accumulator between the MFMA groups of the current tile; three
// 128-wide layers, c1 <= 8: PN2_EUNSUP otherwise).  Both produce the same bits.
extern "C" int pn2_fp_mlp_fused_pre_schedule(int b, int n, int m, int c1, const float* dist, const int* idx,
                                             const float* points1, const float* z, int nlayers, const int* widths,
                                             const float* const* w, const float* const* bias, float* y, int schedule,
                                             void* stream) {
    if (schedule != 0 && schedule != 1) return PN2_EINVAL;
    return fp_mlp_fused_pre_impl(b, n, m, c1, dist, idx, points1, z, nlayers, widths, w, bias, y, schedule, stream);
}

static int fp_mlp_fused_pre_impl(int b, int n, int m, int c1, const float* dist, const int* idx, const float* points1,
                                 const float* z, int nlayers, const int* widths, const float* const* w,
                                 const float* const* bias, float* y, int schedule, void* stream, int ld_points1) {
    if (ld_points1 != 0 && ld_points1 < c1) return PN2_EINVAL;
    if (b <= 0 || n <= 0 || m <= 0 || c1 < 0 || nlayers <= 0) return PN2_EINVAL;
    if (!dist || !idx || !z || !widths || !w || !bias || !y || (c1 > 0 && (!points1 || !w[0]))) return PN2_ENULL;
    if ((long long)b * n > 0x7fffffffLL - 32) return PN2_ERANGE;
    if (nlayers < 2 || nlayers > 3 || (uintptr_t)z % 16 != 0) return PN2_EUNSUP;
    SaFusedParams p{};
    p.schedule = schedule;
    p.n = n; p.m = m; p.c = widths[0]; p.c1 = c1; p.rows = b * n; p.groups = (b * n + 31) / 32;
    p.points = z; p.points1 = points1; p.dist = dist; p.idx = idx; p.out = y; p.ld1 = ld_points1;
    int nt[3] = {0, 0, 0};
    for (int l = 0; l < nlayers; ++l) {
        if (widths[l] <= 0 || widths[l] % 32 != 0 || widths[l] > 128) return PN2_EUNSUP;
        if (!bias[l] || (l > 0 && !w[l])) return PN2_ENULL;
        if (w[l] && (uintptr_t)w[l] % 16 != 0) return PN2_EUNSUP;
        p.w[l] = widths[l]; p.W[l] = w[l] ? w[l] : bias[l] /* never read: c1 == 0 */; p.bias[l] = bias[l];
        nt[l] = widths[l] / 32;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (nlayers * 1000 + nt[0] * 100 + nt[1] * 10 + nt[2]) {
        case 3444: return launch_chain<3, 4, 4, 4, true, true, false, true, true>(p, st);  // FP4 of semantic.json: 131 -> 128 -> 128 -> 128
        case 2440: return schedule == 1 ? PN2_EUNSUP : launch_chain<2, 4, 4, 1, true, true, false, true, true>(p, st);
        default: return PN2_EUNSUP;
    }
}

// pn2_sa_mlp_max_fused / pn2_sa_mlp_rows_fused with the FEATURE part of the first layer hoisted by linearity:
// zf = points @ W1[3:] (b*n rows, widths[0] wide: one per source point, computed by the caller with pn2_linear) replaces
// `points`; w[0] = the 3 xyz rows of the folded first-layer weight (3 x widths[0]).  pool != 0: max over the K neighbours
// (nsample = 32 only here), else the un-pooled (b, m, nsample, w_last) rows.  Same products, summed in a different order.
extern "C" int pn2_sa_mlp_fused_pre(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const float* zf,
                                    const int* idx, int nlayers, const int* widths, const float* const* w,
                                    const float* const* bias, int pool, float* out, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || nlayers <= 0) return PN2_EINVAL;
    if (!xyz || !new_xyz || !zf || !idx || !widths || !w || !bias || !out) return PN2_ENULL;
    if (nlayers > 3 || nsample != 32 || (uintptr_t)zf % 16 != 0) return PN2_EUNSUP;
    if ((long long)b * m * nsample > 0x7fffffffLL - 64) return PN2_ERANGE;
    SaFusedParams p{};
    p.schedule = -1;
    p.n = n; p.m = m; p.c = widths[0]; p.kshift = 5;
    p.groups = (int)(((long long)b * m * nsample) / 32); p.rows = p.groups * 32;
    p.xyz = xyz; p.new_xyz = new_xyz; p.points = zf; p.idx = idx; p.out = out;
    int nt[3] = {0, 0, 0};
    for (int l = 0; l < nlayers; ++l) {
        if (widths[l] <= 0 || widths[l] % 32 != 0 || widths[l] > 128) return PN2_EUNSUP;
        if (!w[l] || !bias[l]) return PN2_ENULL;
        if ((uintptr_t)w[l] % 16 != 0) return PN2_EUNSUP;
        p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l];
        nt[l] = widths[l] / 32;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int key = nlayers * 1000 + nt[0] * 100 + nt[1] * 10 + nt[2];
    if (pool) {
        if (key == 3224) return launch_chain<3, 2, 2, 4, true, false, true, false, true>(p, st);  // SA2 of semantic.json: [64,64,128]
        if (key == 3444) return launch_chain<3, 4, 4, 4, true, false, true, false, true>(p, st);
        return PN2_EUNSUP;
    }
    if (key == 2440) return launch_chain<2, 4, 4, 1, true, false, false, false, true>(p, st);      // SA3: [128,128 | 256 on pn2_linear]
    return PN2_EUNSUP;
}
