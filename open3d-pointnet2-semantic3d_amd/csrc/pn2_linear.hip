// pn2_linear.hip -- one dense layer of the PointNet++ shared MLP on the fp32
// matrix cores of gfx950, plus the grouped-input builder used by the unfused path.
//
// The reference runs each 1x1 conv as tf.nn.conv2d + bias_add + batch_norm + relu
// (util/tf_util.py:181-203) and the K-pool as a separate tf.reduce_max
// (util/pointnet_util.py:167-170), each re-reading the (B,M,K,C) tensor from HBM.
// Here: y = relu(x @ W + b) with the inference BatchNorm folded into (W, b) by the
// host and the max over the K neighbours taken in the MFMA epilogue.
//
// Kernel shape: 256 threads = 4 waves arranged WM x WN; block tile (32*WM) rows x (32*NT*WN)
// cols; each wave owns 32 rows x 32*NT cols = NT accumulators of v_mfma_f32_32x32x2_f32 (exact
// fp32 products, fp32 accumulate: bitwise an fmaf chain).  K is tiled by 32 through LDS with a
// global->register prefetch of tile t+1 under the MFMAs of tile t.
//   * A tile: row-major [rows][32(+4 pad)] written with 16-byte stores straight from the 16-byte
//     global loads (no transpose).  A lane (row = lane&31, half = lane>>5) reads FOUR k values
//     with one ds_read_b128: k = 8t + 4*half + {0..3}.  The 32x32x2 MFMA only requires the A and
//     B operands of a step to carry the SAME two k indices (half 0 / half 1), so visiting the
//     contraction index in this permuted order is exact; row stride 36 floats makes the b128
//     reads bank-conflict free.
//   * B tile: k-major [32][BN(+4)], read one float per lane per step (lane -> consecutive cols).
// Because a wave's 32 rows are exactly one K=32 neighbourhood, the max-pool is an in-register max
// over the 16 accumulator rows + one cross-half exchange.
#include <type_traits>

#include "pn2_common.h"
#include "pn2_mfma_stats.h"
#include "pn2_fwd_narrow.h"
#include "pn2_dgrad_wide.h"

namespace {

template <int U, int N, class F>
__device__ __forceinline__ void static_for(F& f) {
    if constexpr (U < N) {
        f(std::integral_constant<int, U>{});
        static_for<U + 1, N>(f);
    }
}

constexpr int kBK = 32;
constexpr int kAS = kBK + 4;  // A tile row stride (floats): 144 B, keeps b128 accesses 16-B aligned

__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
    // v >= 0 (post-ReLU): integer order == float order
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
}

// WK = 2 splits every k-tile between two waves of the same output tile (intra-workgroup split-K,
// reduced through LDS before the epilogue): small-row layers then launch twice as many workgroups.
// ST = register prefetch depth: the global loads of k-tile kt+ST-1 are issued under the MFMAs of tile
// kt (ST = 2: one tile ahead; default 3, measured best: profiles/r01_linear_prefetch_depth.txt).
// TB = true: the B operand is read TRANSPOSED, B(k, n) = w[n * cin + k] with `cout` REAL output columns (any value; the
// tile is guarded) -- the data gradient of a dense layer, dx (rows, n_in) = dy (rows, n_out) . W^T with W (n_in, n_out)
// row-major: here cin = n_out is the contraction and cout = n_in the output width.
// XF = true (training, VEC_A): the A operand is the PRE-normalisation output of the layer below; its batch norm (+ReLU) is
// applied per input channel k while the tile moves from the prefetch registers to LDS, a = relu?(fma(x, xf.scale[k],
// xf.shift[k])) -- the normalised activation is never written (pn2_bn_relu_forward_deferred).
// GX = 1 / 2 (training, VEC_A, TB): the A operand is the gradient dy LEAVING the batch norm (+ReLU; GX = 2: + max over groups of
// 32 rows) of the layer whose data gradient this is, formed from (y, dz) while the tile moves from the prefetch registers to LDS
// (Pn2GradOnLoad; x is not read).  The per-channel constants -- and for GX = 2 the pooled gradient / maxima / tie counts of the
// block's BM / 32 groups -- are staged once per block in dynamic LDS.
template <int WM, int WN, int NT, bool VEC_A, int WK = 1, int ST = 2, bool TB = false, bool XF = false, int GX = 0>
__global__ void __launch_bounds__(256)
linear_kernel(int rows, int cin, int cout, const float* __restrict__ x,
              const float* __restrict__ w, const float* __restrict__ bias, int relu, int pool,
              float* __restrict__ y, double* __restrict__ stats = nullptr, Pn2BnGradEpilogue gepi = Pn2BnGradEpilogue{},
              Pn2LoadTransform xf = Pn2LoadTransform{}, Pn2GradOnLoad gx = Pn2GradOnLoad{}, Pn2BnFinish fin = Pn2BnFinish{}) {
    static_assert(!XF || (VEC_A && !TB && WK == 1), "the load transform exists for the forward GEMM with 16-byte A loads");
    static_assert(GX == 0 || (VEC_A && TB && WK == 1 && !XF), "the gradient-on-load operand exists for the data gradient with 16-byte A loads");
    static_assert(WM * WN * WK == 4, "4 waves per block");
    static_assert(WK == 1 || (WK == 2 && NT * 16 * 64 * WM * WN <= kBK * (32 * NT * WN + 4)), "reduction buffer must fit the B tile");
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * NT * WN;
    constexpr int BS = BN + 4;                   // B tile row stride (floats)
    constexpr int A_F4 = BM * kBK / 4;           // float4 per A tile
    constexpr int A_PER_T = (A_F4 + 255) / 256;
    constexpr int A_SC = (BM * kBK + 255) / 256; // scalars per thread (non-vector path)
    constexpr int B_F4 = kBK * BN / 4;
    constexpr int B_PER_T = (B_F4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float As[BM * kAS];
    __shared__ __attribute__((aligned(16))) float Bs[kBK * BS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wk = wave / (WM * WN);           // k-split index (0 when WK == 1)
    const int wmn = wave % (WM * WN);
    const int wm = wmn / WN, wn = wmn % WN;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int row0 = blockIdx.x * BM;
    const int col0 = blockIdx.y * BN;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // GX: coef (6, cin) | GX = 2: dz / zmax / ties of the block's BM / 32 groups, each (BM / 32, cin)
    extern __shared__ __attribute__((aligned(16))) float gx_lds[];
    if constexpr (GX != 0) {
        for (int e = tid; e < 6 * cin; e += 256) gx_lds[e] = gx.coef[e];
        if constexpr (GX == 2) {
            constexpr int NG = BM / 32;
            const int groups = rows / 32;
            for (int e = tid; e < NG * cin; e += 256) {
                const int g = e / cin, ch = e - g * cin;
                const int gg = row0 / 32 + g < groups ? row0 / 32 + g : groups - 1;
                gx_lds[6 * cin + e] = gx.dz[(size_t)gg * cin + ch];
                gx_lds[6 * cin + NG * cin + e] = gx.zmax[(size_t)gg * cin + ch];
                gx_lds[6 * cin + 2 * NG * cin + e] = gx.ties[(size_t)gg * cin + ch];
            }
        }
    }

    f32x4 a_gs[GX == 1 ? ST : 1][GX == 1 ? A_PER_T : 1];  // GX = 1: the dz tile beside the y tile
    f32x4 a_vs[ST][VEC_A ? A_PER_T : 1];
    float a_ss[ST][VEC_A ? 1 : A_SC];
    f32x4 b_vs[ST][B_PER_T];
    f32x4 x_sc[XF ? ST : 1], x_sh[XF ? ST : 1];  // XF: (scale, shift) of this thread's four k of the tile (k4 = tid & 7 for every i)

    auto load_tile = [&](int kt, f32x4 (&a_v)[VEC_A ? A_PER_T : 1], float (&a_s)[VEC_A ? 1 : A_SC],
                         f32x4 (&b_v)[B_PER_T], f32x4& xsc, f32x4& xsh,
                         f32x4 (&a_g)[GX == 1 ? A_PER_T : 1]) __attribute__((always_inline)) {
        // Every load is unconditional (addresses clamped into the buffers, out-of-range elements zeroed
        // afterwards): straight-line loads let the compiler use counted s_waitcnt vmcnt(N), which is what
        // keeps ST-1 tiles in flight; loads under a divergent branch force vmcnt(0).
        const int k0 = kt * kBK;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (VEC_A) {
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                const int f = tid + 256 * i;
                const int r = f >> 3, k4 = f & 7;
                const int gr = row0 + r, gk = k0 + k4 * 4;
                const int grc = gr < rows ? gr : rows - 1;
                const int gkc = gk < cin ? gk : cin - 4;  // VEC_A: cin % 4 == 0
                if constexpr (GX != 0) {  // transformed, then zeroed, in store_tile
                    a_v[i] = *reinterpret_cast<const f32x4*>(gx.y + (size_t)grc * cin + gkc);
                    if constexpr (GX == 1) a_g[i] = *reinterpret_cast<const f32x4*>(gx.dz + (size_t)grc * cin + gkc);
                } else {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)grc * cin + gkc);
                    if constexpr (XF) a_v[i] = v;  // transformed, then zeroed, in store_tile
                    else a_v[i] = (gr < rows && gk < cin) ? v : z4;
                }
            }
            if constexpr (XF) {
                const int gk = k0 + (tid & 7) * 4;
                const int gkc = gk < cin ? gk : cin - 4;
                xsc = *reinterpret_cast<const f32x4*>(xf.scale + gkc);
                xsh = *reinterpret_cast<const f32x4*>(xf.shift + gkc);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_SC; ++i) {
                const int e = tid + 256 * i;
                const int k = e & 31, r = e >> 5;
                const int gr = row0 + r, gk = k0 + k;
                const int grc = gr < rows ? gr : rows - 1;
                const int gkc = gk < cin ? gk : cin - 1;
                const float v = x[(size_t)grc * cin + gkc];
                a_s[i] = (r < BM && gr < rows && gk < cin) ? v : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            const int f = tid + 256 * i;
            if constexpr (TB) {
                // thread -> (n = f / 8, four consecutive k): a 16-byte (or 4 scalar) read along a row of W
                const int n = f >> 3, k4 = f & 7;
                const int gn = col0 + n, gk = k0 + k4 * 4;
                const int gnc = gn < cout ? gn : cout - 1;
                f32x4 v;
                if ((cin & 3) == 0) {
                    const int gkc = gk < cin ? gk : cin - 4;
                    v = *reinterpret_cast<const f32x4*>(w + (size_t)gnc * cin + gkc);
                    if (!(gn < cout && gk < cin)) v = z4;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gkq = gk + q < cin ? gk + q : cin - 1;
                        const float e = w[(size_t)gnc * cin + gkq];
                        v[q] = (gn < cout && gk + q < cin) ? e : 0.f;
                    }
                }
                b_v[i] = v;
            } else {
                const int k = f / (BN / 4), n4 = f % (BN / 4);
                const int kc = k0 + k < cin ? k0 + k : cin - 1;
                const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)kc * cout + col0 + n4 * 4);
                b_v[i] = (k0 + k < cin) ? v : z4;
            }
        }
    };
    auto store_tile = [&](int kt, const f32x4 (&a_v)[VEC_A ? A_PER_T : 1], const float (&a_s)[VEC_A ? 1 : A_SC],
                          const f32x4 (&b_v)[B_PER_T], const f32x4& xsc, const f32x4& xsh,
                          const f32x4 (&a_g)[GX == 1 ? A_PER_T : 1]) __attribute__((always_inline)) {
        if constexpr (VEC_A) {
            f32x4 gc[GX != 0 ? 6 : 1];  // GX: this thread's four channels of the tile (k4 = tid & 7 for every i)
            int gch = 0;
            if constexpr (GX != 0) {
                const int gk = kt * kBK + (tid & 7) * 4;
                gch = gk < cin ? gk : cin - 4;
#pragma unroll
                for (int j = 0; j < 6; ++j) gc[j] = *reinterpret_cast<const f32x4*>(gx_lds + j * cin + gch);
            }
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                const int f = tid + 256 * i;
                const int r = f >> 3, k4 = f & 7;
                f32x4 v = a_v[i];
                if constexpr (GX != 0) {
                    const bool live = row0 + r < rows && kt * kBK + k4 * 4 < cin;
                    f32x4 pd, pm, pn;
                    if constexpr (GX == 2) {
                        constexpr int NG = BM / 32;
                        const float* pg = gx_lds + 6 * cin + (r >> 5) * cin + gch;
                        pd = *reinterpret_cast<const f32x4*>(pg);
                        pm = *reinterpret_cast<const f32x4*>(pg + NG * cin);
                        pn = *reinterpret_cast<const f32x4*>(pg + 2 * NG * cin);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t;
                        if constexpr (GX == 2)
                            t = pn2_bn_grad_element_pooled(v[q], pd[q], pm[q], pn[q], gc[0][q], gc[1][q], gc[2][q], gc[3][q],
                                                           gc[4][q], gc[5][q], gx.relu);
                        else
                            t = pn2_bn_grad_element(v[q], a_g[i][q], gc[0][q], gc[1][q], gc[2][q], gc[3][q], gc[4][q], gc[5][q],
                                                    gx.relu);
                        v[q] = live ? t : 0.f;
                    }
                }
                if constexpr (XF) {
                    const bool live = row0 + r < rows && kt * kBK + k4 * 4 < cin;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = __builtin_fmaf(v[q], xsc[q], xsh[q]);
                        t = xf.relu ? fmaxf(t, 0.f) : t;
                        v[q] = live ? t : 0.f;
                    }
                }
                if (f < A_F4) *reinterpret_cast<f32x4*>(As + r * kAS + k4 * 4) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_SC; ++i) {
                const int e = tid + 256 * i;
                const int k = e & 31, r = e >> 5;
                if (r < BM) As[r * kAS + k] = a_s[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            const int f = tid + 256 * i;
            if constexpr (TB) {
                const int n = f >> 3, k4 = f & 7;
                if (f < B_F4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) Bs[(4 * k4 + q) * BS + n] = b_v[i][q];
                }
            } else {
                const int k = f / (BN / 4), n4 = f % (BN / 4);
                if (f < B_F4) *reinterpret_cast<f32x4*>(Bs + k * BS + n4 * 4) = b_v[i];
            }
        }
    };

    const int nkt = (cin + kBK - 1) / kBK;
#pragma unroll
    for (int u = 0; u < ST - 1; ++u)
        load_tile(u < nkt ? u : nkt - 1, a_vs[u], a_ss[u], b_vs[u], x_sc[XF ? u : 0], x_sh[XF ? u : 0], a_gs[GX == 1 ? u : 0]);
    // (always_inline: past a code-size threshold the inliner leaves this lambda a real function, and every array it captures by
    // reference then lives in scratch memory -- measured 876 us instead of 51 for a 131072 x 128 data gradient)
    auto tile_step = [&](auto uc, int kt) __attribute__((always_inline)) {  // statically unrolled: the prefetch ring slots are compile-time
        constexpr int u = decltype(uc)::value;
        __syncthreads();  // previous tile fully consumed
        store_tile(kt, a_vs[u], a_ss[u], b_vs[u], x_sc[XF ? u : 0], x_sh[XF ? u : 0], a_gs[GX == 1 ? u : 0]);
        __syncthreads();
        constexpr int un = (u + ST - 1) % ST;
        // always issued (tile index clamped: the last ST-1 prefetches re-read the last tile and are
        // never stored) so that the loop body stays branch-free around the loads
        load_tile(kt + ST - 1 < nkt ? kt + ST - 1 : nkt - 1, a_vs[un], a_ss[un], b_vs[un], x_sc[XF ? un : 0], x_sh[XF ? un : 0],
                  a_gs[GX == 1 ? un : 0]);
        const float* as = As + (wm * 32 + l31) * kAS + 4 * half;
        const float* bs = Bs + (4 * half) * BS + wn * (NT * 32) + l31;
#pragma unroll
        for (int tt = 0; tt < 4 / WK; ++tt) {
            const int t = tt + wk * (4 / WK);  // this wave's share of the k-tile
            const f32x4 av = *reinterpret_cast<const f32x4*>(as + 8 * t);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float bb = bs[(8 * t + q) * BS + nt * 32];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bb, acc[nt], 0, 0, 0);
                }
            }
        }
    };
    int kt0 = 0;
    for (; kt0 + ST <= nkt; kt0 += ST) {  // full groups: no branches between the counted waits
        auto step = [&](auto uc) __attribute__((always_inline)) { tile_step(uc, kt0 + decltype(uc)::value); };
        static_for<0, ST>(step);
    }
    {
        auto step = [&](auto uc) __attribute__((always_inline)) {
            if (kt0 + decltype(uc)::value < nkt) tile_step(uc, kt0 + decltype(uc)::value);
        };
        static_for<0, ST - 1>(step);
    }

    if constexpr (WK == 2) {  // reduce the two k-halves through LDS (B tile memory is free now)
        __syncthreads();
        float* red = Bs + wmn * (NT * 16 * 64);
        if (wk == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(nt * 16 + r) * 64 + lane] = acc[nt][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] += red[(nt * 16 + r) * 64 + lane];
        }
    }

    // ---- epilogue ------------------------------------------------------------
    // D[i][j]: j = l31, i = (r&3) + 8*(r>>2) + 4*half
    const int wrow0 = row0 + wm * 32;
    if (wk == 0) {  // (WK = 2: the second k-half has handed its tile over and only takes part in the finishing barrier below)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = col0 + wn * (NT * 32) + nt * 32 + l31;
        if constexpr (TB) {  // plain guarded store: no bias / activation / pooling on a data gradient
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < rows && col < cout) y[(size_t)row * cout + col] = acc[nt][r];
            }
            if (gepi.ws) push_column_grad_stats(acc[nt], half, wrow0, rows, col, cout, (unsigned)(blockIdx.x * WM + wm), gepi);
            continue;
        }
        const float bv = bias ? bias[col] : 0.f;
        if (pool <= 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[nt][r] + bv;
                if (relu) v = fmaxf(v, 0.f);
                if (row < rows) y[(size_t)row * cout + col] = v;
            }
            if (stats) push_column_stats(acc[nt], half, col, cout, (unsigned)(blockIdx.x * WM + wm), stats);
        } else if (pool == 16) {
            // rows 0..15 live in regs 0..7, rows 16..31 in regs 8..15 (both halves)
            float v0 = acc[nt][0], v1 = acc[nt][8];
#pragma unroll
            for (int r = 1; r < 8; ++r) { v0 = fmaxf(v0, acc[nt][r]); v1 = fmaxf(v1, acc[nt][8 + r]); }
            v0 = fmaxf(v0, __shfl_xor(v0, 32));
            v1 = fmaxf(v1, __shfl_xor(v1, 32));
            v0 += bv; v1 += bv;
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            const int g0 = wrow0 / 16;
            if (half == 0 && wrow0 < rows) y[(size_t)g0 * cout + col] = v0;
            if (half == 0 && wrow0 + 16 < rows) y[(size_t)(g0 + 1) * cout + col] = v1;
        } else {
            float v = acc[nt][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[nt][r]);
            v = fmaxf(v, __shfl_xor(v, 32));
            v += bv;  // max_i relu(x_i + b) == relu(max_i(x_i) + b): fl(x+b) and relu are monotone
            if (relu) v = fmaxf(v, 0.f);
            if (half == 0 && wrow0 < rows) {
                const int g = wrow0 / pool;
                if (pool == 32) y[(size_t)g * cout + col] = v;
                else atomic_max_nonneg(&y[(size_t)g * cout + col], v);  // pool = 32*t, y pre-zeroed, relu on
            }
        }
    }
    }
    // the last workgroup folds the batch-norm sums this launch has left (and derives the constants): pn2_common.h
    pn2_bn_finish(fin, gridDim.x * gridDim.y, blockIdx.x + gridDim.x * blockIdx.y);
}

template <int WM, int WN, int NT, int WK = 1, int ST = 2>
int launch_linear(int rows, int cin, int cout, const float* x, const float* w, const float* bias,
                  int relu, int pool, float* y, hipStream_t st, double* stats = nullptr, const Pn2BnFinish* fin = nullptr) {
    constexpr int BM = 32 * WM, BN = 32 * NT * WN;
    dim3 grid((rows + BM - 1) / BM, cout / BN);
    const bool vec_a = (cin % 4 == 0) && ((uintptr_t)x % 16 == 0);
    const Pn2BnFinish f = fin ? *fin : Pn2BnFinish{};
#define PN2_LK_ARGS rows, cin, cout, x, w, bias, relu, pool, y, stats, Pn2BnGradEpilogue{}, Pn2LoadTransform{}, Pn2GradOnLoad{}, f
    if constexpr (WM == 4 && NT == 4) {
        // the 128x128 tile exists with 16-byte A loads only (the scalar-load variant spills its accumulators)
        if (!vec_a) return launch_linear<2, 2, 2, 1, ST>(rows, cin, cout, x, w, bias, relu, pool, y, st, stats, fin);
        linear_kernel<WM, WN, NT, true, WK, ST><<<grid, 256, 0, st>>>(PN2_LK_ARGS);
    } else {
        if (vec_a) linear_kernel<WM, WN, NT, true, WK, ST><<<grid, 256, 0, st>>>(PN2_LK_ARGS);
        else linear_kernel<WM, WN, NT, false, WK, ST><<<grid, 256, 0, st>>>(PN2_LK_ARGS);
    }
#undef PN2_LK_ARGS
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// pn2_linear_bn_stats_xf: the forward GEMM + statistics epilogue with the load transform (ST = 3, 16-byte A loads)
template <int WM, int WN, int NT>
int launch_linear_xf(int rows, int cin, int cout, const float* x, const float* w, float* y, hipStream_t st, double* stats,
                     const Pn2LoadTransform& xf, const Pn2BnFinish* fin = nullptr) {
    constexpr int BM = 32 * WM, BN = 32 * NT * WN;
    dim3 grid((rows + BM - 1) / BM, cout / BN);
    linear_kernel<WM, WN, NT, true, 1, 3, false, true><<<grid, 256, 0, st>>>(rows, cin, cout, x, w, nullptr, 0, 0, y, stats,
                                                                            Pn2BnGradEpilogue{}, xf, Pn2GradOnLoad{},
                                                                            fin ? *fin : Pn2BnFinish{});
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// Data gradient of a layer with a handful of outputs (the 9-class head): dx[r][i] = sum_o dy[r][o] * w[i][o], K = n_out <= 16.
// An MFMA tile would pad K to 32 and gather dy with scalar loads (measured 411 us at 131072 x 128 <- 9); this is a
// streaming kernel bound by the dx write: thread = (row, 4 consecutive inputs), W (n_in x K) staged in LDS.
template <int K>
__global__ void __launch_bounds__(256)
dgrad_smallk_kernel(int rows, int n_in, const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx) {
    extern __shared__ float sw[];  // n_in * K
    for (int e = threadIdx.x; e < n_in * K; e += 256) sw[e] = w[e];
    __syncthreads();
    const int groups = (n_in + 3) / 4;
    const long long total = (long long)rows * groups;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long r = t / groups;
        const int i0 = (int)(t - r * groups) * 4;
        float d[K];
#pragma unroll
        for (int o = 0; o < K; ++o) d[o] = dy[r * K + o];
        float acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q < n_in ? i0 + q : n_in - 1;
            float a = 0.f;
#pragma unroll
            for (int o = 0; o < K; ++o) a = __builtin_fmaf(d[o], sw[i * K + o], a);
            acc[q] = a;
        }
        if (i0 + 4 <= n_in && (n_in & 3) == 0) {
            *reinterpret_cast<f32x4*>(dx + r * n_in + i0) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (i0 + q < n_in) dx[r * n_in + i0 + q] = acc[q];
        }
    }
}

// The same when the number of 4-input groups divides the block size (n_in = 128: 32 groups): a thread then keeps the SAME four
// inputs for every row it visits, so their 4 x K weights live in registers and the loop is 9 broadcast loads of dy, 36 fma and one
// 16-byte store per row -- bound by the write of dx (131072 x 128 <- 9: 38 -> ~17 us; the LDS form issues 36 ds_read per store).
template <int K>
__global__ void __launch_bounds__(256)
dgrad_smallk_fixed_kernel(int rows, int n_in, const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx) {
    const int groups = n_in >> 2;                   // n_in % 4 == 0, 256 % groups == 0
    const int g = (int)threadIdx.x % groups;
    const int rpb = 256 / groups;                   // rows per block and pass
    float wr[4][K];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int o = 0; o < K; ++o) wr[q][o] = w[(size_t)(4 * g + q) * K + o];
    for (long long r = (long long)blockIdx.x * rpb + (int)threadIdx.x / groups; r < rows; r += (long long)gridDim.x * rpb) {
        float d[K];
#pragma unroll
        for (int o = 0; o < K; ++o) d[o] = dy[r * K + o];
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float t = 0.f;
#pragma unroll
            for (int o = 0; o < K; ++o) t = __builtin_fmaf(d[o], wr[q][o], t);
            a[q] = t;
        }
        *reinterpret_cast<f32x4*>(dx + r * n_in + 4 * g) = a;
    }
}

template <int K>
int launch_dgrad_smallk(int rows, int n_in, const float* dy, const float* w, float* dx, hipStream_t st) {
    const int groups = (n_in + 3) / 4;
    if ((n_in & 3) == 0 && groups <= 256 && 256 % groups == 0 && ((uintptr_t)dx % 16) == 0) {
        const int rpb = 256 / groups;
        long long g = ((long long)rows + rpb - 1) / rpb;
        if (g > 256 * 16) g = 256 * 16;
        dgrad_smallk_fixed_kernel<K><<<(int)g, 256, 0, st>>>(rows, n_in, dy, w, dx);
        PN2_RETURN_IF_LAUNCH_FAILED();
        return PN2_OK;
    }
    const long long total = (long long)rows * groups;
    long long g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    dgrad_smallk_kernel<K><<<(int)g, 256, (size_t)n_in * K * sizeof(float), st>>>(rows, n_in, dy, w, dx);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// Forward of a layer with a handful of OUTPUTS (the 9-class head, model.py:145-146): y[r][o] = sum_k x[r][k] * w[k][o] + b[o],
// N = n_out <= 16.  The MFMA kernels need 32 output columns (zero-padded weight copy, padded output, slice copy, bias add:
// five launches); this is one streaming kernel bound by the read of x: eight lanes share a row -- lane j of the group loads the
// 16-byte pieces k = 4j + 32i, so a wave reads eight contiguous 128-byte runs per load -- keep N partial sums each and combine
// them with three butterfly steps; W (cin x N) sits in LDS.  cin % 4 == 0, x 16-byte aligned.
template <int N>
__global__ void __launch_bounds__(256)
linear_narrow_kernel(int rows, int cin, const float* __restrict__ x, const float* __restrict__ w,
                     const float* __restrict__ bias, float* __restrict__ y) {
    extern __shared__ float sw[];  // cin * N
    for (int e = threadIdx.x; e < cin * N; e += 256) sw[e] = w[e];
    __syncthreads();
    const int j = threadIdx.x & 7;
    const long long groups = (long long)gridDim.x * 32;
    for (long long r = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); r < (long long)((rows + 31) / 32) * 32; r += groups) {
        const long long rc = r < rows ? r : rows - 1;  // whole groups of eight lanes stay in the loop: the butterflies need them
        const float* __restrict__ px = x + rc * cin;
        float acc[N];
#pragma unroll
        for (int o = 0; o < N; ++o) acc[o] = 0.f;
        for (int k = 4 * j; k < cin; k += 32) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(px + k);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int o = 0; o < N; ++o) acc[o] = __builtin_fmaf(v[q], sw[(k + q) * N + o], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < N; ++o) {
            acc[o] += __shfl_xor(acc[o], 1);
            acc[o] += __shfl_xor(acc[o], 2);
            acc[o] += __shfl_xor(acc[o], 4);
        }
        if (r < rows) {
#pragma unroll
            for (int o = 0; o < N; ++o)
                if ((o & 7) == j) y[r * N + o] = acc[o] + (bias ? bias[o] : 0.f);
        }
    }
}

template <int N>
int launch_linear_narrow(int rows, int cin, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    long long g = ((long long)rows + 31) / 32;
    if (g > 256 * 16) g = 256 * 16;
    linear_narrow_kernel<N><<<(int)g, 256, (size_t)cin * N * sizeof(float), st>>>(rows, cin, x, w, bias, y);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

template <int WM, int WN, int NT>
int launch_linear_dgrad(int rows, int n_in, int n_out, const float* dy, const float* w, float* dx, hipStream_t st,
                        const Pn2BnGradEpilogue& gepi = Pn2BnGradEpilogue{}, const Pn2GradOnLoad* gx = nullptr,
                        const Pn2BnFinish* fin = nullptr) {
    constexpr int BM = 32 * WM, BN = 32 * NT * WN;
    dim3 grid((rows + BM - 1) / BM, (n_in + BN - 1) / BN);
    const Pn2BnFinish f = fin ? *fin : Pn2BnFinish{};
    if (gx) {  // dy formed on load from (y, dz): n_out % 4 == 0 and 16-byte aligned operands checked by the caller
        const size_t lds = sizeof(float) * (size_t)n_out * (6 + (gx->pool ? 3 * (BM / 32) : 0));
        if (gx->pool)
            linear_kernel<WM, WN, NT, true, 1, 3, true, false, 2><<<grid, 256, lds, st>>>(rows, n_out, n_in, nullptr, w, nullptr, 0, 0, dx,
                                                                                       nullptr, gepi, Pn2LoadTransform{}, *gx, f);
        else
            linear_kernel<WM, WN, NT, true, 1, 3, true, false, 1><<<grid, 256, lds, st>>>(rows, n_out, n_in, nullptr, w, nullptr, 0, 0, dx,
                                                                                       nullptr, gepi, Pn2LoadTransform{}, *gx, f);
        PN2_RETURN_IF_LAUNCH_FAILED();
        return PN2_OK;
    }
    const bool vec_a = (n_out % 4 == 0) && ((uintptr_t)dy % 16 == 0);
    if (vec_a)
        linear_kernel<WM, WN, NT, true, 1, 3, true><<<grid, 256, 0, st>>>(rows, n_out, n_in, dy, w, nullptr, 0, 0, dx, nullptr, gepi,
                                                                         Pn2LoadTransform{}, Pn2GradOnLoad{}, f);
    else
        linear_kernel<WM, WN, NT, false, 1, 3, true><<<grid, 256, 0, st>>>(rows, n_out, n_in, dy, w, nullptr, 0, 0, dx, nullptr, gepi,
                                                                          Pn2LoadTransform{}, Pn2GradOnLoad{}, f);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// ---- few-row layers (SA3/SA4/FP1-FP3: 1024 .. 16384 rows onto 256 .. 512 columns) --------------------------------------
// linear_kernel's smallest tile (32 x 128 per workgroup, one 32x32 accumulator per wave, k-tiles of 32 staged through
// LDS behind two barriers) leaves a 1024-row layer on 64 of the 256 CUs and spends more time in barriers and LDS
// round trips than in its 16 MFMAs per k-tile (measured 8-35 % of the MFMA peak on these layers).  Here one workgroup
// owns a (32*TM) x (32*TN) output tile and its 4 waves split the CONTRACTION: wave w takes k in [w*K/4, (w+1)*K/4) and
// feeds v_mfma_f32_32x32x2_f32 straight from global memory (the operands of a few-row layer live in L2):
//   A: lane (row = l&31, half = l>>5) loads 4 consecutive k of its row with one 16-byte load  (k = k0 + 4*half + q),
//   B: lane (col = l&31, half)        loads w[k0 + 4*half + q][col] for q = 0..3 (128-byte segments per half-wave);
// step q of a group contracts exactly those two k -- the MFMA only needs A and B to agree on the k of each half, so
// this permuted visiting order is exact.  No LDS, no barrier in the loop; G groups of loads stay in flight under the
// MFMAs of the previous ones.  The four partial tiles are added in LDS in a fixed order (deterministic), then the
// usual bias / ReLU / max-over-32-rows epilogue.  rows/32/TM x cout/32/TN workgroups: 256 for the 1024-row layers.
template <int TM, int TN, bool VEC_A, int G>
__global__ void __launch_bounds__(256)
linear_splitk_kernel(int rows, int cin, int cout, const float* __restrict__ x, const float* __restrict__ w,
                     const float* __restrict__ bias, int relu, int pool, float* __restrict__ y,
                     double* __restrict__ stats = nullptr, Pn2BnFinish fin = Pn2BnFinish{}) {
    // G = groups of 8 k in flight per wave (a group's MFMAs take ~0.1 us, an L2 round trip ~0.7 us)
    __shared__ float red[3 * TM * TN * 16 * 64];
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = blockIdx.x * (32 * TM), col0 = blockIdx.y * (32 * TN);
    // this wave's share of the contraction, in groups of 8 k
    const int ngroups = (cin + 7) / 8;
    const int g0 = (ngroups * wave) / 4, g1 = (ngroups * (wave + 1)) / 4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const float* __restrict__ xr[TM];
    bool rok[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int r = row0 + a * 32 + l31;
        rok[a] = r < rows;
        xr[a] = x + (size_t)(rok[a] ? r : rows - 1) * cin;
    }
    const float* __restrict__ wc = w + col0 + l31;

    f32x4 av[G][TM];
    float bv[G][4][TN];
    auto fetch = [&](int g, f32x4 (&a_)[TM], float (&b_)[4][TN]) {
        // unconditional clamped loads (counted vmcnt), out-of-range k zeroed on the A side only (B stays finite data)
        const int k = g * 8 + 4 * half;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            f32x4 v;
            if constexpr (VEC_A) {
                const int kc = k + 4 <= cin ? k : cin - 4;
                v = *reinterpret_cast<const f32x4*>(xr[a] + kc);
                if (!(k + 4 <= cin && rok[a])) v = f32x4{0.f, 0.f, 0.f, 0.f};  // VEC_A: cin % 4 == 0, so a group of 4 is all in or all out
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kq = k + q < cin ? k + q : cin - 1;
                    const float e = xr[a][kq];
                    v[q] = (k + q < cin && rok[a]) ? e : 0.f;
                }
            }
            a_[a] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kq = k + q < cin ? k + q : cin - 1;
#pragma unroll
            for (int b = 0; b < TN; ++b) b_[q][b] = wc[(size_t)kq * cout + b * 32];
        }
    };
    auto contract = [&](const f32x4 (&a_)[TM], const float (&b_)[4][TN]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[a][q], b_[q][b], acc[a][b], 0, 0, 0);
    };
    if (g0 < g1) {
        const int last = g1 - 1;
#pragma unroll
        for (int u = 0; u < G; ++u) fetch(g0 + u < g1 ? g0 + u : last, av[u], bv[u]);
        int g = g0;
        for (; g + G <= g1; g += G) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                f32x4 ca[TM];
                float cb[4][TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) ca[a] = av[u][a];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int b = 0; b < TN; ++b) cb[q][b] = bv[u][q][b];
                fetch(g + G + u < g1 ? g + G + u : last, av[u], bv[u]);
                contract(ca, cb);
            }
        }
#pragma unroll
        for (int u = 0; u < G - 1; ++u)
            if (g + u < g1) contract(av[u], bv[u]);
    }
    // fixed-order reduction of the four partial tiles: waves 1..3 park theirs in LDS, wave 0 adds them 1, 2, 3
    if (wave > 0) {
        float* dst = red + (size_t)(wave - 1) * (TM * TN * 16 * 64);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((a * TN + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (wave == 0) {  // (waves 1..3 have handed their tiles over and only take part in the finishing barrier below)
#pragma unroll
    for (int wv = 0; wv < 3; ++wv) {
        const float* src = red + (size_t)wv * (TM * TN * 16 * 64);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += src[((a * TN + b) * 16 + r) * 64 + lane];
    }
    // epilogue: D[i][j], j = l31, i = (r&3) + 8*(r>>2) + 4*half
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int wrow0 = row0 + a * 32;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = col0 + b * 32 + l31;
            const float bb = bias ? bias[col] : 0.f;
            if (pool <= 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = acc[a][b][r] + bb;
                    if (relu) v = fmaxf(v, 0.f);
                    if (row < rows) y[(size_t)row * cout + col] = v;
                }
                if (stats) push_column_stats(acc[a][b], half, col, cout, (unsigned)(blockIdx.x * TM + a), stats);
            } else {  // pool == 32: the tile's 32 rows are one neighbourhood
                float v = acc[a][b][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[a][b][r]);
                v = fmaxf(v, __shfl_xor(v, 32));
                v += bb;  // max_i relu(x_i + b) == relu(max_i(x_i) + b)
                if (relu) v = fmaxf(v, 0.f);
                if (half == 0 && wrow0 < rows) y[(size_t)(wrow0 / 32) * cout + col] = v;
            }
        }
    }
    }
    pn2_bn_finish(fin, gridDim.x * gridDim.y, blockIdx.x + gridDim.x * blockIdx.y);
}

template <int TM, int TN, int G = 2>
int launch_linear_splitk(int rows, int cin, int cout, const float* x, const float* w, const float* bias, int relu,
                         int pool, float* y, hipStream_t st, double* stats = nullptr, const Pn2BnFinish* fin = nullptr) {
    dim3 grid((rows + 32 * TM - 1) / (32 * TM), cout / (32 * TN));
    const bool vec_a = (cin % 4 == 0) && ((uintptr_t)x % 16 == 0);
    const Pn2BnFinish f = fin ? *fin : Pn2BnFinish{};
    if (vec_a) linear_splitk_kernel<TM, TN, true, G><<<grid, 256, 0, st>>>(rows, cin, cout, x, w, bias, relu, pool, y, stats, f);
    else linear_splitk_kernel<TM, TN, false, G><<<grid, 256, 0, st>>>(rows, cin, cout, x, w, bias, relu, pool, y, stats, f);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// Weight gradient of a dense layer (training): dW (cin, cout) = x^T (cin, rows) . dy (rows, cout), the reduction over
// ALL rows (524288 for SA1) that hipBLASLt runs at ~10 % of the memory roofline on these tall-skinny shapes.  One
// wave owns a (32*TM x 32*TN) tile of dW for a chunk of rows and streams x and dy straight from HBM into the MFMA
// operands -- a 32x32x2 step contracts two rows: lanes 0-31 carry row r, lanes 32-63 row r+1, each a contiguous
// 128-byte segment of x (A operand, m = cin index) and of dy (B operand, n = cout index) -- then adds its partial
// tile to dW with fp32 atomics (dW zeroed by the entry point).  Memory-bound: every row is read once per tile column.
// XF: x is the pre-normalisation output of the layer below, the operand is relu?(fma(x, scale[m], shift[m])) (per lane: its
// TM input channels), see linear_kernel.
// GX = 1 / 2: dy is the gradient leaving this layer's batch norm (+ReLU; 2: + max over groups of 32 rows), formed per lane from
// (y, dz) and the constants of its TN output channels while the operand is loaded (Pn2GradOnLoad; the `dy` argument is not read).
// GX = 2: chunk % 32 == 0, so a fetch of 2U rows never straddles a pooling group.
// NWV = waves per workgroup: 4 (two workgroups per CU) or 8 (one workgroup per CU, TWO WAVES PER SIMD with the same number of
// workgroups, hence the same number of closing atomics: one wave's operand transform and loads run under its partner's MFMAs).
template <int TM, int TN, bool XF = false, int GX = 0, int NWV = 4>
__global__ void __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1)
linear_wgrad_kernel(int rows, int cin, int cout, int chunk, const float* __restrict__ x,
                    const float* __restrict__ dy, float* __restrict__ dw, Pn2LoadTransform xf = Pn2LoadTransform{},
                    Pn2GradOnLoad gx = Pn2GradOnLoad{}) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // row bookkeeping stays in SGPRs
    const long long c0 = ((long long)blockIdx.x * NWV + wave) * chunk;
    // a wave past the end contributes zeros (it still takes part in the block reduction)
    const int r0 = c0 < rows ? (int)c0 : rows;
    const int r1 = c0 + chunk < rows ? (int)(c0 + chunk) : rows;
    const int m0 = blockIdx.y * 32 * TM, n0 = blockIdx.z * 32 * TN;
    // per-lane element offsets inside a pair of rows; the pair's base address is wave-uniform (SGPR base + VGPR offset
    // loads: no per-load 64-bit address registers).  Columns past cin / cout read clamped data and are dropped at the
    // store.
    const int hoff = rows > 1 ? half : 0;
    unsigned offa[TM], offb[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) { const int m = m0 + t * 32 + l31; offa[t] = (unsigned)(hoff * cin + (m < cin ? m : cin - 1)); }
#pragma unroll
    for (int t = 0; t < TN; ++t) { const int n = n0 + t * 32 + l31; offb[t] = (unsigned)(hoff * cout + (n < cout ? n : cout - 1)); }
    float xsc[TM], xsh[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int m = m0 + t * 32 + l31;
        xsc[t] = XF ? xf.scale[m < cin ? m : cin - 1] : 1.f;
        xsh[t] = XF ? xf.shift[m < cin ? m : cin - 1] : 0.f;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // Register double buffer: the loads of the NEXT U k-steps (pairs of rows) are in flight while the MFMAs of the
    // current U run.  Loads are unconditional (pair base clamped to the last pair of the matrix) so the compiler waits
    // with counted vmcnt.  A lane contributes iff the row it loaded belongs to this k-step and to this chunk; otherwise
    // A = 0 is enough (the dy values it pairs with are finite data of the matrix).
    constexpr int U = 4;
    const int last_pair = rows > 1 ? rows - 2 : 0;
    // GX: the six batch-norm gradient constants of this lane's TN output channels
    float gc[GX != 0 ? TN : 1][6];
    unsigned offp[GX == 2 ? TN : 1];  // GX = 2: this lane's columns inside a row of the pooled tensors
    if constexpr (GX != 0) {
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int n = n0 + t * 32 + l31;
            const int nc = n < cout ? n : cout - 1;
#pragma unroll
            for (int j = 0; j < 6; ++j) gc[t][j] = gx.coef[(size_t)j * cout + nc];
            if constexpr (GX == 2) offp[t] = (unsigned)nc;
        }
    }
    // per register buffer: x (TM), dy or y (TN), GX = 1: dz (TN), GX = 2: pooled gradient / maximum / tie count of the fetch's group
    struct Buf {
        float av[U][TM], bv[U][TN];
        float bg[GX == 1 ? U : 1][GX == 1 ? TN : 1];
        float pd[GX == 2 ? TN : 1], pm[GX == 2 ? TN : 1], pn[GX == 2 ? TN : 1];
    };
    Buf b0, b1;
    auto fetch = [&](Buf& bf, int r) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rb = r + 2 * u < last_pair ? r + 2 * u : last_pair;
            const float* __restrict__ px = x + (size_t)rb * cin;
#pragma unroll
            for (int t = 0; t < TM; ++t) bf.av[u][t] = px[offa[t]];
            if constexpr (GX == 0) {
                const float* __restrict__ pd = dy + (size_t)rb * cout;
#pragma unroll
                for (int t = 0; t < TN; ++t) bf.bv[u][t] = pd[offb[t]];
            } else {
                const float* __restrict__ py = gx.y + (size_t)rb * cout;
#pragma unroll
                for (int t = 0; t < TN; ++t) bf.bv[u][t] = py[offb[t]];
                if constexpr (GX == 1) {
                    const float* __restrict__ pg = gx.dz + (size_t)rb * cout;
#pragma unroll
                    for (int t = 0; t < TN; ++t) bf.bg[u][t] = pg[offb[t]];
                }
            }
        }
        if constexpr (GX == 2) {  // r is a multiple of 2U = 8 inside a chunk that starts on a multiple of 32: one group per fetch
            const int rc = r < last_pair ? r : last_pair;
            const size_t go = (size_t)(rc >> 5) * cout;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                bf.pd[t] = gx.dz[go + offp[t]];
                bf.pm[t] = gx.zmax[go + offp[t]];
                bf.pn[t] = gx.ties[go + offp[t]];
            }
        }
    };
    auto contract = [&](const Buf& bf, int r) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rb = r + 2 * u < last_pair ? r + 2 * u : last_pair;
            const int loaded = rb + half;
            const bool rv = loaded >= r + 2 * u && loaded < r1;
            float bw[TN];
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                if constexpr (GX == 1)
                    bw[b] = pn2_bn_grad_element(bf.bv[u][b], bf.bg[u][b], gc[b][0], gc[b][1], gc[b][2], gc[b][3], gc[b][4], gc[b][5],
                                                gx.relu);
                else if constexpr (GX == 2)
                    bw[b] = pn2_bn_grad_element_pooled(bf.bv[u][b], bf.pd[b], bf.pm[b], bf.pn[b], gc[b][0], gc[b][1], gc[b][2],
                                                       gc[b][3], gc[b][4], gc[b][5], gx.relu);
                else
                    bw[b] = bf.bv[u][b];
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                float av_t = bf.av[u][a];
                if constexpr (XF) {
                    av_t = __builtin_fmaf(av_t, xsc[a], xsh[a]);
                    av_t = xf.relu ? fmaxf(av_t, 0.f) : av_t;
                }
                const float av_m = rv ? av_t : 0.f;
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_m, bw[b], acc[a][b], 0, 0, 0);
            }
        }
    };
    fetch(b0, r0);
    for (int r = r0; r < r1; r += 4 * U) {
        fetch(b1, r + 2 * U);
        __builtin_amdgcn_sched_barrier(0);
        contract(b0, r);
        __builtin_amdgcn_sched_barrier(0);
        fetch(b0, r + 4 * U);
        __builtin_amdgcn_sched_barrier(0);
        contract(b1, r + 2 * U);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the waves of the block own consecutive row chunks of the SAME tile: add them up through LDS (a tree: log2(NWV) steps)
    // so that only one wave per block issues the global atomics
    extern __shared__ __attribute__((aligned(16))) float wgrad_red[];  // (NWV / 2) x TM * TN * 16 * 64 floats (128 KB at NWV = 8)
    float (*red)[TM * TN * 16 * 64] = reinterpret_cast<float (*)[TM * TN * 16 * 64]>(wgrad_red);
    auto spill = [&](float* dst) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((a * TN + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    };
    auto absorb = [&](const float* src) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += src[((a * TN + b) * 16 + r) * 64 + lane];
    };
#pragma unroll
    for (int h = NWV / 2; h >= 1; h >>= 1) {  // waves [h, 2h) hand their tiles to waves [0, h)
        if (wave >= h && wave < 2 * h) spill(red[wave - h]);
        __syncthreads();
        if (wave < h) absorb(red[wave]);
        __syncthreads();
    }
    if (wave != 0) return;
    // D[i][j]: j = l31 (cout index), i = (r&3) + 8*(r>>2) + 4*half (cin index)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = n0 + b * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < cin && n < cout) atomicAdd(&dw[(size_t)m * cout + n], acc[a][b][r]);
            }
        }
}

// grouped input of an SA layer: out[b,j,k,:] = [xyz[b,idx]-new_xyz[b,j] | points[b,idx]]
// (xyz FIRST: util/pointnet_util.py:43-54).  One thread per output float.
__global__ void __launch_bounds__(256)
sa_group_concat_kernel(int n, int m, int nsample, int c, const float* __restrict__ xyz_all,
                       const float* __restrict__ new_xyz_all, const float* __restrict__ points_all,
                       const int* __restrict__ idx_all, float* __restrict__ out_all) {
    const unsigned cw = 3u + (unsigned)c;
    const unsigned rows = (unsigned)m * nsample;
    const unsigned total = rows * cw;
    const int bi = blockIdx.y;
    const float* __restrict__ xyz = xyz_all + (size_t)bi * n * 3;
    const float* __restrict__ nxyz = new_xyz_all + (size_t)bi * m * 3;
    const float* __restrict__ pts = points_all ? points_all + (size_t)bi * n * c : nullptr;
    const int* __restrict__ idx = idx_all + (size_t)bi * rows;
    float* __restrict__ out = out_all + (size_t)bi * rows * cw;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned row = e / cw, col = e - row * cw;
        const int ii = idx[row];
        float v;
        if (col < 3u) v = xyz[(size_t)ii * 3 + col] - nxyz[(size_t)(row / nsample) * 3 + col];
        else v = pts[(size_t)ii * c + (col - 3u)];
        out[e] = v;
    }
}

}  // namespace

PN2_TUNABLE(int, g_lin_cfg, 0)     // tuning hook (pn2_debug_set(8, v)): 0 = auto, 1..4 = force <4,1,4> / <2,2,2> / <1,4,1> / <1,2,1,split-K>
PN2_TUNABLE(int, g_lin_narrow, 1)  // tuning hook (pn2_debug_set(17, v)): the streaming forward kernel of the narrow training layers (pn2_fwd_narrow.h)
PN2_TUNABLE(int, g_lin_stages, 3)  // tuning hook (pn2_debug_set(5, v)): register prefetch depth of linear_kernel
PN2_TUNABLE(int, g_wgrad_nwv8, 1)  // tuning hook (pn2_debug_set(18, v)): eight-wave workgroups for the 64 x 128 weight-gradient tile over >= 65536 rows
PN2_TUNABLE(int, g_wgrad_waves, 0) // tuning hook (pn2_debug_set(9, v)): waves in flight targeted by pn2_linear_wgrad (0 = auto)

static int linear_impl(int rows, int cin, int cout, const float* x, const float* w,
                       const float* bias, int relu, int pool, float* y, void* stream, double* stats,
                       const Pn2BnFinish* fin = nullptr) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return PN2_EINVAL;
    if (!x || !w || !y) return PN2_ENULL;
    if (cout % 32 != 0 || ((uintptr_t)w % 16) != 0) return PN2_EUNSUP;
    if ((long long)rows + 128 > 0x7fffffffLL) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pool > 1) {
        if (pool != 16 && pool % 32 != 0) return PN2_EUNSUP;
        if (rows % pool != 0) return PN2_EINVAL;
        if (pool > 32) {
            if (!relu) return PN2_EUNSUP;
            hipError_t e = hipMemsetAsync(y, 0, sizeof(float) * (size_t)(rows / pool) * cout, st);
            if (e != hipSuccess) return (int)e;
        }
    }
#define PN2_LIN(WM_, WN_, NT_, WK_)                                                                          \
    (st_depth >= 4 ? launch_linear<WM_, WN_, NT_, WK_, 4>(rows, cin, cout, x, w, bias, relu, pool, y, st, stats, fin)      \
     : st_depth == 3 ? launch_linear<WM_, WN_, NT_, WK_, 3>(rows, cin, cout, x, w, bias, relu, pool, y, st, stats, fin)    \
                     : launch_linear<WM_, WN_, NT_, WK_, 2>(rows, cin, cout, x, w, bias, relu, pool, y, st, stats, fin))
    const int st_depth = g_lin_stages;
    // training forward of a narrow layer over many rows (statistics epilogue, no bias / activation): the streaming kernel
    if (g_lin_cfg == 0 && g_lin_narrow && stats && !bias && !relu && pool <= 1 && fwd_narrow_fits(rows, cin, cout, x, w))
        return launch_fwd_narrow(rows, cin, cout, x, nullptr, w, y, stats, fin, st);
    if (g_lin_cfg >= 5 && (pool <= 1 || pool == 32)) {  // tuning hook: split-K direct-feed tiles
        if (g_lin_cfg == 5) return launch_linear_splitk<1, 1, 2>(rows, cin, cout, x, w, bias, relu, pool, y, st, stats, fin);
        if (g_lin_cfg == 6) return launch_linear_splitk<1, 1, 4>(rows, cin, cout, x, w, bias, relu, pool, y, st, stats, fin);
        if (g_lin_cfg == 7) return launch_linear_splitk<1, 1, 8>(rows, cin, cout, x, w, bias, relu, pool, y, st, stats, fin);
        if (g_lin_cfg == 8 && cout % 64 == 0) return launch_linear_splitk<1, 2, 4>(rows, cin, cout, x, w, bias, relu, pool, y, st, stats, fin);
    }
    // few rows (FP1: 1024): one 32x32 tile per workgroup with the contraction split over its 4 waves fills the chip
    // where the LDS-tiled kernel leaves 3/4 of the CUs idle: 1024x768->256 15.3 -> 9.7 us, 1024x256->256 7.6 -> 6.0 us;
    // from 4096 rows on the LDS-tiled kernel wins (operand re-reads from L2) -- profiles/r02_linear_splitk.txt
    if (g_lin_cfg == 0 && rows <= 2048 && (pool <= 1 || pool == 32))
        return launch_linear_splitk<1, 1, 2>(rows, cin, cout, x, w, bias, relu, pool, y, st, stats, fin);
    if (cout % 128 == 0 && g_lin_cfg != 0 && g_lin_cfg < 5) {  // tuning hook: force a tile configuration
        if (g_lin_cfg == 1) return PN2_LIN(4, 1, 4, 1);
        if (g_lin_cfg == 2) return PN2_LIN(2, 2, 2, 1);
        if (g_lin_cfg == 3) return PN2_LIN(1, 4, 1, 1);
        return PN2_LIN(1, 2, 1, 2);
    }
    if (cout % 128 == 0) {
        // largest tile that still yields >= 2 blocks per CU; small problems get 32-row blocks.  The 128-row tile is
        // only built for 16-byte A loads: with scalar A loads (cin % 4 != 0 or unaligned x) its accumulators spill
        // to scratch (1.7 ms instead of 55 us at 131072 x 134 -> 128), so those shapes take the 64-row tile.
        const long long cb = cout / 128;
        const bool vec_a = (cin % 4 == 0) && ((uintptr_t)x % 16 == 0);
        // one column of 128-wide tiles: the 64 x 128 tile beats the 128 x 128 one (131072 x 128 -> 128: 54.8 vs 62.0 us,
        // 524288 x 128 -> 128 + max: 159 vs 176 us; profiles/r02_linear_splitk.txt)
        if (((rows + 127) / 128) * cb >= 512 && vec_a && cb > 1) return PN2_LIN(4, 1, 4, 1);
        if (((rows + 63) / 64) * cb >= 512) return PN2_LIN(2, 2, 2, 1);
        if (((rows + 31) / 32) * cb >= 256 || cin < 128) return PN2_LIN(1, 4, 1, 1);
        return PN2_LIN(1, 2, 1, 2);  // few rows: 32x64 tiles, split-K in the block
    }
    // few rows onto 64 columns (the hoisted product of SA2: 16384 x 64 -> 64): 32 x 64 tiles with the k-tile split between
    // two waves give 512 workgroups where the 128-row tile leaves half of the CUs idle: 10.3 -> 6.5 us (tools/dbg/lin_time.py)
    if (cout % 64 == 0 && pool <= 1 && rows <= 32768) return PN2_LIN(1, 2, 1, 2);
    if (cout % 64 == 0) return PN2_LIN(4, 1, 2, 1);
    return PN2_LIN(4, 1, 1, 1);
#undef PN2_LIN
}

extern "C" int pn2_linear(int rows, int cin, int cout, const float* x, const float* w,
                          const float* bias, int relu, int pool, float* y, void* stream) {
    return linear_impl(rows, cin, cout, x, w, bias, relu, pool, y, stream, nullptr);
}

// y (rows, cout) = x (rows, cin) . w (cin, cout) + bias for 1 <= cout <= 16 (no activation): the class head of the network
// (model.py:145-146, tf_util.conv1d with activation_fn=None) without padding the layer to an MFMA tile.  cin % 4 == 0,
// cin * cout * 4 <= 48 KB, x 16-byte aligned; otherwise PN2_EUNSUP (pad and call pn2_linear).
extern "C" int pn2_linear_narrow(int rows, int cin, int cout, const float* x, const float* w, const float* bias, float* y,
                                 void* stream) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return PN2_EINVAL;
    if (!x || !w || !y) return PN2_ENULL;
    if (cout > 16 || cin % 4 != 0 || ((uintptr_t)x % 16) != 0 || (size_t)cin * cout * sizeof(float) > 48 * 1024) return PN2_EUNSUP;
    if ((long long)rows + 128 > 0x7fffffffLL) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (cout) {
#define PN2_NW(N_) case N_: return launch_linear_narrow<N_>(rows, cin, x, w, bias, y, st);
        PN2_NW(1) PN2_NW(2) PN2_NW(3) PN2_NW(4) PN2_NW(5) PN2_NW(6) PN2_NW(7) PN2_NW(8) PN2_NW(9) PN2_NW(10) PN2_NW(11)
        PN2_NW(12) PN2_NW(13) PN2_NW(14) PN2_NW(15) PN2_NW(16)
#undef PN2_NW
    }
    return PN2_EUNSUP;
}

// y = x . w (no bias, no activation) AND the per-column sums the following training-mode batch norm needs (tf_util.py:186-204:
// conv2d -> batch_norm_template): every wave adds its tile's column sums of y and y^2 (16 fp32 terms per lane, then fp64) to
// one of kPn2BnSlots copies of the accumulators in `bn_workspace` (pn2_bn_workspace_bytes(cout), ZEROED by the caller);
// pn2_bn_relu_forward_stats folds the copies and normalises.  Saves the statistics pass over y.
extern "C" int pn2_linear_bn_stats(int rows, int cin, int cout, const float* x, const float* w, float* y,
                                   void* bn_workspace, size_t workspace_bytes, void* stream) {
    if (!bn_workspace) return PN2_ENULL;
    if (cout <= 0 || workspace_bytes < sizeof(double) * pn2_bn_ws_doubles(cout, kPn2BnSlots) || ((uintptr_t)bn_workspace % 8) != 0)
        return PN2_EINVAL;
    return linear_impl(rows, cin, cout, x, w, nullptr, 0, 0, y, stream, static_cast<double*>(bn_workspace));
}

// pn2_linear_bn_stats on the PRE-normalisation output x_raw of the layer below: the A operand is
// relu?(fma(x_raw, a_scale[k], a_shift[k])) formed while the tile is staged (a_scale / a_shift (cin) from
// pn2_bn_relu_forward_deferred).  cin % 4 == 0, x_raw / a_scale / a_shift 16-byte aligned, rows > 2048 (the few-row split-K
// tiles have no staging step); otherwise PN2_EUNSUP.
static int linear_bn_stats_xf_impl(int rows, int cin, int cout, const float* x_raw, const float* w, float* y,
                                   void* bn_workspace, size_t workspace_bytes, const float* a_scale, const float* a_shift,
                                   int a_relu, void* stream, const Pn2BnFinish* fin) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return PN2_EINVAL;
    if (!x_raw || !w || !y || !bn_workspace || !a_scale || !a_shift) return PN2_ENULL;
    if (workspace_bytes < sizeof(double) * pn2_bn_ws_doubles(cout, kPn2BnSlots) || ((uintptr_t)bn_workspace % 8) != 0) return PN2_EINVAL;
    if (cout % 32 != 0 || cin % 4 != 0 || rows <= 2048) return PN2_EUNSUP;
    if ((((uintptr_t)x_raw | (uintptr_t)w | (uintptr_t)a_scale | (uintptr_t)a_shift) % 16) != 0) return PN2_EUNSUP;
    if ((long long)rows + 128 > 0x7fffffffLL) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* stats = static_cast<double*>(bn_workspace);
    const Pn2LoadTransform xf{a_scale, a_shift, a_relu};
    if (g_lin_cfg == 0 && g_lin_narrow && fwd_narrow_fits(rows, cin, cout, x_raw, w))
        return launch_fwd_narrow(rows, cin, cout, x_raw, &xf, w, y, stats, fin, st);
#ifdef PN2_TUNING_HOOKS
    if (cout % 128 == 0 && g_lin_cfg == 1) return launch_linear_xf<4, 1, 4>(rows, cin, cout, x_raw, w, y, st, stats, xf, fin);
    if (cout % 128 == 0 && g_lin_cfg == 2) return launch_linear_xf<2, 2, 2>(rows, cin, cout, x_raw, w, y, st, stats, xf, fin);
    if (cout % 128 == 0 && g_lin_cfg == 3) return launch_linear_xf<1, 4, 1>(rows, cin, cout, x_raw, w, y, st, stats, xf, fin);
#endif
    if (cout % 128 == 0) {  // the tile choice of linear_impl, without its 128 x 128 tile (with the transform's registers its
                            // accumulators spill: 630 us instead of ~45 at 32768 x 128 -> 256)
        const long long cb = cout / 128;
        if (((rows + 63) / 64) * cb >= 512) return launch_linear_xf<2, 2, 2>(rows, cin, cout, x_raw, w, y, st, stats, xf, fin);
        return launch_linear_xf<1, 4, 1>(rows, cin, cout, x_raw, w, y, st, stats, xf, fin);
    }
    if (cout % 64 == 0) return launch_linear_xf<4, 1, 2>(rows, cin, cout, x_raw, w, y, st, stats, xf, fin);
    return launch_linear_xf<4, 1, 1>(rows, cin, cout, x_raw, w, y, st, stats, xf, fin);
}

extern "C" int pn2_linear_bn_stats_xf(int rows, int cin, int cout, const float* x_raw, const float* w, float* y,
                                      void* bn_workspace, size_t workspace_bytes, const float* a_scale, const float* a_shift,
                                      int a_relu, void* stream) {
    return linear_bn_stats_xf_impl(rows, cin, cout, x_raw, w, y, bn_workspace, workspace_bytes, a_scale, a_shift, a_relu, stream,
                                   nullptr);
}

// pn2_linear_bn_stats (a_scale == NULL) / pn2_linear_bn_stats_xf whose LAST WORKGROUP also does what used to be the next launch
// (pn2_common.h pn2_bn_finish): finish = 1 folds the slot copies of the column sums (then pn2_bn_relu_forward_mode /
// pn2_bn_relu_forward_pool with stats_mode 3 normalise); finish = 2 also derives what pn2_bn_relu_forward_deferred publishes --
// save_mean / save_invstd, the moving averages, per-channel (scale, shift) -- so a layer that hands its un-normalised output on
// is ONE launch.  tf_util.py:186-204 + :555-581.
extern "C" int pn2_linear_bn_stats_fin(int rows, int cin, int cout, const float* x, const float* w, float* y, void* bn_workspace,
                                       size_t workspace_bytes, const float* a_scale, const float* a_shift, int a_relu,
                                       int finish, const float* gamma, const float* beta, const float* bias, float eps,
                                       float decay, float* running_mean, float* running_var, float* save_mean,
                                       float* save_invstd, float* scale, float* shift, void* stream) {
    if (!bn_workspace) return PN2_ENULL;
    if (finish != 1 && finish != 2) return PN2_EINVAL;
    if (cout <= 0 || workspace_bytes < sizeof(double) * pn2_bn_ws_doubles(cout, kPn2BnSlots) || ((uintptr_t)bn_workspace % 8) != 0)
        return PN2_EINVAL;
    if ((a_scale == nullptr) != (a_shift == nullptr)) return PN2_ENULL;
    Pn2BnFinish f{};
    f.kind = finish; f.c = cout; f.nslots = kPn2BnSlots; f.rows = rows; f.ws = static_cast<double*>(bn_workspace);
    if (finish == 2) {
        if (!gamma || !beta || !save_mean || !save_invstd || (scale == nullptr) != (shift == nullptr)) return PN2_ENULL;
        if ((running_mean == nullptr) != (running_var == nullptr)) return PN2_ENULL;
        f.gamma = gamma; f.beta = beta; f.bias = bias; f.eps = eps; f.decay = decay; f.running_mean = running_mean;
        f.running_var = running_var; f.save_mean = save_mean; f.save_invstd = save_invstd; f.scale = scale; f.shift = shift;
    }
    if (a_scale)
        return linear_bn_stats_xf_impl(rows, cin, cout, x, w, y, bn_workspace, workspace_bytes, a_scale, a_shift, a_relu, stream, &f);
    return linear_impl(rows, cin, cout, x, w, nullptr, 0, 0, y, stream, static_cast<double*>(bn_workspace), &f);
}

// Data gradient of a dense layer (training): dx (rows, cin) = dy (rows, cout) . W^T, W (cin, cout) row-major as the forward
// pass holds it (no transposed copy); any cin / cout.  The reference gets this from tf.gradients of tf.nn.conv2d
// (util/tf_util.py:181-186).
static int linear_dgrad_impl(int rows, int cin, int cout, const float* dy, const float* w, float* dx, void* stream,
                             const Pn2BnGradEpilogue& gepi, const Pn2GradOnLoad* gx = nullptr, const Pn2BnFinish* fin = nullptr) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return PN2_EINVAL;
    if ((!dy && !gx) || !w || !dx) return PN2_ENULL;
    if ((long long)rows + 128 > 0x7fffffffLL) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (gx) {
        if (!gx->y || !gx->dz || !gx->coef || (gx->pool && (!gx->zmax || !gx->ties))) return PN2_ENULL;
        if ((gx->pool != 0 && gx->pool != 32) || (gx->pool && rows % 32 != 0)) return PN2_EINVAL;
        if (cout % 4 != 0 || cout <= 16 || cout > kPn2GxMaxC) return PN2_EUNSUP;
        if ((((uintptr_t)gx->y | (uintptr_t)gx->dz | (uintptr_t)gx->coef) % 16) != 0) return PN2_EUNSUP;
    }
    if (cout <= 16 && (size_t)cin * cout * sizeof(float) <= 48 * 1024) {
        if (gepi.ws) return PN2_EUNSUP;  // the streaming kernel has no accumulator tiles to take the sums from
        switch (cout) {
#define PN2_SK(K_) case K_: return launch_dgrad_smallk<K_>(rows, cin, dy, w, dx, st);
            PN2_SK(1) PN2_SK(2) PN2_SK(3) PN2_SK(4) PN2_SK(5) PN2_SK(6) PN2_SK(7) PN2_SK(8) PN2_SK(9) PN2_SK(10) PN2_SK(11)
            PN2_SK(12) PN2_SK(13) PN2_SK(14) PN2_SK(15) PN2_SK(16)
#undef PN2_SK
        }
    }
    // 128 -> 128 over many rows with dy formed on load: the streaming kernel with two waves per SIMD (pn2_dgrad_wide.h; same dx bits)
    if (g_lin_cfg == 0 && g_lin_narrow && dgrad_wide_fits(rows, cin, cout, gx, w)) return launch_dgrad_wide(rows, cin, w, dx, *gx, gepi, fin, st);
#ifdef PN2_TUNING_HOOKS
    if (cin > 96 && g_lin_cfg == 1) return launch_linear_dgrad<4, 1, 4>(rows, cin, cout, dy, w, dx, st, gepi, gx, fin);
    if (cin > 96 && g_lin_cfg == 2) return launch_linear_dgrad<2, 2, 2>(rows, cin, cout, dy, w, dx, st, gepi, gx, fin);
    if (cin > 96 && g_lin_cfg == 3) return launch_linear_dgrad<1, 4, 1>(rows, cin, cout, dy, w, dx, st, gepi, gx, fin);
#endif
    if (cin <= 32) return launch_linear_dgrad<4, 1, 1>(rows, cin, cout, dy, w, dx, st, gepi, gx, fin);
    if (cin <= 64) return launch_linear_dgrad<4, 1, 2>(rows, cin, cout, dy, w, dx, st, gepi, gx, fin);
    if (cin <= 96) return launch_linear_dgrad<4, 1, 1>(rows, cin, cout, dy, w, dx, st, gepi, gx, fin);
    const long long cb = (cin + 127) / 128;
    if (((rows + 63) / 64) * cb >= 512) return launch_linear_dgrad<2, 2, 2>(rows, cin, cout, dy, w, dx, st, gepi, gx, fin);
    return launch_linear_dgrad<1, 4, 1>(rows, cin, cout, dy, w, dx, st, gepi, gx, fin);
}

extern "C" int pn2_linear_dgrad(int rows, int cin, int cout, const float* dy, const float* w, float* dx, void* stream) {
    return linear_dgrad_impl(rows, cin, cout, dy, w, dx, stream, Pn2BnGradEpilogue{});
}

// pn2_linear_dgrad whose output dx (rows, cin) is the gradient reaching the batch norm (+ReLU) of the layer BELOW (the
// layer that produced this layer's input): y_below (rows, cin) is that layer's pre-normalisation output, gamma / beta /
// save_mean / save_invstd its batch-norm parameters and saved moments.  While the accumulator tiles of dx are at hand the
// kernel adds sum g and sum g * xhat per channel (g = dx * [relu mask]) to the ZEROED batch-norm workspace of that layer
// (pn2_bn_workspace_bytes(cin)); pn2_bn_relu_backward_stats then skips its reduction pass over (dz, y) -- one of the two
// passes of the reference's batch-norm gradient (util/tf_util.py:555-581 via tf.gradients).  PN2_EUNSUP for cout <= 16
// (streaming kernel): call pn2_linear_dgrad + pn2_bn_relu_backward.
extern "C" int pn2_linear_dgrad_bn_grad_stats(int rows, int cin, int cout, const float* dy, const float* w, float* dx,
                                              const float* y_below, const float* gamma, const float* beta,
                                              const float* save_mean, const float* save_invstd, int relu,
                                              void* bn_workspace, size_t workspace_bytes, void* stream) {
    if (!y_below || !gamma || !beta || !save_mean || !save_invstd || !bn_workspace) return PN2_ENULL;
    if (cin <= 0 || workspace_bytes < sizeof(double) * pn2_bn_ws_doubles(cin, kPn2BnSlots) || ((uintptr_t)bn_workspace % 8) != 0)
        return PN2_EINVAL;
    Pn2BnGradEpilogue e{y_below, gamma, beta, save_mean, save_invstd, static_cast<double*>(bn_workspace), relu};
    return linear_dgrad_impl(rows, cin, cout, dy, w, dx, stream, e);
}

// pn2_linear_dgrad of a layer whose upstream gradient dy = the gradient LEAVING its batch norm (+ReLU [+ max over groups of 32
// rows]) is formed while the GEMM loads it, from y (rows, cout) = this layer's pre-normalisation output, dz = the gradient
// reaching the activation ((rows, cout), or with pool = 32 the (rows / 32, cout) gradient of the pooled maxima beside zmax / ties
// of pn2_bn_relu_forward) and coef (6, cout) of pn2_bn_grad_constants: pn2_bn_relu_backward's second pass (read dz, y; write dy)
// is not run and dy is never written (util/tf_util.py:555-581 + :181-186 via tf.gradients).  y_below != null: the epilogue of
// pn2_linear_dgrad_bn_grad_stats for the layer below as well.  cout % 4 == 0, 16 < cout <= 512, else PN2_EUNSUP.
extern "C" int pn2_linear_dgrad_gx(int rows, int cin, int cout, const float* y, const float* dz, const float* coef, int relu,
                                   int pool, const float* zmax, const float* ties, const float* w, float* dx,
                                   const float* y_below, const float* gamma_below, const float* beta_below,
                                   const float* mean_below, const float* invstd_below, int relu_below, void* ws_below,
                                   size_t ws_below_bytes, void* stream) {
    Pn2BnGradEpilogue e{};
    if (y_below) {
        if (!gamma_below || !beta_below || !mean_below || !invstd_below || !ws_below) return PN2_ENULL;
        if (cin <= 0 || ws_below_bytes < sizeof(double) * pn2_bn_ws_doubles(cin, kPn2BnSlots) || ((uintptr_t)ws_below % 8) != 0)
            return PN2_EINVAL;
        e = Pn2BnGradEpilogue{y_below, gamma_below, beta_below, mean_below, invstd_below, static_cast<double*>(ws_below), relu_below};
    }
    const Pn2GradOnLoad gx{y, dz, coef, zmax, ties, relu, pool};
    return linear_dgrad_impl(rows, cin, cout, nullptr, w, dx, stream, e, &gx);
}

// The data gradient in all its training forms as ONE entry point whose last workgroup also finishes the batch-norm reduction of
// the layer BELOW (pn2_common.h pn2_bn_finish):
//   upstream gradient: dy (rows, cout) given (y == NULL) as in pn2_linear_dgrad, or formed on load from (y, dz, coef, relu, pool,
//     zmax, ties) as in pn2_linear_dgrad_gx (dy == NULL);
//   y_below != NULL: pn2_linear_dgrad_bn_grad_stats' epilogue for the layer below, and finish_below = 1: its slot copies folded
//     (pn2_bn_relu_backward_mode with stats_mode 3 follows), 3: folded AND turned into coef_below (6, cin), dgamma_below,
//     dbeta_below (what pn2_bn_grad_constants publishes), 0: left as they are.
// A layer inside a stack then costs two launches in the backward pass (this + its weight gradient) instead of four.
extern "C" int pn2_linear_dgrad_fin(int rows, int cin, int cout, const float* dy, const float* y, const float* dz,
                                    const float* coef, int relu, int pool, const float* zmax, const float* ties, const float* w,
                                    float* dx, const float* y_below, const float* gamma_below, const float* beta_below,
                                    const float* mean_below, const float* invstd_below, int relu_below, void* ws_below,
                                    size_t ws_below_bytes, int finish_below, float* coef_below, float* dgamma_below,
                                    float* dbeta_below, void* stream) {
    if ((dy == nullptr) == (y == nullptr)) return PN2_EINVAL;  // exactly one form of the upstream gradient
    Pn2BnGradEpilogue e{};
    Pn2BnFinish f{};
    if (y_below) {
        if (!gamma_below || !beta_below || !mean_below || !invstd_below || !ws_below) return PN2_ENULL;
        if (cin <= 0 || ws_below_bytes < sizeof(double) * pn2_bn_ws_doubles(cin, kPn2BnSlots) || ((uintptr_t)ws_below % 8) != 0)
            return PN2_EINVAL;
        e = Pn2BnGradEpilogue{y_below, gamma_below, beta_below, mean_below, invstd_below, static_cast<double*>(ws_below), relu_below};
        if (finish_below != 0 && finish_below != 1 && finish_below != 3) return PN2_EINVAL;
        if (finish_below == 3 && (!coef_below || !dgamma_below || !dbeta_below)) return PN2_ENULL;
        f.kind = finish_below; f.c = cin; f.nslots = kPn2BnSlots; f.rows = rows; f.ws = static_cast<double*>(ws_below);
        f.gamma = gamma_below; f.beta = beta_below; f.mean_in = mean_below; f.invstd_in = invstd_below;
        f.coef = coef_below; f.dgamma = dgamma_below; f.dbeta = dbeta_below;
    } else if (finish_below != 0) {
        return PN2_EINVAL;
    }
    if (y) {
        const Pn2GradOnLoad gx{y, dz, coef, zmax, ties, relu, pool};
        return linear_dgrad_impl(rows, cin, cout, nullptr, w, dx, stream, e, &gx, f.kind ? &f : nullptr);
    }
    if (f.kind && cout <= 16) return PN2_EUNSUP;  // the streaming kernel of the class head has no epilogue
    return linear_dgrad_impl(rows, cin, cout, dy, w, dx, stream, e, nullptr, f.kind ? &f : nullptr);
}

#ifdef PN2_TUNING_HOOKS
extern "C" int pn2_debug_set_linear(int what, int value) {
    if (what == 5) { g_lin_stages = value; return 0; }
    if (what == 8) { g_lin_cfg = value; return 0; }
    if (what == 9) { g_wgrad_waves = value; return 0; }
    if (what == 17) { g_lin_narrow = value; return 0; }
    if (what == 18) { g_wgrad_nwv8 = value; return 0; }
    return PN2_EINVAL;
}
#endif  // PN2_TUNING_HOOKS

// dW = x^T . dy (see linear_wgrad_kernel): the training path's weight gradient.  dw (cin, cout) is overwritten.
#ifndef PN2_WGRAD_MIN_CHUNK
#define PN2_WGRAD_MIN_CHUNK 32   // fewest rows a wave of the weight-gradient kernel contracts (64 until r06: the few-row layers are
                                // bound by their chain of dependent fetch rounds; 32: step 3.187 -> 3.161 ms, 16: the same)
#endif
static int linear_wgrad_impl(int rows, int cin, int cout, const float* x, const float* dy, float* dw, void* stream, bool accumulate,
                             const Pn2LoadTransform* xf = nullptr, const Pn2GradOnLoad* gx = nullptr) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return PN2_EINVAL;
    if (!x || (!dy && !gx) || !dw) return PN2_ENULL;
    if (gx) {
        if (!gx->y || !gx->dz || !gx->coef || (gx->pool && (!gx->zmax || !gx->ties))) return PN2_ENULL;
        if ((gx->pool != 0 && gx->pool != 32) || (gx->pool && rows % 32 != 0)) return PN2_EINVAL;
    }
    if ((long long)rows + 128 > 0x7fffffffLL) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)cin * cout, st);
        if (e != hipSuccess) return (int)e;
    }
    // tile: up to 64 x 128 of dW per wave; rows split so that ~1024 waves (256 blocks) are in flight (>= 64 rows each;
    // measured best of 256..8192 over the training layer shapes, tools/wgrad_sweep.py);
    // each block adds its four partial tiles in LDS and issues one set of atomics
    const int tm = cin > 32 ? 2 : 1, tn = cout > 64 ? 4 : (cout > 32 ? 2 : 1);
    const int gy = (cin + 32 * tm - 1) / (32 * tm), gz = (cout + 32 * tn - 1) / (32 * tn);
    // the 64 x 128 tile over many rows: eight waves per workgroup (twice the waves, the same 256 workgroups and closing atomics)
    const bool wide8 = g_wgrad_nwv8 && tm == 2 && tn == 4 && rows >= PN2_STREAM_MIN_ROWS;
    long long waves = (g_wgrad_waves > 0 ? g_wgrad_waves : (wide8 ? 2048 : 1024)) / ((long long)gy * gz);
    if (waves < 16) waves = 16;
    int chunk = (int)((rows + waves - 1) / waves);
    chunk = (chunk + 7) & ~7;
    if (chunk < PN2_WGRAD_MIN_CHUNK) chunk = PN2_WGRAD_MIN_CHUNK;
    const int nchunks = (rows + chunk - 1) / chunk;
    dim3 grid((nchunks + 3) / 4, gy, gz);
    if (wide8) {
        const dim3 grid8((nchunks + 7) / 8, gy, gz);
        constexpr size_t lds8 = sizeof(float) * 4 * 2 * 4 * 16 * 64;  // 128 KB: beyond the default dynamic limit
#define PN2_WG8(XF_, GX_, XFV_, GXV_)                                                                                              \
        do {                                                                                                                       \
            static bool attr_set = false; /* per instantiation; benign race (idempotent call) */                                  \
            if (!attr_set) {                                                                                                       \
                hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_wgrad_kernel<2, 4, XF_, GX_, 8>),         \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);                       \
                if (e8 != hipSuccess) return (int)e8;                                                                              \
                attr_set = true;                                                                                                   \
            }                                                                                                                      \
            linear_wgrad_kernel<2, 4, XF_, GX_, 8><<<grid8, 512, lds8, st>>>(rows, cin, cout, chunk, x, dy, dw, XFV_, GXV_);       \
        } while (0)
        const Pn2LoadTransform xv = xf ? *xf : Pn2LoadTransform{};
        const Pn2GradOnLoad gv = gx ? *gx : Pn2GradOnLoad{};
        if (xf) {
            if (!gx) PN2_WG8(true, 0, xv, gv);
            else if (gx->pool) PN2_WG8(true, 2, xv, gv);
            else PN2_WG8(true, 1, xv, gv);
        } else {
            if (!gx) PN2_WG8(false, 0, xv, gv);
            else if (gx->pool) PN2_WG8(false, 2, xv, gv);
            else PN2_WG8(false, 1, xv, gv);
        }
#undef PN2_WG8
        PN2_RETURN_IF_LAUNCH_FAILED();
        return PN2_OK;
    }
#define PN2_WG_X(TM_, TN_, XF_, XFV_)                                                                                     \
    do {                                                                                                                  \
        constexpr size_t lds_ = sizeof(float) * 2 * TM_ * TN_ * 16 * 64;                                                   \
        if (!gx) linear_wgrad_kernel<TM_, TN_, XF_><<<grid, 256, lds_, st>>>(rows, cin, cout, chunk, x, dy, dw, XFV_);      \
        else if (gx->pool) linear_wgrad_kernel<TM_, TN_, XF_, 2><<<grid, 256, lds_, st>>>(rows, cin, cout, chunk, x, nullptr, dw, XFV_, *gx); \
        else linear_wgrad_kernel<TM_, TN_, XF_, 1><<<grid, 256, lds_, st>>>(rows, cin, cout, chunk, x, nullptr, dw, XFV_, *gx); \
    } while (0)
#define PN2_WG(TM_, TN_)                                              \
    do {                                                              \
        if (xf) PN2_WG_X(TM_, TN_, true, *xf);                        \
        else PN2_WG_X(TM_, TN_, false, Pn2LoadTransform{});           \
    } while (0)
    if (tm == 1 && tn == 1) PN2_WG(1, 1);
    else if (tm == 1 && tn == 2) PN2_WG(1, 2);
    else if (tm == 1 && tn == 4) PN2_WG(1, 4);
    else if (tm == 2 && tn == 1) PN2_WG(2, 1);
    else if (tm == 2 && tn == 2) PN2_WG(2, 2);
    else PN2_WG(2, 4);
#undef PN2_WG
#undef PN2_WG_X
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_linear_wgrad(int rows, int cin, int cout, const float* x, const float* dy, float* dw, void* stream) {
    return linear_wgrad_impl(rows, cin, cout, x, dy, dw, stream, false);
}
// dw += x^T . dy: the caller owns the initial value (e.g. a gradient arena zero-filled once per step)
extern "C" int pn2_linear_wgrad_accumulate(int rows, int cin, int cout, const float* x, const float* dy, float* dw, void* stream) {
    return linear_wgrad_impl(rows, cin, cout, x, dy, dw, stream, true);
}

// pn2_linear_wgrad_accumulate with x = the pre-normalisation output of the layer below (see pn2_linear_bn_stats_xf)
extern "C" int pn2_linear_wgrad_accumulate_xf(int rows, int cin, int cout, const float* x_raw, const float* dy, float* dw,
                                              const float* a_scale, const float* a_shift, int a_relu, void* stream) {
    if (!a_scale || !a_shift) return PN2_ENULL;
    const Pn2LoadTransform xf{a_scale, a_shift, a_relu};
    return linear_wgrad_impl(rows, cin, cout, x_raw, dy, dw, stream, true, &xf);
}

// dw += x^T . dy with dy = the gradient leaving this layer's batch norm formed on load (see pn2_linear_dgrad_gx; same y / dz /
// coef / pool operands); a_scale != null: x is the pre-normalisation output of the layer below (pn2_linear_wgrad_accumulate_xf).
extern "C" int pn2_linear_wgrad_gx(int rows, int cin, int cout, const float* x, const float* a_scale, const float* a_shift,
                                   int a_relu, const float* y, const float* dz, const float* coef, int relu, int pool,
                                   const float* zmax, const float* ties, float* dw, void* stream) {
    if ((a_scale == nullptr) != (a_shift == nullptr)) return PN2_ENULL;
    const Pn2LoadTransform xf{a_scale, a_shift, a_relu};
    const Pn2GradOnLoad gx{y, dz, coef, zmax, ties, relu, pool};
    return linear_wgrad_impl(rows, cin, cout, x, nullptr, dw, stream, true, a_scale ? &xf : nullptr, &gx);
}

// internal helper (exported for the host package's unfused SA path and for tests)
extern "C" int pn2_sa_group_concat(int b, int n, int m, int nsample, int c, const float* xyz,
                                   const float* new_xyz, const float* points, const int* idx,
                                   float* out, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || c < 0) return PN2_EINVAL;
    if (!xyz || !new_xyz || !idx || !out || (c > 0 && !points)) return PN2_ENULL;
    const unsigned long long total = (unsigned long long)m * nsample * (3 + c);
    if (total > 0xffffffffull || b > 65535) return PN2_ERANGE;
    unsigned long long g = (total + 255) / 256;
    unsigned long long cap = (256ull * 8 + b - 1) / b;
    if (g > cap) g = cap;
    dim3 grid((unsigned)g, b);
    sa_group_concat_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
        n, m, nsample, c, xyz, new_xyz, c > 0 ? points : nullptr, idx, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
