// pn2_linear_wres.h -- the dense layer of pn2_linear.hip for MANY rows and a SHORT contraction (cin <= 256), with the weight
// panel resident in LDS.  (The reference: tf.nn.conv2d 1x1 + bias_add + batch_norm + relu, util/tf_util.py:181-203, and its
// gradient through tf.gradients.)
//
// linear_kernel gives every 64 x 128 output tile its own workgroup: at 131072 x 128 -> 128 that is 2048 workgroups of FOUR
// k-tiles each, every one of them re-staging the whole 64 KB weight matrix through LDS behind two barriers per k-tile, with a
// prologue and an epilogue nothing overlaps -- 64-77 us for a layer whose matrix-pipe time and HBM time are both ~27 us.
// Here:
//   * a workgroup stages its weight panel (cin x 128 columns, at most 128 KB) ONCE, already in the order the MFMA wants it:
//     fragment (T, nt) = the 64 x 16 bytes a wave reads with ONE ds_read_b128 per lane for the four steps k = 8T + 4*half + q,
//     q = 0..3, of column block nt (conflict-free: consecutive lanes, consecutive 16 bytes);
//   * it is persistent over row tiles, and its four waves are independent: wave w owns 32-row tiles w, w + 4G, ... and there is
//     NO barrier after the panel is staged;
//   * the A operand never touches LDS: lane (row = l & 31, half = l >> 5) loads the four k = 8T + 4*half + {0..3} of its row with
//     one 16-byte load straight into the register the MFMA reads -- v_mfma_f32_32x32x2_f32 only needs A and B to agree on the
//     k of each half, and this is the visiting order of linear_kernel, so the two kernels return the SAME BITS;
//   * the loads of the next chunk (128 contraction indices of the next tile) are issued before the MFMAs of this one: 16 KB per
//     wave in flight under 256 MFMAs; the operand transforms of the training step (the batch norm of the layer below applied
//     on load; the batch-norm gradient formed on load) run on the registers, with their per-channel constants in LDS.
// Epilogues are linear_kernel's (bias / ReLU / max over a 32-row group / batch-norm sums from the accumulators / the
// last-workgroup finish).
#pragma once
#include "pn2_common.h"
#include <type_traits>

#include "pn2_mfma_stats.h"

namespace {

// KC8: 16-byte loads per lane and chunk (8: 64 contraction indices per chunk, 16: 128); NT: 32-column blocks per workgroup;
// TB: B(k, n) = w[n * cin + k] (data gradient); XF / GX: linear_kernel's operand transforms (GX = 1 only); OCC: workgroups per CU
template <int KC8, int NT, bool TB, bool XF, int GX, int OCC>
__global__ void __launch_bounds__(256, OCC)
linear_wres_kernel(int rows, int cin, int cout, const float* __restrict__ x, const float* __restrict__ w,
                   const float* __restrict__ bias, int relu, int pool, float* __restrict__ y, double* __restrict__ stats,
                   Pn2BnGradEpilogue gepi, Pn2LoadTransform xf, Pn2GradOnLoad gx, Pn2BnFinish fin) {
    static_assert(GX == 0 || GX == 1, "the pooled gradient-on-load form stays with linear_kernel");
    static_assert(!(XF && GX), "one operand transform");
    constexpr int BN = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float wres_lds[];
    float* __restrict__ Bp = wres_lds;                              // (cin / 8, NT, 64 lanes, 4)
    float* __restrict__ aux = wres_lds + (size_t)(cin >> 3) * NT * 256;  // XF: scale[cin] | shift[cin];  GX: coef (6, cin)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int col0 = blockIdx.y * BN;
    const int ntiles = (rows + 31) >> 5;
    const int nch = cin / (KC8 * 8);
    const int tstride = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;

    const float* __restrict__ asrc = GX ? gx.y : x;
    // TWO operand buffers, addressed statically (the item loop below is unrolled by two): the loads of item i + 1 land in the
    // other buffer while the MFMAs of item i read this one.  (One buffer lets the compiler issue the next loads only after the
    // last MFMA that reads the register they overwrite: the prefetch then overlaps nothing.)
    f32x4 pa[2][KC8];
    f32x4 pg[2][GX ? KC8 : 1];
    // every load is unconditional (tile / row clamped into the buffer): counted waits keep the chunk in flight under the MFMAs
    auto issue = [&](auto uc, int t, int c) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        const int tt = t < ntiles ? t : ntiles - 1;
        int row = tt * 32 + l31;
        row = row < rows ? row : rows - 1;
        const size_t off = (size_t)row * cin + c * (KC8 * 8) + 4 * half;
#pragma unroll
        for (int T = 0; T < KC8; ++T) pa[u][T] = *reinterpret_cast<const f32x4*>(asrc + off + 8 * T);
        if constexpr (GX != 0) {
#pragma unroll
            for (int T = 0; T < KC8; ++T) pg[u][T] = *reinterpret_cast<const f32x4*>(gx.dz + off + 8 * T);
        }
    };
    const int my_tiles = tile < ntiles ? (ntiles - tile + tstride - 1) / tstride : 0;
    const int nitems = my_tiles * nch;  // an item = one chunk (KC8 * 8 contraction indices) of one 32-row tile
    if (nitems > 0) issue(std::integral_constant<int, 0>{}, tile, 0);

    // ---- the weight panel, in fragment order (loads in batches of eight 16-byte reads per thread, then the LDS stores) ----
    {
        constexpr int FB = 8;
        const int k4n = cin >> 2;
        const int total = TB ? BN * k4n : cin * (BN / 4);
        for (int e0 = 0; e0 < total; e0 += 256 * FB) {
            f32x4 v[FB];
#pragma unroll
            for (int i = 0; i < FB; ++i) {
                const int e = e0 + tid + 256 * i;
                const int ec = e < total ? e : total - 1;
                if constexpr (TB) {
                    const int n = ec / k4n, k = (ec - n * k4n) * 4;
                    const int nc = col0 + n < cout ? col0 + n : cout - 1;
                    v[i] = *reinterpret_cast<const f32x4*>(w + (size_t)nc * cin + k);
                    if (col0 + n >= cout) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    const int k = ec / (BN / 4), n4 = ec - k * (BN / 4);
                    v[i] = *reinterpret_cast<const f32x4*>(w + (size_t)k * cout + col0 + n4 * 4);
                }
            }
#pragma unroll
            for (int i = 0; i < FB; ++i) {
                const int e = e0 + tid + 256 * i;
                if (e < total) {
                    if constexpr (TB) {
                        const int n = e / k4n, k = (e - n * k4n) * 4;
                        *reinterpret_cast<f32x4*>(Bp + (((((k >> 3) * NT + (n >> 5)) * 64) + ((k >> 2) & 1) * 32 + (n & 31)) << 2)) = v[i];
                    } else {
                        const int k = e / (BN / 4), n4 = e - k * (BN / 4);
                        const int base = ((k >> 3) * NT * 64 + ((k >> 2) & 1) * 32) * 4 + (k & 3);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int n = n4 * 4 + j;
                            Bp[base + ((n >> 5) * 64 + (n & 31)) * 4] = v[i][j];
                        }
                    }
                }
            }
        }
    }
    if constexpr (XF) {
        for (int e = tid; e < cin; e += 256) { aux[e] = xf.scale[e]; aux[cin + e] = xf.shift[e]; }
    }
    if constexpr (GX != 0) {
        for (int e = tid; e < 6 * cin; e += 256) aux[e] = gx.coef[e];
    }
    __syncthreads();

    f32x16 acc[NT];
    int c = 0;  // chunk of the current item within its tile
    auto process = [&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        const int r0 = tile * 32;
        const bool rowv = r0 + l31 < rows;
        if constexpr (XF || GX != 0) {  // the operand transform, in place
#pragma unroll
            for (int T = 0; T < KC8; ++T) {
                f32x4 v = pa[u][T];
                const int kk = (c * KC8 + T) * 8 + 4 * half;  // this lane's four contraction indices of fragment T
                if constexpr (XF) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(aux + kk);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(aux + cin + kk);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = __builtin_fmaf(v[q], sc[q], sh[q]);
                        v[q] = xf.relu ? fmaxf(t, 0.f) : t;
                    }
                } else {
                    f32x4 gc[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) gc[j] = *reinterpret_cast<const f32x4*>(aux + j * cin + kk);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        v[q] = pn2_bn_grad_element(v[q], pg[u][T][q], gc[0][q], gc[1][q], gc[2][q], gc[3][q], gc[4][q], gc[5][q], gx.relu);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = rowv ? v[q] : 0.f;  // rows past the end add nothing (sums; stores are guarded)
                pa[u][T] = v;
            }
        }
        {   // the next item: the next chunk of this tile, or the first one of this wave's next tile
            int nc = c + 1, nt_ = tile;
            if (nc == nch) { nc = 0; nt_ = tile + tstride; }
            issue(std::integral_constant<int, u ^ 1>{}, nt_, nc);
        }
        if (c == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        }
        const float* __restrict__ bfrag = Bp + ((size_t)c * KC8 * NT * 64 + lane) * 4;
#pragma unroll
        for (int T = 0; T < KC8; ++T) {
            f32x4 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(bfrag + (T * NT + nt) * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[u][T][q], b[nt][q], acc[nt], 0, 0, 0);
        }
        if (c + 1 < nch) { ++c; return; }
        c = 0;
        // ---- epilogue: D[i][j], j = l31, i = (r & 3) + 8 * (r >> 2) + 4 * half ---------------------------------------
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + nt * 32 + l31;
            if constexpr (TB) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < rows && col < cout) y[(size_t)row * cout + col] = acc[nt][r];
                }
                if (gepi.ws) push_column_grad_stats(acc[nt], half, r0, rows, col, cout, (unsigned)tile, gepi);
            } else {
                const float bv = bias ? bias[col] : 0.f;
                if (pool <= 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float v = acc[nt][r] + bv;
                        if (relu) v = fmaxf(v, 0.f);
                        if (row < rows) y[(size_t)row * cout + col] = v;
                    }
                    if (stats) push_column_stats(acc[nt], half, col, cout, (unsigned)tile, stats);
                } else {  // pool == 32: the wave's tile is one group
                    float v = acc[nt][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[nt][r]);
                    v = fmaxf(v, __shfl_xor(v, 32));
                    v += bv;  // max_i relu(x_i + b) == relu(max_i(x_i) + b): fl(x + b) and relu are monotone
                    if (relu) v = fmaxf(v, 0.f);
                    if (half == 0) y[(size_t)tile * cout + col] = v;
                }
            }
        }
        tile += tstride;
    };
    for (int it = 0; it < nitems; it += 2) {
        process(std::integral_constant<int, 0>{});
        if (it + 1 < nitems) process(std::integral_constant<int, 1>{});
    }
    // the last workgroup folds the batch-norm sums this launch has left (and derives the constants): pn2_common.h
    pn2_bn_finish(fin, gridDim.x * gridDim.y, blockIdx.x + gridDim.x * blockIdx.y);
}

// Shapes the resident-weight kernel takes: contraction K in {64, 128, 192, 256}, n output columns a multiple of 64, enough 32-row
// tiles to give every SIMD of the chip a wave, 16-byte aligned operands, pool in {0, 1, 32}.
inline bool wres_fits(int rows, int K, int n, int pool, const void* a, const void* w_, bool tb) {
    if (K < 64 || K > 256 || K % 64 != 0) return false;
    if (n % 64 != 0) return false;
    if (pool > 1 && (pool != 32 || rows % 32 != 0)) return false;
    if ((((uintptr_t)a | (uintptr_t)w_) % 16) != 0) return false;
    const int bn = n % 128 == 0 ? 128 : 64;
    const long long wave_tiles = (long long)((rows + 31) / 32) * (n / bn);
    (void)tb;
    return wave_tiles >= 1024;
}

template <int KC8, int NT, bool TB, bool XF, int GX, int OCC>
int launch_wres_one(int rows, int K, int n, const float* x, const float* w, const float* bias, int relu, int pool, float* y,
                    hipStream_t st, double* stats, const Pn2BnGradEpilogue& gepi, const Pn2LoadTransform& xf,
                    const Pn2GradOnLoad& gx, const Pn2BnFinish& fin) {
    constexpr int BN = 32 * NT;
    const size_t lds = sizeof(float) * ((size_t)K * BN + (XF ? 2 * (size_t)K : 0) + (GX ? 6 * (size_t)K : 0));
    auto kern = linear_wres_kernel<KC8, NT, TB, XF, GX, OCC>;
    static bool attr_set = false;  // per instantiation; benign race (idempotent call)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);  // (+ the finish's static word)
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int gy = n / BN;
    const int ntiles = (rows + 31) / 32;
    const int per_cu = lds * 2 <= 150 * 1024 ? OCC : 1;  // workgroups the LDS lets a CU hold
    int G = (256 * per_cu) / gy;
    if (G < 1) G = 1;
    if (G > (ntiles + 3) / 4) G = (ntiles + 3) / 4;
    kern<<<dim3(G, gy), 256, lds, st>>>(rows, K, n, x, w, bias, relu, pool, y, stats, gepi, xf, gx, fin);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// K = contraction length, n = output columns (see wres_fits)
template <bool TB, bool XF, int GX>
int launch_wres(int rows, int K, int n, const float* x, const float* w, const float* bias, int relu, int pool, float* y,
                hipStream_t st, double* stats, const Pn2BnGradEpilogue& gepi, const Pn2LoadTransform& xf, const Pn2GradOnLoad& gx,
                const Pn2BnFinish* fin) {
    const Pn2BnFinish f = fin ? *fin : Pn2BnFinish{};
    constexpr int OCC = GX ? 1 : 2;  // the gradient-on-load form keeps two operand streams in registers: one wave per SIMD
#define PN2_WRES(KC8_, NT_) launch_wres_one<KC8_, NT_, TB, XF, GX, OCC>(rows, K, n, x, w, bias, relu, pool, y, st, stats, gepi, xf, gx, f)
    // chunks of 64 contraction indices: 8 KB per wave in flight under 128 MFMAs (chunks of 128 -- twice the registers, twice the
    // bytes in flight -- measured slower: 65 vs 48 us at 131072 x 128 -> 128)
    return n % 128 == 0 ? PN2_WRES(8, 4) : PN2_WRES(8, 2);
#undef PN2_WRES
}

}  // namespace
