// pn2_bwd_fused.hip -- data gradient AND weight gradient of a NARROW dense layer of the training path in one launch.
//
// The backward pass of a dense + batch-norm layer inside a stack (reference: tf.gradients through tf_util.conv2d ->
// batch_norm_template -> relu, util/tf_util.py:181-204,555-581) reads the same rows three times over when its two GEMMs are two
// launches: pn2_linear_dgrad_gx reads (y, dz) to form dy and writes dx, pn2_linear_wgrad_gx reads (y, dz) AGAIN beside x.  For
// the wide layers (128 channels) both GEMMs are bound by the matrix pipe and nothing is lost; for the narrow ones of the first two
// SA modules (32 / 64 channels over 524288 / 131072 rows) both are plain HBM streams, and the second read of (y, dz) is a third
// of the traffic.  Here a WAVE owns a tile of 32 rows:
//   dy (32 x cout)  formed from (y, dz) and the batch-norm gradient constants (Pn2GradOnLoad) -> the wave's LDS tile Ds,
//   dx (32 x cin)   = Ds . W^T        A = Ds read along the contraction (one ds_read_b128 per four k), B = W^T staged once per
//                                     workgroup in LDS (k-major), + the batch-norm gradient sums of the layer BELOW from the
//                                     accumulator tiles (Pn2BnGradEpilogue), + the finish of that reduction (Pn2BnFinish),
//   dW (cin x cout) += x^T . Ds       A = x straight from global memory in MFMA layout (lane = input channel, the batch norm of
//                                     the layer below applied on load when it was deferred), B = Ds read along the rows,
// all on v_mfma_f32_32x32x2_f32 (exact fp32 products).  No workgroup barrier inside the loop (a wave's LDS traffic is in order);
// the four waves' dW tiles are added in LDS at the end and leave with one set of atomics per workgroup.
#include "pn2_common.h"
#include "pn2_mfma_stats.h"

namespace {

// CI = cin / 32, CO = cout / 32 (1 or 2); GX = 1: dz (rows, cout), 2: pooled over groups of 32 rows; XF: x is the un-normalised
// output of the layer below
template <int CI, int CO, int GX, bool XF>
__global__ void __launch_bounds__(256, 2)
bwd_narrow_kernel(int rows, const float* __restrict__ x, Pn2LoadTransform xf, Pn2GradOnLoad gx, const float* __restrict__ w,
                  float* __restrict__ dx, float* __restrict__ dw, Pn2BnGradEpilogue gepi, Pn2BnFinish fin) {
    constexpr int CIN = 32 * CI, COUT = 32 * CO;
    constexpr int WS = CIN + 4;   // row stride of Wt (k-major: Wt[k][n] = w[n][k])
    constexpr int DS = COUT + 4;  // row stride of a wave's dy tile
    constexpr int C4 = COUT / 4;  // float4 columns of a dy row
    constexpr int NF = 32 * C4 / 64;  // float4 of a tile per lane
    __shared__ __attribute__((aligned(16))) float Wt[COUT * WS];
    __shared__ __attribute__((aligned(16))) float Dall[4 * 32 * DS];
    static_assert(4 * 32 * DS >= CI * CO * 16 * 64, "the dW reduction reuses the dy tiles");
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* __restrict__ Ds = Dall + wave * (32 * DS);

    for (int e = tid; e < CIN * COUT; e += 256) {  // w (CIN, COUT) row-major -> Wt[k][n]
        const int n = e / COUT, k = e - n * COUT;
        Wt[k * WS + n] = w[e];
    }
    // this lane's four dy columns are the same for every tile (64 % C4 == 0): their constants stay in registers
    const int c4 = lane % C4;
    f32x4 gc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) gc[j] = *reinterpret_cast<const f32x4*>(gx.coef + (size_t)j * COUT + c4 * 4);
    float xsc[CI], xsh[CI];
#pragma unroll
    for (int a = 0; a < CI; ++a) {
        xsc[a] = XF ? xf.scale[a * 32 + l31] : 1.f;
        xsh[a] = XF ? xf.shift[a * 32 + l31] : 0.f;
    }
    f32x16 acc_dw[CI][CO];
#pragma unroll
    for (int a = 0; a < CI; ++a)
#pragma unroll
        for (int b = 0; b < CO; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_dw[a][b][r] = 0.f;
    __syncthreads();  // Wt staged

    const int ntiles = rows / 32;
    // (y, dz) of a tile are fetched one tile AHEAD: they are the only loads a tile waits for at its top (x is issued before the
    // dx MFMAs and used after them), and with two waves per SIMD nothing else hides that round trip
    f32x4 yv[NF], gv[GX == 1 ? NF : 1];
    f32x4 pd, pm, pn;
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const int tc = tile < ntiles ? tile : ntiles - 1;
        const int row0 = tc * 32;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int r = (lane + 64 * i) / C4;
            yv[i] = *reinterpret_cast<const f32x4*>(gx.y + (size_t)(row0 + r) * COUT + c4 * 4);
            if constexpr (GX == 1) gv[i] = *reinterpret_cast<const f32x4*>(gx.dz + (size_t)(row0 + r) * COUT + c4 * 4);
        }
        if constexpr (GX == 2) {  // the tile IS one pooling group
            pd = *reinterpret_cast<const f32x4*>(gx.dz + (size_t)tc * COUT + c4 * 4);
            pm = *reinterpret_cast<const f32x4*>(gx.zmax + (size_t)tc * COUT + c4 * 4);
            pn = *reinterpret_cast<const f32x4*>(gx.ties + (size_t)tc * COUT + c4 * 4);
        }
    };
    const int tstep = gridDim.x * 4;
    if ((int)(blockIdx.x * 4 + wave) < ntiles) fetch(blockIdx.x * 4 + wave);
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += tstep) {
        const int row0 = tile * 32;
        // x in MFMA layout for the weight gradient: k-step j contracts rows 2j (lanes 0-31) and 2j + 1 (lanes 32-63)
        float xv[16][CI];
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int a = 0; a < CI; ++a) xv[j][a] = x[(size_t)(row0 + 2 * j + half) * CIN + a * 32 + l31];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int r = (lane + 64 * i) / C4;
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (GX == 2)
                    v[q] = pn2_bn_grad_element_pooled(yv[i][q], pd[q], pm[q], pn[q], gc[0][q], gc[1][q], gc[2][q], gc[3][q], gc[4][q],
                                                      gc[5][q], gx.relu);
                else
                    v[q] = pn2_bn_grad_element(yv[i][q], gv[i][q], gc[0][q], gc[1][q], gc[2][q], gc[3][q], gc[4][q], gc[5][q], gx.relu);
            }
            *reinterpret_cast<f32x4*>(Ds + r * DS + c4 * 4) = v;
        }
        fetch(tile + tstep);  // the next tile's (y, dz) travel under this tile's MFMAs (clamped past the end: never used)
        __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations execute in issue order: the reads below see these writes)
        // ---- dx tile = Ds . W^T
        f32x16 acc_dx[CI];
#pragma unroll
        for (int nt = 0; nt < CI; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_dx[nt][r] = 0.f;
        const float* as = Ds + l31 * DS + 4 * half;
        const float* bs = Wt + (4 * half) * WS + l31;
#pragma unroll
        for (int t = 0; t < COUT / 8; ++t) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(as + 8 * t);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < CI; ++nt)
                    acc_dx[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bs[(8 * t + q) * WS + nt * 32], acc_dx[nt], 0, 0, 0);
        }
        // ---- dW += x^T . Ds
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float bv[CO];
#pragma unroll
            for (int b = 0; b < CO; ++b) bv[b] = Ds[(2 * j + half) * DS + b * 32 + l31];
#pragma unroll
            for (int a = 0; a < CI; ++a) {
                float av = xv[j][a];
                if constexpr (XF) {
                    av = __builtin_fmaf(av, xsc[a], xsh[a]);
                    av = xf.relu ? fmaxf(av, 0.f) : av;
                }
#pragma unroll
                for (int b = 0; b < CO; ++b) acc_dw[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[b], acc_dw[a][b], 0, 0, 0);
            }
        }
        // ---- dx out (+ the batch-norm gradient sums of the layer below from the accumulator tiles)
#pragma unroll
        for (int nt = 0; nt < CI; ++nt) {
            const int col = nt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                dx[(size_t)row * CIN + col] = acc_dx[nt][r];
            }
            if (gepi.ws) push_column_grad_stats(acc_dx[nt], half, row0, rows, col, CIN, (unsigned)tile, gepi);
        }
        __builtin_amdgcn_wave_barrier();  // the next tile's dy overwrites Ds: every read above has been issued
    }
    // ---- the four waves' dW tiles meet in LDS (the dy tiles are free now), one set of atomics per workgroup
    __syncthreads();
    float* __restrict__ red = Dall;
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int a = 0; a < CI; ++a)
#pragma unroll
                for (int b = 0; b < CO; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* p = red + ((a * CO + b) * 16 + r) * 64 + lane;
                        *p = wv == 0 ? acc_dw[a][b][r] : *p + acc_dw[a][b][r];
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < CI * CO * 16 * 64; e += 256) {
        const int ln = e & 63, r = (e >> 6) & 15, tl = e >> 10;
        const int a = tl / CO, b = tl - a * CO;
        const int m = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), n = b * 32 + (ln & 31);
        atomicAdd(&dw[(size_t)m * COUT + n], red[e]);
    }
    pn2_bn_finish(fin, gridDim.x, blockIdx.x);
}

template <int CI, int CO>
int launch_bwd_narrow(int rows, const float* x, const Pn2LoadTransform* xf, const Pn2GradOnLoad& gx, const float* w, float* dx,
                      float* dw, const Pn2BnGradEpilogue& e, const Pn2BnFinish& f, hipStream_t st) {
    const int ntiles = rows / 32;
    int blocks = (ntiles + 3) / 4;
    if (blocks > 512) blocks = 512;  // two workgroups per CU; every workgroup ends in one set of dW atomics
    const Pn2LoadTransform t = xf ? *xf : Pn2LoadTransform{};
#define PN2_BWN(GX_, XF_) bwd_narrow_kernel<CI, CO, GX_, XF_><<<blocks, 256, 0, st>>>(rows, x, t, gx, w, dx, dw, e, f)
    if (gx.pool) {
        if (xf) PN2_BWN(2, true);
        else PN2_BWN(2, false);
    } else {
        if (xf) PN2_BWN(1, true);
        else PN2_BWN(1, false);
    }
#undef PN2_BWN
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

}  // namespace

// dx (rows, cin) = dy . W^T AND dw (cin, cout) += x^T . dy in one launch for a narrow layer: cin, cout in {32, 64}, rows % 32 == 0,
// 16-byte aligned y / dz / coef (else PN2_EUNSUP: call pn2_linear_dgrad_fin + pn2_linear_wgrad_gx).  dy = the gradient leaving the
// layer's batch norm (+ReLU [+ max over groups of 32 rows]) formed on load from (y, dz, coef) as pn2_linear_dgrad_gx forms it;
// a_scale != NULL: x is the un-normalised output of the layer below (pn2_linear_wgrad_accumulate_xf); y_below != NULL: the
// epilogue and the finish of pn2_linear_dgrad_fin for the layer below (finish_below 0 / 1 / 3).  dw is ADDED to (zero it first).
extern "C" int pn2_linear_bwd_fused(int rows, int cin, int cout, const float* x, const float* a_scale, const float* a_shift,
                                    int a_relu, const float* y, const float* dz, const float* coef, int relu, int pool,
                                    const float* zmax, const float* ties, const float* w, float* dx, float* dw,
                                    const float* y_below, const float* gamma_below, const float* beta_below,
                                    const float* mean_below, const float* invstd_below, int relu_below, void* ws_below,
                                    size_t ws_below_bytes, int finish_below, float* coef_below, float* dgamma_below,
                                    float* dbeta_below, void* stream) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return PN2_EINVAL;
    if (!x || !y || !dz || !coef || !w || !dx || !dw) return PN2_ENULL;
    if ((a_scale == nullptr) != (a_shift == nullptr)) return PN2_ENULL;
    if ((pool != 0 && pool != 32) || (pool && (!zmax || !ties))) return pool && (!zmax || !ties) ? PN2_ENULL : PN2_EINVAL;
    if ((cin != 32 && cin != 64) || (cout != 32 && cout != 64) || rows % 32 != 0) return PN2_EUNSUP;
    if ((((uintptr_t)y | (uintptr_t)dz | (uintptr_t)coef | (uintptr_t)zmax | (uintptr_t)ties) % 16) != 0) return PN2_EUNSUP;
    if ((long long)rows + 128 > 0x7fffffffLL) return PN2_ERANGE;
    Pn2BnGradEpilogue e{};
    Pn2BnFinish f{};
    if (y_below) {
        if (!gamma_below || !beta_below || !mean_below || !invstd_below || !ws_below) return PN2_ENULL;
        if (ws_below_bytes < sizeof(double) * pn2_bn_ws_doubles(cin, kPn2BnSlots) || ((uintptr_t)ws_below % 8) != 0) return PN2_EINVAL;
        e = Pn2BnGradEpilogue{y_below, gamma_below, beta_below, mean_below, invstd_below, static_cast<double*>(ws_below), relu_below};
        if (finish_below != 0 && finish_below != 1 && finish_below != 3) return PN2_EINVAL;
        if (finish_below == 3 && (!coef_below || !dgamma_below || !dbeta_below)) return PN2_ENULL;
        f.kind = finish_below; f.c = cin; f.nslots = kPn2BnSlots; f.rows = rows; f.ws = static_cast<double*>(ws_below);
        f.gamma = gamma_below; f.beta = beta_below; f.mean_in = mean_below; f.invstd_in = invstd_below;
        f.coef = coef_below; f.dgamma = dgamma_below; f.dbeta = dbeta_below;
    } else if (finish_below != 0) {
        return PN2_EINVAL;
    }
    const Pn2LoadTransform xf{a_scale, a_shift, a_relu};
    const Pn2GradOnLoad gx{y, dz, coef, zmax, ties, relu, pool};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Pn2LoadTransform* pxf = a_scale ? &xf : nullptr;
    if (cin == 32 && cout == 32) return launch_bwd_narrow<1, 1>(rows, x, pxf, gx, w, dx, dw, e, f, st);
    if (cin == 32 && cout == 64) return launch_bwd_narrow<1, 2>(rows, x, pxf, gx, w, dx, dw, e, f, st);
    if (cin == 64 && cout == 32) return launch_bwd_narrow<2, 1>(rows, x, pxf, gx, w, dx, dw, e, f, st);
    return launch_bwd_narrow<2, 2>(rows, x, pxf, gx, w, dx, dw, e, f, st);
}
