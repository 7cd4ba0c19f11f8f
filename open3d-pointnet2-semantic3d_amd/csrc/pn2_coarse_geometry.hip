// pn2_coarse_geometry.hip -- the sampling / grouping / 3-NN tables of the COARSE levels of the pyramid in ONE launch.
//
// The reference builds every set-abstraction level with three ops of its own -- farthest_point_sample + gather_point,
// query_ball_point (util/pointnet_util.py:36-39, called level after level by model.py:104-113) -- and every feature-
// propagation level with a three_nn (util/pointnet_util.py:300).  Below the first level the clouds are tiny (1024 -> 256 ->
// 64 -> 16 points per cloud in semantic.json): nine launches of 6-13 us each for ~25 us of work, all of them on the critical
// path of a single batch ("Latency").  They depend on coordinates only, and every level's input is the
// previous level's output, so one launch can walk down the pyramid:
//
//   for each level l (source cloud = the level above, n points; m = npoint of the level):
//     sample       pn2_fps_nested's decision on the device: the first m rows when the parent run's tie record allows it
//                  (the common case: no sampling at all), else the register-resident sampler itself (pn2fpsreg::fps_reg_body,
//                  the code fps_reg_kernel runs), which also leaves the tie record for the level below;
//     ball query   one wave per query, 64 candidates per step in index order, hits appended in lane order (= the first
//                  nsample in index order, tf_grouping.cu:18-37), short rows padded with the first hit, empty rows zero --
//                  the bits of pn2_query_ball_point;
//     three_nn     queries = the source cloud, known points = the m samples: pn2nn::three_nn_wave, the routine of
//                  pn2_three_nn (exact float64 ranking, ties -> lowest index; tf_interpolate.cpp:213-243).
//
// Every output is bit-identical to the separate entry points (tests/test_coarse_geometry_gpu.py).
//
// Parallelism.  The searches of a level are latency chains (a 3-NN query group ~4 us, a ball query ~1 us per wave), so one
// workgroup per cloud would take ~100 us (measured).  R workgroups of 4 waves share a cloud with NO communication between
// them: with the shortcut the samples of every level are prefixes of xyz0, known to everybody; without it every workgroup
// runs the (deterministic) sampler itself and all of them write the same picks -- redundant work in the rare tie case
// instead of a flag to spin on.  Work items (3-NN query groups first, then ball queries, all levels in one index space) are
// dealt round-robin to the R x 4 waves of the cloud.  HBM traffic is a few KB per cloud; the kernel is latency-bound.
#include <math.h>

#include "pn2_fps_reg.h"
#include "pn2_three_nn.h"

namespace {

using namespace pn2fpsreg;
using namespace pn2nn;

constexpr int kCgThreads = 256;
constexpr int kCgWaves = kCgThreads / 64;
constexpr int kCgPPT = 4;       // sampler: points per thread
constexpr int kCgMaxLevels = 4;
constexpr int kCgMaxN = kCgThreads * kCgPPT;  // points of the source cloud of the first level
constexpr int kCgMaxR = 64;     // workgroups per cloud at most
constexpr int kCgMaxNnM = 256;  // samples of a level that asks for a 3-NN table

struct CgLevel {
    int m, nsample;
    float thr;       // ball threshold on the squared distance (pn2_ball_threshold)
    int* fps_idx;    // (b, m)
    float* new_xyz;  // (b, m, 3)
    int* bq_idx;     // (b, m, nsample)
    int* bq_cnt;     // (b, m) or null
    float* nn_dist;  // (b, n, 3), n = points of the level above; null = no 3-NN table for this level
    int* nn_idx;     // (b, n, 3)
};
struct CgParams {
    int nlev, n0;
    const float* xyz0;  // (b, n0, 3)
    const int* tie_in;  // (b) or null
    int* tie_out;       // (b) or null: record of the LAST level
    CgLevel lv[kCgMaxLevels];
};

// LDS: the sampler's scratch (phase 1) overlaps the source-cloud copy + the 3-NN wave scratch (phase 2); then the level
// table (source / sample pointers of every level, written in phase 1) and the tie record
constexpr int kCgFpsBytes = (kFpsRegHead + 16 * kCgMaxN + 4 * kCgMaxN + 15) & ~15;
constexpr int kCgPhase2Bytes = 16 * kCgMaxN + kCgWaves * kNnWaveLdsBytes;
constexpr int kCgUnionBytes = kCgFpsBytes > kCgPhase2Bytes ? kCgFpsBytes : kCgPhase2Bytes;
struct CgTable { const float* src[kCgMaxLevels]; const float* nw[kCgMaxLevels]; int n[kCgMaxLevels]; int tie; int pad; };
constexpr int kCgLdsBytes = kCgUnionBytes + (int)sizeof(CgTable);

template <int FMODE, int BMODE>
__global__ void __launch_bounds__(kCgThreads)
coarse_geometry_kernel(CgParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* fps_smem = smem;
    float4* sxyz = reinterpret_cast<float4*>(smem);
    unsigned char* nn_smem = smem + 16 * kCgMaxN;
    CgTable* tab = reinterpret_cast<CgTable*>(smem + kCgUnionBytes);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.y, r = blockIdx.x;

    // ---- phase 1: the samples of every level (pn2_fps_nested's decision, per cloud) ------------------------------------
    {
        int n = p.n0;
        const float* src = p.xyz0 + (size_t)c * n * 3;  // (no __restrict__: a level's samples are the next level's source cloud)
        if (tid == 0) tab->tie = p.tie_in ? p.tie_in[c] : 0;  // 0: nothing is known about the source cloud -> sample
        __syncthreads();
#pragma unroll 1
        for (int l = 0; l < p.nlev; ++l) {
            const CgLevel L = p.lv[l];
            const int m = L.m;
            int* out = L.fps_idx + (size_t)c * m;
            float* nx = L.new_xyz + (size_t)c * m * 3;
            const int T = tab->tie;
            const float* nw;
            if (T >= m && m < n) {  // the first m rows; workgroup 0 of the cloud writes them out (m == n: see fps_nested_shortcut_b)
                if (r == 0) {
                    for (int jj = tid; jj < m; jj += kCgThreads) out[jj] = jj;
                    for (int e = tid; e < m * 3; e += kCgThreads) nx[e] = src[e];
                }
                nw = src;
            } else {  // every workgroup of the cloud samples (same picks, same stores) and leaves the record for the level below
                __syncthreads();  // everybody has read the old record
                fps_reg_body<kCgThreads, kCgPPT, FMODE, true, true>(n, m, src, out, nx, &tab->tie, fps_smem);
                nw = nx;
            }
            if (tid == 0) { tab->src[l] = src; tab->nw[l] = nw; tab->n[l] = n; }
            __syncthreads();  // out / nx (global), the table and the record (LDS) are visible to the workgroup
            src = nw;
            n = m;
        }
        if (p.tie_out && tid == 0 && r == 0) p.tie_out[c] = tab->tie;
    }
    // ---- phase 2: the searches, dealt round-robin over the waves of the cloud's workgroups ------------------------------
    const int GW = gridDim.x * kCgWaves, gw = r * kCgWaves + wave;
    int base = 0;
    const NnWaveLds S = nn_wave_lds(nn_smem, wave);
    // three_nn: queries = the source cloud, known points = the samples (groups of kNnQ queries)
#pragma unroll 1
    for (int l = 0; l < p.nlev; ++l) {
        const CgLevel L = p.lv[l];
        if (!L.nn_dist) continue;
        const int n = tab->n[l], m = L.m;
        const float* src = tab->src[l];
        const float* nw = tab->nw[l];
        float* nd = L.nn_dist + (size_t)c * n * 3;
        int* ni = L.nn_idx + (size_t)c * n * 3;
        const int first = ((gw - base) % GW + GW) % GW;  // this wave's first group of the level
        // (m <= kCgMaxNnM: the 16-chunk form of the routine keeps 1024 candidates in 186 VGPRs -- one wave per SIMD)
        if (m <= 64) three_nn_wave<1>(n, m, src, nw, nd, ni, first, GW, S);
        else three_nn_wave<4>(n, m, src, nw, nd, ni, first, GW, S);
        base += (n + kNnQ - 1) / kNnQ;
    }
    // ball query: one wave per query, the source cloud in LDS
    const float* staged = nullptr;
    int staged_n = 0;
#pragma unroll 1
    for (int l = 0; l < p.nlev; ++l) {
        const CgLevel L = p.lv[l];
        const int n = tab->n[l], m = L.m, ns = L.nsample;
        const float* src = tab->src[l];
        const float* nw = tab->nw[l];
        if (src != staged || n > staged_n) {  // (with the shortcut every source cloud is a prefix of the first one)
            __syncthreads();  // the previous copy / the 3-NN scratch it overlaps are no longer read
            for (int k = tid; k < n; k += kCgThreads) sxyz[k] = make_float4(src[k * 3 + 0], src[k * 3 + 1], src[k * 3 + 2], 0.f);
            __syncthreads();
            staged = src;
            staged_n = n;
        }
        const float thr = L.thr;
        for (int q = ((gw - base) % GW + GW) % GW; q < m; q += GW) {
            const float4 qp = nw == staged ? sxyz[q] : make_float4(nw[q * 3 + 0], nw[q * 3 + 1], nw[q * 3 + 2], 0.f);
            const float qx = qp.x, qy = qp.y, qz = qp.z;
            int* __restrict__ row = L.bq_idx + ((size_t)c * m + q) * ns;
            int cnt = 0, first = 0;
            for (int b0 = 0; b0 < n && cnt < ns; b0 += 64) {  // full queries stop collecting (tf_grouping.cu:20-21)
                const int k = b0 + lane;
                const float4 pt = sxyz[k < n ? k : n - 1];
                const float s = pn2_sqdist<BMODE>(qx - pt.x, qy - pt.y, qz - pt.z);
                const unsigned long long mask = __ballot(k < n && s <= thr);
                if (mask != 0ull) {
                    const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (((mask >> lane) & 1ull) && pos < ns) row[pos] = k;
                    if (cnt == 0) first = b0 + __ffsll((long long)mask) - 1;
                    cnt += __popcll(mask);
                }
            }
            cnt = cnt < ns ? cnt : ns;
            // short rows repeat the first hit (tf_grouping.cu:32-36); empty rows are zero (as pn2_query_ball_point)
            for (int e = cnt + lane; e < ns; e += 64) row[e] = cnt > 0 ? first : 0;
            if (L.bq_cnt && lane == 0) L.bq_cnt[(size_t)c * m + q] = cnt;
        }
        base += m;
    }
}

PN2_TUNABLE(int, g_cg_units, 6)  // tuning hook (pn2_debug_set(15, v)): cost units per wave (a 3-NN query group = 5, a ball query = 1);
                                 // configs[1], graph-timed: 24: 39.5 us, 12: 30.2, 8: 27.3, 6: 25.8, 3: 25.3 (nine separate launches: 52.1)

template <int FMODE, int BMODE>
int launch_coarse_geometry(int b, int R, const CgParams& p, hipStream_t st) {
    coarse_geometry_kernel<FMODE, BMODE><<<dim3(R, b), kCgThreads, kCgLdsBytes, st>>>(p);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

template <int FMODE>
int dispatch_bmode(int b, int R, const CgParams& p, int bq_mode, hipStream_t st) {
    switch (bq_mode) {
        case PN2_ARITH_STRICT: return launch_coarse_geometry<FMODE, PN2_ARITH_STRICT>(b, R, p, st);
        case PN2_ARITH_FMA: return launch_coarse_geometry<FMODE, PN2_ARITH_FMA>(b, R, p, st);
        case PN2_ARITH_FMA_ALT: return launch_coarse_geometry<FMODE, PN2_ARITH_FMA_ALT>(b, R, p, st);
    }
    return PN2_EINVAL;
}

}  // namespace

#ifdef PN2_TUNING_HOOKS
extern "C" int pn2_debug_set_coarse(int what, int value) { if (what == 15 && value > 0) { g_cg_units = value; return 0; } return -1; }
#endif  // PN2_TUNING_HOOKS

extern "C" int pn2_coarse_geometry(int b, int n0, int nlev, const int* npoint, const float* radius, const int* nsample,
                                   const float* xyz0, const int* tie_in, int* const* fps_idx, float* const* new_xyz,
                                   int* const* bq_idx, int* const* bq_cnt, float* const* nn_dist, int* const* nn_idx,
                                   int* tie_out, int fps_arith_mode, int bq_arith_mode, void* stream) {
    if (!npoint || !radius || !nsample || !xyz0 || !fps_idx || !new_xyz || !bq_idx) return PN2_ENULL;
    if (b <= 0 || n0 <= 0 || nlev <= 0) return PN2_EINVAL;
    if (nlev > kCgMaxLevels || n0 > kCgMaxN || b > 65535) return PN2_EUNSUP;
    CgParams p;
    p.nlev = nlev; p.n0 = n0; p.xyz0 = xyz0; p.tie_in = tie_in; p.tie_out = tie_out;
    int n = n0;
    for (int l = 0; l < nlev; ++l) {
        CgLevel& L = p.lv[l];
        if (npoint[l] <= 0 || nsample[l] <= 0) return PN2_EINVAL;
        if (npoint[l] > n) return PN2_EUNSUP;  // the stand-alone sampler repeats points then; not needed below level 1
        if (!fps_idx[l] || !new_xyz[l] || !bq_idx[l]) return PN2_ENULL;
        const bool nn = nn_dist && nn_dist[l];
        if (nn && (!nn_idx || !nn_idx[l])) return PN2_ENULL;
        if (nn && npoint[l] < 3) return PN2_EINVAL;  // as pn2_three_nn
        if (nn && npoint[l] > kCgMaxNnM) return PN2_EUNSUP;
        L.m = npoint[l]; L.nsample = nsample[l]; L.thr = pn2_ball_threshold(radius[l]);
        L.fps_idx = fps_idx[l]; L.new_xyz = new_xyz[l]; L.bq_idx = bq_idx[l]; L.bq_cnt = bq_cnt ? bq_cnt[l] : nullptr;
        L.nn_dist = nn ? nn_dist[l] : nullptr; L.nn_idx = nn ? nn_idx[l] : nullptr;
        n = npoint[l];
    }
    // workgroups per cloud: g_cg_units cost units per wave (a 3-NN query group = 5, a ball query = 1)
    long long cost = 0;
    n = n0;
    for (int l = 0; l < nlev; ++l) {
        if (p.lv[l].nn_dist) cost += 5LL * ((n + kNnQ - 1) / kNnQ);
        cost += npoint[l];
        n = npoint[l];
    }
    long long R = (cost + kCgWaves * g_cg_units - 1) / (kCgWaves * g_cg_units);
    R = R < 1 ? 1 : (R > kCgMaxR ? kCgMaxR : R);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (fps_arith_mode) {
        case PN2_ARITH_STRICT: return dispatch_bmode<PN2_ARITH_STRICT>(b, (int)R, p, bq_arith_mode, st);
        case PN2_ARITH_FMA: return dispatch_bmode<PN2_ARITH_FMA>(b, (int)R, p, bq_arith_mode, st);
        case PN2_ARITH_FMA_ALT: return dispatch_bmode<PN2_ARITH_FMA_ALT>(b, (int)R, p, bq_arith_mode, st);
    }
    return PN2_EINVAL;
}
