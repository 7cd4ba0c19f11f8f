// pn2_grouping.hip -- ball query, group_point and its gradient for gfx950.
// MI355X-native replacements for tf_ops/tf_grouping.cu:3-90 (reference).
//
// Ball query: the reference runs ONE 256-thread block per batch element with a
// serial per-thread scan (tf_grouping.cu:138-144).  Here one wave64 owns QPW
// queries: it streams the dataset 64 candidates at a time (one per lane, reused
// for all QPW queries), tests them with a wave ballot, and appends hits in lane
// order == index order with an mbcnt prefix count, so "the FIRST nsample points
// in the ball" (tf_grouping.cu:20-21) is preserved exactly.  The per-candidate
// sqrtf of the reference is replaced by an exactly equivalent threshold on the
// squared distance, computed on the host (see ball_threshold()).
//
// group_point: a row copy out[b,j,k,:] = points[b,idx[b,j,k],:]; HBM-bound, the
// (b,m,nsample,c) write dominates.  16-byte lanes along c, rows contiguous per
// wave, non-temporal stores so the gathered `points` stay L2-resident.
#include <math.h>

#include "pn2_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ball_threshold(radius): pn2_ball_threshold, pn2_common.h
inline float ball_threshold(float radius) { return pn2_ball_threshold(radius); }

constexpr int kBqThreads = 256;
constexpr int kBqWaves = kBqThreads / 64;

struct BqChunk { float x, y, z; };

// Scan one 64-candidate chunk for the wave's QPW queries.  TAIL: the chunk is the partial last
// one (lanes past n are masked).  CHECK: some query may already be full, so each query is tested
// against nsample first (tf_grouping.cu:20-21); while no query is full the test is skipped.
// Returns true when every query of the wave is full.
template <int MODE, int QPW, bool TAIL, bool CHECK>
__device__ __forceinline__ bool bq_scan(const BqChunk& cur, int c0, int lane, int n, float thr, int nsample,
                                        const float (&qx)[QPW], const float (&qy)[QPW], const float (&qz)[QPW],
                                        int (&cnt)[QPW], int (&first)[QPW], int* __restrict__ srow,
                                        bool& any_full) {
    // All QPW hit masks are produced first (7 VALU each, the compare writes an SGPR pair directly),
    // OR-ed, and ONE branch per chunk decides whether anything has to be appended: scalar
    // instructions cost ~4 cycles each and serialise with VALU issue (PMC: profiles/), so the
    // common no-hit chunk must not pay per-query branches.
    const int k = c0 + lane;
    unsigned long long mk[QPW];
    unsigned long long any = 0ull;
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
        const float s = pn2_sqdist<MODE>(qx[q] - cur.x, qy[q] - cur.y, qz[q] - cur.z);
        bool hit = s <= thr;
        if constexpr (TAIL) hit = hit && (k < n);
        mk[q] = __ballot(hit);
        if constexpr (CHECK) mk[q] = cnt[q] < nsample ? mk[q] : 0ull;  // full queries stop collecting (:20-21)
        any |= mk[q];
    }
    if (any != 0ull) {
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            const unsigned long long mask = mk[q];
            if (mask != 0ull) {
                const int pos = cnt[q] + (int)__builtin_amdgcn_mbcnt_hi(
                                             (unsigned)(mask >> 32),
                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (((mask >> lane) & 1ull) && pos < nsample) srow[q * nsample + pos] = k;
                if (cnt[q] == 0) first[q] = c0 + __ffsll((long long)mask) - 1;
                cnt[q] += __popcll(mask);
                any_full = any_full || (cnt[q] >= nsample);
            }
        }
    }
    if constexpr (CHECK) {
        bool all_full = true;
#pragma unroll
        for (int q = 0; q < QPW; ++q) all_full = all_full && (cnt[q] >= nsample);
        return all_full;
    }
    return false;
}

// One wave64 owns QPW queries.  The dataset streams through registers 64 candidates at a time
// (one global_load_dwordx3 per lane per chunk, shared by the QPW queries, prefetched chunks
// ahead with unconditional clamped loads so the compiler can keep counted vmcnt waits); hits are
// appended in lane order == index order into an LDS row buffer and the finished rows are written
// to HBM once, coalesced.  No VMEM store inside the scan loop.
template <int MODE, int QPW>
__global__ void __launch_bounds__(kBqThreads)
ball_query_kernel(int n, int m, float thr, int nsample, const float* __restrict__ xyz1_all,
                  const float* __restrict__ xyz2_all, int* __restrict__ idx_all,
                  int* __restrict__ cnt_all) {
    extern __shared__ int srows[];  // [kBqWaves][QPW][nsample]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    const int q0 = (blockIdx.x * kBqWaves + wave) * QPW;
    if (q0 >= m) return;  // wave-uniform; the kernel has no barriers
    const BqChunk* __restrict__ xyz1 = reinterpret_cast<const BqChunk*>(xyz1_all + (size_t)bi * n * 3);
    const float* __restrict__ xyz2 = xyz2_all + (size_t)bi * m * 3;
    int* __restrict__ idx = idx_all + ((size_t)bi * m + q0) * nsample;
    int* __restrict__ cnt_out = cnt_all + (size_t)bi * m + q0;
    int* srow = srows + wave * QPW * nsample;

    float qx[QPW], qy[QPW], qz[QPW];
    int cnt[QPW], first[QPW];
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
        const int jq = q0 + q < m ? q0 + q : m - 1;
        qx[q] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(xyz2[jq * 3 + 0])));
        qy[q] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(xyz2[jq * 3 + 1])));
        qz[q] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(xyz2[jq * 3 + 2])));
        cnt[q] = q0 + q < m ? 0 : nsample;  // out-of-range queries are "already full"
        first[q] = 0;
    }

    const int last = n - 1;
    auto load = [&](int c0) {  // unconditional: out-of-range lanes re-read the last point (masked by `valid`)
        const int kk = c0 + lane;
        return xyz1[kk < last ? kk : last];
    };
    // out-of-range queries start "full", which forces the checked path for this wave
    bool any_full = q0 + QPW > m;
    auto scan = [&](const BqChunk& cur, int c0) -> bool {
        if (c0 + 64 > n) return bq_scan<MODE, QPW, true, true>(cur, c0, lane, n, thr, nsample, qx, qy, qz, cnt, first, srow, any_full);
        if (any_full) return bq_scan<MODE, QPW, false, true>(cur, c0, lane, n, thr, nsample, qx, qy, qz, cnt, first, srow, any_full);
        return bq_scan<MODE, QPW, false, false>(cur, c0, lane, n, thr, nsample, qx, qy, qz, cnt, first, srow, any_full);
    };
    // 4 chunk registers rotate by name (loop unrolled x4): three loads stay in flight per wave
    BqChunk r0 = load(0), r1 = load(64), r2 = load(128), r3 = load(192);
    for (int c0 = 0; c0 < n; c0 += 256) {
        if (scan(r0, c0)) break;
        r0 = load(c0 + 256);
        if (c0 + 64 >= n || scan(r1, c0 + 64)) break;
        r1 = load(c0 + 320);
        if (c0 + 128 >= n || scan(r2, c0 + 128)) break;
        r2 = load(c0 + 384);
        if (c0 + 192 >= n || scan(r3, c0 + 192)) break;
        r3 = load(c0 + 448);
    }
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
        if (q0 + q < m) {
            const int c = cnt[q] < nsample ? cnt[q] : nsample;
            // short rows: remaining slots repeat the first hit (tf_grouping.cu:32-36);
            // empty rows are zero-filled (documented divergence: reference leaves them uninitialised)
            const int fill = c > 0 ? first[q] : 0;
            for (int l = lane; l < nsample; l += 64)
                idx[q * nsample + l] = l < c ? srow[q * nsample + l] : fill;
            if (lane == 0) cnt_out[q] = c;  // tf_grouping.cu:41
        }
    }
}

// ---- ball query, lane = query ---------------------------------------------------------------
// 512 threads = 8 waves serve 64 consecutive queries of one batch element: lane l of EVERY wave is
// query q0 + l, and wave s scans candidate segment s (n/8 consecutive points).  Candidates are
// wave-uniform, so their coordinates arrive through scalar loads (SGPR broadcast): the inner loop
// is 3 sub + mul + 2 fma + compare per lane with no cross-lane traffic, no vector loads to wait
// for and one branch per candidate (taken only when some lane hits).  Each wave appends its lane's
// hits (ascending index inside the segment) to a private LDS list; after one barrier the segment
// counts are prefix-summed per query, every wave drops its entries at their final positions of an
// LDS row image, and the 64 finished rows (contiguous in HBM) are written coalesced with the
// first-hit padding of tf_grouping.cu:32-36 applied on the fly.  Exactly the reference result:
// segments are visited in index order and only the first nsample positions are kept.
constexpr int kBq2Waves = 16;  // 16 segments: 1024 threads; 16-bit list entries keep the lists at 64 KB for nsample = 32
constexpr int kBq2Threads = 64 * kBq2Waves;

template <int MODE>
__global__ void __launch_bounds__(kBq2Threads)
ball_query_lane_kernel(int n, int m, float thr, int nsample, int seg,
                       const float* __restrict__ xyz1_all, const float* __restrict__ xyz2_all,
                       int* __restrict__ idx_all, int* __restrict__ cnt_all) {
    extern __shared__ int smem_i[];
    // layout: rows[64][nsample] | segcnt[W][64] | total[64] | first[64] | list[W][nsample][64] (16-bit, segment-relative)
    int* rows = smem_i;
    int* segcnt = rows + 64 * nsample;
    int* total = segcnt + kBq2Waves * 64;
    int* firsth = total + 64;
    unsigned short* list = reinterpret_cast<unsigned short*>(firsth + 64);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bi = blockIdx.y;
    const int q0 = blockIdx.x * 64;
    const float* __restrict__ xyz1 = xyz1_all + (size_t)bi * n * 3;
    const float* __restrict__ xyz2 = xyz2_all + (size_t)bi * m * 3;
    const int q = q0 + lane < m ? q0 + lane : m - 1;  // surplus lanes shadow the last query, never stored
    const float qx = xyz2[q * 3 + 0], qy = xyz2[q * 3 + 1], qz = xyz2[q * 3 + 2];

    int lo = wave * seg;
    int hi = lo + seg < n ? lo + seg : n;
    if (lo > n) lo = n;
    unsigned short* mylist = list + wave * nsample * 64 + lane;
    int cnt = 0;
    // Candidates are processed 8 at a time: their 24 coordinates are wave-uniform and arrive as wide
    // scalar loads (two groups in flight: A/B name rotation, no register copies); the 8 compares write
    // SGPR masks that are OR-ed so the common "nobody hit" case costs ONE branch per 8 candidates.
    struct Grp { float c[24]; };
    auto gload = [&](int k0) {
        Grp g;
        const float* __restrict__ p = xyz1 + (size_t)(k0 < n - 8 ? k0 : (n >= 8 ? n - 8 : 0)) * 3;  // clamped, never out of bounds
#pragma unroll
        for (int u = 0; u < 24; ++u) g.c[u] = p[u];
        return g;
    };
    auto append = [&](int k) {
        if (cnt < nsample) mylist[cnt * 64] = (unsigned short)(k - lo);
        ++cnt;
    };
    auto gtest = [&](const Grp& g, int k0) {
        unsigned long long mk[8];
        unsigned long long any = 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float sd = pn2_sqdist<MODE>(qx - g.c[3 * u], qy - g.c[3 * u + 1], qz - g.c[3 * u + 2]);
            mk[u] = __ballot(sd <= thr);
            any |= mk[u];
        }
        if (any != 0ull) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (mk[u] != 0ull && ((mk[u] >> lane) & 1ull)) append(k0 + u);
        }
    };
    int k = lo;
    if (n >= 8) {
        Grp ga = gload(k), gb = gload(k + 8);
        for (; k + 16 <= hi; k += 16) {
            gtest(ga, k);
            ga = gload(k + 16);
            gtest(gb, k + 8);
            gb = gload(k + 24);
        }
        if (k + 8 <= hi) { gtest(ga, k); k += 8; }
    }
    for (; k < hi; ++k) {  // tail (< 8 candidates)
        const float sd = pn2_sqdist<MODE>(qx - xyz1[k * 3 + 0], qy - xyz1[k * 3 + 1], qz - xyz1[k * 3 + 2]);
        if (sd <= thr) append(k);
    }
    segcnt[wave * 64 + lane] = cnt < nsample ? cnt : nsample;
    __syncthreads();

    // prefix over the segments (8 LDS reads per lane)
    int before = 0, tot = 0, fseg = -1;
#pragma unroll
    for (int sgi = 0; sgi < kBq2Waves; ++sgi) {
        const int c = segcnt[sgi * 64 + lane];
        if (sgi < wave) before += c;
        if (fseg < 0 && c > 0) fseg = sgi;
        tot += c;
    }
    const int mine = cnt < nsample ? cnt : nsample;
    for (int e = 0; e < mine; ++e) {
        const int pos = before + e;
        if (pos < nsample) rows[lane * nsample + pos] = lo + (int)mylist[e * 64];
    }
    if (wave == 0) {
        total[lane] = tot < nsample ? tot : nsample;
        firsth[lane] = fseg >= 0 ? fseg * seg + (int)list[fseg * nsample * 64 + lane] : 0;  // entry 0 of the first non-empty segment
    }
    __syncthreads();

    // coalesced write-out of the 64 rows (contiguous in HBM)
    const int nq = m - q0 < 64 ? m - q0 : 64;
    int* __restrict__ out = idx_all + ((size_t)bi * m + q0) * nsample;
    for (int e = tid; e < nq * nsample; e += kBq2Threads) {
        const int r = e / nsample, pos = e - r * nsample;
        out[e] = pos < total[r] ? rows[e] : firsth[r];  // padding with the first hit; empty rows -> 0
    }
    if (tid < nq) cnt_all[(size_t)bi * m + q0 + tid] = total[tid];
}

template <int MODE>
int launch_ball_query_lane(int b, int n, int m, float thr, int nsample, const float* xyz1,
                           const float* xyz2, int* idx, int* cnt, hipStream_t st) {
    int seg = (n + kBq2Waves - 1) / kBq2Waves;
    seg = (seg + 3) & ~3;
    const size_t lds = ((size_t)64 * nsample + kBq2Waves * 64 + 128) * sizeof(int) + (size_t)kBq2Waves * nsample * 64 * sizeof(unsigned short);
    if (seg > 65535) return PN2_ERANGE;
    auto kern = ball_query_lane_kernel<MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((m + 63) / 64, b);
    kern<<<grid, kBq2Threads, lds, st>>>(n, m, thr, nsample, seg, xyz1, xyz2, idx, cnt);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// ---- multi-radius ball query (MSG set abstraction, util/pointnet_util.py:219-282: one query set, several
// (radius, nsample) pairs; the reference re-scans xyz once per radius, :245-250) ----------------------------
// Same lane = query organisation with kBqmWaves candidate segments; every squared distance is computed ONCE and
// compared with the R thresholds, each radius keeps its own LDS hit lists / row image.  Bit-identical to R
// separate pn2_query_ball_point calls.
constexpr int kBqmWaves = 8;  // 8 segments: the R lists of nsample_r 16-bit entries stay within LDS
constexpr int kBqmThreads = 64 * kBqmWaves;
constexpr int kBqmMaxR = 3;

struct BqmParams {
    int n, m, seg, nr;
    float thr[kBqmMaxR];
    int ns[kBqmMaxR];
    const float* xyz1;
    const float* xyz2;
    int* idx[kBqmMaxR];
    int* cnt[kBqmMaxR];
};

template <int MODE, int R>
__global__ void __launch_bounds__(kBqmThreads)
ball_query_multi_kernel(BqmParams p) {
    extern __shared__ int smem_i[];
    // per radius r: rows[64][ns_r] | segcnt[W][64] | total[64] | first[64];  then the 16-bit lists [W][ns_r][64]
    int* rows[R]; int* segcnt[R]; int* total[R]; int* firsth[R];
    unsigned short* list[R];
    {
        int* q = smem_i;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            rows[r] = q; q += 64 * p.ns[r];
            segcnt[r] = q; q += kBqmWaves * 64;
            total[r] = q; q += 64;
            firsth[r] = q; q += 64;
        }
        unsigned short* l = reinterpret_cast<unsigned short*>(q);
#pragma unroll
        for (int r = 0; r < R; ++r) { list[r] = l; l += kBqmWaves * p.ns[r] * 64; }
    }
    const int n = p.n, m = p.m, seg = p.seg;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bi = blockIdx.y;
    const int q0 = blockIdx.x * 64;
    const float* __restrict__ xyz1 = p.xyz1 + (size_t)bi * n * 3;
    const float* __restrict__ xyz2 = p.xyz2 + (size_t)bi * m * 3;
    const int q = q0 + lane < m ? q0 + lane : m - 1;  // surplus lanes shadow the last query, never stored
    const float qx = xyz2[q * 3 + 0], qy = xyz2[q * 3 + 1], qz = xyz2[q * 3 + 2];
    int lo = wave * seg;
    int hi = lo + seg < n ? lo + seg : n;
    if (lo > n) lo = n;
    unsigned short* mylist[R];
    int cnt[R];
    float thr[R];
    int ns[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mylist[r] = list[r] + wave * p.ns[r] * 64 + lane;
        cnt[r] = 0; thr[r] = p.thr[r]; ns[r] = p.ns[r];
    }
    struct Grp { float c[24]; };
    auto gload = [&](int k0) {
        Grp g;
        const float* __restrict__ pp = xyz1 + (size_t)(k0 < n - 8 ? k0 : (n >= 8 ? n - 8 : 0)) * 3;  // clamped
#pragma unroll
        for (int u = 0; u < 24; ++u) g.c[u] = pp[u];
        return g;
    };
    auto append = [&](int k, float sd) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (sd <= thr[r]) {
                if (cnt[r] < ns[r]) mylist[r][cnt[r] * 64] = (unsigned short)(k - lo);
                ++cnt[r];
            }
        }
    };
    auto gtest = [&](const Grp& g, int k0) {
        float sd[8];
        unsigned long long any = 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            sd[u] = pn2_sqdist<MODE>(qx - g.c[3 * u], qy - g.c[3 * u + 1], qz - g.c[3 * u + 2]);
            bool h = false;
#pragma unroll
            for (int r = 0; r < R; ++r) h = h || (sd[u] <= thr[r]);
            any |= __ballot(h);
        }
        if (any != 0ull) {  // one branch per 8 candidates in the common "nobody hit" case
#pragma unroll
            for (int u = 0; u < 8; ++u) append(k0 + u, sd[u]);
        }
    };
    int k = lo;
    if (n >= 8) {
        Grp ga = gload(k), gb = gload(k + 8);
        for (; k + 16 <= hi; k += 16) {
            gtest(ga, k);
            ga = gload(k + 16);
            gtest(gb, k + 8);
            gb = gload(k + 24);
        }
        if (k + 8 <= hi) { gtest(ga, k); k += 8; }
    }
    for (; k < hi; ++k)  // tail (< 8 candidates)
        append(k, pn2_sqdist<MODE>(qx - xyz1[k * 3 + 0], qy - xyz1[k * 3 + 1], qz - xyz1[k * 3 + 2]));
#pragma unroll
    for (int r = 0; r < R; ++r) segcnt[r][wave * 64 + lane] = cnt[r] < ns[r] ? cnt[r] : ns[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int before = 0, tot = 0, fseg = -1;
#pragma unroll
        for (int sgi = 0; sgi < kBqmWaves; ++sgi) {
            const int c = segcnt[r][sgi * 64 + lane];
            if (sgi < wave) before += c;
            if (fseg < 0 && c > 0) fseg = sgi;
            tot += c;
        }
        const int mine = cnt[r] < ns[r] ? cnt[r] : ns[r];
        for (int e = 0; e < mine; ++e) {
            const int pos = before + e;
            if (pos < ns[r]) rows[r][lane * ns[r] + pos] = lo + (int)mylist[r][e * 64];
        }
        if (wave == 0) {
            total[r][lane] = tot < ns[r] ? tot : ns[r];
            firsth[r][lane] = fseg >= 0 ? fseg * seg + (int)list[r][fseg * ns[r] * 64 + lane] : 0;
        }
    }
    __syncthreads();
    const int nq = m - q0 < 64 ? m - q0 : 64;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int* __restrict__ out = p.idx[r] + ((size_t)bi * m + q0) * ns[r];
        for (int e = tid; e < nq * ns[r]; e += kBqmThreads) {
            const int row = e / ns[r], pos = e - row * ns[r];
            out[e] = pos < total[r][row] ? rows[r][e] : firsth[r][row];  // first-hit padding; empty rows -> 0
        }
        if (tid < nq) p.cnt[r][(size_t)bi * m + q0 + tid] = total[r][tid];
    }
}

template <int MODE, int R>
int launch_ball_query_multi(int b, const BqmParams& p, hipStream_t st) {
    size_t lds = 0;
    for (int r = 0; r < R; ++r)
        lds += ((size_t)64 * p.ns[r] + kBqmWaves * 64 + 128) * sizeof(int) + (size_t)kBqmWaves * p.ns[r] * 64 * sizeof(unsigned short);
    if (lds > 150 * 1024) return PN2_EUNSUP;
    auto kern = ball_query_multi_kernel<MODE, R>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((p.m + 63) / 64, b);
    kern<<<grid, kBqmThreads, lds, st>>>(p);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// ---- ball query on a per-block uniform grid -------------------------------------------------------------
// For clouds that fit LDS (n <= kBqgMaxN) every workgroup (64 consecutive queries of one batch element, 16
// waves) first bins the whole cloud into a uniform grid IN LDS -- bounding box, cell ids, histogram, prefix
// sum, scatter of (x, y, z, index) in cell order -- and then each wave answers 4 queries by testing only the
// 27 cells around the query with the reference's own fp32 expression.  Cells are at least radius*(1+1e-4)
// wide, so every point with sqrtf(d2) < radius lies in those cells; the hits (typically < 32 of ~100
// candidates instead of a scan over all n points) are ordered by index with a rank pass, which reproduces
// "first nsample in index order" (tf_grouping.cu:18-37).  A query with more than kBqgCap hits (dense clouds)
// is re-done over the same 27 cells: each lane marks its hits in a per-query LDS bitmap indexed by point number, and a
// prefix count over the bitmap hands out the first nsample set bits in index order (no scan of the cloud).  Results are
// bit-identical to the scan kernels.
constexpr int kBqgThreads = 1024;
constexpr int kBqgWaves = kBqgThreads / 64;
constexpr int kBqgMaxN = 8192;
constexpr int kBqgPPT = kBqgMaxN / kBqgThreads;  // points per thread while building
constexpr int kBqgDim = 16;                       // cells per axis (at most)
constexpr int kBqgCells = kBqgDim * kBqgDim * kBqgDim;
constexpr int kBqgCap = 64;                       // hits per query kept before falling back
constexpr int kBqgQPW = 4;                        // queries per wave (64 per workgroup)

struct BqgGrid {
    float lo[3], inv_h[3];
    int dim[3];
};

__device__ __forceinline__ int bqg_cell1(float x, float lo, float inv_h, int dim) {
    int c = (int)floorf((x - lo) * inv_h);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

// The cell-sorted form of one cloud for a given radius, as the query kernel wants it in LDS: x / y / z and original index
// of every point in cell order, the start of every cell, the grid.  Built in LDS by bqg_build (one workgroup);
// pn2_ball_query_bin stores it to a caller-provided workspace ONCE per cloud so that the query workgroups (16 per cloud at
// m = 1024) copy it instead of each re-binning the whole cloud (10 of their 18 us, 3.8x the compulsory HBM traffic).
struct BqgLds {
    float *sx, *sy, *sz, *red;
    int *ccount, *rowbuf;
    BqgGrid* grid;
    unsigned short *sidx, *cstart, *hits;
};
__device__ __forceinline__ BqgLds bqg_carve(int* smem_i, int np) {
    // layout (np = n rounded up to 64): sx[np] sy[np] sz[np] | ccount[kBqgCells+1] | rowbuf[W][64] | red[W][6] |
    //         grid | sidx[np] (u16) | cstart[kBqgCells+2] (u16) | hits[W][4][kBqgCap] (u16)
    BqgLds L;
    L.sx = reinterpret_cast<float*>(smem_i);
    L.sy = L.sx + np;
    L.sz = L.sy + np;
    L.ccount = reinterpret_cast<int*>(L.sz + np);
    L.rowbuf = L.ccount + kBqgCells + 1;
    L.red = reinterpret_cast<float*>(L.rowbuf + kBqgWaves * 64);
    L.grid = reinterpret_cast<BqgGrid*>(L.red + kBqgWaves * 6);
    L.sidx = reinterpret_cast<unsigned short*>(L.grid + 1);
    L.cstart = L.sidx + np;
    L.hits = L.cstart + kBqgCells + 2;
    return L;
}
inline size_t bqg_lds_bytes(int np) {
    return (size_t)np * 12 + (size_t)(kBqgCells + 1) * 4 + kBqgWaves * 64 * 4 + kBqgWaves * 6 * 4 + sizeof(BqgGrid) +
           (size_t)np * 2 + (size_t)(kBqgCells + 2) * 2 + (size_t)kBqgWaves * 4 * kBqgCap * 2 + 64;
}
// workspace of pn2_ball_query_bin, per cloud: float sx[np] sy[np] sz[np] | u16 sidx[np] | u16 cstart[kBqgCells+2] | BqgGrid
__host__ __device__ inline size_t bqg_ws_stride(int n) {
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    return (np * 14 + (size_t)(kBqgCells + 2) * 2 + sizeof(BqgGrid) + 255) & ~(size_t)255;
}

// bounding box -> grid -> LDS histogram (the atomic returns the rank inside the cell) -> prefix sum -> scatter.
// All kBqgThreads threads of the workgroup; ends with a barrier.
__device__ __forceinline__ void bqg_build(const BqgLds& L, int n, float radius, const float* __restrict__ xyz1, int ld = 3) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // thread t owns points t, t+1024, ...: a wave's load instruction covers 768 contiguous bytes.  (Tried: 8 consecutive
    // points per thread read as six 16-byte loads -- every instruction then touches 48 cache lines; +7.5 us.)
    float px[kBqgPPT], py[kBqgPPT], pz[kBqgPPT];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < kBqgPPT; ++j) {
        const int k = tid + kBqgThreads * j;
        const int kc = k < n ? k : n - 1;
        px[j] = xyz1[kc * ld + 0]; py[j] = xyz1[kc * ld + 1]; pz[j] = xyz1[kc * ld + 2];
        mn[0] = fminf(mn[0], px[j]); mx[0] = fmaxf(mx[0], px[j]);
        mn[1] = fminf(mn[1], py[j]); mx[1] = fmaxf(mx[1], py[j]);
        mn[2] = fminf(mn[2], pz[j]); mx[2] = fmaxf(mx[2], pz[j]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
        if (lane == 0) { L.red[wave * 6 + a] = mn[a]; L.red[wave * 6 + 3 + a] = mx[a]; }
    }
    for (int e = tid; e < kBqgCells + 1; e += kBqgThreads) L.ccount[e] = 0;
    __syncthreads();
    if (tid < 3) {  // one thread per axis
        const int a = tid;
        const float h0 = radius * 1.0001f + 1e-30f;  // every hit is within radius*(1+1e-6) per axis: 27 cells suffice
        float lo = L.red[a], hi = L.red[3 + a];
        for (int w = 1; w < kBqgWaves; ++w) { lo = fminf(lo, L.red[w * 6 + a]); hi = fmaxf(hi, L.red[w * 6 + 3 + a]); }
        const float ext = hi - lo;
        float h = fmaxf(h0, ext * (1.00001f / kBqgDim));
        if (!(h > 0.f) || !(h < 3.0e38f)) h = 1.0f;
        const float inv = 1.0f / h;
        int d = (int)floorf(ext * inv) + 1;
        d = d < 1 ? 1 : (d > kBqgDim ? kBqgDim : d);
        L.grid->lo[a] = lo; L.grid->inv_h[a] = inv; L.grid->dim[a] = d;
    }
    __syncthreads();
    const BqgGrid G = *L.grid;
    int cell[kBqgPPT], rnk[kBqgPPT];
#pragma unroll
    for (int j = 0; j < kBqgPPT; ++j) {
        const int k = tid + kBqgThreads * j;
        const int cx = bqg_cell1(px[j], G.lo[0], G.inv_h[0], G.dim[0]);
        const int cy = bqg_cell1(py[j], G.lo[1], G.inv_h[1], G.dim[1]);
        const int cz = bqg_cell1(pz[j], G.lo[2], G.inv_h[2], G.dim[2]);
        cell[j] = (cz * G.dim[1] + cy) * G.dim[0] + cx;
        rnk[j] = k < n ? atomicAdd(&L.ccount[cell[j]], 1) : 0;
    }
    __syncthreads();
    {   // exclusive scan of ccount[0..kBqgCells): 4 cells per thread, wave scan, wave totals through `red`
        int v[4], s = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = L.ccount[tid * 4 + i]; s += v[i]; }
        int inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        int* wtot = reinterpret_cast<int*>(L.red);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wtot[w];
        int run = base + inc - s;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            L.cstart[tid * 4 + i] = (unsigned short)run;
            run += v[i];
        }
        if (tid == kBqgThreads - 1) L.cstart[kBqgCells] = (unsigned short)run;  // == n
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kBqgPPT; ++j) {
        const int k = tid + kBqgThreads * j;
        if (k < n) {
            const int pos = (int)L.cstart[cell[j]] + rnk[j];
            L.sx[pos] = px[j]; L.sy[pos] = py[j]; L.sz[pos] = pz[j];
            L.sidx[pos] = (unsigned short)k;
        }
    }
    __syncthreads();
}

// LDS <-> workspace copies of the cell-sorted cloud (16-byte lanes; every array is 16-byte aligned in both places except
// cstart, whose byte length is a multiple of 4)
template <bool TO_LDS>
__device__ __forceinline__ void bqg_copy(const BqgLds& L, int np, unsigned char* __restrict__ ws) {
    const int tid = threadIdx.x;
    float* gx = reinterpret_cast<float*>(ws);
    unsigned short* gi = reinterpret_cast<unsigned short*>(gx + 3 * (size_t)np);
    unsigned short* gc = gi + np;
    BqgGrid* gg = reinterpret_cast<BqgGrid*>(gc + kBqgCells + 2);
    auto mv = [&](void* lds, void* glb, int bytes) {  // bytes % 4 == 0; widest lanes both addresses and the length allow
        const unsigned al = ((unsigned)(size_t)lds | (unsigned)(size_t)glb | (unsigned)bytes);
        if ((al & 15u) == 0u) {
            uint4* a4 = reinterpret_cast<uint4*>(lds);
            uint4* g4 = reinterpret_cast<uint4*>(glb);
            for (int i = tid; i < bytes / 16; i += kBqgThreads) { if (TO_LDS) a4[i] = g4[i]; else g4[i] = a4[i]; }
        } else if ((al & 7u) == 0u) {
            uint2* a2 = reinterpret_cast<uint2*>(lds);
            uint2* g2 = reinterpret_cast<uint2*>(glb);
            for (int i = tid; i < bytes / 8; i += kBqgThreads) { if (TO_LDS) a2[i] = g2[i]; else g2[i] = a2[i]; }
        } else {
            unsigned* a = reinterpret_cast<unsigned*>(lds);
            unsigned* g = reinterpret_cast<unsigned*>(glb);
            for (int i = tid; i < bytes / 4; i += kBqgThreads) { if (TO_LDS) a[i] = g[i]; else g[i] = a[i]; }
        }
    };
    mv(L.sx, gx, np * 12);  // sx | sy | sz are contiguous in both
    mv(L.sidx, gi, np * 2);
    mv(L.cstart, gc, (kBqgCells + 2) * 2);
    mv(L.grid, gg, (int)sizeof(BqgGrid));
}

__global__ void __launch_bounds__(kBqgThreads)
ball_query_bin_kernel(int n, float radius, const float* __restrict__ xyz1_all, int ld1, unsigned char* __restrict__ ws_all, size_t stride) {
    extern __shared__ int smem_i[];
    const int np = (n + 63) & ~63;
    const BqgLds L = bqg_carve(smem_i, np);
    bqg_build(L, n, radius, xyz1_all + (size_t)blockIdx.x * n * ld1, ld1);
    bqg_copy<false>(L, np, ws_all + (size_t)blockIdx.x * stride);
}

template <int MODE, bool PRE>
__global__ void __launch_bounds__(kBqgThreads)
ball_query_grid_kernel(int n, int m, float radius, float thr, int nsample, const float* __restrict__ xyz1_all,
                       const float* __restrict__ xyz2_all, int* __restrict__ idx_all, int* __restrict__ cnt_all,
                       unsigned char* __restrict__ ws_all, size_t ws_stride, int ld1) {
    extern __shared__ int smem_i[];
    const int np = (n + 63) & ~63;
    const BqgLds L = bqg_carve(smem_i, np);
    float* sx = L.sx; float* sy = L.sy; float* sz = L.sz;
    int* ccount = L.ccount; int* rowbuf = L.rowbuf;
    // rowbuf[0..63]: first hit of every query of the workgroup; [64..128]: local numbers of its over-full queries and, in the
    // last slot, their count (the launch already asks for all 160 KB: no static LDS beside it)
    int* s_dense = rowbuf + kBqgWaves * kBqgQPW;
    if (threadIdx.x == 0) s_dense[kBqgWaves * kBqgQPW] = 0;  // (bqg_build / the copy below end with a barrier)
    unsigned short* sidx = L.sidx; unsigned short* cstart = L.cstart; unsigned short* hits = L.hits;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware block remap (speed only; workgroup L runs on XCD L % 8): the gridDim.x workgroups of a cloud all read
    // that cloud, so they should share ONE L2 -- in launch order they land on 8 different XCDs and every L2 fetches every
    // cloud from HBM.
    int bxq = blockIdx.x, bi = blockIdx.y;
    {
        const unsigned nwg = gridDim.x * gridDim.y;
        if ((nwg & 7u) == 0u) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned swz = (lin & 7u) * (nwg >> 3) + (lin >> 3);
            bxq = (int)(swz % gridDim.x);
            bi = (int)(swz / gridDim.x);
        }
    }
    const float* __restrict__ xyz2 = xyz2_all + (size_t)bi * m * 3;
    if constexpr (PRE) {
        bqg_copy<true>(L, np, ws_all + (size_t)bi * ws_stride);
        __syncthreads();
    } else {
        bqg_build(L, n, radius, xyz1_all + (size_t)bi * n * ld1, ld1);  // (ld1: row stride of the cloud in floats, 3 = dense)
    }
    const BqgGrid G = *L.grid;
    // ---- queries: a wave answers 4 at a time, one per 16-lane group ------------------------------------------
    // (the 27 cells of a query hold ~100 candidates in 9 short runs: 64-lane steps would idle most lanes)
    const int grp = lane >> 4, l16 = lane & 15;
    unsigned short* myhits = hits + (wave * 4 + grp) * kBqgCap;
    int* myfirst = rowbuf + wave * 4 + grp;
    const int qbase = bxq * (kBqgWaves * kBqgQPW) + wave * kBqgQPW;
    {
        const int q = qbase + grp;
        const bool qv = q < m;
        const int qc = qv ? q : m - 1;
        const float qx = xyz2[qc * 3 + 0], qy = xyz2[qc * 3 + 1], qz = xyz2[qc * 3 + 2];
        const int cx = bqg_cell1(qx, G.lo[0], G.inv_h[0], G.dim[0]);
        const int cy = bqg_cell1(qy, G.lo[1], G.inv_h[1], G.dim[1]);
        const int cz = bqg_cell1(qz, G.lo[2], G.inv_h[2], G.dim[2]);
        const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < G.dim[0] ? cx + 1 : G.dim[0] - 1;
        int nh = 0;
        // a query whose hit list overflows (dense balls: unit-normal clouds at r = 0.5 have ~270 hits around their centre) is
        // redone by the whole wave below; its list pass stops at the overflow instead of scanning the remaining cells
#pragma unroll 1
        for (int r = 0; r < 9; ++r) {
            const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
            int s0 = 0, e0 = 0;
            if (qv && nh <= kBqgCap && z >= 0 && z < G.dim[2] && y >= 0 && y < G.dim[1]) {
                const int rowc = (z * G.dim[1] + y) * G.dim[0];
                s0 = cstart[rowc + x0];
                e0 = cstart[rowc + x1 + 1];  // the 3 x-cells are contiguous in the sorted arrays
            }
            for (int p0 = s0; __any(p0 < e0); p0 += 16) {
                const int p = p0 + l16;
                const int pc = p < e0 ? p : (e0 > 0 ? e0 - 1 : 0);
                const float sd = pn2_sqdist<MODE>(qx - sx[pc], qy - sy[pc], qz - sz[pc]);
                const bool hit = p < e0 && sd <= thr;
                const unsigned gm = (unsigned)(__ballot(hit) >> (16 * grp)) & 0xffffu;
                if (hit) {
                    const int pos = nh + __popc(gm & ((1u << l16) - 1u));
                    if (pos < kBqgCap) myhits[pos] = sidx[pc];
                }
                nh += __popc(gm);
                if (nh > kBqgCap) e0 = 0;  // overflow: this group is done with the list pass (nh is recounted from the bitmap)
            }
        }
        int* __restrict__ out = idx_all + ((size_t)bi * m + qc) * nsample;
        const bool sparse = nh <= kBqgCap;
        if (qv && sparse) {
            // order the hits by index: rank = number of hits with a smaller index (indices are distinct)
            for (int h = l16; h < nh; h += 16) {
                const int me = myhits[h];
                int rank = 0;
                for (int j = 0; j < nh; ++j) rank += (int)myhits[j] < me ? 1 : 0;
                if (rank < nsample) out[rank] = me;
                if (rank == 0) *myfirst = me;
            }
        }
        // queries with more than kBqgCap hits (dense balls), the whole wave per query: the hits of the 27 cells set
        // bits of an 8192-bit map (the histogram memory is free by now), which is then read out in index order.
        // They are SHARED OUT over the workgroup's waves through an LDS list (r04): a unit-normal cloud at r = 0.5 has ~6 of
        // them among a workgroup's 64 queries, unevenly spread, and a wave that owns three of them made the other 15 wait.
        // (Their rows are full -- more than kBqgCap >= nsample hits -- so the owning group needs nothing back.)
        if (qv && !sparse && l16 == 0) s_dense[atomicAdd(&s_dense[kBqgWaves * kBqgQPW], 1)] = wave * kBqgQPW + grp;
        __syncthreads();
        const int ndense = s_dense[kBqgWaves * kBqgQPW];
        for (int di = wave; di < ndense; di += kBqgWaves) {  // wave-uniform
            const int q2 = bxq * (kBqgWaves * kBqgQPW) + s_dense[di];
            const float ax = xyz2[q2 * 3 + 0], ay = xyz2[q2 * 3 + 1], az = xyz2[q2 * 3 + 2];
            const int dx = bqg_cell1(ax, G.lo[0], G.inv_h[0], G.dim[0]);
            const int dy = bqg_cell1(ay, G.lo[1], G.inv_h[1], G.dim[1]);
            const int dz = bqg_cell1(az, G.lo[2], G.inv_h[2], G.dim[2]);
            const int u0 = dx > 0 ? dx - 1 : 0, u1 = dx + 1 < G.dim[0] ? dx + 1 : G.dim[0] - 1;
            unsigned* bm = reinterpret_cast<unsigned*>(ccount) + wave * 256;
#pragma unroll
            for (int w = 0; w < 4; ++w) bm[lane * 4 + w] = 0u;
            for (int r = 0; r < 9; ++r) {
                const int z = dz + r / 3 - 1, y = dy + r % 3 - 1;
                if (z < 0 || z >= G.dim[2] || y < 0 || y >= G.dim[1]) continue;
                const int rowc = (z * G.dim[1] + y) * G.dim[0];
                const int s0 = cstart[rowc + u0], e0 = cstart[rowc + u1 + 1];
                for (int p = s0 + lane; p < e0; p += 64) {
                    const float sd = pn2_sqdist<MODE>(ax - sx[p], ay - sy[p], az - sz[p]);
                    if (sd <= thr) {
                        const unsigned id = sidx[p];
                        atomicOr(&bm[id >> 5], 1u << (id & 31u));
                    }
                }
            }
            // lane l owns words 4l..4l+3 (indices 128l .. 128l+127): exclusive prefix of the set-bit counts
            unsigned wv[4];
            int mine = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { wv[w] = bm[lane * 4 + w]; mine += __popc(wv[w]); }
            int inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            int pos = inc - mine;
            int* __restrict__ o2 = idx_all + ((size_t)bi * m + q2) * nsample;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned bits = wv[w];
                while (bits != 0u && pos < nsample) {
                    const int bit = __ffs(bits) - 1;
                    bits &= bits - 1u;
                    o2[pos] = (lane * 4 + w) * 32 + bit;
                    ++pos;
                }
            }
        }
        // short rows repeat the first hit (tf_grouping.cu:32-36); empty rows are zero (documented divergence)
        const int cnt = nh < nsample ? nh : nsample;
        if (qv) {
            const int first = cnt > 0 ? *myfirst : 0;
            for (int l = cnt + l16; l < nsample; l += 16) out[l] = first;
            if (l16 == 0) cnt_all[(size_t)bi * m + q] = cnt;
        }
    }
}

template <int MODE>
int launch_ball_query_grid(int b, int n, int m, float radius, float thr, int nsample, const float* xyz1,
                           const float* xyz2, int* idx, int* cnt, hipStream_t st, unsigned char* bins = nullptr, int ld1 = 3) {
    const int np = (n + 63) & ~63;
    const size_t lds = bqg_lds_bytes(np);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ball_query_grid_kernel<MODE, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ball_query_grid_kernel<MODE, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((m + kBqgWaves * kBqgQPW - 1) / (kBqgWaves * kBqgQPW), b);
    if (bins) ball_query_grid_kernel<MODE, true><<<grid, kBqgThreads, lds, st>>>(n, m, radius, thr, nsample, xyz1, xyz2, idx, cnt, bins, bqg_ws_stride(n), ld1);
    else ball_query_grid_kernel<MODE, false><<<grid, kBqgThreads, lds, st>>>(n, m, radius, thr, nsample, xyz1, xyz2, idx, cnt, nullptr, 0, ld1);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

PN2_TUNABLE(int, g_bq_variant, 0)  // tuning hook (pn2_debug_set(2, v)): 0 = auto, 1 = wave-per-queries kernel, 2 = lane kernel, 3 = LDS grid kernel

PN2_TUNABLE(int, g_bq_qpw, 8)  // tuning hook (pn2_debug_set(1, v))

template <int MODE, int QPW>
int launch_ball_query_q(int b, int n, int m, float thr, int nsample, const float* xyz1,
                        const float* xyz2, int* idx, int* cnt, hipStream_t st) {
    dim3 grid((m + kBqWaves * QPW - 1) / (kBqWaves * QPW), b);
    const size_t lds = (size_t)kBqWaves * QPW * nsample * sizeof(int);
    if (lds > 64 * 1024) return PN2_ERANGE;
    ball_query_kernel<MODE, QPW><<<grid, kBqThreads, lds, st>>>(n, m, thr, nsample, xyz1, xyz2, idx, cnt);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// kernel: 0 = choose by shape, 1 = wave-per-queries scan, 2 = lane-per-query scan, 3 = LDS grid (explicit values come from
// pn2_query_ball_point_kernel, the test / diagnostic door; every kernel returns the same bits)
template <int MODE>
int launch_ball_query(int b, int n, int m, float radius, float thr, int nsample, const float* xyz1,
                      const float* xyz2, int* idx, int* cnt, hipStream_t st, int kernel = 0) {
    if (kernel == 0) kernel = g_bq_variant;
    // per-block LDS grid: the whole cloud fits LDS, enough points / queries to amortise building it in every
    // workgroup, and a neighbourhood size (nsample <= 32 is the caller's own estimate of the hits per ball) for which
    // the hit lists stay short; dense balls are cheaper on the ordered scan kernels (profiles/r01_ball_query_grid.txt)
    const bool grid_ok = n <= kBqgMaxN && n >= 4096 && m >= 256 && nsample <= 64 && radius < 1e18f;
    if ((kernel == 3 && n <= kBqgMaxN && nsample <= 64 && radius < 1e18f) || (kernel == 0 && grid_ok))
        return launch_ball_query_grid<MODE>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, cnt, st);
    // lane = query kernel whenever its LDS lists fit and there are enough queries to fill 64 lanes
    const bool lane_ok = nsample <= 64 && m >= 32 && n <= 65535 * kBq2Waves;
    if ((kernel == 2 && nsample <= 64 && n <= 65535 * kBq2Waves) || (kernel == 0 && lane_ok))
        return launch_ball_query_lane<MODE>(b, n, m, thr, nsample, xyz1, xyz2, idx, cnt, st);
    // fewer queries per wave when there are too few queries to fill the chip (>= ~4 waves/SIMD wanted)
    int qpw = g_bq_qpw;
    while (qpw > 2 && (long long)b * ((m + qpw - 1) / qpw) < 4096) qpw >>= 1;
    if (qpw >= 8) return launch_ball_query_q<MODE, 8>(b, n, m, thr, nsample, xyz1, xyz2, idx, cnt, st);
    if (qpw >= 4) return launch_ball_query_q<MODE, 4>(b, n, m, thr, nsample, xyz1, xyz2, idx, cnt, st);
    return launch_ball_query_q<MODE, 2>(b, n, m, thr, nsample, xyz1, xyz2, idx, cnt, st);
}

// ---- group_point -----------------------------------------------------------
// grid.y = batch; e indexes the (m*nsample*c/VEC) vector elements of one batch.  UNR independent
// 16-byte gathers are in flight per thread before the first store (memory-level parallelism);
// NT selects non-temporal stores (the output is written once and never re-read by this kernel).
PN2_TUNABLE(int, g_gp_variant, 0)  // tuning hook (pn2_debug_set(3, v)): bit0 = plain stores, bits 4.. = blocks-per-CU override

template <typename VT, int VEC, int UNR, bool NT>
__global__ void __launch_bounds__(256)
group_point_kernel(int n, int c, unsigned per_batch_rows, const float* __restrict__ points_all,
                   const int* __restrict__ idx_all, float* __restrict__ out_all, int g_remap) {
    const unsigned cv = (unsigned)c / VEC;
    const unsigned total = per_batch_rows * cv;
    // XCD-aware block remap (speed only; workgroup L runs on XCD L % 8): every XCD gathers from whole batch
    // elements of its own instead of from all of `points`
    unsigned bx = blockIdx.x;
    int bi = blockIdx.y;
    if (g_remap) {
        const unsigned nwg = gridDim.x * gridDim.y;
        if ((nwg & 7u) == 0u) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned swz = (lin & 7u) * (nwg >> 3) + (lin >> 3);
            bx = swz % gridDim.x;
            bi = (int)(swz / gridDim.x);
        }
    }
    const VT* __restrict__ points = reinterpret_cast<const VT*>(points_all + (size_t)bi * n * c);
    const int* __restrict__ idx = idx_all + (size_t)bi * per_batch_rows;
    VT* __restrict__ out = reinterpret_cast<VT*>(out_all + (size_t)bi * per_batch_rows * c);
    const bool pow2 = (cv & (cv - 1)) == 0;
    const unsigned sh = 31 - __builtin_clz(cv | 1u);
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned e0 = bx * blockDim.x + threadIdx.x; e0 < total; e0 += stride * UNR) {
        VT v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned e = e0 + u * stride;
            if (e < total) {
                unsigned row, col;
                if (pow2) { row = e >> sh; col = e & (cv - 1); }
                else { row = e / cv; col = e - row * cv; }
                v[u] = points[(size_t)idx[row] * cv + col];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned e = e0 + u * stride;
            if (e < total) {
                if constexpr (NT) __builtin_nontemporal_store(v[u], &out[e]);
                else out[e] = v[u];
            }
        }
    }
}

template <typename VT, int VEC>
__global__ void __launch_bounds__(256)
group_point_grad_kernel(int n, int c, unsigned per_batch_rows, const float* __restrict__ grad_out_all,
                        const int* __restrict__ idx_all, float* __restrict__ grad_points_all) {
    const unsigned cv = (unsigned)c / VEC;
    const unsigned total = per_batch_rows * cv;
    const int bi = blockIdx.y;
    const VT* __restrict__ go = reinterpret_cast<const VT*>(grad_out_all + (size_t)bi * per_batch_rows * c);
    const int* __restrict__ idx = idx_all + (size_t)bi * per_batch_rows;
    float* __restrict__ gp = grad_points_all + (size_t)bi * n * c;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned row = e / cv, col = e - row * cv;
        const int ii = idx[row];
        const VT v = go[e];
        float* dst = gp + (size_t)ii * c + col * VEC;
        if constexpr (VEC == 4) {
            atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y);
            atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
        } else {
            atomicAdd(dst, v);  // tf_grouping.cu:85-86
        }
    }
}

// ---- selection sort / top-k (tf_grouping.cu:95-136) ------------------------------------------------
// One wave per (batch, query) row; the row (values + permutation) lives in LDS.  Round s finds the
// minimum of positions [s, n) under the order (value, position) -- exactly what the reference's
// ascending scan with a strict '<' selects -- with a strided lane scan + a 6-step wave reduction,
// then swaps it into position s like the reference does, so the WHOLE output row (sorted head and
// permuted tail) is bit-identical.
__global__ void __launch_bounds__(64)
selection_sort_kernel(int n, int m, int k, const float* __restrict__ dist_all,
                      int* __restrict__ outi_all, float* __restrict__ out_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_smem[];
    float* v = reinterpret_cast<float*>(ss_smem);
    int* id = reinterpret_cast<int*>(v + n);
    const int lane = threadIdx.x;
    const size_t rowoff = ((size_t)blockIdx.y * m + blockIdx.x) * n;
    const float* __restrict__ src = dist_all + rowoff;
    for (int t = lane; t < n; t += 64) { v[t] = src[t]; id[t] = t; }
    __syncthreads();
    const int kk = k < n ? k : n;
    for (int s = 0; s < kk; ++s) {
        float bv = v[s];
        int bt = s;
        for (int t = s + 1 + lane; t < n; t += 64) {
            const float x = v[t];
            if (x < bv) { bv = x; bt = t; }  // strict: a lane keeps its lowest position among equals
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float pv = __shfl_xor(bv, o);
            const int pt = __shfl_xor(bt, o);
            if (pv < bv || (pv == bv && pt < bt)) { bv = pv; bt = pt; }
        }
        if (lane == 0 && bt != s) {
            const float tv = v[bt]; v[bt] = v[s]; v[s] = tv;
            const int ti = id[bt]; id[bt] = id[s]; id[s] = ti;
        }
        __syncthreads();
    }
    for (int t = lane; t < n; t += 64) { out_all[rowoff + t] = v[t]; outi_all[rowoff + t] = id[t]; }
}

inline int grid_x_for(unsigned long long total, int block, int batches) {
    unsigned long long g = (total + block - 1) / block;
    unsigned long long cap = (256ull * 8 + batches - 1) / batches;  // ~8 blocks per CU overall
    if (cap < 1) cap = 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#ifdef PN2_TUNING_HOOKS
extern "C" int pn2_debug_set_grouping(int what, int value) {
    if (what == 1) { g_bq_qpw = value; return 0; }
    if (what == 2) { g_bq_variant = value; return 0; }
    if (what == 3) { g_gp_variant = value; return 0; }
    return PN2_EINVAL;
}
#endif  // PN2_TUNING_HOOKS

static int query_ball_point_impl(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2,
                                 int* idx, int* pts_cnt, int arith_mode, int kernel, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0) return PN2_EINVAL;
    if (!(radius > 0.0f)) return PN2_EINVAL;  // tf_grouping.cpp:80-83 "expects positive radius"
    if (!xyz1 || !xyz2 || !idx || !pts_cnt) return PN2_ENULL;
    if ((long long)n * 3 > 0x7fffffffLL || (long long)m * 3 > 0x7fffffffLL || b > 65535) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float thr = ball_threshold(radius);
    switch (arith_mode) {
        case PN2_ARITH_STRICT: return launch_ball_query<PN2_ARITH_STRICT>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, pts_cnt, st, kernel);
        case PN2_ARITH_FMA: return launch_ball_query<PN2_ARITH_FMA>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, pts_cnt, st, kernel);
        case PN2_ARITH_FMA_ALT: return launch_ball_query<PN2_ARITH_FMA_ALT>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, pts_cnt, st, kernel);
        default: return PN2_EINVAL;
    }
}

extern "C" int pn2_query_ball_point(int b, int n, int m, float radius, int nsample,
                                    const float* xyz1, const float* xyz2, int* idx, int* pts_cnt,
                                    int arith_mode, void* stream) {
    return query_ball_point_impl(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt, arith_mode, 0, stream);
}

// pn2_query_ball_point on a cloud whose rows are ld1 floats apart (the xyz columns of a (b,n,6) xyz+rgb batch read in place;
// model.py:26-29).  Only the shapes the LDS-grid kernel takes (it reads the cloud once, into LDS); anything else: PN2_EUNSUP
// and the caller queries a dense copy.  Bit-identical to pn2_query_ball_point on that copy.
extern "C" int pn2_query_ball_point_ld(int b, int n, int m, float radius, int nsample, const float* xyz1, int ld1,
                                       const float* xyz2, int* idx, int* pts_cnt, int arith_mode, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || ld1 < 3) return PN2_EINVAL;
    if (!(radius > 0.0f)) return PN2_EINVAL;
    if (!xyz1 || !xyz2 || !idx || !pts_cnt) return PN2_ENULL;
    if ((long long)n * ld1 > 0x7fffffffLL || (long long)m * 3 > 0x7fffffffLL || b > 65535) return PN2_ERANGE;
    if (!(g_bq_variant == 0 && n <= kBqgMaxN && n >= 4096 && m >= 256 && nsample <= 64 && radius < 1e18f)) return PN2_EUNSUP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float thr = ball_threshold(radius);
    switch (arith_mode) {
        case PN2_ARITH_STRICT: return launch_ball_query_grid<PN2_ARITH_STRICT>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, pts_cnt, st, nullptr, ld1);
        case PN2_ARITH_FMA: return launch_ball_query_grid<PN2_ARITH_FMA>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, pts_cnt, st, nullptr, ld1);
        case PN2_ARITH_FMA_ALT: return launch_ball_query_grid<PN2_ARITH_FMA_ALT>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, pts_cnt, st, nullptr, ld1);
        default: return PN2_EINVAL;
    }
}

// Bin a batch of clouds once for a radius (see BqgLds): workspace = b * pn2_ball_query_bin_bytes(n) bytes, 256-byte aligned.
// PN2_EUNSUP when the shape is outside the LDS-grid kernel's range (the caller then uses pn2_query_ball_point).
extern "C" size_t pn2_ball_query_bin_bytes(int n) { return (n >= 1 && n <= kBqgMaxN) ? bqg_ws_stride(n) : 0; }

// pn2_ball_query_bin with the cloud's rows ld1 floats apart (ld1 >= 3: the xyz columns of a (b,n,6) xyz+rgb batch read in place,
// model.py:26-29); the bins are those of a dense copy, bit for bit.
extern "C" int pn2_ball_query_bin_ld(int b, int n, float radius, const float* xyz1, int ld1, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (b <= 0 || n <= 0 || ld1 < 3) return PN2_EINVAL;
    if (!(radius > 0.0f)) return PN2_EINVAL;
    if (!xyz1 || !workspace) return PN2_ENULL;
    if (n > kBqgMaxN || !(radius < 1e18f)) return PN2_EUNSUP;
    if (b > 65535 || (long long)n * ld1 > 0x7fffffffLL) return PN2_ERANGE;
    const size_t stride = bqg_ws_stride(n);
    if (workspace_bytes < stride * (size_t)b || ((uintptr_t)workspace & 255) != 0) return PN2_EINVAL;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ball_query_bin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    ball_query_bin_kernel<<<b, kBqgThreads, bqg_lds_bytes((n + 63) & ~63), static_cast<hipStream_t>(stream)>>>(
        n, radius, xyz1, ld1, static_cast<unsigned char*>(workspace), stride);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_ball_query_bin(int b, int n, float radius, const float* xyz1, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    return pn2_ball_query_bin_ld(b, n, radius, xyz1, 3, workspace, workspace_bytes, stream);
}

// query_ball_point on a cloud binned by pn2_ball_query_bin(b, n, radius, xyz1, ...) -- same radius, same xyz1: the query
// workgroups copy the cell-sorted cloud instead of re-binning it.  Bit-identical to pn2_query_ball_point.
extern "C" int pn2_query_ball_point_binned(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2,
                                           const void* bins, int* idx, int* pts_cnt, int arith_mode, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0) return PN2_EINVAL;
    if (!(radius > 0.0f)) return PN2_EINVAL;
    if (!xyz1 || !xyz2 || !idx || !pts_cnt || !bins) return PN2_ENULL;
    if (n > kBqgMaxN || nsample > 64 || !(radius < 1e18f)) return PN2_EUNSUP;
    if ((long long)m * 3 > 0x7fffffffLL || b > 65535) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float thr = ball_threshold(radius);
    unsigned char* ws = const_cast<unsigned char*>(static_cast<const unsigned char*>(bins));
    switch (arith_mode) {
        case PN2_ARITH_STRICT: return launch_ball_query_grid<PN2_ARITH_STRICT>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, pts_cnt, st, ws);
        case PN2_ARITH_FMA: return launch_ball_query_grid<PN2_ARITH_FMA>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, pts_cnt, st, ws);
        case PN2_ARITH_FMA_ALT: return launch_ball_query_grid<PN2_ARITH_FMA_ALT>(b, n, m, radius, thr, nsample, xyz1, xyz2, idx, pts_cnt, st, ws);
        default: return PN2_EINVAL;
    }
}

// Diagnostic door: the same operator on an explicitly chosen kernel (1 wave-per-queries scan, 2 lane-per-query scan,
// 3 LDS grid; a kernel whose preconditions do not hold falls through to the next one).  Stateless -- the parity tests
// use it to hold EVERY kernel to the oracle, not just the one the shape heuristic picks.
extern "C" int pn2_query_ball_point_kernel(int b, int n, int m, float radius, int nsample,
                                           const float* xyz1, const float* xyz2, int* idx, int* pts_cnt,
                                           int arith_mode, int kernel, void* stream) {
    if (kernel < 0 || kernel > 3) return PN2_EINVAL;
    return query_ball_point_impl(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt, arith_mode, kernel, stream);
}

// One scan of xyz1 for several (radius, nsample) pairs -- see include/pn2_abi.h.
extern "C" int pn2_query_ball_point_multi(int b, int n, int m, int nradius, const float* radii, const int* nsamples,
                                          const float* xyz1, const float* xyz2, int* const* idx, int* const* pts_cnt,
                                          int arith_mode, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nradius <= 0) return PN2_EINVAL;
    if (!radii || !nsamples || !xyz1 || !xyz2 || !idx || !pts_cnt) return PN2_ENULL;
    if (nradius > kBqmMaxR) return PN2_EUNSUP;
    if ((long long)n * 3 > 0x7fffffffLL || (long long)m * 3 > 0x7fffffffLL || b > 65535) return PN2_ERANGE;
    BqmParams p{};
    p.n = n; p.m = m; p.nr = nradius; p.xyz1 = xyz1; p.xyz2 = xyz2;
    p.seg = (((n + kBqmWaves - 1) / kBqmWaves) + 3) & ~3;
    if (p.seg > 65535) return PN2_ERANGE;
    for (int r = 0; r < nradius; ++r) {
        if (!(radii[r] > 0.0f) || nsamples[r] <= 0) return PN2_EINVAL;
        if (!idx[r] || !pts_cnt[r]) return PN2_ENULL;
        p.thr[r] = ball_threshold(radii[r]); p.ns[r] = nsamples[r]; p.idx[r] = idx[r]; p.cnt[r] = pts_cnt[r];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (arith_mode < 0 || arith_mode > 2) return PN2_EINVAL;
    // radii whose neighbourhoods are small and whose cloud fits LDS are answered by the per-workgroup grid kernel
    // (a few us each, see launch_ball_query); the remaining ones share one scan
    {
        BqmParams rest = p;
        int nrest = 0;
        for (int r = 0; r < nradius; ++r) {
            const bool grid_ok = g_bq_variant == 0 && n <= kBqgMaxN && n >= 4096 && m >= 256 && nsamples[r] <= 64 && radii[r] < 1e18f;
            if (grid_ok) {
                const int rc = pn2_query_ball_point(b, n, m, radii[r], nsamples[r], xyz1, xyz2, idx[r], pts_cnt[r], arith_mode, stream);
                if (rc != PN2_OK) return rc;
            } else {
                rest.thr[nrest] = p.thr[r]; rest.ns[nrest] = p.ns[r]; rest.idx[nrest] = p.idx[r]; rest.cnt[nrest] = p.cnt[r];
                ++nrest;
            }
        }
        if (nrest == 0) return PN2_OK;
        if (nrest < nradius) {
            if (nrest == 1) {
                for (int r = 0; r < nradius; ++r)
                    if (idx[r] == rest.idx[0])
                        return pn2_query_ball_point(b, n, m, radii[r], nsamples[r], xyz1, xyz2, idx[r], pts_cnt[r], arith_mode, stream);
            }
            p = rest; p.nr = nrest; nradius = nrest;
        }
    }
    if (nradius == 1) return pn2_query_ball_point(b, n, m, radii[0], nsamples[0], xyz1, xyz2, idx[0], pts_cnt[0], arith_mode, stream);
#define PN2_BQM(MODE_) (nradius == 2 ? launch_ball_query_multi<MODE_, 2>(b, p, st) : launch_ball_query_multi<MODE_, 3>(b, p, st))
    switch (arith_mode) {
        case PN2_ARITH_STRICT: return PN2_BQM(PN2_ARITH_STRICT);
        case PN2_ARITH_FMA: return PN2_BQM(PN2_ARITH_FMA);
        case PN2_ARITH_FMA_ALT: return PN2_BQM(PN2_ARITH_FMA_ALT);
        default: return PN2_EINVAL;
    }
#undef PN2_BQM
}

// selectionSortLauncher(b,n,m,k,dist,outi,out)  tf_grouping.cu:145-149, tf_grouping.cpp:135
extern "C" int pn2_selection_sort(int b, int n, int m, int k, const float* dist, int* outi, float* out,
                                  void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || k <= 0) return PN2_EINVAL;  // "SelectionSort expects positive k" :142-144
    if (!dist || !outi || !out) return PN2_ENULL;
    const size_t lds = (size_t)n * 8;
    if (lds > 150 * 1024 || m > 65535 * 32 || b > 65535) return PN2_ERANGE;  // row must fit LDS (n <= 19200)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(selection_sort_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(m, b);
    selection_sort_kernel<<<grid, 64, lds, static_cast<hipStream_t>(stream)>>>(n, m, k, dist, outi, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_group_point(int b, int n, int c, int m, int nsample, const float* points,
                               const int* idx, float* out, void* stream) {
    if (b <= 0 || n <= 0 || c <= 0 || m <= 0 || nsample <= 0) return PN2_EINVAL;
    if (!points || !idx || !out) return PN2_ENULL;
    const unsigned long long rows = (unsigned long long)m * nsample;
    if (rows * (unsigned long long)c > 0xffffffffull || b > 65535) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec4 = (c % 4 == 0) && (((uintptr_t)points | (uintptr_t)out) % 16 == 0);
    if (vec4) {
        const unsigned long long tot = rows * (c / 4);
        int gx = 1;
        int bpc = g_gp_variant >> 4;  // blocks per CU overall; measured best at 64 (profiles/r01_group_point_sweep.txt)
        if (bpc <= 0) bpc = 64;
        {
            unsigned long long cap = (256ull * bpc + b - 1) / b, g = (tot / 4 + 255) / 256;
            gx = (int)(g < cap ? (g < 1 ? 1 : g) : cap);
        }
        dim3 grid(gx, b);
        if (g_gp_variant & 1) group_point_kernel<f32x4, 4, 4, false><<<grid, 256, 0, st>>>(n, c, (unsigned)rows, points, idx, out, (g_gp_variant & 2) ? 0 : 1);
        else group_point_kernel<f32x4, 4, 4, true><<<grid, 256, 0, st>>>(n, c, (unsigned)rows, points, idx, out, (g_gp_variant & 2) ? 0 : 1);
    } else {
        dim3 grid(grid_x_for(rows * c, 256, b), b);
        group_point_kernel<float, 1, 4, true><<<grid, 256, 0, st>>>(n, c, (unsigned)rows, points, idx, out, (g_gp_variant & 2) ? 0 : 1);
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_group_point_grad(int b, int n, int c, int m, int nsample, const float* grad_out,
                                    const int* idx, float* grad_points, void* stream) {
    if (b <= 0 || n <= 0 || c <= 0 || m <= 0 || nsample <= 0) return PN2_EINVAL;
    if (!grad_out || !idx || !grad_points) return PN2_ENULL;
    const unsigned long long rows = (unsigned long long)m * nsample;
    if (rows * (unsigned long long)c > 0xffffffffull || b > 65535) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, st);
    if (e != hipSuccess) return (int)e;
    const bool vec4 = (c % 4 == 0) && ((uintptr_t)grad_out % 16 == 0);
    if (vec4) {
        dim3 grid(grid_x_for(rows * (c / 4), 256, b), b);
        group_point_grad_kernel<f32x4, 4><<<grid, 256, 0, st>>>(n, c, (unsigned)rows, grad_out, idx, grad_points);
    } else {
        dim3 grid(grid_x_for(rows * c, 256, b), b);
        group_point_grad_kernel<float, 1><<<grid, 256, 0, st>>>(n, c, (unsigned)rows, grad_out, idx, grad_points);
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// internal helpers of pn2_interpolate.hip (list-and-gather row scatter)
extern "C" size_t pn2_scatter_rows_workspace_bytes(int b, int nent, int nsrc);
extern "C" int pn2_scatter_rows_gather(int b, int nent, int div, int c, int nsrc, const float* rows_in, const int* idx,
                                       const float* weight, float* out, void* workspace, void* stream);

extern "C" size_t pn2_group_point_grad_workspace_bytes(int b, int n, int m, int nsample) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0) return 0;
    return pn2_scatter_rows_workspace_bytes(b, m * nsample, n);
}

// group_point gradient with caller-provided scratch: large levels build a per-point list of the grouped rows that
// reference it and gather (no float atomics); small ones, c % 4 != 0 or workspace == NULL run pn2_group_point_grad.
extern "C" int pn2_group_point_grad_ws(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx,
                                       float* grad_points, void* workspace, size_t workspace_bytes, void* stream) {
    if (b <= 0 || n <= 0 || c <= 0 || m <= 0 || nsample <= 0) return PN2_EINVAL;
    if (!grad_out || !idx || !grad_points) return PN2_ENULL;
    const unsigned long long rows = (unsigned long long)m * nsample;
    const bool gather_ok = workspace && c % 4 == 0 && c <= 1024 && (((uintptr_t)grad_out | (uintptr_t)grad_points) % 16) == 0 &&
                           rows <= 0x7fffffffull && b <= 65535;
    if (!gather_ok || (unsigned long long)b * rows * c < (1ull << 20))
        return pn2_group_point_grad(b, n, c, m, nsample, grad_out, idx, grad_points, stream);
    if (workspace_bytes < pn2_group_point_grad_workspace_bytes(b, n, m, nsample) || ((uintptr_t)workspace % 4) != 0) return PN2_EINVAL;
    return pn2_scatter_rows_gather(b, (int)rows, 1, c, n, grad_out, idx, nullptr, grad_points, workspace, stream);
}
