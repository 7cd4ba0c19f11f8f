// pn2_three_nn.h -- the exact float64 3-NN search of pn2_three_nn as a per-wave device routine, shared by the stand-alone
// kernel (pn2_interpolate.hip) and the one-launch coarse-level geometry (pn2_coarse_geometry.hip).
#pragma once
#include <math.h>

#include "pn2_common.h"

namespace pn2nn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNnThreads = 256;
constexpr int kNnWaves = kNnThreads / 64;
constexpr int kNnQ = 8;        // queries per wave
constexpr int kNnList = 32;    // per-query candidate list (LDS)

struct NnPoint { float x, y, z; };

// wave64 maximum of non-negative floats (as int bits), uniform result
__device__ __forceinline__ int nn_wave_imax(int v) {
    asm volatile(
        "s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// v_min_f32 without the canonicalising v_max the compiler adds for llvm.minnum (operands are FMA results /
// +inf, never signalling NaNs)
__device__ __forceinline__ float nn_fmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float nn_fmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#ifndef PN2_NN_STAGES
#define PN2_NN_STAGES 9  // tools/nn_stage_ab.py builds truncated variants (timing breakdown only)
#endif

// value of lane perm(l) inside every group of 8 lanes; CTRL: 0xB1 = l^1, 0x4E = l^2, 0x141 = 7-l
template <int CTRL>
__device__ __forceinline__ float nn_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// (t0<=t1<=t2) <- three smallest of the multiset {t0,t1,t2} U {b0,b1,b2} (both sorted)
__device__ __forceinline__ void nn_merge3(float& t0, float& t1, float& t2, float b0, float b1, float b2) {
    const float r0 = fminf(t0, b0);
    const float r1 = fminf(fminf(t1, b1), fmaxf(t0, b0));
    const float r2 = fminf(fminf(fminf(t2, b2), fmaxf(t0, b1)), fmaxf(t1, b0));
    t0 = r0; t1 = r1; t2 = r2;
}

// Exact float64 3-NN (reference semantics: tf_interpolate.cpp:20-28 -> FLANN L2<double>,
// ((0+dx*dx)+dy*dy)+dz*dz, ties -> lowest index).  One wave owns kNnQ queries; the known points sit
// one per lane in registers (16 chunks = 1024 points at a time) and are shared by the wave's queries.
//
// The fp32 passes only RANK candidates, so they use the expanded form around a per-cloud centre o
// (= known point 0):  with Q = fl(q - o), C = fl(c - o),
//     s(q, c) = fma(-2Qx, Cx, fma(-2Qy, Cy, fma(-2Qz, Cz, |C|^2)))  ~  |q - c|^2 - |Q|^2
// -- 3 FMAs per pair instead of 3 sub + mul + 2 fma.  Error bound (u = 2^-24, R = largest |component| of
// any Q or C): centring moves the true distance by <= 24 u R^2, |C|^2 carries <= 9 u R^2, the three
// FMAs <= 27 u R^2, so  | s + |Q|^2 - |q - c|^2 | <= E := 64 u R^2  for every pair.
//   pass 1  every lane keeps min s over its own candidates; v3 = 3rd smallest of the 64 lane minima
//           (three distinct candidates), found for all 8 queries at once by a transposed top-3 merge
//           (LDS transpose, 8 lanes per query, 3 DPP merge steps).  The true 3rd-NN distance is
//           <= v3 + |Q|^2 + E, so every true top-3 candidate (ties included) has s <= v3 + 2E;
//           thr = v3 + 3E absorbs the rounding of the addition.
//   pass 2  the few lanes whose minimum is inside thr re-test their own candidates and append the hits
//           (s <= thr, a handful per query) to the query's list in LDS, in arbitrary order;
//   refine  8 lanes per query take the exact float64 distance of the listed candidates (raw coordinates re-read, all
//           loads of the wave in flight together) and rank them by (distance, index); ranks 0..2 are written --
//           exactly what a strict '<' insertion over a full ascending scan yields (ties -> lowest index).
// A list overflow (> 32 candidates inside thr: heavy duplication, or a dynamic range (extent/spacing)^2
// approaching 2^24 that makes E useless) falls back to a full float64 scan of that query by one lane.
// kNnChunks = candidate chunks (of 64) held in registers at a time (16 -> 1024 points; small known
// sets instantiate 4 or 1 so that the unrolled chunk loops do no dead work).
// Per-wave LDS scratch of three_nn_wave (16-byte aligned pieces).
struct NnWaveLds {
    int* wl;      // [kNnQ * kNnList] candidate lists
    double* wd;   // [kNnQ * kNnList] their float64 distances
    float* wm;    // [kNnQ * 64] lane minima, transposed selection (16-byte aligned)
    float* wq;    // [kNnQ * 4] (-2Qx, -2Qy, -2Qz, thr) per query (16-byte aligned)
    float* wr;    // [kNnQ * 4] raw query coordinates (16-byte aligned)
    int* wc;      // [kNnQ] list lengths
};
constexpr int kNnWaveLdsBytes = kNnQ * kNnList * 12 + kNnQ * 64 * 4 + kNnQ * 4 * 8 + kNnQ * 4;  // 5408
// carve wave w's scratch out of `base` (16-byte aligned, kNnWaveLdsBytes * waves bytes)
__device__ __forceinline__ NnWaveLds nn_wave_lds(unsigned char* base, int w) {
    unsigned char* p = base + (size_t)w * kNnWaveLdsBytes;
    NnWaveLds s;
    s.wd = reinterpret_cast<double*>(p);                  p += kNnQ * kNnList * 8;
    s.wm = reinterpret_cast<float*>(p);                   p += kNnQ * 64 * 4;
    s.wq = reinterpret_cast<float*>(p);                   p += kNnQ * 4 * 4;
    s.wr = reinterpret_cast<float*>(p);                   p += kNnQ * 4 * 4;
    s.wl = reinterpret_cast<int*>(p);                     p += kNnQ * kNnList * 4;
    s.wc = reinterpret_cast<int*>(p);
    return s;
}

// ONE wave's share of a three_nn call on one cloud: xyz1 (n,3) queries, xyz2 (m,3) known points, dist / idx (n,3) -- all
// already offset to the cloud -- query groups grp, grp + gstride, ... (wave-uniform).  No barriers.
template <int kNnChunks>
__device__ __forceinline__ void three_nn_wave(int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                              float* __restrict__ dist_all, int* __restrict__ idx_all, int grp, int gstride,
                                              const NnWaveLds& S, int ld1 = 3) {  // ld1: row stride of the queries in floats
    const int lane = threadIdx.x & 63;
    // PERSISTENT waves (r03): a wave keeps the candidates in registers (m <= 64 * kNnChunks: the common case) and walks
    // query groups grp, grp + stride, ...; the next group's coordinates are fetched (one coalesced load, 24 lanes) while
    // this group is worked on.  With one group per wave the kernel was bound by the latency of its own start-up loads
    // (48 candidate + 24 query loads in front of 0.3 us of pass-1 arithmetic), not by VALU issue.
    const int ngroups = (n + kNnQ - 1) / kNnQ;
    if (grp >= ngroups) return;  // wave-uniform, no barriers in this routine
    const NnPoint* __restrict__ cand = reinterpret_cast<const NnPoint*>(xyz2);
    int* wl = S.wl;
    double* wd = S.wd;
    float* wm = S.wm;
    float* wq = S.wq;
    float* wr = S.wr;
    int* wc = S.wc;

    const float ox = xyz2[0], oy = xyz2[1], oz = xyz2[2];
    const int last = m - 1;
    const int nblk = (m + 64 * kNnChunks - 1) / (64 * kNnChunks);
    float cx[kNnChunks], cy[kNnChunks], cz[kNnChunks], cc[kNnChunks];
    float rc = 0.f;  // largest |component| of this lane's centred candidates (grows monotonically: E stays a bound)
    auto load_block = [&](int blk) {
#pragma unroll
        for (int t = 0; t < kNnChunks; ++t) {
            const int k = (blk * kNnChunks + t) * 64 + lane;
            const NnPoint p = cand[k < last ? k : last];  // clamped; out-of-range lanes are masked by index below
            const float x = p.x - ox, y = p.y - oy, z = p.z - oz;
            cx[t] = x; cy[t] = y; cz[t] = z;
            cc[t] = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
            rc = fmaxf(rc, fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))));
        }
    };
    // lane l < 24 holds coordinate l % 3 of query q0 + l / 3 (queries past n re-read the last one; never written)
    auto load_queries = [&](int g) {
        const int l = lane < 3 * kNnQ ? lane : 3 * kNnQ - 1;
        int jq = g * kNnQ + l / 3;
        jq = jq < n ? jq : n - 1;
        return xyz1[jq * ld1 + l % 3];
    };
    float qnext = load_queries(grp);
    if (nblk == 1) load_block(0);
  for (; grp < ngroups; grp += gstride) {
    const int q0 = grp * kNnQ;
    const float qv = qnext;
    if (grp + gstride < ngroups) qnext = load_queries(grp + gstride);
    float ax[kNnQ], ay[kNnQ], az[kNnQ];
    float rq = 0.f;
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) {
        const float rx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qv), 3 * q + 0));
        const float ry = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qv), 3 * q + 1));
        const float rz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qv), 3 * q + 2));
        if (lane == 0) *reinterpret_cast<f32x4*>(wr + q * 4) = f32x4{rx, ry, rz, 0.f};
        const float x = rx - ox, y = ry - oy, z = rz - oz;
        ax[q] = -2.0f * x; ay[q] = -2.0f * y; az[q] = -2.0f * z;  // exact scalings
        rq = fmaxf(rq, fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))));
    }
    auto rank = [&](int q, int t) {
        return __builtin_fmaf(ax[q], cx[t], __builtin_fmaf(ay[q], cy[t], __builtin_fmaf(az[q], cz[t], cc[t])));
    };

    // ---- pass 1 -------------------------------------------------------------------------------
    float mn[kNnQ];
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) mn[q] = INFINITY;
    for (int blk = 0; blk < nblk; ++blk) {
        if (nblk > 1) load_block(blk);  // single block: resident for the whole kernel
        auto one_chunk = [&](int t) {
            const int k = (blk * kNnChunks + t) * 64 + lane;
            const int cbase = (blk * kNnChunks + t) * 64;
            if (cbase + 64 <= m) {  // full chunk (wave-uniform): no masking
#pragma unroll
                for (int q = 0; q < kNnQ; ++q) mn[q] = nn_fmin(mn[q], rank(q, t));
            } else if (cbase < m) {  // partial last chunk: lanes past m re-read point m-1 and must not count
                const bool valid = k < m;
#pragma unroll
                for (int q = 0; q < kNnQ; ++q) mn[q] = nn_fmin(mn[q], valid ? rank(q, t) : INFINITY);
            }
        };
        if constexpr (kNnChunks >= 2) {
#pragma unroll
            for (int t = 0; t < kNnChunks; t += 2) {
                if ((blk * kNnChunks + t + 2) * 64 <= m) {  // two full chunks (wave-uniform): one v_min3 per query
#pragma unroll
                    for (int q = 0; q < kNnQ; ++q) mn[q] = nn_fmin3(mn[q], rank(q, t), rank(q, t + 1));
                } else {
                    one_chunk(t);
                    one_chunk(t + 1);
                }
            }
        } else {
            one_chunk(0);
        }
    }
    if constexpr (PN2_NN_STAGES < 2) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < kNnQ; ++q) acc += mn[q];
        if (acc == 12345.f) dist_all[lane] = acc;
        continue;
    }
    // E = 64 u R^2 (rounded up)
    const float R = fmaxf(__int_as_float(nn_wave_imax(__float_as_int(rc))), rq);
    const float E3 = R * R * (3.0f * 64.0f * 5.9604645e-8f * 1.001f);
    // transposed selection: lane (g = lane>>3, part = lane&7) takes 8 of query g's 64 lane minima
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) wm[q * 64 + lane] = mn[q];
    float t0, t1, t2;
    {
        const int g = lane >> 3, part = lane & 7;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(wm + g * 64 + part * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(wm + g * 64 + part * 8 + 4);
        t0 = fminf(fminf(v0[0], v0[1]), v0[2]);
        t2 = fmaxf(fmaxf(v0[0], v0[1]), v0[2]);
        t1 = __builtin_amdgcn_fmed3f(v0[0], v0[1], v0[2]);
        const float rest[5] = {v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float d = rest[i];
            const float n0 = fminf(t0, d);
            const float n1 = __builtin_amdgcn_fmed3f(t0, t1, d);
            const float n2 = __builtin_amdgcn_fmed3f(t1, t2, d);
            t0 = n0; t1 = n1; t2 = n2;
        }
        nn_merge3(t0, t1, t2, nn_dpp<0xB1>(t0), nn_dpp<0xB1>(t1), nn_dpp<0xB1>(t2));
        nn_merge3(t0, t1, t2, nn_dpp<0x4E>(t0), nn_dpp<0x4E>(t1), nn_dpp<0x4E>(t2));
        nn_merge3(t0, t1, t2, nn_dpp<0x141>(t0), nn_dpp<0x141>(t1), nn_dpp<0x141>(t2));
    }
    float thr[kNnQ];
#pragma unroll
    for (int q = 0; q < kNnQ; ++q)
        thr[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t2), q * 8)) + E3;  // +inf stays +inf

    if constexpr (PN2_NN_STAGES < 3) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < kNnQ; ++q) acc += thr[q];
        if (acc == 12345.f) dist_all[lane] = acc;
        continue;
    }
    // ---- pass 2: collect ----------------------------------------------------------------------
    // Only lanes whose pass-1 minimum is inside thr hold a candidate of that query (typically 3-5 of
    // the 64 lanes), so instead of re-testing every (query, chunk) pair wave-wide, every lane walks its
    // own short work list of queries (bit q of `todo`), re-evaluates s for its <= kNnChunks candidates
    // with the query's parameters fetched from LDS, and appends the hits through an LDS counter.  The
    // list order is arbitrary; the refine step orders by (distance, index).
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) {
        if (lane == 0) {
            *reinterpret_cast<f32x4*>(wq + q * 4) = f32x4{ax[q], ay[q], az[q], thr[q]};
            wc[q] = 0;
        }
    }
    unsigned todo = 0u;
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) todo |= (mn[q] <= thr[q] ? 1u : 0u) << q;
    for (int blk = 0; blk < nblk; ++blk) {
        if (nblk > 1) load_block(blk);  // single block: still resident from pass 1
        // chunks t < nvalid of this block hold a real candidate for this lane (k < m)
        int nvalid = (m - lane - blk * kNnChunks * 64 + 63) >> 6;
        nvalid = nvalid < 0 ? 0 : (nvalid > kNnChunks ? kNnChunks : nvalid);
        const unsigned vmask = (1u << nvalid) - 1u;
        unsigned td = todo;
        while (td != 0u) {
            const int q = __ffs(td) - 1;
            td &= td - 1u;
            const f32x4 qp = *reinterpret_cast<const f32x4*>(wq + q * 4);
            unsigned hm = 0u;
#pragma unroll
            for (int t = 0; t < kNnChunks; ++t) {
                const float sv = __builtin_fmaf(qp[0], cx[t], __builtin_fmaf(qp[1], cy[t], __builtin_fmaf(qp[2], cz[t], cc[t])));
                hm |= (sv <= qp[3] ? 1u : 0u) << t;
            }
            hm &= vmask;
            while (hm != 0u) {
                const int t = __ffs(hm) - 1;
                hm &= hm - 1u;
                const int kk = (blk * kNnChunks + t) * 64 + lane;
                const int pos = atomicAdd(&wc[q], 1);  // may exceed kNnList: overflow marker
                if (pos < kNnList) wl[q * kNnList + pos] = kk;  // the exact distance is taken in the refine step
            }
        }
    }
    int cnt[kNnQ];
#pragma unroll
    for (int q = 0; q < kNnQ; ++q) cnt[q] = __builtin_amdgcn_readfirstlane(wc[q]);

    if constexpr (PN2_NN_STAGES < 4) {
        int acc = 0;
#pragma unroll
        for (int q = 0; q < kNnQ; ++q) acc += cnt[q];
        if (acc == 12345) idx_all[lane] = acc + wl[lane];
        continue;
    }
    // ---- refine: 8 lanes per query; every listed candidate is ranked by (distance, index) ---------
    // rank = number of listed candidates that precede it; ranks 0..2 are the answer -- exactly what a
    // strict '<' insertion over a full ascending scan yields (ties keep the lowest index).  Indices are
    // distinct, so ranks are distinct; the three candidates that defined v3 are always listed.
    {
        const int g = lane >> 3, part = lane & 7;
        int cg = 0;
#pragma unroll
        for (int q = 0; q < kNnQ; ++q) cg = g == q ? cnt[q] : cg;
        if (q0 + g < n) {
            const size_t o = (size_t)(q0 + g) * 3;
            if (cg <= kNnList) {
                // exact float64 distances from the RAW coordinates (the centred ones are rounded): every lane fetches its
                // own entries, all of a wave's loads in flight together (inside pass 2's divergent loop each hit paid
                // its own global round trip)
                const double qx = (double)wr[g * 4 + 0], qy = (double)wr[g * 4 + 1], qz = (double)wr[g * 4 + 2];
                for (int e = part; e < cg; e += 8) {
                    const int kk = wl[g * kNnList + e];
                    const double dx = qx - (double)xyz2[kk * 3 + 0];
                    const double dy = qy - (double)xyz2[kk * 3 + 1];
                    const double dz = qz - (double)xyz2[kk * 3 + 2];
                    wd[g * kNnList + e] = (dx * dx + dy * dy) + dz * dz;  // contraction is off
                }
                for (int e = part; e < cg; e += 8) {
                    const double d = wd[g * kNnList + e];
                    const int kk = wl[g * kNnList + e];
                    int rank = 0;
                    for (int j0 = 0; j0 < cg; j0 += 8) {  // 8 entries per LDS round trip (slots past cg: stale, masked)
                        double dj[8];
                        int kj[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { dj[u] = wd[g * kNnList + j0 + u]; kj[u] = wl[g * kNnList + j0 + u]; }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            rank += (j0 + u < cg && (dj[u] < d || (dj[u] == d && kj[u] < kk))) ? 1 : 0;
                    }
                    if (rank < 3) { dist_all[o + rank] = (float)d; idx_all[o + rank] = kk; }
                }
            } else if (part == 0) {  // overflow: full float64 scan in ascending index order
                const int jq = q0 + g;
                const double dqx = xyz1[jq * ld1 + 0], dqy = xyz1[jq * ld1 + 1], dqz = xyz1[jq * ld1 + 2];
                double b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
                int i1 = 0, i2 = 0, i3 = 0;
                for (int kk = 0; kk < m; ++kk) {
                    const double dx = dqx - (double)xyz2[kk * 3 + 0];
                    const double dy = dqy - (double)xyz2[kk * 3 + 1];
                    const double dz = dqz - (double)xyz2[kk * 3 + 2];
                    const double d = (dx * dx + dy * dy) + dz * dz;  // contraction is off
                    if (d < b3) {  // strict: ties keep the lowest index
                        if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = kk; }
                        else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = kk; }
                        else { b3 = d; i3 = kk; }
                    }
                }
                dist_all[o + 0] = (float)b1; dist_all[o + 1] = (float)b2; dist_all[o + 2] = (float)b3;
                idx_all[o + 0] = i1; idx_all[o + 1] = i2; idx_all[o + 2] = i3;
            }
        }
    }
  }  // query groups
}

}  // namespace pn2nn
