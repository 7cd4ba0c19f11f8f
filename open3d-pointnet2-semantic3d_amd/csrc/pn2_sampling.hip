// pn2_sampling.hip -- farthest point sampling, gather_point and its gradient.
// MI355X-native replacements for tf_ops/tf_sampling.cu:111-206 (reference).
//
// FPS design (one workgroup per batch element, everything on-chip):
//   * each thread keeps PPT points (x,y,z) and their running min distance in
//     VGPRs for the whole kernel -- the reference re-reads/re-writes a global
//     `temp` row every round (tf_sampling.cu:139,152);
//   * a round = PPT distance updates per thread, a wave64 DPP max, ONE barrier,
//     a 16-lane DPP max over the per-wave candidates; the winner's coordinates
//     come from an LDS float4 copy of the cloud (one broadcast ds_read_b128);
//   * the reference tie-break (max distance, then k mod 512, then k -- the
//     512-thread strided scan + left-biased tree of tf_sampling.cu:153-170) is
//     reproduced exactly: thread t owns k = t + NT*i (NT a multiple of 512, so all
//     its points share the residue t mod 512 and are scanned in ascending k with
//     a strict '>'), lanes of a wave have ascending distinct residues (lowest
//     lane among equals = lowest residue), and waves are merged on the 64-bit
//     key (dist bits << 32 | ~((k&511)<<22 | k>>9)).
// The kernel is latency-bound (m-1 dependent rounds), not HBM-bound: its HBM
// traffic is b*n*12 + b*m*4 bytes in total.
#include "pn2_fps_common.h"
#include "pn2_fps_reg.h"

namespace {

using namespace pn2fpsreg;

using pn2fps::kLazyCap;
using pn2fps::wave_imax_from;
using pn2fps::wave_umax_all;

// max over lanes 0..15 (row 0), result from lane 15
__device__ __forceinline__ int row0_imax(int v) {
    asm volatile(
        "s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 15);
}
__device__ __forceinline__ unsigned row0_umin(unsigned v) {
    asm volatile(
        "s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}

// Merge the per-wave candidates {dist bits, tiekey}; returns the winning index (uniform).
// Value-only max first; the 64-bit order (dist desc, tiekey asc) is only resolved on ties.
__device__ __forceinline__ int fps_merge_waves(const uint2* slots, int nwaves, int lane) {
    uint2 sv = make_uint2(0x80000000u, 0xFFFFFFFFu);  // "no candidate": negative as int
    if (lane < nwaves) sv = slots[lane];
    const int vmax = row0_imax((int)sv.x);
    const unsigned long long tie = __ballot((int)sv.x == vmax) & 0xFFFFull;
    unsigned key;
    if (__popcll(tie) == 1) {
        key = (unsigned)__builtin_amdgcn_readlane((int)sv.y, __ffsll((long long)tie) - 1);
    } else {
        key = row0_umin((int)sv.x == vmax ? sv.y : 0xFFFFFFFFu);
    }
    return fps_untiekey(key);
}

// Generic per-wave candidate from a per-thread (best, bestk) pair (used by the streaming kernel).
__device__ __forceinline__ uint2 fps_wave_candidate(float best, int bestk) {
    const int bits = __float_as_int(best);  // -1.0f sentinel is negative
    const int wmax = wave_imax(bits);
    if (wmax < 0) return make_uint2(0x80000000u, 0xFFFFFFFFu);
    const unsigned long long mask = __ballot(bits == wmax);
    const int src = __ffsll((long long)mask) - 1;  // lowest lane = lowest residue (NT % 512 == 0)
    const int kw = __builtin_amdgcn_readlane(bestk, src);
    return make_uint2((unsigned)wmax, fps_tiekey(kw));
}

__device__ __forceinline__ float wave_fmin_all(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_fmax_all(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// fps_reg_kernel: pn2fpsreg::fps_reg_body on cloud blockIdx.x, after the nested-sampling shortcut
template <int NT, int PPT, int MODE, bool LDS_XYZ, bool TRACK>
__global__ void __launch_bounds__(NT)
fps_reg_kernel(int n, int m, const float* __restrict__ xyz_all, int ld, int* __restrict__ out_all,
               float* __restrict__ new_xyz_all, const int* __restrict__ tie_in, int* __restrict__ tie_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * ld;
    int* __restrict__ out = out_all + (size_t)blockIdx.x * m;
    float* __restrict__ nxyz = new_xyz_all ? new_xyz_all + (size_t)blockIdx.x * m * 3 : nullptr;
    if (fps_nested_shortcut(n, m, xyz, out, nxyz, tie_in, tie_out, NT, ld)) return;
    fps_reg_body<NT, PPT, MODE, LDS_XYZ, TRACK>(n, m, xyz, out, nxyz, tie_out ? tie_out + blockIdx.x : nullptr, smem, ld);
}

PN2_TUNABLE(int, g_fps_variant, 0)  // tuning hook (pn2_debug_set(0, v)), see dispatch_fps

// ---- lazy multi-pick FPS (2048 < n <= 8192) --------------------------------------------------------------------------
// Same picks as fps_reg_kernel (= the reference, tf_sampling.cu:111-176), but a synchronised pass over the cloud delivers
// ~15 picks instead of one.  min() is associative, so the running minimum td(k) may lag behind the picks: td_stale(k) >=
// td(k), and a point whose stale td is below a threshold tau cannot be the next pick while some listed point stays >= tau.
//   PHASE A (all waves): apply the PENDING picks of the previous phase to the points, then LIST every point with
//     td >= tau (64-bit key = (td bits : ~tiekey) as everywhere in this file) in LDS; every wave also publishes its own
//     maximum key.                                                                                  -- barrier --
//   PHASE B (wave 0, lane = candidate): repeatedly take the 64-bit maximum of the list (= the global maximum: unlisted
//     points have td < tau), emit it as the next pick, and lower the candidates' td by their distance to it -- a ~25
//     instruction dependent chain per pick with no barrier and no pass over the cloud -- until the best candidate
//     drops below tau (or 64 picks).  The picks become the pending list.                            -- barrier --
//   tau = (1 - eps) * (td of the last pick); eps follows the list length (target 12..42 of 64 entries).  An empty or
//   overflowing list falls back to ONE pick from the 16 per-wave maxima (exact: the global maximum is one of them).
// The result never depends on tau / eps / list capacity, only the number of phases does (tools/fps_lazy_sim.py: 1023
// picks in ~60 phases at n = 8192 on scene-, normal-, uniform- and lattice-distributed clouds).
// Phase A is pruned: the cloud is sorted along a Hilbert curve inside the workgroup (LDS counting sort over 16^3 cells of
// the bounding box) and dealt out so that (wave w, register row i) holds 64 consecutive sorted points = one compact BUCKET with an
// exact bounding box in SGPRs.  A pending pick p can lower td inside a bucket only if lb(p, box)^2 * (1 - 1e-6) <= G, G =
// the td of the first pending pick when it was picked (>= every stale td).  Lane p tests pick p against row i's box (one
// VALU pass per row, ballot = work mask of the row); only the surviving (row, pick) pairs -- ~12 % of them -- run the
// distance update.  Skipped pairs are provably no-ops, so the result is bit-identical.  Neighbouring buckets go to
// different waves (bucket q -> wave q % NW) so the buckets around a pick spread over the SIMDs.
// Tie-break (max td, then lowest k mod 512, then lowest k; tf_sampling.cu:153-170) rides in the low word of every key.
__device__ __forceinline__ unsigned fps_spread4(unsigned v) {  // b3b2b1b0 -> bits 9,6,3,0
    v = (v | (v << 4)) & 0x0C3u;
    return (v | (v << 2)) & 0x249u;
}
// Index of cell (x, y, z) of a 16^3 grid along a 3-D Hilbert curve (Skilling's axes-to-transpose transform, 4 bits per
// axis).  Unlike the Morton order it has no jumps, so 64 consecutive points never straddle two distant octants: the
// buckets that did made their waves evaluate 2-3x the (row, pick) pairs of the others (tools/fps_lazy_sim.py).
__device__ __forceinline__ unsigned fps_hilbert4(unsigned x0, unsigned x1, unsigned x2) {
    unsigned X[3] = {x0, x1, x2};
#pragma unroll
    for (unsigned Q = 8; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    unsigned t = 0;
#pragma unroll
    for (unsigned Q = 8; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return (fps_spread4(X[0]) << 2) | (fps_spread4(X[1]) << 1) | fps_spread4(X[2]);
}
constexpr int kFpsCells = 4096;  // 16 x 16 x 16 cells, Hilbert-ordered
PN2_TUNABLE(long long*, g_fps_stats, nullptr)  // tuning builds: device buffer of 16 counters written by block 0 (tools/fps_ab.py)
constexpr int kLazyHead = 2176;  // bytes in front of the cloud copy (ctrl, wcand, cand, pend, bbw, wsum)

inline size_t fps_lazy_bytes(int n, int m) {
    size_t r = (size_t)kFpsCells * 4;
    if ((size_t)n * 2 > r) r = (size_t)n * 2;
    if ((size_t)m * 4 > r) r = (size_t)m * 4;
    return kLazyHead + (size_t)n * 16 + r;
}

// Set-up shared by the lazy kernels: load the cloud (NT threads, LPT points each), copy it to LDS, sort it along the
// Hilbert curve (LDS counting sort over 16^3 cells) and deal the buckets (64 consecutive sorted points) to the worker
// waves -- worker wi (wave-uniform, -1 for a wave that owns no rows) holds bucket q = i * NWK + wi in register row i -- with
// the exact box of the lane's own row (lane l: row l / (64 / PPT)).  Leaves hist / perm dead (the caller may alias them).
template <int NT, int LPT, int PPT, int NWK>
__device__ __forceinline__ void fps_lazy_setup(int n, int wi, const float* __restrict__ xyz, int ld, float4* sxyz, int* hist, float* bbw, int* wsum,
                                               float (&px)[PPT], float (&py)[PPT], float (&pz)[PPT], double (&mk)[PPT],
                                               float& bx0, float& by0, float& bz0, float& bx1, float& by1, float& bz1) {
    constexpr int NW = NT / 64;
    constexpr int EPT = (kFpsCells + NT - 1) / NT;  // histogram entries per thread in the scan
    static_assert(NT % 64 == 0 && NW <= 16 && NWK <= NW && NT * LPT >= NWK * PPT * 64 && NWK * PPT * 64 <= 16384, "16 wave slots");
    unsigned short* perm = reinterpret_cast<unsigned short*>(hist);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float lx[LPT], ly[LPT], lz[LPT];
    // ---- 1. load in original order, cloud copy in LDS, bounding box ----------------------------------------------
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int k = tid + NT * i;
        lx[i] = ly[i] = lz[i] = 0.f;
        if (k < n) {
            lx[i] = xyz[k * ld + 0]; ly[i] = xyz[k * ld + 1]; lz[i] = xyz[k * ld + 2];
            sxyz[k] = make_float4(lx[i], ly[i], lz[i], 0.f);
            lo[0] = fminf(lo[0], lx[i]); hi[0] = fmaxf(hi[0], lx[i]);
            lo[1] = fminf(lo[1], ly[i]); hi[1] = fmaxf(hi[1], ly[i]);
            lo[2] = fminf(lo[2], lz[i]); hi[2] = fmaxf(hi[2], lz[i]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = wave_fmin_all(lo[a]); hi[a] = wave_fmax_all(hi[a]); }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { bbw[wave * 6 + a] = lo[a]; bbw[wave * 6 + 3 + a] = hi[a]; }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) if (tid + NT * e < kFpsCells) hist[tid + NT * e] = 0;
    __syncthreads();
    float scl;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = bbw[a], h = bbw[3 + a];
        for (int w = 1; w < NW; ++w) { l = fminf(l, bbw[w * 6 + a]); h = fmaxf(h, bbw[w * 6 + 3 + a]); }
        lo[a] = l;
        hi[a] = h - l;
    }
    {   // cubic cells (one scale for the three axes): buckets come out compact in every direction
        const float ext = fmaxf(fmaxf(hi[0], hi[1]), hi[2]);
        scl = ext > 0.f ? 16.0f / ext : 0.f;  // degenerate cloud (or inf/garbage): everything in cell 0
        if (!(scl < 3e38f)) scl = 0.f;
    }
    // ---- 2. counting sort by Hilbert cell: rank within the cell from the LDS atomic, exclusive scan of the histogram
    //         (the order inside a cell is arbitrary: only locality depends on it, never the result)
    int code[LPT], rnk[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int k = tid + NT * i;
        code[i] = 0; rnk[i] = 0;
        if (k < n) {
            int c[3];
            const float q[3] = {lx[i], ly[i], lz[i]};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float f = (q[a] - lo[a]) * scl;
                int ci = (int)f;
                ci = ci < 0 ? 0 : (ci > 15 ? 15 : ci);
                if (!(f == f)) ci = 0;
                c[a] = ci;
            }
            code[i] = (int)fps_hilbert4((unsigned)c[0], (unsigned)c[1], (unsigned)c[2]);
            rnk[i] = atomicAdd(&hist[code[i]], 1);
        }
    }
    __syncthreads();
    {
        int v[EPT], sum = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) { v[e] = tid * EPT + e < kFpsCells ? hist[tid * EPT + e] : 0; sum += v[e]; }
        int inc = sum;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int run = base + inc - sum;
#pragma unroll
        for (int e = 0; e < EPT; ++e) { if (tid * EPT + e < kFpsCells) hist[tid * EPT + e] = run; run += v[e]; }
    }
    __syncthreads();
    int pos[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) pos[i] = hist[code[i]] + rnk[i];
    __syncthreads();  // hist is dead from here: the permutation aliases it
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int k = tid + NT * i;
        if (k < n) perm[pos[i]] = (unsigned short)k;
    }
    __syncthreads();
    // ---- 3. deal the sorted cloud; exact bucket boxes (an empty bucket keeps the inverted box: lb = +inf, never touched)
    // Box test layout: lane l tests row l / PP against pick slot l % PP, so ONE pass of ~15 VALU tests PP = 64 / PPT
    // pending picks against all the rows of the wave; the lane keeps only its own row's box.
    constexpr int PP = 64 / PPT;
    static_assert(PPT >= 2 && PPT * PP == 64, "rows per wave: 2, 4, 8 or 16");
    bx0 = by0 = bz0 = 3e38f; bx1 = by1 = bz1 = -3e38f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = (i * NWK + wi) * 64 + lane;
        float l0 = 3e38f, l1 = 3e38f, l2 = 3e38f, h0 = -3e38f, h1 = -3e38f, h2 = -3e38f;
        if (wi >= 0 && p < n) {
            const int k = perm[p];
            const float4 q = sxyz[k];
            px[i] = q.x; py[i] = q.y; pz[i] = q.z;
            mk[i] = __hiloint2double(__float_as_int(1e38f), (int)~fps_tiekey(k));  // tf_sampling.cu:124-126
            l0 = h0 = q.x; l1 = h1 = q.y; l2 = h2 = q.z;
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            mk[i] = __hiloint2double(__float_as_int(-1.0f), 0);  // no point: a negative high word, below every real key
        }
        l0 = wave_fmin_all(l0); l1 = wave_fmin_all(l1); l2 = wave_fmin_all(l2);
        h0 = wave_fmax_all(h0); h1 = wave_fmax_all(h1); h2 = wave_fmax_all(h2);
        if (lane / PP == i) { bx0 = l0; by0 = l1; bz0 = l2; bx1 = h0; by1 = h1; bz1 = h2; }
    }
    __syncthreads();  // perm is dead from here
}

template <int NT, int PPT, int MODE, bool TRACK>
__global__ void __launch_bounds__(NT)
fps_lazy_kernel(int n, int m, const float* __restrict__ xyz_all, int ld, int* __restrict__ out_all,
                float* __restrict__ new_xyz_all, long long* __restrict__ stats, const int* __restrict__ tie_in,
                int* __restrict__ tie_out) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: int ctrl[16] | u64 wcand[16] | u64 cand[64] | float4 pend[64] | float bbw[16][6], int wsum[16] |
    //         float4 sxyz[n] | R: int hist[4096] -> u16 perm[n] -> int spick[m]
    int* ctrl = reinterpret_cast<int*>(smem);  // [0] cnt (LDS atomic) | [4..7] np, j, tau_hi, G bits (one 16-byte read)
    unsigned long long* wcand = reinterpret_cast<unsigned long long*>(smem + 64);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem + 192);
    float4* pend = reinterpret_cast<float4*>(smem + 704);
    float* bbw = reinterpret_cast<float*>(smem + 1728);
    int* wsum = reinterpret_cast<int*>(bbw + 6 * 16);
    float4* sxyz = reinterpret_cast<float4*>(smem + kLazyHead);
    int* hist = reinterpret_cast<int*>(sxyz + n);
    int* spick = hist;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * ld;
    int* __restrict__ out = out_all + (size_t)blockIdx.x * m;
    float* __restrict__ nxyz = new_xyz_all ? new_xyz_all + (size_t)blockIdx.x * m * 3 : nullptr;
    if (fps_nested_shortcut(n, m, xyz, out, nxyz, tie_in, tie_out, NT, ld)) return;
    constexpr bool track = TRACK;
    int* ttl = ctrl + 8;                          // pn2fps::tie_* record {strict, benign, zero}, written by wave 0
    int* wtie = reinterpret_cast<int*>(bbw);      // per-wave tie class of the one-pick fallback (bbw is dead after the set-up)

    float px[PPT], py[PPT], pz[PPT];
    double mk[PPT];
    float bx0, by0, bz0, bx1, by1, bz1;
    constexpr int PP = 64 / PPT;
    fps_lazy_setup<NT, PPT, PPT, NW>(n, wave, xyz, ld, sxyz, hist, bbw, wsum, px, py, pz, mk, bx0, by0, bz0, bx1, by1, bz1);
    if (tid == 0) {
        spick[0] = 0;  // first pick is index 0 (tf_sampling.cu:122-123)
        pend[0] = sxyz[0];
        ctrl[0] = 0; ctrl[1] = 0;
        pn2fps::tie_init(ttl);
        *reinterpret_cast<int4*>(ctrl + 4) = make_int4(1, 1, __float_as_int(1e38f), __float_as_int(1e38f));
    }
    __syncthreads();

    // ---- 4. phases ----------------------------------------------------------------------------------------------------
    float eps = 0.2f;  // wave 0 only
#ifdef PN2_TUNING_HOOKS
    const bool do_stats = stats != nullptr;
#else
    constexpr bool do_stats = false;  // the counters below compile away in the shipped library
#endif
    long long st_a = 0, st_w1 = 0, st_b = 0, st_w2 = 0, st_ph = 0, st_empty = 0, st_over = 0, st_cnt = 0, st_pairs = 0, st_upd = 0;
    const long long st_t00 = do_stats ? (long long)__builtin_readcyclecounter() : 0;
    // the list counter alternates between ctrl[0] and ctrl[1]: a phase appends to one while wave 0 clears the other,
    // whose last readers passed the previous phase's closing barrier
    for (int ph = 0;; ph ^= 1) {
        const long long st_t0 = do_stats ? (long long)__builtin_readcyclecounter() : 0;
        const int4 cw = *reinterpret_cast<const int4*>(ctrl + 4);
        const int np = __builtin_amdgcn_readfirstlane(cw.x);
        const int jdone = __builtin_amdgcn_readfirstlane(cw.y);
        if (jdone >= m) break;
        const int tau_hi = __builtin_amdgcn_readfirstlane(cw.z);
        const float G = __int_as_float(__builtin_amdgcn_readfirstlane(cw.w));
        // ---- phase A: pending picks against the rows, PP picks per pass (lanes beyond the list test a point at infinity)
        const float Gs = fmaxf(G * 1.000002f, 1e-30f);  // skip a (row, pick) pair iff lb > Gs: lb * (1 - 1e-6) > G, and never on underflow
        for (int s0 = 0; s0 < np; s0 += PP) {
            const int pi = s0 + (lane & (PP - 1));
            float qx = __builtin_inff(), qy = qx, qz = qx;
            if (pi < np) { const float4 q = pend[pi]; qx = q.x; qy = q.y; qz = q.z; }
            const float ex = fmaxf(fmaxf(bx0 - qx, qx - bx1), 0.f);
            const float ey = fmaxf(fmaxf(by0 - qy, qy - by1), 0.f);
            const float ez = fmaxf(fmaxf(bz0 - qz, qz - bz1), 0.f);
            const float lb = (ex * ex + ey * ey) + ez * ez;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(lb <= Gs);
            if (do_stats) st_pairs += __popcll(mask);
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                unsigned mi = (unsigned)(mask >> (i * PP)) & (unsigned)((1ull << PP) - 1ull);
                while (mi) {
                    const int p = __builtin_ctz(mi);  // lane p (row 0's group) holds pick slot p
                    mi &= mi - 1;
                    const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), p));
                    const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), p));
                    const float z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), p));
                    const float d = pn2_sqdist<MODE>(px[i] - x1, py[i] - y1, pz[i] - z1);
                    const int di = __float_as_int(d), oh = __double2hiint(mk[i]);  // d >= +0: int order == float order
                    mk[i] = __hiloint2double(di < oh ? di : oh, __double2loint(mk[i]));  // min(d, td) :151 on the high word
                }
            }
        }
        const long long st_tu = do_stats ? (long long)__builtin_readcyclecounter() : 0;
        if (do_stats) st_upd += st_tu - st_t0;
        double tr[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) tr[i] = mk[i];
#pragma unroll
        for (int w = PPT; w > 1; w = (w + 1) / 2) {
#pragma unroll
            for (int g = 0; g < w / 2; ++g) tr[g] = fps_dmax(tr[g], tr[w - 1 - g]);
        }
        const int best = __double2hiint(tr[0]);
        const unsigned bl = (unsigned)__double2loint(tr[0]);
        if (best >= tau_hi) {  // rare lanes: list every point of mine that reaches tau (one LDS atomic per lane)
            unsigned c = 0;
#pragma unroll
            for (int i = 0; i < PPT; ++i) c += __double2hiint(mk[i]) >= tau_hi ? 1u : 0u;
            unsigned slot_i;
            const unsigned caddr = (unsigned)(size_t)(smem) + 4u * (unsigned)ph;  // &ctrl[ph]
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(slot_i) : "v"(caddr), "v"(c) : "memory");
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                if (__double2hiint(mk[i]) >= tau_hi) {
                    if (slot_i < (unsigned)kLazyCap) cand[slot_i] = (unsigned long long)__double_as_longlong(mk[i]);
                    ++slot_i;
                }
            }
        }
        const long long st_t1 = do_stats ? (long long)__builtin_readcyclecounter() : 0;
        __syncthreads();
        const long long st_t2 = do_stats ? (long long)__builtin_readcyclecounter() : 0;
        const int cnt = __builtin_amdgcn_readfirstlane(ctrl[ph]);
        const bool use_list = cnt >= 1 && cnt <= kLazyCap;
        if (!use_list) {
            // empty or overflowing list (~1 phase in 8): ONE pick from the NW per-wave maxima instead (the global maximum is
            // one of them).  Every wave publishes its own maximum key; one more barrier.
            const int wh = wave_imax(best);
            const unsigned long long bal = __ballot(best == wh);
            unsigned wl;
            if (__popcll(bal) == 1) wl = (unsigned)__builtin_amdgcn_readlane((int)bl, __ffsll((long long)bal) - 1);
            else wl = wave_umax_all(best == wh ? bl : 0u);  // equal td across lanes: lowest tie key = largest low word
            if (lane == 0) wcand[wave] = ((unsigned long long)(unsigned)wh << 32) | wl;
            if (track) {
                // ties INSIDE the wave are invisible in its published maximum: class 0 = my maximum is unique in my rows,
                // 1 = shared only by points coinciding with my winner, 2 = shared by a point elsewhere
                int cls = 0;
                if (wh >= 0) {
                    const float4 wq = sxyz[fps_untiekey(~wl)];
                    int ct = 0;
                    unsigned long long oth = 0ull;
#pragma unroll
                    for (int i = 0; i < PPT; ++i) {
                        const bool t = __double2hiint(mk[i]) == wh;
                        ct += __popcll(__builtin_amdgcn_ballot_w64(t));
                        oth |= __builtin_amdgcn_ballot_w64(t && (px[i] != wq.x || py[i] != wq.y || pz[i] != wq.z));
                    }
                    cls = oth != 0ull ? 2 : (ct > 1 ? 1 : 0);
                }
                if (lane == 0) wtie[wave] = cls;
            }
            __syncthreads();
        }
        // ---- phase B: wave 0 picks from the list
        if (wave == 0) {
            if (do_stats) { st_ph++; st_empty += cnt == 0; st_over += cnt > kLazyCap; st_cnt += use_list ? cnt : 0; }
            unsigned long long key = 0x8000000000000000ull;  // negative high word: never the maximum
            if (use_list) { if (lane < cnt) key = cand[lane]; }
            else if (lane < NW) key = wcand[lane];
            int chi = (int)(unsigned)(key >> 32);
            const unsigned clo = (unsigned)key;
            int ck = fps_untiekey(~clo);
            if (chi < 0) ck = 0;
            const float4 cq = sxyz[ck];
            const int limit = use_list ? tau_hi : (int)0x80000000;
            int maxp = use_list ? kLazyCap : 1;
            if (maxp > m - jdone) maxp = m - jdone;
            int pk_k, g_first, d_last;
            float pk_x, pk_y, pk_z;
            const int lim = limit < 0 ? 0 : limit;  // valid td are >= 0; lanes without a candidate are negative
            const int npick = pn2fps::pick_phase<MODE, pn2fps::NoPost, TRACK>(chi, clo, cq.x, cq.y, cq.z, lim, maxp, pk_k, pk_x, pk_y, pk_z,
                                                                               g_first, d_last, pn2fps::NoPost(), ttl, jdone, lane);
            if (track && !use_list && npick == 1) {  // the waves holding the maximum: their own in-wave tie classes
                const int cls = (lane < NW && chi == g_first) ? wtie[lane] : 0;
                const unsigned long long c2 = __builtin_amdgcn_ballot_w64(cls == 2), c1 = __builtin_amdgcn_ballot_w64(cls == 1);
                if ((c1 | c2) != 0ull && lane == 0) pn2fps::tie_note(ttl, jdone, g_first == 0 || c2 != 0ull, g_first == 0);
            }
            if (lane < npick) {
                spick[jdone + lane] = pk_k;
                pend[lane] = make_float4(pk_x, pk_y, pk_z, 0.f);
            }
            int j = jdone + npick;
            eps = pn2fps::adapt_eps(eps, cnt);
            if (npick == 0) j = m;  // unreachable (a non-empty list always yields a pick); never spin
            if (lane == 0) {
                ctrl[ph ^ 1] = 0;
                *reinterpret_cast<int4*>(ctrl + 4) = make_int4(npick, j, __float_as_int(__int_as_float(d_last) * (1.0f - eps)), g_first);
            }
        }
        const long long st_t3 = do_stats ? (long long)__builtin_readcyclecounter() : 0;
        __syncthreads();
        if (do_stats) {
            const long long st_t4 = (long long)__builtin_readcyclecounter();
            st_a += st_t1 - st_t0; st_w1 += st_t2 - st_t1; st_b += st_t3 - st_t2; st_w2 += st_t4 - st_t3;
        }
    }
    if (do_stats && blockIdx.x == 0 && tid == 0) {
        stats[0] = st_ph; stats[1] = st_empty; stats[2] = st_over; stats[3] = st_cnt; stats[4] = st_a; stats[5] = st_w1;
        stats[6] = st_b; stats[7] = st_w2; stats[8] = (long long)__builtin_readcyclecounter() - st_t00; stats[9] = st_pairs;
    }
    if (do_stats && blockIdx.x == 0 && lane == 0) { stats[16 + wave] = st_a; stats[32 + wave] = st_pairs; stats[48 + wave] = st_upd; }
    if (track && tid == 0 && tie_out) tie_out[blockIdx.x] = pn2fps::tie_first(ttl);
    for (int jj = tid; jj < m; jj += NT) {
        const int k = spick[jj];
        out[jj] = k;
        if (nxyz) {
            const float4 p = sxyz[k];
            nxyz[jj * 3 + 0] = p.x; nxyz[jj * 3 + 1] = p.y; nxyz[jj * 3 + 2] = p.z;
        }
    }
}

template <int NT, int PPT, int MODE>
int launch_fps_lazy(int b, int n, int m, const float* inp, int ld, int* out, float* nxyz, const int* tie_in, int* tie_out, hipStream_t st) {
    const size_t bytes = fps_lazy_bytes(n, m);
    auto kern = tie_out ? fps_lazy_kernel<NT, PPT, MODE, true> : fps_lazy_kernel<NT, PPT, MODE, false>;
    static bool attr_set[2] = {false, false};  // per instantiation; benign race (idempotent call)
    if (!attr_set[tie_out != nullptr]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set[tie_out != nullptr] = true;
    }
    kern<<<b, NT, bytes, st>>>(n, m, inp, ld, out, nxyz, g_fps_stats, tie_in, tie_out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// Generic fallback for n > PN2_FPS_MAX_REG_POINTS: running min in the caller's
// `temp` rows (one row per resident block, like tf_sampling.cu:124), points
// streamed from L2.  Blocks stride over the batch (grid <= 32) so the reference's
// (32,n) scratch is sufficient.
template <int MODE>
__global__ void __launch_bounds__(1024)
fps_stream_kernel(int b, int n, int m, const float* __restrict__ xyz_all,
                  float* __restrict__ temp_all, int* __restrict__ out_all) {
    constexpr int NT = 1024;
    __shared__ uint2 slots[2 * kFpsSlotsMax];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* __restrict__ temp = temp_all + (size_t)blockIdx.x * n;
    for (int bi = blockIdx.x; bi < b; bi += gridDim.x) {
        const float* __restrict__ xyz = xyz_all + (size_t)bi * n * 3;
        int* __restrict__ out = out_all + (size_t)bi * m;
        for (int k = tid; k < n; k += NT) temp[k] = 1e38f;
        if (tid == 0) out[0] = 0;
        __syncthreads();
        int old = 0;
        for (int j = 1; j < m; ++j) {
            const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
            float best = -1.0f;
            int bestk = 0;
            for (int k = tid; k < n; k += NT) {
                const float d = pn2_sqdist<MODE>(xyz[k * 3 + 0] - x1, xyz[k * 3 + 1] - y1,
                                                 xyz[k * 3 + 2] - z1);
                const float td = temp[k];
                const float d2 = fminf(d, td);
                if (d2 != td) temp[k] = d2;
                if (d2 > best) { best = d2; bestk = k; }
            }
            const uint2 cand = fps_wave_candidate(best, bestk);
            uint2* s = slots + (j & 1) * kFpsSlotsMax;
            if (lane == 0) s[wave] = cand;
            __syncthreads();
            old = fps_merge_waves(s, NT / 64, lane);
            if (tid == 0) out[j] = old;
        }
        __syncthreads();  // temp / slots reuse by the next batch element
    }
}

template <int NT, int PPT, int MODE>
int launch_fps_reg(int b, int n, int m, const float* inp, int ld, int* out, float* nxyz, const int* tie_in, int* tie_out, hipStream_t st) {
    const size_t slots_bytes = kFpsRegHead;
    const size_t xyz_bytes = (size_t)n * sizeof(float4);
    // 160 KiB LDS per CU; keep the cloud (and the pick list) in LDS when they fit (n <= 8192 -> 128 KiB + 4m)
    const size_t pick_bytes = (size_t)m * sizeof(int);
    if (slots_bytes + xyz_bytes + pick_bytes <= 158 * 1024) {
        auto kern = tie_out ? fps_reg_kernel<NT, PPT, MODE, true, true> : fps_reg_kernel<NT, PPT, MODE, true, false>;
        static int attr_bytes[2] = {0, 0};  // per instantiation; benign race (idempotent call)
        if (attr_bytes[tie_out != nullptr] < (int)(slots_bytes + xyz_bytes + pick_bytes)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_bytes[tie_out != nullptr] = 160 * 1024;
        }
        kern<<<b, NT, slots_bytes + xyz_bytes + pick_bytes, st>>>(n, m, inp, ld, out, nxyz, tie_in, tie_out);
    } else if constexpr (NT != 64) {
        auto kern = tie_out ? fps_reg_kernel<NT, PPT, MODE, false, true> : fps_reg_kernel<NT, PPT, MODE, false, false>;
        kern<<<b, NT, slots_bytes, st>>>(n, m, inp, ld, out, nxyz, tie_in, tie_out);
    } else {
        return PN2_ERANGE;  // unreachable: a single-wave cloud (n <= 256) always fits LDS unless m is absurd
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

template <int MODE>
int dispatch_fps(int b, int n, int m, const float* inp, int ld, float* temp, int* out, float* nxyz, const int* tie_in, int* tie_out,
                 hipStream_t st) {
    // Any (threads, points per thread) layout is exact (every point carries its own tie-break key); the choice is
    // latency only.  A round's sync skeleton costs ~40 ns for one wave (no barrier, no LDS atomic), ~130 ns for 4
    // waves and ~180 ns for 16 (tools/round_ubench.hip), a lone wave issues ~0.45 instructions/ns, 4 waves per SIMD ~1/ns.
    // Measured (B=16, ns per round; profiles/r02_fps_experiments.txt): n=256 <64,4> 255 vs <256,1> 316; n=1024 <512,2> 327
    // vs <1024,1> 355; n=2048 <512,4> 360 vs <1024,2> 391; n=4096 <1024,4> 440 vs <512,8> 451; n=8192 <1024,8> 617.
    // Lazy multi-pick kernel (~15 picks per synchronised pass) where it wins: n = 4096: 174 vs 225 us, n = 8192: 342 vs 633 us
    // (B = 16; profiles/r03_fps_lazy.txt).  At n <= 2048 its serial picking wave (~400 cycles per pick) costs what a whole
    // round of the one-pick kernels below costs (93 vs 84 us at n = 1024), so those keep the small levels.
    // g_fps_variant (tuning builds): 2 = one-pick kernels everywhere, 3 = lazy kernel from n > 512.
    if (g_fps_variant != 2 && n > (g_fps_variant == 3 ? 512 : 2048) && n <= 8192 && fps_lazy_bytes(n, m) <= 160 * 1024) {
        if (n <= 1024) return launch_fps_lazy<256, 4, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 2048) return launch_fps_lazy<512, 4, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 4096) return launch_fps_lazy<1024, 4, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        return launch_fps_lazy<1024, 8, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
    }
    if (g_fps_variant == 1) {  // A/B hook: one point per thread up to 1024 threads (the round-1 layout)
        if (n <= 64) return launch_fps_reg<64, 1, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 128) return launch_fps_reg<128, 1, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 256) return launch_fps_reg<256, 1, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 512) return launch_fps_reg<512, 1, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 1024) return launch_fps_reg<1024, 1, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
    } else {
        if (n <= 64) return launch_fps_reg<64, 1, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 128) return launch_fps_reg<64, 2, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 256) return launch_fps_reg<64, 4, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 512) return launch_fps_reg<256, 2, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 1024) return launch_fps_reg<512, 2, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
        if (n <= 2048) return launch_fps_reg<512, 4, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
    }
    if (n <= 2048) return launch_fps_reg<1024, 2, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
    if (n <= 4096) return launch_fps_reg<1024, 4, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
    if (n <= 8192) return launch_fps_reg<1024, 8, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
    if (n <= 16384) return launch_fps_reg<1024, 16, MODE>(b, n, m, inp, ld, out, nxyz, tie_in, tie_out, st);
    if (ld != 3) return PN2_EUNSUP;  // the streaming kernel reads dense rows
    if (!temp) return PN2_ENULL;
    const int grid = b < 32 ? b : 32;
    if (tie_out) {  // the streaming kernel keeps no tie record: "tied at step 0" = the next level always samples for real
        hipError_t e = hipMemsetAsync(tie_out, 0, sizeof(int) * (size_t)b, st);
        if (e != hipSuccess) return (int)e;
    }
    fps_stream_kernel<MODE><<<grid, 1024, 0, st>>>(b, n, m, inp, temp, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return nxyz ? 1000000 : PN2_OK;  // sentinel: caller still has to run the separate gather
}

// ---- gather_point / grad ---------------------------------------------------
// one thread per output float: coalesced 4-byte writes, 12-byte gathered reads.
__global__ void gather_point_kernel(long long total, int n, int m,
                                    const float* __restrict__ inp,
                                    const int* __restrict__ idx, float* __restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long row = e / 3;  // (i*m + j)
        const int c = (int)(e - row * 3);
        const long long i = row / m;
        const int a = idx[row];
        out[e] = inp[(i * n + a) * 3 + c];
    }
}

__global__ void gather_point_grad_kernel(long long total, int n, int m,
                                         const float* __restrict__ out_g,
                                         const int* __restrict__ idx,
                                         float* __restrict__ inp_g) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long row = e / 3;
        const int c = (int)(e - row * 3);
        const long long i = row / m;
        const int a = idx[row];
        atomicAdd(&inp_g[(i * n + a) * 3 + c], out_g[e]);  // tf_sampling.cu:201-203
    }
}

inline int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    const long long cap = 256LL * 8;  // 256 CUs x 8 blocks, grid-stride beyond
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---- prob_sample (tf_sampling.cu:7-110): categorical sampling by inverting the running sum of the weights -------
// The drawn index depends on the exact fp32 running sum, so the kernel keeps the reference's summation ORDER:
// chunks of 8192 values; inside a chunk groups of four (v2 = a1+a0, t = a3+a2, v3 = a2+v2, v4 = t+v2; a ragged last
// group is summed left to right), a strided up-sweep / down-sweep over the group totals (total[i] += total[i-s]), the
// preceding inclusive total added to every later group, the chunk offset added last and carried with a compensated
// two-term update.  One 1024-thread workgroup per row, the chunk in LDS.  (oracle_prob_sample restates the same.)
constexpr int kPsChunk = 8192;
constexpr int kPsThreads = 1024;

__global__ void __launch_bounds__(kPsThreads)
prob_cumsum_kernel(int n, const float* __restrict__ inp_all, float* __restrict__ out_all) {
    __shared__ float b4[kPsChunk];
    __shared__ float tot[kPsChunk / 4];
    const int tid = threadIdx.x;
    const float* __restrict__ a_row = inp_all + (size_t)blockIdx.x * n;
    float* __restrict__ o_row = out_all + (size_t)blockIdx.x * n;
    float running = 0.f, running2 = 0.f;  // identical in every thread
    for (int j = 0; j < n; j += kPsChunk) {
        const int ni = n - j < kPsChunk ? n - j : kPsChunk;
        const int n4 = (ni + 3) & ~3, ng = n4 >> 2;
        const float* __restrict__ a = a_row + j;
        for (int g = tid; g < ng; g += kPsThreads) {
            const int k = 4 * g;
            if (k + 3 < ni) {
                const float v1 = a[k];
                const float v2 = a[k + 1] + v1;
                const float v3 = a[k + 2];
                const float v4 = (a[k + 3] + v3) + v2;
                b4[k] = v1; b4[k + 1] = v2; b4[k + 2] = v3 + v2; b4[k + 3] = v4;
                tot[g] = v4;
            } else {
                float v = 0.f;
                for (int k2 = k; k2 < ni; ++k2) { v += a[k2]; b4[k2] = v; }
                for (int k2 = ni; k2 < n4; ++k2) b4[k2] = v;
                tot[g] = v;
            }
        }
        __syncthreads();
        int s = 1;
        for (; 2 * s <= ng; s <<= 1) {  // up-sweep
            for (int k = tid; k < ng / (2 * s); k += kPsThreads) {
                const int i1 = 2 * s * (k + 1) - 1;
                tot[i1] += tot[i1 - s];
            }
            __syncthreads();
        }
        for (s >>= 1; s >= 1; s >>= 1) {  // down-sweep, starting at the last up-sweep stride
            for (int k = tid; k < (ng - s) / (2 * s); k += kPsThreads) {
                const int i1 = s * (2 * k + 3) - 1;
                tot[i1] += tot[i1 - s];
            }
            __syncthreads();
        }
        for (int g = 1 + tid; g < ng; g += kPsThreads) {
            const float p = tot[g - 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) b4[4 * g + e] += p;
        }
        __syncthreads();
        for (int k = tid; k < ni; k += kPsThreads) o_row[j + k] = b4[k] + running;
        const float t = tot[ng - 1] + running2;
        const float r2 = running + t;
        running2 = t - (r2 - running);
        running = r2;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
prob_search_kernel(int n, int m, int base, const float* __restrict__ cs_all, const float* __restrict__ query_all,
                   int* __restrict__ out_all) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const float* __restrict__ cs = cs_all + (size_t)blockIdx.y * n;
    const float q = query_all[(size_t)blockIdx.y * m + j] * cs[n - 1];
    int r = n - 1;
    for (int k = base; k >= 1; k >>= 1)
        if (r >= k && cs[r - k] >= q) r -= k;
    out_all[(size_t)blockIdx.y * m + j] = r;
}

}  // namespace

// probsampleLauncher(b,n,m,inp_p,inp_r,temp,out)  tf_sampling.cu:212-216, tf_sampling.cpp:72
extern "C" int pn2_prob_sample(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out,
                               void* stream) {
    if (b <= 0 || n <= 0 || m <= 0) return PN2_EINVAL;
    if (!inp_p || !inp_r || !temp || !out) return PN2_ENULL;
    if (b > 65535) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    prob_cumsum_kernel<<<b, kPsThreads, 0, st>>>(n, inp_p, temp);
    int base = 1;
    while (base < n) base <<= 1;
    prob_search_kernel<<<dim3((m + 255) / 256, b), 256, 0, st>>>(n, m, base, temp, inp_r, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

#ifdef PN2_TUNING_HOOKS
// undocumented tuning/experiment hook (not part of the ABI header)
extern "C" int pn2_debug_set_grouping(int what, int value);
extern "C" int pn2_debug_set_linear(int what, int value);
extern "C" int pn2_debug_set_bn(int what, int value);
extern "C" int pn2_debug_set_fused(int what, int value);
extern "C" int pn2_debug_set_fps_large(int what, int value);
extern "C" int pn2_debug_set_interp(int what, int value);
extern "C" int pn2_debug_set_coarse(int what, int value);
extern "C" int pn2_debug_set_fps_stats(long long* dev_ptr) { g_fps_stats = dev_ptr; return 0; }
extern "C" int pn2_debug_set(int what, int value) {
    if (what == 0) { g_fps_variant = value; return 0; }
    if (what == 5 || what == 8 || what == 9 || what == 17 || what == 18) return pn2_debug_set_linear(what, value);
    if (what == 10) return pn2_debug_set_bn(what, value);
    if (what == 11) return pn2_debug_set_fps_large(what, value);
    if (what == 12) return pn2_debug_set_interp(what, value);
    if (what == 15) return pn2_debug_set_coarse(what, value);
    if (what == 6 || what == 7 || what == 13 || what == 14 || what == 16) return pn2_debug_set_fused(what, value);
    return pn2_debug_set_grouping(what, value);
}
#endif  // PN2_TUNING_HOOKS

static int fps_entry(int b, int n, int m, const float* inp, float* temp, int* out, float* nxyz,
                     int arith_mode, void* stream, const int* tie_in = nullptr, int* tie_out = nullptr, int ld = 3) {
    if (b <= 0 || n <= 0 || m <= 0 || ld < 3) return PN2_EINVAL;
    if (!inp || !out) return PN2_ENULL;
    if ((long long)n * ld > 0x7fffffffLL) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (arith_mode) {
        case PN2_ARITH_STRICT: return dispatch_fps<PN2_ARITH_STRICT>(b, n, m, inp, ld, temp, out, nxyz, tie_in, tie_out, st);
        case PN2_ARITH_FMA: return dispatch_fps<PN2_ARITH_FMA>(b, n, m, inp, ld, temp, out, nxyz, tie_in, tie_out, st);
        case PN2_ARITH_FMA_ALT: return dispatch_fps<PN2_ARITH_FMA_ALT>(b, n, m, inp, ld, temp, out, nxyz, tie_in, tie_out, st);
        default: return PN2_EINVAL;
    }
}

extern "C" int pn2_farthest_point_sample(int b, int n, int m, const float* inp, float* temp,
                                         int* out, int arith_mode, void* stream) {
    return fps_entry(b, n, m, inp, temp, out, nullptr, arith_mode, stream);
}

extern "C" int pn2_gather_point(int b, int n, int m, const float* inp, const int* idx,
                                float* out, void* stream);

// farthest_point_sample + gather_point in one call (util/pointnet_util.py:36-37 always runs them
// back to back): out (b,m) indices and new_xyz (b,m,3) = inp[out].  For n <= 16384 the FPS kernel
// writes the coordinates itself; beyond that the streaming kernel is followed by the gather kernel.
extern "C" int pn2_fps_gather(int b, int n, int m, const float* inp, float* temp, int* out,
                              float* new_xyz, int arith_mode, void* stream) {
    if (!new_xyz) return PN2_ENULL;
    const int rc = fps_entry(b, n, m, inp, temp, out, new_xyz, arith_mode, stream);
    if (rc == 1000000) return pn2_gather_point(b, n, m, inp, out, new_xyz, stream);
    return rc;
}

// Nested farthest point sampling: pn2_farthest_point_sample / pn2_fps_gather (new_xyz may be NULL) with the tie record
// of the level above.  tie_in (b) int32 or NULL: the first tied step of the run that PRODUCED inp (inp = that run's
// new_xyz, rows in pick order); a cloud with tie_in[i] >= m gets idx = 0..m-1 and new_xyz = its first m rows without
// sampling (bit-identical to the sampled result, see fps_nested_shortcut), any other cloud is sampled as always.
// tie_out (b) int32 or NULL: this level's record for the level below.
extern "C" int pn2_fps_nested(int b, int n, int m, const float* inp, float* temp, int* out, float* new_xyz,
                              const int* tie_in, int* tie_out, int arith_mode, void* stream) {
    const int rc = fps_entry(b, n, m, inp, temp, out, new_xyz, arith_mode, stream, tie_in, tie_out);
    if (rc == 1000000) return pn2_gather_point(b, n, m, inp, out, new_xyz, stream);
    if (rc == PN2_OK && tie_out && n == 1 && m > 1) {
        // a one-point cloud repeats its point from step 1 on: the maximum is 0 there (a strict step by definition) but there is
        // no SECOND holder for the kernels' tie branches to see
        hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(tie_out), 1, (size_t)b, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return (int)e;
    }
    return rc;
}

// pn2_fps_nested on a cloud whose rows are `ld` floats apart (ld >= 3; the xyz columns of a (b,n,6) xyz+rgb batch read in
// place: model.py:26-29 slices them out of the input tensor).  n <= 16384 (the register-resident kernels); beyond: PN2_EUNSUP.
extern "C" int pn2_fps_nested_ld(int b, int n, int m, const float* inp, int ld, int* out, float* new_xyz,
                                 const int* tie_in, int* tie_out, int arith_mode, void* stream) {
    if (n > 16384) return PN2_EUNSUP;
    const int rc = fps_entry(b, n, m, inp, nullptr, out, new_xyz, arith_mode, stream, tie_in, tie_out, ld);
    if (rc == PN2_OK && tie_out && n == 1 && m > 1) {
        hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(tie_out), 1, (size_t)b, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return (int)e;
    }
    return rc;
}

extern "C" int pn2_gather_point(int b, int n, int m, const float* inp, const int* idx,
                                float* out, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0) return PN2_EINVAL;
    if (!inp || !idx || !out) return PN2_ENULL;
    const long long total = (long long)b * m * 3;
    gather_point_kernel<<<grid_for(total, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
        total, n, m, inp, idx, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

extern "C" int pn2_gather_point_grad(int b, int n, int m, const float* out_g, const int* idx,
                                     float* inp_g, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0) return PN2_EINVAL;
    if (!out_g || !idx || !inp_g) return PN2_ENULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(inp_g, 0, sizeof(float) * (size_t)b * n * 3, st);
    if (e != hipSuccess) return (int)e;
    const long long total = (long long)b * m * 3;
    gather_point_grad_kernel<<<grid_for(total, 256), 256, 0, st>>>(total, n, m, out_g, idx, inp_g);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
