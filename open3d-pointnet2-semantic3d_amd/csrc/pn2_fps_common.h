// pn2_fps_common.h -- pieces shared by the farthest-point-sampling kernels (pn2_sampling.hip, pn2_fps_bucket.hip).
#pragma once
#include "pn2_common.h"

namespace pn2fps {

// Tie-break of the reference (tf_sampling.cu:153-170: 512-thread strided scan with a strict '>', left-biased tree):
// among equal distances the lowest (k mod 512, k) wins.  Every point carries ~tiekey(k) in the low word of its 64-bit
// key (td bits : ~tiekey), so a plain 64-bit max reproduces the reference's order.
__device__ __forceinline__ unsigned tiekey(int k) { return (((unsigned)k & 511u) << 22) | ((unsigned)k >> 9); }
__device__ __forceinline__ int untiekey(unsigned key) { return (int)(((key & 0x3FFFFFu) << 9) | (key >> 22)); }

// wave64 max of a 32-bit signed key by fused DPP (one VALU per step; s_nop 1 = the 2 wait states a DPP read needs after
// a VALU write of the same VGPR), reading the source register in place; uniform result.
__device__ __forceinline__ int wave_imax_from(int src) {
    int v;
    asm volatile(
        "s_nop 1\n"
        "v_max_i32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        : "=&v"(v) : "v"(src));
    return __builtin_amdgcn_readlane(v, 63);
}
// wave64 max of an unsigned key by fused DPP (uniform result); the tie paths use it, so clouds with many duplicated rows
// (every pick of a row that has a twin is a tie) pay 6 VALU instead of 6 cross-lane shuffles per tied pick
__device__ __forceinline__ unsigned wave_umax_dpp(unsigned src) {
    unsigned v;
    asm volatile(
        "s_nop 1\n"
        "v_max_u32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
        "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
        "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        : "=&v"(v) : "v"(src));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_umax_all(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)v, o); v = t > v ? t : v; }
    return v;
}

constexpr int kLazyCap = 64;  // candidates per phase = lanes of the picking wave

// PHASE B of the lazy multi-pick scheme (see pn2_sampling.hip, fps_lazy_kernel): ONE wave, lane = candidate
// (chi = td bits, negative = no candidate; clo = ~tiekey; cx, cy, cz = coordinates).  Repeatedly: wave max of td (6 fused
// DPP steps), ballot of the lanes holding it (one, unless td ties: then the largest low word = lowest tie key decides),
// the winner's coordinates by v_readlane, every candidate's td lowered by its distance to it -- until the best candidate
// drops below `lim` or `maxp` picks.  The picks are parked in lane `npick` of four registers (v_writelane) for the
// caller to store: no memory traffic and no exec juggling on the chain.  Returns the number of picks; g_first = td of the
// first pick (>= every td in the cloud when it was picked), d_last = td of the last one.
struct NoPost { __device__ __forceinline__ void operator()(int, int, float, float, float, int) const {} };

// Tie bookkeeping of the nested-sampling shortcut (pn2_fps_nested, pn2_sampling.hip).  A pick is TIED when more than one
// point of the cloud holds the maximum td at that step.  `strict` = first step with a tie between points of different
// coordinates, or with td == 0 (every remaining point is a duplicate of a picked one); `benign` = first step whose tied
// points all coincide with the winner (their td drops to 0 with the pick: they are picked only once td == 0 everywhere).
// Steps are pick numbers (pick 0 = index 0 is never a choice); kNoTie = no such step.
constexpr int kNoTie = 0x7fffffff;
// The record lives in three LDS words {strict, benign, zero} written from the rare tie branches only (a register copy
// would ride through the pick loop as loop-carried values: +10 scalar moves on a ~50 instruction dependent chain).
__device__ __forceinline__ void tie_init(int* t) { t[0] = kNoTie; t[1] = kNoTie; t[2] = 0; }
__device__ __forceinline__ void tie_note(int* t, int step, bool is_strict, bool is_zero) {  // ONE lane calls this
    atomicMin(&t[is_strict ? 0 : 1], step);
    if (is_zero) t[2] = 1;  // the maximum td reached 0 (always a strict step): from there on coincident twins are picked too
}
// first step at which FPS restricted to this run's picks could leave the identity: the first strict tie -- or the first
// benign one when the run went on until td == 0, because only then the coincident twin is among the picks
__device__ __forceinline__ int tie_first(const int* t) { return t[2] != 0 && t[1] < t[0] ? t[1] : t[0]; }

// on_pick(npick, pk_k, pk_x, pk_y, pk_z, g_first) is called after every pick with the parked registers (the streaming
// kernel posts the picks to the worker waves in batches from it).
// TRACK: record tied picks in the LDS words tt[3] (steps counted from jbase; lane = lane id); all of it lives in the rare tie branch.
template <int MODE, class OnPick = NoPost, bool TRACK = false>
__device__ __forceinline__ int pick_phase(int chi, unsigned clo, float cx, float cy, float cz, int lim, int maxp,
                                          int& pk_k, float& pk_x, float& pk_y, float& pk_z, int& g_first, int& d_last,
                                          OnPick on_pick = OnPick(), int* tt = nullptr, int jbase = 0, int lane = 0) {
    int npick = 0;
    g_first = -1; d_last = 0;
    pk_k = 0; pk_x = pk_y = pk_z = 0.f;
    const int ck = untiekey(~clo);
    while (npick < maxp) {
        const int bh = wave_imax_from(chi);
        if (bh < lim) break;
        unsigned long long bal = __builtin_amdgcn_ballot_w64(chi == bh);
        if (__builtin_expect(__popcll(bal) != 1, 0)) {
            if constexpr (TRACK) {
                // do the holders of the maximum all coincide?  (compared with the FIRST holder, not the winner: the hot path
                // below must not share a value with this branch, or the compiler threads the branch through it)
                const int Lr = __builtin_ctzll(bal);
                const float rx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), Lr));
                const float ry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), Lr));
                const float rz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), Lr));
                const unsigned long long other = __builtin_amdgcn_ballot_w64(chi == bh && (cx != rx || cy != ry || cz != rz));
                if (lane == 0) tie_note(tt, jbase + npick, bh == 0 || other != 0ull, bh == 0);
            }
            const unsigned lm = wave_umax_dpp(chi == bh ? clo : 0u);
            bal = __builtin_amdgcn_ballot_w64(chi == bh && clo == lm);
        }
        const int L = __builtin_ctzll(bal);
        const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), L));
        const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), L));
        const float z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), L));
        const int kk = __builtin_amdgcn_readlane(ck, L);
        // (gfx9 constant bus: one SGPR per instruction, so the lane select travels in m0, which the compiler only ever
        // sets right before a use of its own)
        // s_nop: the parked values come out of v_readlane (a VALU write of an SGPR); gfx940+ wants 2 wait states before a VALU
        // reads such an SGPR, and the hazard recogniser does not insert them in front of inline asm that CONSUMES a register
        asm volatile("s_mov_b32 m0, %8\n s_nop 0\n v_writelane_b32 %0, %4, m0\n v_writelane_b32 %1, %5, m0\n v_writelane_b32 %2, %6, m0\n v_writelane_b32 %3, %7, m0"
                     : "+v"(pk_k), "+v"(pk_x), "+v"(pk_y), "+v"(pk_z) : "s"(kk), "s"(x1), "s"(y1), "s"(z1), "s"(npick));
        const float d = pn2_sqdist<MODE>(cx - x1, cy - y1, cz - z1);
        const int di = __float_as_int(d);
        chi = di < chi ? di : chi;              // lanes without a candidate stay negative
        g_first = bh > g_first ? bh : g_first;  // td of the picks never increases: the maximum is the first one
        d_last = bh;
        ++npick;
        on_pick(npick, pk_k, pk_x, pk_y, pk_z, g_first);
    }
    return npick;
}

// eps follows the list length (target 12..42 of 64 entries): the result never depends on it, only the number of phases
__device__ __forceinline__ float adapt_eps(float eps, int cnt) {
    if (cnt == 0) return fminf(0.5f, eps * 2.0f);
    if (cnt > kLazyCap) return eps * 0.5f;
    if (cnt < 12) return fminf(0.5f, eps * 1.3f);
    if (cnt > 42) return eps * 0.8f;
    return eps;
}

}  // namespace pn2fps
