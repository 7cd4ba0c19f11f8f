// pn2_fps_reg.h -- the register-resident farthest-point-sampling round loop (one workgroup per cloud) as a device routine,
// shared by pn2_sampling.hip (fps_reg_kernel) and the one-launch coarse-level geometry (pn2_coarse_geometry.hip: its tie
// fallback), plus the nested-sampling shortcut.  Reference: tf_ops/tf_sampling.cu:111-176.
#pragma once
#include "pn2_fps_common.h"

namespace pn2fpsreg {

constexpr int kFpsSlotsMax = 16;  // waves per workgroup <= 16
constexpr int kFpsRegHead = 48;   // fps_reg_kernel: bytes of LDS in front of the cloud copy (key slots + tie record)

__device__ __forceinline__ unsigned fps_tiekey(int k) {
    return (((unsigned)k & 511u) << 22) | ((unsigned)k >> 9);
}
__device__ __forceinline__ int fps_untiekey(unsigned key) {
    return (int)(((key & 0x3FFFFFu) << 9) | (key >> 22));
}

// Fused-DPP wave64 reductions on 32-bit keys (one VALU op per step; the s_nop 1
// are the 2 wait states a DPP read needs after a VALU write of the same VGPR).
// All distances are >= +0, so their bit patterns order like signed/unsigned ints;
// the sentinel -1.0f (threads without points) is a negative int.
__device__ __forceinline__ int wave_imax(int v) {
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
__device__ __forceinline__ unsigned wave_umin_all(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)v, o); v = t < v ? t : v; }
    return v;
}

// Exact 64-bit max of two (td bits : ~tiekey) pairs read as doubles.  The high word is an fp32 pattern in [0, bits(1e38f)]
// (or bits(-1.0f) for "no point"), so the doubles are finite; pairs with td < 2^-126-ish map to fp64 DENORMALS (high word
// < 0x00100000), which v_max_f64 orders correctly only because the kernel runs with fp64 denormals enabled -- the
// default float mode of HIP kernels on gfx9 (FP64/FP16 denormals on, MODE.FP_DENORM = 0b11xx; only fp32 denormals are
// affected by -fgpu-flush-denormals-to-zero).  tests/test_ops_gpu.py::test_fps_duplicates_and_degenerate and the
// lattice tests (td == 0 everywhere) would fail if that ever changed.  (Inline asm: no canonicalisation inserted.)
__device__ __forceinline__ double fps_dmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Nested sampling (pn2_fps_nested).  When `inp` is itself the (b,n,3) output of an FPS + gather of a larger cloud, its
// rows are that run's picks in pick order, and FPS restricted to those rows retraces them: at step j the row j holds
// the maximum td over the WHOLE parent cloud, hence over the subset, with bit-identical td values (same operands, same
// expression) -- the pick is j itself unless another row TIES with it (then this level's tie-break by (j mod 512, j)
// may differ from the parent's).  The parent run reports per cloud the first step at which its maximum was not unique
// (pn2fps::TieTrack); a level asked for m <= that step is the identity: idx = 0..m-1, new_xyz = the first m rows.  The
// answer is exact, not approximate: tests/test_ref_gpu.py holds it against the reference's kernel on tie-free and
// tie-heavy clouds.  Returns true when the workgroup took the shortcut (uniform per workgroup).
// (tie_in_b / tie_out_b: THIS cloud's slots of the tie records, or null)
__device__ __forceinline__ bool fps_nested_shortcut_b(int n, int m, const float* __restrict__ xyz, int* __restrict__ out,
                                                      float* __restrict__ nxyz, const int* __restrict__ tie_in_b,
                                                      int* __restrict__ tie_out_b, int nthreads, int ld = 3) {
    if (tie_in_b == nullptr) return false;
    const int T = __builtin_amdgcn_readfirstlane(*tie_in_b);
    // m == n is refused as well: the sampler does not look at ties of its LAST pick (a consumer normally asks for fewer
    // picks than the level above made), and a level that takes every row needs that step too.
    if (T < m || m >= n) return false;
    for (int jj = threadIdx.x; jj < m; jj += nthreads) out[jj] = jj;
    if (nxyz) for (int e = threadIdx.x; e < m * 3; e += nthreads) nxyz[e] = xyz[(e / 3) * ld + e % 3];
    if (tie_out_b && threadIdx.x == 0) *tie_out_b = T;  // the prefix of a prefix: the same bound holds below
    return true;
}
__device__ __forceinline__ bool fps_nested_shortcut(int n, int m, const float* __restrict__ xyz, int* __restrict__ out,
                                                    float* __restrict__ nxyz, const int* __restrict__ tie_in,
                                                    int* __restrict__ tie_out, int nthreads, int ld = 3) {
    return fps_nested_shortcut_b(n, m, xyz, out, nxyz, tie_in ? tie_in + blockIdx.x : nullptr,
                                 tie_out ? tie_out + blockIdx.x : nullptr, nthreads, ld);
}

// NT threads, thread t owns points k = t + NT*i (i < PPT) in VGPRs for the whole kernel: coordinates and ONE 64-bit
// register pair per point,
//          (td bits : ~tiekey(k))        td = running min distance (>= +0: int order == float order),
// read as a double.  For these bit patterns v_max_f64 is an exact 64-bit max, i.e. exactly the reference's order
// (max td, then lowest k mod 512, then lowest k; tf_sampling.cu:153-170): the tie-break key rides through every
// max, there is no "which of my points was it" search in the round and no constraint on the thread layout.
// A round:
//   1. distance update (fp32 sub/mul/fma: 2-cycle VALU pipe) + v_min_i32 on the high words, v_max_f64 tree;
//   2. wave max of the high word by fused DPP; the lanes holding it (normally one) publish their pair with ONE
//      LDS atomic max (ds_max_u64) -- the LDS unit merges lanes and waves;
//   3. one barrier, one broadcast read of the winning pair, one broadcast read of the winner's xyz.
// Three key slots rotate so the reset of a slot never races with its readers.  A single-wave block (NT == 64)
// needs neither LDS atomics nor barriers: ballot + v_readlane.
// TRACK: also write the run's tie record (pn2fps::tie_*) for the level below.
// fps_reg_body: ONE cloud (xyz (n,3), out (m), nxyz (m,3) or null, tie_out_b = the cloud's slot of the tie record or null)
// by the calling workgroup of NT threads; smem = kFpsRegHead + (LDS_XYZ ? 16 n + 4 m : 0) bytes, 16-byte aligned.
// ld = row stride of xyz in floats (3 = dense; 6 = the xyz columns of a (n,6) xyz+rgb cloud read in place).
template <int NT, int PPT, int MODE, bool LDS_XYZ, bool TRACK>
__device__ __forceinline__ void fps_reg_body(int n, int m, const float* __restrict__ xyz, int* __restrict__ out,
                                             float* __restrict__ nxyz, int* __restrict__ tie_out_b, unsigned char* smem,
                                             int ld = 3) {
    static_assert(NT != 64 || LDS_XYZ, "single-wave path keeps the cloud in LDS");
    // layout: 4 x u64 key slots (3 used), 4 x i32 tie record (3 used) | float4 xyz[n] (if LDS_XYZ) | int picks[m] (if LDS_XYZ)
    // The picks are kept in LDS and written to HBM once, coalesced, after the last round (together
    // with their coordinates when the fused gather is requested): no global store sits on the
    // round-to-round critical path of wave 0.
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem);
    int* ttl = reinterpret_cast<int*>(smem + 4 * sizeof(unsigned long long));  // pn2fps::tie_* (TRACK only)
    float4* sxyz = reinterpret_cast<float4*>(smem + kFpsRegHead);
    int* spick = reinterpret_cast<int*>(sxyz + (LDS_XYZ ? n : 0));

    const int tid = threadIdx.x;
    // nxyz: optional fused gather_point (tf_sampling.cu:178-191): the coordinates of every pick pass through
    // this routine anyway, so new_xyz[j] = xyz[out[j]] costs three extra stores per round
    // TRACK: a second holder of the previous round's maximum is looked for at the top of the next round, when the winner's
    // coordinates are at hand anyway
    int pw_hi = -2;               // the previous winner's td bits (-2: none yet, never a point's high word)
    unsigned pw_lo = 0u;

    float px[PPT], py[PPT], pz[PPT];
    double mk[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + NT * i;
        int hi;
        if (k < n) {
            px[i] = xyz[k * ld + 0];
            py[i] = xyz[k * ld + 1];
            pz[i] = xyz[k * ld + 2];
            hi = __float_as_int(1e38f);  // tf_sampling.cu:124-126
            if constexpr (LDS_XYZ) sxyz[k] = make_float4(px[i], py[i], pz[i], 0.f);
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            hi = __float_as_int(-1.0f);  // never selected: a negative double, below every real pair
        }
        mk[i] = __hiloint2double(hi, (int)~fps_tiekey(k));
    }
    if (tid < 4) slots[tid] = 0ull;
    if (TRACK && tid == 0) pn2fps::tie_init(ttl);
    if (tid == 0) {  // first pick is index 0 (tf_sampling.cu:122-123)
        if constexpr (LDS_XYZ) spick[0] = 0; else out[0] = 0;
    }
    __syncthreads();

    int old = 0;
    int slot = 1;  // j % 3
    for (int j = 1; j < m; ++j) {
        float x1, y1, z1;
        if constexpr (LDS_XYZ) {
            const float4 p = sxyz[old];
            x1 = p.x; y1 = p.y; z1 = p.z;
        } else {
            x1 = xyz[old * ld + 0]; y1 = xyz[old * ld + 1]; z1 = xyz[old * ld + 2];
            if (nxyz && tid == 0) { nxyz[(j - 1) * 3 + 0] = x1; nxyz[(j - 1) * 3 + 1] = y1; nxyz[(j - 1) * 3 + 2] = z1; }
        }
        if constexpr (TRACK) {
            // mk still holds the td the previous maximum was taken over: exactly ONE point of the cloud -- the winner, owned
            // by lane (old % NT) -- may carry its value.  Branch-free count per wave: `dup` collects the lanes that hold it
            // in MORE than one of their rows (any count >= 2: an OR/XOR parity would miss three), the population of the OR
            // counts the lanes.
            unsigned long long m_or = 0ull, dup = 0ull;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const unsigned long long hb = __builtin_amdgcn_ballot_w64(__double2hiint(mk[i]) == pw_hi);
                dup |= m_or & hb; m_or |= hb;
            }
            const int expect = (((old & (NT - 1)) >> 6) == (tid >> 6)) ? 1 : 0;
            if (__builtin_expect(dup != 0ull || __popcll(m_or) != expect, 0)) {
#pragma unroll
                for (int i = 0; i < PPT; ++i) {
                    if (__double2hiint(mk[i]) == pw_hi && (unsigned)__double2loint(mk[i]) != pw_lo)
                        pn2fps::tie_note(ttl, j - 1, pw_hi == 0 || px[i] != x1 || py[i] != y1 || pz[i] != z1, pw_hi == 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float d = pn2_sqdist<MODE>(px[i] - x1, py[i] - y1, pz[i] - z1);
            const int di = __float_as_int(d), oh = __double2hiint(mk[i]);  // d >= +0: int order == float order
            mk[i] = __hiloint2double(di < oh ? di : oh, __double2loint(mk[i]));  // min(d, td) :151 on the high word
        }
        double tr[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) tr[i] = mk[i];
#pragma unroll
        for (int w = PPT; w > 1; w = (w + 1) / 2) {
#pragma unroll
            for (int g = 0; g < w / 2; ++g) tr[g] = fps_dmax(tr[g], tr[w - 1 - g]);
        }
        const int best = __double2hiint(tr[0]);
        const int wmax = wave_imax(best);
        if constexpr (NT == 64) {
            // one wave: the winner is resolved with a ballot (equal td across lanes: lowest tie key = largest low
            // word); no LDS atomic, no barrier in the round
            const unsigned long long bal = __ballot(best == wmax);
            const unsigned lo = (unsigned)__double2loint(tr[0]);
            unsigned wl;
            if (__popcll(bal) == 1) wl = (unsigned)__builtin_amdgcn_readlane((int)lo, __ffsll((long long)bal) - 1);
            else wl = ~wave_umin_all(best == wmax ? ~lo : 0xFFFFFFFFu);
            old = fps_untiekey(~wl);
            if (tid == 0) spick[j] = old;
            if constexpr (TRACK) { pw_hi = wmax; pw_lo = wl; }
            continue;
        }
        if (best == wmax && wmax >= 0) {  // normally a single lane of the wave
            // one ds_max_u64 per winning lane (normally exactly one per wave); written as asm so the
            // compiler's uniform-address atomic optimiser does not wrap it in a per-lane scalar loop
            const unsigned long long comp = (unsigned long long)__double_as_longlong(tr[0]);
            const unsigned saddr = (unsigned)(size_t)(&slots[slot]);  // LDS byte address (low 32 bits of the generic pointer)
            asm volatile("ds_max_u64 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(saddr), "v"(comp) : "memory");
        }
        __syncthreads();
        const unsigned long long win = slots[slot];
        old = fps_untiekey(~(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)win));
        if constexpr (TRACK) { pw_hi = (int)(unsigned)(win >> 32); pw_lo = (unsigned)win; }
        const int nxt = slot == 2 ? 0 : slot + 1;          // (j+1) % 3
        if (tid == 0) {
            slots[nxt == 2 ? 0 : nxt + 1] = 0ull;          // (j+2) % 3: last read after barrier j-1, next used in round j+2
            if constexpr (LDS_XYZ) spick[j] = old; else out[j] = old;
        }
        slot = nxt;
    }
    if constexpr (LDS_XYZ) {
        __syncthreads();
        for (int jj = tid; jj < m; jj += NT) {
            const int k = spick[jj];
            out[jj] = k;
            if (nxyz) {
                const float4 p = sxyz[k];
                nxyz[jj * 3 + 0] = p.x; nxyz[jj * 3 + 1] = p.y; nxyz[jj * 3 + 2] = p.z;
            }
        }
    } else if (nxyz && tid == 0) {  // coordinates of the last pick
        nxyz[(m - 1) * 3 + 0] = xyz[old * ld + 0]; nxyz[(m - 1) * 3 + 1] = xyz[old * ld + 1]; nxyz[(m - 1) * 3 + 2] = xyz[old * ld + 2];
    }
    if constexpr (TRACK) {  // (ties of the LAST pick are not looked at: a consumer asks for fewer picks than this level made)
        __syncthreads();
        if (tid == 0 && tie_out_b) *tie_out_b = pn2fps::tie_first(ttl);
    }
}

}  // namespace pn2fpsreg
