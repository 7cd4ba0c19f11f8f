// pn2_fps_bucket.hip -- exact farthest point sampling for clouds that do not fit one workgroup's registers
// (n > 16384; BASELINE configs[4]: N = 65536 -> npoint 4096).  The reference kernel (tf_sampling.cu:111-176) and
// the streaming fallback in pn2_sampling.hip re-read all n running minima every round: one workgroup, 1 MB per
// round, 11 us per round.  Here the cloud is first sorted along a space-filling (Hilbert) curve and cut into buckets of 64
// consecutive points (one wave each).  A round only touches the buckets the new pick can change:
//     bucket b is skipped  iff  lb(pick, bbox_b) * (1 - 1e-6) > max_{p in b} td[p]
// where lb is the squared distance from the pick to the bucket's bounding box: every point of a skipped bucket
// has fp32 distance >= lb*(1-1e-6) > its td, so min(td, d) == td -- bit-identical to updating it.  Each bucket
// caches its best (td, tie-break) key and the coordinates of that point, so the global argmax is a reduction
// over <= 2048 cached keys in LDS.  Same picks as the reference: first pick 0, winner = max td, ties -> lowest
// (k mod 512, k) on the ORIGINAL indices.  One 1024-thread workgroup per cloud, ~2 us per round.
#include <hipcub/hipcub.hpp>

#include "pn2_fps_common.h"

namespace {

PN2_TUNABLE(long long*, g_fb_stats, nullptr)  // tuning builds: 16 counters written by block 0
PN2_TUNABLE(int, g_fb_variant, 0)  // tuning hook (pn2_debug_set(11, v)): 1 = one-pick-per-round kernel
constexpr int kFbThreads = 1024;
constexpr int kFbWaves = kFbThreads / 64;
constexpr int kFbMaxBuckets = 2048;  // n <= 131072
constexpr int kFbBPT = kFbMaxBuckets / kFbThreads;  // buckets per thread (2)
constexpr size_t kFbLds = (size_t)kFbMaxBuckets * (16 + 8 + 24 + 2);

struct FbLayout {  // byte offsets into the caller's workspace (256-byte aligned)
    size_t bbox, keys_in, keys_out, vals_in, vals_out, sorted, td, seg, cub, total, cub_bytes;
    int npad;
};

inline size_t fb_align(size_t x) { return (x + 255) & ~(size_t)255; }

FbLayout fb_layout(int b, int n) {
    FbLayout L{};
    L.npad = (n + 63) & ~63;
    size_t o = 0;
    const size_t bn = (size_t)b * n, bp = (size_t)b * L.npad;
    L.bbox = o; o = fb_align(o + (size_t)b * 6 * 4);
    L.keys_in = o; o = fb_align(o + bn * 4);
    L.keys_out = o; o = fb_align(o + bn * 4);
    L.vals_in = o; o = fb_align(o + bn * 4);
    L.vals_out = o; o = fb_align(o + bn * 4);
    L.sorted = o; o = fb_align(o + bp * 16);
    L.td = o; o = fb_align(o + bp * 4);
    L.seg = o; o = fb_align(o + (size_t)(b + 1) * 4);
    size_t cub = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cub, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                             (int*)nullptr, b <= 32 ? b * n : n, 0, 32);
    L.cub_bytes = cub;
    L.cub = o; o = fb_align(o + cub);
    L.total = o;
    return L;
}

__device__ __forceinline__ unsigned fb_ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fb_unord(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ unsigned fb_tiekey(int k) { return (((unsigned)k & 511u) << 22) | ((unsigned)k >> 9); }

__global__ void fb_init_kernel(int b, unsigned* bbox) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b * 6) bbox[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
}

__global__ void __launch_bounds__(256)
fb_bbox_kernel(int n, const float* __restrict__ xyz_all, unsigned* __restrict__ bbox_all) {
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.y * n * 3;
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const unsigned o = fb_ord(xyz[(size_t)i * 3 + a]);
            mn[a] = o < mn[a] ? o : mn[a];
            mx[a] = o > mx[a] ? o : mx[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const unsigned wmn = ~pn2_wave_umax(~mn[a]);
        const unsigned wmx = pn2_wave_umax(mx[a]);
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&bbox_all[blockIdx.y * 6 + a], wmn);
            atomicMax(&bbox_all[blockIdx.y * 6 + 3 + a], wmx);
        }
    }
}

__device__ __forceinline__ unsigned fb_spread10(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// sort key = 30-bit index of the point's cell along a 3-D HILBERT curve over a 1024^3 grid of CUBIC cells spanning the
// cloud's bounding box (Skilling's axes-to-transpose transform; ordering only: it decides which points share a bucket,
// never a result).  Hilbert rather than Morton: a Z-order run of 64 points can straddle a jump between distant octants;
// such buckets have huge boxes and are visited by most picks (15 vs 9 buckets per pick late in a 65536-point run).
__global__ void __launch_bounds__(256)
fb_keys_kernel(int n, const float* __restrict__ xyz_all, const unsigned* __restrict__ bbox_all,
               unsigned* __restrict__ keys, int* __restrict__ vals, int batch_key) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int bi = blockIdx.y;
    float lo[3], ext = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = fb_unord(bbox_all[bi * 6 + a]);
        ext = fmaxf(ext, fb_unord(bbox_all[bi * 6 + 3 + a]) - lo[a]);
    }
    unsigned X[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float t = ext > 0.f ? (xyz_all[((size_t)bi * n + i) * 3 + a] - lo[a]) / ext : 0.f;
        int q = (int)(t * 1023.0f);
        if (!(t == t)) q = 0;
        X[a] = (unsigned)(q < 0 ? 0 : (q > 1023 ? 1023 : q));
    }
    for (unsigned Q = 512; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (X[a] & Q) X[0] ^= P;
            else { const unsigned t = (X[0] ^ X[a]) & P; X[0] ^= t; X[a] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    unsigned t = 0;
    for (unsigned Q = 512; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    const unsigned code = (fb_spread10(X[0]) << 2) | (fb_spread10(X[1]) << 1) | fb_spread10(X[2]);
    keys[(size_t)bi * n + i] = batch_key ? (((unsigned)bi << 27) | (code >> 3)) : code;
    vals[(size_t)bi * n + i] = i;
}

// sorted[bi][j] = (x, y, z, original index) in curve order; padded to a multiple of 64 with copies of the last
// point (same index: a duplicate candidate can never change the winner); td = 1e38 (tf_sampling.cu:124-126)
__global__ void __launch_bounds__(256)
fb_gather_kernel(int n, int npad, const float* __restrict__ xyz_all, const int* __restrict__ vals,
                 float4* __restrict__ sorted, float* __restrict__ td) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npad) return;
    const int bi = blockIdx.y;
    const int src = vals[(size_t)bi * n + (j < n ? j : n - 1)];
    const float* __restrict__ p = xyz_all + ((size_t)bi * n + src) * 3;
    sorted[(size_t)bi * npad + j] = make_float4(p[0], p[1], p[2], __int_as_float(src));
    td[(size_t)bi * npad + j] = 1e38f;
}

// 64-bit max over the wave (uniform result)
__device__ __forceinline__ unsigned long long fb_wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
        const unsigned long long t = ((unsigned long long)hi << 32) | lo;
        v = t > v ? t : v;
    }
    return v;
}

template <int MODE>
__global__ void __launch_bounds__(kFbThreads)
fps_bucket_kernel(int n, int npad, int m, const float* __restrict__ xyz_all, const float4* __restrict__ sorted_all,
                  float* __restrict__ td_all, int* __restrict__ out_all, float* __restrict__ new_xyz_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
    float4* bwin = reinterpret_cast<float4*>(fb_smem);                                        // coordinates (+ original index) of every bucket's best point
    unsigned long long* bkey = reinterpret_cast<unsigned long long*>(bwin + kFbMaxBuckets);   // best key of every bucket
    float (*bb)[kFbMaxBuckets] = reinterpret_cast<float (*)[kFbMaxBuckets]>(bkey + kFbMaxBuckets);  // bounding boxes [6][.]
    unsigned short* alist = reinterpret_cast<unsigned short*>(&bb[6][0]);                     // buckets the current pick can change
    __shared__ unsigned long long slots[4];
    __shared__ float4 spick;
    __shared__ int acount;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bi = blockIdx.x;
    const float* __restrict__ xyz = xyz_all + (size_t)bi * n * 3;
    const float4* __restrict__ sorted = sorted_all + (size_t)bi * npad;
    float* __restrict__ td = td_all + (size_t)bi * npad;
    int* __restrict__ out = out_all + (size_t)bi * m;
    float* __restrict__ nxyz = new_xyz_all ? new_xyz_all + (size_t)bi * m * 3 : nullptr;
    const int nb = npad >> 6;

    // bucket bounding boxes: wave w handles buckets w, w+16, ...
    for (int bk = wave; bk < nb; bk += kFbWaves) {
        const float4 p = sorted[bk * 64 + lane];
        float lo[3] = {p.x, p.y, p.z}, hi[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
                hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
            }
        if (lane == 0) {
            bb[0][bk] = lo[0]; bb[1][bk] = lo[1]; bb[2][bk] = lo[2];
            bb[3][bk] = hi[0]; bb[4][bk] = hi[1]; bb[5][bk] = hi[2];
        }
    }
    for (int bk = tid; bk < kFbMaxBuckets; bk += kFbThreads) bkey[bk] = 0ull;
    if (tid < 4) slots[tid] = 0ull;
    if (tid == 0) {
        acount = 0;
        spick = make_float4(xyz[0], xyz[1], xyz[2], 0.f);  // first pick is index 0 (tf_sampling.cu:122-123)
        out[0] = 0;
    }
    __syncthreads();
    float blo[kFbBPT][3], bhi[kFbBPT][3];
    float bmax[kFbBPT];  // max td of the bucket (float value of its key)
    unsigned long long mykey[kFbBPT];
#pragma unroll
    for (int u = 0; u < kFbBPT; ++u) {
        const int bk = tid + kFbThreads * u;
        const int bc = bk < nb ? bk : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { blo[u][a] = bb[a][bc]; bhi[u][a] = bb[3 + a][bc]; }
        bmax[u] = bk < nb ? 1e38f : -1.0f;  // everything is "changed" by the first pick
        mykey[u] = 0ull;
    }
    int slot = 1;
    for (int j = 1; j < m; ++j) {
        const float4 pk = spick;
        const float x1 = pk.x, y1 = pk.y, z1 = pk.z;
        if (nxyz && tid == 0) { nxyz[(j - 1) * 3 + 0] = x1; nxyz[(j - 1) * 3 + 1] = y1; nxyz[(j - 1) * 3 + 2] = z1; }
        // (a) which buckets can the pick change?
#pragma unroll
        for (int u = 0; u < kFbBPT; ++u) {
            const float ex = fmaxf(fmaxf(blo[u][0] - x1, x1 - bhi[u][0]), 0.f);
            const float ey = fmaxf(fmaxf(blo[u][1] - y1, y1 - bhi[u][1]), 0.f);
            const float ez = fmaxf(fmaxf(blo[u][2] - z1, z1 - bhi[u][2]), 0.f);
            const float lb = (ex * ex + ey * ey + ez * ez) * 0.999999f;
            if (lb <= bmax[u]) alist[atomicAdd(&acount, 1)] = (unsigned short)(tid + kFbThreads * u);
        }
        __syncthreads();
        // (b) update them, one wave per bucket
        const int na = acount;
        for (int i = wave; i < na; i += kFbWaves) {
            const int bk = alist[i];
            const float4 p = sorted[bk * 64 + lane];
            const float told = td[bk * 64 + lane];
            const float d = pn2_sqdist<MODE>(p.x - x1, p.y - y1, p.z - z1);
            const float tn = fminf(d, told);  // min(d, td) tf_sampling.cu:151 (both >= +0, never NaN by contract)
            td[bk * 64 + lane] = tn;
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(tn) << 32) | (unsigned)(~fb_tiekey(__float_as_int(p.w)));
            const unsigned long long best = fb_wave_max_u64(key);
            if (key == best) {  // one lane (padding copies share a key: identical writes)
                bkey[bk] = best;
                bwin[bk] = p;
            }
        }
        __syncthreads();
        // (c) argmax over the cached bucket keys
        if (tid == 0) acount = 0;
        unsigned long long tbest = 0ull;
#pragma unroll
        for (int u = 0; u < kFbBPT; ++u) {
            const int bk = tid + kFbThreads * u;
            if (bk < nb) {
                mykey[u] = bkey[bk];
                bmax[u] = __uint_as_float((unsigned)(mykey[u] >> 32));
                tbest = mykey[u] > tbest ? mykey[u] : tbest;
            }
        }
        const unsigned long long wbest = fb_wave_max_u64(tbest);
        if (lane == 0) atomicMax(&slots[slot], wbest);
        __syncthreads();
        const unsigned long long win = slots[slot];
#pragma unroll
        for (int u = 0; u < kFbBPT; ++u) {
            const int bk = tid + kFbThreads * u;
            if (bk < nb && mykey[u] == win) {  // exactly one thread: keys are unique per point (pads share a bucket)
                const float4 w = bwin[bk];
                spick = w;
                out[j] = __float_as_int(w.w);
            }
        }
        if (tid == 0) {
            const int nxt = slot == 2 ? 0 : slot + 1;
            slots[nxt == 2 ? 0 : nxt + 1] = 0ull;  // (j+2) % 3
        }
        slot = slot == 2 ? 0 : slot + 1;
        __syncthreads();
    }
    if (nxyz && tid == 0) {
        const float4 pk = spick;
        nxyz[(m - 1) * 3 + 0] = pk.x; nxyz[(m - 1) * 3 + 1] = pk.y; nxyz[(m - 1) * 3 + 2] = pk.z;
    }
}


// ---- lazy multi-pick variant (default) --------------------------------------------------------------------------------
// Same scheme as fps_lazy_kernel (pn2_sampling.hip) for clouds that live in L2 / HBM: a PHASE applies the pending picks
// of the previous phase to the buckets they can change, lists every point with td >= tau in LDS (64-bit key +
// coordinates), and ONE wave then takes ~17 exact picks from the list without touching the cloud (pn2fps::pick_phase)
// -- instead of three barriers and an L2 round trip per pick (2.5 us each at n = 65536).
//   A1  thread = bucket: which pending picks can lower a td of my bucket?  lb(pick, box)^2 <= bmax (1 + 2e-6), bmax = the
//       bucket's exact maximum after its last visit (>= every stale td in it).  Buckets with a non-empty pick mask, or
//       with bmax >= tau (they hold candidates), go to the work list (LDS).
//   A2  wave = work-list entry (4 entries' loads in flight): load 64 points + td, apply the masked picks (coordinates by
//       v_readlane from lanes holding the pending list), store td where it changed, new bucket maximum (DPP), append the
//       lanes with td >= tau to the candidate list.
//   B   wave 0: picks while the best candidate stays >= tau; they are written straight to `out` / `new_xyz`.
// Empty or overflowing list (~1 phase in 8; every phase of an all-duplicates cloud): ONE exact pick from a full
// reduction -- global maximum of the bucket maxima, the buckets that reach it are scanned for the 64-bit maximum key
// (ds_max_u64), its coordinates come from the original cloud.
constexpr int kFlHead = 128;
#ifndef PN2_FL_U
#define PN2_FL_U 4
#endif
#ifndef PN2_FL_DB
#define PN2_FL_DB 1
#endif
constexpr int kFlSupers = kFbMaxBuckets / 16;  // super-bucket = 16 consecutive buckets (compact: the curve has no jumps)
constexpr size_t kFlLds = kFlHead + (size_t)kFbMaxBuckets * (24 + 4 + 8 + 2) + 64 * (8 + 16 + 16) + kFlSupers * (8 + 4);

template <int MODE>
__global__ void __launch_bounds__(kFbThreads)
fps_bucket_lazy_kernel(int n, int npad, int m, const float* __restrict__ xyz_all, const float4* __restrict__ sorted_all,
                       float* __restrict__ td_all, int* __restrict__ out_all, float* __restrict__ new_xyz_all,
                       long long* __restrict__ stats) {
#ifdef PN2_TUNING_HOOKS
    const bool do_stats = stats != nullptr && blockIdx.x == 0;
#else
    constexpr bool do_stats = false;  // the counters compile away in the shipped library
#endif
    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // A1, A2, fallback, B, phases, fallbacks, work entries, list entries
    auto now = [&]() -> long long { return do_stats ? (long long)__builtin_readcyclecounter() : 0; };
    extern __shared__ __attribute__((aligned(16))) unsigned char fl_smem[];
    // layout: int ctrl[32] | float bb[6][2048] | int bmaxhi[2048] | u64 wl_mask[2048] | u16 wl_bk[2048] |
    //         u64 cand_key[64] | float4 cand_xyz[64] | float4 pend[64] | u64 smask[128] | int sbmax[128]
    int* ctrl = reinterpret_cast<int*>(fl_smem);  // [0],[1] list counters (alternate) | [2] work count | [3] max bmax (fallback)
                                                  // [4..7] np, j, tau_hi, - | [8..9] u64 fallback winner key
    float (*bb)[kFbMaxBuckets] = reinterpret_cast<float (*)[kFbMaxBuckets]>(fl_smem + kFlHead);
    int* bmaxhi = reinterpret_cast<int*>(&bb[6][0]);
    unsigned long long* wl_mask = reinterpret_cast<unsigned long long*>(bmaxhi + kFbMaxBuckets);
    unsigned short* wl_bk = reinterpret_cast<unsigned short*>(wl_mask + kFbMaxBuckets);
    unsigned long long* cand_key = reinterpret_cast<unsigned long long*>(wl_bk + kFbMaxBuckets);
    float4* cand_xyz = reinterpret_cast<float4*>(cand_key + 64);
    float4* pend = cand_xyz + 64;
    unsigned long long* smask = reinterpret_cast<unsigned long long*>(pend + 64);  // per super-bucket: pending picks that reach its box
    int* sbmax = reinterpret_cast<int*>(smask + kFlSupers);                       // per super-bucket: max bmaxhi of its buckets
    unsigned long long* fslot = reinterpret_cast<unsigned long long*>(ctrl + 8);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bi = blockIdx.x;
    const float* __restrict__ xyz = xyz_all + (size_t)bi * n * 3;
    const float4* __restrict__ sorted = sorted_all + (size_t)bi * npad;
    float* __restrict__ td = td_all + (size_t)bi * npad;
    int* __restrict__ out = out_all + (size_t)bi * m;
    float* __restrict__ nxyz = new_xyz_all ? new_xyz_all + (size_t)bi * m * 3 : nullptr;
    const int nb = npad >> 6;

    // bucket bounding boxes: wave w handles buckets w, w+16, ...
    for (int bk = wave; bk < nb; bk += kFbWaves) {
        const float4 p = sorted[bk * 64 + lane];
        float lo[3] = {p.x, p.y, p.z}, hi[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
                hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
            }
        if (lane == 0) {
            bb[0][bk] = lo[0]; bb[1][bk] = lo[1]; bb[2][bk] = lo[2];
            bb[3][bk] = hi[0]; bb[4][bk] = hi[1]; bb[5][bk] = hi[2];
        }
    }
    for (int bk = tid; bk < kFbMaxBuckets; bk += kFbThreads) bmaxhi[bk] = bk < nb ? __float_as_int(1e38f) : __float_as_int(-1.0f);
    if (tid == 0) {
        const float x0 = xyz[0], y0 = xyz[1], z0 = xyz[2];  // first pick is index 0 (tf_sampling.cu:122-123)
        out[0] = 0;
        if (nxyz) { nxyz[0] = x0; nxyz[1] = y0; nxyz[2] = z0; }
        pend[0] = make_float4(x0, y0, z0, 0.f);
        ctrl[0] = 0; ctrl[1] = 0; ctrl[2] = 0; ctrl[3] = -1;
        *reinterpret_cast<int4*>(ctrl + 4) = make_int4(1, 1, __float_as_int(1e38f), 0);
        *fslot = 0ull;
    }
    __syncthreads();
    float blo[kFbBPT][3], bhi[kFbBPT][3];
#pragma unroll
    for (int u = 0; u < kFbBPT; ++u) {
        const int bk = tid + kFbThreads * u;
        const int bc = bk < nb ? bk : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { blo[u][a] = bb[a][bc]; bhi[u][a] = bb[3 + a][bc]; }
    }
    // super-bucket boxes (union of 16 bucket boxes): lane = super-bucket in the coarse test; supers s and s + 64
    const int ns = (nb + 15) >> 4;
    float slo[2][3], shi[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sb = lane + 64 * u;
#pragma unroll
        for (int a = 0; a < 3; ++a) { slo[u][a] = 3e38f; shi[u][a] = -3e38f; }
        if (sb < ns) {
            for (int e = 0; e < 16; ++e) {
                const int bk = sb * 16 + e;
                if (bk < nb) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) { slo[u][a] = fminf(slo[u][a], bb[a][bk]); shi[u][a] = fmaxf(shi[u][a], bb[3 + a][bk]); }
                }
            }
        }
    }
    const unsigned wcaddr = (unsigned)(size_t)(fl_smem) + 8u;  // &ctrl[2]
    float eps = 0.2f;  // wave 0 only
    for (int ph = 0;; ph ^= 1) {
        const int4 cw = *reinterpret_cast<const int4*>(ctrl + 4);
        const int np = __builtin_amdgcn_readfirstlane(cw.x);
        const int jdone = __builtin_amdgcn_readfirstlane(cw.y);
        if (jdone >= m) break;
        const int tau_hi = __builtin_amdgcn_readfirstlane(cw.z);
        const long long t0 = now();
        // ---- A1: pick mask of every bucket, coarse to fine
        // S0: thread = bucket: bound of every super-bucket = max of its 16 bucket maxima (one 16-lane DPP row)
        int bmh[kFbBPT];
#pragma unroll
        for (int u = 0; u < kFbBPT; ++u) {
            bmh[u] = -1;
            if (kFbThreads * u >= nb) break;  // uniform
            const int bk = tid + kFbThreads * u;
            bmh[u] = bk < nb ? bmaxhi[bk] : -1;
            int r = bmh[u];
            asm volatile(
                "s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                : "+v"(r));
            if ((lane & 15) == 15) sbmax[bk >> 4] = r;
        }
        if (tid < kFlSupers) smask[tid] = 0ull;
        __syncthreads();
        // S1: lane = super-bucket, wave w takes pending picks w, w + 16, ...: which picks reach which super-bucket
        for (int p = wave; p < np; p += kFbWaves) {
            const float4 q = pend[p];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int sb = lane + 64 * u;
                if (64 * u >= ns) break;  // uniform
                const int sh = sb < ns ? sbmax[sb] : -1;
                const float Gs = fmaxf(__int_as_float(sh) * 1.000002f, 1e-30f);
                const float ex = fmaxf(fmaxf(slo[u][0] - q.x, q.x - shi[u][0]), 0.f);
                const float ey = fmaxf(fmaxf(slo[u][1] - q.y, q.y - shi[u][1]), 0.f);
                const float ez = fmaxf(fmaxf(slo[u][2] - q.z, q.z - shi[u][2]), 0.f);
                const float lb = (ex * ex + ey * ey) + ez * ez;
                if (sh >= 0 && lb <= Gs) atomicOr(&smask[sb], 1ull << p);  // one address per lane
            }
        }
        __syncthreads();
        // S2: thread = bucket: only the picks that reach my super-bucket are tested against my own box
#pragma unroll
        for (int u = 0; u < kFbBPT; ++u) {
            if (kFbThreads * u >= nb) break;  // uniform
            const int bk = tid + kFbThreads * u;
            const float Gs = fmaxf(__int_as_float(bmh[u]) * 1.000002f, 1e-30f);  // skip a pick iff lb > Gs (never on underflow)
            unsigned long long mask = 0ull;
            unsigned long long sm = bmh[u] >= 0 ? smask[bk >> 4] : 0ull;
            while (sm) {
                const int p = __builtin_ctzll(sm);
                sm &= sm - 1;
                const float4 q = pend[p];
                const float ex = fmaxf(fmaxf(blo[u][0] - q.x, q.x - bhi[u][0]), 0.f);
                const float ey = fmaxf(fmaxf(blo[u][1] - q.y, q.y - bhi[u][1]), 0.f);
                const float ez = fmaxf(fmaxf(blo[u][2] - q.z, q.z - bhi[u][2]), 0.f);
                const float lb = (ex * ex + ey * ey) + ez * ez;
                if (lb <= Gs) mask |= 1ull << p;
            }
            if (mask != 0ull || (bmh[u] >= 0 && bmh[u] >= tau_hi)) {
                unsigned slot_i;
                const unsigned one = 1u;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(slot_i) : "v"(wcaddr), "v"(one) : "memory");
                wl_bk[slot_i] = (unsigned short)bk;
                wl_mask[slot_i] = mask;
            }
        }
        __syncthreads();
        const long long t1 = now();
        // ---- A2: visit the listed buckets
        {
            const int nw = __builtin_amdgcn_readfirstlane(ctrl[2]);
            float qx = 0.f, qy = 0.f, qz = 0.f;  // lane p holds pending pick p
            if (lane < np) { const float4 q = pend[lane]; qx = q.x; qy = q.y; qz = q.z; }
            const unsigned laddr = (unsigned)(size_t)(fl_smem) + 4u * (unsigned)ph;  // &ctrl[ph]
            // U entries' loads in flight while the previous U are processed (an L2 round trip is ~1 us)
            constexpr int U = PN2_FL_U;
            struct Batch { int bks[U]; unsigned long long msk[U]; float4 pt[U]; float tdv[U]; };
            auto load = [&](int e0, Batch& B) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + kFbWaves * u;
                    B.bks[u] = -1; B.msk[u] = 0ull;
                    if (e < nw) {  // uniform
                        B.bks[u] = __builtin_amdgcn_readfirstlane((int)wl_bk[e]);
                        const unsigned long long mm = wl_mask[e];
                        B.msk[u] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mm >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mm);
                        B.pt[u] = sorted[B.bks[u] * 64 + lane];
                        B.tdv[u] = td[B.bks[u] * 64 + lane];
                    }
                }
            };
            auto process = [&](const Batch& B) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (B.bks[u] < 0) break;  // uniform
                    int hi = __float_as_int(B.tdv[u]);
                    const int hi0 = hi;
                    unsigned long long mm = B.msk[u];
                    while (mm) {
                        const int p = __builtin_ctzll(mm);
                        mm &= mm - 1;
                        const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), p));
                        const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), p));
                        const float z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), p));
                        const float d = pn2_sqdist<MODE>(B.pt[u].x - x1, B.pt[u].y - y1, B.pt[u].z - z1);
                        const int di = __float_as_int(d);  // d >= +0: int order == float order
                        hi = di < hi ? di : hi;            // min(d, td) tf_sampling.cu:151
                    }
                    if (B.msk[u] != 0ull) {  // (a bucket listed only because it holds candidates keeps its td and its maximum)
                        if (hi != hi0) td[B.bks[u] * 64 + lane] = __int_as_float(hi);
                        const int wh = pn2fps::wave_imax_from(hi);
                        if (lane == 0) bmaxhi[B.bks[u]] = wh;
                    }
                    if (hi >= tau_hi) {  // candidates of this phase (rare lanes)
                        unsigned slot_i;
                        const unsigned one = 1u;
                        asm volatile("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(slot_i) : "v"(laddr), "v"(one) : "memory");
                        if (slot_i < (unsigned)pn2fps::kLazyCap) {
                            cand_key[slot_i] = ((unsigned long long)(unsigned)hi << 32) | (unsigned)(~fb_tiekey(__float_as_int(B.pt[u].w)));
                            cand_xyz[slot_i] = B.pt[u];
                        }
                    }
                }
            };
            Batch b0, b1;
            int e0 = wave;
            if (!PN2_FL_DB) {
                for (; e0 < nw; e0 += kFbWaves * U) { load(e0, b0); process(b0); }
            } else if (e0 < nw) {
                load(e0, b0);
                for (;;) {
                    e0 += kFbWaves * U;
                    const bool more1 = e0 < nw;
                    if (more1) load(e0, b1);
                    process(b0);
                    if (!more1) break;
                    e0 += kFbWaves * U;
                    const bool more0 = e0 < nw;
                    if (more0) load(e0, b0);
                    process(b1);
                    if (!more0) break;
                }
            }
        }
        __syncthreads();
        const long long t2 = now();
        const int cnt = __builtin_amdgcn_readfirstlane(ctrl[ph]);
        const bool use_list = cnt >= 1 && cnt <= pn2fps::kLazyCap;
        if (do_stats) { st[4]++; st[5] += use_list ? 0 : 1; st[6] += ctrl[2]; st[7] += use_list ? cnt : 0; }
        if (!use_list) {
            // ---- fallback: one exact pick from a full reduction
            if (tid == 0) ctrl[2] = 0;
            {
                int mx = -1;
#pragma unroll
                for (int u = 0; u < kFbBPT; ++u) {
                    const int bk = tid + kFbThreads * u;
                    if (bk < nb) { const int v = bmaxhi[bk]; mx = v > mx ? v : mx; }
                }
                mx = pn2fps::wave_imax_from(mx);
                if (lane == 0) atomicMax(&ctrl[3], mx);
            }
            __syncthreads();
            const int G = __builtin_amdgcn_readfirstlane(ctrl[3]);
#pragma unroll
            for (int u = 0; u < kFbBPT; ++u) {
                const int bk = tid + kFbThreads * u;
                if (bk < nb && bmaxhi[bk] == G) {
                    unsigned slot_i;
                    const unsigned one = 1u;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(slot_i) : "v"(wcaddr), "v"(one) : "memory");
                    wl_bk[slot_i] = (unsigned short)bk;
                }
            }
            __syncthreads();
            {
                const int nw = __builtin_amdgcn_readfirstlane(ctrl[2]);
                for (int e = wave; e < nw; e += kFbWaves) {
                    const int bk = __builtin_amdgcn_readfirstlane((int)wl_bk[e]);
                    const float4 p = sorted[bk * 64 + lane];
                    const int hi = __float_as_int(td[bk * 64 + lane]);
                    const unsigned lo = hi == G ? (unsigned)(~fb_tiekey(__float_as_int(p.w))) : 0u;
                    const unsigned wl = pn2fps::wave_umax_all(lo);
                    if (lane == 0) {
                        const unsigned long long comp = ((unsigned long long)(unsigned)G << 32) | wl;
                        const unsigned saddr = (unsigned)(size_t)(fl_smem) + 32u;  // fslot
                        asm volatile("ds_max_u64 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(saddr), "v"(comp) : "memory");
                    }
                }
            }
            __syncthreads();
        }
        const long long t3 = now();
        // ---- B: wave 0 picks
        if (wave == 0) {
            int chi = __float_as_int(-1.0f);
            unsigned clo = 0u;
            float cx = 0.f, cy = 0.f, cz = 0.f;
            if (use_list) {
                if (lane < cnt) {
                    const unsigned long long key = cand_key[lane];
                    const float4 c = cand_xyz[lane];
                    chi = (int)(unsigned)(key >> 32); clo = (unsigned)key; cx = c.x; cy = c.y; cz = c.z;
                }
            } else if (lane == 0) {
                const unsigned long long key = *fslot;
                chi = (int)(unsigned)(key >> 32); clo = (unsigned)key;
                const int k = pn2fps::untiekey(~clo);
                cx = xyz[k * 3 + 0]; cy = xyz[k * 3 + 1]; cz = xyz[k * 3 + 2];
            }
            int maxp = use_list ? pn2fps::kLazyCap : 1;
            if (maxp > m - jdone) maxp = m - jdone;
            const int lim = use_list ? (tau_hi < 0 ? 0 : tau_hi) : 0;
            int pk_k, g_first, d_last;
            float pk_x, pk_y, pk_z;
            const int npick = pn2fps::pick_phase<MODE>(chi, clo, cx, cy, cz, lim, maxp, pk_k, pk_x, pk_y, pk_z, g_first, d_last);
            if (lane < npick) {
                out[jdone + lane] = pk_k;
                if (nxyz) { float* o = nxyz + (size_t)(jdone + lane) * 3; o[0] = pk_x; o[1] = pk_y; o[2] = pk_z; }
                pend[lane] = make_float4(pk_x, pk_y, pk_z, 0.f);
            }
            int j = jdone + npick;
            eps = pn2fps::adapt_eps(eps, cnt);
            if (npick == 0) j = m;  // unreachable (a non-empty list / the full reduction always yield a pick); never spin
            if (lane == 0) {
                ctrl[ph ^ 1] = 0; ctrl[2] = 0; ctrl[3] = -1;
                *fslot = 0ull;
                *reinterpret_cast<int4*>(ctrl + 4) = make_int4(npick, j, __float_as_int(__int_as_float(d_last) * (1.0f - eps)), g_first);
            }
        }
        __syncthreads();
        if (do_stats) { const long long t4 = now(); st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2; st[3] += t4 - t3; }
    }
    if (do_stats && tid == 0)
        for (int i = 0; i < 8; ++i) stats[i] = st[i];
}

}  // namespace

#ifdef PN2_TUNING_HOOKS
extern "C" int pn2_debug_set_fps_large_stats(long long* dev_ptr) { g_fb_stats = dev_ptr; return 0; }
extern "C" int pn2_debug_set_fps_large(int what, int value) { if (what == 11) { g_fb_variant = value; return 0; } return -1; }
#endif

extern "C" size_t pn2_fps_large_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    return fb_layout(b, n).total;
}

extern "C" int pn2_fps_large(int b, int n, int m, const float* inp, void* workspace, size_t workspace_bytes, int* out,
                             float* new_xyz, int arith_mode, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0) return PN2_EINVAL;
    if (!inp || !out || !workspace) return PN2_ENULL;
    if (n > kFbMaxBuckets * 64 || b > 65535) return PN2_ERANGE;
    if (arith_mode < 0 || arith_mode > 2) return PN2_EINVAL;
    const FbLayout L = fb_layout(b, n);
    if (workspace_bytes < L.total || ((uintptr_t)workspace & 255) != 0) return PN2_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    unsigned* bbox = reinterpret_cast<unsigned*>(ws + L.bbox);
    unsigned* keys_in = reinterpret_cast<unsigned*>(ws + L.keys_in);
    unsigned* keys_out = reinterpret_cast<unsigned*>(ws + L.keys_out);
    int* vals_in = reinterpret_cast<int*>(ws + L.vals_in);
    int* vals_out = reinterpret_cast<int*>(ws + L.vals_out);
    float4* sorted = reinterpret_cast<float4*>(ws + L.sorted);
    float* td = reinterpret_cast<float*>(ws + L.td);
    fb_init_kernel<<<(b * 6 + 255) / 256, 256, 0, st>>>(b, bbox);
    const int gx = (n + 255) / 256;
    fb_bbox_kernel<<<dim3(gx < 256 ? gx : 256, b), 256, 0, st>>>(n, inp, bbox);
    fb_keys_kernel<<<dim3(gx, b), 256, 0, st>>>(n, inp, bbox, keys_in, vals_in, (b > 1 && b <= 32) ? 1 : 0);
    PN2_RETURN_IF_LAUNCH_FAILED();
    if (b <= 32) {
        // ONE device-wide sort for the whole batch: key = cloud id above the curve index (27 bits of it when b > 1: the
        // index is hierarchical, its top bits are the index on a coarser grid), so cloud bi lands in [bi * n, (bi + 1) * n)
        int cbits = 0;
        while ((1 << cbits) < b) ++cbits;
        size_t cub = L.cub_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub, keys_in, keys_out, vals_in, vals_out, b * n, 0,
                                                         b == 1 ? 30 : 27 + cbits, st);
        if (e != hipSuccess) return (int)e;
    } else {
        for (int bi = 0; bi < b; ++bi) {
            size_t cub = L.cub_bytes;
            hipError_t e = hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub, keys_in + (size_t)bi * n, keys_out + (size_t)bi * n,
                                                             vals_in + (size_t)bi * n, vals_out + (size_t)bi * n, n, 0, 30, st);
            if (e != hipSuccess) return (int)e;
        }
    }
    fb_gather_kernel<<<dim3((L.npad + 255) / 256, b), 256, 0, st>>>(n, L.npad, inp, vals_out, sorted, td);
    static bool attr_set = false;
    if (!attr_set) {
        const void* ks[6] = {reinterpret_cast<const void*>(fps_bucket_kernel<PN2_ARITH_STRICT>), reinterpret_cast<const void*>(fps_bucket_kernel<PN2_ARITH_FMA>),
                             reinterpret_cast<const void*>(fps_bucket_kernel<PN2_ARITH_FMA_ALT>), reinterpret_cast<const void*>(fps_bucket_lazy_kernel<PN2_ARITH_STRICT>),
                             reinterpret_cast<const void*>(fps_bucket_lazy_kernel<PN2_ARITH_FMA>), reinterpret_cast<const void*>(fps_bucket_lazy_kernel<PN2_ARITH_FMA_ALT>)};
        for (int i = 0; i < 6; ++i) {
            hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)(i < 3 ? kFbLds : kFlLds));
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    if (g_fb_variant == 1) {  // tuning builds: the one-pick-per-round kernel (2.5 us per pick at n = 65536)
        switch (arith_mode) {
            case PN2_ARITH_STRICT: fps_bucket_kernel<PN2_ARITH_STRICT><<<b, kFbThreads, kFbLds, st>>>(n, L.npad, m, inp, sorted, td, out, new_xyz); break;
            case PN2_ARITH_FMA: fps_bucket_kernel<PN2_ARITH_FMA><<<b, kFbThreads, kFbLds, st>>>(n, L.npad, m, inp, sorted, td, out, new_xyz); break;
            default: fps_bucket_kernel<PN2_ARITH_FMA_ALT><<<b, kFbThreads, kFbLds, st>>>(n, L.npad, m, inp, sorted, td, out, new_xyz); break;
        }
    } else {
        switch (arith_mode) {
            case PN2_ARITH_STRICT: fps_bucket_lazy_kernel<PN2_ARITH_STRICT><<<b, kFbThreads, kFlLds, st>>>(n, L.npad, m, inp, sorted, td, out, new_xyz, g_fb_stats); break;
            case PN2_ARITH_FMA: fps_bucket_lazy_kernel<PN2_ARITH_FMA><<<b, kFbThreads, kFlLds, st>>>(n, L.npad, m, inp, sorted, td, out, new_xyz, g_fb_stats); break;
            default: fps_bucket_lazy_kernel<PN2_ARITH_FMA_ALT><<<b, kFbThreads, kFlLds, st>>>(n, L.npad, m, inp, sorted, td, out, new_xyz, g_fb_stats); break;
        }
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
