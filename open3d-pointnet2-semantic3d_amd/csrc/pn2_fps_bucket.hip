// pn2_fps_bucket.hip -- exact farthest point sampling for clouds that do not fit one workgroup's registers
// (n > 16384; BASELINE configs[4]: N = 65536 -> npoint 4096).  The reference kernel (tf_sampling.cu:111-176) and
// the streaming fallback in pn2_sampling.hip re-read all n running minima every round: one workgroup, 1 MB per
// round, 11 us per round.  Here the cloud is first sorted along a Morton curve and cut into buckets of 64
// consecutive points (one wave each).  A round only touches the buckets the new pick can change:
//     bucket b is skipped  iff  lb(pick, bbox_b) * (1 - 1e-6) > max_{p in b} td[p]
// where lb is the squared distance from the pick to the bucket's bounding box: every point of a skipped bucket
// has fp32 distance >= lb*(1-1e-6) > its td, so min(td, d) == td -- bit-identical to updating it.  Each bucket
// caches its best (td, tie-break) key and the coordinates of that point, so the global argmax is a reduction
// over <= 2048 cached keys in LDS.  Same picks as the reference: first pick 0, winner = max td, ties -> lowest
// (k mod 512, k) on the ORIGINAL indices.  One 1024-thread workgroup per cloud, ~2 us per round.
#include <hipcub/hipcub.hpp>

#include "pn2_common.h"

namespace {

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
    hipcub::DeviceRadixSort::SortPairs(nullptr, cub, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                       (int*)nullptr, n, 0, 30);
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

// sort key = 30-bit Morton code of the point's cell on a 1024^3 grid over the cloud's bounding box (ordering
// only: it decides which points share a bucket, never a result)
__global__ void __launch_bounds__(256)
fb_keys_kernel(int n, const float* __restrict__ xyz_all, const unsigned* __restrict__ bbox_all,
               unsigned* __restrict__ keys, int* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int bi = blockIdx.y;
    unsigned c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = fb_unord(bbox_all[bi * 6 + a]), hi = fb_unord(bbox_all[bi * 6 + 3 + a]);
        const float ext = hi - lo;
        const float t = ext > 0.f ? (xyz_all[((size_t)bi * n + i) * 3 + a] - lo) / ext : 0.f;
        int q = (int)(t * 1023.0f);
        c[a] = (unsigned)(q < 0 ? 0 : (q > 1023 ? 1023 : q));
    }
    keys[(size_t)bi * n + i] = fb_spread10(c[0]) | (fb_spread10(c[1]) << 1) | (fb_spread10(c[2]) << 2);
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

}  // namespace

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
    fb_keys_kernel<<<dim3(gx, b), 256, 0, st>>>(n, inp, bbox, keys_in, vals_in);
    PN2_RETURN_IF_LAUNCH_FAILED();
    for (int bi = 0; bi < b; ++bi) {  // clouds this large come one or a few at a time
        size_t cub = L.cub_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub, keys_in + (size_t)bi * n, keys_out + (size_t)bi * n,
                                                         vals_in + (size_t)bi * n, vals_out + (size_t)bi * n, n, 0, 30, st);
        if (e != hipSuccess) return (int)e;
    }
    fb_gather_kernel<<<dim3((L.npad + 255) / 256, b), 256, 0, st>>>(n, L.npad, inp, vals_out, sorted, td);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(fps_bucket_kernel<PN2_ARITH_STRICT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFbLds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(fps_bucket_kernel<PN2_ARITH_FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFbLds);
        hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(fps_bucket_kernel<PN2_ARITH_FMA_ALT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFbLds);
        if (e1 != hipSuccess) return (int)e1;
        if (e2 != hipSuccess) return (int)e2;
        if (e3 != hipSuccess) return (int)e3;
        attr_set = true;
    }
    switch (arith_mode) {
        case PN2_ARITH_STRICT:
            fps_bucket_kernel<PN2_ARITH_STRICT><<<b, kFbThreads, kFbLds, st>>>(n, L.npad, m, inp, sorted, td, out, new_xyz);
            break;
        case PN2_ARITH_FMA:
            fps_bucket_kernel<PN2_ARITH_FMA><<<b, kFbThreads, kFbLds, st>>>(n, L.npad, m, inp, sorted, td, out, new_xyz);
            break;
        default:
            fps_bucket_kernel<PN2_ARITH_FMA_ALT><<<b, kFbThreads, kFbLds, st>>>(n, L.npad, m, inp, sorted, td, out, new_xyz);
            break;
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
