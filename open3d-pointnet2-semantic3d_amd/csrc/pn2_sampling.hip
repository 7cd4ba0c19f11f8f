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
#include "pn2_common.h"

namespace {

constexpr int kFpsSlotsMax = 16;  // waves per workgroup <= 16

__device__ __forceinline__ unsigned fps_tiekey(int k) {
    return (((unsigned)k & 511u) << 22) | ((unsigned)k >> 9);
}

// Merge per-wave candidates; returns the winning index (uniform).
__device__ __forceinline__ int fps_merge_waves(unsigned long long* slots, int nwaves,
                                               int lane) {
    unsigned long long v = lane < nwaves ? slots[lane] : 0ull;
    v = pn2_row0_u64max(v);
    unsigned key = ~(unsigned)v;  // tiekey of the winner
    return (int)(((key & 0x3FFFFFu) << 9) | (key >> 22));
}

// Per-wave candidate from per-thread (best, bestk): uniform 64-bit key.
__device__ __forceinline__ unsigned long long fps_wave_candidate(float best, int bestk) {
    const bool valid = best >= 0.0f;  // threads without points keep best = -1 (tf_sampling.cu:133)
    const unsigned bits = valid ? __float_as_uint(best) : 0u;
    const unsigned wmax = pn2_wave_umax(bits);
    const unsigned long long mask = __ballot(valid && bits == wmax);
    if (mask == 0ull) return 0ull;  // whole wave has no point
    const int src = __ffsll((long long)mask) - 1;  // lowest lane = lowest residue
    const int kw = __builtin_amdgcn_readlane(bestk, src);
    return ((unsigned long long)wmax << 32) | (unsigned)(~fps_tiekey(kw));
}

template <int NT, int PPT, int MODE, bool LDS_XYZ>
__global__ void __launch_bounds__(NT)
fps_reg_kernel(int n, int m, const float* __restrict__ xyz_all, int* __restrict__ out_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: [2][16] u64 slots | float4 xyz[n] (if LDS_XYZ)
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem);
    float4* sxyz = reinterpret_cast<float4*>(smem + 2 * kFpsSlotsMax * sizeof(unsigned long long));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    constexpr int NW = NT / 64;
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    int* __restrict__ out = out_all + (size_t)blockIdx.x * m;

    float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + NT * i;
        if (k < n) {
            px[i] = xyz[k * 3 + 0];
            py[i] = xyz[k * 3 + 1];
            pz[i] = xyz[k * 3 + 2];
            md[i] = 1e38f;  // tf_sampling.cu:124-126
            if constexpr (LDS_XYZ) sxyz[k] = make_float4(px[i], py[i], pz[i], 0.f);
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            md[i] = -1.0f;  // never beats best = -1 under strict '>'
        }
    }
    if (tid == 0) out[0] = 0;  // first pick is index 0 (tf_sampling.cu:122-123)
    __syncthreads();

    int old = 0;
    for (int j = 1; j < m; ++j) {
        float x1, y1, z1;
        if constexpr (LDS_XYZ) {
            const float4 p = sxyz[old];
            x1 = p.x; y1 = p.y; z1 = p.z;
        } else {
            x1 = xyz[old * 3 + 0]; y1 = xyz[old * 3 + 1]; z1 = xyz[old * 3 + 2];
        }
        float best = -1.0f;
        int bestk = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float d = pn2_sqdist<MODE>(px[i] - x1, py[i] - y1, pz[i] - z1);
            const float d2 = fminf(d, md[i]);  // min(d, td) :151
            md[i] = d2;
            if (d2 > best) {  // strict, ascending k :153
                best = d2;
                bestk = tid + NT * i;
            }
        }
        const unsigned long long cand = fps_wave_candidate(best, bestk);
        unsigned long long* s = slots + (j & 1) * kFpsSlotsMax;
        if (lane == 0) s[wave] = cand;
        __syncthreads();
        old = fps_merge_waves(s, NW, lane);
        if (tid == 0) out[j] = old;
    }
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
    __shared__ unsigned long long slots[2 * kFpsSlotsMax];
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
            const unsigned long long cand = fps_wave_candidate(best, bestk);
            unsigned long long* s = slots + (j & 1) * kFpsSlotsMax;
            if (lane == 0) s[wave] = cand;
            __syncthreads();
            old = fps_merge_waves(s, NT / 64, lane);
            if (tid == 0) out[j] = old;
        }
        __syncthreads();  // temp / slots reuse by the next batch element
    }
}

template <int NT, int PPT, int MODE>
int launch_fps_reg(int b, int n, int m, const float* inp, int* out, hipStream_t st) {
    const size_t slots_bytes = 2 * kFpsSlotsMax * sizeof(unsigned long long);
    const size_t xyz_bytes = (size_t)n * sizeof(float4);
    // 160 KiB LDS per CU; keep the cloud in LDS when it fits (n <= 8192 -> 128 KiB)
    if (slots_bytes + xyz_bytes <= 144 * 1024) {
        auto kern = fps_reg_kernel<NT, PPT, MODE, true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(slots_bytes + xyz_bytes));
        if (e != hipSuccess) return (int)e;
        kern<<<b, NT, slots_bytes + xyz_bytes, st>>>(n, m, inp, out);
    } else {
        fps_reg_kernel<NT, PPT, MODE, false><<<b, NT, slots_bytes, st>>>(n, m, inp, out);
    }
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

template <int MODE>
int dispatch_fps(int b, int n, int m, const float* inp, float* temp, int* out, hipStream_t st) {
    // NT must be a multiple of 512 (tie-break argument in the file header).
    if (n <= 512) return launch_fps_reg<512, 1, MODE>(b, n, m, inp, out, st);
    if (n <= 1024) return launch_fps_reg<512, 2, MODE>(b, n, m, inp, out, st);
    if (n <= 2048) return launch_fps_reg<512, 4, MODE>(b, n, m, inp, out, st);
    if (n <= 4096) return launch_fps_reg<1024, 4, MODE>(b, n, m, inp, out, st);
    if (n <= 8192) return launch_fps_reg<1024, 8, MODE>(b, n, m, inp, out, st);
    if (n <= 16384) return launch_fps_reg<1024, 16, MODE>(b, n, m, inp, out, st);
    if (!temp) return PN2_ENULL;
    const int grid = b < 32 ? b : 32;
    fps_stream_kernel<MODE><<<grid, 1024, 0, st>>>(b, n, m, inp, temp, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
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

}  // namespace

extern "C" int pn2_farthest_point_sample(int b, int n, int m, const float* inp, float* temp,
                                         int* out, int arith_mode, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0) return PN2_EINVAL;
    if (!inp || !out) return PN2_ENULL;
    if ((long long)n * 3 > 0x7fffffffLL) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (arith_mode) {
        case PN2_ARITH_STRICT: return dispatch_fps<PN2_ARITH_STRICT>(b, n, m, inp, temp, out, st);
        case PN2_ARITH_FMA: return dispatch_fps<PN2_ARITH_FMA>(b, n, m, inp, temp, out, st);
        case PN2_ARITH_FMA_ALT: return dispatch_fps<PN2_ARITH_FMA_ALT>(b, n, m, inp, temp, out, st);
        default: return PN2_EINVAL;
    }
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
