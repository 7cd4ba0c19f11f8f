// What does a SIMD partner's activity cost the wave that drives the matrix pipe?  Waves 0..3 of an 8-wave workgroup (one per
// SIMD) run the fused-MLP MFMA loop (weights from LDS, 4 accumulators) for a fixed number of steps and report their cycles;
// waves 4..7 (their SIMD partners) meanwhile: 0 = exit, 1 = dependent VALU chain, 2 = scattered 16-byte gathers (the FP front
// end's z rows: 32 rows x 2 halves per instruction), 3 = LDS reads, 4 = coalesced stores, 5 = s_sleep spin.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_interference_ubench.hip -o tools/mfma_interference_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512, 1) k(int mode, int iters, const float* __restrict__ table, int rows, float* sink,
                                            long long* cyc) {
    extern __shared__ float w[];  // [64 steps][2][128]
    __shared__ int done;
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 2 * 128; i += blockDim.x) w[i] = 1e-3f * i;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave < 4) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float act[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) act[r] = lane * 1e-3f + r;
        const float* wl = w + half * 128 + l31;
        constexpr int PF = 4;
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            float wq[PF][4];
#pragma unroll
            for (int p = 0; p < PF; ++p)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wq[p][nt] = wl[p * 256 + nt * 32];
#pragma unroll
            for (int s = 0; s < 64; ++s) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[s % PF][nt], act[s & 15], acc[nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int sn = s + PF < 64 ? s + PF : 63;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wq[s % PF][nt] = wl[sn * 256 + nt * 32];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const long long t1 = __builtin_readcyclecounter();
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += acc[i][r];
        if (sm == 123.456f) sink[threadIdx.x] = sm;
        if (lane == 0) { cyc[blockIdx.x * 4 + wave] = t1 - t0; __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        return;
    }
    if (mode == 0) return;
    float v = lane;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    unsigned r = (blockIdx.x * 977u + threadIdx.x * 131u) % (unsigned)rows;
    long long n = 0;
    while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
        if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 64; ++i) v = v * 1.0001f + 0.5f;
        } else if (mode == 2) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {  // 12 loads in flight, like one column chunk of the front end
                const f32x4* p = reinterpret_cast<const f32x4*>(table + (size_t)((r + l31 * 7919u + i * 104729u) % (unsigned)rows) * 128) + half + 2 * (i & 3);
                const f32x4 t = *p;
                a += t;
            }
            r = r * 1664525u + 1013904223u;
        } else if (mode == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v += w[(lane + i * 64 + (int)n) & 16383];
        } else if (mode == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sink[4096 + (size_t)blockIdx.x * 65536 + ((n * 16 + i) & 255) * 64 + lane] = v;
        } else {
            __builtin_amdgcn_s_sleep(4);
        }
        ++n;
    }
    if (v + a[0] + a[1] + a[2] + a[3] == 123.456f) sink[threadIdx.x] = v;
    if (lane == 0) cyc[1024 + blockIdx.x * 4 + (wave - 4)] = n;
}

int main() {
    const int rows = 16384;
    float *table, *sink; long long* cyc;
    (void)hipMalloc(&table, (size_t)rows * 128 * 4); (void)hipMemset(table, 0, (size_t)rows * 128 * 4);
    (void)hipMalloc(&sink, (4096 + 256 * 65536) * 4); (void)hipMalloc(&cyc, 4096 * 8);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const char* names[] = {"partner absent", "partner: VALU chain", "partner: scattered 16 B gathers", "partner: LDS reads",
                           "partner: coalesced stores", "partner: s_sleep spin"};
    const int iters = 20;
    for (int mode = 0; mode < 6; ++mode) {
        (void)hipMemset(cyc, 0, 4096 * 8);
        k<<<256, 512, 64 * 1024>>>(mode, 2, table, rows, sink, cyc);
        k<<<256, 512, 64 * 1024>>>(mode, iters, table, rows, sink, cyc);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(4096);
        (void)hipMemcpy(h.data(), cyc, 4096 * 8, hipMemcpyDeviceToHost);
        double s = 0, pn = 0; for (int i = 0; i < 1024; ++i) { s += h[i]; pn += h[1024 + i]; }
        const double per_layer = s / 1024 / iters;  // cycles per 256 MFMAs (one 128 -> 128 layer of a tile)
        printf("%-34s %8.0f cycles per 256 MFMAs (%.3f of the pipe), partner iterations per wave %.0f\n", names[mode], per_layer,
               16384.0 / per_layer, pn / 1024);
    }
    return 0;
}
