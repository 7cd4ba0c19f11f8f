// MFMA loop fed like the fused MLP kernels: A operand (weights) streamed from LDS with a PF-step register
// prefetch, B operand from registers, 4 accumulators per wave.  Shows how much of the 155 TFLOP/s sustained
// MFMA rate survives the LDS->VGPR->MFMA dependency at 2 waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_lds.hip -o tools/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PF, bool BARRIER>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
    extern __shared__ float w[];  // [steps][2][128]
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    for (int i = threadIdx.x; i < 64 * 2 * 128; i += blockDim.x) w[i] = 1e-3f * i;
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float act[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) act[r] = lane * 1e-3f + r;
    const float* wl = w + half * 128 + l31;
    for (int it = 0; it < iters; ++it) {
        float wq[PF + 1][4];
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wq[p][nt] = wl[p * 256 + nt * 32];
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const int sn = s + PF < 64 ? s + PF : 63;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wq[(s + PF) % (PF + 1)][nt] = wl[sn * 256 + nt * 32];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[s % (PF + 1)][nt], act[s & 15], acc[nt], 0, 0, 0);
            if (BARRIER) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += acc[i][r];
    if (sm == 123.456f) out[threadIdx.x] = sm;
}

template <int PF, bool BARRIER>
void run(int iters) {
    float* out; (void)hipMalloc(&out, 4096);
    auto kern = k<PF, BARRIER>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    kern<<<256, 512, 64 * 1024>>>(out, 2); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    kern<<<256, 512, 64 * 1024>>>(out, iters);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    const double flops = 256.0 * 8 * iters * 64.0 * 4 * 4096.0;
    printf("prefetch=%d sched_barrier=%d: %.1f TFLOP/s (%.3f ms)\n", PF, (int)BARRIER, flops / ms * 1e-9, ms);
    (void)hipFree(out);
}

template <int PF>
void run_short(int iters, int launches) {  // many short launches: what a 1056-MFMA-per-wave kernel can reach
    float* out; (void)hipMalloc(&out, 4096);
    auto kern = k<PF, true>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    kern<<<256, 512, 64 * 1024>>>(out, 2); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    for (int l = 0; l < launches; ++l) kern<<<256, 512, 64 * 1024>>>(out, iters);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    const double flops = 256.0 * 8 * iters * 64.0 * 4 * 4096.0 * launches;
    printf("short: prefetch=%d iters=%d x %d launches: %.1f TFLOP/s (%.1f us per launch)\n", PF, iters, launches,
           flops / ms * 1e-9, ms * 1e3 / launches);
    (void)hipFree(out);
}

// ---- one wave per SIMD (r03: can ONE wave keep the matrix pipe busy?  tools/chain_stage_ab.py says a lone wave of the FP4
// kernel needs 34 k cycles for a 128 -> 128 layer, twice the pipe time).  VEC: the four column tiles' weights of a k-step
// in one ds_read_b128 (layout [step][half][l31][4]) instead of four ds_read_b32.
template <int PF, bool VEC, int THREADS>
__global__ void __launch_bounds__(THREADS) k1(float* out, int iters) {
    extern __shared__ float w[];
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    for (int i = threadIdx.x; i < 64 * 2 * 128; i += blockDim.x) w[i] = 1e-3f * i;
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float act[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) act[r] = lane * 1e-3f + r;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const float* wl = VEC ? w + (half * 32 + l31) * 4 : w + half * 128 + l31;
    auto ld = [&](float (&q)[4], int s) {
        if constexpr (VEC) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wl + s * 256);
            q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) q[nt] = wl[s * 256 + nt * 32];
        }
    };
    for (int it = 0; it < iters; ++it) {
        float wq[PF + 1][4];
#pragma unroll
        for (int p = 0; p < PF; ++p) ld(wq[p], p);
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            ld(wq[(s + PF) % (PF + 1)], s + PF < 64 ? s + PF : 63);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[s % (PF + 1)][nt], act[s & 15], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += acc[i][r];
    if (sm == 123.456f) out[threadIdx.x] = sm;
}

template <int PF, bool VEC, int THREADS>
void run1(int iters) {
    float* out; (void)hipMalloc(&out, 4096);
    auto kern = k1<PF, VEC, THREADS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    kern<<<256, THREADS, 128 * 1024>>>(out, 2); (void)hipDeviceSynchronize();   // 128 KB: one workgroup per CU
    (void)hipEventRecord(s);
    kern<<<256, THREADS, 128 * 1024>>>(out, iters);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    const double flops = 256.0 * (THREADS / 64) * iters * 64.0 * 4 * 4096.0;
    printf("waves/SIMD=%d prefetch=%d b128=%d: %.1f TFLOP/s (%.3f ms)\n", THREADS / 256, PF, (int)VEC, flops / ms * 1e-9, ms);
    (void)hipFree(out);
}

int main() {
    run1<4, false, 256>(200); run1<4, true, 256>(200); run1<8, false, 256>(200); run1<8, true, 256>(200);
    run1<4, false, 512>(200); run1<4, true, 512>(200); run1<2, true, 256>(200);
    run_short<4>(4, 20); run_short<4>(8, 20); run_short<4>(16, 20); run_short<1>(4, 20);
    run<1, true>(200); run<1, false>(200); run<2, true>(200); run<2, false>(200); run<4, true>(200); run<4, false>(200);
    return 0;
}
