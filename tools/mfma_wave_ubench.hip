// How dense can ONE wave per SIMD keep the fp32 matrix pipe?  The fused-MLP loop shape (weights = A operand streamed from LDS
// with a 4-step register prefetch, activations = B operand from registers) with T row tiles per wave sharing each weight
// read (T x 4 accumulators of 16 registers): T = 1 at 2 waves/SIMD is what sa_fused_kernel runs; T = 1 and T = 2 at one
// wave per SIMD are the candidates of the software-pipelined chain kernel.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_wave_ubench.hip -o tools/mfma_wave_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int T, int NTHREADS, int MINB>
__global__ void __launch_bounds__(NTHREADS, MINB) k(float* out, int iters) {
    extern __shared__ float w[];  // [steps][2][128]
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    for (int i = threadIdx.x; i < 64 * 2 * 128; i += blockDim.x) w[i] = 1e-3f * i;
    __syncthreads();
    f32x16 acc[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
    float act[T][16];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[t][r] = lane * 1e-3f + r + t;
    const float* wl = w + half * 128 + l31;
    constexpr int PF = 4;
    for (int it = 0; it < iters; ++it) {
        float wq[PF][4];
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wq[p][nt] = wl[p * 256 + nt * 32];
#pragma unroll
        for (int s = 0; s < 64; ++s) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[s % PF][nt], act[t][s & 15], acc[t][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int sn = s + PF < 64 ? s + PF : 63;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wq[s % PF][nt] = wl[sn * 256 + nt * 32];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += acc[t][i][r];
    if (sm == 123.456f) out[threadIdx.x] = sm;
}

template <int T, int NTHREADS, int MINB>
void run(const char* name, int iters) {
    float* out; (void)hipMalloc(&out, 4096);
    auto kern = k<T, NTHREADS, MINB>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    const int grid = 256 * MINB;
    kern<<<grid, NTHREADS, 64 * 1024>>>(out, 2); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    kern<<<grid, NTHREADS, 64 * 1024>>>(out, iters);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    const double flops = (double)grid * (NTHREADS / 64) * iters * 64.0 * 4 * T * 4096.0;
    printf("%-40s %.1f TFLOP/s (%.3f of 157.3)\n", name, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 157.3e12);
    (void)hipFree(out);
}

int main() {
    run<1, 512, 1>("T=1, 8 waves/CU (2 per SIMD)", 400);
    run<1, 256, 1>("T=1, 4 waves/CU (1 per SIMD)", 400);
    run<2, 256, 1>("T=2, 4 waves/CU (1 per SIMD)", 400);
    run<2, 512, 1>("T=2, 8 waves/CU (2 per SIMD, 256 regs)", 400);
    run<1, 768, 1>("T=1, 12 waves/CU (3 per SIMD)", 400);
    return 0;
}
