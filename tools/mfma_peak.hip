// Sustained fp32 MFMA rate on gfx950: v_mfma_f32_32x32x2_f32 with NACC independent accumulators per wave,
// W waves per SIMD, no memory traffic.  Calibrates what "100 %" means for the fused MLP kernels
// (the 157.3 TFLOP/s peak assumes 2.4 GHz sustained).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f - a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int NACC>
void run(int wavesPerSimd, int iters) {
    float* out; hipMalloc(&out, 4096);
    const int threads = 64 * 4 * wavesPerSimd > 512 ? 512 : 64 * 4 * wavesPerSimd;
    const int blocksPerCu = (64 * 4 * wavesPerSimd) / threads;
    const int grid = 256 * blocksPerCu;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<NACC><<<grid, threads>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(s);
    k<NACC><<<grid, threads>>>(out, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double flops = (double)grid * (threads / 64) * iters * 16.0 * NACC * 4096.0;
    printf("NACC=%d waves/SIMD=%d: %.1f TFLOP/s (%.3f ms)\n", NACC, wavesPerSimd, flops / ms * 1e-9, ms);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) { run<1>(w, 4000); run<2>(w, 2000); run<4>(w, 1000); }
    // long run: sustained clocks
    run<4>(2, 40000);
    return 0;
}
