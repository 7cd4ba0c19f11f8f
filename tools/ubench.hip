// Micro-benchmarks of VALU instruction throughput on gfx950 (informs the FPS kernel design).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP 256
template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = 1.0001f, d = 0.5f;
    const f32x2 pc = {c, c}, pd = {d, d};
    int i0 = threadIdx.x, i1 = i0 * 3, i2 = i0 * 5, i3 = i0 * 7;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) {  // v_fma_f32 x8 independent
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (KIND == 1) {  // v_pk_fma_f32 x4 (8 floats)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc), "v"(pd));
            } else if (KIND == 2) {  // v_min_i32 x8
                asm volatile("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4\n"
                             "v_max_i32 %0, %0, %5\n v_max_i32 %1, %1, %5\n v_max_i32 %2, %2, %5\n v_max_i32 %3, %3, %5"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(it), "v"(iters));
            } else if (KIND == 3) {  // v_max3_i32 x8
                asm volatile("v_max3_i32 %0, %0, %4, %1\n v_max3_i32 %1, %1, %4, %2\n v_max3_i32 %2, %2, %4, %3\n v_max3_i32 %3, %3, %4, %0\n"
                             "v_max3_i32 %0, %0, %5, %1\n v_max3_i32 %1, %1, %5, %2\n v_max3_i32 %2, %2, %5, %3\n v_max3_i32 %3, %3, %5, %0"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(it), "v"(iters));
            } else if (KIND == 4) {  // dependent chain of fused dpp max (row_shr:1) with required nops
                asm volatile("v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                             "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                             "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                             "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                             "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
                             "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n"
                             "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1"
                             : "+v"(i0) : "v"(i1));
            } else if (KIND == 5) {  // v_cmp_gt + v_cndmask pairs (x4)
                asm volatile("v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %0, vcc\n v_cmp_gt_i32 vcc, %2, %3\n v_cndmask_b32 %3, %3, %2, vcc\n"
                             "v_cmp_gt_i32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_i32 vcc, %3, %2\n v_cndmask_b32 %2, %2, %3, vcc"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc");
            } else if (KIND == 7) {  // v_sub_f32 / v_mul_f32 / v_min_f32 / v_max_f32
                asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %9\n"
                             "v_sub_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (KIND == 8) {  // v_min_f32 only
                asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %9\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %9\n"
                             "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %9\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (KIND == 9) {  // v_cmp_eq_u32 -> SGPR pair (e64), 8 of them
                asm volatile("v_cmp_eq_u32 s[20:21], %0, %4\n v_cmp_eq_u32 s[22:23], %1, %4\n v_cmp_eq_u32 s[24:25], %2, %4\n v_cmp_eq_u32 s[26:27], %3, %4\n"
                             "v_cmp_eq_u32 s[28:29], %0, %5\n v_cmp_eq_u32 s[30:31], %1, %5\n v_cmp_eq_u32 s[32:33], %2, %5\n v_cmp_eq_u32 s[34:35], %3, %5"
                             : : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(it), "v"(iters) : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
            } else if (KIND == 10) {  // v_sub_f32 only
                asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %9\n"
                             "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %9\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (KIND == 11) {  // v_max3_f32
                asm volatile("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %9, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %9, %4\n"
                             "v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %9, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %9, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (KIND == 12) {  // s_ SALU ops: s_bitcmp1 + s_cselect (8 salu)
                asm volatile("s_bitcmp1_b32 %0, 3\n s_cselect_b32 s20, 1, s20\n s_bitcmp1_b32 %0, 4\n s_cselect_b32 s20, 2, s20\n"
                             "s_bitcmp1_b32 %0, 5\n s_cselect_b32 s20, 3, s20\n s_bitcmp1_b32 %0, 6\n s_cselect_b32 s20, 4, s20"
                             : : "s"(iters) : "s20", "scc");
            } else if (KIND == 6) {  // v_sub_f32 + v_pk_mul mix: pk_add x4
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %0, %0, %5\n v_pk_mul_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %5\n v_pk_mul_f32 %3, %3, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pd), "v"(pc));
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + i0 + i1 + i2 + i3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int threads, float* out, long long* cyc) {
    const int iters = 200;
    k<KIND><<<1, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipEventRecord(s); k<KIND><<<1, threads>>>(out, cyc, iters); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n_inst = (double)iters * REP;  // instructions per wave
    int waves_per_simd = threads / 256 ? threads / 256 : 1;
    printf("%-28s threads=%4d  clock64 ticks/inst/wave=%.3f  wall ns/inst/wave=%.3f  (waves/SIMD=%d => per-SIMD ns/inst=%.3f)\n", name, threads,
           c / n_inst, ms * 1e6 / n_inst, waves_per_simd, ms * 1e6 / n_inst / waves_per_simd);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1 << 12);
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fma_f32", threads, out, cyc);
        run<1>("v_pk_fma_f32 (2 fma/inst)", threads, out, cyc);
        run<6>("v_pk_add/mul_f32", threads, out, cyc);
        run<2>("v_min/max_i32", threads, out, cyc);
        run<3>("v_max3_i32", threads, out, cyc);
        run<5>("v_cmp+v_cndmask (per inst)", threads, out, cyc);
        run<4>("dpp max chain (per 8 slots)", threads, out, cyc);
        run<7>("v_sub/mul/min/max_f32 mix", threads, out, cyc);
        run<8>("v_min_f32", threads, out, cyc);
        run<10>("v_sub_f32", threads, out, cyc);
        run<11>("v_max3_f32", threads, out, cyc);
        run<9>("v_cmp_eq_u32 -> sgpr", threads, out, cyc);
        run<12>("salu bitcmp+cselect", threads, out, cyc);
    }
    return 0;
}
