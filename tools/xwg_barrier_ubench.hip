// Cost of one barrier between G workgroups through L2 (atomic arrive + polling), the per-round price a multi-workgroup FPS
// for one large cloud (BASELINE configs[4]: B = 1, N = 65536) would pay on top of its local work.
//   hipcc --offload-arch=gfx950 -O3 tools/xwg_barrier_ubench.hip -o tools/xwg_barrier_ubench && tools/xwg_barrier_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void barrier_kernel(int rounds, int G, unsigned* counter, unsigned long long* key, int* fail) {
    const int tid = threadIdx.x;
    unsigned long long best = 0;
    for (int r = 0; r < rounds; ++r) {
        if (tid == 0) {
            // publish a candidate (one 64-bit atomic max, as the FPS winner exchange would), then arrive
            atomicMax(&key[r & 1], ((unsigned long long)(r + 1) << 32) | (unsigned)(blockIdx.x * 7 + r));
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)G * (unsigned)(r + 1);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 2000000) { *fail = 1; break; }  // bounded: never hang the box
            }
            __threadfence();
            best = __hip_atomic_load(&key[r & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid == 0 && blockIdx.x == 0) key[(r + 1) & 1] = 0;  // reset the other slot for the next round
        __syncthreads();
    }
    if (tid == 0 && best == 42) *fail = 2;
}

int main() {
    unsigned* counter; unsigned long long* key; int* fail;
    hipMalloc(&counter, 4); hipMalloc(&key, 16); hipMalloc(&fail, 4);
    const int rounds = 20000;
    for (int G : {1, 2, 4, 8, 16, 32}) {
        for (int threads : {64, 1024}) {
            hipMemset(counter, 0, 4); hipMemset(key, 0, 16); hipMemset(fail, 0, 4);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            barrier_kernel<<<G, threads>>>(rounds, G, counter, key, fail);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            int f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
            printf("G=%2d workgroups x %4d threads: %.0f ns per round%s\n", G, threads, ms * 1e6 / rounds, f ? "  (spin limit hit)" : "");
        }
    }
    return 0;
}
