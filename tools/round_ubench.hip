// Cost of the pieces of a barrier-synchronised "round" on one gfx950 CU (informs the FPS kernel design).
// Build: hipcc --offload-arch=gfx950 -O3 tools/round_ubench.hip -o tools/round_ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

// KIND bits: 1 = barrier, 2 = uniform key read (ds_read_b64 + readfirstlane), 4 = dependent xyz read (ds_read_b128),
//            8 = ds_max_u64 by lane 0 of every wave + waitcnt, 16 = 24 dependent VALU (fma chain), 32 = 24 independent-ish VALU (3 chains of 8)
//            64 = wave DPP max reduce (6 steps) + readlane, 128 = 12 dependent SALU
template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, int rounds) {
    __shared__ unsigned long long slots[4];
    __shared__ float4 xyz[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    xyz[tid] = make_float4(tid * 0.5f, tid * 0.25f, tid, 0.f);
    if (tid < 4) slots[tid] = (unsigned long long)(tid * 7 + 1);
    __syncthreads();
    int old = 0;
    float acc = tid * 1e-3f, a1 = acc + 1, a2 = acc + 2;
    int iv = tid;
    unsigned s = 1;
    for (int j = 1; j < rounds; ++j) {
        float4 p = make_float4(1.f, 2.f, 3.f, 0.f);
        if (KIND & 4) p = xyz[old & 1023];
        if (KIND & 16) {
#pragma unroll
            for (int r = 0; r < 24; ++r) acc = __builtin_fmaf(acc, 1.0001f, p.x);
        }
        if (KIND & 32) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { acc = __builtin_fmaf(acc, 1.0001f, p.x); a1 = __builtin_fmaf(a1, 1.0001f, p.y); a2 = __builtin_fmaf(a2, 1.0001f, p.z); }
        }
        if (KIND & 64) {
            int v = __float_as_int(acc) ^ iv;
            asm volatile(
                "s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n"
                : "+v"(v));
            iv = __builtin_amdgcn_readlane(v, 63);
        }
        if (KIND & 128) {
#pragma unroll
            for (int r = 0; r < 12; ++r) s = __builtin_amdgcn_readfirstlane(s) * 3u + 1u;
        }
        if (KIND & 8) {
            if (lane == 0) {
                const unsigned saddr = (unsigned)(size_t)(&slots[j & 1]);
                const unsigned long long comp = ((unsigned long long)(unsigned)__float_as_int(acc) << 32) | (unsigned)(iv + j);
                asm volatile("ds_max_u64 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(saddr), "v"(comp) : "memory");
            }
        }
        if (KIND & 1) __syncthreads();
        if (KIND & 2) {
            const unsigned long long win = slots[j & 1];
            old = __builtin_amdgcn_readfirstlane((int)(unsigned)win) + j;
        }
    }
    out[blockIdx.x * blockDim.x + tid] = acc + a1 + a2 + old + iv + s;
}

template <int KIND>
void run(const char* name, float* out) {
    const int rounds = 4096;
    printf("%-58s", name);
    for (int threads : {64, 256, 512, 1024}) {
        k<KIND><<<16, threads>>>(out, rounds);
        hipDeviceSynchronize();
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        hipEventRecord(s); k<KIND><<<16, threads>>>(out, rounds); hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        printf("  NT=%4d: %6.1f ns/round", threads, ms * 1e6 / rounds);
    }
    printf("\n");
}

int main() {
    float* out; hipMalloc(&out, 1 << 22);
    run<1>("barrier only", out);
    run<1 | 2>("barrier + key read", out);
    run<1 | 2 | 4>("barrier + key read + dependent xyz read", out);
    run<1 | 2 | 4 | 8>("  + ds_max_u64 (lane 0 of each wave) + wait", out);
    run<1 | 2 | 4 | 8 | 16>("  + 24 dependent fma", out);
    run<1 | 2 | 4 | 8 | 32>("  + 24 fma in 3 chains", out);
    run<1 | 2 | 4 | 8 | 64>("  + wave DPP max + readlane", out);
    run<1 | 2 | 4 | 8 | 128>("  + 12 dependent SALU(readfirstlane+mul+add)", out);
    run<16>("24 dependent fma only (no sync)", out);
    run<32>("24 fma in 3 chains only", out);
    run<64>("wave DPP max + readlane only", out);
    run<4 | 2>("key read + dependent xyz read, no barrier", out);
    run<8>("ds_max_u64 + wait only", out);
    return 0;
}
