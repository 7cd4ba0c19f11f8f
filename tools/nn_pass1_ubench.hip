// Pass 1 of three_nn (per-query minimum of the expanded squared distance over 1024 known points) in four forms, to decide
// how csrc/pn2_interpolate.hip should compute it.  B = 16 clouds, n = 8192 queries, m = 1024 known points.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/nn_pass1_ubench.hip -o tools/nn_pass1_ubench ; run on the GPU box.
//   A  lane = known point (16 per lane in registers), 8 queries per group: 3 v_fma + v_min per pair  (the shipped form)
//   B  same, one v_min3 per two known points
//   C  same, two queries per v_pk_fma_f32 (known point broadcast by op_sel), v_min3
//   D  lane = query (32 per wave), 32 known points x 32 queries per v_mfma_f32_32x32x2_f32 pair (K = 4: x, y, z, |c|^2 * 1),
//      16 v_min per tile into 16 per-register-slot minima
//   E  D with v_min3 over two tiles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kQ = 8;

__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int VAR>
__global__ void __launch_bounds__(256) lane_is_point(int n, int m, const float* __restrict__ xyz1_all, const float* __restrict__ xyz2_all,
                                                     float* __restrict__ out_all, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bi = blockIdx.y;
    const float* xyz1 = xyz1_all + (size_t)bi * n * 3;
    const float* xyz2 = xyz2_all + (size_t)bi * m * 3;
    float cx[16], cy[16], cz[16], cc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int k = t * 64 + lane;
        cx[t] = xyz2[k * 3]; cy[t] = xyz2[k * 3 + 1]; cz[t] = xyz2[k * 3 + 2];
        cc[t] = __builtin_fmaf(cz[t], cz[t], __builtin_fmaf(cy[t], cy[t], cx[t] * cx[t]));
    }
    const int ngroups = n / kQ, gstride = gridDim.x * 4;
    for (int rep = 0; rep < reps; ++rep)
    for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gstride) {
        const int l = lane < 24 ? lane : 23;
        const float qv = xyz1[(grp * kQ + l / 3) * 3 + l % 3];
        float ax[kQ], ay[kQ], az[kQ], mn[kQ];
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            ax[q] = -2.f * __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qv), 3 * q));
            ay[q] = -2.f * __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qv), 3 * q + 1));
            az[q] = -2.f * __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(qv), 3 * q + 2));
            mn[q] = INFINITY;
        }
        if (VAR == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int q = 0; q < kQ; ++q)
                    mn[q] = vmin(mn[q], __builtin_fmaf(ax[q], cx[t], __builtin_fmaf(ay[q], cy[t], __builtin_fmaf(az[q], cz[t], cc[t]))));
        } else if (VAR == 1) {
#pragma unroll
            for (int t = 0; t < 16; t += 2)
#pragma unroll
                for (int q = 0; q < kQ; ++q) {
                    const float s0 = __builtin_fmaf(ax[q], cx[t], __builtin_fmaf(ay[q], cy[t], __builtin_fmaf(az[q], cz[t], cc[t])));
                    const float s1 = __builtin_fmaf(ax[q], cx[t + 1], __builtin_fmaf(ay[q], cy[t + 1], __builtin_fmaf(az[q], cz[t + 1], cc[t + 1])));
                    mn[q] = vmin3(mn[q], s0, s1);
                }
        } else {
#pragma unroll
            for (int t = 0; t < 16; t += 2)
#pragma unroll
                for (int q = 0; q < kQ; q += 2) {
                    const f32x2 a = {ax[q], ax[q + 1]}, b = {ay[q], ay[q + 1]}, c = {az[q], az[q + 1]};
                    f32x2 s0 = {cc[t], cc[t]}, s1 = {cc[t + 1], cc[t + 1]};
                    s0 = __builtin_elementwise_fma(c, (f32x2){cz[t], cz[t]}, s0);
                    s1 = __builtin_elementwise_fma(c, (f32x2){cz[t + 1], cz[t + 1]}, s1);
                    s0 = __builtin_elementwise_fma(b, (f32x2){cy[t], cy[t]}, s0);
                    s1 = __builtin_elementwise_fma(b, (f32x2){cy[t + 1], cy[t + 1]}, s1);
                    s0 = __builtin_elementwise_fma(a, (f32x2){cx[t], cx[t]}, s0);
                    s1 = __builtin_elementwise_fma(a, (f32x2){cx[t + 1], cx[t + 1]}, s1);
                    mn[q] = vmin3(mn[q], s0[0], s1[0]);
                    mn[q + 1] = vmin3(mn[q + 1], s0[1], s1[1]);
                }
        }
        // wave minimum per query (not part of the shipped pass 1, keeps the work alive): lane q writes
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            float v = mn[q];
#pragma unroll
            for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o));
            r = lane == q ? v : r;
        }
        if (lane < kQ) out_all[(size_t)bi * n + grp * kQ + lane] = r;
    }
}

template <int VAR>
__global__ void __launch_bounds__(256) lane_is_query(int n, int m, const float* __restrict__ xyz1_all, const float* __restrict__ xyz2_all,
                                                     float* __restrict__ out_all, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bi = blockIdx.y;
    const int col = lane & 31, half = lane >> 5;
    const float* xyz1 = xyz1_all + (size_t)bi * n * 3;
    const float* xyz2 = xyz2_all + (size_t)bi * m * 3;
    // A operands: tile t (32 known points), MFMA 1: k = half -> x | y, MFMA 2: z | |c|^2
    float a1[32], a2[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const int k = t * 32 + col;
        const float x = xyz2[k * 3], y = xyz2[k * 3 + 1], z = xyz2[k * 3 + 2];
        a1[t] = half ? y : x;
        a2[t] = half ? __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)) : z;
    }
    const int ngroups = n / 32, gstride = gridDim.x * 4;
    for (int rep = 0; rep < reps; ++rep)
    for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gstride) {
        const int q = grp * 32 + col;
        const float qx = xyz1[q * 3], qy = xyz1[q * 3 + 1], qz = xyz1[q * 3 + 2];
        const float b1 = half ? -2.f * qy : -2.f * qx;
        const float b2 = half ? 1.f : -2.f * qz;
        float mn[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) mn[r] = INFINITY;
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (VAR == 0) {
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1, zero, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t], b2, d, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) mn[r] = vmin(mn[r], d[r]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 32; t += 2) {
                f32x16 d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1, zero, 0, 0, 0);
                f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t + 1], b1, zero, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t], b2, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t + 1], b2, d1, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) mn[r] = vmin3(mn[r], d0[r], d1[r]);
            }
        }
        float v = mn[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) v = fminf(v, mn[r]);
        v = fminf(v, __shfl_xor(v, 32));
        if (half == 0) out_all[(size_t)bi * n + q] = v;
    }
}

int main() {
    const int B = 16, n = 8192, m = 1024;
    std::vector<float> h1((size_t)B * n * 3), h2((size_t)B * m * 3);
    srand(1);
    for (auto& v : h1) v = (rand() % 20001 - 10000) * 5e-4f;
    for (int b = 0; b < B; ++b) for (int k = 0; k < m * 3; ++k) h2[(size_t)b * m * 3 + k] = h1[(size_t)b * n * 3 + k];
    float *d1, *d2, *o, *oref;
    hipMalloc(&d1, h1.size() * 4); hipMalloc(&d2, h2.size() * 4); hipMalloc(&o, (size_t)B * n * 4); hipMalloc(&oref, (size_t)B * n * 4);
    hipMemcpy(d1, h1.data(), h1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d2, h2.data(), h2.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ref((size_t)B * n), got((size_t)B * n);
    auto run = [&](const char* name, auto launch, bool is_ref) {
        for (int reps : {1, 3}) for (int blocks : {64, 256}) {   // workgroups per cloud (x 16 clouds)
            float* dst = is_ref && blocks == 64 && reps == 1 ? oref : o;
            for (int i = 0; i < 3; ++i) launch(blocks, dst, reps);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) launch(blocks, dst, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double err = 0;
            if (!(is_ref && blocks == 64 && reps == 1)) {
                hipMemcpy(got.data(), dst, got.size() * 4, hipMemcpyDeviceToHost);
                for (size_t i = 0; i < got.size(); ++i) err = fmax(err, fabs((double)got[i] - ref[i]));
            } else hipMemcpy(ref.data(), oref, ref.size() * 4, hipMemcpyDeviceToHost);
            printf("%-28s reps %d blocks/cloud %3d  %7.2f us   max |diff to A| %.3g\n", name, reps, blocks, ms * 1000 / 20, err);
        }
    };
    run("A lane=point fma+min", [&](int g, float* dst, int reps) { lane_is_point<0><<<dim3(g, B), 256>>>(n, m, d1, d2, dst, reps); }, true);
    run("B lane=point fma+min3", [&](int g, float* dst, int reps) { lane_is_point<1><<<dim3(g, B), 256>>>(n, m, d1, d2, dst, reps); }, false);
    run("C lane=point pk_fma+min3", [&](int g, float* dst, int reps) { lane_is_point<2><<<dim3(g, B), 256>>>(n, m, d1, d2, dst, reps); }, false);
    run("D lane=query mfma+min", [&](int g, float* dst, int reps) { lane_is_query<0><<<dim3(g, B), 256>>>(n, m, d1, d2, dst, reps); }, false);
    run("E lane=query mfma+min3", [&](int g, float* dst, int reps) { lane_is_query<1><<<dim3(g, B), 256>>>(n, m, d1, d2, dst, reps); }, false);
    return 0;
}
