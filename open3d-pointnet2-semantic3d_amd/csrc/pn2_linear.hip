// pn2_linear.hip -- one dense layer of the PointNet++ shared MLP on the fp32
// matrix cores of gfx950, plus the grouped-input builder used by the unfused path.
//
// The reference runs each 1x1 conv as tf.nn.conv2d + bias_add + batch_norm + relu
// (util/tf_util.py:181-203) and the K-pool as a separate tf.reduce_max
// (util/pointnet_util.py:167-170), each re-reading the (B,M,K,C) tensor from HBM.
// Here: y = relu(x @ W + b) with the inference BatchNorm folded into (W, b) by the
// host and the max over the K neighbours taken in the MFMA epilogue.
//
// Kernel shape: 256 threads = 4 waves; block tile 128 rows x BN cols (BN = 32*NT),
// each wave owns 32 rows x BN cols = NT accumulators of v_mfma_f32_32x32x2_f32
// (exact fp32 products, fp32 accumulate: bitwise an fmaf chain).  K is tiled by
// 16 through double-buffered LDS (one barrier per tile), global->register
// prefetch of tile t+1 overlaps the MFMAs of tile t.  LDS layouts are k-major with
// row strides 130 / BN+4 floats so both operand reads (lane -> consecutive
// floats) and the transposing A stores are bank-conflict free.
// Because a wave's 32 rows are exactly one K=32 neighbourhood, the max-pool is an
// in-register max over the 16 accumulator rows + one cross-half exchange.
#include "pn2_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128;
constexpr int kBK = 16;
constexpr int kAS = kBM + 2;  // A tile row stride (floats)

__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
    // v >= 0 (post-ReLU): integer order == float order
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
}

template <int NT, bool VEC_A>
__global__ void __launch_bounds__(256)
linear_kernel(int rows, int cin, int cout, const float* __restrict__ x,
              const float* __restrict__ w, const float* __restrict__ bias, int relu, int pool,
              float* __restrict__ y) {
    constexpr int BN = NT * 32;
    constexpr int BS = BN + 4;  // B tile row stride (floats), keeps float4 stores 16-B aligned
    constexpr int B_F4 = kBK * BN / 4;             // float4 per B tile
    constexpr int B_PER_T = (B_F4 + 255) / 256;    // float4 per thread
    __shared__ __attribute__((aligned(16))) float As[2][kBK * kAS];
    __shared__ __attribute__((aligned(16))) float Bs[2][kBK * BS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int row0 = blockIdx.x * kBM;
    const int col0 = blockIdx.y * BN;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // ---- staging registers -------------------------------------------------
    f32x4 a_v[2];      // VEC_A: 2 float4 per thread
    float a_s[8];      // scalar path: 8 floats per thread
    f32x4 b_v[B_PER_T];

    auto load_tile = [&](int kt) {
        const int k0 = kt * kBK;
        if constexpr (VEC_A) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + 256 * i;
                const int r = f >> 2, k4 = f & 3;
                const int gr = row0 + r, gk = k0 + k4 * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gr < rows && gk < cin)
                    v = *reinterpret_cast<const f32x4*>(x + (size_t)gr * cin + gk);
                a_v[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i;
                const int k = e & 15, r = e >> 4;
                const int gr = row0 + r, gk = k0 + k;
                a_s[i] = (gr < rows && gk < cin) ? x[(size_t)gr * cin + gk] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            const int f = tid + 256 * i;
            const int k = f / (BN / 4), n4 = f % (BN / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f < B_F4 && k0 + k < cin)
                v = *reinterpret_cast<const f32x4*>(w + (size_t)(k0 + k) * cout + col0 + n4 * 4);
            b_v[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* as = As[buf];
        float* bs = Bs[buf];
        if constexpr (VEC_A) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + 256 * i;
                const int r = f >> 2, k4 = f & 3;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) as[(k4 * 4 + jj) * kAS + r] = a_v[i][jj];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i;
                const int k = e & 15, r = e >> 4;
                as[k * kAS + r] = a_s[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            const int f = tid + 256 * i;
            const int k = f / (BN / 4), n4 = f % (BN / 4);
            if (f < B_F4) *reinterpret_cast<f32x4*>(bs + k * BS + n4 * 4) = b_v[i];
        }
    };

    const int nkt = (cin + kBK - 1) / kBK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const float* as = As[buf] + wave * 32 + l31;
        const float* bs = Bs[buf] + l31;
#pragma unroll
        for (int ks = 0; ks < kBK / 2; ++ks) {
            const int k = ks * 2 + half;
            const float a = as[k * kAS];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float bb = bs[k * BS + nt * 32];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[nt], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------
    // D[i][j]: j = l31, i = (r&3) + 8*(r>>2) + 4*half
    const int wrow0 = row0 + wave * 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = col0 + nt * 32 + l31;
        const float bv = bias ? bias[col] : 0.f;
        if (pool <= 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[nt][r] + bv;
                if (relu) v = fmaxf(v, 0.f);
                if (row < rows) y[(size_t)row * cout + col] = v;
            }
        } else if (pool == 16) {
            // rows 0..15 live in regs 0..7, rows 16..31 in regs 8..15 (both halves)
            float v0 = acc[nt][0], v1 = acc[nt][8];
#pragma unroll
            for (int r = 1; r < 8; ++r) { v0 = fmaxf(v0, acc[nt][r]); v1 = fmaxf(v1, acc[nt][8 + r]); }
            v0 = fmaxf(v0, __shfl_xor(v0, 32));
            v1 = fmaxf(v1, __shfl_xor(v1, 32));
            v0 += bv; v1 += bv;
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            const int g0 = wrow0 / 16;
            if (half == 0 && wrow0 < rows) y[(size_t)g0 * cout + col] = v0;
            if (half == 0 && wrow0 + 16 < rows) y[(size_t)(g0 + 1) * cout + col] = v1;
        } else {
            float v = acc[nt][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[nt][r]);
            v = fmaxf(v, __shfl_xor(v, 32));
            v += bv;  // max_i relu(x_i + b) == relu(max_i(x_i) + b): fl(x+b) and relu are monotone
            if (relu) v = fmaxf(v, 0.f);
            if (half == 0 && wrow0 < rows) {
                const int g = wrow0 / pool;
                if (pool == 32) y[(size_t)g * cout + col] = v;
                else atomic_max_nonneg(&y[(size_t)g * cout + col], v);  // pool = 32*t, y pre-zeroed, relu on
            }
        }
    }
}

template <int NT>
int launch_linear(int rows, int cin, int cout, const float* x, const float* w, const float* bias,
                  int relu, int pool, float* y, hipStream_t st) {
    dim3 grid((rows + kBM - 1) / kBM, cout / (NT * 32));
    const bool vec_a = (cin % 4 == 0) && ((uintptr_t)x % 16 == 0);
    if (vec_a) linear_kernel<NT, true><<<grid, 256, 0, st>>>(rows, cin, cout, x, w, bias, relu, pool, y);
    else linear_kernel<NT, false><<<grid, 256, 0, st>>>(rows, cin, cout, x, w, bias, relu, pool, y);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

// grouped input of an SA layer: out[b,j,k,:] = [xyz[b,idx]-new_xyz[b,j] | points[b,idx]]
// (xyz FIRST: util/pointnet_util.py:43-54).  One thread per output float.
__global__ void __launch_bounds__(256)
sa_group_concat_kernel(int n, int m, int nsample, int c, const float* __restrict__ xyz_all,
                       const float* __restrict__ new_xyz_all, const float* __restrict__ points_all,
                       const int* __restrict__ idx_all, float* __restrict__ out_all) {
    const unsigned cw = 3u + (unsigned)c;
    const unsigned rows = (unsigned)m * nsample;
    const unsigned total = rows * cw;
    const int bi = blockIdx.y;
    const float* __restrict__ xyz = xyz_all + (size_t)bi * n * 3;
    const float* __restrict__ nxyz = new_xyz_all + (size_t)bi * m * 3;
    const float* __restrict__ pts = points_all ? points_all + (size_t)bi * n * c : nullptr;
    const int* __restrict__ idx = idx_all + (size_t)bi * rows;
    float* __restrict__ out = out_all + (size_t)bi * rows * cw;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned row = e / cw, col = e - row * cw;
        const int ii = idx[row];
        float v;
        if (col < 3u) v = xyz[(size_t)ii * 3 + col] - nxyz[(size_t)(row / nsample) * 3 + col];
        else v = pts[(size_t)ii * c + (col - 3u)];
        out[e] = v;
    }
}

}  // namespace

extern "C" int pn2_linear(int rows, int cin, int cout, const float* x, const float* w,
                          const float* bias, int relu, int pool, float* y, void* stream) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return PN2_EINVAL;
    if (!x || !w || !y) return PN2_ENULL;
    if (cout % 32 != 0 || ((uintptr_t)w % 16) != 0) return PN2_EUNSUP;
    if ((long long)rows + kBM > 0x7fffffffLL) return PN2_ERANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pool > 1) {
        if (pool != 16 && pool % 32 != 0) return PN2_EUNSUP;
        if (rows % pool != 0) return PN2_EINVAL;
        if (pool > 32) {
            if (!relu) return PN2_EUNSUP;
            hipError_t e = hipMemsetAsync(y, 0, sizeof(float) * (size_t)(rows / pool) * cout, st);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (cout % 128 == 0) return launch_linear<4>(rows, cin, cout, x, w, bias, relu, pool, y, st);
    if (cout % 64 == 0) return launch_linear<2>(rows, cin, cout, x, w, bias, relu, pool, y, st);
    return launch_linear<1>(rows, cin, cout, x, w, bias, relu, pool, y, st);
}

// internal helper (exported for the host package's unfused SA path and for tests)
extern "C" int pn2_sa_group_concat(int b, int n, int m, int nsample, int c, const float* xyz,
                                   const float* new_xyz, const float* points, const int* idx,
                                   float* out, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || c < 0) return PN2_EINVAL;
    if (!xyz || !new_xyz || !idx || !out || (c > 0 && !points)) return PN2_ENULL;
    const unsigned long long total = (unsigned long long)m * nsample * (3 + c);
    if (total > 0xffffffffull || b > 65535) return PN2_ERANGE;
    unsigned long long g = (total + 255) / 256;
    unsigned long long cap = (256ull * 8 + b - 1) / b;
    if (g > cap) g = cap;
    dim3 grid((unsigned)g, b);
    sa_group_concat_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
        n, m, nsample, c, xyz, new_xyz, c > 0 ? points : nullptr, idx, out);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
