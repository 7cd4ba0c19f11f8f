// pn2_sa_fused_bf16.hip -- bf16 variant of the fused set-abstraction MLP (BASELINE configs[4]: large-scene
// inference, K = 64 neighbours, bf16 features): gather(idx) -> [xyz - centre | features] -> up to 3 x
// (1x1 conv + bias + ReLU) -> max over the K neighbours, on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).
// Same sub-graph as pn2_sa_fused.hip (util/pointnet_util.py:43-54,150-170, inference BN folded); no
// reference kernel exists for it and the reference has no reduced-precision mode, so the arithmetic is
// DEFINED here (and restated by oracle.sa_module_bf16):
//   * features are given in bf16; xyz[idx] - new_xyz is computed in fp32 and rounded to bf16 (RNE);
//   * weights are rounded to bf16 (RNE) while they are staged into LDS; biases stay fp32;
//   * products are exact, accumulation is fp32 (MFMA);
//   * a hidden activation is bf16(relu(acc + bias)); the output is max_k relu(acc + bias) in fp32.
//
// Mapping (same chaining trick as the fp32 kernel): one wave owns one (b, j) neighbourhood = RT tiles of 32
// rows.  Lane l carries neighbour (l & 31) of every tile and, as MFMA operand, the 8 contraction indices of
// its half-wave.  Hidden layers are computed transposed (A = weights, B = activations): the accumulator of
// lane (n, half) holds channels (r&3)+8(r>>2)+4*half, r = 0..15, so registers 0..7 / 8..15 of the two
// half-waves are exactly the 16 contraction indices of one 32x32x16 step of the next layer -- packing them to
// bf16x8 in-lane IS the next B operand; the weights sit in LDS pre-permuted to that order.  Only the pairing
// of A and B elements matters, so no assumption is made about which k a (half, j) slot "really" is.
// The last layer is computed un-transposed so the K-max is in-lane + one half-wave exchange.  Every weight
// fragment read from LDS (16 B per lane) feeds the MFMAs of all RT row tiles.
#include "pn2_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SaBf16Params {
    int n, m, c, groups, nsample;
    int w[3];
    const float* xyz;
    const float* new_xyz;
    const __bf16* points;
    const int* idx;
    const float* W[3];
    const float* bias[3];
    float* out;
};

__device__ __forceinline__ int bchan(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// input row of W (layer 1) feeding slot j of half-wave h at k-step s; -1 = zero
__device__ __forceinline__ int l1_row(int s, int h, int j, int c) {
    if (s == 0) return (h == 0 && j < 3) ? j : -1;  // [dx dy dz 0 0 0 0 0 | 0 x 8]
    const int ch = 16 * (s - 1) + 8 * h + j;
    return ch < c ? 3 + ch : -1;
}
// input row of W (layer >= 2) feeding slot j of half-wave h at k-step s (s = 2*tile + block of the producer)
__device__ __forceinline__ int ln_row(int s, int h, int j) { return (s >> 1) * 32 + bchan(8 * (s & 1) + j, h); }

template <int NT, bool LAST>
__device__ __forceinline__ void mfma_bf(f32x16 (&acc)[NT], const bf16x8 (&wf)[NT], bf16x8 act) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (LAST) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(act, wf[nt], acc[nt], 0, 0, 0);
        else acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], act, acc[nt], 0, 0, 0);
    }
}

template <int NT>
__device__ __forceinline__ void load_wf(bf16x8 (&wf)[NT], const bf16x8* __restrict__ wl) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wf[nt] = wl[nt * 32];
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
}

// hidden epilogue: h[nt][blk] = bf16x8(relu(acc[nt][8*blk + j] + bias[channel]))  (transposed layout)
template <int NT>
__device__ __forceinline__ void pack_hidden(const f32x16 (&acc)[NT], const float* __restrict__ sbias, int half,
                                            bf16x8 (&h)[NT][2]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * blk + j;
                h[nt][blk][j] = (__bf16)fmaxf(acc[nt][r] + sbias[nt * 32 + bchan(r, half)], 0.f);
            }
}

// dense layer fed from packed hidden activations, for RT row tiles at once
template <int RT, int NTP, int NT, bool LAST>
__device__ __forceinline__ void layer_from_hidden(const bf16x8 (&h)[RT][NTP][2], f32x16 (&acc)[RT][NT],
                                                  const bf16x8* __restrict__ wp, int w, int half, int l31) {
    const bf16x8* __restrict__ wl = wp + half * w + l31;
    bf16x8 wc[NT];
    load_wf<NT>(wc, wl);
#pragma unroll
    for (int s = 0; s < NTP * 2; ++s) {
        bf16x8 wn[NT];
        if (s + 1 < NTP * 2) load_wf<NT>(wn, wl + (s + 1) * 2 * w);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) mfma_bf<NT, LAST>(acc[rt], wc, h[rt][s >> 1][s & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < NTP * 2) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wc[nt] = wn[nt];
        }
    }
}

// last-layer epilogue: max over all RT*32 rows, + bias, relu, 32 floats per tile
template <int RT, int NT>
__device__ __forceinline__ void pool_store(const f32x16 (&acc)[RT][NT], const float* __restrict__ sbias,
                                           float* __restrict__ orow, int half, int l31) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float v = acc[0][nt][0];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) v = fmaxf(v, acc[rt][nt][r]);
        v = fmaxf(v, __shfl_xor(v, 32));
        v = fmaxf(v + sbias[nt * 32 + l31], 0.f);  // max_i relu(x_i + b) == relu(max_i x_i + b)
        if (half == 0) orow[nt * 32 + l31] = v;
    }
}

template <int L, int NT1, int NT2, int NT3, int RT, int NW>
__global__ void __launch_bounds__(NW * 64, 2)
sa_fused_bf16_kernel(SaBf16Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int W1 = NT1 * 32, W2 = NT2 * 32, W3 = NT3 * 32;
    constexpr int NTH = NW * 64;
    const int c = p.c;
    const int steps1 = 1 + (c + 15) / 16;

    bf16x8* wp1 = reinterpret_cast<bf16x8*>(smem_raw);
    bf16x8* wp2 = wp1 + steps1 * 2 * W1;
    bf16x8* wp3 = wp2 + (L >= 2 ? (W1 / 16) * 2 * W2 : 0);
    float* sb1 = reinterpret_cast<float*>(wp3 + (L >= 3 ? (W2 / 16) * 2 * W3 : 0));
    float* sb2 = sb1 + W1;
    float* sb3 = sb2 + (L >= 2 ? W2 : 0);

    // ---- weight staging: fp32 global -> bf16 (RNE) -> LDS, pre-permuted [k-step][half][out channel] x 8 ----
    auto stage = [&](bf16x8* dst, int steps, int wd, const float* __restrict__ src, auto row_of) {
        const int count = steps * 2 * wd;
        for (int e = tid; e < count; e += NTH) {
            const int mcol = e % wd, sh = e / wd;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = row_of(sh >> 1, sh & 1, j);
                v[j] = (__bf16)(row >= 0 ? src[(size_t)row * wd + mcol] : 0.f);
            }
            dst[e] = v;
        }
    };
    stage(wp1, steps1, W1, p.W[0], [&](int s, int h, int j) { return l1_row(s, h, j, c); });
    for (int e = tid; e < W1; e += NTH) sb1[e] = p.bias[0][e];
    if constexpr (L >= 2) {
        stage(wp2, W1 / 16, W2, p.W[1], [&](int s, int h, int j) { return ln_row(s, h, j); });
        for (int e = tid; e < W2; e += NTH) sb2[e] = p.bias[1][e];
    }
    if constexpr (L >= 3) {
        stage(wp3, W2 / 16, W3, p.W[2], [&](int s, int h, int j) { return ln_row(s, h, j); });
        for (int e = tid; e < W3; e += NTH) sb3[e] = p.bias[2][e];
    }
    __syncthreads();

    constexpr int WOUT = L == 1 ? W1 : (L == 2 ? W2 : W3);
    constexpr bool LAST1 = (L == 1);
    const int K = RT * 32;
    for (int g = blockIdx.x * NW + wave; g < p.groups; g += gridDim.x * NW) {
        const int bi = g / p.m;
        size_t prow[RT];
        bf16x8 xb[RT];  // k-step 0 operand: [dx dy dz 0 0 0 0 0] in half-wave 0, zeros in half-wave 1
        {
            const float cxv = p.new_xyz[(size_t)g * 3 + 0];
            const float cyv = p.new_xyz[(size_t)g * 3 + 1];
            const float czv = p.new_xyz[(size_t)g * 3 + 2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int ii = p.idx[(size_t)g * K + rt * 32 + l31];
                prow[rt] = (size_t)bi * p.n + ii;
                const float rx = p.xyz[prow[rt] * 3 + 0] - cxv;  // grouped_xyz -= tile(new_xyz) :44-46 (fp32)
                const float ry = p.xyz[prow[rt] * 3 + 1] - cyv;
                const float rz = p.xyz[prow[rt] * 3 + 2] - czv;
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
                if (half == 0) { v[0] = (__bf16)rx; v[1] = (__bf16)ry; v[2] = (__bf16)rz; }
                xb[rt] = v;
            }
        }
        f32x16 a1[RT][NT1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) zero_acc<NT1>(a1[rt]);
        const bf16x8* __restrict__ w1l = wp1 + half * W1 + l31;
        {
            bf16x8 wf[NT1];
            load_wf<NT1>(wf, w1l);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) mfma_bf<NT1, LAST1>(a1[rt], wf, xb[rt]);
        }
        {
            const int nt16 = steps1 - 1;  // feature k-steps (16 channels each; c % 16 == 0 checked by the host)
            const bf16x8* fp[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) fp[rt] = reinterpret_cast<const bf16x8*>(p.points + prow[rt] * c) + half;
            bf16x8 cur[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) cur[rt] = fp[rt][0];
            bf16x8 wc[NT1];
            load_wf<NT1>(wc, w1l + 2 * W1);
            for (int t = 0; t < nt16; ++t) {
                const int tn = (t + 1 < nt16 ? t + 1 : t);
                bf16x8 nxt[RT], wn[NT1];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) nxt[rt] = fp[rt][tn * 2];  // unconditional (clamped): counted vmcnt
                load_wf<NT1>(wn, w1l + (1 + tn) * 2 * W1);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) mfma_bf<NT1, LAST1>(a1[rt], wc, cur[rt]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) cur[rt] = nxt[rt];
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt) wc[nt] = wn[nt];
            }
        }
        float* __restrict__ orow = p.out + (size_t)g * WOUT;
        if constexpr (L == 1) {
            pool_store<RT, NT1>(a1, sb1, orow, half, l31);
        } else {
            bf16x8 h1[RT][NT1][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) pack_hidden<NT1>(a1[rt], sb1, half, h1[rt]);
            f32x16 a2[RT][NT2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) zero_acc<NT2>(a2[rt]);
            layer_from_hidden<RT, NT1, NT2, L == 2>(h1, a2, wp2, W2, half, l31);
            if constexpr (L == 2) {
                pool_store<RT, NT2>(a2, sb2, orow, half, l31);
            } else {
                bf16x8 h2[RT][NT2][2];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) pack_hidden<NT2>(a2[rt], sb2, half, h2[rt]);
                f32x16 a3[RT][NT3];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) zero_acc<NT3>(a3[rt]);
                layer_from_hidden<RT, NT2, NT3, true>(h2, a3, wp3, W3, half, l31);
                pool_store<RT, NT3>(a3, sb3, orow, half, l31);
            }
        }
    }
}

template <int L, int NT1, int NT2, int NT3, int RT>
int launch_bf16(const SaBf16Params& p, hipStream_t st) {
    constexpr int W1 = NT1 * 32, W2 = NT2 * 32, W3 = NT3 * 32;
    const int steps1 = 1 + (p.c + 15) / 16;
    size_t bytes = (size_t)steps1 * 2 * W1 * 16 + W1 * 4;
    if (L >= 2) bytes += (size_t)(W1 / 16) * 2 * W2 * 16 + W2 * 4;
    if (L >= 3) bytes += (size_t)(W2 / 16) * 2 * W3 * 16 + W3 * 4;
    if (bytes > 150 * 1024) return PN2_EUNSUP;
    constexpr int NW = 8;
    auto kern = sa_fused_bf16_kernel<L, NT1, NT2, NT3, RT, NW>;
    static bool attr_set = false;  // per instantiation; benign race (idempotent call)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    int grid = bytes > 78 * 1024 ? 256 : 512;
    const int need = (p.groups + NW - 1) / NW;
    if (grid > need) grid = need;
    kern<<<grid, NW * 64, bytes, st>>>(p);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

}  // namespace

// See include/pn2_abi.h.  points_bf16: (b,n,c) bfloat16 (upper 16 bits of an IEEE fp32), 16-byte aligned rows.
extern "C" int pn2_sa_mlp_max_fused_bf16(int b, int n, int m, int nsample, int c, const float* xyz,
                                         const float* new_xyz, const void* points_bf16, const int* idx,
                                         int nlayers, const int* widths, const float* const* w,
                                         const float* const* bias, float* out, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || c <= 0 || nlayers <= 0) return PN2_EINVAL;
    if (!xyz || !new_xyz || !idx || !widths || !w || !bias || !out || !points_bf16) return PN2_ENULL;
    if ((nsample != 32 && nsample != 64) || nlayers > 3 || c % 16 != 0 || ((uintptr_t)points_bf16 % 16) != 0)
        return PN2_EUNSUP;
    if ((long long)b * m > 0x7fffffffLL / 64) return PN2_ERANGE;
    SaBf16Params p{};
    p.n = n; p.m = m; p.c = c; p.groups = b * m; p.nsample = nsample;
    p.xyz = xyz; p.new_xyz = new_xyz; p.points = static_cast<const __bf16*>(points_bf16); p.idx = idx; p.out = out;
    int nt[3] = {0, 0, 0};
    for (int l = 0; l < nlayers; ++l) {
        if (widths[l] <= 0 || widths[l] % 32 != 0 || widths[l] > 128) return PN2_EUNSUP;
        if (!w[l] || !bias[l]) return PN2_ENULL;
        p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l];
        nt[l] = widths[l] / 32;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int key = nlayers * 1000 + nt[0] * 100 + nt[1] * 10 + nt[2];
#define PN2_BF_CASE(L_, A_, B_, C_)                                                              \
    case (L_ * 1000 + A_ * 100 + B_ * 10 + C_):                                                  \
        return nsample == 64 ? launch_bf16<L_, A_, (B_ ? B_ : 1), (C_ ? C_ : 1), 2>(p, st)       \
                             : launch_bf16<L_, A_, (B_ ? B_ : 1), (C_ ? C_ : 1), 1>(p, st);
    switch (key) {
        PN2_BF_CASE(1, 4, 0, 0)
        PN2_BF_CASE(2, 4, 4, 0)
        PN2_BF_CASE(3, 4, 4, 4)
        PN2_BF_CASE(3, 2, 2, 4)
        PN2_BF_CASE(2, 2, 4, 0)
        default: return PN2_EUNSUP;
    }
#undef PN2_BF_CASE
}
