// pn2_mlp_wide.hip -- the coarse levels' shared MLPs (SA4, FP2, FP3 of the SSG model: 4096 .. 16384 rows, layer widths
// 128 / 256 / 512) as ONE launch per level: up to three 1x1-conv layers (+bias, ReLU; inference BN folded by the host,
// util/tf_util.py:181-203, pointnet_util.py:150-170 and :312-325) chained inside a workgroup, optionally behind the SA
// front end (group_point + centre + concat, pointnet_util.py:39-54) or the FP front end (inverse-distance weights +
// three_interpolate + concat, pointnet_util.py:300-311), and in front of the max over the K = 32 neighbours.
//
// Why not the register-resident chain of pn2_sa_fused.hip: a 256-wide layer needs 256 KB of weights (LDS holds 160 KB) and
// 2 x 128 accumulator registers per 32-row tile.  Why not one pn2_linear per layer (round 1): each of those launches
// drains and refills the chip around ~7-10 us of MFMA work (8 - 55 % of the MFMA peak on these shapes).
//
// Mapping (v_mfma_f32_32x32x2_f32, exact fp32):
//   * a workgroup (4 waves) owns ONE 32-row tile through all layers; the tile's activations live in LDS
//     (act[row][k], row stride K+4 floats: conflict-free 16-byte reads), ping-pong between two buffers;
//   * a layer (K -> N, N in {128, 256, 512}) is cut into N/128 column blocks x 4/(N/128) slices of the contraction: every
//     wave owns a 32 x 128 accumulator block (four 32x32 tiles) for its K slice; the slices are added through LDS in a
//     fixed order (deterministic);
//   * A operand: one ds_read_b128 per 8 k (lane = row l&31, half-wave h reads k0+4h .. k0+4h+3);
//     B operand: straight from global memory / L2 -- lane (l&31, h) loads w[k0+4h+q][cb + 4*(l&31) .. +3] with one
//     16-byte load per k: the four floats are the B values of the wave's four accumulator tiles, i.e. tile t of a wave
//     holds the output columns cb + 4j + t (j = 0..31).  Four groups of 8 k are in flight under the 16 MFMAs of the
//     current one (an L2 round trip is ~1 us, a group's MFMAs ~0.4 us); the first groups of the NEXT layer are issued
//     before the current layer's epilogue (weights do not depend on it);
//   * the weights stream from L2 once per workgroup and layer (256 KB x rows/32: 64 MB per 256-wide layer at 8192 rows,
//     ~5 us of the XCDs' aggregate L2 bandwidth beside ~9 us of MFMA time).
#include "pn2_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWideDepth = 4;  // groups of 8 k in flight per wave (2 / 6 / 8 measured the same or slower, EXPERIMENTS.md)

struct WideParams {
    int rows, cin, x_stride, nlayers, pool, relu_last;
    int w[3];
    const float* x;  // mode 0: (rows, x_stride) input rows
    // mode 1 (SA front end): rows = b*m*32 grouped neighbours, cin = 3 + c
    int n, m, c;
    const float* xyz;
    const float* new_xyz;
    const float* points;
    const int* idx;
    // mode 2 (FP front end): rows = b*n unknown points, cin = c2 + c1; idx / dist (b,n,3), points2 (b,m,c2) = `points`,
    // points1 (b,n,c1) or NULL
    const float* dist;
    const float* points1;
    int c1, c2;
    const float* W[3];
    const float* bias[3];
    float* y;
    // first layer hoisted by linearity (pn2_*_mlp_wide_pre): zpre = (source rows, w[0]) = points2 @ W0[interpolated rows]
    // (FP, b*m rows) or points @ W0[feature rows] (SA, b*n rows).  The layer-0 tile then holds only the skip-link channels
    // (FP: cin = c1) or dx dy dz (SA: cin = 3), and the layer-0 epilogue adds the blended / gathered rows of zpre.
    const float* zpre;
    int rowtab_off;     // floats from the start of LDS: the FP front end's 32 x 8 row table when zpre is set
    int sa[2];          // row strides (floats) of the two activation buffers
    int scratch_off;    // floats from the start of LDS (dedicated scratch)
    int scratch_alias;  // 1: the K-slice partial sums are parked in the layer's (consumed) input buffer
};

// GATHER: the A tile of layer 0 is the SA front end, K order [features (c) | dx dy dz | zero pad to a multiple of 8]; the
// host hands over W0 with its rows in that order.
// INTERP: the A tile of layer 0 is the FP front end [three_interpolate(points2, idx, w(dist)) | points1]
// (pointnet_util.py:300-311), formed with the float expressions of fp_interp_concat_* / the reference ops.
enum { kWidePlain = 0, kWideGather = 1, kWideInterp = 2 };
template <int MODE>
__global__ void __launch_bounds__(256, 2)
mlp_wide_kernel(WideParams p) {
    constexpr bool GATHER = MODE == kWideGather;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int row0 = tile * 32;
    // LDS addresses are formed as smem + integer offset everywhere: selecting between POINTERS to the two buffers at run
    // time degrades them to generic pointers, the operand reads become flat_load, and flat loads (unordered against
    // global loads) force s_waitcnt vmcnt(0) in front of every group of MFMAs -- no load/MFMA overlap at all (measured).
    const int buf_off[2] = {0, 32 * p.sa[0]};
    float* const buf0 = smem;

    // ---- per-layer roles of this wave; the B operands (weights) of a layer go in flight before its input is ready ------
    struct LayerCfg { int N, wnt, ks_n, nq, ks, cb, g0, g1; const float* wcol; };
    auto make_cfg = [&](int l, int Kt) {
        LayerCfg c;
        c.N = p.w[l];
        c.wnt = c.N >> 7;        // column blocks of 128
        c.ks_n = 4 / c.wnt;      // slices of the contraction
        c.nq = wave % c.wnt;
        c.ks = wave / c.wnt;
        c.cb = c.nq * 128;
        const int ngroups = (Kt + 7) >> 3;
        c.g0 = (ngroups * c.ks) / c.ks_n;
        c.g1 = (ngroups * (c.ks + 1)) / c.ks_n;
        c.wcol = p.W[l] + c.cb + 4 * l31;
        return c;
    };
    f32x4 bq[kWideDepth][4];
    // W_l has round8(K_l) rows (the host pads with zero rows; in GATHER mode it also moves the three coordinate rows behind
    // the feature rows): no clamping or row mapping in the loop, four 16-byte loads off one pointer per group
    auto fetch_b = [&](const LayerCfg& c, int g, f32x4 (&b_)[4]) {
        const float* __restrict__ wp = c.wcol + (size_t)(8 * g + 4 * half) * c.N;
#pragma unroll
        for (int q = 0; q < 4; ++q) b_[q] = *reinterpret_cast<const f32x4*>(wp + (size_t)q * c.N);
    };
    auto prefetch_b = [&](const LayerCfg& c) {
#pragma unroll
        for (int u = 0; u < kWideDepth; ++u) fetch_b(c, c.g0 + u < c.g1 ? c.g0 + u : c.g1 - 1, bq[u]);
    };
    const LayerCfg first = make_cfg(0, p.cin);
    prefetch_b(first);

    // ---- stage the input tile ------------------------------------------------------------------------------------
    const int K0 = (p.cin + 7) & ~7;
    // eight independent global loads in flight per thread (a plain load -> LDS store loop pays one L2 round trip per
    // iteration: measured 15 us of a 512-workgroup launch)
    auto stage4 = [&](int count4, int cv, int sa, auto src_of) {  // src_of(r, j) -> address of 4 floats
        for (int base = 0; base < count4; base += 256 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int e = base + u * 256 + tid;
                e = e < count4 ? e : count4 - 1;
                const int r = e / cv, j = e - r * cv;
                v[u] = *reinterpret_cast<const f32x4*>(src_of(r, j));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * 256 + tid;
                if (e < count4) {
                    const int r = e / cv, j = e - r * cv;
                    *reinterpret_cast<f32x4*>(buf0 + r * sa + 4 * j) = v[u];
                }
            }
        }
    };
    if constexpr (GATHER) {
        const int c = p.c, cv = c >> 2;  // c % 4 == 0
        const int bi = tile / p.m;       // one tile = one centre (K = 32)
        const float* __restrict__ pts = p.points + (size_t)bi * p.n * c;
        const int* __restrict__ idx = p.idx + (size_t)tile * 32;
        float* myrow = buf0 + (tid & 31) * p.sa[0];
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (tid < 32) {
            const float* __restrict__ q = p.xyz + ((size_t)bi * p.n + idx[tid]) * 3;
            const float* __restrict__ ctr = p.new_xyz + (size_t)tile * 3;
            dx = q[0] - ctr[0]; dy = q[1] - ctr[1]; dz = q[2] - ctr[2];  // pointnet_util.py:44-46
        }
        if (p.zpre) {  // features hoisted: the tile is [dx dy dz | 0 0 0 0 0]
            if (tid < 32) {
                myrow[0] = dx; myrow[1] = dy; myrow[2] = dz;
                for (int k = 3; k < K0; ++k) myrow[k] = 0.f;
            }
        } else {
        stage4(32 * cv, cv, p.sa[0], [&](int r, int j) { return pts + (size_t)idx[r] * c + 4 * j; });
        if (tid < 32) {
            myrow[c] = dx; myrow[c + 1] = dy; myrow[c + 2] = dz;
            for (int k = c + 3; k < K0; ++k) myrow[k] = 0.f;
        }
        }
    } else if constexpr (MODE == kWideInterp) {
        const int c2 = p.c2, c1 = p.c1, cv2 = c2 >> 2, cv1 = c1 >> 2;  // both % 4 == 0
        const int bi = row0 / p.n;  // p.n % 32 == 0: a tile never straddles two clouds
        const float* __restrict__ p2 = p.points + (size_t)bi * p.m * c2;
        // 8 floats per row: w1 w2 w3 - | i1 i2 i3 -  (buffer 1 is idle until layer 0 ends; with zpre the table is still
        // needed in the layer-0 epilogue, which writes buffer 1: a region of its own)
        float* rowtab = smem + (p.zpre ? p.rowtab_off : buf_off[1]);
        if (tid < 32) {
            const size_t r = (size_t)row0 + tid;
            const float d1 = fmaxf(p.dist[r * 3 + 0], 1e-10f), d2 = fmaxf(p.dist[r * 3 + 1], 1e-10f);
            const float d3 = fmaxf(p.dist[r * 3 + 2], 1e-10f);
            const float r1 = 1.0f / d1, r2 = 1.0f / d2, r3 = 1.0f / d3;  // IEEE divisions (pointnet_util.py:300-303)
            const float norm = (r1 + r2) + r3;
            *reinterpret_cast<f32x4*>(rowtab + tid * 8) = f32x4{r1 / norm, r2 / norm, r3 / norm, 0.f};
            *reinterpret_cast<f32x4*>(rowtab + tid * 8 + 4) =
                f32x4{__int_as_float(p.idx[r * 3 + 0]), __int_as_float(p.idx[r * 3 + 1]), __int_as_float(p.idx[r * 3 + 2]), 0.f};
        }
        __syncthreads();
        const int count4 = p.zpre ? 0 : 32 * cv2;  // interpolated channels hoisted: nothing to stage for them
        for (int base = 0; base < count4; base += 256 * 4) {  // 4 elements x 3 gathers in flight per thread
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int e = base + u * 256 + tid;
                e = e < count4 ? e : count4 - 1;
                const int r = e / cv2, j = e - r * cv2;
                const f32x4 wq = *reinterpret_cast<const f32x4*>(rowtab + r * 8);
                const f32x4 iq = *reinterpret_cast<const f32x4*>(rowtab + r * 8 + 4);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(p2 + (size_t)__float_as_int(iq[0]) * c2 + 4 * j);
                const f32x4 x2 = *reinterpret_cast<const f32x4*>(p2 + (size_t)__float_as_int(iq[1]) * c2 + 4 * j);
                const f32x4 x3 = *reinterpret_cast<const f32x4*>(p2 + (size_t)__float_as_int(iq[2]) * c2 + 4 * j);
                v[u] = (x1 * wq[0] + x2 * wq[1]) + x3 * wq[2];  // tf_interpolate.cpp:322-324, unfused
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = base + u * 256 + tid;
                if (e < count4) {
                    const int r = e / cv2, j = e - r * cv2;
                    *reinterpret_cast<f32x4*>(buf0 + r * p.sa[0] + 4 * j) = v[u];
                }
            }
        }
        if (c1 > 0) {
            const float* __restrict__ p1 = p.points1 + (size_t)row0 * c1;
            for (int e = tid; e < 32 * cv1; e += 256) {
                const int r = e / cv1, j = e - r * cv1;
                *reinterpret_cast<f32x4*>(buf0 + r * p.sa[0] + (p.zpre ? 0 : c2) + 4 * j) = *reinterpret_cast<const f32x4*>(p1 + (size_t)r * c1 + 4 * j);
            }
        }
        if (K0 > p.cin) {
            const int padw = K0 - p.cin;
            for (int e = tid; e < 32 * padw; e += 256) buf0[(e / padw) * p.sa[0] + p.cin + e % padw] = 0.f;
        }
    } else {
        const float* __restrict__ x = p.x;
        if ((p.x_stride & 3) == 0 && (p.cin & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
            const int cv = p.cin >> 2;
            stage4(32 * cv, cv, p.sa[0], [&](int r, int j) {
                const int gr = row0 + r < p.rows ? row0 + r : p.rows - 1;
                return x + (size_t)gr * p.x_stride + 4 * j;
            });
            if (row0 + 32 > p.rows) {  // ragged last tile: rows past the end are zero
                __syncthreads();
                for (int e = tid; e < 32 * p.cin; e += 256) {
                    const int r = e / p.cin;
                    if (row0 + r >= p.rows) buf0[r * p.sa[0] + e - r * p.cin] = 0.f;
                }
            }
        } else {
            const int total = 32 * p.cin;
            for (int base = 0; base < total; base += 256 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int e = base + u * 256 + tid;
                    e = e < total ? e : total - 1;
                    const int r = e / p.cin, k = e - r * p.cin;
                    const int gr = row0 + r < p.rows ? row0 + r : p.rows - 1;
                    v[u] = x[(size_t)gr * p.x_stride + k];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = base + u * 256 + tid;
                    if (e < total) {
                        const int r = e / p.cin, k = e - r * p.cin;
                        buf0[r * p.sa[0] + k] = row0 + r < p.rows ? v[u] : 0.f;
                    }
                }
            }
        }
        if (K0 > p.cin) {
            const int padw = K0 - p.cin;
            for (int e = tid; e < 32 * padw; e += 256) buf0[(e / padw) * p.sa[0] + p.cin + e % padw] = 0.f;
        }
    }
    __syncthreads();

    // (the layer loop below starts with the B operands of layer 0 already in flight: see `prefetch_b` before the staging)
    LayerCfg cur = first;
#pragma unroll 1
    for (int l = 0; l < p.nlayers; ++l) {
        const int N = cur.N, wnt = cur.wnt, ks_n = cur.ks_n, nq = cur.nq, ks = cur.ks, cb = cur.cb, g0 = cur.g0, g1 = cur.g1;
        const int in_off = buf_off[l & 1];
        const int sin = p.sa[l & 1];
        const float* __restrict__ arow = smem + in_off + l31 * sin + 4 * half;

        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        auto contract = [&](const f32x4& a_, const f32x4 (&b_)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[q], b_[q][t], acc[t], 0, 0, 0);
        };
        if (g0 < g1) {
            const int last = g1 - 1;
            f32x4 aq[kWideDepth];
#pragma unroll
            for (int u = 0; u < kWideDepth; ++u) aq[u] = *reinterpret_cast<const f32x4*>(arow + 8 * (g0 + u < g1 ? g0 + u : last));
            int g = g0;
            // steady state: every refill is a real group (no clamping in the loop)
            for (; g + 2 * kWideDepth <= g1; g += kWideDepth) {
#pragma unroll
                for (int u = 0; u < kWideDepth; ++u) {
                    // sched_barrier: keep "16 MFMAs of slot u, then the refill of slot u" in program order.  Left alone the
                    // scheduler sinks all loads of the iteration below its 64 MFMAs and waits for them (vmcnt(0)) at the top
                    // of the next one: no load ever overlaps an MFMA (measured 48 % of the MFMA rate on an idle chip).
                    contract(aq[u], bq[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int gn = g + kWideDepth + u;
                    aq[u] = *reinterpret_cast<const f32x4*>(arow + 8 * gn);
                    fetch_b(cur, gn, bq[u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // drain: the slots hold groups g .. g+D-1 (those < g1 are real), rem more groups follow them
            const int rem = g1 - g - kWideDepth;  // < D
#pragma unroll
            for (int u = 0; u < kWideDepth; ++u) {
                if (g + u < g1) contract(aq[u], bq[u]);
                if (u < rem) {
                    aq[u] = *reinterpret_cast<const f32x4*>(arow + 8 * (g + kWideDepth + u));
                    fetch_b(cur, g + kWideDepth + u, bq[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < kWideDepth - 1; ++u)
                if (u < rem) contract(aq[u], bq[u]);
        }
        // the next layer's weights do not depend on this layer's result: put them in flight under the epilogue
        const bool lastl = l == p.nlayers - 1;
        LayerCfg nxt = cur;
        if (!lastl) {
            nxt = make_cfg(l + 1, N);
            prefetch_b(nxt);
        }

        // ---- add the K slices (fixed order: 2,3 onto 0,1, then 1 onto 0), then the epilogue on the slice-0 waves ------
        if (ks_n > 1) {
            const int sc_off = p.scratch_alias ? in_off : p.scratch_off;
            if (p.scratch_alias) __syncthreads();  // every wave has finished reading the layer's input
            auto park = [&](int block) {
                float* dst = smem + sc_off + block * (64 * 64);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * 64 + lane] = acc[t][r];
            };
            auto add = [&](int block) {
                const float* src = smem + sc_off + block * (64 * 64);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += src[(t * 16 + r) * 64 + lane];
            };
            if (ks_n == 4) {  // one column block, four slices
                if (ks >= 2) park(ks - 2);
                __syncthreads();
                if (ks < 2) add(ks);
                __syncthreads();
                if (ks == 1) park(0);
                __syncthreads();
                if (ks == 0) add(0);
            } else {          // two column blocks, two slices each
                if (ks == 1) park(nq);
                __syncthreads();
                if (ks == 0) add(nq);
            }
        }
        if (ks == 0 && l == 0 && p.zpre) {
            // hoisted part of layer 0: this lane's four output columns (cb + 4 l31 .. +3) of the source rows of zpre
            if constexpr (MODE == kWideInterp) {
                const float* __restrict__ zb = p.zpre + (size_t)(row0 / p.n) * p.m * N + cb + 4 * l31;
                const float* rowtab = smem + p.rowtab_off;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const f32x4 wq = *reinterpret_cast<const f32x4*>(rowtab + row * 8);
                    const f32x4 iq = *reinterpret_cast<const f32x4*>(rowtab + row * 8 + 4);
                    const f32x4 z1 = *reinterpret_cast<const f32x4*>(zb + (size_t)__float_as_int(iq[0]) * N);
                    const f32x4 z2 = *reinterpret_cast<const f32x4*>(zb + (size_t)__float_as_int(iq[1]) * N);
                    const f32x4 z3 = *reinterpret_cast<const f32x4*>(zb + (size_t)__float_as_int(iq[2]) * N);
                    const f32x4 z = (z1 * wq[0] + z2 * wq[1]) + z3 * wq[2];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][r] += z[t];
                }
            } else if constexpr (MODE == kWideGather) {
                const float* __restrict__ zb = p.zpre + (size_t)(tile / p.m) * p.n * N + cb + 4 * l31;
                const int* __restrict__ idx = p.idx + (size_t)tile * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const f32x4 z = *reinterpret_cast<const f32x4*>(zb + (size_t)idx[row] * N);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][r] += z[t];
                }
            }
        }
        if (ks == 0) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias[l] + cb + 4 * l31);
            const bool relu = !lastl || p.relu_last;
            if (!lastl) {
                float* __restrict__ out = smem + buf_off[(l + 1) & 1];
                const int so = p.sa[(l + 1) & 1];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    f32x4 v = {acc[0][r] + bv[0], acc[1][r] + bv[1], acc[2][r] + bv[2], acc[3][r] + bv[3]};
                    v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                    *reinterpret_cast<f32x4*>(out + row * so + cb + 4 * l31) = v;
                }
            } else if (p.pool == 32) {
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float mx = acc[t][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[t][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    mx += bv[t];  // max_i relu(x_i + b) == relu(max_i(x_i) + b)
                    v[t] = relu ? fmaxf(mx, 0.f) : mx;
                }
                if (half == 0) *reinterpret_cast<f32x4*>(p.y + (size_t)tile * N + cb + 4 * l31) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    f32x4 v = {acc[0][r] + bv[0], acc[1][r] + bv[1], acc[2][r] + bv[2], acc[3][r] + bv[3]};
                    if (relu) v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                    if (row < p.rows) *reinterpret_cast<f32x4*>(p.y + (size_t)row * N + cb + 4 * l31) = v;
                }
            }
        }
        if (!lastl) __syncthreads();  // the next layer's input is complete; this layer's input and the scratch are free
        cur = nxt;
    }
}

int launch_mlp_wide(WideParams& p, int mode, hipStream_t st) {
    // LDS: buffer 0 holds the layer-0 and layer-2 inputs, buffer 1 the layer-1 input
    const int k0 = (p.cin + 7) & ~7;
    int ka = k0, kb = 0;
    if (p.nlayers >= 2) kb = p.w[0];
    if (p.nlayers >= 3 && p.w[1] > ka) ka = p.w[1];
    p.sa[0] = ka + 4;
    p.sa[1] = kb + 4 > 8 ? kb + 4 : 8;  // >= 8: the FP front end parks its 32 x 8 row table there
    // K-slice partial sums: two 16 KB blocks; parked in the layer's consumed input buffer when every sliced layer's input
    // is that large, else in a region of their own
    bool sliced = false, alias_ok = true;
    int kin = k0;
    for (int l = 0; l < p.nlayers; ++l) {
        if (p.w[l] < 512) {
            sliced = true;
            if ((size_t)32 * p.sa[l & 1] < (size_t)2 * 64 * 64) alias_ok = false;
        }
        kin = p.w[l];
    }
    (void)kin;
    p.scratch_off = 32 * (p.sa[0] + p.sa[1]);
    p.scratch_alias = sliced && alias_ok;
    p.rowtab_off = p.scratch_off + (sliced && !alias_ok ? 2 * 64 * 64 : 0);
    const size_t lds = sizeof(float) * ((size_t)p.rowtab_off + (p.zpre && mode == kWideInterp ? 32 * 8 : 0));
    if (lds > 160 * 1024) return PN2_EUNSUP;
    const void* kern = mode == kWideGather ? reinterpret_cast<const void*>(mlp_wide_kernel<kWideGather>)
                       : mode == kWideInterp ? reinterpret_cast<const void*>(mlp_wide_kernel<kWideInterp>)
                                             : reinterpret_cast<const void*>(mlp_wide_kernel<kWidePlain>);
    static bool attr_set[3] = {false, false, false};
    if (!attr_set[mode]) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set[mode] = true;
    }
    const int tiles = (p.rows + 31) / 32;
    if (mode == kWideGather) mlp_wide_kernel<kWideGather><<<tiles, 256, lds, st>>>(p);
    else if (mode == kWideInterp) mlp_wide_kernel<kWideInterp><<<tiles, 256, lds, st>>>(p);
    else mlp_wide_kernel<kWidePlain><<<tiles, 256, lds, st>>>(p);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

int check_layers(int nlayers, const int* widths, const float* const* w, const float* const* bias) {
    if (nlayers < 1 || nlayers > 3) return PN2_EUNSUP;
    if (!widths || !w || !bias) return PN2_ENULL;
    for (int l = 0; l < nlayers; ++l) {
        if (widths[l] != 128 && widths[l] != 256 && widths[l] != 512) return PN2_EUNSUP;
        if (!w[l] || !bias[l]) return PN2_ENULL;
        if ((((uintptr_t)w[l]) | ((uintptr_t)bias[l])) & 15) return PN2_EUNSUP;
    }
    return PN2_OK;
}

}  // namespace

extern "C" int pn2_mlp_wide(int rows, int cin, int x_stride, const float* x, int nlayers, const int* widths,
                            const float* const* w, const float* const* bias, int relu_last, int pool, float* y, void* stream) {
    if (rows <= 0 || cin <= 0 || x_stride < cin) return PN2_EINVAL;
    if (!x || !y) return PN2_ENULL;
    int rc = check_layers(nlayers, widths, w, bias);
    if (rc != PN2_OK) return rc;
    if (pool != 0 && pool != 32) return PN2_EUNSUP;
    if (pool == 32 && rows % 32 != 0) return PN2_EINVAL;
    if (((uintptr_t)y) & 15) return PN2_EUNSUP;
    if ((long long)rows + 64 > 0x7fffffffLL) return PN2_ERANGE;
    WideParams p = {};
    p.rows = rows; p.cin = cin; p.x_stride = x_stride; p.nlayers = nlayers; p.pool = pool; p.relu_last = relu_last;
    p.x = x; p.y = y;
    for (int l = 0; l < nlayers; ++l) { p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l]; }
    return launch_mlp_wide(p, kWidePlain, static_cast<hipStream_t>(stream));
}

extern "C" int pn2_sa_mlp_wide(int b, int n, int m, int nsample, int c, const float* xyz, const float* new_xyz,
                               const float* points, const int* idx, int nlayers, const int* widths, const float* const* w,
                               const float* const* bias, int pool, float* y, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0) return PN2_EINVAL;
    if (!xyz || !new_xyz || !points || !idx || !y) return PN2_ENULL;
    if (nsample != 32 || c % 4 != 0 || (((uintptr_t)points | (uintptr_t)y) & 15)) return PN2_EUNSUP;
    int rc = check_layers(nlayers, widths, w, bias);
    if (rc != PN2_OK) return rc;
    if ((long long)b * m * 32 + 64 > 0x7fffffffLL) return PN2_ERANGE;
    WideParams p = {};
    p.rows = b * m * 32; p.cin = 3 + c; p.x_stride = 0; p.nlayers = nlayers; p.pool = pool ? 32 : 0; p.relu_last = 1;
    p.n = n; p.m = m; p.c = c; p.xyz = xyz; p.new_xyz = new_xyz; p.points = points; p.idx = idx; p.y = y;
    for (int l = 0; l < nlayers; ++l) { p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l]; }
    return launch_mlp_wide(p, kWideGather, static_cast<hipStream_t>(stream));
}

extern "C" int pn2_fp_mlp_wide(int b, int n, int m, int c1, int c2, const float* dist, const int* idx, const float* points1,
                               const float* points2, int nlayers, const int* widths, const float* const* w,
                               const float* const* bias, float* y, void* stream) {
    if (b <= 0 || n <= 0 || m < 3 || c2 <= 0 || c1 < 0) return PN2_EINVAL;
    if (!dist || !idx || !points2 || !y || (c1 > 0 && !points1)) return PN2_ENULL;
    if (n % 32 != 0 || c2 % 4 != 0 || c1 % 4 != 0 || (((uintptr_t)points2 | (uintptr_t)points1 | (uintptr_t)y) & 15)) return PN2_EUNSUP;
    int rc = check_layers(nlayers, widths, w, bias);
    if (rc != PN2_OK) return rc;
    if ((long long)b * n + 64 > 0x7fffffffLL) return PN2_ERANGE;
    WideParams p = {};
    p.rows = b * n; p.cin = c2 + c1; p.nlayers = nlayers; p.pool = 0; p.relu_last = 1;
    p.n = n; p.m = m; p.c1 = c1; p.c2 = c2; p.dist = dist; p.idx = idx; p.points = points2; p.points1 = points1; p.y = y;
    for (int l = 0; l < nlayers; ++l) { p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l]; }
    return launch_mlp_wide(p, kWideInterp, static_cast<hipStream_t>(stream));
}

// pn2_fp_mlp_wide with the first layer's product with the interpolated channels hoisted by linearity (see
// pn2_fp_mlp_fused_pre): z = points2 @ W0[:c2] (b*m rows, widths[0] wide) replaces points2; w[0] = the c1 skip-link rows of
// the folded first-layer weight, zero-padded to a multiple of 8 rows.  c1 > 0, c1 % 4 == 0.
extern "C" int pn2_fp_mlp_wide_pre(int b, int n, int m, int c1, const float* dist, const int* idx, const float* points1,
                                   const float* z, int nlayers, const int* widths, const float* const* w,
                                   const float* const* bias, float* y, void* stream) {
    if (b <= 0 || n <= 0 || m < 3 || c1 <= 0) return PN2_EINVAL;
    if (!dist || !idx || !z || !y || !points1) return PN2_ENULL;
    if (n % 32 != 0 || c1 % 4 != 0 || (((uintptr_t)z | (uintptr_t)points1 | (uintptr_t)y) & 15)) return PN2_EUNSUP;
    int rc = check_layers(nlayers, widths, w, bias);
    if (rc != PN2_OK) return rc;
    if ((long long)b * n + 64 > 0x7fffffffLL) return PN2_ERANGE;
    WideParams p = {};
    p.rows = b * n; p.cin = c1; p.nlayers = nlayers; p.pool = 0; p.relu_last = 1;
    p.n = n; p.m = m; p.c1 = c1; p.c2 = 0; p.dist = dist; p.idx = idx; p.points = nullptr; p.points1 = points1; p.y = y; p.zpre = z;
    for (int l = 0; l < nlayers; ++l) { p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l]; }
    return launch_mlp_wide(p, kWideInterp, static_cast<hipStream_t>(stream));
}

// pn2_sa_mlp_wide with the FEATURE part of the first layer hoisted (see pn2_sa_mlp_fused_pre): zf = points @ W0[3:] (b*n
// rows, widths[0] wide) replaces points; w[0] = the 3 xyz rows of the folded first-layer weight + 5 zero rows (8 x widths[0]).
extern "C" int pn2_sa_mlp_wide_pre(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const float* zf,
                                   const int* idx, int nlayers, const int* widths, const float* const* w,
                                   const float* const* bias, int pool, float* y, void* stream) {
    if (b <= 0 || n <= 0 || m <= 0) return PN2_EINVAL;
    if (!xyz || !new_xyz || !zf || !idx || !y) return PN2_ENULL;
    if (nsample != 32 || (((uintptr_t)zf | (uintptr_t)y) & 15)) return PN2_EUNSUP;
    int rc = check_layers(nlayers, widths, w, bias);
    if (rc != PN2_OK) return rc;
    if ((long long)b * m * 32 + 64 > 0x7fffffffLL) return PN2_ERANGE;
    WideParams p = {};
    p.rows = b * m * 32; p.cin = 3; p.x_stride = 0; p.nlayers = nlayers; p.pool = pool ? 32 : 0; p.relu_last = 1;
    p.n = n; p.m = m; p.c = 4; p.xyz = xyz; p.new_xyz = new_xyz; p.points = zf; p.idx = idx; p.y = y; p.zpre = zf;
    for (int l = 0; l < nlayers; ++l) { p.w[l] = widths[l]; p.W[l] = w[l]; p.bias[l] = bias[l]; }
    return launch_mlp_wide(p, kWideGather, static_cast<hipStream_t>(stream));
}
