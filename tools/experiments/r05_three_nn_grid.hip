// pn2_three_nn_grid.hip -- three_nn (tf_interpolate.cpp:213-243: exact float64 3-NN, squared L2 ascending, ties -> lowest index)
// through a uniform grid over the KNOWN points, for the shapes where the all-pairs kernel (pn2_three_nn.h) is the larger cost:
// many queries against a few hundred to a couple of thousand known points (FP4 of semantic.json: 8192 queries x 1024 samples per
// cloud).  r05.
//
// Every workgroup bins the cloud's m known points into a grid in LDS (cells ~1.2 x the mean spacing wide: ~1.7 points per cell
// of a volume-filling cloud, more where the points concentrate) and keeps them there in cell order.  A lane owns a query: it
// walks the 3 x 3 x 3 cells around the query's cell -- 9 runs of x-contiguous cells, its own row first -- and keeps the kTgKeep
// candidates with the smallest float32 squared distance (a branch-free sorted insertion: the lanes of a wave walk different
// cells, a rarely-taken slow path would be taken by SOME lane on every trip).  Then the kept candidates get their exact
// float64 distance from the raw coordinates with the reference's operation order ((dx*dx + dy*dy) + dz*dz, contraction off) and
// are ranked by (distance, index): exactly what the all-pairs scan with a strict '<' yields, PROVIDED that
//   (a) no discarded candidate can be among the three: the worst kept float32 distance is above (1 + 2^-19) x the third exact
//       distance (float32 distances are the true ones within 6 * 2^-24 relative) -- or nothing was discarded;
//   (b) no point outside the searched block can: the third exact distance is STRICTLY below the squared distance from the
//       query to the nearest face of the block that has cells behind it (faces shrunk by 1e-4 cell widths for the rounding
//       of the binning).
// A query that cannot show both (near-ties beyond the kept list: lattices, duplicated points; sparse corners; fewer than three
// points around) is redone by its whole wave over all m points in float64.  No approximation anywhere: every output is the
// all-pairs kernel's, bit for bit (tests/test_ops_gpu.py holds both to the oracle and to each other).
#include "pn2_common.h"

namespace {

constexpr int kTgThreads = 512;
constexpr int kTgWaves = kTgThreads / 64;
constexpr int kTgMaxM = 2048;    // known points per cloud (LDS: 14 bytes each)
constexpr int kTgDim = 16;       // cells per axis at most
constexpr int kTgCells = kTgDim * kTgDim * kTgDim;

struct TgGrid {
    float lo[3], inv_h[3];
    double dlo[3], dh[3];
    int dim[3];
};

__device__ __forceinline__ int tg_cell1(float x, float lo, float inv_h, int dim) {
    int c = (int)floorf((x - lo) * inv_h);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

// (d, k) precedes (e, j): smaller distance, ties -> lower index
__device__ __forceinline__ bool tg_before(double d, int k, double e, int j) { return d < e || (d == e && k < j); }

// insert (d, k) into the sorted triple
__device__ __forceinline__ void tg_insert(double d, int k, double& d1, double& d2, double& d3, int& i1, int& i2, int& i3) {
    if (tg_before(d, k, d3, i3)) {
        if (tg_before(d, k, d2, i2)) {
            d3 = d2; i3 = i2;
            if (tg_before(d, k, d1, i1)) { d2 = d1; i2 = i1; d1 = d; i1 = k; }
            else { d2 = d; i2 = k; }
        } else { d3 = d; i3 = k; }
    }
}

__device__ __forceinline__ double tg_shfl_xor(double v, int o) {
    const int lo = __shfl_xor(__double2loint(v), o), hi = __shfl_xor(__double2hiint(v), o);
    return __hiloint2double(hi, lo);
}

constexpr int kTgKeep = 5;  // candidates kept per query by float32 distance (3 + 2 spares for near-ties)

// exact float64 visit of candidate p (fallback scan): insert into the sorted triple
struct TgState {
    double d1, d2, d3;
    int i1, i2, i3;
};
__device__ __forceinline__ void tg_visit(TgState& t, float fx, float fy, float fz, const float4* __restrict__ kp, int p) {
    const float4 c = kp[p];  // (x, y, z, original index as bits)
    const double dx = (double)fx - (double)c.x, dy = (double)fy - (double)c.y, dz = (double)fz - (double)c.z;
    const double d = (dx * dx + dy * dy) + dz * dz;  // contraction is off: the oracle's operation order
    tg_insert(d, __float_as_int(c.w), t.d1, t.d2, t.d3, t.i1, t.i2, t.i3);
}

// branch-free insertion of (s, p) into the ascending list sk[] / pk[] (the largest entry falls out)
__device__ __forceinline__ void tg_keep(float (&sk)[kTgKeep], int (&pk)[kTgKeep], float s, int p) {
#pragma unroll
    for (int j = 0; j < kTgKeep; ++j) {
        const bool lt = s < sk[j];
        const float ts = lt ? sk[j] : s;
        const int tp = lt ? pk[j] : p;
        sk[j] = lt ? s : sk[j];
        pk[j] = lt ? p : pk[j];
        s = ts; p = tp;
    }
}

__global__ void __launch_bounds__(kTgThreads)
three_nn_grid_kernel(int n, int m, const float* __restrict__ xyz1_all, int ld1, const float* __restrict__ xyz2_all,
                     float* __restrict__ dist_all, int* __restrict__ idx_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: TgGrid | float4 kp[m] (x, y, z, index; cell order: ONE 16-byte LDS read per candidate -- the lanes of a wave walk
    //         different cells, three scattered 4-byte reads per candidate made the LDS pipe the bottleneck) |
    //         int ccount[kTgCells + 1] | float red[kTgWaves][6] | u16 cstart[kTgCells + 2] | fallback scratch
    TgGrid* grid = reinterpret_cast<TgGrid*>(smem);
    float4* kp = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(smem) + ((sizeof(TgGrid) + 15) & ~(size_t)15));
    int* ccount = reinterpret_cast<int*>(kp + m);
    float* red = reinterpret_cast<float*>(ccount + kTgCells + 1);
    unsigned short* cstart = reinterpret_cast<unsigned short*>(red + kTgWaves * 6 + 2);
    int* fb_list = reinterpret_cast<int*>(cstart + kTgCells + 2 + 2);            // [kTgThreads + 1]: unfinished queries, count last
    double* fb_d = reinterpret_cast<double*>(fb_list + kTgThreads + 1 + ((kTgThreads + 1) & 1));  // [kTgWaves * 3] per-wave triples
    int* fb_i = reinterpret_cast<int*>(fb_d + kTgWaves * 3);                       // [kTgWaves * 3]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bi = blockIdx.y;
    const float* __restrict__ xyz2 = xyz2_all + (size_t)bi * m * 3;
    const float* __restrict__ xyz1 = xyz1_all + (size_t)bi * n * ld1;

    // the query of this thread: its load is in flight while the grid is built
    const int q = blockIdx.x * kTgThreads + tid;
    const bool qv = q < n;
    const int qc = qv ? q : n - 1;
    const float fx = xyz1[(size_t)qc * ld1 + 0], fy = xyz1[(size_t)qc * ld1 + 1], fz = xyz1[(size_t)qc * ld1 + 2];

    // ---- build: bounding box -> grid -> histogram (the atomic returns the rank inside the cell) -> prefix sum -> scatter -------
    constexpr int PPT = kTgMaxM / kTgThreads;
    float px[PPT], py[PPT], pz[PPT];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = tid + kTgThreads * j;
        const int kc = k < m ? k : m - 1;
        px[j] = xyz2[kc * 3 + 0]; py[j] = xyz2[kc * 3 + 1]; pz[j] = xyz2[kc * 3 + 2];
        mn[0] = fminf(mn[0], px[j]); mx[0] = fmaxf(mx[0], px[j]);
        mn[1] = fminf(mn[1], py[j]); mx[1] = fmaxf(mx[1], py[j]);
        mn[2] = fminf(mn[2], pz[j]); mx[2] = fmaxf(mx[2], pz[j]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[wave * 6 + a] = mn[a]; red[wave * 6 + 3 + a] = mx[a]; }
    }
    for (int c = tid; c <= kTgCells; c += kTgThreads) ccount[c] = 0;
    __syncthreads();
    if (tid == 0) {
        float lo[3], hi[3], ext[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = red[a]; hi[a] = red[3 + a];
#pragma unroll
            for (int w = 1; w < kTgWaves; ++w) { lo[a] = fminf(lo[a], red[w * 6 + a]); hi[a] = fmaxf(hi[a], red[w * 6 + 3 + a]); }
            ext[a] = hi[a] - lo[a];
        }
        // target cell width: 1.2 x the mean spacing of the points over the axes that have an extent
        // (the product of the extents relative to the largest one: no overflow for any finite cloud)
        const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
        float rel = 1.0f;
        int deff = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) if (ext[a] > 1e-3f * emax && ext[a] > 0.f) { rel *= ext[a] / emax; ++deff; }
        const float per = rel / (float)m;
        float h0 = 1.2f * emax * (deff == 3 ? cbrtf(per) : (deff == 2 ? sqrtf(per) : per));
        if (deff == 0 || !(h0 > 0.f) || !isfinite(h0)) h0 = 1.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int d = 1;
            if (ext[a] > 1e-3f * emax && ext[a] > 0.f) {
                const float qd = ext[a] / h0;
                d = qd >= (float)kTgDim ? kTgDim : (int)ceilf(qd);
                d = d < 1 ? 1 : d;
            }
            // cells slightly wider than ext / d so that the largest coordinate still falls into cell d - 1 by arithmetic
            const float h = d > 1 ? (ext[a] / (float)d) * 1.0001f : 0.f;
            grid->lo[a] = lo[a];
            grid->inv_h[a] = d > 1 ? 1.0f / h : 0.f;
            grid->dlo[a] = (double)lo[a];
            grid->dh[a] = (double)h;
            grid->dim[a] = d;
        }
    }
    __syncthreads();
    const TgGrid G = *grid;
    int cellid[PPT], rank[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = tid + kTgThreads * j;
        cellid[j] = (tg_cell1(pz[j], G.lo[2], G.inv_h[2], G.dim[2]) * G.dim[1] + tg_cell1(py[j], G.lo[1], G.inv_h[1], G.dim[1])) * G.dim[0] +
                    tg_cell1(px[j], G.lo[0], G.inv_h[0], G.dim[0]);
        rank[j] = k < m ? atomicAdd(&ccount[cellid[j]], 1) : 0;
    }
    __syncthreads();
    // exclusive prefix sum of the kTgCells counts: 8 consecutive cells per thread, then a scan of the 512 partial sums
    {
        constexpr int CPT = kTgCells / kTgThreads;
        int loc[CPT], sum = 0;
#pragma unroll
        for (int c = 0; c < CPT; ++c) { loc[c] = ccount[tid * CPT + c]; sum += loc[c]; }
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        int* wsum = reinterpret_cast<int*>(red);  // (red was consumed before the barrier above)
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int run = base + inc - sum;
#pragma unroll
        for (int c = 0; c < CPT; ++c) { cstart[tid * CPT + c] = (unsigned short)run; run += loc[c]; }
        if (tid == kTgThreads - 1) { cstart[kTgCells] = (unsigned short)run; cstart[kTgCells + 1] = (unsigned short)run; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = tid + kTgThreads * j;
        if (k < m) {
            const int pos = (int)cstart[cellid[j]] + rank[j];
            kp[pos] = make_float4(px[j], py[j], pz[j], __int_as_float(k));
        }
    }
    __syncthreads();

    // ---- the query ------------------------------------------------------------------------------------------------------------
    const double qx = (double)fx, qy = (double)fy, qz = (double)fz;
    const int cx = tg_cell1(fx, G.lo[0], G.inv_h[0], G.dim[0]);
    const int cy = tg_cell1(fy, G.lo[1], G.inv_h[1], G.dim[1]);
    const int cz = tg_cell1(fz, G.lo[2], G.inv_h[2], G.dim[2]);
    const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < G.dim[0] ? cx + 1 : G.dim[0] - 1;
    const int y0 = cy > 0 ? cy - 1 : 0, y1 = cy + 1 < G.dim[1] ? cy + 1 : G.dim[1] - 1;
    const int z0 = cz > 0 ? cz - 1 : 0, z1 = cz + 1 < G.dim[2] ? cz + 1 : G.dim[2] - 1;
    float sk[kTgKeep];
    int pk[kTgKeep];
#pragma unroll
    for (int j = 0; j < kTgKeep; ++j) { sk[j] = INFINITY; pk[j] = -1; }
    int ncand = 0;  // candidates met (all of them are kept while ncand <= kTgKeep)
#ifndef TG_NO_MAIN
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
        // the query's own row first: its candidates are the near ones, the list settles early
        const int dzr = r < 3 ? 0 : (r < 6 ? -1 : 1), dyr = (r % 3 == 0) ? 0 : (r % 3 == 1 ? -1 : 1);
        const int z = cz + dzr, y = cy + dyr;
        int s0 = 0, e0 = 0;
        if (z >= 0 && z < G.dim[2] && y >= 0 && y < G.dim[1]) {
            const int rowc = (z * G.dim[1] + y) * G.dim[0];
            s0 = cstart[rowc + x0]; e0 = cstart[rowc + x1 + 1];  // the x-cells of a row are contiguous
        }
        ncand += e0 - s0;
        for (int p = s0; p < e0; p += 2) {  // two candidates per trip: their loads and distances overlap
            const int pb = p + 1 < e0 ? p + 1 : p;
            const float4 ca = kp[p], cb = kp[pb];
            const float aa = fx - ca.x, ab = fy - ca.y, ac = fz - ca.z, ba = fx - cb.x, bb = fy - cb.y, bc = fz - cb.z;
            const float sa = (aa * aa + ab * ab) + ac * ac;
            const float sb = p + 1 < e0 ? (ba * ba + bb * bb) + bc * bc : INFINITY;
            tg_keep(sk, pk, sa, p);
            tg_keep(sk, pk, sb, pb);
        }
    }
#endif
    // exact float64 distances of the kept candidates, ranked by (distance, index)
    TgState t;
    t.d1 = t.d2 = t.d3 = INFINITY;
    t.i1 = t.i2 = t.i3 = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < kTgKeep; ++j)
        if (pk[j] >= 0 && sk[j] < INFINITY) tg_visit(t, fx, fy, fz, kp, pk[j]);
    // (a) nothing discarded could belong to the three: every discarded candidate has a float32 distance >= the worst kept one
    const bool kept_all = ncand <= kTgKeep;
    const bool list_ok = kept_all || ((double)sk[kTgKeep - 1] > t.d3 * 1.0000019 && sk[kTgKeep - 1] < 3.0e38f);
    // nearest face of the searched block that has cells behind it
    double fmin = INFINITY;
    {
        const double m4 = 1e-4;
        double f;
        if (x0 > 0) { f = qx - (G.dlo[0] + (x0 + m4) * G.dh[0]); fmin = f < fmin ? f : fmin; }
        if (x1 < G.dim[0] - 1) { f = (G.dlo[0] + (x1 + 1 - m4) * G.dh[0]) - qx; fmin = f < fmin ? f : fmin; }
        if (y0 > 0) { f = qy - (G.dlo[1] + (y0 + m4) * G.dh[1]); fmin = f < fmin ? f : fmin; }
        if (y1 < G.dim[1] - 1) { f = (G.dlo[1] + (y1 + 1 - m4) * G.dh[1]) - qy; fmin = f < fmin ? f : fmin; }
        if (z0 > 0) { f = qz - (G.dlo[2] + (z0 + m4) * G.dh[2]); fmin = f < fmin ? f : fmin; }
        if (z1 < G.dim[2] - 1) { f = (G.dlo[2] + (z1 + 1 - m4) * G.dh[2]) - qz; fmin = f < fmin ? f : fmin; }
    }
    // final iff three points were found and the third is strictly nearer than anything outside the block can be
    bool done = t.i3 != 0x7fffffff && list_ok && (fmin == INFINITY || (fmin > 0.0 && t.d3 < fmin * fmin));
#ifdef TG_REPORT
    if (!done) t.i1 = -1 - (t.i3 == 0x7fffffff ? 1 : (!list_ok ? 2 : 3));  // diagnosis: why a query was not final
    done = true;
#endif
#ifdef TG_NO_FALLBACK
    done = true;
#endif
    // ---- the WORKGROUP redoes its unfinished queries over all m points in float64 (typically none or one of its 512: a lone
    //      wave doing this made the kernel wait ~10 us for it) ----------------------------------------------------------------------
    if (tid == 0) fb_list[kTgThreads] = 0;
    __syncthreads();
    if (qv && !done) fb_list[atomicAdd(&fb_list[kTgThreads], 1)] = tid;
    __syncthreads();
    const int nfb = fb_list[kTgThreads];
    for (int f = 0; f < nfb; ++f) {
        const int owner = fb_list[f];
        const int oq = blockIdx.x * kTgThreads + owner;
        const float ax = xyz1[(size_t)oq * ld1 + 0], ay = xyz1[(size_t)oq * ld1 + 1], az = xyz1[(size_t)oq * ld1 + 2];
        TgState e;
        e.d1 = e.d2 = e.d3 = INFINITY;
        e.i1 = e.i2 = e.i3 = 0x7fffffff;
        for (int p = tid; p < m; p += kTgThreads) tg_visit(e, ax, ay, az, kp, p);
        // three rounds per wave: the wave's best head is the next answer; the lane that holds it pops it
#pragma unroll
        for (int rnd = 0; rnd < 3; ++rnd) {
            double bd = e.d1;
            int bk = e.i1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double od = tg_shfl_xor(bd, o);
                const int ok = __shfl_xor(bk, o);
                if (tg_before(od, ok, bd, bk)) { bd = od; bk = ok; }
            }
            if (bk == e.i1 && bk != 0x7fffffff) { e.d1 = e.d2; e.i1 = e.i2; e.d2 = e.d3; e.i2 = e.i3; e.d3 = INFINITY; e.i3 = 0x7fffffff; }
            if (lane == 0) { fb_d[wave * 3 + rnd] = bd; fb_i[wave * 3 + rnd] = bk; }
        }
        __syncthreads();
        if (tid == owner) {  // merge the kTgWaves sorted triples
            TgState g;
            g.d1 = g.d2 = g.d3 = INFINITY;
            g.i1 = g.i2 = g.i3 = 0x7fffffff;
            for (int w = 0; w < kTgWaves * 3; ++w)
                if (fb_i[w] != 0x7fffffff) tg_insert(fb_d[w], fb_i[w], g.d1, g.d2, g.d3, g.i1, g.i2, g.i3);
            t = g;
        }
        __syncthreads();
    }
    if (qv) {
        const size_t o = ((size_t)bi * n + q) * 3;
        dist_all[o + 0] = (float)t.d1; dist_all[o + 1] = (float)t.d2; dist_all[o + 2] = (float)t.d3;
        idx_all[o + 0] = t.i1; idx_all[o + 1] = t.i2; idx_all[o + 2] = t.i3;
    }
}

inline size_t tg_lds_bytes(int m) {
    const size_t mp = ((size_t)m + 3) & ~(size_t)3;
    (void)mp;
    return ((sizeof(TgGrid) + 15) & ~(size_t)15) + (size_t)m * 16 + (size_t)(kTgCells + 1) * 4 + (kTgWaves * 6 + 2) * 4 +
           (size_t)(kTgCells + 4) * 2 + (size_t)(kTgThreads + 2) * 4 + (size_t)kTgWaves * 3 * 12 + 64;
}

}  // namespace

// launched by pn2_three_nn(_ld / _kernel) (pn2_interpolate.hip); PN2_EUNSUP when the shape is outside this kernel's range
int pn2_three_nn_grid_launch(int b, int n, int m, const float* xyz1, int ld1, const float* xyz2, float* dist, int* idx,
                             hipStream_t st) {
    if (m < 3 || m > kTgMaxM || b > 65535) return PN2_EUNSUP;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(three_nn_grid_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((n + kTgThreads - 1) / kTgThreads, b);
    three_nn_grid_kernel<<<grid, kTgThreads, tg_lds_bytes(m), st>>>(n, m, xyz1, ld1, xyz2, dist, idx);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

#ifdef TG_STANDALONE
extern "C" int tg_launch(int b, int n, int m, const float* xyz1, int ld1, const float* xyz2, float* dist, int* idx, void* st) {
    return pn2_three_nn_grid_launch(b, n, m, xyz1, ld1, xyz2, dist, idx, static_cast<hipStream_t>(st));
}
#endif
