// NOT COMPILED INTO THE LIBRARY -- kept as the record of a round-2 experiment ("tried and dropped",
// and profiles/r02_fps_experiments.txt).  Spatially sorted, bucket-skipping FPS: bit-exact (passed every FPS parity test,
// incl. oracle/_ref), touches only ~11 % of the points per round, and is SLOWER than the every-point kernel
// (781 ns/round with 4 waves, 642 with 8, 681 with 16 vs 620): a lone wave issues one dependent instruction per 2-4 ns,
// so the ~110 instructions of the skip logic + hierarchical max cost more than the 16-wave brute-force pass they replace.
// It plugged into pn2_sampling.hip between fps_reg_kernel and fps_stream_kernel (helpers: fps_tiekey, wave_imax, fps_dmax...).
int g_fps_variant = 0;  // tuning hook (pn2_debug_set(0, v)), see dispatch_fps

// ---- spatially sorted, bucket-skipping FPS (1024 < n <= 8192) ---------------------------------------------------
// Same exact result as fps_reg_kernel with a fraction of its per-round work.  fps_reg_kernel's round is VALU-issue
// bound (every one of the n points is re-tested against every pick: 91 VALU per wave-round x 16 waves on one CU).
// But a pick can only lower the running minimum td(k) of points closer to it than td(k), i.e. (late in the run)
// of a small neighbourhood.  So the cloud is first sorted along a Morton curve inside the workgroup (16x16x16
// cells over the bounding box, LDS counting sort) and dealt out so that (wave w, register slot i) holds 64
// consecutive sorted points = one spatially compact BUCKET with an exact bounding box kept by lane i of wave w.
// A round then
//   1. every wave tests its PPT bucket boxes against the pick (lanes 0..PPT-1, ~14 VALU) :
//          skip bucket  <=>  lb(pick, box) * (1 - 1e-6) > wmax      (wmax = the wave's current max td >= every td
//                                                                     of the bucket; lb = squared distance to the box)
//      the fp32 distance of any point of the box is >= lb * (1 - 1e-6) (both carry <= a few ulp of relative error),
//      hence > td(k): min(d, td) leaves td(k) untouched -- the skipped work is provably a no-op, the result bit-identical;
//   2. only the touched buckets (wave-uniform branches over the ballot mask) run the distance update; a wave that
//      touched anything recomputes its max + tie-break key, the others re-publish their cached key;
//   3. one ds_max_u64 per wave, one barrier, broadcast reads of the key and of the winner's coordinates -- as before.
// The tie-break (max distance, then k mod 512, then k; tf_sampling.cu:153-170) no longer comes from the thread
// layout: every point carries its own 32-bit key fps_tiekey(k) of its ORIGINAL index.
__device__ __forceinline__ unsigned fps_spread4(unsigned v) {  // b3b2b1b0 -> bits 9,6,3,0
    v = (v | (v << 4)) & 0x0C3u;
    return (v | (v << 2)) & 0x249u;
}
constexpr int kFpsCells = 4096;  // 16 x 16 x 16 Morton-coded cells


template <int NT, int PPT, int MODE>
__global__ void __launch_bounds__(NT)
fps_sorted_kernel(int n, int m, const float* __restrict__ xyz_all, int* __restrict__ out_all,
                  float* __restrict__ new_xyz_all, int dbg) {
    static_assert(PPT <= 32 && NT % 64 == 0 && NT * PPT <= 8192 && kFpsCells % NT == 0, "bucket boxes live in lanes < PPT; u16 permutation");
    constexpr int NW = NT / 64;
    constexpr int EPT = kFpsCells / NT;  // histogram entries per thread in the scan
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: 4 x u64 key slots | float bbw[NW][6] + int wsum[NW] | float4 sxyz[n] | int hist[4096] (later u16 perm[n]) | int picks[m]
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem);
    float* bbw = reinterpret_cast<float*>(smem + 32);
    int* wsum = reinterpret_cast<int*>(bbw + 6 * 16);
    float4* sxyz = reinterpret_cast<float4*>(smem + 32 + (6 * 16 + 16) * 4);
    int* hist = reinterpret_cast<int*>(sxyz + n);
    unsigned short* perm = reinterpret_cast<unsigned short*>(hist);
    int* spick = hist + kFpsCells;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    int* __restrict__ out = out_all + (size_t)blockIdx.x * m;
    float* __restrict__ nxyz = new_xyz_all ? new_xyz_all + (size_t)blockIdx.x * m * 3 : nullptr;

    float px[PPT], py[PPT], pz[PPT];

    // ---- 1. load in original order, cloud copy in LDS, bounding box --------------------------------------------
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + NT * i;
        if (k < n) {
            px[i] = xyz[k * 3 + 0]; py[i] = xyz[k * 3 + 1]; pz[i] = xyz[k * 3 + 2];
            sxyz[k] = make_float4(px[i], py[i], pz[i], 0.f);
            lo[0] = fminf(lo[0], px[i]); hi[0] = fmaxf(hi[0], px[i]);
            lo[1] = fminf(lo[1], py[i]); hi[1] = fmaxf(hi[1], py[i]);
            lo[2] = fminf(lo[2], pz[i]); hi[2] = fmaxf(hi[2], pz[i]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = wave_fmin_all(lo[a]); hi[a] = wave_fmax_all(hi[a]); }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { bbw[wave * 6 + a] = lo[a]; bbw[wave * 6 + 3 + a] = hi[a]; }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) hist[tid + NT * e] = 0;
    if (tid < 4) slots[tid] = 0ull;
    if (tid == 0) spick[0] = 0;  // first pick is index 0 (tf_sampling.cu:122-123)
    __syncthreads();
    float scl[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = bbw[a], h = bbw[3 + a];
        for (int w = 1; w < NW; ++w) { l = fminf(l, bbw[w * 6 + a]); h = fmaxf(h, bbw[w * 6 + 3 + a]); }
        lo[a] = l;
        const float ext = h - l;
        scl[a] = ext > 0.f ? 16.0f / ext : 0.f;  // degenerate axis (or inf/garbage): everything in cell 0
        if (!(scl[a] < 3e38f)) scl[a] = 0.f;
    }
    // ---- 2. counting sort by Morton cell: rank within cell from the LDS atomic, exclusive scan of the histogram ---
    int code[PPT], rnk[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + NT * i;
        code[i] = 0; rnk[i] = 0;
        if (k < n) {
            int c[3];
            const float q[3] = {px[i], py[i], pz[i]};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float f = (q[a] - lo[a]) * scl[a];
                int ci = (int)f;            // only locality depends on it, never the result
                ci = ci < 0 ? 0 : (ci > 15 ? 15 : ci);
                if (!(f == f)) ci = 0;
                c[a] = ci;
            }
            code[i] = (int)(fps_spread4((unsigned)c[0]) | (fps_spread4((unsigned)c[1]) << 1) | (fps_spread4((unsigned)c[2]) << 2));
            rnk[i] = atomicAdd(&hist[code[i]], 1);
        }
    }
    __syncthreads();
    {
        int v[EPT], s = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) { v[e] = hist[tid * EPT + e]; s += v[e]; }
        int inc = s;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int run = base + inc - s;
#pragma unroll
        for (int e = 0; e < EPT; ++e) { hist[tid * EPT + e] = run; run += v[e]; }
    }
    __syncthreads();
    int pos[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) pos[i] = hist[code[i]] + rnk[i];
    __syncthreads();  // hist is dead from here: the permutation aliases it
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + NT * i;
        if (k < n) perm[pos[i]] = (unsigned short)k;
    }
    __syncthreads();
    // ---- 3. deal the sorted cloud: bucket q = 64 consecutive sorted points -> wave q % NW, register slot q / NW
    //         (neighbouring buckets go to different waves: the few buckets a pick touches spread over the SIMDs);
    //         lane i of a wave keeps the exact box of the wave's bucket i.
    //      Every point is ONE 64-bit register pair  (td bits + kFpsBias) : ~tiekey,  read as a double: all patterns
    //      are positive normal doubles, so v_max_f64 orders them exactly like the reference's (max td, then lowest
    //      k mod 512, then lowest k) -- value and tie-break travel through every max in one instruction.
    float bx0 = 3e38f, by0 = 3e38f, bz0 = 3e38f, bx1 = -3e38f, by1 = -3e38f, bz1 = -3e38f;
    double mk[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = (i * NW + wave) * 64 + lane;
        float l0 = 3e38f, l1 = 3e38f, l2 = 3e38f, h0 = -3e38f, h1 = -3e38f, h2 = -3e38f;
        if (p < n) {
            const int k = perm[p];
            const float4 q = sxyz[k];
            px[i] = q.x; py[i] = q.y; pz[i] = q.z;
            mk[i] = __hiloint2double(__float_as_int(1e38f) + kFpsBias, (int)~fps_tiekey(k));  // tf_sampling.cu:124-126
            l0 = h0 = q.x; l1 = h1 = q.y; l2 = h2 = q.z;
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            mk[i] = 0.0;  // no point: below every real entry (their high words are >= kFpsBias)
        }
        l0 = wave_fmin_all(l0); l1 = wave_fmin_all(l1); l2 = wave_fmin_all(l2);
        h0 = wave_fmax_all(h0); h1 = wave_fmax_all(h1); h2 = wave_fmax_all(h2);
        if (lane == i) { bx0 = l0; by0 = l1; bz0 = l2; bx1 = h0; by1 = h1; bz1 = h2; }
    }
    // (an empty bucket keeps the inverted box: its lower bound is +inf, it is never touched)
    constexpr int NG = (PPT + 3) / 4;  // groups of 4 slots with a cached per-lane maximum
    double gm[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        gm[g] = mk[4 * g];
#pragma unroll
        for (int e = 1; e < 4; ++e) if (4 * g + e < PPT) gm[g] = fps_dmax(gm[g], mk[4 * g + e]);
    }

    // ---- 4. rounds ------------------------------------------------------------------------------------------------
    int old = 0, slot = 1;
    int vmax = __float_as_int(1e38f);           // td of the current pick when it was chosen: >= every td in the cloud
    unsigned long long ccomp = 0ull;            // this wave's cached candidate (max of its mk[], as raw bits); 0 = none
    bool first = true;
    for (int j = 1; j < m; ++j) {
        const float4 pk = sxyz[old];
        const float x1 = pk.x, y1 = pk.y, z1 = pk.z;
        // bucket test (lanes >= PPT and empty buckets hold the inverted box -> +inf -> never touched)
        const float ex = fmaxf(fmaxf(bx0 - x1, x1 - bx1), 0.f);
        const float ey = fmaxf(fmaxf(by0 - y1, y1 - by1), 0.f);
        const float ez = fmaxf(fmaxf(bz0 - z1, z1 - bz1), 0.f);
        const float lb = (ex * ex + ey * ey) + ez * ez;
        const bool skip = (__float_as_int(lb * 0.999999f) > vmax) && (lb > 1e-30f);
        unsigned smask = (unsigned)__ballot(!skip && lane < PPT);
        if (dbg == 1 && j > 1) smask = 0u;                       // timing probe: fixed path only
        if (dbg == 2 && j > 1) smask = (wave == (j & (NW - 1))) ? 1u : 0u;  // probe: exactly one wave touches one bucket
        if (dbg == 3 && j > 1) smask = 1u;                       // probe: every wave touches one bucket
        smask = (unsigned)__builtin_amdgcn_readfirstlane((int)smask);  // wave-uniform: the slot branches below are scalar
        if (smask != 0u || first) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const unsigned gmask = (smask >> (4 * g)) & 0xFu;
                if (gmask != 0u) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * g + e;
                        if (i < PPT && (gmask & (1u << e))) {
                            const float d = pn2_sqdist<MODE>(px[i] - x1, py[i] - y1, pz[i] - z1);
                            const int dh = __float_as_int(d) + kFpsBias;   // d >= +0: int order == float order
                            const int oh = __double2hiint(mk[i]);
                            mk[i] = __hiloint2double(dh < oh ? dh : oh, __double2loint(mk[i]));  // min(d, td) :151
                        }
                    }
                    gm[g] = mk[4 * g];
#pragma unroll
                    for (int e = 1; e < 4; ++e) if (4 * g + e < PPT) gm[g] = fps_dmax(gm[g], mk[4 * g + e]);
                }
            }
            double tr[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) tr[g] = gm[g];
#pragma unroll
            for (int w = NG; w > 1; w = (w + 1) / 2) {
#pragma unroll
                for (int g = 0; g < w / 2; ++g) tr[g] = fps_dmax(tr[g], tr[w - 1 - g]);
            }
            const int bh = __double2hiint(tr[0]);
            const unsigned bl = (unsigned)__double2loint(tr[0]);
            const int wh = wave_imax(bh);
            const unsigned long long bal = __ballot(bh == wh);
            unsigned wl;
            if (__popcll(bal) == 1) wl = (unsigned)__builtin_amdgcn_readlane((int)bl, __ffsll((long long)bal) - 1);
            else wl = ~wave_umin_all(bh == wh ? ~bl : 0xFFFFFFFFu);  // equal td across lanes: lowest tie key = largest low word
            ccomp = wh >= kFpsBias ? (((unsigned long long)(unsigned)wh << 32) | wl) : 0ull;
            first = false;
        }
        if (lane == 0 && ccomp != 0ull) {
            const unsigned saddr = (unsigned)(size_t)(&slots[slot]);
            asm volatile("ds_max_u64 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(saddr), "v"(ccomp) : "memory");
        }
        __syncthreads();
        const unsigned long long win = slots[slot];
        const unsigned wlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)win);
        vmax = __builtin_amdgcn_readfirstlane((int)(unsigned)(win >> 32)) - kFpsBias;
        old = fps_untiekey(~wlo);
        const int nxt = slot == 2 ? 0 : slot + 1;
        if (tid == 0) {
            slots[nxt == 2 ? 0 : nxt + 1] = 0ull;
            spick[j] = old;
        }
        slot = nxt;
    }
    __syncthreads();
    for (int jj = tid; jj < m; jj += NT) {
        const int k = spick[jj];
        out[jj] = k;
        if (nxyz) {
            const float4 p = sxyz[k];
            nxyz[jj * 3 + 0] = p.x; nxyz[jj * 3 + 1] = p.y; nxyz[jj * 3 + 2] = p.z;
        }
    }
}

template <int NT, int PPT, int MODE>
int launch_fps_sorted(int b, int n, int m, const float* inp, int* out, float* nxyz, hipStream_t st) {
    const size_t bytes = 32 + (6 * 16 + 16) * 4 + (size_t)n * sizeof(float4) + kFpsCells * sizeof(int) + (size_t)m * sizeof(int);
    auto kern = fps_sorted_kernel<NT, PPT, MODE>;
    static bool attr_set = false;  // per instantiation; benign race (idempotent call)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    kern<<<b, NT, bytes, st>>>(n, m, inp, out, nxyz, g_fps_variant >= 10 ? g_fps_variant / 10 : 0);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}
inline bool fps_sorted_fits(int n, int m) {
    return 32 + (6 * 16 + 16) * 4 + (size_t)n * 16 + kFpsCells * 4 + (size_t)m * 4 <= 160 * 1024;
}

