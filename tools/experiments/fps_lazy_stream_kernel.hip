// NOT COMPILED INTO THE LIBRARY -- record of a round-3 experiment (DESIGN.md section 4 "tried and dropped", profiles/r03_fps_lazy.txt).
// Streaming variant of fps_lazy_kernel (csrc/pn2_sampling.hip): the picking wave posts its picks to LDS in batches while it
// goes on picking and the worker waves apply them concurrently (sequence-numbered mailbox, no flag is ever reset, wave 0
// never waits inside a phase).  Bit-exact (fps tests green on the MI355X), and NOT faster: 352-381 us vs 347 us at
// B=16, n=8192 -> 1024.  The posting (exec-masked LDS writes + a release store every 4 picks) and the polling waves cost
// the picking chain ~100 cycles per pick (407 -> 500), which eats what the hidden update pass saves; the last batch and
// the candidate-list tail (~2000 cycles) stay on the critical path either way.  With the picking wave alone on its SIMD
// (S0FREE) the chain is not faster either: it is not issue contention.
// It plugged into pn2_sampling.hip after fps_lazy_kernel (helpers: fps_lazy_setup, pn2fps::pick_phase with an on_pick callback).

// ---- streaming variant: the picking wave and the worker waves run CONCURRENTLY -----------------------------------------
// In fps_lazy_kernel a phase is serial: [all waves apply the pending picks + list] | barrier | [wave 0 picks, 15 waves
// idle] | barrier, ~55 % of it the picking wave's dependent chain.  Here wave 0 ONLY picks and owns no points; it posts its
// picks to LDS in batches of PP = 64 / PPT while it goes on picking, and the NWK worker waves apply every batch as soon as it
// is posted (same box tests, same arithmetic).  When the list is exhausted the workers have one batch left, build the next
// candidate list, and ONE barrier starts the next phase: a phase costs the picking chain + a short tail.
//   posted (LDS, written by wave 0 only):  pend[i] coordinates of pick i of this phase | post = (phase_seq << 8) | count |
//   gph = td of the phase's first pick (bound of the box tests) | done = phase_seq once the last batch and tau are out.
//   Sequence numbers instead of flags: nothing is ever reset, a worker that is late reads an old sequence number and waits.
// Wave 0 never waits for a worker inside a phase (only at the barrier), so the polling cannot deadlock.
// Same picks as every other FPS kernel of this file (bit-exact; tests/test_ops_gpu.py, tests/test_ref_gpu.py).
// S0FREE: 16 waves, the picking wave alone on its SIMD (wave w sits on SIMD w % 4): waves 4, 8, 12 own nothing and only
// attend the barriers, the 12 waves of the other three SIMDs are the workers.
template <int NWK, int PPT, int MODE, bool S0FREE>
__global__ void __launch_bounds__(S0FREE ? 1024 : 64 * (NWK + 1))
fps_lazy_stream_kernel(int n, int m, const float* __restrict__ xyz_all, int* __restrict__ out_all,
                       float* __restrict__ new_xyz_all, long long* __restrict__ stats) {
#ifdef PN2_TUNING_HOOKS
    const bool do_stats = stats != nullptr && blockIdx.x == 0;
#else
    constexpr bool do_stats = false;
#endif
    long long st_b = 0, st_wait = 0, st_ph = 0, st_tail = 0, st_cons = 0;
    auto now = [&]() -> long long { return do_stats ? (long long)__builtin_readcyclecounter() : 0; };
    const long long st_t00 = now();
    constexpr int NT = S0FREE ? 1024 : 64 * (NWK + 1);
    constexpr int NW = NT / 64;
    static_assert(!S0FREE || NWK == 12, "three SIMDs of four waves");
    constexpr int PP = 64 / PPT;
    constexpr int LPT = (NWK * PPT * 64 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: int ctrl[16] | u64 wcand[16] | u64 cand[64] | float4 pend[64] | float bbw[16][6], int wsum[16] |
    //         float4 sxyz[n] | R: int hist[4096] -> u16 perm[n] -> int spick[m]
    int* ctrl = reinterpret_cast<int*>(smem);  // [0],[1] list counters | [4..7] -, j, tau_hi, - | [8] post | [9] done | [10] gph
    unsigned long long* wcand = reinterpret_cast<unsigned long long*>(smem + 64);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem + 192);
    float4* pend = reinterpret_cast<float4*>(smem + 704);
    float* bbw = reinterpret_cast<float*>(smem + 1728);
    int* wsum = reinterpret_cast<int*>(bbw + 6 * 16);
    float4* sxyz = reinterpret_cast<float4*>(smem + kLazyHead);
    int* hist = reinterpret_cast<int*>(sxyz + n);
    int* spick = hist;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    int* __restrict__ out = out_all + (size_t)blockIdx.x * m;
    float* __restrict__ nxyz = new_xyz_all ? new_xyz_all + (size_t)blockIdx.x * m * 3 : nullptr;

    float px[PPT], py[PPT], pz[PPT];
    double mk[PPT];
    float bx0, by0, bz0, bx1, by1, bz1;
    const int wi = S0FREE ? ((wave & 3) == 0 ? -1 : (wave >> 2) * 3 + (wave & 3) - 1) : wave - 1;  // worker index, -1: no rows
    const bool worker = wi >= 0;
    fps_lazy_setup<NT, LPT, PPT, NWK>(n, wi, xyz, sxyz, hist, bbw, wsum, px, py, pz, mk, bx0, by0, bz0, bx1, by1, bz1);
    if (tid == 0) {
        spick[0] = 0;  // first pick is index 0 (tf_sampling.cu:122-123): posted as the one-pick phase 1
        pend[0] = sxyz[0];
        ctrl[0] = 0; ctrl[1] = 0;
        *reinterpret_cast<int4*>(ctrl + 4) = make_int4(0, 1, __float_as_int(1e38f), 0);
        ctrl[10] = __float_as_int(1e38f);
        ctrl[11] = 0;
        ctrl[8] = (1 << 8) | 1;
        ctrl[9] = 1;
    }
    __syncthreads();
    if (wave == 0) __builtin_amdgcn_s_setprio(3);  // the picking chain is the critical path of every phase

    float eps = 0.2f;  // wave 0 only
    int best = 0;      // workers: high word of the lane's best key after the tail (fallback list)
    unsigned bl = 0u;
    for (int seq = 1, ph = 0;; ++seq, ph ^= 1) {
        const long long t_top = now();
        if (worker) {
            // ---- workers: apply the picks of phase `seq` as they are posted
            int consumed = 0;
            for (int spins = 0;;) {
                // `done` BEFORE `post`: if the phase is seen finished, the post word read after it is the final one
                const int fin = __hip_atomic_load(&ctrl[9], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int post = __hip_atomic_load(&ctrl[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int avail = (post >> 8) == seq ? (post & 255) : 0;
                if (avail > consumed) {
                    const int hi_p = avail - consumed > PP ? consumed + PP : avail;
                    const float G = __int_as_float(__builtin_amdgcn_readfirstlane(ctrl[10]));
                    const float Gs = fmaxf(G * 1.000002f, 1e-30f);  // skip a (row, pick) pair iff lb > Gs
                    const int pi = consumed + (lane & (PP - 1));
                    float qx = __builtin_inff(), qy = qx, qz = qx;
                    if (pi < hi_p) { const float4 q = pend[pi]; qx = q.x; qy = q.y; qz = q.z; }
                    const float ex = fmaxf(fmaxf(bx0 - qx, qx - bx1), 0.f);
                    const float ey = fmaxf(fmaxf(by0 - qy, qy - by1), 0.f);
                    const float ez = fmaxf(fmaxf(bz0 - qz, qz - bz1), 0.f);
                    const float lb = (ex * ex + ey * ey) + ez * ez;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(lb <= Gs);
#pragma unroll
                    for (int i = 0; i < PPT; ++i) {
                        unsigned mi = (unsigned)(mask >> (i * PP)) & (unsigned)((1ull << PP) - 1ull);
                        while (mi) {
                            const int p = __builtin_ctz(mi);  // lane p (row 0's group) holds pick slot p
                            mi &= mi - 1;
                            const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), p));
                            const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), p));
                            const float z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), p));
                            const float d = pn2_sqdist<MODE>(px[i] - x1, py[i] - y1, pz[i] - z1);
                            const int di = __float_as_int(d), oh = __double2hiint(mk[i]);  // d >= +0: int order == float order
                            mk[i] = __hiloint2double(di < oh ? di : oh, __double2loint(mk[i]));  // min(d, td) :151
                        }
                    }
                    consumed = hi_p;
                } else if (fin == seq) {
                    break;
                } else {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) { ctrl[11] = 1; break; }  // ~1 s without a post: never hang the GPU; the result is poisoned below
                }
            }
            // ---- tail: candidate list of the next phase
            const long long t_tail = now();
            st_cons += t_tail - t_top;
            const int tau_hi = __builtin_amdgcn_readfirstlane(ctrl[6]);
            double tr[PPT];
#pragma unroll
            for (int i = 0; i < PPT; ++i) tr[i] = mk[i];
#pragma unroll
            for (int w = PPT; w > 1; w = (w + 1) / 2) {
#pragma unroll
                for (int g = 0; g < w / 2; ++g) tr[g] = fps_dmax(tr[g], tr[w - 1 - g]);
            }
            best = __double2hiint(tr[0]);
            bl = (unsigned)__double2loint(tr[0]);
            if (best >= tau_hi) {  // rare lanes: list every point of mine that reaches tau (one LDS atomic per lane)
                unsigned c = 0;
#pragma unroll
                for (int i = 0; i < PPT; ++i) c += __double2hiint(mk[i]) >= tau_hi ? 1u : 0u;
                unsigned slot_i;
                const unsigned caddr = (unsigned)(size_t)(smem) + 4u * (unsigned)ph;  // &ctrl[ph]
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(slot_i) : "v"(caddr), "v"(c) : "memory");
#pragma unroll
                for (int i = 0; i < PPT; ++i) {
                    if (__double2hiint(mk[i]) >= tau_hi) {
                        if (slot_i < (unsigned)kLazyCap) cand[slot_i] = (unsigned long long)__double_as_longlong(mk[i]);
                        ++slot_i;
                    }
                }
            }
        }
        const long long t_bar0 = now();
        if (worker) st_tail += t_bar0 - t_top;
        __syncthreads();  // the list of phase seq + 1 is complete; every pick so far is applied
        const long long t_bar1 = now();
        st_wait += t_bar1 - t_bar0; st_ph++;
        const int jdone = __builtin_amdgcn_readfirstlane(ctrl[5]);
        if (jdone >= m) break;
        const int tau_hi = __builtin_amdgcn_readfirstlane(ctrl[6]);
        const int cnt = __builtin_amdgcn_readfirstlane(ctrl[ph]);
        const bool use_list = cnt >= 1 && cnt <= kLazyCap;
        if (!use_list) {
            // empty or overflowing list: ONE pick from the per-wave maxima instead (the global maximum is one of them)
            if (lane == 0) wcand[wave] = 0x8000000000000000ull;  // waves without rows: never the maximum
            if (worker) {
                const int wh = wave_imax(best);
                const unsigned long long bal = __ballot(best == wh);
                unsigned wl;
                if (__popcll(bal) == 1) wl = (unsigned)__builtin_amdgcn_readlane((int)bl, __ffsll((long long)bal) - 1);
                else wl = wave_umax_all(best == wh ? bl : 0u);  // equal td across lanes: lowest tie key = largest low word
                if (lane == 0) wcand[wave] = ((unsigned long long)(unsigned)wh << 32) | wl;
            }
            __syncthreads();
        }
        if (wave == 0) {
            // ---- the picking wave: phase seq + 1
            const int nseq = seq + 1;
            unsigned long long key = 0x8000000000000000ull;  // negative high word: never the maximum
            if (use_list) { if (lane < cnt) key = cand[lane]; }
            else if (lane < NW) key = wcand[lane];
            const int chi = (int)(unsigned)(key >> 32);
            const unsigned clo = (unsigned)key;
            int ck = fps_untiekey(~clo);
            if (chi < 0) ck = 0;
            const float4 cq = sxyz[ck];
            if (lane == 0) ctrl[ph ^ 1] = 0;  // the list counter of the NEXT tail (its last readers passed the barrier above)
            int maxp = use_list ? kLazyCap : 1;
            if (maxp > m - jdone) maxp = m - jdone;
            const int lim = use_list ? (tau_hi < 0 ? 0 : tau_hi) : 0;
            int pk_k, g_first, d_last;
            float pk_x, pk_y, pk_z;
            int posted = 0;
            auto post_upto = [&](int np_, int g_first_, float qx_, float qy_, float qz_) {
                if (lane >= posted && lane < np_) pend[lane] = make_float4(qx_, qy_, qz_, 0.f);
                if (posted == 0 && lane == 0) ctrl[10] = g_first_;
                __hip_atomic_store(&ctrl[8], (nseq << 8) | np_, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                posted = np_;
            };
            const int npick = pn2fps::pick_phase<MODE>(chi, clo, cq.x, cq.y, cq.z, lim, maxp, pk_k, pk_x, pk_y, pk_z, g_first, d_last,
                [&](int np_, int, float qx_, float qy_, float qz_, int g_first_) {
                    if ((np_ & (PP - 1)) == 0) post_upto(np_, g_first_, qx_, qy_, qz_);
                });
            if (posted < npick) post_upto(npick, g_first, pk_x, pk_y, pk_z);
            if (lane < npick) spick[jdone + lane] = pk_k;
            int j = jdone + npick;
            eps = pn2fps::adapt_eps(eps, cnt);
            if (npick == 0) j = m;  // unreachable (a non-empty list always yields a pick); never spin
            if (lane == 0) {
                ctrl[5] = j;
                ctrl[6] = __float_as_int(__int_as_float(d_last) * (1.0f - eps));
            }
            __hip_atomic_store(&ctrl[9], nseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            st_b += now() - t_bar1;
        }
    }
    if (do_stats && lane == 0) {
        if (wave == 0) { stats[0] = st_ph; stats[6] = st_b; stats[5] = st_wait; stats[8] = now() - st_t00; stats[4] = 0; stats[7] = 0; }
        stats[16 + wave] = st_tail; stats[48 + wave] = st_cons; stats[32 + wave] = st_wait;
    }
    const bool poisoned = ctrl[11] != 0;  // a worker gave up waiting (cannot happen: wave 0 never waits inside a phase)
    for (int jj = tid; jj < m; jj += NT) {
        const int k = poisoned ? 0 : spick[jj];
        out[jj] = poisoned ? -1 : k;
        if (nxyz) {
            const float4 p = sxyz[k];
            nxyz[jj * 3 + 0] = p.x; nxyz[jj * 3 + 1] = p.y; nxyz[jj * 3 + 2] = p.z;
        }
    }
}

template <int NWK, int PPT, int MODE, bool S0FREE>
int launch_fps_lazy_stream(int b, int n, int m, const float* inp, int* out, float* nxyz, hipStream_t st) {
    const size_t bytes = fps_lazy_bytes(n, m);
    auto kern = fps_lazy_stream_kernel<NWK, PPT, MODE, S0FREE>;
    static bool attr_set = false;  // per instantiation; benign race (idempotent call)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    kern<<<b, S0FREE ? 1024 : 64 * (NWK + 1), bytes, st>>>(n, m, inp, out, nxyz, g_fps_stats);
    PN2_RETURN_IF_LAUNCH_FAILED();
    return PN2_OK;
}

