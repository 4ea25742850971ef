// general attract kernel (detector + cycle-state cache) and its launchers; the lean kernel that takes
// over once attractors are cached lives in bsx_lean.hip
#include "bsx_kernels_common.h"

namespace bsx {

// ------------------------------------------------------------------------------------------------
// attract: attract.py:262-302 semantics (S5-S7, S9, S10) for problems [first, first + count).
//
// Lane life cycle: IDLE -> [WARM: T_p steps under perturbations] -> FAST: step + cycle-cache lookup
// only (taken when cached attractors exist; ends at mu on the first cached cycle state) -> if nothing
// is hit within kFastSteps the search restarts from s(T_p) in BRENT (detector + lookups) -> ADVANCE ->
// MU -> DONE.  With a warm cache almost every lane ends in FAST, which carries no detector state.
//
// Minimum waves per SIMD requested from the register allocator (a 512-thread workgroup is 2 per SIMD).
constexpr int attract_min_waves(int nw) { return nw <= 2 ? 4 : 2; }

template <int NW, int K, int LM>
__global__ __launch_bounds__(kBlock, attract_min_waves(NW)) void k_attract(const AttractParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
    const NetView<NW, K, LM> nv = stage_network<NW, K, LM>(P.net, smem, smem_free);
    const int lane = threadIdx.x & 63;
    const bool has_warmup = (P.sp.tp_origin | P.sp.n_pv) != 0;                      // wave-uniform
    const bool simple_space = bsx::simple_space(P.sp);
    const bool use_cache = P.cc.enabled != 0;
    const uint32_t fast_steps = P.fast_steps;
    const uint32_t service_lanes = kServiceLanes;
    const uint32_t cmask = P.cc.lds_slots - 1;

    // LDS mirror of the cycle-state cache (pointer arithmetic on `smem` keeps the LDS address space:
    // a round trip through an integer would turn every probe into a flat load)
    uint32_t* lc = smem + (((uint32_t)(smem_free - smem) + 3u) & ~3u);
    const uint32_t lc_words = kCacheHeaderWords + P.cc.lds_slots * CacheLayout<NW>::kStride;
    uint32_t cc_seen = 0, cc_states = 0, cc_attr = 0, cc_rounds = 0;       // meaningful in thread 0 only
    uint32_t fm0[NW], fv0[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) { fm0[w] = P.sp.fixmask[w]; fv0[w] = P.sp.fixval[w]; }
    if (use_cache) {
        for (uint32_t i = threadIdx.x; i < lc_words; i += blockDim.x) lc[i] = 0;
        __syncthreads();
        if (threadIdx.x == 0) cache_pull<NW, K>(P.cc, nv, fm0, fv0, lc, cc_seen, cc_states, cc_attr);
        __syncthreads();
    }

    TableSlot<NW> slot;
#pragma unroll
    for (int w = 0; w < NW; ++w) slot.key[w] = 0;
    slot.length = 0; slot.count = 0; slot.sum_l = 0; slot.sum_l2 = 0;

    // A: current state / hare / second pointer, B: tortoise / first pointer, C: min code since the
    // tortoise moved, D: s(T_p) until the cycle closes or a cached state is hit, then the attractor key.
    uint32_t A[NW], B[NW], C[NW], D[NW], fm[NW], fv[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) { A[w] = B[w] = C[w] = D[w] = 0; fm[w] = fm0[w]; fv[w] = fv0[w]; }
    // t: steps since T_p (absolute time during warm-up); cnt: pointer advance / mu; sub: which pointer
    // moves next in PH_MU, the found flag in PH_DONE; pub: the result came from the detector
    uint32_t phase = PH_IDLE, t = 0, tp = P.sp.tp_origin, lam = 0, power = 1, cnt = 0, sub = 0, pub = 0;
    uint32_t exec32 = 0;
    // found iff mu + lambda <= max_t - T_p (S7); Brent needs at most 3x that many steps
    uint32_t cap_rel, brent_limit;
    if (P.cap_rel_inf || P.max_t - tp >= (kStepLimit / 4)) { cap_rel = 0xFFFFFFFFu; brent_limit = kStepLimit; }
    else { cap_rel = (uint32_t)(P.max_t - tp); brent_limit = 3u * cap_rel + 2u; }
    // vis: number of cached attractors that were completely visible when this lane's search started.
    // Only those may end it: a cycle that becomes visible while the lane is already walking on it
    // would be hit at a later state than the entry point (mu too large).
    uint32_t vis = 0;
    uint64_t pv_digits = 0, my_p = 0;

    uint32_t ck[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) ck[w] = 0;
    using acc_t = uint64_t;
    constexpr acc_t kSqGuard = (acc_t)(1ull << 62);
    uint32_t clen = 0, ccnt = 0;
    acc_t csl = 0, csl2 = 0;

    acc_t steps_ref = 0, steps_exec = 0;       // steps_ref: found problems only; the others add max_t each
    uint32_t n_none = 0, n_capfail = 0, limit_hits = 0, n_cached = 0;

#ifdef BSX_DIAG
    unsigned long long dbg_iters = 0, dbg_service = 0;
#endif
    WaveQueue q{0, 0, true};

    // start of the search at s(T_p) = A: snapshot the cache, look s(T_p) itself up, pick the mode
    auto begin_search = [&]() {
        copy_words<NW>(D, A);
        t = 0; pub = 0;
        vis = use_cache ? cache_visible(lc) : 0u;
        uint32_t l2 = 0, k2[NW];
        if (vis && cache_lookup<NW>(lc, cmask, vis, A, l2, k2)) {
            phase = PH_DONE; lam = l2; cnt = 0; sub = (l2 <= cap_rel) ? 1u : 0u;     // mu = 0
            copy_words<NW>(D, k2);
        } else if (vis) {
            phase = PH_FAST;
        } else {
            phase = PH_BRENT; lam = 0; power = 1;
            copy_words<NW>(B, A); copy_words<NW>(C, A);
        }
    };

    for (;;) {
        const uint32_t n_run = __popcll(__ballot(phase >= PH_WARM));
        const uint32_t n_pend = __popcll(__ballot(phase == PH_DONE));
        const uint32_t n_wait = 64u - n_run;                       // pending + idle
        const bool work_left = q.more || q.next < q.end;
        if (n_run == 0 && n_pend == 0 && !work_left) break;
        const bool service = (n_pend && (n_pend >= service_lanes || n_run == 0 || !work_left)) ||
                             (work_left && (n_wait >= service_lanes || n_run == 0));
#ifdef BSX_DIAG
        ++dbg_iters;
        if (service) ++dbg_service;
#endif
        if (service) {
            // ---- resolved problems: statistics, per-problem record, aggregation
            bool flush = false, want_pub = false;
            uint32_t fk[NW], flen = 0, fcnt = 0;
            uint64_t fsl = 0, fsl2 = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) fk[w] = 0;
            if (phase == PH_DONE) {
                phase = PH_IDLE;
                const bool found = sub != 0;
                const uint32_t traj32 = tp + cnt;
                const acc_t traj_l = traj32;
                steps_exec += exec32;
                // reference loop stops at T_p + mu + lambda when found, at max_t otherwise (model.py:201)
                steps_ref += found ? (acc_t)(traj32 + lam) : (acc_t)0;
                n_capfail += found ? 0u : 1u;
                const bool keep = found && (uint64_t)lam <= P.max_len;          // attract.py:294
                want_pub = found && pub && use_cache && lam <= kCycleCacheMaxLen;
                n_cached += (found && !pub) ? 1u : 0u;              // ended on a cached cycle state
                if (P.per_problem) {
                    ProblemRec32 r;
#pragma unroll
                    for (int w = 0; w < kMaxW32; ++w) r.key[w] = 0;
                    if (keep) {
#pragma unroll
                        for (int w = 0; w < NW; ++w) r.key[w] = D[w];
                    }
                    r.length = keep ? lam : 0; r.trajectory_l = keep ? traj32 : 0; r.found = keep; r.pad = 0;
                    P.per_problem[my_p] = r;
                }
                const acc_t sq = (acc_t)traj32 * traj32;
                if (!keep) ++n_none;
                else if (ccnt && eq_words<NW>(ck, D) && csl2 < kSqGuard) { ++ccnt; csl += traj_l; csl2 += sq; }
                else {
                    if (ccnt) { flush = true; copy_words<NW>(fk, ck); flen = clen; fcnt = ccnt; fsl = csl; fsl2 = csl2; }
                    copy_words<NW>(ck, D); clen = lam; ccnt = 1; csl = traj_l; csl2 = sq;
                }
            }
            if (__ballot(flush)) table_merge<NW>(P, slot, lane, flush, fk, flen, fcnt, fsl, fsl2);

            // ---- cycle-state cache upkeep (rare): pull what others published; one lane per newly
            //      detected attractor appends it to the journal
            if (use_cache) {
                if (threadIdx.x == 0 && (++cc_rounds & 31u) == 0)
                    cache_pull<NW, K>(P.cc, nv, fm0, fv0, lc, cc_seen, cc_states, cc_attr);
                uint64_t cand = __ballot(want_pub);
                while (cand) {
                    const int src = __builtin_ctzll(cand);
                    uint32_t k[NW];
#pragma unroll
                    for (int w = 0; w < NW; ++w) k[w] = __builtin_amdgcn_readlane(D[w], src);
                    cand &= ~__ballot(want_pub && eq_words<NW>(D, k));
                    if (lane == src) cache_publish<NW>(P.cc, D, lam);
                }
            }

            // ---- refill idle lanes with the next problems of the wave's chunk
            if (work_left) {
                if (q.next == q.end) {
                    const uint64_t base = grab_chunk(&P.ctr->cursor, P.chunk, lane);
                    if (base >= P.count) q.more = false;
                    else { q.next = base; q.end = (base + P.chunk < P.count) ? base + P.chunk : P.count; }
                }
                const uint64_t avail = q.end - q.next;
                const uint64_t idle = __ballot(phase == PH_IDLE);
                if (avail && idle) {
                    const uint32_t rank = rank_below(idle);
                    if (phase == PH_IDLE && rank < avail) {
                        my_p = q.next + rank;
                        if (P.offsets) my_p = P.offsets[my_p];
                        exec32 = 0;
                        if (P.states) {                 // discovery from explicit states (cube passes): no enumeration
#pragma unroll
                            for (int w = 0; w < NW; ++w) A[w] = P.states[my_p * NW + w];
                        } else if (simple_space) {
                            init_problem_simple<NW>(P.sp, my_p, A);
                        } else {
                            Problem<NW> pr;
                            init_problem<NW>(P.sp, my_p, pr);
                            copy_words<NW>(A, pr.s); copy_words<NW>(fm, pr.fm); copy_words<NW>(fv, pr.fv);
                            pv_digits = pr.pv_digits; tp = pr.tp;
                            if (P.cap_rel_inf || P.max_t - tp >= (kStepLimit / 4)) { cap_rel = 0xFFFFFFFFu; brent_limit = kStepLimit; }
                            else { cap_rel = (uint32_t)(P.max_t - tp); brent_limit = 3u * cap_rel + 2u; }
                        }
                        if (has_warmup && tp > 0) { phase = PH_WARM; t = 0; }
                        else begin_search();
                    }
                    const uint64_t n_idle = (uint64_t)__popcll(idle);
                    q.next += n_idle < avail ? n_idle : avail;
                }
            }
        }       // no `continue`: a single back edge keeps the loop-carried registers in place (no copy chains)

        // ---- one network update per lane per iteration
        const bool step_b = (phase == PH_MU) && sub;
        uint32_t cur[NW], nxt[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) cur[w] = step_b ? B[w] : A[w];
        net_step<NW, K>(nv, cur, fm, fv, nxt);
        exec32 += (phase >= PH_WARM) ? 1u : 0u;

        // ---- FAST / BRENT: the new state is s(T_p + t + 1); is it a known cycle state?
        bool hit = false;
        uint32_t l2 = 0, k2[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) k2[w] = 0;
        if (vis && (phase == PH_FAST || phase == PH_BRENT)) hit = cache_lookup<NW>(lc, cmask, vis, nxt, l2, k2);

        if (phase == PH_FAST) {
            const uint32_t t1 = t + 1;
            const bool ok = hit && t1 <= cap_rel && l2 <= cap_rel - t1;         // mu + lambda <= max_t - T_p
            const bool restart = !hit && t1 >= fast_steps;                       // not on a cached cycle yet
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                A[w] = restart ? D[w] : nxt[w];
                B[w] = restart ? D[w] : B[w];
                C[w] = restart ? D[w] : C[w];
                D[w] = hit ? k2[w] : D[w];
            }
            t = restart ? 0u : t1;
            lam = hit ? l2 : 0u;
            power = 1;
            cnt = ok ? t1 : 0u;
            sub = ok ? 1u : 0u;
            phase = hit ? PH_DONE : (restart ? PH_BRENT : PH_FAST);
        } else if (phase == PH_BRENT) {
            // Brent's detector, written without nested branches (every value is a select)
            const uint32_t t1 = t + 1, lam1 = lam + 1;
            const bool e = !hit && eq_words<NW>(nxt, B);            // hare met the tortoise: cycle closed
            const bool tele = !e && lam1 == power;                  // tortoise jumps to the hare
            const bool lower = tele || lt_words<NW>(nxt, C);
            const bool over = !e && !hit && t1 >= brent_limit;
            const bool too_long = e && lam1 > cap_rel;              // lambda alone exceeds max_t - T_p
            const bool go = e && !too_long;
            const bool ok = hit && t1 <= cap_rel && l2 <= cap_rel - t1;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t key_w = C[w];                        // min code over the cycle when e
                C[w] = lower ? nxt[w] : C[w];
                A[w] = go ? D[w] : nxt[w];
                B[w] = go ? D[w] : (tele ? nxt[w] : B[w]);
                D[w] = hit ? k2[w] : (go ? key_w : D[w]);
            }
            power = tele ? power << 1 : power;
            lam = hit ? l2 : (tele ? 0u : lam1);                    // = lambda when e
            t = t1;
            cnt = ok ? t1 : 0u;
            sub = ok ? 1u : 0u;
            pub = go ? 1u : pub;
            limit_hits += (over && cap_rel == 0xFFFFFFFFu) ? 1u : 0u;
            phase = hit ? PH_DONE : (go ? PH_ADVANCE : ((over || too_long) ? PH_DONE : PH_BRENT));
        } else if (phase >= PH_WARM) {
            // rare phases: warm-up under perturbations, the mu pass after a detection
            if (has_warmup && phase == PH_WARM) {
                ++t;
                apply_perturbations<NW>(P.sp, t, pv_digits, nxt);
                copy_words<NW>(A, nxt);
                if (t == tp) begin_search();
            } else if (phase == PH_ADVANCE) {
                // second pointer y = A moves lambda steps ahead of x = B = s(T_p)
                ++cnt;
                copy_words<NW>(A, nxt);
                if (cnt == lam) {
                    cnt = 0; sub = 0;
                    if (eq_words<NW>(A, B)) { phase = PH_DONE; sub = 1; }                       // mu = 0
                    else if (lam >= cap_rel && cap_rel != 0xFFFFFFFFu) { phase = PH_DONE; }     // mu >= 1: mu + lam > cap
                    else phase = PH_MU;
                }
            } else if (phase == PH_MU) {
                // lagged two-pointer pass, one network update per iteration: y, then x, then compare
                if (!sub) { copy_words<NW>(A, nxt); sub = 1; }
                else {
                    copy_words<NW>(B, nxt); ++cnt;
                    if (eq_words<NW>(A, B)) { phase = PH_DONE; sub = 1; }                       // mu = cnt
                    else if (cnt + lam >= cap_rel && cap_rel != 0xFFFFFFFFu) { phase = PH_DONE; sub = 0; cnt = 0; }
                    else sub = 0;
                }
            }
        }
    }

    // ---- epilogue: lane caches -> wave table -> HBM log; counters
    table_merge<NW>(P, slot, lane, ccnt != 0, ck, clen, ccnt, csl, csl2);
    if (slot.count) log_append<NW, true>(P, slot.key, slot.length, slot.count, slot.sum_l, slot.sum_l2);
#ifdef BSX_DIAG
    if (lane == 0) { atomicAdd(&P.ctr->wave_iters, dbg_iters); atomicAdd(&P.ctr->service_rounds, dbg_service); }
#endif
    wave_atomic_add(&P.ctr->steps_ref, (unsigned long long)steps_ref +
                                           (P.cap_rel_inf ? 0ull : (unsigned long long)n_capfail * P.max_t), lane);
    wave_atomic_add(&P.ctr->steps_exec, (unsigned long long)steps_exec, lane);
    wave_atomic_add(&P.ctr->n_none, (unsigned long long)n_none, lane);
    wave_atomic_add(&P.ctr->step_limit_hits, limit_hits, lane);
    wave_atomic_add(&P.ctr->n_cache_resolved, (unsigned long long)n_cached, lane);
}

template <int NW, int K>
static hipError_t launch_attract_nk(int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    const void* fn;
    BSX_KERNEL_FOR_MODE(k_attract, NW, K, lut_mode, fn);
    if (!fn) return hipErrorInvalidValue;
    void* args[] = {const_cast<AttractParams*>(&P)};
    return hipLaunchKernel(fn, grid, dim3(kBlock), args, shmem, st);
}
template <int NW, int K>
static hipError_t configure_attract_nk(int lut_mode, dim3, size_t shmem, hipStream_t, const int&) {
    const void* fn;
    BSX_KERNEL_FOR_MODE(k_attract, NW, K, lut_mode, fn);
    if (!fn) return hipErrorInvalidValue;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}

hipError_t launch_attract(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    BSX_DISPATCH(launch_attract_nk)
}
// Allow the instantiation used by a network to take `shmem` bytes of dynamic LDS (above 64 KiB this
// must be requested explicitly).
hipError_t configure_attract(int nw, int k, int lut_mode, size_t shmem) {
    const dim3 grid(1);
    const hipStream_t st = nullptr;
    const int P = 0;
    BSX_DISPATCH(configure_attract_nk)
}

}  // namespace bsx
