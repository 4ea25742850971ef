// Lean attract kernel: the hot loop of attract once the cycle-state cache knows the attractors.
#include "bsx_kernels_common.h"

namespace bsx {

// ------------------------------------------------------------------------------------------------
// k_attract_lean: attract.py:262-302 semantics for problems whose trajectory reaches a CACHED attractor
// within P.fast_steps steps after T_p; every other problem is handed to the general kernel
// (k_attract, bsx_attract.hip) through the straggler list, so the pair resolves every problem once.
//
// Preconditions (bsx_run_attract checks them): no fixed-node / perturbation variations, the 'any'
// nodes are nodes 0..a-1 with a <= 64, at most a short uniform warm-up (origin perturbations), at
// least one attractor in the journal, P.count < 2^32.
//
// Why a separate kernel: with a warm cache a trajectory needs nothing but  state -> next state ->
// "is it a cached cycle state?"  until the first hit at t = mu; then key / lambda / trajectory_l are
// known.  There is no detector state (tortoise, min code, powers), the cache mirror is
// read-only for the whole launch, and a result is just (attractor tag, mu):
//   * results of cached attractors 1..kTagAcc are summed per lane in registers by predicated adds,
//     later ones (and register sums about to overflow) by LDS atomics into per-workgroup accumulators;
//     nothing crosses lanes before the kernel ends, when every workgroup writes one log record per
//     attractor; the step counters follow from those sums;
//   * t runs from -T_p: time <= 0 is the warm-up (perturbations applied, hits ignored), so warm-up and
//     search are the same loop body;
//   * resolved lanes wait for a service round (record + refill), entered when P.pad lanes (default
//     kLeanServiceLanes) are free: a round costs about as much as one step, so it is shared.
constexpr uint32_t kLeanServiceLanes = 32;
constexpr uint32_t kRegSumGuard = 0x7FFF0000u;      // per-lane 32-bit sums of trajectory_l^2 stay below 2^32

// (LUT + cache mirror take ~52 KiB of LDS per workgroup for n = 64: 3 workgroups = 6 waves per SIMD)
constexpr int lean_min_waves(int nw) { return nw <= 2 ? 6 : nw == 4 ? 4 : 2; }

// Lane status word: 0 = running; a hit leaves the entry's tag word (attractor number, maybe with the
// continue flag); the two codes below have bit 31 clear and are no attractor numbers.
constexpr uint32_t kResLost = 0x7FFFFFFEu;
constexpr uint32_t kResIdle = 0x7FFFFFFFu;

template <int NW, int K, int LM>
__global__ __launch_bounds__(kBlock, lean_min_waves(NW)) void k_attract_lean(const AttractParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
    const NetView<NW, K, LM> nv = stage_network<NW, K, LM>(P.net, smem, smem_free);
    const int lane = threadIdx.x & 63;
    const uint32_t tp = P.sp.tp_origin;                     // uniform: no variations here
    const bool has_warmup = tp != 0;
    const int32_t fast_steps = (int32_t)P.fast_steps;
    const uint32_t service_lanes = P.pad ? P.pad : kLeanServiceLanes;
    const uint32_t cmask = P.cc.lds_slots - 1;
    constexpr int S = CacheLayout<NW>::kStride;
    constexpr uint32_t kAccs = (uint32_t)kTagAcc + kLdsAcc;          // attractors this kernel resolves

    // LDS: [network tables][cache mirror][per attractor: sum l^2, sum l (u64), count, length (u32), key]
    uint32_t* lc = smem + (((uint32_t)(smem_free - smem) + 3u) & ~3u);
    const uint32_t lc_words = kCacheHeaderWords + P.cc.lds_slots * S;
    unsigned long long* acc_sl2 = reinterpret_cast<unsigned long long*>(lc + ((lc_words + 1u) & ~1u));
    unsigned long long* acc_sl = acc_sl2 + kAccs;
    unsigned int* acc_cnt = reinterpret_cast<unsigned int*>(acc_sl + kAccs);
    uint32_t* lamtab = acc_cnt + kAccs;
    uint32_t* keytab = lamtab + kAccs;
    // sibling merge: per-wave tables -- 64 member-mask accumulators, then 64 one-byte lane ids
    const bool merge = P.merge != 0;                        // uniform
    uint32_t* dd_acc = keytab + kAccs * NW + (uint32_t)(threadIdx.x >> 6) * (64u + kMergeSlots / 4u);      // 256 B + the id bytes per wave
    // the same tables as LDS-address-space pointers (volatile ds_* accesses instead of flat ones)
    typedef volatile uint32_t __attribute__((address_space(3))) lds_vu32;
    typedef volatile uint8_t __attribute__((address_space(3))) lds_vu8;
    lds_vu32* const dd_acc3 = (lds_vu32*)(__attribute__((address_space(3))) uint32_t*)dd_acc;
    lds_vu8* const dd_ids3 = (lds_vu8*)(dd_acc3 + 64);

    uint32_t fm0[NW], fv0[NW];
    uint32_t any_fixed = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { fm0[w] = P.sp.fixmask[w]; fv0[w] = P.sp.fixval[w]; any_fixed |= fm0[w]; }
    const bool has_fixed = any_fixed != 0;                  // uniform
    for (uint32_t i = threadIdx.x; i < lc_words; i += blockDim.x) lc[i] = 0;
    for (uint32_t i = threadIdx.x; i < kAccs; i += blockDim.x) { acc_sl2[i] = 0; acc_sl[i] = 0; acc_cnt[i] = 0; lamtab[i] = 0; }
    dd_acc[lane] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        // only the first kAccs attractors of the journal enter the mirror, so every occupied slot is a
        // hit candidate and the probe needs no visibility compare; the rest become stragglers
        uint32_t seen = 0, n_states = 0, n_attr = 0;
        cache_pull<NW, K>(P.cc, nv, fm0, fv0, lc, seen, n_states, n_attr, kAccs);
    }
    __syncthreads();
    const uint32_t* cbase = lc + kCacheHeaderWords;
    for (uint32_t sl = threadIdx.x; sl < P.cc.lds_slots; sl += blockDim.x) {
        const uint32_t tg = cbase[sl * S + NW] & kTagMask;
        if (tg) {       // every entry of an attractor writes the same values
            lamtab[tg - 1] = cbase[sl * S + NW + 1];
#pragma unroll
            for (int w = 0; w < NW; ++w) keytab[(tg - 1) * NW + w] = cbase[sl * S + NW + 2 + w];
        }
    }
    __syncthreads();

    // found iff mu + lambda <= max_t - T_p (S7)
    const uint32_t cap_rel = (P.cap_rel_inf || P.max_t - tp >= (kStepLimit / 4)) ? 0xFFFFFFFFu : (uint32_t)(P.max_t - tp);

    uint32_t A[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) A[w] = 0;
    int32_t t = 0;                          // time relative to T_p; = mu once the lane has its hit
    // my_p: the lane's problem (offset in the launch); with merging the base of its group of kMergeGroup
    // consecutive problems, and `members` the mask of those whose trajectories are in this lane's state
    uint32_t res = kResIdle, my_p = 0, members = 1;
    uint32_t tcnt[kTagAcc], tsl[kTagAcc], tsl2[kTagAcc];
#pragma unroll
    for (int j = 0; j < kTagAcc; ++j) tcnt[j] = tsl[j] = tsl2[j] = 0;
    // steps of results that bypass the accumulators (lost, not found, longer than max_len); the steps
    // of accumulated results follow from the sums at the end
    unsigned long long extra_ref = 0;
    uint32_t nexec = 0;                     // network updates this lane really executed (a merged lane steps once for all its members)
    uint32_t n_none = 0, n_capfail = 0;
    WaveQueue q{0, 0, true};
    uint32_t since_service = 0;
#ifdef BSX_DIAG
    unsigned long long dbg_iters = 0, dbg_service = 0;
#endif

    // probe of state s: is it a cached cycle state?  -> the entry's tag word (0 = no)
    uint32_t last_hash = 0;                 // full hash of the state probed last (reused by the merge)
    auto probe = [&](const uint32_t (&s)[NW]) -> uint32_t {
        const uint32_t hfull = hash_state<NW>(s);
        last_hash = hfull;
        uint32_t h = hfull & cmask;
        const uint32_t* e = cbase + h * S;
        uint32_t et, d;
        // (the asm keeps the unused last word alive: a 16-byte ds_read_b128 takes 4 LDS cycles, the
        //  12-byte ds_read_b96 the compiler would narrow it to takes 8)
        if constexpr (NW == 1) {
            uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(e, 16));
            asm volatile("" : "+v"(v.z), "+v"(v.w));
            d = v.x ^ s[0]; et = v.y;
        } else if constexpr (NW == 2) {
            uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(e, 16));
            asm volatile("" : "+v"(v.w));
            d = (v.x ^ s[0]) | (v.y ^ s[1]); et = v.z;          // one 16-byte read, no short circuit
        } else {
            d = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) d |= e[w] ^ s[w];
            et = e[NW];
        }
        bool hit = (d == 0) & (et != 0);
        bool walking = (et >> 31) != 0 && !hit;             // a later insert skipped over this slot
        if (__builtin_expect(__ballot(walking) != 0, 0)) {          // collision chain: rare
            while (walking) {
                h = (h + 1) & cmask;
                const uint32_t* f = cbase + h * S;
                uint32_t d2 = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) d2 |= f[w] ^ s[w];
                const uint32_t ft = f[NW];
                const bool here = (d2 == 0) & (ft != 0);
                if (here) { hit = true; et = ft; }
                walking = (ft >> 31) != 0 && !here;
            }
        }
        return hit ? et : 0u;
    };

    for (;;) {
        const uint64_t running = __ballot(res == 0);
        const bool work_left = q.more || q.next < q.end;
        if (!running && !work_left && !__ballot(res != kResIdle)) break;
        const uint32_t n_free = 64u - (uint32_t)__popcll(running);
#ifdef BSX_DIAG
        ++dbg_iters;
#endif
        // a round is also forced every 32 iterations, so that lanes running past fast_steps are noticed
        // even when no lane of the wave ever gets free
        if ((work_left && n_free >= (merge && service_lanes < kMergeGroup ? kMergeGroup : service_lanes)) || !running ||
            ++since_service >= 32u) {
            since_service = 0;
#ifdef BSX_DIAG
            ++dbg_service;
#endif
            // ---- service round: record results, hand lost problems over, refill
            if (res == 0 && t >= fast_steps) res = kResLost;    // no cached cycle state within fast_steps
            if (res != 0 && res != kResIdle) {
                if (res == kResLost) {
                    if (merge) {            // the class goes back as (group base, member mask)
                        atomicAdd(&P.ctr->n_stragglers, (unsigned long long)__popc(members));
                        const unsigned long long at = atomicAdd(&P.ctr->straggler_classes, 1ull);
                        if (2 * at + 1 < P.stragglers_cap) { P.stragglers[2 * at] = my_p; P.stragglers[2 * at + 1] = members; }
                        else atomicOr(&P.ctr->straggler_overflow, 1u);
                    } else {
                        const unsigned long long at = atomicAdd(&P.ctr->n_stragglers, 1ull);
                        if (at < P.stragglers_cap) P.stragglers[at] = my_p;
                        else atomicOr(&P.ctr->straggler_overflow, 1u);
                    }
                    nexec += tp + (uint32_t)t;
                } else {
                    const uint32_t tg = res & kTagMask, mu = (uint32_t)t, traj = tp + mu;
                    const uint32_t m = (uint32_t)__popc(members);       // trajectories that share this result
                    nexec += traj;
                    const uint32_t lam = lamtab[tg - 1];
                    const bool found = mu <= cap_rel && lam <= cap_rel - mu;
                    const bool keep = found && (uint64_t)lam <= P.max_len;          // attract.py:294
                    if (P.per_problem) {
                        ProblemRec32 r;
#pragma unroll
                        for (int w = 0; w < kMaxW32; ++w) r.key[w] = 0;
                        if (keep) {
#pragma unroll
                            for (int w = 0; w < NW; ++w) r.key[w] = keytab[(tg - 1) * NW + w];
                        }
                        r.length = keep ? lam : 0; r.trajectory_l = keep ? traj : 0; r.found = keep; r.pad = 0;
                        if (merge) {
                            for (uint32_t left = members; left; left &= left - 1) P.per_problem[my_p + (uint32_t)__builtin_ctz(left)] = r;
                        } else {
                            P.per_problem[my_p] = r;
                        }
                    }
                    if (__builtin_expect(!keep, 0)) {
                        n_none += m;
                        n_capfail += found ? 0u : m;                        // these add max_t each at the end
                        extra_ref += found ? (unsigned long long)m * (traj + lam) : 0ull;   // model.py:201
                    } else {
                        const uint32_t wl = m * traj, wsq = wl * traj;      // weighted by the members
                        bool in_regs = false;
#pragma unroll
                        for (int j = 0; j < kTagAcc; ++j) {
                            const bool here = tg == (uint32_t)(j + 1) && tsl2[j] < kRegSumGuard;
                            tcnt[j] += here ? m : 0u;
                            tsl[j] += here ? wl : 0u;
                            tsl2[j] += here ? wsq : 0u;
                            in_regs = in_regs || here;
                        }
                        if (!in_regs) {
                            atomicAdd(&acc_cnt[tg - 1], m);
                            atomicAdd(&acc_sl[tg - 1], (unsigned long long)wl);
                            atomicAdd(&acc_sl2[tg - 1], (unsigned long long)wsq);
                        }
                    }
                }
                res = kResIdle;
            }
            if (work_left) {
                if (q.next == q.end) {
                    const uint64_t base = grab_chunk(&P.ctr->cursor, P.chunk, lane);
                    if (base >= P.count) q.more = false;
                    else { q.next = base; q.end = (base + P.chunk < P.count) ? base + P.chunk : P.count; }
                }
                const uint64_t avail = q.end - q.next;
                const uint64_t idle = __ballot(res == kResIdle);
                if (avail && idle) {
                    const uint32_t rank = rank_below(idle);
                    // with merging whole groups of kMergeGroup consecutive problems are loaded together (chunks
                    // start on group boundaries), so that their members share a clock and a member mask
                    uint64_t take = (uint64_t)__popcll(idle);
                    if (merge) take &= ~(uint64_t)(kMergeGroup - 1);
                    if (res == kResIdle && rank < take && rank < avail) {
                        const uint32_t off = (uint32_t)(q.next + rank);
                        init_problem_simple<NW>(P.sp, (uint64_t)off, A);
                        my_p = merge ? (off & ~(kMergeGroup - 1)) : off;
                        members = merge ? 1u << (off & (kMergeGroup - 1)) : 1u;
                        t = -(int32_t)tp;
                        res = has_warmup ? 0u : probe(A);       // s(T_p) = s(0) itself may be a cycle state: mu = 0
                    }
                    q.next += take < avail ? take : avail;
                }
            }
        }       // no `continue`: one back edge keeps the loop-carried registers in place (no copy chains)

        // ---- one step for every running lane, then the lookup of the new state.  Waiting and idle lanes
        //      are masked off: their LDS reads would only add bank conflicts.
        if (res == 0) {
            uint32_t nxt[NW];
            net_step<NW, K>(nv, A, fm0, fv0, nxt, has_fixed);
            ++t;
            if (has_warmup) {
                if (t <= 0) apply_perturbations<NW>(P.sp, (uint32_t)((int32_t)tp + t), 0ull, nxt);
            }
            uint32_t et = probe(nxt);
            if (has_warmup) et = t >= 0 ? et : 0u;          // states before T_p do not count
#pragma unroll
            for (int w = 0; w < NW; ++w) A[w] = nxt[w];
            res = et;               // 0 = keep running; running too long is noticed in the service round
        }

        // ---- sibling merge: trajectories of one group that are in the same state (at the same time: they
        //      were loaded together) continue as one lane with the union of their member masks.  Each
        //      running lane posts its id in a per-wave slot picked by the state's hash; whoever reads another
        //      lane's id there compares states through ds_bpermute and, if equal, hands its members over.
        //      (Lanes on a cached cycle state were resolved by the probe above and take no part, so every
        //      member of a merged lane has the same mu.  A slot collision only postpones a merge.)
        if (merge) {
            const bool cand = res == 0;
            // (sibling states differ in a few bits, so the slot needs a mixing hash: two multiplies per iteration)
            // resident groups are mostly consecutive, so two group bits keep them out of each other's slots
            const uint32_t slot = ((last_hash * 0x85EBCA6Bu) >> 26) | (((my_p >> 5) & 3u) << 6);
            // (volatile: the compiler must not forward a lane's own store to its load -- another lane's
            //  store to the same slot may have come later)
            if (cand) dd_ids3[slot] = (uint8_t)lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t w = cand ? (uint32_t)dd_ids3[slot] : (uint32_t)lane;
            // every lane takes part in the permutes (a lane masked off would deliver nothing to its readers),
            // so they come before any short-circuit logic
            uint32_t differ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w * 4u), (int)my_p) ^ my_p;
            differ |= (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w * 4u), (int)res);
#pragma unroll
            for (int i = 0; i < NW; ++i) differ |= (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w * 4u), (int)A[i]) ^ A[i];
            const bool same = cand & (w != (uint32_t)lane) & (differ == 0u);
            if (__ballot(same)) {                       // uniform: most iterations of old lanes merge nothing
                if (same) {
                    atomicOr(&dd_acc[w], members);
                    nexec += tp + (uint32_t)t;          // the steps this lane did on its own
                    res = kResIdle;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (res == 0) {
                    const uint32_t got = dd_acc3[lane];
                    if (got) { members |= got; dd_acc3[lane] = 0; }
                }
            }
        }
    }

    // ---- epilogue: per-lane sums -> workgroup accumulators -> one log record per attractor and workgroup
#pragma unroll
    for (int j = 0; j < kTagAcc; ++j) {
        if (tcnt[j]) {
            atomicAdd(&acc_cnt[j], tcnt[j]);
            atomicAdd(&acc_sl[j], (unsigned long long)tsl[j]);
            atomicAdd(&acc_sl2[j], (unsigned long long)tsl2[j]);
        }
    }
    __syncthreads();
    for (uint32_t a = threadIdx.x; a < kAccs; a += blockDim.x) {
        const uint32_t cn = acc_cnt[a];
        if (!cn) continue;
        uint32_t k[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) k[w] = keytab[a * NW + w];
        const unsigned long long sl = acc_sl[a];
        log_append<NW>(P, k, lamtab[a], cn, sl, acc_sl2[a]);
        extra_ref += sl + (unsigned long long)cn * lamtab[a];                   // + lambda each (model.py:201)
    }
#ifdef BSX_DIAG
    if (lane == 0) { atomicAdd(&P.ctr->wave_iters, dbg_iters); atomicAdd(&P.ctr->service_rounds, dbg_service); }
#endif
    wave_atomic_add(&P.ctr->steps_ref, extra_ref + (P.cap_rel_inf ? 0ull : (unsigned long long)n_capfail * P.max_t), lane);
    wave_atomic_add(&P.ctr->steps_exec, (unsigned long long)nexec, lane);
    wave_atomic_add(&P.ctr->n_none, (unsigned long long)n_none, lane);
}

template <int NW, int K>
static hipError_t launch_lean_nk(int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    const void* fn;
    BSX_KERNEL_FOR_MODE(k_attract_lean, NW, K, lut_mode, fn);
    if (!fn) return hipErrorInvalidValue;
    void* args[] = {const_cast<AttractParams*>(&P)};
    return hipLaunchKernel(fn, grid, dim3(kBlock), args, shmem, st);
}
template <int NW, int K>
static hipError_t configure_lean_nk(int lut_mode, dim3, size_t shmem, hipStream_t, int& blocks_per_cu) {
    const void* fn;
    BSX_KERNEL_FOR_MODE(k_attract_lean, NW, K, lut_mode, fn);
    if (!fn) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fn, kBlock, shmem);
}

hipError_t launch_attract_fast(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    BSX_DISPATCH(launch_lean_nk)
}
// Also reports how many workgroups of the kernel fit on one CU (registers + LDS), which sizes its grid.
hipError_t configure_attract_fast(int nw, int k, int lut_mode, size_t shmem, int* blocks_per_cu) {
    const dim3 grid(1);
    const hipStream_t st = nullptr;
    int& P = *blocks_per_cu;
    BSX_DISPATCH(configure_lean_nk)
}

}  // namespace bsx
