// Attract sweep kernel with a class pool: sibling trajectories that meet are stepped once.
// (kernel templates; instantiated per state width in bsx_pool_nw{1,2,4,8}.hip so that the four builds run in parallel,
// dispatch over the width in bsx_pool.hip)
#pragma once
#include "bsx_kernels_common.h"

namespace bsx {

// ------------------------------------------------------------------------------------------------
// k_attract_pool: attract.py:262-302 semantics for the problems that the lean kernel (bsx_lean.hip)
// covers -- plain enumeration, attractors cached -- when no per-lane bookkeeping has to survive an
// iteration.  Same results, same straggler hand-over to the general kernel; what changes is which
// network updates are executed.
//
// Observation: a sweep enumerates initial states that differ in a few low bits.  In an ordered or
// critical network most of those differences die out within a step or two, i.e. sibling trajectories
// run into the SAME state at the SAME time and are identical from there on (north-star network: the 64
// states of an aligned group of 64 problems are 9.5 distinct states after one update, 3.8 after three).
// A "class" is such a set of trajectories: (state, time, group base, 64-bit member mask).  A class is
// stepped once; when its state is a cached cycle state at time t every member has mu = t (classes on a
// cycle are resolved before they can take part in a merge, so no member was on the cycle earlier),
// and count / sum l / sum l^2 grow by m, m*l, m*l^2 with m = popcount(members).
//
// Each wave keeps its classes in a ring buffer in LDS (the pool) and runs one of two stages per
// iteration, all 64 lanes doing one network update either way:
//   fresh stage: the next 64 consecutive problems -- init, update, lookup (s(T_p) itself is looked up only when s(T_p + 1) is a cycle state);
//   pool stage (when the pool holds at least 64 classes, or the input is used up): the 64 oldest
//                classes -- update, lookup.
// After the lookup: resolved lanes are accumulated, lanes past the FAST length go to the straggler list
// as (group base, member mask), the rest is deduplicated (lanes of one group in the same state merge
// their masks: per-wave hash slots + ds_bpermute compare, as in the lean kernel) and the survivors are
// appended to the pool.  Nothing but the accumulators lives in registers across iterations, so there is
// no per-lane state machine and no service round.
// Cube pass (P.merge == 3, DESIGN.md "cube collapse"): the first update of an aligned block of 2^a consecutive
// problems depends only on the RELEVANT free digits (the host finds them from the truth tables restricted to
// the block's fixed bits), so the fresh stage enumerates the 2^r assignments of those digits instead of the
// 2^a problems and every class starts with 2^(a-r) members (64-bit member counts).  A member whose s(0) is
// itself a cycle state (mu = 0) differs from its class representative only in irrelevant bits: the mirror
// gets a second entry per cached cycle state inside the block -- the state with those bits cleared, flagged
// kTagRep -- which a t = 0 probe of the representative hits (at most one member per class can be a cycle
// state: two would share their successor).  Classes that are still unresolved at the step limit are listed
// with their state for the host (discovery of uncached attractors, then the pass is repeated).
// Workgroup = 12 waves sharing one LUT and cache mirror: with n = 64 that is 39 KiB + 12 x 3.3 KiB of LDS,
// so two workgroups fit a CU = 6 waves per SIMD, 3 from each (with 8-wave workgroups and 128-class rings
// it was 4).  The kernel is bound by the latency of its dependent LDS round trips, so waves matter.
constexpr int kPoolBlock = kPoolBlockThreads;
constexpr int kPoolWaves = kPoolBlock / 64;
constexpr uint32_t kPoolGroup = 64;             // problems loaded together = lanes

constexpr int pool_min_waves(int nw) { return nw <= 2 ? 6 : 2; }

// Rare blocks of the kernel's loop (listing a class, a member that is a cycle state itself, the next chunk of work) read
// their parameters through the opaque kernel-argument pointer (bsx_kernels_common.h: BSX_KERNARG), so the loads and the
// address arithmetic stay inside the rare block.  Through `P` the compiler hoists them in front of the loop, where they hold
// scalar registers for the whole launch -- 36 of them were spilled to VGPR lanes and scratch in round 2's build
// (profiles/r03_kernel_resources.csv).
#define BSX_RARE_PARAMS(name) BSX_KERNARG(AttractParams, name)

// OR the digits of `d` into `s` along the deposit plan (init_problem_simple without the origin).  With a
// wave-uniform `d` this is scalar work.
template <int NW>
__device__ __forceinline__ void deposit_runs(const DevSpace& sp, uint64_t d, uint32_t (&s)[NW], uint32_t run_first = 0,
                                             uint32_t run_end = 0xFFFFFFFFu) {
    const uint32_t end = run_end < sp.n_runs ? run_end : sp.n_runs;
    for (uint32_t r = run_first; r < end; ++r) {
        const uint32_t desc = sp.deposit[2 * r], mask = sp.deposit[2 * r + 1];
        const uint32_t piece = ((uint32_t)(d >> (desc & 63u)) & mask) << ((desc >> 16) & 31u);
        const uint32_t word = (desc >> 8) & 7u;
#pragma unroll
        for (int w = 0; w < NW; ++w) s[w] |= (word == (uint32_t)w) ? piece : 0u;
    }
}

// Per-parent level (bsx_device.h: LeafProgram): a batch of 64 parents with all their children is a long piece of serial work
// for one wave (0.15 ms at 512 children each) while most waves of a launch have nothing to do, so the children -- in units
// of one 32-children word -- are shared out over a power of two of work items per batch, enough for two per wave.
__device__ __forceinline__ uint32_t leaf_parts(uint32_t kb, unsigned long long listed, uint64_t n_waves) {
    const uint32_t units = kb > 5u ? 1u << (kb - 5u) : 1u;
    const uint64_t batches = (listed + 63ull) / 64ull;
    uint32_t parts = 1;
    while (parts < units && batches * parts < 2ull * n_waves) parts <<= 1;
    return parts;
}

// CUBE = true: the cube-pass build of the kernel (P.merge == 3): member counts only, no member masks, no
// per-problem records; kept apart so that neither build carries the other's registers.
// LOWER = true (with CUBE): the build for the lower levels of a chained cascade -- entry-based passes, whose classes are all
// settled by their first lookup (see the main loop), so it has no pool, no rings in LDS and two batches of 64 classes in
// flight per iteration.
template <int NW, int K, int LM, bool CUBE, bool LOWER = false>
__global__ __launch_bounds__(kPoolBlock, pool_min_waves(NW)) void k_attract_pool(const AttractParams P) {
    static_assert(CUBE || !LOWER, "the lower-level build is a cube build");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
#ifdef BSX_DIAG
    const unsigned long long dbg_t0 = wall_clock64();
#endif
    // work items and their split over the waves: from the host, or -- a lower level of a chained cascade, whose size
    // only the level above knows -- from that level's descriptor (bsx_device.h: LevelDesc)
    uint64_t n_items = P.count, chunk_first = P.chunk_first;
    uint32_t chunk = P.chunk;
    if constexpr (CUBE) {
        if (P.level_in) {                                   // uniform
            const unsigned long long listed = P.level_in->n_entries;
            n_items = P.level_in->abort ? 0ull : listed << P.entry_shift;
            if constexpr (LOWER) {
                // per-parent level: the children of a batch of 64 parents are shared out over `parts` work items (below)
                if (P.leaf && n_items) n_items = ((listed + 63ull) & ~63ull) * leaf_parts(P.leaf->kb, listed, (uint64_t)gridDim.x * kPoolWaves);
            }
            if (n_items == 0) {                             // nothing was handed down (or the level above overflowed)
                if (threadIdx.x == 0 && P.near_counts) P.near_counts[blockIdx.x] = 0;
                return;
            }
            const uint64_t n_waves = (uint64_t)gridDim.x * kPoolWaves;
            if (n_items < (1ull << 28)) { chunk_first = ((n_items + n_waves - 1) / n_waves + 63) / 64 * 64; chunk = 0; }
            else { chunk_first = 4096; chunk = 4096; }
            // a short list keeps only some workgroups busy (wave w of workgroup g takes share w * gridDim.x + g, so those are
            // spread over the chip): the others leave before they stage anything
            if ((uint64_t)blockIdx.x * chunk_first >= n_items) {
                if (threadIdx.x == 0 && P.near_counts) P.near_counts[blockIdx.x] = 0;
                return;
            }
        }
    }
    if constexpr (CUBE) {
        if (threadIdx.x == 0 && !P.mirror_out) atomicMax(&P.ctr->t_first_not, ~(unsigned long long)wall_clock64());
    }
    const NetView<NW, K, LM> nv = stage_network<NW, K, LM>(P.net, smem, smem_free);
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tp = P.sp.tp_origin;                     // uniform: no variations here
    const bool has_warmup = tp != 0;
    // P.merge == 2: classes carry a member COUNT instead of the member mask, so classes of different groups
    // may merge too (same state, same time); a class that would have to go back to the general kernel cannot
    // be taken apart again, so it raises the abort flag and the host repeats the tile with member masks.
    const bool counting = CUBE || P.merge >= 2;             // uniform
    constexpr bool cube = CUBE;                             // work items are relevant-digit assignments
    const int32_t fast_steps = (int32_t)P.fast_steps;
    const uint32_t cmask = P.cc.lds_slots - 1;
    constexpr int S = CacheLayout<NW>::kStride;
    constexpr uint32_t kAccs = (uint32_t)kTagAcc + kLdsAcc;
    // ring record: state, group base, members lo/hi, time.  A cube pass has no group base and packs the time next to
    // the high word of its member count (count < 2^49, -256 <= t < 2^12 - 256): NW + 2 words -- one 16-byte access
    // at NW = 2 -- and what the shorter records free goes into a longer ring.
    constexpr uint32_t R = cube ? NW + 2 : pool_rec_words(NW);
    constexpr uint32_t kCap = cube ? (kPoolCap * pool_rec_words(NW)) / (NW + 2) : kPoolCap;

    // LDS: [network tables][cache mirror][per attractor: sum l^2, sum l (u64), count, length (u32), key]
    //      [per wave: pool records | 64 x 8 B member accumulators | 256 one-byte lane ids]
    uint32_t* lc = smem + (((uint32_t)(smem_free - smem) + 3u) & ~3u);
    const uint32_t lc_words = kCacheHeaderWords + P.cc.lds_slots * S;
    unsigned long long* acc_sl2 = reinterpret_cast<unsigned long long*>(lc + ((lc_words + 1u) & ~1u));     // 128-bit sum l^2: low,
    unsigned long long* acc_sl2h = acc_sl2 + kAccs;                                                         // high
    unsigned long long* acc_sl = acc_sl2h + kAccs;
    unsigned long long* acc_cnt = acc_sl + kAccs;
    uint32_t* lamtab = reinterpret_cast<uint32_t*>(acc_cnt + kAccs);
    uint32_t* keytab = lamtab + kAccs;
    constexpr uint32_t kWaveWords = LOWER ? 0u : kPoolCap * pool_rec_words(NW) + 128 + kPoolSlots / 4;
    constexpr uint32_t kFoundWords = LOWER ? kLowerFoundWords : 0u;     // (the other builds borrow wave 0's ring for that list)
    constexpr uint32_t kLowerCntWords = LOWER ? 2u * kAccs : 0u;        // lower-level build, per wave: classes by outcome (tag, hit / miss)
    static_assert(kAccs % 2 == 0 && kWaveWords % 2 == 0, "64-bit LDS atomics need 8-byte aligned tables");
    // what does not end on a kept attractor (no-attractor counts, reference steps of the reference's loop, cap failures):
    // rare with production caps, so it is summed in the workgroup's LDS as it happens instead of in three 64-bit registers
    // per lane that lived across the whole loop (and were spilled to scratch)
    unsigned long long* wg_ctr = reinterpret_cast<unsigned long long*>(keytab + ((kAccs * NW + 1u) & ~1u));    // [0] none, [1] cap failures, [2] reference steps
    uint32_t* midtab = reinterpret_cast<uint32_t*>(wg_ctr + 4);  // cube pass: deposits of class-index bits 6..11, [64][NW]
    uint32_t* wave_base = midtab + 64 * NW + kFoundWords + wave * (kWaveWords + kLowerCntWords);
    typedef volatile uint32_t __attribute__((address_space(3))) lds_vu32;
    typedef volatile uint8_t __attribute__((address_space(3))) lds_vu8;
    lds_vu32* const pool = (lds_vu32*)(__attribute__((address_space(3))) uint32_t*)wave_base;
    lds_vu32* const dd_acc = pool + kPoolCap * pool_rec_words(NW);           // [64][2]
    lds_vu8* const dd_ids = (lds_vu8*)(dd_acc + 128);

    uint32_t fm0[NW], fv0[NW];
    uint32_t any_fixed = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { fm0[w] = P.sp.fixmask[w]; fv0[w] = P.sp.fixval[w]; any_fixed |= fm0[w]; }
    const bool has_fixed = any_fixed != 0;                  // uniform
    const uint32_t* const image = P.mirror_image;           // uniform: the mirror as built by an earlier launch (bsx_device.h)
    if (image) { for (uint32_t i = threadIdx.x; i < lc_words; i += blockDim.x) lc[i] = image[i]; }
    else { for (uint32_t i = threadIdx.x; i < lc_words; i += blockDim.x) lc[i] = 0; }
    for (uint32_t i = threadIdx.x; i < kAccs; i += blockDim.x) { acc_sl2[i] = 0; acc_sl2h[i] = 0; acc_sl[i] = 0; acc_cnt[i] = 0; lamtab[i] = 0; }
    if (threadIdx.x < 4) wg_ctr[threadIdx.x] = 0;
    if constexpr (!LOWER) { dd_acc[2 * lane] = 0; dd_acc[2 * lane + 1] = 0; }
    else { wave_base[lane] = 0; wave_base[64 + lane] = 0; }
    __syncthreads();
    if (!image && threadIdx.x == 0) {
        uint32_t seen = 0, n_states = 0, n_attr = 0;
        cache_pull<NW, K>(P.cc, nv, fm0, fv0, lc, seen, n_states, n_attr, kAccs);
    }
    __syncthreads();
    if (P.mirror_out) {                                     // uniform: this launch only builds the image
        for (uint32_t i = threadIdx.x; i < lc_words; i += blockDim.x) P.mirror_out[i] = lc[i];
        return;
    }
    const uint32_t* cbase = lc + kCacheHeaderWords;
    for (uint32_t sl = threadIdx.x; sl < P.cc.lds_slots; sl += blockDim.x) {
        const uint32_t tg = cbase[sl * S + NW] & kTagMask;
        if (tg) {
            lamtab[tg - 1] = cbase[sl * S + NW + 1];
#pragma unroll
            for (int w = 0; w < NW; ++w) keytab[(tg - 1) * NW + w] = cbase[sl * S + NW + 2 + w];
        }
    }
    __syncthreads();
    if constexpr (cube) {
        for (uint32_t i = threadIdx.x; i < 64u; i += blockDim.x) {
            uint32_t m[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) m[w] = 0;
            deposit_runs<NW>(P.sp, (uint64_t)i << 6, m, 6u, 12u);
#pragma unroll
            for (int w = 0; w < NW; ++w) midtab[i * NW + w] = m[w];
        }
    }
    // (a deep pass hands every class with members that close to a cycle to the level below instead)
    const uint32_t depth = (cube && P.cube_depth > 1u) ? P.cube_depth : 1u;   // uniform: updates of a fresh class before its first lookup
    const bool from_entries = cube && P.entries != nullptr;                   // uniform: a lower level of a cascade
    // cube pass: cached cycle states inside the block get a second mirror entry, their class representative.
    // All threads look through the mirror; the few states found are inserted by thread 0.
    if constexpr (cube) {
        if (threadIdx.x == 0) { lc[1] = 0; lc[2] = 0; }     // lc[1]: cycle states inside the block, lc[2]: classes listed for the level below
        __syncthreads();
        if (depth == 1u && !has_warmup) {
            constexpr uint32_t kFoundCap = LOWER ? kLowerFoundWords : kCap * R;     // slots of the states found go to wave 0's ring (not in use yet)
            auto inside = [&](uint32_t sl) -> bool {
                const uint32_t* e = cbase + sl * S;
                const uint32_t tw = e[NW];
                if ((tw & kTagMask) == 0 || (tw & kTagRep)) return false;
                uint32_t outside = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) outside |= (e[w] ^ P.sp.origin[w]) & ~P.cube_free[w];
                return outside == 0;
            };
            lds_vu32* const found0 = (lds_vu32*)(__attribute__((address_space(3))) uint32_t*)(midtab + 64 * NW);
            for (uint32_t sl = threadIdx.x; sl < P.cc.lds_slots; sl += blockDim.x) {
                if (!inside(sl)) continue;
                const uint32_t at = atomicAdd((uint32_t*)(__attribute__((address_space(3))) uint32_t*)&lc[1], 1u);
                if (at < kFoundCap) found0[at] = sl;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t n_in = lc[1];
                auto add_rep = [&](uint32_t sl) {
                    const uint32_t* e = cbase + sl * S;
                    uint32_t rep[NW], key[NW], differs = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        rep[w] = e[w] & ~P.cube_umask[w];
                        differs |= rep[w] ^ e[w];
                        key[w] = e[NW + 2 + w];
                    }
                    if (differs) cache_insert_lds<NW>(lc, cmask, rep, e[NW + 1], key, (e[NW] & kTagMask) | kTagRep);
                };
                if (n_in <= kFoundCap) {
                    for (uint32_t i = 0; i < n_in; ++i) add_rep(found0[i]);
                } else {                                    // (more than the list holds: one thread walks the mirror)
                    for (uint32_t sl = 0; sl < P.cc.lds_slots; ++sl) if (inside(sl)) add_rep(sl);
                }
            }
        }
    }
    __syncthreads();
    // uniform: some member may have mu = 0 (with a warm-up the search starts at s(T_p), which all members share)
    const bool t0_lookup = cube && !has_warmup && depth == 1u && __builtin_amdgcn_readfirstlane(lc[1]) != 0;
    // cube pass: a wave's 64 classes differ in the six lowest relevant digits only; where those land in the
    // state is the same in every iteration, the rest of the class index is wave-uniform (scalar deposit)
    uint32_t lane_part[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) lane_part[w] = 0;
    if constexpr (cube) deposit_runs<NW>(P.sp, (uint64_t)lane, lane_part);
    // ... and bits 12 and up change once per 4096 classes: their deposit is kept (scalar registers) with its tag
    uint32_t u_hi[NW];
    uint64_t u_hi_tag = ~0ull;
#pragma unroll
    for (int w = 0; w < NW; ++w) u_hi[w] = 0;

    const uint32_t cap_rel = (P.cap_rel_inf || P.max_t - tp >= (kStepLimit / 4)) ? 0xFFFFFFFFu : (uint32_t)(P.max_t - tp);

    // member counts: 64 bits in a cube pass (a class stands for up to 2^48 problems), 32 bits otherwise (a tile
    // has at most 2^28 problems) -- the plain build keeps its registers
    using cnt_t = std::conditional_t<CUBE, unsigned long long, uint32_t>;
    uint32_t nexec = 0;
    // work queue: every wave's first chunk is fixed (wave w of workgroup g takes chunk w * gridDim.x + g: a pass with fewer
    // chunks than waves runs on all CUs with few waves each), only the chunks after those come from the shared cursor --
    // a small pass has no traffic on that one address at all
    const uint64_t first_dyn = (uint64_t)gridDim.x * kPoolWaves * chunk_first;
    WaveQueue q{0, 0, chunk != 0 && n_items > first_dyn};
    {
        const uint64_t b = ((uint64_t)wave * gridDim.x + blockIdx.x) * chunk_first;
        if (b < n_items) { q.next = b; q.end = (b + chunk_first < n_items) ? b + chunk_first : n_items; }
    }
    uint32_t head = 0, count = 0;                           // pool ring (uniform)
#ifdef BSX_DIAG
    unsigned long long dbg_iters = 0, dbg_fresh = 0, dbg_fresh_keep = 0, dbg_pool_in = 0, dbg_pool_keep = 0, dbg_merged = 0;
    bool dbg_is_fresh = false;
#endif

    // cube pass: the representative state of class `pos + lane`
    uint32_t ptag = 0;                      // entry-based pass: tag of the cycle the parent class's common state lies on
    auto fresh_state = [&](uint64_t pos, bool lv, uint32_t (&S0)[NW]) {
        // (a cube's plan has one run per enumerated digit, so run r is class-index bit r; pos is a multiple of 64)
        if ((pos >> 12) != u_hi_tag) {                      // uniform, rare
            u_hi_tag = pos >> 12;
#pragma unroll
            for (int w = 0; w < NW; ++w) u_hi[w] = P.sp.origin[w];
            deposit_runs<NW>(P.sp, pos & ~0xFFFull, u_hi, 12u);
        }
        const uint32_t mid = (((uint32_t)pos >> 6) & 63u) * NW;            // uniform: one broadcast read per word
#pragma unroll
        for (int w = 0; w < NW; ++w) S0[w] = u_hi[w] | midtab[mid + w] | lane_part[w];
        if (P.entries) {                                    // uniform: a listed class of the level above, plus this level's digits
            const uint64_t e = (pos + lane) >> P.entry_shift;
            if (lv) {                                       // entry = the listed class's state + the tag of its cycle
#pragma unroll
                for (int w = 0; w < NW; ++w) S0[w] |= P.entries[e * (NW + 1) + w];
                ptag = P.entries[e * (NW + 1) + NW];
            }
        }
    };

    // is `s` a cached cycle state?  -> the entry's tag word (0 = no); `hfull` = the state's hash
    uint32_t hit_len = 0;                   // NW <= 2: the entry's length word comes with the probe's 16-byte read
    // (representative entries of a cube pass count only for the t = 0 probe: `reps`)
    auto probe = [&](const uint32_t (&s)[NW], uint32_t& hfull, bool reps = false) -> uint32_t {
        const uint32_t ignore = reps ? 0u : kTagRep;
        hfull = hash_state<NW>(s);
        uint32_t h = hfull & cmask;
        const uint32_t* e = cbase + h * S;
        uint32_t et, d;
        if constexpr (NW == 1) {
            uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(e, 16));
            asm volatile("" : "+v"(v.w));
            d = v.x ^ s[0]; et = v.y; hit_len = v.z;
        } else if constexpr (NW == 2) {
            uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(e, 16));
            d = (v.x ^ s[0]) | (v.y ^ s[1]); et = v.z; hit_len = v.w;
        } else {
            d = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) d |= e[w] ^ s[w];
            et = e[NW];
        }
        bool hit = (d == 0) & (et != 0) & ((et & ignore) == 0);
        bool walking = (et >> 31) != 0 && !hit;
        if (__builtin_expect(__ballot(walking) != 0, 0)) {
            while (walking) {
                h = (h + 1) & cmask;
                const uint32_t* f = cbase + h * S;
                uint32_t d2 = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) d2 |= f[w] ^ s[w];
                const uint32_t ft = f[NW];
                const bool here = (d2 == 0) & (ft != 0) & ((ft & ignore) == 0);
                if (here) { hit = true; et = ft; hit_len = f[NW + 1]; }
                walking = (ft >> 31) != 0 && !here;
            }
        }
        return hit ? et : 0u;
    };

    // m members end on the cycle state with tag word `tagw` at time mu (attract.py:291-298 for each of them)
    auto account = [&](uint32_t tagw, cnt_t m, uint32_t mu, uint32_t lam, bool& keep_out) {
        const uint32_t tg = tagw & kTagMask, traj = tp + mu;
        const bool found = mu <= cap_rel && lam <= cap_rel - mu;
        const bool keep = found && (uint64_t)lam <= P.max_len;              // attract.py:294
        keep_out = keep;
        if (__builtin_expect(!keep, 0)) {
            atomicAdd(&wg_ctr[0], (unsigned long long)m);
            if (found) atomicAdd(&wg_ctr[2], (unsigned long long)m * (traj + lam));      // model.py:201
            else atomicAdd(&wg_ctr[1], (unsigned long long)m);
        } else {
            // Straight into the workgroup's accumulators: a wave resolves less than one class per
            // iteration on average (64 problems end as one or two classes), so the LDS atomics do not
            // queue up, and no per-lane sums have to be carried in registers.
            const unsigned long long wl = (unsigned long long)m * traj;     // cube: m < 2^49, traj < 2^13; else m < 2^29
            atomicAdd(&acc_cnt[tg - 1], (unsigned long long)m);
            atomicAdd(&acc_sl[tg - 1], wl);
            if constexpr (cube) {                                           // 128-bit sum of m * traj^2
                const unsigned long long lo = wl * traj, hi = __umul64hi(wl, (unsigned long long)traj);
                const unsigned long long old = atomicAdd(&acc_sl2[tg - 1], lo);
                const unsigned long long up = hi + ((old + lo < old) ? 1ull : 0ull);
                if (up) atomicAdd(&acc_sl2h[tg - 1], up);
            } else {
                atomicAdd(&acc_sl2[tg - 1], wl * traj);                     // < 2^57 per add, < 2^64 per workgroup
            }
        }
    };

    // one problem (sign = +1) or one problem less (-1) on the cycle state with tag word `tagw` at time mu: the same
    // bookkeeping as account() in absolute numbers, straight into the global correction sums (a handful per pass)
    auto account_fix = [&](uint32_t tagw, uint32_t mu, uint32_t lam, long long sign) {
        BSX_RARE_PARAMS(Pr);
        asm volatile("" : "+v"(mu));                        // (opaque: nothing derived from the literal is kept across the loop)
        const uint32_t tg = tagw & kTagMask, traj = tp + mu;
        const bool found = mu <= cap_rel && lam <= cap_rel - mu;
        const bool keep = found && (uint64_t)lam <= Pr->max_len;
        Counters* c = Pr->ctr;
        if (found) atomicAdd(&c->fix_ref, (unsigned long long)(sign * (long long)(traj + lam)));
        else atomicAdd(&c->fix_capfail, (unsigned long long)sign);
        if (!keep) { atomicAdd(&c->fix_none, (unsigned long long)sign); return; }
        // (the attractor may have no whole class in this pass -- a time cap between mu = 0 and mu = 1 -- so its length
        // and key, which otherwise come with the workgroup sums, are stored here as well: same values from everybody)
        c->acc_len[tg - 1] = lam;
#pragma unroll
        for (int w = 0; w < NW; ++w) c->acc_key[tg - 1][w] = keytab[(tg - 1) * NW + w];
        atomicAdd(&c->fix_cnt[tg - 1], (unsigned long long)sign);
        atomicAdd(&c->fix_sl[tg - 1], (unsigned long long)(sign * (long long)traj));
        atomicAdd(&c->fix_sl2[tg - 1], (unsigned long long)(sign * (long long)traj * (long long)traj));
    };

#ifdef BSX_DIAG
    const unsigned long long dbg_t1 = wall_clock64();
#endif
    if constexpr (LOWER) {
        // ---- lower level of a cascade: every class is settled by its first lookup (the reasoning is at `from_entries` in
        // the general loop below), so an iteration is: two batches of 64 classes -- entry + this level's digits -> `depth`
        // updates -> one lookup -> listed for the level below (depth > 1), or booked by outcome with a ballot.  The two
        // batches are independent chains of LDS / HBM round trips that overlap; nothing lives across iterations.
        // Outcomes are counted per wave in LDS -- slot 2 (tag - 1) for the classes that are resolved by the lookup itself
        // (depth 1), + 1 for those that enter their parent's cycle with the next update -- and booked once, at the end:
        // the bookkeeping of an outcome (a 128-bit product, the caps) is some fifty instructions, which per batch and
        // outcome was half of what a batch of one-update classes executed.
        uint32_t* const wave_cnt = wave_base;
        if (P.leaf) {
            // ---- depth-1 level evaluated per parent (bsx_device.h: LeafProgram).  Work items are the listed entries themselves.
            const LeafProgram* const L = P.leaf;
            const uint32_t kb = L->kb, n_dep = L->n_dep;
            // (the children are taken 512 at a time: 16 words of 32; digits 9.. of the child index number the pieces)
            const uint32_t kb_in = kb < 9u ? kb : 9u;
            const uint32_t n_words = kb_in > 5u ? 1u << (kb_in - 5u) : 1u;                     // <= 16
            const uint32_t word_mask = kb_in >= 5u ? 0xFFFFFFFFu : (1u << (1u << kb_in)) - 1u;  // children in a word
            uint32_t indep[NW], added[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) { indep[w] = L->indep[w]; added[w] = L->added[w]; }
            // the dependent rules go to LDS (the prologue's list of cycle states inside the block is done with its 256 words): the
            // loops below read one per (candidate, rule), and a scalar load from HBM there was most of a work item's latency
            static_assert(sizeof(LeafDep) == 16 && kLeafMaxDeps * 4 <= kLowerFoundWords, "the rules fit the found list's words");
            uint32_t* const dep_lds = midtab + 64 * NW;
            for (uint32_t i = threadIdx.x; i < n_dep * 4u; i += blockDim.x) dep_lds[i] = reinterpret_cast<const uint32_t*>(L->dep)[i];
            __syncthreads();
            // work item = (parent, part of its children): items [part * listed64, (part + 1) * listed64) are part `part` of all parents
            const unsigned long long listed = P.level_in->n_entries, listed64 = (listed + 63ull) & ~63ull;
            const uint32_t parts = leaf_parts(kb, listed, (uint64_t)gridDim.x * kPoolWaves);
            const uint32_t units_per_part = (kb > 5u ? 1u << (kb - 5u) : 1u) / parts;         // 32-children words
            const uint32_t children_per_part = kb > 5u ? units_per_part * 32u : 1u << kb;
            for (;;) {
                if (q.next == q.end) {
                    if (!q.more) break;
                    BSX_RARE_PARAMS(Pr);
                    const uint64_t b = first_dyn + grab_chunk(&Pr->ctr->cursor, chunk, (int)lane);
                    if (b >= n_items) { q.more = false; continue; }
                    q.next = b; q.end = (b + chunk < n_items) ? b + chunk : n_items;
                }
                const uint64_t avail = q.end - q.next;
                const uint32_t n = avail < 64u ? (uint32_t)avail : 64u;
                const uint32_t part = (uint32_t)(q.next / listed64);                // uniform (shares are multiples of 64)
                const unsigned long long pbase = q.next - (unsigned long long)part * listed64;
                const bool lv = lane < n && pbase + lane < listed;
                // this part's children: pieces [piece0, piece0 + n_pieces_here), of each the words [w0, w0 + n_words_here)
                const uint32_t unit0 = part * units_per_part;
                const uint32_t piece0 = unit0 >> 4, w0 = units_per_part >= 16u ? 0u : unit0 & 15u;
                const uint32_t n_pieces_here = units_per_part >= 16u ? units_per_part >> 4 : 1u;
                const uint32_t w_end = units_per_part >= 16u ? n_words : w0 + units_per_part;
                uint32_t Sp[NW], Y[NW], tagp = 0;       // the parent's representative, its first update
#pragma unroll
                for (int w = 0; w < NW; ++w) { Sp[w] = 0; Y[w] = 0; }
                if (lv) {
                    const uint32_t* ent = P.entries + (pbase + lane) * (NW + 1);
#pragma unroll
                    for (int w = 0; w < NW; ++w) Sp[w] = ent[w];
                    tagp = ent[NW];
                    net_step<NW, K>(nv, Sp, fm0, fv0, Y, has_fixed);
                    nexec += part == 0u ? 1u : 0u;      // (the other parts repeat it: not counted as work)
                }
                q.next += n;
                uint32_t hits = 0;
                for (uint32_t sl = 0; sl < P.cc.lds_slots; ++sl) {          // (uniform: every lane looks at the same entry)
                    const uint32_t* e = cbase + sl * S;
                    const uint32_t tagw = __builtin_amdgcn_readfirstlane(e[NW]);
                    if ((tagw & kTagMask) == 0) continue;
                    {
                        // A cycle state inside the block is the representative of its class either as a flagged entry of its own
                        // (its irrelevant bits cleared) or, if it has none set, as itself.  Is that class a child of this parent --
                        // equal to the parent's representative outside the added digits?  Then one member of it has mu = 0
                        // (the class itself is counted among the hits below, at mu = 1).
                        uint32_t d0 = 0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) d0 |= (e[w] ^ Sp[w]) & ~added[w];
                        if (lv && d0 == 0 && part == 0u) {
                            const uint32_t lam0 = lamtab[(tagw & kTagMask) - 1];
                            account_fix(tagw, 0u, lam0, 1ll);
                            account_fix(tagw, 1u, lam0, -1ll);
                        }
                        if (tagw & kTagRep) continue;
                    }
                    uint32_t d = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) d |= (e[w] ^ Y[w]) & indep[w];
                    const bool ok = lv && d == 0;
                    if (!__ballot(ok)) continue;
                    // the children whose first update is exactly this cycle state: dependent nodes, 32 children per word
                    uint32_t n_hit = 0;
                    for (uint32_t piece = piece0; piece < piece0 + n_pieces_here; ++piece) {
                        uint32_t match[16];
#pragma unroll
                        for (int w = 0; w < 16; ++w) match[w] = ((uint32_t)w >= w0 && (uint32_t)w < w_end) ? word_mask : 0u;
                        for (uint32_t di = 0; di < n_dep; ++di) {
                            LeafDep dep;                                        // uniform (one broadcast read)
                            {
                                const uint32_t* const dw = dep_lds + 4u * di;
                                const uint32_t d0w = __builtin_amdgcn_readfirstlane(dw[0]), d1w = __builtin_amdgcn_readfirstlane(dw[1]);
                                const uint32_t d2w = __builtin_amdgcn_readfirstlane(dw[2]);
                                dep.in[0] = (uint16_t)d0w; dep.in[1] = (uint16_t)(d0w >> 16); dep.in[2] = (uint16_t)d1w; dep.in[3] = (uint16_t)(d1w >> 16);
                                dep.node = (uint16_t)d2w; dep.k = (uint16_t)(d2w >> 16);
                                dep.tt = __builtin_amdgcn_readfirstlane(dw[3]);
                            }
                            const uint32_t node = dep.node, k = dep.k;
                            uint32_t want_bit = 0;                              // this cycle state's value of the node (uniform)
#pragma unroll
                            for (int w = 0; w < NW; ++w) want_bit |= e[w] & (((node >> 5) == (uint32_t)w) ? 1u << (node & 31u) : 0u);
                            const uint32_t want = want_bit ? 0xFFFFFFFFu : 0u;
                            uint32_t selp[kLeafMaxK];                           // inputs that are parent bits: one broadcast per lane
#pragma unroll
                            for (int j = 0; j < (int)kLeafMaxK; ++j)
                                selp[j] = ((uint32_t)j < k && !(dep.in[j] & 0x8000u)) ? 0u - get_bit<NW>(Sp, dep.in[j]) : 0u;
#pragma unroll
                            for (int w = 0; w < 16; ++w) {
                                if ((uint32_t)w >= w0 && (uint32_t)w < w_end) { // uniform
                                    const uint32_t wg = piece * 16u + (uint32_t)w;     // the word's number among all children's
                                    uint32_t sel[kLeafMaxK];
#pragma unroll
                                    for (int j = 0; j < (int)kLeafMaxK; ++j) {
                                        const uint32_t q_digit = dep.in[j] & 0x7FFFu;
                                        const uint32_t pat = q_digit == 0u ? 0xAAAAAAAAu : q_digit == 1u ? 0xCCCCCCCCu : q_digit == 2u ? 0xF0F0F0F0u :
                                                             q_digit == 3u ? 0xFF00FF00u : q_digit == 4u ? 0xFFFF0000u :
                                                             (((wg >> (q_digit - 5u)) & 1u) ? 0xFFFFFFFFu : 0u);
                                        sel[j] = (dep.in[j] & 0x8000u) ? pat : selp[j];
                                    }
                                    // mux tree over the table bits (inputs beyond k select the low half: their selector is 0)
                                    uint32_t v[1 << (kLeafMaxK - 1)];
#pragma unroll
                                    for (int i = 0; i < (1 << (kLeafMaxK - 1)); ++i) {
                                        const uint32_t lo = 0u - ((dep.tt >> (2 * i)) & 1u), hi = 0u - ((dep.tt >> (2 * i + 1)) & 1u);
                                        v[i] = (sel[0] & hi) | (~sel[0] & lo);
                                    }
#pragma unroll
                                    for (int j = 1; j < (int)kLeafMaxK; ++j)
#pragma unroll
                                        for (int i = 0; i < (1 << (kLeafMaxK - 1 - j)); ++i) v[i] = bfi(sel[j], v[2 * i + 1], v[2 * i]);
                                    match[w] &= ~(v[0] ^ want);
                                }
                            }
                        }
#pragma unroll
                        for (int w = 0; w < 16; ++w) n_hit += (uint32_t)__popc(match[w]);
                    }
                    n_hit = ok ? n_hit : 0u;
                    if (n_hit) atomicAdd(&wave_cnt[2u * ((tagw & kTagMask) - 1u)], n_hit);
                    hits += n_hit;
                }
                if (lv) atomicAdd(&wave_cnt[2u * (tagp - 1u) + 1u], children_per_part - hits);
            }
        } else {
#ifndef BSX_LOWER_BATCHES
#define BSX_LOWER_BATCHES 2
#endif
        constexpr int B = BSX_LOWER_BATCHES;
        for (;;) {
            if (q.next == q.end) {
                if (!q.more) break;
                BSX_RARE_PARAMS(Pr);
                const uint64_t b = first_dyn + grab_chunk(&Pr->ctr->cursor, chunk, (int)lane);
                if (b >= n_items) { q.more = false; continue; }
                q.next = b; q.end = (b + chunk < n_items) ? b + chunk : n_items;
            }
            uint32_t S[B][NW], tagp[B], et[B];
            bool lv[B];
            uint64_t pos[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const uint64_t avail = q.end - q.next;
                const uint32_t n = avail < 64u ? (uint32_t)avail : 64u;
                lv[b] = lane < n;
                pos[b] = q.next;
#pragma unroll
                for (int w = 0; w < NW; ++w) S[b][w] = 0;
                tagp[b] = 0;
                if (n) { fresh_state(pos[b], lv[b], S[b]); tagp[b] = ptag; }
                q.next += n;
            }
            if (t0_lookup) {                                // (depth 1: a member that is a cycle state itself, as below)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    uint32_t h0;
                    const uint32_t et0 = lv[b] ? probe(S[b], h0, true) : 0u;
                    if (et0) {
                        const uint32_t lam0 = lamtab[(et0 & kTagMask) - 1];
                        bool kept;
                        uint32_t one = 1u;
                        unsigned long long m0 = 1ull << P.cube_shift;
                        asm volatile("" : "+v"(one), "+v"(m0));
                        account(et0, m0, one, lam0, kept);
                        account_fix(et0, 0u, lam0, 1ll);
                        account_fix(et0, 1u, lam0, -1ll);
                        lv[b] = false;
                    }
                }
            }
            for (uint32_t i = 1; i < depth; ++i) {
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    if (lv[b]) {
                        uint32_t nx[NW];
                        net_step<NW, K>(nv, S[b], fm0, fv0, nx, has_fixed);
#pragma unroll
                        for (int w = 0; w < NW; ++w) S[b][w] = nx[w];
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                et[b] = 0;
                if (lv[b]) {
                    uint32_t nx[NW], hf;
                    net_step<NW, K>(nv, S[b], fm0, fv0, nx, has_fixed);
                    et[b] = probe(nx, hf);
                    nexec += depth;
                }
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                if (depth > 1u) {
                    // F^depth(x) on a cycle: the members' entry times differ -- list the class for the level below
                    const bool near = lv[b] && et[b] != 0;
                    const uint64_t nb = __ballot(near);
                    if (nb) {
                        BSX_RARE_PARAMS(Pr);
                        uint32_t S0[NW];
                        fresh_state(pos[b], near, S0);
                        uint32_t at0 = 0;
                        if (lane == 0) at0 = atomicAdd((uint32_t*)(__attribute__((address_space(3))) uint32_t*)&lc[2], (uint32_t)__popcll(nb));
                        const uint32_t at = __builtin_amdgcn_readfirstlane(at0) + rank_below(nb);
                        if (near && at < Pr->near_cap) {
                            uint32_t* seg = Pr->near + ((uint64_t)blockIdx.x * Pr->near_cap + at) * (NW + 1);
#pragma unroll
                            for (int w = 0; w < NW; ++w) seg[w] = S0[w];
                            seg[NW] = et[b] & kTagMask;
                        }
                    }
                    if (near) lv[b] = false;
                }
                const bool hit = lv[b] && et[b] != 0;                       // (depth 1 only)
                const uint32_t otag = hit ? (et[b] & kTagMask) : tagp[b];
                bool todo = lv[b];
                uint64_t tb = __ballot(todo);
                while (tb) {
                    const int first = __builtin_ctzll(tb);
                    const uint32_t tg = (uint32_t)__builtin_amdgcn_readlane((int)otag, first);
                    const bool h0 = __builtin_amdgcn_readlane((int)(hit ? 1u : 0u), first) != 0;
                    const bool same = todo && otag == tg && hit == h0;
                    const uint64_t sb = __ballot(same);
                    if (lane == (uint32_t)first) atomicAdd(&wave_cnt[2u * (tg - 1u) + (h0 ? 0u : 1u)], (uint32_t)__popcll(sb));
                    todo = todo && !same;
                    tb &= ~sb;
                }
            }
        }
        }       // (per-child loop)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t half = 0; half < 2; ++half) {         // lane l books slots l and 64 + l
            const uint32_t idx = half * 64u + lane;
            const uint32_t n_classes = *(volatile uint32_t*)&wave_cnt[idx];
            if (n_classes) {
                const uint32_t tg = idx / 2u + 1u;
                bool kept;
                account(tg, (cnt_t)(((unsigned long long)n_classes) << P.cube_shift), depth + (idx & 1u), lamtab[tg - 1], kept);
            }
        }
    }
    if constexpr (!LOWER) for (;;) {
        const bool input = q.more || q.next < q.end;
        if (!input && count == 0) break;
#ifdef BSX_DIAG
        ++dbg_iters;
#endif
        uint32_t A[NW], base = 0, mlo = 0, mhi = 0, res = 0, hfull = 0;
        int32_t t = 0;
        bool live = false, deep_fresh = false;
        uint64_t fresh_pos = 0;
        if (count > kCap - kPoolGroup || !input) {
            // ---- pool stage: the oldest classes (their states were looked up when they were stored)
            const uint32_t n = count < 64u ? count : 64u;
            live = lane < n;
            uint32_t ri = head + lane;
            ri -= ri >= kCap ? kCap : 0u;
            const uint32_t r = ri * R;
            if constexpr (cube) {
                uint32_t packed;
                if constexpr (NW == 2) {
                    typedef volatile bsx_u32x4 __attribute__((address_space(3))) lds_v4;
                    const bsx_u32x4 v = *(lds_v4*)(pool + r);
                    A[0] = live ? v.x : 0u; A[1] = live ? v.y : 0u; mlo = v.z; packed = v.w;
                } else {
#pragma unroll
                    for (int w = 0; w < NW; ++w) A[w] = live ? pool[r + w] : 0u;
                    mlo = pool[r + NW]; packed = pool[r + NW + 1];
                }
                mhi = packed & 0x1FFFFu;
                t = (int32_t)(packed >> 17) - 256;
            } else if constexpr (NW % 2 == 0) {        // 8-byte accesses (records are 8-byte aligned for even NW)
                typedef volatile bsx_u32x2 __attribute__((address_space(3))) lds_v2;
                lds_v2* rec = (lds_v2*)(pool + r);
#pragma unroll
                for (int w = 0; w < NW; w += 2) { const bsx_u32x2 v = rec[w / 2]; A[w] = live ? v.x : 0u; A[w + 1] = live ? v.y : 0u; }
                const bsx_u32x2 b0 = rec[NW / 2], b1 = rec[NW / 2 + 1];
                base = b0.x; mlo = b0.y; mhi = b1.x; t = (int32_t)b1.y;
            } else {
#pragma unroll
                for (int w = 0; w < NW; ++w) A[w] = live ? pool[r + w] : 0u;
                base = pool[r + NW]; mlo = pool[r + NW + 1]; mhi = pool[r + NW + 2]; t = (int32_t)pool[r + NW + 3];
            }
            head += n;
            head -= head >= kCap ? kCap : 0u;
            count -= n;
#ifdef BSX_DIAG
            dbg_is_fresh = false;
            dbg_pool_in += n;
#endif
        } else {
            // ---- fresh stage: the next 64 consecutive problems (chunks start on multiples of 64)
            if (q.next == q.end) {
                BSX_RARE_PARAMS(Pr);
                const uint64_t b = first_dyn + grab_chunk(&Pr->ctr->cursor, chunk, (int)lane);
                if (b >= n_items) { q.more = false; continue; }
                q.next = b; q.end = (b + chunk < n_items) ? b + chunk : n_items;
            }
            const uint64_t avail = q.end - q.next;
            const uint32_t n = avail < 64u ? (uint32_t)avail : 64u;
            live = lane < n;
            base = (uint32_t)q.next;
            if constexpr (cube) {                           // q.next is a multiple of 64 and the class index starts at 0
                fresh_pos = q.next;
                fresh_state(fresh_pos, live, A);
            } else {
                // (lane made opaque: otherwise first_digits + lane is kept across the loop as a 64-bit per-lane value -- and spilled)
                uint32_t lane_here = lane;
                asm volatile("" : "+v"(lane_here));
                init_problem_simple<NW>(P.sp, q.next + lane_here, A);
            }
            if (counting) {                                 // (mlo, mhi) = 64-bit member count
                const unsigned long long members = 1ull << (cube ? P.cube_shift : 0u);
                mlo = (uint32_t)members; mhi = (uint32_t)(members >> 32);
            } else {                                        // (mlo, mhi) = member mask
                mlo = lane < 32u ? 1u << lane : 0u;
                mhi = lane < 32u ? 0u : 1u << (lane - 32u);
            }
            t = -(int32_t)tp;
            q.next += n;
            if constexpr (cube) {
                if (t0_lookup) {
                    // Is a member of this class itself a cycle state x0 (mu = 0)?  Its representative entry says so.
                    // Then the class is resolved on the spot: the common s(1) = f(x0) lies on x0's cycle, and no other
                    // member is a cycle state (two would share their successor), so everybody else has mu = 1.  Member
                    // counts are in units of 2^unit_shift problems (bsx_device.h), so the class is booked whole at mu = 1
                    // and the one member is moved to mu = 0 through the signed, absolute correction sums.
                    uint32_t h0;
                    const uint32_t et0 = live ? probe(A, h0, true) : 0u;
                    if (et0) {
                        const uint32_t lam0 = NW <= 2 ? hit_len : lamtab[(et0 & kTagMask) - 1];
                        bool kept;
                        uint32_t one = 1u;
                        unsigned long long m0 = ((unsigned long long)mhi << 32) | mlo;
                        asm volatile("" : "+v"(one), "+v"(m0));
                        account(et0, m0, one, lam0, kept);
                        account_fix(et0, 0u, lam0, 1ll);
                        account_fix(et0, 1u, lam0, -1ll);
                        live = false;
                    }
                }
            }
            if constexpr (cube) {
                if (depth > 1u) {
                    // deep pass: the members of a class share F^depth(x), not the states before it.  With a warm-up
                    // (depth <= T_p, the host sees to that) they share s(T_p) and so all that counts: nothing to list.
                    deep_fresh = !has_warmup;
                    for (uint32_t i = 1; i < depth; ++i) {
                        ++t;
                        if (live) {
                            uint32_t nx[NW];
                            net_step<NW, K>(nv, A, fm0, fv0, nx, has_fixed);
                            if (has_warmup) apply_perturbations<NW>(P.sp, (uint32_t)((int32_t)tp + t), 0ull, nx);      // (t <= 0 here)
#pragma unroll
                            for (int w = 0; w < NW; ++w) A[w] = nx[w];
                        }
                    }
                    if (live) nexec += depth - 1u;
                }
            }
#ifdef BSX_DIAG
            ++dbg_fresh;
            dbg_is_fresh = true;
#endif
        }

        // ---- one update per live, unresolved lane, then the lookup of the new state
        if (live && res == 0) {
            uint32_t nxt[NW];
            net_step<NW, K>(nv, A, fm0, fv0, nxt, has_fixed);
            ++t;
            ++nexec;
            if (has_warmup) {
                if (t <= 0) apply_perturbations<NW>(P.sp, (uint32_t)((int32_t)tp + t), 0ull, nxt);
            }
            uint32_t et = probe(nxt, hfull);
            if (has_warmup) et = t >= 0 ? et : 0u;          // states before T_p do not count
            // s(T_p) = s(0) itself may be a cycle state (mu = 0).  It is only looked up when s(1) is one --
            // a successor of a cycle state is a cycle state -- instead of for every fresh problem.
            if (!has_warmup && !cube && __ballot(et != 0 && t == 1)) {
                if (et != 0 && t == 1) {
                    uint32_t h0;
                    const uint32_t len1 = hit_len;
                    const uint32_t et0 = probe(A, h0);
                    if (et0) { et = et0; t = 0; } else hit_len = len1;
                }
            }
#pragma unroll
            for (int w = 0; w < NW; ++w) A[w] = nxt[w];
            res = et;
        }
        if constexpr (cube) {
            if (deep_fresh) {
                // F^depth(x) on a cycle: the members' entry times differ (<= depth) -- list the class for the level below
                const bool near = live && res != 0;
                const uint64_t nb = __ballot(near);
                if (nb) {
                    BSX_RARE_PARAMS(Pr);
                    uint32_t S0[NW];
                    fresh_state(fresh_pos, near, S0);
                    uint32_t at0 = 0;
                    if (lane == 0) at0 = atomicAdd((uint32_t*)(__attribute__((address_space(3))) uint32_t*)&lc[2], (uint32_t)__popcll(nb));    // the workgroup's own segment
                    const uint32_t at = __builtin_amdgcn_readfirstlane(at0) + rank_below(nb);
                    if (near && at < Pr->near_cap) {
                        uint32_t* seg = Pr->near + ((uint64_t)blockIdx.x * Pr->near_cap + at) * (NW + 1);
#pragma unroll
                        for (int w = 0; w < NW; ++w) seg[w] = S0[w];
                        seg[NW] = res & kTagMask;
                    }
                }
                if (near) { live = false; res = 0; }
            }
            if (from_entries) {
                // A class of an entry-based pass is settled by its first lookup.  Its parent was listed because the
                // parent's common state F^(depth+1)(x) is a cycle state; so either F^depth(x), just looked up, is one as
                // well -- the class is listed in turn (above) or, at depth 1, resolved with mu = 1 -- or every member enters
                // the parent's cycle with the next update: mu = depth + 1, nothing left to step.  Fresh classes all carry
                // one unit and share t, so the lanes of an outcome (tag, mu) are counted with a ballot and booked by one of
                // them -- and the iteration ends here: nothing of such a pass ever reaches the pool.
                const bool hit = live && res != 0;                          // (depth 1 only: deeper hits were listed)
                uint32_t otag = hit ? (res & kTagMask) : ptag;
                bool todo = live;
                uint64_t tb = __ballot(todo);
                while (tb) {
                    const int first = __builtin_ctzll(tb);
                    const uint32_t tg = (uint32_t)__builtin_amdgcn_readlane((int)otag, first);
                    const bool h0 = __builtin_amdgcn_readlane((int)(hit ? 1u : 0u), first) != 0;
                    const bool same = todo && otag == tg && hit == h0;
                    const uint64_t sb = __ballot(same);
                    if (lane == (uint32_t)first) {
                        bool kept;
                        account(tg, (cnt_t)((((unsigned long long)mhi << 32) | mlo) * (unsigned long long)__popcll(sb)),
                                (uint32_t)t + (h0 ? 0u : 1u), lamtab[tg - 1], kept);
                    }
                    todo = todo && !same;
                    tb &= ~sb;
                }
                continue;
            }
        }

        // ---- candidates for the merge post their lane id now; the slot is read back right away and used
        //      after the resolve block, whose arithmetic hides the two LDS round trips
        bool cand = live && res == 0 && t < fast_steps;
        // (sibling states differ in a few bits: the slot needs a mixing hash; 24-bit multiplies are full rate)
        uint32_t slot;
        if constexpr (cube) {
            // hash_state has folded the upper half of the hash into its low 16 bits, so one 24-bit multiply mixes
            // all of it: five instructions fewer than the two-multiply mix below and, on cube passes, 3 % fewer
            // updates (on plain tiles it was 2 % more, so they keep theirs)
            slot = (__umul24(hfull ^ (uint32_t)t, 0x9E3779u) >> 17) & (kPoolSlots - 1);
        } else {
            const uint32_t hx = hfull ^ (hfull >> 15) ^ (counting ? (uint32_t)t : base >> 6);
            slot = ((__umul24(hx, 0x9E3779u) ^ __umul24(hx >> 11, 0x85EBCBu)) >> 12) & (kPoolSlots - 1);
        }
        if (cand) dd_ids[slot] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t w = cand ? (uint32_t)dd_ids[slot] : lane;

        // ---- resolved classes: every member has mu = t
        cnt_t m;
        if constexpr (cube) m = ((unsigned long long)mhi << 32) | mlo;
        else m = counting ? mlo : (uint32_t)(__popc(mlo) + __popc(mhi));
        if (live && res != 0) {
            const uint32_t tg = res & kTagMask;
            const uint32_t lam = NW <= 2 ? hit_len : lamtab[tg - 1];
            bool keep;
            account(res, m, (uint32_t)t, lam, keep);
            if (!cube && P.per_problem) {
                ProblemRec32 r;
#pragma unroll
                for (int w = 0; w < kMaxW32; ++w) r.key[w] = 0;
                if (keep) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) r.key[w] = keytab[(tg - 1) * NW + w];
                }
                r.length = keep ? lam : 0; r.trajectory_l = keep ? tp + (uint32_t)t : 0; r.found = keep; r.pad = 0;
                for (uint32_t left = mlo; left; left &= left - 1) P.per_problem[base + (uint32_t)__builtin_ctz(left)] = r;
                for (uint32_t left = mhi; left; left &= left - 1) P.per_problem[base + 32u + (uint32_t)__builtin_ctz(left)] = r;
            }
        }
        // ---- classes past the FAST length go back as (group base, member mask)
        if (live && res == 0 && t >= fast_steps) {
            BSX_RARE_PARAMS(Pr);
            Counters* const c = Pr->ctr;
            uint32_t* const list = Pr->stragglers;
            const uint64_t list_cap = Pr->stragglers_cap;
            if (counting && !cube) atomicOr(&c->straggler_overflow, 2u);    // members unknown: the host repeats the tile
            atomicAdd(&c->n_stragglers, (unsigned long long)m);
            const unsigned long long at = atomicAdd(&c->straggler_classes, 1ull);
            if (cube) {
                // (state, t, member count): the host runs the detector from the state; an attractor it did not
                // know yet means the pass is repeated, otherwise the class simply ran past the time cap
                constexpr uint32_t kRec = NW + 3;
                if (kRec * (at + 1) <= list_cap) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) list[kRec * at + w] = A[w];
                    list[kRec * at + NW] = (uint32_t)t; list[kRec * at + NW + 1] = mlo; list[kRec * at + NW + 2] = mhi;
                } else atomicOr(&c->straggler_overflow, 1u);
            } else if (3 * at + 2 < list_cap) { list[3 * at] = base; list[3 * at + 1] = mlo; list[3 * at + 2] = mhi; }
            else atomicOr(&c->straggler_overflow, 1u);
        }

        // ---- merge lanes of one group that are in the same state (same group = same time)
        {
            // every lane takes part in the permutes (a lane masked off would deliver nothing to its readers)
            uint32_t differ = counting ? 0u : (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w * 4u), (int)base) ^ base;
            // the slot's owner must be a candidate itself and at the same time: one permute for both
            const uint32_t tc = ((uint32_t)t << 1) | (cand ? 0u : 1u);
            differ |= (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w * 4u), (int)tc) ^ ((uint32_t)t << 1);
#pragma unroll
            for (int i = 0; i < NW; ++i) differ |= (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w * 4u), (int)A[i]) ^ A[i];
            const bool same = cand & (w != lane) & (differ == 0u);
#ifdef BSX_DIAG
            dbg_merged += __popcll(__ballot(same));
#endif
            if (__ballot(same)) {
                if (same) {
                    if constexpr (cube) {
                        atomicAdd((unsigned long long*)(__attribute__((address_space(3))) unsigned long long*)&dd_acc[2 * w],
                                  ((unsigned long long)mhi << 32) | mlo);
                    } else if (counting) {
                        atomicAdd((uint32_t*)(__attribute__((address_space(3))) uint32_t*)&dd_acc[2 * w], mlo);
                    } else {
                        if (mlo) atomicOr((uint32_t*)(__attribute__((address_space(3))) uint32_t*)&dd_acc[2 * w], mlo);
                        if (mhi) atomicOr((uint32_t*)(__attribute__((address_space(3))) uint32_t*)&dd_acc[2 * w + 1], mhi);
                    }
                    cand = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (cand) {
                    uint32_t glo, ghi;
                    if constexpr (cube) {
                        typedef volatile bsx_u32x2 __attribute__((address_space(3))) lds_v2;
                        const bsx_u32x2 gv = *(lds_v2*)(dd_acc + 2 * lane);
                        glo = gv.x; ghi = gv.y;
                    } else { glo = dd_acc[2 * lane]; ghi = dd_acc[2 * lane + 1]; }
                    if (glo | ghi) {
                        if constexpr (cube) {
                            const unsigned long long sum = (((unsigned long long)mhi << 32) | mlo) + (((unsigned long long)ghi << 32) | glo);
                            mlo = (uint32_t)sum; mhi = (uint32_t)(sum >> 32);
                        } else if (counting) { mlo += glo; }
                        else { mlo |= glo; mhi |= ghi; }
                        dd_acc[2 * lane] = 0; dd_acc[2 * lane + 1] = 0;
                    }
                }
            }
        }

        // ---- survivors go (back) to the pool
        const uint64_t keepers = __ballot(cand);
#ifdef BSX_DIAG
        if (dbg_is_fresh) dbg_fresh_keep += __popcll(keepers); else dbg_pool_keep += __popcll(keepers);
#endif
        if (cand) {
            const uint32_t rank = rank_below(keepers);
            uint32_t tail = head + count;                   // uniform; head < cap, count <= cap
            tail -= tail >= kCap ? kCap : 0u;
            uint32_t ri = tail + rank;
            ri -= ri >= kCap ? kCap : 0u;
            const uint32_t r = ri * R;
            if constexpr (cube) {
                const uint32_t packed = mhi | ((uint32_t)(t + 256) << 17);
                if constexpr (NW == 2) {
                    typedef volatile bsx_u32x4 __attribute__((address_space(3))) lds_v4;
                    bsx_u32x4 v;
                    v.x = A[0]; v.y = A[1]; v.z = mlo; v.w = packed;
                    *(lds_v4*)(pool + r) = v;
                } else {
#pragma unroll
                    for (int w = 0; w < NW; ++w) pool[r + w] = A[w];
                    pool[r + NW] = mlo; pool[r + NW + 1] = packed;
                }
            } else if constexpr (NW % 2 == 0) {
                typedef volatile bsx_u32x2 __attribute__((address_space(3))) lds_v2;
                lds_v2* rec = (lds_v2*)(pool + r);
#pragma unroll
                for (int w = 0; w < NW; w += 2) { bsx_u32x2 v; v.x = A[w]; v.y = A[w + 1]; rec[w / 2] = v; }
                bsx_u32x2 b0, b1;
                b0.x = base; b0.y = mlo; b1.x = mhi; b1.y = (uint32_t)t;
                rec[NW / 2] = b0; rec[NW / 2 + 1] = b1;
            } else {
#pragma unroll
                for (int w = 0; w < NW; ++w) pool[r + w] = A[w];
                pool[r + NW] = base; pool[r + NW + 1] = mlo; pool[r + NW + 2] = mhi; pool[r + NW + 3] = (uint32_t)t;
            }
        }
        count += (uint32_t)__popcll(keepers);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- epilogue: workgroup accumulators -> one log record per attractor and workgroup
    __syncthreads();
#ifdef BSX_DIAG
    const unsigned long long dbg_t2 = wall_clock64();
#endif
    // (wave 0: lane a owns accumulator a; the workgroup's records take one step of the log cursor)
    static_assert(kAccs == 64, "one lane per accumulator");
    if (wave == 0) {
        const uint32_t a = lane;
        const unsigned long long cn = acc_cnt[a];
        const uint64_t nb = __ballot(cn != 0);
        if (nb) {
            unsigned long long at0 = 0;
            if (!cube && lane == 0) at0 = atomicAdd(&P.ctr->log_cursor, (unsigned long long)__popcll(nb));
            const unsigned long long at = bcast64(at0, 0) + rank_below(nb);
            if (cn) {
                const unsigned long long sl = acc_sl[a];
                if constexpr (cube) {
                    // (one global atomic per sum: 64 addresses, a few hundred adds each per launch)
                    Counters* c = P.ctr;
                    atomicAdd(&c->acc_cnt[a], cn);
                    atomicAdd(&c->acc_sl[a], sl);
                    const unsigned long long lo = acc_sl2[a];
                    const unsigned long long old = atomicAdd(&c->acc_sl2_lo[a], lo);
                    const unsigned long long up = acc_sl2h[a] + ((old + lo < old) ? 1ull : 0ull);
                    if (up) atomicAdd(&c->acc_sl2_hi[a], up);
                    c->acc_len[a] = lamtab[a];                      // (every workgroup stores the same values)
#pragma unroll
                    for (int w = 0; w < NW; ++w) c->acc_key[a][w] = keytab[a * NW + w];
                } else if (at < P.log_cap) {
                    LogRec r;
#pragma unroll
                    for (int w = 0; w < kMaxW32; ++w) r.key[w] = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) r.key[w] = keytab[a * NW + w];
                    r.length = lamtab[a]; r.pad = 0; r.count = cn; r.sum_l = sl; r.sum_l2_lo = acc_sl2[a]; r.sum_l2_hi = acc_sl2h[a];
                    P.log[at] = r;
                } else {
                    atomicOr(&P.ctr->log_overflow, 1u);
                }
                atomicAdd(&wg_ctr[2], sl + cn * lamtab[a]);                         // + lambda each (model.py:201)
            }
        }
    }
#ifdef BSX_DIAG
    if (lane == 0) {
        atomicAdd(&P.ctr->wave_iters, dbg_iters); atomicAdd(&P.ctr->service_rounds, dbg_fresh);
        atomicAdd(&P.ctr->diag[0], dbg_fresh_keep); atomicAdd(&P.ctr->diag[1], dbg_pool_in);
        atomicAdd(&P.ctr->diag[2], dbg_pool_keep); atomicAdd(&P.ctr->diag[3], dbg_merged);
    }
    if (threadIdx.x == 0) {
        const unsigned long long dbg_t3 = wall_clock64();
        atomicAdd(&P.ctr->phase_sum[0], dbg_t1 - dbg_t0); atomicMax(&P.ctr->phase_max[0], dbg_t1 - dbg_t0);
        atomicAdd(&P.ctr->phase_sum[1], dbg_t2 - dbg_t1); atomicMax(&P.ctr->phase_max[1], dbg_t2 - dbg_t1);
        atomicAdd(&P.ctr->phase_sum[2], dbg_t3 - dbg_t2); atomicMax(&P.ctr->phase_max[2], dbg_t3 - dbg_t2);
    }
#endif
    // counters: summed over the workgroup in LDS first (the accumulators are free now), one global atomic each
    {
        const unsigned long long w_exec = wave_sum((unsigned long long)nexec);
        if (lane == 0 && w_exec) atomicAdd(&wg_ctr[3], w_exec);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if constexpr (cube) {
            if (P.near_counts) {
                const uint32_t listed = lc[2];
                P.near_counts[blockIdx.x] = listed;
                if (listed) atomicAdd(&P.ctr->near_classes, (unsigned long long)listed);
                if (listed > P.near_cap) atomicOr(&P.ctr->near_overflow, 1u);
            }
        }
        const unsigned long long ref = wg_ctr[2] + (P.cap_rel_inf ? 0ull : wg_ctr[1] * P.max_t);
        if (ref) atomicAdd(&P.ctr->steps_ref, ref);
        if (wg_ctr[3]) atomicAdd(&P.ctr->steps_exec, wg_ctr[3]);
        if (wg_ctr[0]) atomicAdd(&P.ctr->n_none, wg_ctr[0]);
        if constexpr (cube) atomicMax(&P.ctr->t_last, (unsigned long long)wall_clock64());
    }
}

// ------------------------------------------------------------------------------------------------
// k_digit_lifetimes (bsx_device.h: LifetimeParams): ordering heuristic of the cube collapse.
template <int NW, int K, int LM>
__global__ __launch_bounds__(1024) void k_digit_lifetimes(const LifetimeParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
    const NetView<NW, K, LM> nv = stage_network<NW, K, LM>(P.net, smem, smem_free);
    const uint32_t digit = threadIdx.x / kLifeTrials, trial = threadIdx.x % kLifeTrials;
    if (digit >= P.n_digits) return;
    uint32_t fm[NW], fv[NW], x[NW], y[NW];
    uint32_t any_fixed = 0;
    const uint32_t node = P.node[digit];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        fm[w] = P.fixmask[w]; fv[w] = P.fixval[w]; any_fixed |= fm[w];
        uint32_t rnd = (trial * 0x9E3779B1u + (uint32_t)w * 0x85EBCA6Bu + 0x27D4EB2Fu) * 0xC2B2AE35u;
        rnd ^= rnd >> 15; rnd *= 0x2C1B3C6Du; rnd ^= rnd >> 12;
        const uint32_t bit = ((node >> 5) == (uint32_t)w) ? 1u << (node & 31) : 0u;
        x[w] = (P.base[w] | (trial ? rnd & P.free_mask[w] : 0u)) & ~bit;       // trial 0: every other free digit 0
        y[w] = x[w] | bit;
    }
    uint32_t t = 0;
    while (t < kLifeSteps && !eq_words<NW>(x, y)) {
        uint32_t nx[NW], ny[NW];
        net_step<NW, K>(nv, x, fm, fv, nx, any_fixed != 0);
        net_step<NW, K>(nv, y, fm, fv, ny, any_fixed != 0);
        copy_words<NW>(x, nx); copy_words<NW>(y, ny);
        ++t;
    }
    atomicAdd(&P.out[digit], t);
}


template <int NW, int K>
static hipError_t launch_life_nk(int lut_mode, dim3, size_t shmem, hipStream_t st, const LifetimeParams& P) {
    const void* fn;
    BSX_KERNEL_FOR_MODE(k_digit_lifetimes, NW, K, lut_mode, fn);
    if (!fn) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    void* args[] = {const_cast<LifetimeParams*>(&P)};
    return hipLaunchKernel(fn, dim3(1), dim3(1024), args, shmem, st);
}

template <int NW, int K>
static const void* pool_kernel_for(int lut_mode, bool cube, bool lower = false) {
    const void* fn = nullptr;
    if (lower) {
        if (lut_mode == kLutLdsByte) fn = (const void*)k_attract_pool<NW, K, kLutLdsByte, true, true>;
        else if (lut_mode == kLutGlobal) fn = (const void*)k_attract_pool<NW, K, kLutGlobal, true, true>;
        else if constexpr (NW >= 4) { if (lut_mode == kLutLdsNibble) fn = (const void*)k_attract_pool<NW, K, kLutLdsNibble, true, true>; }
    } else if (cube) {
        if (lut_mode == kLutLdsByte) fn = (const void*)k_attract_pool<NW, K, kLutLdsByte, true>;
        else if (lut_mode == kLutGlobal) fn = (const void*)k_attract_pool<NW, K, kLutGlobal, true>;
        else if constexpr (NW >= 4) { if (lut_mode == kLutLdsNibble) fn = (const void*)k_attract_pool<NW, K, kLutLdsNibble, true>; }
    } else {
        if (lut_mode == kLutLdsByte) fn = (const void*)k_attract_pool<NW, K, kLutLdsByte, false>;
        else if (lut_mode == kLutGlobal) fn = (const void*)k_attract_pool<NW, K, kLutGlobal, false>;
        else if constexpr (NW >= 4) { if (lut_mode == kLutLdsNibble) fn = (const void*)k_attract_pool<NW, K, kLutLdsNibble, false>; }
    }
    return fn;
}
template <int NW, int K>
static hipError_t launch_pool_nk(int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    const void* fn = pool_kernel_for<NW, K>(lut_mode, P.merge == 3, P.merge == 3 && P.lower_build != 0);
    if (!fn) return hipErrorInvalidValue;
    void* args[] = {const_cast<AttractParams*>(&P)};
    return hipLaunchKernel(fn, grid, dim3(kPoolBlock), args, shmem, st);
}
template <int NW, int K>
static hipError_t configure_pool_nk(int lut_mode, dim3, size_t shmem, hipStream_t, int& blocks_per_cu) {
    for (int cube = 0; cube < 3; ++cube) {                  // plain, cube, lower-level cube
        const void* fn = pool_kernel_for<NW, K>(lut_mode, cube != 0, cube == 2);
        if (!fn) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        if (!cube) { e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fn, kPoolBlock, shmem); if (e != hipSuccess) return e; }
    }
    return hipSuccess;
}

}  // namespace bsx

// One translation unit per state width -- two for the wide states, whose instantiations take minutes to compile: odd and
// even numbers of predecessor slots (BSX_POOL_KMASK, bit k = instantiate K = k).  BSX_POOL_TU(2, ) defines launch_pool_nw2 /
// configure_pool_nw2 / launch_life_nw2; BSX_POOL_TU(8, a) launch_pool_nw8a ...
#ifndef BSX_POOL_KMASK
#define BSX_POOL_KMASK 0x7E
#endif
#if BSX_POOL_KMASK & (1 << 1)
#define BSX_PK1(FN, NWV) case 1: return FN<NWV, 1>(lut_mode, grid, shmem, st, P);
#else
#define BSX_PK1(FN, NWV)
#endif
#if BSX_POOL_KMASK & (1 << 2)
#define BSX_PK2(FN, NWV) case 2: return FN<NWV, 2>(lut_mode, grid, shmem, st, P);
#else
#define BSX_PK2(FN, NWV)
#endif
#if BSX_POOL_KMASK & (1 << 3)
#define BSX_PK3(FN, NWV) case 3: return FN<NWV, 3>(lut_mode, grid, shmem, st, P);
#else
#define BSX_PK3(FN, NWV)
#endif
#if BSX_POOL_KMASK & (1 << 4)
#define BSX_PK4(FN, NWV) case 4: return FN<NWV, 4>(lut_mode, grid, shmem, st, P);
#else
#define BSX_PK4(FN, NWV)
#endif
#if BSX_POOL_KMASK & (1 << 5)
#define BSX_PK5(FN, NWV) case 5: return FN<NWV, 5>(lut_mode, grid, shmem, st, P);
#else
#define BSX_PK5(FN, NWV)
#endif
#if BSX_POOL_KMASK & (1 << 6)
#define BSX_PK6(FN, NWV) case 6: return FN<NWV, 6>(lut_mode, grid, shmem, st, P);
#else
#define BSX_PK6(FN, NWV)
#endif
#define BSX_POOL_DISPATCH_K(FN, NWV)                                                                                  \
    switch (k) {                                                                                                      \
        BSX_PK1(FN, NWV) BSX_PK2(FN, NWV) BSX_PK3(FN, NWV) BSX_PK4(FN, NWV) BSX_PK5(FN, NWV) BSX_PK6(FN, NWV)          \
        default: return hipErrorInvalidValue;                                                                         \
    }

#define BSX_POOL_TU(NWV, TAG)                                                                                         \
    namespace bsx {                                                                                                   \
    hipError_t launch_pool_nw##NWV##TAG(int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) { \
        BSX_POOL_DISPATCH_K(launch_pool_nk, NWV)                                                                      \
    }                                                                                                                 \
    hipError_t configure_pool_nw##NWV##TAG(int k, int lut_mode, size_t shmem, int* blocks_per_cu) {                   \
        const dim3 grid(1);                                                                                           \
        const hipStream_t st = nullptr;                                                                               \
        int& P = *blocks_per_cu;                                                                                      \
        BSX_POOL_DISPATCH_K(configure_pool_nk, NWV)                                                                   \
    }                                                                                                                 \
    hipError_t launch_life_nw##NWV##TAG(int k, int lut_mode, size_t shmem, hipStream_t st, const LifetimeParams& P) { \
        const dim3 grid(1);                                                                                           \
        BSX_POOL_DISPATCH_K(launch_life_nk, NWV)                                                                      \
    }                                                                                                                 \
    }
