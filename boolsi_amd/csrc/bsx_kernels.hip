// gfx950 kernels of the Boolean-network state-update engine.
//
// Work decomposition: one LANE = one trajectory, state packed in NW 32-bit registers.
// A step gathers every node's predecessor bits with byte-indexed LUTs staged in LDS (one lookup per
// 8 state bits yields the gathered words of all K predecessor slots at once) and evaluates all
// truth tables of the network together as a K-level v_bfi mux tree over bit-packed table masks.
// Trajectories end at data-dependent times, so waves are persistent: a lane whose problem is
// resolved immediately takes the next problem index of its wave's chunk (wave-level dequeue with
// __ballot/__popcll), keeping all 64 lanes busy.  Cycle detection is Brent's algorithm per lane;
// the minimum state code of the cycle (the attractor key) is tracked during the detection lap, the
// trajectory length mu comes from a lagged two-pointer pass.  Results are aggregated in registers
// (per-lane run of equal keys), then in a per-wave table (one slot per lane, matched with ballots),
// then appended to a log in HBM.
//
// Replaces (reference file:line): apply_update_rules model.py:16-28, fixed nodes 31-49, simulate_step
// 52-73, warm-up 76-128, detection loop 152-236, solvers attract.py:262-302 / target.py:109-133 /
// simulate.py:97-131, store_attractor attract.py:374-402, problem enumeration batching.py:160-282.
#include <hip/hip_runtime.h>
#include "bsx_device.h"

namespace bsx {

// (a & sel) | (b & ~sel) in one VALU op.  Written as asm because with loop-invariant a/b the compiler
// prefers two ops (and + xor with a precomputed a^b), which doubles the cost of the mux tree.
__device__ __forceinline__ uint32_t bfi(uint32_t sel, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(sel), "v"(a), "v"(b));
    return r;
}

template <int NW>
__device__ __forceinline__ bool eq_words(const uint32_t (&a)[NW], const uint32_t (&b)[NW]) {
    uint32_t d = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) d |= a[w] ^ b[w];
    return d == 0;
}

// a < b as big integers, word NW-1 most significant (state code order, model.py:131-149)
template <int NW>
__device__ __forceinline__ bool lt_words(const uint32_t (&a)[NW], const uint32_t (&b)[NW]) {
    if constexpr (NW == 1) return a[0] < b[0];
    if constexpr (NW == 2)
        return (((uint64_t)a[1] << 32) | a[0]) < (((uint64_t)b[1] << 32) | b[0]);     // v_cmp_lt_u64
    bool lt = false, eq = true;
#pragma unroll
    for (int w = NW - 1; w >= 0; --w) {
        lt = lt || (eq && a[w] < b[w]);
        eq = eq && (a[w] == b[w]);
    }
    return lt;
}

template <int NW>
__device__ __forceinline__ void copy_words(uint32_t (&dst)[NW], const uint32_t (&src)[NW]) {
#pragma unroll
    for (int w = 0; w < NW; ++w) dst[w] = src[w];
}

// Bit access by (wave-uniform or per-lane) node number.  Written as mask arithmetic over ALL words so
// that the state arrays keep static indices and stay in registers (a select over array elements gets
// turned into a dynamically indexed access by the compiler, which would push the arrays to scratch).
template <int NW>
__device__ __forceinline__ uint32_t get_bit(const uint32_t (&s)[NW], uint32_t node) {
    const uint32_t m = 1u << (node & 31);
    uint32_t acc = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) acc |= s[w] & (((node >> 5) == (uint32_t)w) ? m : 0u);
    return acc ? 1u : 0u;
}

template <int NW>
__device__ __forceinline__ void put_bit(uint32_t (&s)[NW], uint32_t node, uint32_t v) {
    const uint32_t m = 1u << (node & 31);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t mw = ((node >> 5) == (uint32_t)w) ? m : 0u;
        s[w] = (s[w] & ~mw) | (v ? mw : 0u);
    }
}

// ------------------------------------------------------------------------------------------------
// Network tables as seen by a workgroup: LUT and masks either in LDS or (large networks) in HBM/L2.
template <int NW, int K>
struct NetView {
    static constexpr bool kMasksInRegs = (K <= 3);
    const uint32_t* lut;     // LDS or global
    const uint32_t* masks;   // LDS (used when the 2^K * NW mask words do not fit the register budget)
    uint32_t mreg[kMasksInRegs ? (1 << K) * NW : 1];
    uint32_t n_wide;
    const uint32_t* wide_desc;
    const uint32_t* wide_preds;
    const uint32_t* wide_tt;
};

// Load one LUT entry (N consecutive words) with the widest loads its size allows.
template <int N>
__device__ __forceinline__ void load_entry(const uint32_t* e, uint32_t (&dst)[N]) {
    if constexpr (N % 4 == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(__builtin_assume_aligned(e, 16));
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const uint4 v = p[i];
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
        }
    } else if constexpr (N % 2 == 0) {
        const uint2* p = reinterpret_cast<const uint2*>(__builtin_assume_aligned(e, 8));
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const uint2 v = p[i];
            dst[2 * i] = v.x; dst[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) dst[i] = e[i];
    }
}

// One synchronous update of all nodes (model.py:16-28) + fixed nodes as constants (model.py:31-49).
template <int NW, int K>
__device__ __forceinline__ void net_step(const NetView<NW, K>& nv, const uint32_t (&s)[NW],
                                         const uint32_t (&fm)[NW], const uint32_t (&fv)[NW],
                                         uint32_t (&out)[NW]) {
    uint32_t g[K][NW];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int w = 0; w < NW; ++w) g[j][w] = 0;

    // gather: one LUT entry per 8 state bits.  All NW*4 lookups are issued unconditionally (the LUT
    // is zero-padded to whole 32-bit words of state) so the LDS reads overlap instead of each
    // waiting behind a branch.
    // Lookups are issued in batches sized to keep the in-flight entries within ~64 registers.
    constexpr int kEntry = K * NW;
    constexpr int kBatch = (64 / kEntry) < 1 ? 1 : ((64 / kEntry) > NW * 4 ? NW * 4 : (64 / kEntry));
#pragma unroll
    for (int c0 = 0; c0 < NW * 4; c0 += kBatch) {
        uint32_t e[kBatch][kEntry];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int ch = c0 + b;
            if (ch < NW * 4) {
                const uint32_t v = (s[ch >> 2] >> ((ch & 3) * 8)) & 0xFFu;
                load_entry<kEntry>(nv.lut + ((uint32_t)(ch << 8) + v) * kEntry, e[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b)
            if (c0 + b < NW * 4) {
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int w = 0; w < NW; ++w) g[j][w] |= e[b][j * NW + w];
            }
    }

    // mux tree over the bit-packed truth-table masks: level j selects on predecessor slot j
    uint32_t r[1 << (K - 1)][NW];
#pragma unroll
    for (int i = 0; i < (1 << (K - 1)); ++i)
#pragma unroll
        for (int w = 0; w < NW; ++w)
            r[i][w] = NetView<NW, K>::kMasksInRegs
                          ? bfi(g[0][w], nv.mreg[(2 * i + 1) * NW + w], nv.mreg[(2 * i) * NW + w])
                          : bfi(g[0][w], nv.masks[(2 * i + 1) * NW + w], nv.masks[(2 * i) * NW + w]);
#pragma unroll
    for (int j = 1; j < K; ++j)
#pragma unroll
        for (int i = 0; i < (1 << (K - 1 - j)); ++i)
#pragma unroll
            for (int w = 0; w < NW; ++w) r[i][w] = bfi(g[j][w], r[2 * i + 1][w], r[2 * i][w]);
#pragma unroll
    for (int w = 0; w < NW; ++w) out[w] = r[0][w];

    // nodes with more than kMaxMuxK predecessors: explicit table lookup
    for (uint32_t q = 0; q < nv.n_wide; ++q) {
        const uint32_t node = nv.wide_desc[4 * q], k = nv.wide_desc[4 * q + 1];
        const uint32_t* preds = nv.wide_preds + nv.wide_desc[4 * q + 2];
        const uint32_t* tt = nv.wide_tt + nv.wide_desc[4 * q + 3];
        uint32_t idx = 0;
        for (uint32_t j = 0; j < k; ++j) idx |= get_bit<NW>(s, preds[j]) << j;
        const uint32_t bit = (tt[idx >> 5] >> (idx & 31)) & 1u;
        put_bit<NW>(out, node, bit);
    }

#pragma unroll
    for (int w = 0; w < NW; ++w) out[w] = (out[w] & ~fm[w]) | fv[w];
}

// ------------------------------------------------------------------------------------------------
// Problem enumeration (batching.py:160-229): offset p within the run -> initial state, fixed-node
// masks, perturbation-variation digits, last perturbation time.
template <int NW>
struct Problem {
    uint32_t s[NW];
    uint32_t fm[NW];
    uint32_t fv[NW];
    uint64_t pv_digits;   // 2 bits per perturbation variation
    uint32_t tp;
};

__device__ __forceinline__ int digit_state(uint32_t range, uint32_t digit) {
    // batching.py:171-175; -1 = absent
    if (range == 0) return digit ? 0 : -1;
    if (range == 1) return digit ? 1 : -1;
    if (range == 2) return digit ? 1 : 0;
    return digit == 0 ? -1 : (digit == 1 ? 0 : 1);
}

template <int NW>
__device__ __forceinline__ void init_problem(const DevSpace& sp, uint64_t p, Problem<NW>& pr) {
    // digits = first_digits + p; what spills over bit n_any goes to the variant number
    uint64_t d[5];
    unsigned long long carry = p;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned long long a = sp.first_digits[w];
        const unsigned long long sum = a + carry;
        carry = (sum < a) ? 1ull : 0ull;
        d[w] = sum;
    }
    d[4] = carry;
    const uint32_t sw = sp.n_any >> 6, sb = sp.n_any & 63;
    uint64_t lo = d[0], hi = d[1];   // words sw, sw+1
#pragma unroll
    for (int w = 1; w < 5; ++w) {
        lo = (sw == (uint32_t)w) ? d[w] : lo;
        hi = (sw + 1 == (uint32_t)w) ? d[w] : hi;
    }
    if (sw >= 4) hi = 0;
    uint64_t over = sb ? ((lo >> sb) | (hi << (64 - sb))) : lo;
    uint64_t variant = sp.first_variant + over;
    // keep only the n_any digits
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if ((uint32_t)w > sw) d[w] = 0;
        else if ((uint32_t)w == sw) d[w] = sb ? (d[w] & ((1ull << sb) - 1)) : 0;
    }

#pragma unroll
    for (int w = 0; w < NW; ++w) {
        pr.s[w] = sp.origin[w];
        pr.fm[w] = sp.fixmask[w];
        pr.fv[w] = sp.fixval[w];
    }
    if (sp.identity_any) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint64_t word = d[w >> 1];
            pr.s[w] |= (uint32_t)((w & 1) ? (word >> 32) : word);
        }
    } else {
        uint64_t sh[4] = {d[0], d[1], d[2], d[3]};
        for (uint32_t j = 0; j < sp.n_any; ++j) {
            put_bit<NW>(pr.s, sp.any_nodes[j], (uint32_t)(sh[0] & 1));
            sh[0] = (sh[0] >> 1) | (sh[1] << 63);
            sh[1] = (sh[1] >> 1) | (sh[2] << 63);
            sh[2] = (sh[2] >> 1) | (sh[3] << 63);
            sh[3] >>= 1;
        }
    }
    for (uint32_t j = 0; j < sp.n_fv; ++j) {
        const uint32_t node = sp.fv[2 * j], range = sp.fv[2 * j + 1];
        uint32_t digit;
        if (range == 3) { digit = (uint32_t)(variant % 3); variant /= 3; }
        else { digit = (uint32_t)(variant & 1); variant >>= 1; }
        const int st = digit_state(range, digit);
        if (st >= 0) { put_bit<NW>(pr.fm, node, 1); put_bit<NW>(pr.fv, node, (uint32_t)st); }
    }
    pr.pv_digits = 0;
    pr.tp = sp.tp_origin;
    for (uint32_t j = 0; j < sp.n_pv; ++j) {
        const uint32_t t = sp.pv[3 * j], range = sp.pv[3 * j + 2];
        uint32_t digit;
        if (range == 3) { digit = (uint32_t)(variant % 3); variant /= 3; }
        else { digit = (uint32_t)(variant & 1); variant >>= 1; }
        pr.pv_digits |= (uint64_t)digit << (2 * j);
        if (digit_state(range, digit) >= 0 && t > pr.tp) pr.tp = t;     // model.py:125
    }
}

// Perturbation override after the rules at time t (model.py:68-71): origin schedule, then the
// problem's variation entries (which win over an origin entry of the same (t, node), batching.py:198-207).
template <int NW>
__device__ __forceinline__ void apply_perturbations(const DevSpace& sp, uint32_t t, uint64_t pv_digits,
                                                    uint32_t (&s)[NW]) {
    if (t <= sp.tp_origin) {
#pragma unroll
        for (int w = 0; w < NW; ++w)
            s[w] = (s[w] & ~sp.sched_clr[t * NW + w]) | sp.sched_set[t * NW + w];
    }
    for (uint32_t j = 0; j < sp.n_pv; ++j) {
        if (sp.pv[3 * j] == t) {
            const int st = digit_state(sp.pv[3 * j + 2], (uint32_t)(pv_digits >> (2 * j)) & 3u);
            if (st >= 0) put_bit<NW>(s, sp.pv[3 * j + 1], (uint32_t)st);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Workgroup prologue: stage LUT + masks into LDS.
template <int NW, int K, bool LDS_LUT>
__device__ __forceinline__ NetView<NW, K> stage_network(const DevNet& net, uint32_t* smem, uint32_t*& smem_free) {
    NetView<NW, K> nv;
    uint32_t* p = smem;
    uint32_t* smasks = p;
    const uint32_t n_masks = (1u << K) * NW;
    for (uint32_t i = threadIdx.x; i < n_masks; i += blockDim.x) smasks[i] = net.masks[i];
    p += (n_masks + 3u) & ~3u;
    if (LDS_LUT) {
        uint32_t* slut = p;
        const uint4* src = reinterpret_cast<const uint4*>(net.lut);
        uint4* dst = reinterpret_cast<uint4*>(slut);
        const uint32_t n4 = net.lut_words >> 2;     // lut_words is a multiple of 4 (256 entries per chunk)
        for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
        p += net.lut_words;
        nv.lut = slut;
    } else {
        nv.lut = net.lut;
    }
    __syncthreads();
    nv.masks = smasks;
    if constexpr (NetView<NW, K>::kMasksInRegs) {
#pragma unroll
        for (int i = 0; i < (1 << K) * NW; ++i) nv.mreg[i] = smasks[i];
    } else {
        nv.mreg[0] = 0;
    }
    nv.n_wide = net.n_wide;
    nv.wide_desc = net.wide_desc;
    nv.wide_preds = net.wide_preds;
    nv.wide_tt = net.wide_tt;
    smem_free = p;
    return nv;
}

__device__ __forceinline__ uint64_t bcast64(uint64_t v, int src_lane) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src_lane);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}

// ------------------------------------------------------------------------------------------------
// Per-wave attractor table: slot i lives in lane i's registers, so a probe is one compare per lane
// and two ballots (attract.py:374-402 store_attractor, integer sums instead of Chan's float update).
template <int NW>
struct TableSlot {
    uint32_t key[NW];
    uint32_t length;
    uint32_t count;
    uint64_t sum_l;
    uint64_t sum_l2;
};

template <int NW>
__device__ __forceinline__ void log_append(const AttractParams& P, const uint32_t (&key)[NW], uint32_t length,
                                           uint32_t count, uint64_t sl, uint64_t sl2) {
    const unsigned long long at = atomicAdd(&P.ctr->log_cursor, 1ull);
    if (at < P.log_cap) {
        LogRec r;
#pragma unroll
        for (int w = 0; w < kMaxW32; ++w) r.key[w] = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) r.key[w] = key[w];
        r.length = length; r.count = count; r.sum_l = sl; r.sum_l2 = sl2;
        P.log[at] = r;
    } else {
        atomicOr(&P.ctr->log_overflow, 1u);
    }
}

// Merge the records of all lanes flagged in `want` into the wave's table (wave-uniform loop).
template <int NW>
__device__ __forceinline__ void table_merge(const AttractParams& P, TableSlot<NW>& slot, int lane, bool want,
                                            const uint32_t (&key)[NW], uint32_t length, uint32_t count,
                                            uint64_t sl, uint64_t sl2) {
    uint64_t todo = __ballot(want);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        uint32_t k[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) k[w] = __builtin_amdgcn_readlane(key[w], src);
        const uint32_t len = __builtin_amdgcn_readlane(length, src);
        const uint32_t cnt = __builtin_amdgcn_readlane(count, src);
        const uint64_t a = bcast64(sl, src), b = bcast64(sl2, src);
        // a slot whose 64-bit sum of squares is nearly full stops matching; the record then opens
        // another slot (or goes to the log) and the host merge adds them up in 128 bits
        const bool same = slot.count != 0 && eq_words<NW>(slot.key, k) && slot.sum_l2 < (1ull << 62);
        const uint64_t hit = __ballot(same);
        const uint64_t empty = __ballot(slot.count == 0);
        if (hit) {
            if (lane == __builtin_ctzll(hit)) { slot.count += cnt; slot.sum_l += a; slot.sum_l2 += b; }
        } else if (empty) {
            if (lane == __builtin_ctzll(empty)) {
                copy_words<NW>(slot.key, k);
                slot.length = len; slot.count = cnt; slot.sum_l = a; slot.sum_l2 = b;
            }
        } else if (lane == 0) {
            log_append<NW>(P, k, len, cnt, a, b);    // all 64 slots taken: straight to the HBM log
        }
    }
}

// Lane phases.  PH_DONE = result computed, waiting for the wave's next service round.
enum Phase : uint32_t { PH_IDLE = 0, PH_DONE = 1, PH_WARM = 2, PH_FAST = 3, PH_BRENT = 4, PH_ADVANCE = 5, PH_MU = 6 };

// Results are recorded and free lanes refilled in "service rounds", entered when at least this many
// lanes of the wave are waiting.  Enumeration (index -> problem) and aggregation cost several network
// updates, so they are run for many lanes at once instead of whenever one lane finishes.
constexpr uint32_t kServiceLanes = 12;

// Wave-level dequeue of problem offsets.
struct WaveQueue {
    uint64_t next, end;
    bool more;
};

__device__ __forceinline__ uint64_t grab_chunk(unsigned long long* cursor, uint32_t chunk, int lane) {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(cursor, (unsigned long long)chunk);
    return bcast64(base, 0);
}

// ------------------------------------------------------------------------------------------------
// Cycle-state cache (bsx_device.h).  LDS mirror entry: [state NW][tag][length][key NW] padded to a
// multiple of 4 words, so the probe of a lookup is one aligned 16-byte read for NW <= 2.
template <int NW>
struct CacheLayout {
    static constexpr int kStride = ((2 * NW + 2) + 3) & ~3;
};

template <int NW>
__device__ __forceinline__ uint32_t hash_state(const uint32_t (&s)[NW]) {
    uint32_t h = s[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        constexpr int kRot[8] = {0, 5, 10, 15, 20, 25, 3, 8};      // all in 1..31: no out-of-range shift
        h ^= (s[w] << kRot[w]) | (s[w] >> (32 - kRot[w]));
    }
    h ^= h >> 16;
    h ^= h >> 8;
    return h;
}

// LDS mirror: word 0 of the header = number of attractors whose states are all inserted ("visible");
// an entry's tag is the 1-based sequence number of its attractor and counts only when <= visible.
constexpr int kCacheHeaderWords = 4;

// One probe: loads the whole entry with no control flow in between (so the reads are issued together
// with whatever else the caller has in flight) and classifies it.
template <int NW>
struct CacheProbe {
    uint32_t tag, length;
    uint32_t key[NW];
    bool same;                  // entry holds exactly this state
};

template <int NW>
__device__ __forceinline__ CacheProbe<NW> cache_probe(const uint32_t* base, uint32_t h, const uint32_t (&s)[NW]) {
    constexpr int S = CacheLayout<NW>::kStride;
    const uint32_t* e = base + h * S;
    CacheProbe<NW> p;
    if constexpr (NW == 1) {
        const uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(e, 16));
        p.tag = v.y; p.length = v.z; p.key[0] = v.w; p.same = v.x == s[0];
    } else if constexpr (NW == 2) {
        const uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(e, 16));
        const uint2 k = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(e + 4, 8));
        p.tag = v.z; p.length = v.w; p.key[0] = k.x; p.key[1] = k.y;
        p.same = v.x == s[0] && v.y == s[1];
    } else {
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { d |= e[w] ^ s[w]; p.key[w] = e[NW + 2 + w]; }
        p.tag = e[NW]; p.length = e[NW + 1]; p.same = d == 0;
    }
    return p;
}

// Is `s` a state of an attractor with sequence number <= visible?  The first probe is branch-free;
// only lanes that land on another attractor's (or a not yet visible) entry keep walking the chain
// (the table is at most half full, so a walk ends at an empty tag).
template <int NW>
__device__ __forceinline__ bool cache_lookup(const uint32_t* lc, uint32_t mask, uint32_t visible,
                                             const uint32_t (&s)[NW], uint32_t& length, uint32_t (&key)[NW],
                                             uint32_t* tag_out = nullptr) {
    const uint32_t* base = lc + kCacheHeaderWords;
    uint32_t h = hash_state<NW>(s) & mask;
    CacheProbe<NW> p = cache_probe<NW>(base, h, s);
    bool hit = p.tag != 0 && p.tag <= visible && p.same;
    bool walking = p.tag != 0 && !hit;
    if (__builtin_expect(__ballot(walking) != 0, 0)) {
        while (walking) {
            h = (h + 1) & mask;
            const CacheProbe<NW> q = cache_probe<NW>(base, h, s);
            const bool here = q.tag != 0 && q.tag <= visible && q.same;
            if (here) { p = q; hit = true; }
            walking = q.tag != 0 && !here;
        }
    }
    length = p.length;
#pragma unroll
    for (int w = 0; w < NW; ++w) key[w] = p.key[w];
    if (tag_out) *tag_out = p.tag;
    return hit;
}

__device__ __forceinline__ uint32_t cache_visible(const uint32_t* lc) {
    return __hip_atomic_load(lc, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Single-thread insert into the LDS mirror (only thread 0 of a workgroup writes it).
template <int NW>
__device__ __forceinline__ void cache_insert_lds(uint32_t* lc, uint32_t mask, const uint32_t (&s)[NW],
                                                 uint32_t length, const uint32_t (&key)[NW], uint32_t tag) {
    constexpr int S = CacheLayout<NW>::kStride;
    uint32_t* base = lc + kCacheHeaderWords;
    uint32_t h = hash_state<NW>(s) & mask;
    while (base[h * S + NW] != 0) h = (h + 1) & mask;
    uint32_t* e = base + h * S;
#pragma unroll
    for (int w = 0; w < NW; ++w) { e[w] = s[w]; e[NW + 2 + w] = key[w]; }
    e[NW + 1] = length;
    e[NW] = tag;
}

// Thread 0: take the attractors published since `seen` (by any workgroup) from the HBM journal,
// regenerate their cycles from the key and make each cycle visible in the LDS mirror at once.
template <int NW, int K>
__device__ __forceinline__ void cache_pull(const CycleCache& cc, const NetView<NW, K>& nv,
                                           const uint32_t (&fm)[NW], const uint32_t (&fv)[NW], uint32_t* lc,
                                           uint32_t& seen, uint32_t& n_states, uint32_t& n_attr) {
    const uint32_t mask = cc.lds_slots - 1;
    uint32_t jc = __hip_atomic_load(cc.journal_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (jc > kCycleJournalCap) jc = kCycleJournalCap;
    while (seen < jc) {
        const CycleRecord* r = &cc.journal[seen];
        if (__hip_atomic_load(&r->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) break;   // next time
        uint32_t key[NW], s[NW], nxt[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) key[w] = __hip_atomic_load(&r->key[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t len = __hip_atomic_load(&r->length, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++seen;
        uint32_t l2, k2[NW];
        if (cache_lookup<NW>(lc, mask, n_attr, key, l2, k2)) continue;              // duplicate record
        if (len == 0 || len > kCycleCacheMaxLen || n_states + len > cc.lds_slots / 2) continue;   // does not fit
        const uint32_t tag = n_attr + 1;
        copy_words<NW>(s, key);
        for (uint32_t i = 0; i < len; ++i) {
            cache_insert_lds<NW>(lc, mask, s, len, key, tag);
            net_step<NW, K>(nv, s, fm, fv, nxt);
            copy_words<NW>(s, nxt);
        }
        n_states += len;
        n_attr = tag;
        __hip_atomic_store(lc, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Append (key, length) to the journal unless its fingerprint has been claimed already.  Rare path.
template <int NW>
__device__ __forceinline__ void cache_publish(const CycleCache& cc, const uint32_t (&key)[NW], uint32_t length) {
    uint32_t fp = hash_state<NW>(key) * 0x9E3779B1u;
#pragma unroll
    for (int w = 0; w < NW; ++w) fp = (fp ^ key[w]) * 0x85EBCA6Bu;
    fp |= 1u;                                                      // never 0
    uint32_t h = (fp >> 7) & (kCycleClaimSlots - 1);
    for (int probe = 0; probe < 32; ++probe) {
        const unsigned int was = atomicCAS(&cc.claims[h], 0u, fp);
        if (was == fp) return;                                     // published (or being published) already
        if (was == 0u) {
            const unsigned int j = atomicAdd(cc.journal_count, 1u);
            if (j >= kCycleJournalCap) return;
            CycleRecord* r = &cc.journal[j];
#pragma unroll
            for (int w = 0; w < NW; ++w) __hip_atomic_store(&r->key[w], key[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&r->length, length, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&r->ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        h = (h + 1) & (kCycleClaimSlots - 1);
    }
}

// Enumeration fast path: no variations, 'any' nodes = nodes 0..n_any-1 with n_any <= 64.
template <int NW>
__device__ __forceinline__ void init_problem_simple(const DevSpace& sp, uint64_t p, uint32_t (&s)[NW]) {
    const uint64_t d = sp.first_digits[0] + p;
#pragma unroll
    for (int w = 0; w < NW; ++w) s[w] = sp.origin[w];
    s[0] |= (uint32_t)d;
    if constexpr (NW > 1) s[1] |= (uint32_t)(d >> 32);
}

// ------------------------------------------------------------------------------------------------
// attract: attract.py:262-302 semantics (S5-S7, S9, S10) for problems [first, first + count).
//
// Lane life cycle: IDLE -> [WARM: T_p steps under perturbations] -> FAST: step + cycle-cache lookup
// only (taken when cached attractors exist; ends at mu on the first cached cycle state) -> if nothing
// is hit within kFastSteps the search restarts from s(T_p) in BRENT (detector + lookups) -> ADVANCE ->
// MU -> DONE.  With a warm cache almost every lane ends in FAST, which carries no detector state.
//
// FAST_ONLY = true builds the lean kernel used once attractors are cached: only IDLE / FAST / DONE
// exist, the cache mirror is static for the whole launch, the probe of the current state is issued
// together with the gather reads of the next step, and a lane that hits nothing within
// P.fast_steps hands its problem to the general kernel through the straggler list.
// Minimum waves per SIMD requested from the register allocator (a 512-thread workgroup is 2 per SIMD).
constexpr int attract_min_waves(int nw, bool fast) {
    return fast ? (nw == 1 ? 8 : nw == 2 ? 6 : nw == 4 ? 4 : 2) : (nw <= 2 ? 4 : 2);
}

template <int NW, int K, bool LDS_LUT, bool FAST_ONLY>
__global__ __launch_bounds__(kBlock, attract_min_waves(NW, FAST_ONLY)) void k_attract(const AttractParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
    const NetView<NW, K> nv = stage_network<NW, K, LDS_LUT>(P.net, smem, smem_free);
    const int lane = threadIdx.x & 63;
    const bool has_warmup = !FAST_ONLY && (P.sp.tp_origin | P.sp.n_pv) != 0;       // wave-uniform
    const bool simple_space = FAST_ONLY || (P.sp.identity_any && P.sp.n_any <= 64 && !P.sp.n_fv && !P.sp.n_pv);
    const bool use_cache = FAST_ONLY || P.cc.enabled != 0;
    const uint32_t fast_steps = P.fast_steps;
    const uint32_t service_lanes = P.pad ? P.pad : kServiceLanes;
    const uint32_t cmask = P.cc.lds_slots - 1;

    // LDS mirror of the cycle-state cache (pointer arithmetic on `smem` keeps the LDS address space:
    // a round trip through an integer would turn every probe into a flat load)
    uint32_t* lc = smem + (((uint32_t)(smem_free - smem) + 3u) & ~3u);
    uint32_t cc_seen = 0, cc_states = 0, cc_attr = 0, cc_rounds = 0;       // meaningful in thread 0 only
    uint32_t fm0[NW], fv0[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) { fm0[w] = P.sp.fixmask[w]; fv0[w] = P.sp.fixval[w]; }
    if (use_cache) {
        for (uint32_t i = threadIdx.x; i < kCacheHeaderWords + P.cc.lds_slots * CacheLayout<NW>::kStride; i += blockDim.x)
            lc[i] = 0;
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
    uint32_t clen = 0, ccnt = 0;
    uint64_t csl = 0, csl2 = 0;

    uint64_t steps_ref = 0, steps_exec = 0;
    uint32_t n_none = 0, limit_hits = 0;

    WaveQueue q{0, 0, true};
    if constexpr (FAST_ONLY) vis = cache_visible(lc);      // static for the launch: nobody inserts

    // start of the search at s(T_p) = A: snapshot the cache, look s(T_p) itself up, pick the mode
    auto begin_search = [&]() {
        if constexpr (FAST_ONLY) {          // s(T_p) itself is probed by the first iteration
            t = 0; phase = PH_FAST;
            return;
        }
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
        const bool service = (n_pend && (n_pend >= kServiceLanes || n_run == 0 || !work_left)) ||
                             (work_left && (n_wait >= kServiceLanes || n_run == 0));
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
                const uint64_t traj_l = traj32;
                steps_exec += exec32;
                // reference loop stops at T_p + mu + lambda when found, at max_t otherwise (model.py:201)
                steps_ref += found ? (uint64_t)(traj32 + lam) : (P.cap_rel_inf ? 0ull : P.max_t);
                const bool keep = found && (uint64_t)lam <= P.max_len;          // attract.py:294
                want_pub = found && pub && use_cache && lam <= kCycleCacheMaxLen;
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
                const uint64_t sq = (uint64_t)traj32 * traj32;
                if (!keep) ++n_none;
                else if (ccnt && eq_words<NW>(ck, D) && csl2 < (1ull << 62)) { ++ccnt; csl += traj_l; csl2 += sq; }
                else {
                    if (ccnt) { flush = true; copy_words<NW>(fk, ck); flen = clen; fcnt = ccnt; fsl = csl; fsl2 = csl2; }
                    copy_words<NW>(ck, D); clen = lam; ccnt = 1; csl = traj_l; csl2 = sq;
                }
            }
            if (__ballot(flush)) table_merge<NW>(P, slot, lane, flush, fk, flen, fcnt, fsl, fsl2);

            // ---- cycle-state cache upkeep (rare): pull what others published; one lane per newly
            //      detected attractor appends it to the journal
            if (!FAST_ONLY && use_cache) {
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
                    const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
                    if (phase == PH_IDLE && rank < avail) {
                        my_p = q.next + rank;
                        if (!FAST_ONLY && P.offsets) my_p = P.offsets[my_p];
                        exec32 = 0;
                        if (simple_space) {
                            init_problem_simple<NW>(P.sp, my_p, A);
                        } else if constexpr (!FAST_ONLY) {
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
            continue;       // re-evaluate the wave state (nothing to step if every lane is idle)
        }

        if constexpr (FAST_ONLY) {
            // ---- lean iteration: probe the current state s(T_p + t) and compute s(T_p + t + 1) together
            uint32_t nxt[NW], l2 = 0, k2[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) k2[w] = 0;
            const bool hit = cache_lookup<NW>(lc, cmask, vis, A, l2, k2);
            net_step<NW, K>(nv, A, fm0, fv0, nxt);
            if (phase == PH_FAST) {
                ++exec32;
                const bool ok = hit && t <= cap_rel && l2 <= cap_rel - t;       // mu + lambda <= max_t - T_p
                const bool lost = !hit && t >= fast_steps;
                if (lost) {
                    const unsigned long long at = atomicAdd(&P.ctr->n_stragglers, 1ull);
                    if (at < P.stragglers_cap) P.stragglers[at] = (uint32_t)my_p;
                    else atomicOr(&P.ctr->straggler_overflow, 1u);
                    steps_exec += exec32;
                }
#pragma unroll
                for (int w = 0; w < NW; ++w) { D[w] = hit ? k2[w] : D[w]; A[w] = nxt[w]; }
                lam = l2;
                cnt = ok ? t : 0u;
                sub = ok ? 1u : 0u;
                ++t;
                phase = hit ? PH_DONE : (lost ? PH_IDLE : PH_FAST);
            }
            continue;
        }

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
    if (slot.count) log_append<NW>(P, slot.key, slot.length, slot.count, slot.sum_l, slot.sum_l2);
    atomicAdd(&P.ctr->steps_ref, (unsigned long long)steps_ref);
    atomicAdd(&P.ctr->steps_exec, (unsigned long long)steps_exec);
    if (n_none) atomicAdd(&P.ctr->n_none, (unsigned long long)n_none);
    if (limit_hits) atomicAdd(&P.ctr->step_limit_hits, limit_hits);
}

template <int NW>
__device__ __forceinline__ bool target_hit(const uint32_t (&s)[NW], const uint32_t (&tm)[NW], const uint32_t (&tc)[NW]) {
    uint32_t d = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) d |= (s[w] & tm[w]) ^ tc[w];
    return d == 0;
}

// ------------------------------------------------------------------------------------------------
// target: stop at the first t >= T_p with (state & mask) == code; a trajectory that closes its cycle
// (or hits max_t) first reaches nothing (target.py:109-133 over model.py:152-236, S12).
// Output: t_hit[p] for every problem (kNotReached if none); k_compact_* turn it into the hit list.
constexpr uint32_t kNotReached = 0xFFFFFFFFu;

template <int NW, int K, bool LDS_LUT>
__global__ __launch_bounds__(kBlock, NW <= 2 ? 4 : 2) void k_target(const TargetParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
    const NetView<NW, K> nv = stage_network<NW, K, LDS_LUT>(P.net, smem, smem_free);
    const int lane = threadIdx.x & 63;
    const bool simple_space = P.sp.identity_any && P.sp.n_any <= 64 && !P.sp.n_fv && !P.sp.n_pv;

    uint32_t A[NW], B[NW], fm[NW], fv[NW], tm[NW], tc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) { A[w] = B[w] = 0; fm[w] = P.sp.fixmask[w]; fv[w] = P.sp.fixval[w]; tm[w] = P.tmask[w]; tc[w] = P.tcode[w]; }
    // t: absolute time; lam/power: Brent's counters (only to notice that the cycle closed)
    uint32_t phase = PH_IDLE, t = 0, tp = P.sp.tp_origin, lam = 0, power = 1;
    const uint32_t t_cap = (P.cap_rel_inf || P.max_t >= kStepLimit) ? kStepLimit : (uint32_t)P.max_t;
    uint64_t pv_digits = 0, my_p = 0;
    uint64_t steps_exec = 0;
    uint32_t limit_hits = 0, n_hits = 0;
    WaveQueue q{0, 0, true};

    for (;;) {
        const uint32_t n_run = __popcll(__ballot(phase >= PH_WARM));
        const bool work_left = q.more || q.next < q.end;
        if (n_run == 0 && !work_left) break;
        if (work_left && (64u - n_run >= kServiceLanes || n_run == 0)) {
            // ---- refill idle lanes with the next problems of the wave's chunk
            if (q.next == q.end) {
                const uint64_t base = grab_chunk(&P.ctr->cursor, P.chunk, lane);
                if (base >= P.count) q.more = false;
                else { q.next = base; q.end = (base + P.chunk < P.count) ? base + P.chunk : P.count; }
            }
            const uint64_t avail = q.end - q.next;
            const uint64_t idle = __ballot(phase == PH_IDLE);
            if (avail && idle) {
                const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
                if (phase == PH_IDLE && rank < avail) {
                    my_p = q.next + rank;
                    if (simple_space) {
                        init_problem_simple<NW>(P.sp, my_p, A);
                    } else {
                        Problem<NW> pr;
                        init_problem<NW>(P.sp, my_p, pr);
                        copy_words<NW>(A, pr.s); copy_words<NW>(fm, pr.fm); copy_words<NW>(fv, pr.fv);
                        pv_digits = pr.pv_digits; tp = pr.tp;
                    }
                    t = 0; lam = 0; power = 1;
                    copy_words<NW>(B, A);
                    phase = tp > 0 ? PH_WARM : PH_BRENT;
                }
                const uint64_t n_idle = (uint64_t)__popcll(idle);
                q.next += n_idle < avail ? n_idle : avail;
            }
            continue;
        }

        // ---- check the current state (covers s(T_p), model.py:200), then one network update
        uint32_t nxt[NW];
        net_step<NW, K>(nv, A, fm, fv, nxt);
        if (phase == PH_BRENT) {
            const bool reached = target_hit<NW>(A, tm, tc);
            const bool capped = !reached && t >= t_cap;
            const uint32_t lam1 = lam + 1;
            const bool closed = !reached && !capped && eq_words<NW>(nxt, B);     // every state has been checked
            const bool tele = !closed && lam1 == power;
            if (reached | capped | closed) {
                P.t_hit[my_p] = reached ? t : kNotReached;
                n_hits += reached ? 1u : 0u;
                limit_hits += (capped && t_cap == kStepLimit) ? 1u : 0u;
                steps_exec += t;
                phase = PH_IDLE;
            } else {
#pragma unroll
                for (int w = 0; w < NW; ++w) { A[w] = nxt[w]; B[w] = tele ? nxt[w] : B[w]; }
                power = tele ? power << 1 : power;
                lam = tele ? 0u : lam1;
                ++t;
            }
        } else if (phase == PH_WARM) {
            ++t;
            apply_perturbations<NW>(P.sp, t, pv_digits, nxt);
            copy_words<NW>(A, nxt);
            if (t == tp) { phase = PH_BRENT; lam = 0; power = 1; copy_words<NW>(B, A); }
        }
    }
    atomicAdd(&P.ctr->steps_ref, (unsigned long long)steps_exec);
    atomicAdd(&P.ctr->steps_exec, (unsigned long long)steps_exec);
    if (n_hits) atomicAdd(&P.ctr->log_cursor, (unsigned long long)n_hits);
    if (limit_hits) atomicAdd(&P.ctr->step_limit_hits, limit_hits);
}

// Ordered stream compaction of t_hit[] into the hit list: pass 1 counts hits per segment, the host
// scans the (small) count array, pass 2 writes each segment's hits at its base in index order.
constexpr uint32_t kCompactSegment = 4096;

__global__ __launch_bounds__(256) void k_compact_count(const uint32_t* t_hit, uint64_t count, uint32_t* seg_counts) {
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kCompactSegment;
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < kCompactSegment; i += 256)
        if (base + i < count && t_hit[base + i] != kNotReached) ++mine;
    atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) seg_counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_compact_write(const uint32_t* t_hit, uint64_t count, const uint64_t* seg_base,
                                                       HitRec* hits, uint64_t hits_cap) {
    __shared__ uint32_t wave_tot[4];
    const uint64_t base = (uint64_t)blockIdx.x * kCompactSegment;
    uint64_t out = seg_base[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t c0 = 0; c0 < kCompactSegment; c0 += 256) {
        const uint64_t p = base + c0 + threadIdx.x;
        const uint32_t t = p < count ? t_hit[p] : kNotReached;
        const bool hit = t != kNotReached;
        const uint64_t m = __ballot(hit);
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        uint32_t before = __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        const uint32_t all = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        if (hit && out + before < hits_cap) { hits[out + before].offset = p; hits[out + before].t = t; }
        out += all;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// simulate: s(0..T) by plain stepping (simulate.py:97-131 == S11); sinks: trajectory, final state, digest.
template <int NW, int K, bool LDS_LUT>
__global__ __launch_bounds__(kBlock) void k_simulate(const SimParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
    const NetView<NW, K> nv = stage_network<NW, K, LDS_LUT>(P.net, smem, smem_free);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t steps = 0;
    for (uint64_t qi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; qi < P.count; qi += stride) {
        const uint64_t p = P.offsets ? P.offsets[qi] : qi;
        const uint64_t T = P.t_len ? P.t_len[qi] : P.max_t;
        Problem<NW> pr;
        init_problem<NW>(P.sp, p, pr);
        uint32_t s[NW];
        copy_words<NW>(s, pr.s);
        uint64_t* out = P.traj ? P.traj + (P.out_offsets ? P.out_offsets[qi] : qi * (P.max_t + 1) * P.w64) : nullptr;
        uint64_t dg = kDigestSeed;
        for (uint64_t t = 0;; ++t) {
#pragma unroll
            for (int w = 0; w < (NW + 1) / 2; ++w) {
                uint64_t word = s[2 * w];
                if (2 * w + 1 < NW) word |= (uint64_t)s[2 * w + 1] << 32;
                if ((uint32_t)w < P.w64) {
                    if (out) out[t * P.w64 + w] = word;
                    dg = (dg ^ word) * kDigestPrime;
                }
            }
            if (t == T) break;
            uint32_t nxt[NW];
            net_step<NW, K>(nv, s, pr.fm, pr.fv, nxt);
            if (t + 1 <= pr.tp) apply_perturbations<NW>(P.sp, (uint32_t)(t + 1), pr.pv_digits, nxt);
            copy_words<NW>(s, nxt);
            ++steps;
        }
        if (P.final_states) {
#pragma unroll
            for (int w = 0; w < (NW + 1) / 2; ++w) {
                uint64_t word = s[2 * w];
                if (2 * w + 1 < NW) word |= (uint64_t)s[2 * w + 1] << 32;
                if ((uint32_t)w < P.w64) P.final_states[qi * P.w64 + w] = word;
            }
        }
        if (P.digests) P.digests[qi] = dg;
    }
    atomicAdd(&P.ctr->steps_ref, (unsigned long long)steps);
    atomicAdd(&P.ctr->steps_exec, (unsigned long long)steps);
}

// ------------------------------------------------------------------------------------------------
// Bit-sliced simulate: the "state bit-matrix" formulation for fixed-length runs (simulate.py:97-131
// with T_p = max_t, e.g. BASELINE config 5).  All trajectories advance in lock step, so nothing
// diverges: lane l of a wave owns trajectories 32l..32l+31 of its group, row i of the matrix holds
// node i of those 32 trajectories in one 32-bit word, and one step evaluates every node's truth table
// for 2048 trajectories at once: K conflict-free LDS reads (row p_j, this lane), a mux tree of v_bfi
// whose leaves are the (wave-uniform) truth-table bits, one LDS write.  ~0.4 VALU + 0.04 LDS
// instructions per node update of one trajectory, against ~2 + 0.4 for the one-trajectory-per-lane
// kernel at n = 128, K = 3.
constexpr int kSlicedBatch = 4;      // nodes evaluated between LDS write-backs (independent reads overlap)
constexpr int kSlicedWaves = 4;      // waves of a workgroup share the 2048 trajectories and split the nodes

template <int NW, int K>
__global__ __launch_bounds__(64 * kSlicedWaves) void k_simulate_sliced(const SlicedParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = P.n_nodes, rows = P.n_rows;      // rows is a multiple of kSlicedBatch * kSlicedWaves
    const uint32_t rows_per_wave = rows / kSlicedWaves;
    uint32_t* desc = smem;                      // [rows][8]
    uint32_t* buf0 = desc + rows * 8;           // [rows][64]
    uint32_t* buf1 = buf0 + rows * 64;
    for (uint32_t i = threadIdx.x; i < rows * 8; i += blockDim.x) desc[i] = P.desc[i];

    const uint64_t n_groups = (P.count + 2047) / 2048;
    for (uint64_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
        const uint64_t base = group * 2048 + (uint64_t)lane * 32;
        // ---- initial states: one 32-node word at a time, packed words of the lane's 32 trajectories
        //      into buf1 as scratch (32 x 64 words), then transposed into 32 rows of buf0; the waves
        //      split the trajectories (k) and then the rows (b)
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            __syncthreads();
            for (uint32_t k = wave * 8; k < wave * 8 + 8; ++k) {
                Problem<NW> pr;
                if (base + k < P.count) init_problem<NW>(P.sp, base + k, pr);
                else pr.s[w] = 0;
                buf1[k * 64 + lane] = pr.s[w];
            }
            __syncthreads();
            for (uint32_t b = wave * 8; b < wave * 8 + 8; ++b) {
                const uint32_t node = w * 32 + b;
                if (node >= rows) break;
                uint32_t row = 0;
                if (node < n) {
                    for (uint32_t k = 0; k < 32; ++k) row |= ((buf1[k * 64 + lane] >> b) & 1u) << k;
                }
                buf0[node * 64 + lane] = row;
            }
        }
        __syncthreads();

        // ---- T synchronous updates; wave v evaluates rows [v * rows / 4, (v + 1) * rows / 4)
        uint32_t* cur = buf0;
        uint32_t* nxt = buf1;
        uint32_t sched_at = 0;
        for (uint64_t t = 1; t <= P.max_t; ++t) {
            for (uint32_t i0 = wave * rows_per_wave; i0 < (wave + 1) * rows_per_wave; i0 += kSlicedBatch) {
                uint32_t out[kSlicedBatch];
#pragma unroll
                for (int u = 0; u < kSlicedBatch; ++u) {
                    const uint32_t* d = desc + (i0 + u) * 8;
                    const uint4 d0 = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(d, 16));
                    const uint4 d1 = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(d + 4, 16));
                    const uint32_t pred[6] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y};
                    const uint32_t tt[2] = {d1.z, d1.w};
                    uint32_t g[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) g[j] = cur[pred[j] * 64 + lane];
                    // leaves: truth-table bits as all-ones / all-zeros words, selected by predecessor 0
                    uint32_t r[1 << (K - 1)];
#pragma unroll
                    for (int idx = 0; idx < (1 << (K - 1)); ++idx) {
                        const uint32_t hi = 0u - ((tt[(2 * idx + 1) >> 5] >> ((2 * idx + 1) & 31)) & 1u);
                        const uint32_t lo = 0u - ((tt[(2 * idx) >> 5] >> ((2 * idx) & 31)) & 1u);
                        r[idx] = (g[0] & hi) | (~g[0] & lo);
                    }
#pragma unroll
                    for (int j = 1; j < K; ++j)
#pragma unroll
                        for (int idx = 0; idx < (1 << (K - 1 - j)); ++idx)
                            r[idx] = bfi(g[j], r[2 * idx + 1], r[2 * idx]);
                    out[u] = r[0];
                }
#pragma unroll
                for (int u = 0; u < kSlicedBatch; ++u) nxt[(i0 + u) * 64 + lane] = out[u];
            }
            __syncthreads();
            // perturbation override at time t (model.py:68-71): whole rows, the schedule is the same
            // for every trajectory (spaces with variations use the per-lane kernel)
            while (sched_at < P.n_sched && P.sched[3 * sched_at] < t) ++sched_at;
            const uint32_t sched_first = sched_at;
            while (sched_at < P.n_sched && P.sched[3 * sched_at] == t) {
                if (wave == 0) nxt[P.sched[3 * sched_at + 1] * 64 + lane] = P.sched[3 * sched_at + 2] ? 0xFFFFFFFFu : 0u;
                ++sched_at;
            }
            if (sched_at != sched_first) __syncthreads();      // uniform: every wave walks the same schedule
            uint32_t* swap = cur; cur = nxt; nxt = swap;
        }

        // ---- final states back to one word sequence per trajectory (waves split the trajectories)
        for (uint32_t k = wave * 8; k < wave * 8 + 8; ++k) {
            if (base + k >= P.count) break;
            uint32_t s[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                uint32_t word = 0;
                for (uint32_t b = 0; b < 32; ++b) {
                    const uint32_t node = w * 32 + b;
                    if (node < n) word |= ((cur[node * 64 + lane] >> k) & 1u) << b;
                }
                s[w] = word;
            }
#pragma unroll
            for (int w = 0; w < (NW + 1) / 2; ++w) {
                uint64_t word = s[2 * w];
                if (2 * w + 1 < NW) word |= (uint64_t)s[2 * w + 1] << 32;
                if ((uint32_t)w < P.w64) P.final_states[(base + k) * P.w64 + w] = word;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        atomicAdd(&P.ctr->steps_ref, (unsigned long long)(P.count * P.max_t));
        atomicAdd(&P.ctr->steps_exec, (unsigned long long)(P.count * P.max_t));
    }
}

template <int NW, int K>
static hipError_t launch_sliced_nk(bool, dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P) {
    hipError_t e = hipFuncSetAttribute((const void*)k_simulate_sliced<NW, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_simulate_sliced<NW, K>), grid, dim3(64 * kSlicedWaves), shmem, st, P);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Launch dispatch over (NW, K, LDS_LUT).  NW in {1,2,4,8}; K in 1..6.
template <int NW, int K>
static hipError_t launch_attract_nk(bool lds, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    if (lds) hipLaunchKernelGGL((k_attract<NW, K, true, false>), grid, dim3(kBlock), shmem, st, P);
    else hipLaunchKernelGGL((k_attract<NW, K, false, false>), grid, dim3(kBlock), shmem, st, P);
    return hipGetLastError();
}
template <int NW, int K>
static hipError_t launch_attract_fast_nk(bool lds, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    if (lds) hipLaunchKernelGGL((k_attract<NW, K, true, true>), grid, dim3(kBlock), shmem, st, P);
    else hipLaunchKernelGGL((k_attract<NW, K, false, true>), grid, dim3(kBlock), shmem, st, P);
    return hipGetLastError();
}
template <int NW, int K>
static hipError_t launch_target_nk(bool lds, dim3 grid, size_t shmem, hipStream_t st, const TargetParams& P) {
    if (lds) hipLaunchKernelGGL((k_target<NW, K, true>), grid, dim3(kBlock), shmem, st, P);
    else hipLaunchKernelGGL((k_target<NW, K, false>), grid, dim3(kBlock), shmem, st, P);
    return hipGetLastError();
}
template <int NW, int K>
static hipError_t launch_simulate_nk(bool lds, dim3 grid, size_t shmem, hipStream_t st, const SimParams& P) {
    if (lds) hipLaunchKernelGGL((k_simulate<NW, K, true>), grid, dim3(kBlock), shmem, st, P);
    else hipLaunchKernelGGL((k_simulate<NW, K, false>), grid, dim3(kBlock), shmem, st, P);
    return hipGetLastError();
}

#define BSX_DISPATCH_K(FN, NWV)                                                         \
    switch (k) {                                                                        \
        case 1: return FN<NWV, 1>(lds, grid, shmem, st, P);                             \
        case 2: return FN<NWV, 2>(lds, grid, shmem, st, P);                             \
        case 3: return FN<NWV, 3>(lds, grid, shmem, st, P);                             \
        case 4: return FN<NWV, 4>(lds, grid, shmem, st, P);                             \
        case 5: return FN<NWV, 5>(lds, grid, shmem, st, P);                             \
        case 6: return FN<NWV, 6>(lds, grid, shmem, st, P);                             \
        default: return hipErrorInvalidValue;                                           \
    }

#define BSX_DISPATCH(FN)                                                                \
    switch (nw) {                                                                       \
        case 1: BSX_DISPATCH_K(FN, 1)                                                   \
        case 2: BSX_DISPATCH_K(FN, 2)                                                   \
        case 4: BSX_DISPATCH_K(FN, 4)                                                   \
        case 8: BSX_DISPATCH_K(FN, 8)                                                   \
        default: return hipErrorInvalidValue;                                           \
    }

hipError_t launch_attract(int nw, int k, bool lds, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    BSX_DISPATCH(launch_attract_nk)
}
hipError_t launch_attract_fast(int nw, int k, bool lds, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    BSX_DISPATCH(launch_attract_fast_nk)
}
hipError_t launch_target(int nw, int k, bool lds, dim3 grid, size_t shmem, hipStream_t st, const TargetParams& P) {
    BSX_DISPATCH(launch_target_nk)
}
hipError_t launch_compact(const uint32_t* t_hit, uint64_t count, uint32_t* seg_counts, const uint64_t* seg_base,
                          HitRec* hits, uint64_t hits_cap, bool write_pass, hipStream_t st) {
    const uint32_t blocks = (uint32_t)((count + kCompactSegment - 1) / kCompactSegment);
    if (!write_pass) hipLaunchKernelGGL(k_compact_count, dim3(blocks), dim3(256), 0, st, t_hit, count, seg_counts);
    else hipLaunchKernelGGL(k_compact_write, dim3(blocks), dim3(256), 0, st, t_hit, count, seg_base, hits, hits_cap);
    return hipGetLastError();
}
hipError_t launch_simulate(int nw, int k, bool lds, dim3 grid, size_t shmem, hipStream_t st, const SimParams& P) {
    BSX_DISPATCH(launch_simulate_nk)
}
hipError_t launch_simulate_sliced(int nw, int k, dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P) {
    const bool lds = true;
    BSX_DISPATCH(launch_sliced_nk)
}

template <int NW, int K>
static hipError_t configure_nk(bool lds, dim3, size_t shmem, hipStream_t, const int&) {
    const int bytes = (int)shmem;
    hipError_t e;
    if (lds) {
        e = hipFuncSetAttribute((const void*)k_attract<NW, K, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_attract<NW, K, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_target<NW, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_simulate<NW, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    } else {
        e = hipFuncSetAttribute((const void*)k_attract<NW, K, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_attract<NW, K, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_target<NW, K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_simulate<NW, K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
    return e;
}

// Allow the instantiation used by a network to take `shmem` bytes of dynamic LDS (above 64 KiB this
// must be requested explicitly).
hipError_t configure_kernels(int nw, int k, bool lds, size_t shmem) {
    const dim3 grid(1);
    const hipStream_t st = nullptr;
    const int P = 0;
    BSX_DISPATCH(configure_nk)
}

}  // namespace bsx
