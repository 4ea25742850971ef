// Functional-graph mode of attract (no reference analogue -- the reference re-simulates
// every trajectory, mpi.py:521-538).  For a full sweep over n <= 32 nodes that are all 'any', the network is a
// function f on N = 2^n states, and every problem's (attractor key, lambda, mu) follows from N-sized arrays
// in HBM instead of N trajectories:
//   A  succ[s] = f(s)                                    one network update per STATE, coalesced 4-byte store
//   B  land[s] = f^(2^k)(s) by pointer doubling          k = ceil(log2(cap + 1)) gather rounds, ping-pong arrays
//   C  distinct landing points -> bitmap -> candidates; a candidate that returns to itself within the cap is
//      a cycle state: (key = min state of its cycle, lambda) goes into a small hash table `cyc`
//   D  pair[s] = (p, d) with f^d(s) = p, halting at cycle states: init (s, 0) / (f(s), 1), then in-place
//      pointer jumping  (p, d) <- (pair[p].p, d + pair[p].d)  until nothing changes: d = mu, p = entry state
//   E  aggregate (count, sum l, sum l^2) per attractor over the problems [first, first + count)
// Origin perturbations (a warm-up of T_p steps, model.py:76-128) only move the start of the search: one more
// array warm[s] = s(T_p), and problem s reads the pair of warm[s]; trajectory_l = T_p + mu.
// Every pass streams whole arrays, so this is the one mode of the engine whose bound really is HBM.  The
// 64-bit pair is read and written whole, so the invariant f^d(s) = p survives any interleaving of phase D.
#include "bsx_kernels_common.h"

namespace bsx {

struct CycEntry { uint32_t state_p1; uint32_t key; uint32_t length; uint32_t pad; };    // state + 1 (0 = empty); n = 32: see kFullState

// n = 32: state 0xFFFFFFFF + 1 wraps to 0 = "empty"; that one state is stored under this code instead
// (pad = 1 marks it), so no state is lost.
__device__ __forceinline__ uint32_t cyc_code(uint32_t s) { return s + 1u; }

struct FgParams {
    DevNet net;
    DevSpace sp;                // origin fixed nodes and perturbation schedule (no variations in this mode)
    uint64_t n_states;
    uint32_t* succ;
    uint32_t warm_steps;        // 0: succ[s] = f(s); T_p > 0: succ[s] = s(T_p) from s(0) = s under the origin perturbations
};

template <int K, int LM>
__global__ __launch_bounds__(kBlock, 4) void k_fg_succ(const FgParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
    const NetView<1, K, LM> nv = stage_network<1, K, LM>(P.net, smem, smem_free);
    const uint32_t fm[1] = {P.sp.fixmask[0]}, fv[1] = {P.sp.fixval[0]};
    const bool has_fixed = fm[0] != 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < P.n_states; s += stride) {
        uint32_t cur[1] = {(uint32_t)s};
        uint32_t nxt[1];
        net_step<1, K>(nv, cur, fm, fv, nxt, has_fixed);
        for (uint32_t t = 1; t <= P.warm_steps; ++t) {          // warm-up map (model.py:76-128), uniform trip count
            apply_perturbations<1>(P.sp, t, 0ull, nxt);
            if (t == P.warm_steps) break;
            cur[0] = nxt[0];
            net_step<1, K>(nv, cur, fm, fv, nxt, has_fixed);
        }
        P.succ[s] = nxt[0];
    }
}

__global__ __launch_bounds__(256) void k_fg_double(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) out[s] = in[in[s]];
}

// bitmap of the distinct landing points (a set bit is seen by a plain load first: after the first few
// thousand states nearly every thread finds its bit set and issues no atomic)
__global__ __launch_bounds__(256) void k_fg_mark(const uint32_t* __restrict__ land, uint64_t n, uint32_t* bits) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) {
        const uint32_t p = land[s], m = 1u << (p & 31);
        if (!(__hip_atomic_load(&bits[p >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & m)) atomicOr(&bits[p >> 5], m);
    }
}

__global__ __launch_bounds__(256) void k_fg_collect(uint32_t* bits, uint64_t n_words, uint32_t* cand, uint32_t cand_cap,
                                                    unsigned int* cursor) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        uint32_t v = bits[w];
        if (!v) continue;
        bits[w] = 0;                                    // leave the bitmap clean for the next use
        for (; v; v &= v - 1) {
            const unsigned int at = atomicAdd(cursor, 1u);
            if (at < cand_cap) cand[at] = (uint32_t)(w * 32u) + (uint32_t)__builtin_ctz(v);
        }
    }
}

// One thread per candidate: does it return to itself within `walk_cap` steps?  Then it is a cycle state.
__global__ __launch_bounds__(256) void k_fg_cycles(const uint32_t* __restrict__ succ, const uint32_t* cand, uint32_t n_cand,
                                                   uint64_t walk_cap, CycEntry* cyc, uint32_t cyc_mask, unsigned int* n_cyclic,
                                                   unsigned int* n_open) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand) return;
    const uint32_t c = cand[i];
    uint32_t x = succ[c], key = c;
    uint64_t len = 1;
    while (x != c && len < walk_cap) { key = x < key ? x : key; x = succ[x]; ++len; }
    if (x != c) { atomicAdd(n_open, 1u); return; }      // not cyclic, or a cycle longer than the cap
    atomicAdd(n_cyclic, 1u);
    const uint32_t code = cyc_code(c);
    uint32_t h = (c * 0x9E3779B1u) >> 7;
    for (;;) {
        h &= cyc_mask;
        if (code == 0u) {                               // state 0xFFFFFFFF: the dedicated last slot
            cyc[cyc_mask + 1].state_p1 = 1u; cyc[cyc_mask + 1].key = key; cyc[cyc_mask + 1].length = (uint32_t)len; cyc[cyc_mask + 1].pad = 1u;
            return;
        }
        const uint32_t was = atomicCAS(&cyc[h].state_p1, 0u, code);
        if (was == 0u || was == code) { cyc[h].key = key; cyc[h].length = (uint32_t)len; return; }
        ++h;
    }
}

__device__ __forceinline__ bool cyc_find(const CycEntry* cyc, uint32_t cyc_mask, uint32_t s, uint32_t& key, uint32_t& length) {
    const uint32_t code = cyc_code(s);
    if (code == 0u) {
        const CycEntry e = cyc[cyc_mask + 1];
        key = e.key; length = e.length;
        return e.pad == 1u;
    }
    uint32_t h = (s * 0x9E3779B1u) >> 7;
    for (;;) {
        h &= cyc_mask;
        const CycEntry e = cyc[h];
        if (e.state_p1 == code) { key = e.key; length = e.length; return true; }
        if (e.state_p1 == 0u) return false;
        ++h;
    }
}

__global__ __launch_bounds__(256) void k_fg_pair_init(const uint32_t* __restrict__ succ, const CycEntry* cyc, uint32_t cyc_mask,
                                                      unsigned long long* pair, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) {
        uint32_t k, l;
        const bool on = cyc_find(cyc, cyc_mask, (uint32_t)s, k, l);
        pair[s] = on ? (unsigned long long)s : ((unsigned long long)succ[s] | (1ull << 32));      // low word p, high word d
    }
}

__global__ __launch_bounds__(256) void k_fg_pair_jump(unsigned long long* pair, uint64_t n, uint32_t d_cap, unsigned int* changed) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool any = false;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) {
        const unsigned long long a = pair[s];
        const uint32_t d = (uint32_t)(a >> 32);
        if (d == 0u || d > d_cap) continue;             // a cycle state, or already beyond every cap
        const unsigned long long b = __hip_atomic_load(&pair[(uint32_t)a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t d2 = (uint32_t)(b >> 32);
        if (d2 == 0u) continue;                         // p is a cycle state: d = mu
        const unsigned long long dn = (unsigned long long)d + d2;
        __hip_atomic_store(&pair[s], (unsigned long long)(uint32_t)b | ((dn > 0xFFFFFFFFull ? 0xFFFFFFFFull : dn) << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        any = true;
    }
    if (__ballot(any) && (threadIdx.x & 63) == 0) atomicOr(changed, 1u);
}

// Phase E.  Results of a wave are grouped by key with ballots (neighbouring states mostly share their
// attractor), one lane per group adds them to the workgroup's LDS table, which is flushed into the HBM
// attractor table at the end.
constexpr uint32_t kFgSlots = 256;

__global__ __launch_bounds__(256) void k_fg_aggregate(const unsigned long long* __restrict__ pair, const CycEntry* cyc, uint32_t cyc_mask,
                                                      const uint32_t* __restrict__ warm, uint32_t tp,
                                                      uint64_t first, uint64_t count, uint64_t cap_rel, uint64_t max_len,
                                                      uint64_t max_t, const AttractParams P) {
    __shared__ uint32_t s_key[kFgSlots], s_len[kFgSlots];          // key + 1 (0 = empty)
    __shared__ unsigned long long s_cnt[kFgSlots], s_sl[kFgSlots], s_sl2[kFgSlots], s_sl2h[kFgSlots];
    for (uint32_t i = threadIdx.x; i < kFgSlots; i += blockDim.x) { s_key[i] = 0; s_len[i] = 0; s_cnt[i] = 0; s_sl[i] = 0; s_sl2[i] = 0; s_sl2h[i] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long n_none = 0, steps_ref = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (count + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        bool valid = i < count;
        uint32_t key = 0, lam = 0, mu = 0;
        if (valid) {
            const unsigned long long a = pair[warm ? (uint64_t)warm[first + i] : first + i];      // search starts at s(T_p)
            mu = (uint32_t)(a >> 32);
            const bool on = cyc_find(cyc, cyc_mask, (uint32_t)a, key, lam);
            const bool found = on && (uint64_t)mu + lam <= cap_rel;         // S7: mu + lambda <= max_t - T_p
            steps_ref += found ? (unsigned long long)tp + mu + lam : max_t; // model.py:201
            mu += tp;                                                       // trajectory_l = T_p + mu (attract.py:291-298)
            if (!found || (uint64_t)lam > max_len) { ++n_none; valid = false; }
        }
        uint64_t todo = __ballot(valid);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            const uint32_t k = __builtin_amdgcn_readlane(key, src);
            const bool same = valid && key == k;
            todo &= ~__ballot(same);
            const unsigned long long m = __popcll(__ballot(same));
            const unsigned long long sl = wave_sum(same ? (unsigned long long)mu : 0ull);
            const unsigned long long sq = same ? (unsigned long long)mu * mu : 0ull;   // < 2^64
            const unsigned long long sq_lo = wave_sum(sq & 0xFFFFFFFFull), sq_hi = wave_sum(sq >> 32);     // no overflow: 64 terms
            if (lane == src) {
                uint32_t h = (k * 0x9E3779B1u) >> 9;
                bool placed = false;
                for (uint32_t probe = 0; probe < kFgSlots && !placed; ++probe, ++h) {
                    h &= kFgSlots - 1;
                    const uint32_t was = atomicCAS(&s_key[h], 0u, k + 1u);
                    if (was == 0u || was == k + 1u) {
                        if (k + 1u == 0u) break;                    // key 0xFFFFFFFF cannot be coded here: straight to HBM
                        s_len[h] = lam;
                        atomicAdd(&s_cnt[h], m);
                        atomicAdd(&s_sl[h], sl);
                        const unsigned long long add_lo = sq_lo + (sq_hi << 32), add_hi = (sq_hi >> 32) + ((sq_lo + (sq_hi << 32) < sq_lo) ? 1ull : 0ull);
                        const unsigned long long old = atomicAdd(&s_sl2[h], add_lo);
                        const unsigned long long up = add_hi + ((old + add_lo < old) ? 1ull : 0ull);
                        if (up) atomicAdd(&s_sl2h[h], up);
                        placed = true;
                    }
                }
                if (!placed) {                                  // LDS table full (or the uncodable key): HBM table directly
                    const uint32_t kk[1] = {k};
                    const unsigned long long add_lo = sq_lo + (sq_hi << 32), add_hi = (sq_hi >> 32) + ((sq_lo + (sq_hi << 32) < sq_lo) ? 1ull : 0ull);
                    atomicAdd(&P.ctr->table_inserts, 1ull);
                    table_insert<1>(P, kk, lam, m, sl, add_lo, add_hi);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kFgSlots; i += blockDim.x) {
        if (!s_key[i]) continue;
        const uint32_t kk[1] = {s_key[i] - 1u};
        atomicAdd(&P.ctr->table_inserts, 1ull);
        table_insert<1>(P, kk, s_len[i], s_cnt[i], s_sl[i], s_sl2[i], s_sl2h[i]);
    }
    wave_atomic_add(&P.ctr->n_none, n_none, lane);
    wave_atomic_add(&P.ctr->steps_ref, steps_ref, lane);
}

// ---- launchers -------------------------------------------------------------------------------------------
template <int K>
static const void* fg_succ_kernel(int lut_mode) {
    if (lut_mode == kLutLdsByte) return (const void*)k_fg_succ<K, kLutLdsByte>;
    if (lut_mode == kLutGlobal) return (const void*)k_fg_succ<K, kLutGlobal>;
    return nullptr;
}

hipError_t launch_fg_succ(int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const DevNet& net, const DevSpace& sp,
                          uint64_t n_states, uint32_t* succ, uint32_t warm_steps) {
    FgParams P{net, sp, n_states, succ, warm_steps};
    const void* fn = nullptr;
    switch (k) {
        case 1: fn = fg_succ_kernel<1>(lut_mode); break;
        case 2: fn = fg_succ_kernel<2>(lut_mode); break;
        case 3: fn = fg_succ_kernel<3>(lut_mode); break;
        case 4: fn = fg_succ_kernel<4>(lut_mode); break;
        case 5: fn = fg_succ_kernel<5>(lut_mode); break;
        case 6: fn = fg_succ_kernel<6>(lut_mode); break;
        default: break;
    }
    if (!fn) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    void* args[] = {&P};
    return hipLaunchKernel(fn, grid, dim3(kBlock), args, shmem, st);
}

static dim3 fg_grid(uint64_t n, uint32_t cus) { return dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, (uint64_t)cus * 16))); }

hipError_t launch_fg_double(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t cus, hipStream_t st) {
    hipLaunchKernelGGL(k_fg_double, fg_grid(n, cus), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}
hipError_t launch_fg_mark(const uint32_t* land, uint64_t n, uint32_t* bits, uint32_t cus, hipStream_t st) {
    hipLaunchKernelGGL(k_fg_mark, fg_grid(n, cus), dim3(256), 0, st, land, n, bits);
    return hipGetLastError();
}
hipError_t launch_fg_collect(uint32_t* bits, uint64_t n_words, uint32_t* cand, uint32_t cand_cap, unsigned int* cursor, uint32_t cus, hipStream_t st) {
    hipLaunchKernelGGL(k_fg_collect, fg_grid(n_words, cus), dim3(256), 0, st, bits, n_words, cand, cand_cap, cursor);
    return hipGetLastError();
}
hipError_t launch_fg_cycles(const uint32_t* succ, const uint32_t* cand, uint32_t n_cand, uint64_t walk_cap, void* cyc, uint32_t cyc_mask,
                            unsigned int* n_cyclic, unsigned int* n_open, hipStream_t st) {
    hipLaunchKernelGGL(k_fg_cycles, dim3((n_cand + 255) / 256), dim3(256), 0, st, succ, cand, n_cand, walk_cap,
                       static_cast<CycEntry*>(cyc), cyc_mask, n_cyclic, n_open);
    return hipGetLastError();
}
hipError_t launch_fg_pair_init(const uint32_t* succ, const void* cyc, uint32_t cyc_mask, unsigned long long* pair, uint64_t n, uint32_t cus, hipStream_t st) {
    hipLaunchKernelGGL(k_fg_pair_init, fg_grid(n, cus), dim3(256), 0, st, succ, static_cast<const CycEntry*>(cyc), cyc_mask, pair, n);
    return hipGetLastError();
}
hipError_t launch_fg_pair_jump(unsigned long long* pair, uint64_t n, uint32_t d_cap, unsigned int* changed, uint32_t cus, hipStream_t st) {
    hipLaunchKernelGGL(k_fg_pair_jump, fg_grid(n, cus), dim3(256), 0, st, pair, n, d_cap, changed);
    return hipGetLastError();
}
hipError_t launch_fg_aggregate(const unsigned long long* pair, const void* cyc, uint32_t cyc_mask, const uint32_t* warm, uint32_t tp,
                               uint64_t first, uint64_t count,
                               uint64_t cap_rel, uint64_t max_len, uint64_t max_t, const AttractParams& P, uint32_t cus, hipStream_t st) {
    hipLaunchKernelGGL(k_fg_aggregate, fg_grid(count, cus), dim3(256), 0, st, pair, static_cast<const CycEntry*>(cyc), cyc_mask, warm, tp,
                       first, count, cap_rel, max_len, max_t, P);
    return hipGetLastError();
}
size_t fg_cyc_entry_bytes() { return sizeof(CycEntry); }

}  // namespace bsx
