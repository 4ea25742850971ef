// Host-side helpers shared by the translation units behind include/bsx.h (bsx_api.cpp: handle, network, problem
// space, target, simulate; bsx_attract_api.cpp: attract): kernel launch prototypes, range checks, wide integers for
// the sums of sweeps beyond 2^64 problems, the cube analysis.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdint>
#include <string>
#include <vector>

#include "bsx_engine.h"

namespace bsx {
hipError_t launch_attract(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P);
hipError_t launch_attract_fast(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P);
hipError_t launch_target(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const TargetParams& P);
hipError_t launch_simulate(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const SimParams& P);
hipError_t launch_simulate_sliced(int nw, int k, dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P);
hipError_t launch_simulate_sliced64(int nw, int k, dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P);
hipError_t launch_compact(const uint32_t* t_hit, uint64_t count, uint32_t* seg_counts, const uint64_t* seg_base,
                          HitRec* hits, uint64_t hits_cap, bool write_pass, hipStream_t st);
hipError_t configure_attract(int nw, int k, int lut_mode, size_t shmem);
hipError_t configure_attract_fast(int nw, int k, int lut_mode, size_t shmem, int* blocks_per_cu);
hipError_t launch_attract_pool(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P);
hipError_t configure_attract_pool(int nw, int k, int lut_mode, size_t shmem, int* blocks_per_cu);
size_t pool_extra_bytes(uint32_t nw);
size_t pool_lower_extra_bytes(uint32_t nw);
hipError_t launch_digit_lifetimes(int nw, int k, int lut_mode, size_t shmem, hipStream_t st, const LifetimeParams& P);
hipError_t launch_compact_near(const uint32_t* seg, const uint32_t* counts, uint32_t n_seg, uint64_t cap, uint32_t nw, uint32_t* out, LevelDesc* desc, hipStream_t stream);
hipError_t launch_publish(const uint32_t* src, uint32_t* host_dst, uint32_t words, uint32_t* host_flag, uint32_t seq, unsigned int* ticket,
                          hipStream_t stream);
hipError_t launch_fg_succ(int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const DevNet& net, const DevSpace& sp,
                          uint64_t n_states, uint32_t* succ, uint32_t warm_steps);
hipError_t launch_fg_double(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t cus, hipStream_t st);
hipError_t launch_fg_mark(const uint32_t* land, uint64_t n, uint32_t* bits, uint32_t cus, hipStream_t st);
hipError_t launch_fg_collect(uint32_t* bits, uint64_t n_words, uint32_t* cand, uint32_t cand_cap, unsigned int* cursor, uint32_t cus, hipStream_t st);
hipError_t launch_fg_cycles(const uint32_t* succ, const uint32_t* cand, uint32_t n_cand, uint64_t walk_cap, void* cyc, uint32_t cyc_mask,
                            unsigned int* n_cyclic, unsigned int* n_open, hipStream_t st);
hipError_t launch_fg_pair_init(const uint32_t* succ, const void* cyc, uint32_t cyc_mask, unsigned long long* pair, uint64_t n, uint32_t cus, hipStream_t st);
hipError_t launch_fg_pair_jump(unsigned long long* pair, uint64_t n, uint32_t d_cap, unsigned int* changed, uint32_t cus, hipStream_t st);
hipError_t launch_fg_aggregate(const unsigned long long* pair, const void* cyc, uint32_t cyc_mask, const uint32_t* warm, uint32_t tp,
                               uint64_t first, uint64_t count,
                               uint64_t cap_rel, uint64_t max_len, uint64_t max_t, const AttractParams& P, uint32_t cus, hipStream_t st);
size_t fg_cyc_entry_bytes();
hipError_t launch_table_drain(LogRec* tab, uint64_t slots, LogRec* out, uint64_t out_cap, unsigned long long* cursor, hipStream_t st);
hipError_t configure_target(int nw, int k, int lut_mode, size_t shmem);
hipError_t configure_simulate(int nw, int k, int lut_mode, size_t shmem);
}  // namespace bsx

namespace bsx {

typedef unsigned __int128 u128;

struct Launch {
    dim3 grid;
    uint32_t chunk;
};

inline Launch plan_persistent(const bsx_engine* h, uint64_t count, size_t shmem) {
    const uint32_t cus = (uint32_t)h->prop.multiProcessorCount;
    uint32_t per_cu = (uint32_t)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(shmem, 1));
    per_cu = std::max(1u, std::min(per_cu, 4u));
    uint64_t blocks = (uint64_t)cus * per_cu;
    const uint64_t need = (count + kBlock - 1) / kBlock;
    blocks = std::max<uint64_t>(1, std::min(blocks, need));
    const uint64_t waves = blocks * kWavesPerBlock;
    uint64_t chunk = count / (waves * 8);
    chunk = std::min<uint64_t>(4096, std::max<uint64_t>(64, chunk));
    chunk = (chunk / 64) * 64;
    return Launch{dim3((uint32_t)blocks), (uint32_t)chunk};
}

// [first, first + count) must lie inside the problem space (2^n_any initial states x variants): the fast
// enumeration paths add the offset to the digits without looking, so an over-long range would otherwise
// spill into nodes that are not 'any' and return plausible but wrong counts.
inline int check_range(bsx_handle h, const bsx_index* first, u128 count) {
    if (!first) return fail(h, BSX_ERR_INVALID, "first index is null");
    const uint32_t n_any = h->sp.n_any;
    for (uint32_t b = n_any; b < 64 * BSX_MAX_WORDS; ++b)
        if ((first->init_digits[b >> 6] >> (b & 63)) & 1ull)
            return fail(h, BSX_ERR_INVALID, "init_digits has bits at or above n_any");
    if (!h->variant_count_saturated && first->variant >= h->variant_count)
        return fail(h, BSX_ERR_INVALID, "variant number outside the problem space");
    if (count == 0) return BSX_OK;
    // last = init_digits + (count - 1), up to 257 bits; what lies above bit n_any carries into the variant
    uint64_t sum[6] = {0, 0, 0, 0, 0, 0};
    u128 add = count - 1, carry = 0;
    for (int w = 0; w < 5; ++w) {
        carry += (w < 4 ? first->init_digits[w] : 0ull);
        carry += (uint64_t)add;
        add >>= 64;
        sum[w] = (uint64_t)carry;
        carry >>= 64;
    }
    sum[5] = (uint64_t)carry;
    u128 over = 0;                                          // (sum >> n_any); at most 129 bits, of which 128 are kept ...
    for (uint32_t b = n_any; b < 384 && b < n_any + 128; ++b)
        over |= (u128)((sum[b >> 6] >> (b & 63)) & 1ull) << (b - n_any);
    bool beyond = false;                                    // ... and anything above them is past every space
    for (uint32_t b = n_any + 128; b < 384; ++b) beyond = beyond || ((sum[b >> 6] >> (b & 63)) & 1ull);
    const u128 last_variant = (u128)first->variant + over;
    if (beyond || (last_variant >> 64) != 0 || (!h->variant_count_saturated && (uint64_t)last_variant >= h->variant_count))
        return fail(h, BSX_ERR_INVALID, "first + count runs past the end of the problem space");
    return BSX_OK;
}

inline int check_max_t(bsx_handle h, uint64_t max_t) {
    if (max_t != BSX_T_INF && max_t < h->tp_max)
        return fail(h, BSX_ERR_INVALID, "max_t is below the last perturbation time (origin schedule or a variation)");
    return BSX_OK;
}

inline void set_first(DevSpace& sp, const bsx_index* first) {
    for (int w = 0; w < 4; ++w) sp.first_digits[w] = first->init_digits[w];
    sp.first_variant = first->variant;
}

// first + delta as a problem index: the digits at or above n_any carry into the variant number (batching.py:212-255)
inline bsx_index index_plus(const bsx_index& first, u128 delta, uint32_t n_any) {
    bsx_index r = first;
    u128 carry = 0;
    for (int w = 0; w < 4; ++w) {
        carry += r.init_digits[w];
        carry += (uint64_t)delta;
        delta >>= 64;
        r.init_digits[w] = (uint64_t)carry;
        carry >>= 64;
    }
    if (n_any < 256) {
        uint64_t over = 0;
        for (uint32_t b = n_any; b < 256 && b < n_any + 64; ++b) {
            over |= ((r.init_digits[b >> 6] >> (b & 63)) & 1ull) << (b - n_any);
            r.init_digits[b >> 6] &= ~(1ull << (b & 63));
        }
        for (uint32_t b = n_any + 64; b < 256; ++b) r.init_digits[b >> 6] &= ~(1ull << (b & 63));   // (check_range: none set)
        r.variant += over;
    }
    return r;
}

inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---- wide unsigned integers: counts and sums of sweeps over more than 2^64 problems ------------------------------
struct U256 {
    uint64_t w[4] = {0, 0, 0, 0};
    void add_at(uint64_t v, int word) {                     // += v << (64 * word)
        for (int i = word; i < 4 && v; ++i) { const uint64_t s = w[i] + v; v = s < v ? 1 : 0; w[i] = s; }
    }
    void add(const U256& o) {
        uint64_t carry = 0;
        for (int i = 0; i < 4; ++i) {
            const u128 s = (u128)w[i] + o.w[i] + carry;
            w[i] = (uint64_t)s; carry = (uint64_t)(s >> 64);
        }
    }
    void add128(u128 v) { U256 t; t.w[0] = (uint64_t)v; t.w[1] = (uint64_t)(v >> 64); add(t); }
    // += (hi:lo) << shift, shift < 128
    void add_shifted(uint64_t lo, uint64_t hi, uint32_t shift) {
        U256 t;
        t.w[0] = lo; t.w[1] = hi;
        const uint32_t ws = shift >> 6, bs = shift & 63;
        U256 r;
        for (int i = 3; i >= 0; --i) {
            const int src = i - (int)ws;
            uint64_t v = 0;
            if (src >= 0) v = t.w[src] << bs;
            if (bs && src - 1 >= 0) v |= t.w[src - 1] >> (64 - bs);
            r.w[i] = v;
        }
        add(r);
    }
    // += sign-extended v (two's complement): the result is known to be non-negative
    void add_signed(int64_t v) {
        U256 t;
        t.w[0] = (uint64_t)v;
        t.w[1] = t.w[2] = t.w[3] = v < 0 ? ~0ull : 0ull;
        add(t);
    }
    // += a * b
    void add_mul(u128 a, uint64_t b) {
        const u128 p0 = (u128)(uint64_t)a * b, p1 = (u128)(uint64_t)(a >> 64) * b;
        U256 t;
        t.w[0] = (uint64_t)p0;
        const u128 mid = (p0 >> 64) + (uint64_t)p1;
        t.w[1] = (uint64_t)mid;
        t.w[2] = (uint64_t)((mid >> 64) + (uint64_t)(p1 >> 64));
        add(t);
    }
    bool fits(int words) const { for (int i = words; i < 4; ++i) if (w[i]) return false; return true; }
};

// One aggregated attractor inside the library (every sum wide; narrowed to the caller's record at the boundary).
struct WideRec {
    uint64_t key[BSX_MAX_WORDS] = {0, 0, 0, 0};
    uint64_t length = 0;
    u128 count = 0;
    U256 sum_l, sum_l2;
};

// attractor key as the table key of the host-side merge (zero padded to the longest state)
using Key8 = std::array<uint32_t, kMaxW32>;
struct Key8Hash {
    size_t operator()(const Key8& k) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (uint32_t w : k) h = (h ^ w) * 0xBF58476D1CE4E5B9ull;
        return (size_t)(h ^ (h >> 29));
    }
};
inline Key8 key8(const uint32_t* words) { Key8 k; std::copy(words, words + kMaxW32, k.begin()); return k; }

// ---- cube collapse: which of the `a` lowest initial-state digits can the FIRST update of the
// block starting at digit value d_lo depend on?  (bsx_attract_api.cpp: build_cube / plan_cube; target's summary
// passes use them too)
struct Cube {
    uint64_t d_lo;              // first digit value (multiple of 2^a)
    uint32_t a;                 // log2 of the problems in the block
    // A SUB-BLOCK fixes some of the block's a lowest digits as well (fix_mask / fix_vals over the digit index): the block is the
    // disjoint union of the sub-blocks of a split, and fixing a well-chosen digit makes many others irrelevant
    // (bsx_attract_api.cpp: plan_split).  free_digits = the digits that vary, n_free of them: the sub-block has 2^n_free problems.
    uint64_t fix_mask = 0, fix_vals = 0, free_digits = 0;
    uint32_t n_free = 0;
    std::vector<uint32_t> rel;  // relevant digits: ascending from build_cube, then in class-index bit order
    uint32_t base[kMaxW32];     // the block's fixed bits, free bits zero
    DevSpace sp;                // enumeration of the relevant digits' assignments (plan_cube)
    uint32_t umask[kMaxW32];    // node bits of the irrelevant free digits
    uint32_t free_mask[kMaxW32];
    bool ok = false;            // false: more deposit runs than the kernels take
};
void build_cube(const bsx_engine* h, uint64_t d_lo, uint32_t a, Cube& c, const uint32_t* fixmask = nullptr,
                uint64_t fix_mask = 0, uint64_t fix_vals = 0);
void plan_cube(const bsx_engine* h, Cube& c);

constexpr uint32_t kCubeMinBits = 16;           // cube collapse: smallest aligned block handled as a cube
constexpr uint32_t kCubeMaxBits = 63;           // ... and the largest (member counts are 64-bit; a 2^64 space is two blocks)

}  // namespace bsx
