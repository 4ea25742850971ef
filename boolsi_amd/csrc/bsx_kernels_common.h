#pragma once
// Device functions shared by the gfx950 kernels of the Boolean-network state-update engine
// (bsx_attract.hip, bsx_target.hip, bsx_simulate.hip).
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
#include <type_traits>
#include "bsx_device.h"

namespace bsx {

// (a & sel) | (b & ~sel) in one VALU op.  Written as asm because with loop-invariant a/b the compiler
// prefers two ops (and + xor with a precomputed a^b), which doubles the cost of the mux tree.
__device__ __forceinline__ uint32_t bfi(uint32_t sel, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(sel), "v"(a), "v"(b));
    return r;
}

// The kernel's (only, by-value) parameter struct as an opaque pointer into the kernel-argument segment (offset 0).
// What is read through it is loaded where it is used: the compiler cannot hoist the loads, or anything computed from
// them, in front of the surrounding loop -- which is what it does with accesses through the parameter itself, and what
// fills the scalar registers of the hot loop with values only a rare or one-off block needs.
#define BSX_KERNARG(Type, name)                                                                          \
    const Type* name = (const Type*)__builtin_amdgcn_kernarg_segment_ptr();                              \
    asm volatile("" : "+s"(name))

// Number of set bits of `mask` below this lane's position (a lane's rank among the flagged lanes): v_mbcnt_lo/hi, no
// per-lane (1 << lane) - 1 constant that would have to live in two VGPRs across the loop.
__device__ __forceinline__ uint32_t rank_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
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
// LUT modes: where the gather LUT lives and how the state is cut into lookups.
constexpr int kLutGlobal = 0;       // one entry per state byte, read through L2 (the table does not fit LDS)
constexpr int kLutLdsByte = 1;      // one entry per state byte, in LDS
constexpr int kLutLdsNibble = 2;    // one entry per 4 state bits, in LDS: 16x smaller table, twice the lookups
                                    // (networks beyond 64 nodes whose byte table would not fit)

template <int NW, int K, int LM = kLutGlobal>
struct NetView {
    static constexpr bool LDS = LM != kLutGlobal;
    static constexpr bool kMasksInRegs = (K <= 3) || (K == 4 && NW == 1);     // up to 16 registers beyond K = 3
    // LDS = true: the gather LUT sits in LDS right behind the masks, and the kernel's dynamic LDS starts
    // at LDS address 0 (checked by stage_network), so an entry's address is a compile-time constant plus
    // the scaled byte -- no base-pointer add per lookup.
    static constexpr uint32_t kLutLdsAddr = ((((1u << K) * NW) + 3u) & ~3u) * 4u;
    const uint32_t* lut;     // LDS or global
    const uint32_t* masks;   // LDS (used when the 2^K * NW mask words do not fit the register budget)
    uint32_t mreg[kMasksInRegs ? (1 << K) * NW : 1];
    uint32_t n_wide;
    const uint32_t* wide_desc;
    const uint32_t* wide_preds;
    const uint32_t* wide_tt;
};

// ((v >> 8*B) & 0xFF) * scale in one instruction (SDWA byte select on the second operand; the scale
// lives in a VGPR because SDWA takes no literal).
__device__ __forceinline__ uint32_t byte_times(uint32_t v, uint32_t scale, int B) {   // B: constant after unrolling
    uint32_t r;
    if (B == 0)
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(scale), "v"(v));
    else if (B == 1)
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(scale), "v"(v));
    else if (B == 2)
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(scale), "v"(v));
    else
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(scale), "v"(v));
    return r;
}

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

// Same from an LDS byte address (address space 3 pointer made from the integer).
typedef uint32_t bsx_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t bsx_u32x4 __attribute__((ext_vector_type(4)));
template <int N>
__device__ __forceinline__ void load_entry_lds(uint32_t byte_addr, uint32_t (&dst)[N]) {
    typedef const uint32_t __attribute__((address_space(3))) lds_u32;
    typedef const bsx_u32x2 __attribute__((address_space(3))) lds_u32x2;
    typedef const bsx_u32x4 __attribute__((address_space(3))) lds_u32x4;
    if constexpr (N % 4 == 0) {
        lds_u32x4* p = reinterpret_cast<lds_u32x4*>(byte_addr);
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const bsx_u32x4 v = p[i];
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
        }
    } else if constexpr (N % 2 == 0) {
        lds_u32x2* p = reinterpret_cast<lds_u32x2*>(byte_addr);
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const bsx_u32x2 v = p[i];
            dst[2 * i] = v.x; dst[2 * i + 1] = v.y;
        }
    } else {
        lds_u32* p = reinterpret_cast<lds_u32*>(byte_addr);
#pragma unroll
        for (int i = 0; i < N; ++i) dst[i] = p[i];
    }
}

// One synchronous update of all nodes (model.py:16-28) + fixed nodes as constants (model.py:31-49).
// `has_fixed` (wave-uniform) = some node is fixed; without fixed nodes the final mask pass is skipped.
template <int NW, int K, int LM>
__device__ __forceinline__ void net_step(const NetView<NW, K, LM>& nv, const uint32_t (&s)[NW],
                                         const uint32_t (&fm)[NW], const uint32_t (&fv)[NW],
                                         uint32_t (&out)[NW], bool has_fixed = true) {
    uint32_t g[K][NW];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int w = 0; w < NW; ++w) g[j][w] = 0;

    // gather: one LUT entry per 8 (or 4) state bits.  All lookups are issued unconditionally (the LUT
    // is zero-padded to whole 32-bit words of state) so the LDS reads overlap instead of each
    // waiting behind a branch.
    // Lookups are issued in batches sized to keep the in-flight entries within ~64 registers.
    constexpr bool LDS = LM != kLutGlobal;
    constexpr int kEntry = K * NW;
    constexpr int kLookups = (LM == kLutLdsNibble) ? NW * 8 : NW * 4;
    constexpr int kBatch = (64 / kEntry) < 1 ? 1 : ((64 / kEntry) > kLookups ? kLookups : (64 / kEntry));
#pragma unroll
    for (int c0 = 0; c0 < kLookups; c0 += kBatch) {
        uint32_t e[kBatch][kEntry];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int ch = c0 + b;
            if (ch < kLookups) {
                if constexpr (LM == kLutLdsNibble) {
                    const uint32_t off = ((s[ch >> 3] >> ((ch & 7) * 4)) & 15u) * (uint32_t)(kEntry * 4);
                    load_entry_lds<kEntry>(NetView<NW, K, LM>::kLutLdsAddr + (uint32_t)(ch << 4) * kEntry * 4u + off, e[b]);
                } else if constexpr (kEntry == 6) {
                    // 24-byte entries (n <= 64 with K = 3, n <= 32 with K = 6) are stored as two planes -- the first four
                    // words of every entry (16 bytes, one b128 read), then the last two (one b64 read): three b64 reads at
                    // a 24-byte stride kept the LDS pipeline 95 % busy with two thirds of it bank conflicts (k_attract<2,3,1>
                    // on a chaotic network, profiles/r03_pmc_chaotic.json)
                    const uint32_t off = byte_times(s[ch >> 2], 16u, ch & 3);
                    constexpr uint32_t kPlaneB = (uint32_t)kLookups * 256u * 16u;       // bytes of the first plane
                    uint32_t lo4[4], hi2[2];
                    if constexpr (LDS) {
                        load_entry_lds<4>(NetView<NW, K, LM>::kLutLdsAddr + (uint32_t)(ch << 8) * 16u + off, lo4);
                        load_entry_lds<2>(NetView<NW, K, LM>::kLutLdsAddr + kPlaneB + (uint32_t)(ch << 8) * 8u + (off >> 1), hi2);
                    } else {
                        const char* base = reinterpret_cast<const char*>(nv.lut);
                        load_entry<4>(reinterpret_cast<const uint32_t*>(base + (uint32_t)(ch << 8) * 16u + off), lo4);
                        load_entry<2>(reinterpret_cast<const uint32_t*>(base + kPlaneB + (uint32_t)(ch << 8) * 8u + (off >> 1)), hi2);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) e[b][i] = lo4[i];
                    e[b][4] = hi2[0]; e[b][5] = hi2[1];
                } else {
                    // byte offset of the entry within its chunk's table: (byte ch of the state) * entry size,
                    // one SDWA multiply instead of extract + scale
                    const uint32_t off = byte_times(s[ch >> 2], (uint32_t)(kEntry * 4), ch & 3);
                    if constexpr (LDS)
                        load_entry_lds<kEntry>(NetView<NW, K, LM>::kLutLdsAddr + (uint32_t)(ch << 8) * kEntry * 4u + off, e[b]);
                    else
                        load_entry<kEntry>(reinterpret_cast<const uint32_t*>(
                                               reinterpret_cast<const char*>(nv.lut + (uint32_t)(ch << 8) * kEntry) + off), e[b]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b)
            if (c0 + b < kLookups) {
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
            r[i][w] = NetView<NW, K, LM>::kMasksInRegs
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

    if (has_fixed) {
#pragma unroll
        for (int w = 0; w < NW; ++w) out[w] = (out[w] & ~fm[w]) | fv[w];
    }
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
template <int NW, int K, int LM>
__device__ __forceinline__ NetView<NW, K, LM> stage_network(const DevNet& net, uint32_t* smem, uint32_t*& smem_free) {
    constexpr bool LDS_LUT = LM != kLutGlobal;
    NetView<NW, K, LM> nv;
    if constexpr (LDS_LUT) {
        // net_step addresses the LUT by absolute LDS byte offsets: the dynamic LDS block must start at 0
        // (true for kernels without static LDS; anything else is a build error, so stop loudly)
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)smem != 0u) __builtin_trap();
    }
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
    if constexpr (NetView<NW, K, LM>::kMasksInRegs) {
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

// Fold digest of a trajectory (simulate's checksum sink, include/bsx.h): FNV-1a over the 64-bit words of
// X = xor of all s(t), Y = xor of the s(t) at times with digest_ybit(t), and s(T).  Built from XORs of whole
// states so that the bit-sliced kernels can keep it per row.
__device__ __forceinline__ uint32_t digest_ybit(uint32_t t) { return (t * 0x9E3779B1u) >> 31; }
template <int NW>
__device__ __forceinline__ uint64_t digest_fold_words(uint64_t dg, const uint32_t (&s)[NW], uint32_t w64) {
#pragma unroll
    for (int w = 0; w < (NW + 1) / 2; ++w) {
        uint64_t word = s[2 * w];
        if (2 * w + 1 < NW) word |= (uint64_t)s[2 * w + 1] << 32;
        if ((uint32_t)w < w64) dg = (dg ^ word) * kDigestPrime;
    }
    return dg;
}

__device__ __forceinline__ uint64_t bcast64(uint64_t v, int src_lane) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src_lane);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}

// Counter totals leave a wave as ONE atomic: 64 lanes adding to the same address are 64 serialised
// operations at the L2 atomic unit, and with every wave of the grid doing that at the end of a launch
// the queue behind one address was 20 % of the lean attract kernel's run time.
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, off, 64);
        const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ void wave_atomic_add(unsigned long long* dst, unsigned long long v, int lane) {
    const unsigned long long total = wave_sum(v);
    if (lane == 0 && total) atomicAdd(dst, total);
}
__device__ __forceinline__ void wave_atomic_add(unsigned int* dst, unsigned int v, int lane) {
    const unsigned long long total = wave_sum((unsigned long long)v);
    if (lane == 0 && total) atomicAdd(dst, (unsigned int)total);
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
__device__ __forceinline__ uint32_t hash_state(const uint32_t (&s)[NW]);

// HBM attractor table: insert-or-add by key.  One loop, every exit inside the loop body: a lane that wins a
// slot finishes writing it in the same iteration in which its wave-mates find it busy, so spinning lanes of
// the same wave cannot starve the writer.
template <int NW>
__device__ __forceinline__ void table_insert(const AttractParams& P, const uint32_t (&key)[NW], uint32_t length,
                                          uint64_t count, uint64_t sl, uint64_t sl2, uint64_t sl2_hi) {
    uint32_t hsh = hash_state<NW>(key) * 0x9E3779B1u;
#pragma unroll
    for (int w = 0; w < NW; ++w) hsh = (hsh ^ key[w]) * 0x85EBCA6Bu;
    uint64_t at = (uint64_t)(hsh ^ (hsh >> 15)) & P.table_mask;
    uint32_t probes = 0, spins = 0;
    bool done = false;
    while (!done) {
        LogRec* e = P.table + at;
        uint32_t st = __hip_atomic_load(&e->pad, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        bool mine = false;
        if (st == 0u) {
            st = atomicCAS(&e->pad, 0u, 1u);
            if (st == 0u) {
#pragma unroll
                for (int w = 0; w < kMaxW32; ++w) e->key[w] = w < NW ? key[w < NW ? w : 0] : 0u;
                e->length = length;
                __hip_atomic_store(&e->pad, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                mine = true;
            }
        }
        if (!mine && st == 1u) {                // somebody is writing the key: look again (bounded)
            if (++spins > (1u << 20)) { atomicOr(&P.ctr->table_overflow, 2u); done = true; }
            continue;
        }
        bool same = mine;
        if (!mine) {                            // st == 2: ready
            uint32_t d = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) d |= __hip_atomic_load(&e->key[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ key[w];
            same = d == 0;
        }
        if (same) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&e->count), (unsigned long long)count);
            atomicAdd(reinterpret_cast<unsigned long long*>(&e->sum_l), (unsigned long long)sl);
            const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(&e->sum_l2_lo), (unsigned long long)sl2);
            const unsigned long long up = sl2_hi + ((old + sl2 < old) ? 1ull : 0ull);
            if (up) atomicAdd(reinterpret_cast<unsigned long long*>(&e->sum_l2_hi), up);
            done = true;
        } else {
            at = (at + 1) & P.table_mask;
            spins = 0;
            if (++probes > 4096u) { atomicOr(&P.ctr->table_overflow, 1u); done = true; }
        }
    }
}

// kTable: the caller's records may spill into the HBM table (general kernel); the lean / pool kernels write at
// most one record per workgroup and attractor and are built without that code.
template <int NW, bool kTable = false>
__device__ __forceinline__ void log_append(const AttractParams& P, const uint32_t (&key)[NW], uint32_t length,
                                           uint64_t count, uint64_t sl, uint64_t sl2, uint64_t sl2_hi = 0) {
    const unsigned long long at = atomicAdd(&P.ctr->log_cursor, 1ull);
    if (at < P.log_cap) {
        LogRec r;
#pragma unroll
        for (int w = 0; w < kMaxW32; ++w) r.key[w] = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) r.key[w] = key[w];
        r.length = length; r.pad = 0; r.count = count; r.sum_l = sl; r.sum_l2_lo = sl2; r.sum_l2_hi = sl2_hi;
        P.log[at] = r;
    } else if (kTable && P.table) {             // log full: straight into the HBM table
        if constexpr (kTable) {
            atomicAdd(&P.ctr->table_inserts, 1ull);
            table_insert<NW>(P, key, length, count, sl, sl2, sl2_hi);
        }
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
            log_append<NW, true>(P, k, len, cnt, a, b);    // all 64 slots taken: straight to the HBM log
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
// Bit 31 of the tag word is the "continue" flag: some later insert had to skip over this (occupied)
// slot, so a lookup that lands here on a different state must look at the next slot too.  Without the
// flag every lookup landing on an occupied slot would have to walk on, and with 64 lanes probing
// mostly non-cycle states that was the case in four wave iterations out of five.
constexpr int kCacheHeaderWords = 4;
constexpr uint32_t kTagCont = 0x80000000u;
constexpr uint32_t kTagRep = 0x40000000u;      // pool kernel, cube pass: entry is the class representative of a cycle
                                               // state (its irrelevant free bits cleared); it counts at t = 0 only
constexpr uint32_t kTagMask = 0x3FFFFFFFu;

// One probe: loads the whole entry with no control flow in between (so the reads are issued together
// with whatever else the caller has in flight) and classifies it.
template <int NW>
struct CacheProbe {
    uint32_t tagw, length;      // tag word (tag | continue flag)
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
        p.tagw = v.y; p.length = v.z; p.key[0] = v.w; p.same = v.x == s[0];
    } else if constexpr (NW == 2) {
        const uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(e, 16));
        const uint2 k = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(e + 4, 8));
        p.tagw = v.z; p.length = v.w; p.key[0] = k.x; p.key[1] = k.y;
        p.same = ((v.x ^ s[0]) | (v.y ^ s[1])) == 0;
    } else {
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { d |= e[w] ^ s[w]; p.key[w] = e[NW + 2 + w]; }
        p.tagw = e[NW]; p.length = e[NW + 1]; p.same = d == 0;
    }
    return p;
}

// Is `s` a state of an attractor with sequence number <= visible?  The first probe is branch-free;
// only lanes that land on a different state in a slot flagged "continue" keep walking the chain.
template <int NW>
__device__ __forceinline__ bool cache_lookup(const uint32_t* lc, uint32_t mask, uint32_t visible,
                                             const uint32_t (&s)[NW], uint32_t& length, uint32_t (&key)[NW],
                                             uint32_t* tag_out = nullptr) {
    const uint32_t* base = lc + kCacheHeaderWords;
    uint32_t h = hash_state<NW>(s) & mask;
    CacheProbe<NW> p = cache_probe<NW>(base, h, s);
    bool hit = ((p.tagw & kTagMask) - 1u < visible) & p.same;
    bool walking = (p.tagw >> 31) != 0 && !hit;
    if (__builtin_expect(__ballot(walking) != 0, 0)) {
        while (walking) {
            h = (h + 1) & mask;
            const CacheProbe<NW> q = cache_probe<NW>(base, h, s);
            const bool here = ((q.tagw & kTagMask) - 1u < visible) & q.same;
            if (here) { p = q; hit = true; }
            walking = (q.tagw >> 31) != 0 && !here;
        }
    }
    length = p.length;
#pragma unroll
    for (int w = 0; w < NW; ++w) key[w] = p.key[w];
    if (tag_out) *tag_out = p.tagw & kTagMask;
    return hit;
}

__device__ __forceinline__ uint32_t cache_visible(const uint32_t* lc) {
    return __hip_atomic_load(lc, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Single-thread insert into the LDS mirror (only thread 0 of a workgroup writes it).  Occupied slots
// on the way get their "continue" flag set; readers that raced past them before that miss a state of
// an attractor that is not visible to them yet anyway.
template <int NW>
__device__ __forceinline__ void cache_insert_lds(uint32_t* lc, uint32_t mask, const uint32_t (&s)[NW],
                                                 uint32_t length, const uint32_t (&key)[NW], uint32_t tag) {
    constexpr int S = CacheLayout<NW>::kStride;
    uint32_t* base = lc + kCacheHeaderWords;
    uint32_t h = hash_state<NW>(s) & mask;
    while (base[h * S + NW] != 0) {
        base[h * S + NW] |= kTagCont;
        h = (h + 1) & mask;
    }
    uint32_t* e = base + h * S;
#pragma unroll
    for (int w = 0; w < NW; ++w) { e[w] = s[w]; e[NW + 2 + w] = key[w]; }
    e[NW + 1] = length;
    e[NW] = tag;
}

// Thread 0: take the attractors published since `seen` (by any workgroup) from the HBM journal,
// regenerate their cycles from the key and make each cycle visible in the LDS mirror at once.
template <int NW, int K, int LM>
__device__ __forceinline__ void cache_pull(const CycleCache& cc, const NetView<NW, K, LM>& nv,
                                           const uint32_t (&fm)[NW], const uint32_t (&fv)[NW], uint32_t* lc,
                                           uint32_t& seen, uint32_t& n_states, uint32_t& n_attr,
                                           uint32_t max_attr = 0xFFFFFFFFu) {
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
        if (len == 0 || len > kCycleCacheMaxLen || n_states + len > cc.lds_slots / 2 || n_attr >= max_attr) continue;   // does not fit
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

// Enumeration fast path: no variations, at most 64 'any' nodes that are either nodes 0..n_any-1 or a
// few runs of consecutive nodes (simple_space below).
template <int NW>
__device__ __forceinline__ void init_problem_simple(const DevSpace& sp, uint64_t p, uint32_t (&s)[NW]) {
    const uint64_t d = sp.first_digits[0] + p;
#pragma unroll
    for (int w = 0; w < NW; ++w) s[w] = sp.origin[w];
    if (sp.identity_any) {
        s[0] |= (uint32_t)d;
        if constexpr (NW > 1) s[1] |= (uint32_t)(d >> 32);
    } else {
        for (uint32_t r = 0; r < sp.n_runs; ++r) {          // uniform trip count and run descriptors
            const uint32_t desc = sp.deposit[2 * r], mask = sp.deposit[2 * r + 1];
            const uint32_t piece = ((uint32_t)(d >> (desc & 63u)) & mask) << ((desc >> 16) & 31u);
            const uint32_t word = (desc >> 8) & 7u;
#pragma unroll
            for (int w = 0; w < NW; ++w) s[w] |= (word == (uint32_t)w) ? piece : 0u;
        }
    }
}

__device__ __forceinline__ bool simple_space(const DevSpace& sp) {
    return sp.n_any <= 64 && (sp.identity_any || sp.n_runs) && !sp.n_fv && !sp.n_pv;
}


}  // namespace bsx

// Address of the instantiation of KERNEL for a LUT mode (nibble tables are only built beyond 64 nodes).
#define BSX_KERNEL_FOR_MODE(KERNEL, NWV, KV, MODE, OUT)                                             \
    do {                                                                                            \
        OUT = nullptr;                                                                              \
        if ((MODE) == kLutLdsByte) OUT = (const void*)KERNEL<NWV, KV, kLutLdsByte>;                 \
        else if ((MODE) == kLutGlobal) OUT = (const void*)KERNEL<NWV, KV, kLutGlobal>;              \
        else if constexpr ((NWV) >= 4) { if ((MODE) == kLutLdsNibble) OUT = (const void*)KERNEL<NWV, KV, kLutLdsNibble>; } \
    } while (0)

// Launch dispatch over (NW, K): NW in {1,2,4,8}; K in 1..6.
#define BSX_DISPATCH_K(FN, NWV)                                                         \
    switch (k) {                                                                        \
        case 1: return FN<NWV, 1>(lut_mode, grid, shmem, st, P);                             \
        case 2: return FN<NWV, 2>(lut_mode, grid, shmem, st, P);                             \
        case 3: return FN<NWV, 3>(lut_mode, grid, shmem, st, P);                             \
        case 4: return FN<NWV, 4>(lut_mode, grid, shmem, st, P);                             \
        case 5: return FN<NWV, 5>(lut_mode, grid, shmem, st, P);                             \
        case 6: return FN<NWV, 6>(lut_mode, grid, shmem, st, P);                             \
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
