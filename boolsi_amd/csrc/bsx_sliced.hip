// bit-sliced simulate kernel (final states of fixed-length runs)
#include "bsx_kernels_common.h"

namespace bsx {

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

// ------------------------------------------------------------------------------------------------
// Second generation for K <= 3 and n <= 128 (config 5's shape): the per-row constants never change
// during a run, so each wave keeps them in registers for the whole launch.
//   * workgroup = 16 waves on one CU, group = 4096 trajectories; lane l owns trajectories 64l..64l+63 as
//     one 8-byte element per row, so a row is 512 contiguous bytes and every access is a conflict-free
//     ds_read_b64 / ds_write_b64 (256 B/clk instead of 128 for the 4-byte rows of the kernel above);
//   * wave v evaluates rows [v*R/16, (v+1)*R/16), at most 8: per row K LDS byte addresses and 2^K leaf
//     words (truth-table bits as all-ones / all-zeros) live in VGPRs -- a step is K reads, 2^K - 1
//     v_bfi per 32 trajectories and one write per row, no descriptor traffic, no address arithmetic;
//   * the two state buffers are interleaved row by row (row i: 512 B of buffer 0, 512 B of buffer 1),
//     so "the other buffer" is the immediate offset 512 and the step loop is unrolled by two instead
//     of swapping pointers (a buffer of n = 128 rows is 64 KiB, beyond the 16-bit ds offset).
constexpr int kS64Waves = 16;
constexpr int kS64RowsPerWave = 8;
constexpr uint32_t kS64Group = 4096;
constexpr uint32_t kS64RowBytes = 1024;

__device__ __forceinline__ uint32_t s64_skew(uint32_t k) { return k + (k >> 6); }   // scratch index, conflict-free column reads

// v_bfi_b32 with the third operand in an SGPR: the truth-table leaves are wave-uniform all-ones / all-zeros
// words, and a VOP3 instruction may read one scalar register -- so half of them need no VGPR.
__device__ __forceinline__ uint32_t bfi_s(uint32_t sel, uint32_t a, uint32_t b_uniform) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(sel), "v"(a), "s"(b_uniform));
    return r;
}

// One step for the rows of a wave.  DIGEST: the values just computed -- s(t) of the wave's rows -- are folded into the
// rows' digest accumulators as they are written (X always, Y when `ymask` is all ones); the rare perturbation override
// of a row corrects the fold where it overwrites the row (k_simulate_sliced64).
template <int K, uint32_t RD, uint32_t WR, bool DIGEST>
__device__ __forceinline__ void s64_step(const uint32_t (&addr)[kS64RowsPerWave][K],
                                         const uint32_t (&leaf_v)[kS64RowsPerWave][1 << (K - 1)],
                                         const uint32_t (&leaf_s)[kS64RowsPerWave][1 << (K - 1)], uint32_t out_addr,
                                         uint32_t rpw, bsx_u32x2 (&acc_x)[kS64RowsPerWave], bsx_u32x2 (&acc_y)[kS64RowsPerWave],
                                         uint32_t ymask) {
    typedef const bsx_u32x2 __attribute__((address_space(3))) lds_rd;
    typedef bsx_u32x2 __attribute__((address_space(3))) lds_wr;
    // rows in flight between the reads and the writes: two (their LDS round trips overlap); the digest build, whose
    // accumulators take 32 registers, one -- with two it spilled them
    constexpr int B = DIGEST ? 1 : 2;
#pragma unroll
    for (int r0 = 0; r0 < kS64RowsPerWave; r0 += B) {
        if ((uint32_t)r0 >= rpw) break;                 // uniform
        bsx_u32x2 g[B][K];
#pragma unroll
        for (int u = 0; u < B; ++u) {
#pragma unroll
            for (int j = 0; j < K; ++j) g[u][j] = *reinterpret_cast<lds_rd*>(addr[r0 + u][j] + RD);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int r = r0 + u;
            uint32_t x[1 << (K - 1)], y[1 << (K - 1)];
#pragma unroll
            for (int i = 0; i < (1 << (K - 1)); ++i) {
                x[i] = bfi_s(g[u][0].x, leaf_v[r][i], leaf_s[r][i]);
                y[i] = bfi_s(g[u][0].y, leaf_v[r][i], leaf_s[r][i]);
            }
#pragma unroll
            for (int j = 1; j < K; ++j)
#pragma unroll
                for (int i = 0; i < (1 << (K - 1 - j)); ++i) {
                    x[i] = bfi(g[u][j].x, x[2 * i + 1], x[2 * i]);
                    y[i] = bfi(g[u][j].y, y[2 * i + 1], y[2 * i]);
                }
            bsx_u32x2 o;
            o.x = x[0]; o.y = y[0];
            if constexpr (DIGEST) {
                acc_x[r].x ^= o.x; acc_x[r].y ^= o.y;
                acc_y[r].x ^= o.x & ymask; acc_y[r].y ^= o.y & ymask;
            }
            if ((uint32_t)r < rpw) *reinterpret_cast<lds_wr*>(out_addr + (uint32_t)r * kS64RowBytes + WR) = o;     // uniform
        }
    }
}

template <int NW, int K, bool DIGEST>
__global__ __launch_bounds__(64 * kS64Waves, 4) void k_simulate_sliced64(const SlicedParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    typedef bsx_u32x2 __attribute__((address_space(3))) lds_v2;
    // absolute LDS addressing (as net_step): the dynamic block must start at LDS address 0
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)smem != 0u) __builtin_trap();
    // (readfirstlane tells the compiler the wave index is uniform)
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n = P.n_nodes, rows = P.n_rows;          // rows: multiple of 16, <= 128
    const uint32_t rpw = rows / kS64Waves;
    const uint32_t scratch = rows * kS64RowBytes;           // byte address of the transposition scratch

    const uint32_t out_addr = wave * rpw * kS64RowBytes + lane * 8u;

    // bits `bit` of the rows at byte offset `src` (+ node * row bytes) -> the NW state words of one trajectory
    auto gather = [&](uint32_t src, uint32_t bit, uint32_t (&s)[NW]) {
        // (n made opaque here: otherwise the 128 `node < n` conditions are evaluated once in front of everything and
        // held in scalar registers -- spilled to VGPR lanes -- through the step loop)
        uint32_t n_here = n;
        asm volatile("" : "+s"(n_here));
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            uint32_t word = 0;
            for (uint32_t b = 0; b < 32; ++b) {
                const uint32_t node = (uint32_t)w * 32 + b;
                if (node < n_here) word |= ((*reinterpret_cast<lds_u32*>(node * kS64RowBytes + src) >> bit) & 1u) << b;
            }
            s[w] = word;
        }
    };

    const uint64_t n_groups = (P.count + kS64Group - 1) / kS64Group;
    for (uint64_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
        const uint64_t base = group * kS64Group;
        // ---- initial states into buffer 0, one 32-node word at a time through the scratch
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            __syncthreads();
            {
                BSX_KERNARG(SlicedParams, Pk);      // (the enumeration's uniform conditions on the space stay in this block)
                for (uint32_t k = threadIdx.x; k < kS64Group; k += blockDim.x) {
                    Problem<NW> pr;
                    pr.s[w] = 0;
                    if (base + k < Pk->count) init_problem<NW>(Pk->sp, base + k, pr);
                    *reinterpret_cast<lds_u32*>(scratch + s64_skew(k) * 4u) = pr.s[w];
                }
            }
            __syncthreads();
            // (lane made opaque: the 64 scratch addresses below are otherwise computed once in front of the group loop,
            // kept for the whole launch and spilled to scratch)
            uint32_t lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            for (uint32_t b = wave * 2; b < wave * 2 + 2; ++b) {        // 16 waves x 2 = the word's 32 nodes
                const uint32_t node = (uint32_t)w * 32 + b;
                if (node >= rows) break;
                uint32_t lo = 0, hi = 0;
                if (node < n) {
                    for (uint32_t k = 0; k < 32; ++k) {
                        lo |= ((*reinterpret_cast<lds_u32*>(scratch + s64_skew(lane_t * 64 + k) * 4u) >> b) & 1u) << k;
                        hi |= ((*reinterpret_cast<lds_u32*>(scratch + s64_skew(lane_t * 64 + 32 + k) * 4u) >> b) & 1u) << k;
                    }
                }
                *reinterpret_cast<lds_u32*>(node * kS64RowBytes + lane_t * 8u) = lo;
                *reinterpret_cast<lds_u32*>(node * kS64RowBytes + lane_t * 8u + 4u) = hi;
            }
        }
        __syncthreads();

        // ---- the run's constants, loaded and expanded HERE, in front of every run, through an opaque pointer (so that the
        // expansion cannot move in front of the group loop): per row K LDS addresses and 2^K leaf words -- leaf 2i+1 in a
        // VGPR, leaf 2i in an SGPR (a VOP3 instruction reads one scalar register).  Held across the transpositions they
        // were spilled to VGPR lanes and read back in every step (20 of the step's 80 instructions).
        uint32_t addr[kS64RowsPerWave][K], leaf_v[kS64RowsPerWave][1 << (K - 1)], leaf_s[kS64RowsPerWave][1 << (K - 1)];
        bsx_u32x2 acc_x[kS64RowsPerWave], acc_y[kS64RowsPerWave];
        {
            const uint32_t* desc = P.desc;
            asm volatile("" : "+s"(desc));
#pragma unroll
            for (int r = 0; r < kS64RowsPerWave; ++r) {
                const bool live = (uint32_t)r < rpw;
                const uint32_t* d = desc + (size_t)(live ? wave * rpw + (uint32_t)r : 0u) * 8;      // uniform: scalar loads
#pragma unroll
                for (int j = 0; j < K; ++j) addr[r][j] = d[j] * kS64RowBytes + lane * 8u;
                const uint32_t tt = live ? d[6] : 0u;           // 2^K <= 8 table bits
#pragma unroll
                for (int i = 0; i < (1 << (K - 1)); ++i) {
                    leaf_v[r][i] = 0u - ((tt >> (2 * i + 1)) & 1u);
                    leaf_s[r][i] = __builtin_amdgcn_readfirstlane(0u - ((tt >> (2 * i)) & 1u));
                }
                acc_x[r].x = acc_x[r].y = 0; acc_y[r].x = acc_y[r].y = 0;
                if constexpr (DIGEST) {                         // s(0) of the wave's rows (t = 0 belongs to Y iff digest_ybit(0))
                    if (live) {
                        const uint32_t ymask0 = 0u - digest_ybit(0u);
                        const bsx_u32x2 v = *reinterpret_cast<lds_v2*>(out_addr + (uint32_t)r * kS64RowBytes);
                        acc_x[r] = v;
                        acc_y[r].x = v.x & ymask0; acc_y[r].y = v.y & ymask0;
                    }
                }
            }
        }

        // ---- T synchronous updates, buffer parity = (t - 1) & 1
        // (32-bit clocks: max_t < 2^30, bsx_api.cpp)
        uint32_t sched_at = 0;
        const uint32_t t_end = (uint32_t)P.max_t, n_sched = P.n_sched;
        const uint32_t* const sched = P.sched;
        uint32_t next_t = n_sched ? sched[0] : 0xFFFFFFFFu;
        for (uint32_t t = 1; t <= t_end; ++t) {
            const bool odd = (t & 1u) != 0;                 // odd steps read buffer 0 and write buffer 1
            const uint32_t ymask = DIGEST ? 0u - digest_ybit(t) : 0u;     // the step folds what it writes: s(t)
            if (odd) s64_step<K, 0u, 512u, DIGEST>(addr, leaf_v, leaf_s, out_addr, rpw, acc_x, acc_y, ymask);
            else s64_step<K, 512u, 0u, DIGEST>(addr, leaf_v, leaf_s, out_addr, rpw, acc_x, acc_y, ymask);
            // perturbation override at time t (model.py:68-71): whole rows of the buffer just written.
            // Every wave walks the (uniform) schedule and the wave that owns the row overwrites what it
            // has just stored, so the step's one barrier covers the override too.
            while (next_t <= t) {
                if (next_t == t) {
                    const uint32_t node = sched[3 * sched_at + 1];
                    const uint32_t mine = node - wave * rpw;            // uniform
                    if (mine < rpw) {
                        const uint32_t v = sched[3 * sched_at + 2] ? 0xFFFFFFFFu : 0u;
                        const uint32_t a = node * kS64RowBytes + (odd ? 512u : 0u) + lane * 8u;
                        if constexpr (DIGEST) {
                            // the fold took the computed value: replace it by the overriding one (the wave reads its own store)
                            const bsx_u32x2 was = *reinterpret_cast<lds_v2*>(a);
                            const uint32_t dx = was.x ^ v, dy = was.y ^ v;
#pragma unroll
                            for (int r = 0; r < kS64RowsPerWave; ++r) {
                                const uint32_t sel = (uint32_t)r == mine ? 0xFFFFFFFFu : 0u;    // (static indices: the arrays stay in registers)
                                acc_x[r].x ^= dx & sel; acc_x[r].y ^= dy & sel;
                                acc_y[r].x ^= dx & sel & ymask; acc_y[r].y ^= dy & sel & ymask;
                            }
                        }
                        *reinterpret_cast<lds_u32*>(a) = v;
                        *reinterpret_cast<lds_u32*>(a + 4u) = v;
                    }
                }
                ++sched_at;
                next_t = sched_at < n_sched ? sched[3 * sched_at] : 0xFFFFFFFFu;
            }
            __syncthreads();
        }
        const uint32_t cur = (P.max_t & 1ull) ? 512u : 0u;      // buffer holding s(max_t)

        // ---- digests: X and Y rows go through the idle buffer, one after the other
        uint64_t dg[kS64Group / (64 * kS64Waves)];
        if constexpr (DIGEST) {
            const uint32_t idle = cur ^ 512u;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int r = 0; r < kS64RowsPerWave; ++r) {
                    if ((uint32_t)r >= rpw) break;
                    *reinterpret_cast<lds_v2*>(out_addr + (uint32_t)r * kS64RowBytes + idle) = pass ? acc_y[r] : acc_x[r];
                }
                __syncthreads();
                uint32_t q = 0;
                for (uint32_t k = threadIdx.x; k < kS64Group; k += blockDim.x, ++q) {
                    uint32_t s[NW];
                    gather((k >> 6) * 8u + ((k >> 5) & 1u) * 4u + idle, k & 31u, s);
                    dg[q] = digest_fold_words<NW>(pass ? dg[q] : kDigestSeed, s, P.w64);
                }
                __syncthreads();
            }
        }

        // ---- final states: thread handles trajectories tid, tid + 1024, ...
        uint32_t q = 0;
        BSX_KERNARG(SlicedParams, Po);
        const uint64_t count_o = Po->count;
        const uint32_t w64_o = Po->w64;
        uint64_t* const final_o = Po->final_states;
        uint64_t* const digests_o = Po->digests;
        for (uint32_t k = threadIdx.x; k < kS64Group; k += blockDim.x, ++q) {
            if (base + k >= count_o) break;
            uint32_t s[NW];
            gather((k >> 6) * 8u + ((k >> 5) & 1u) * 4u + cur, k & 31u, s);
            if constexpr (DIGEST) digests_o[base + k] = digest_fold_words<NW>(dg[q], s, w64_o);
            if (final_o) {
#pragma unroll
                for (int w = 0; w < (NW + 1) / 2; ++w) {
                    uint64_t word = s[2 * w];
                    if (2 * w + 1 < NW) word |= (uint64_t)s[2 * w + 1] << 32;
                    if ((uint32_t)w < w64_o) final_o[(base + k) * w64_o + w] = word;
                }
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
static hipError_t launch_sliced_nk(int, dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P) {
    hipError_t e = hipFuncSetAttribute((const void*)k_simulate_sliced<NW, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_simulate_sliced<NW, K>), grid, dim3(64 * kSlicedWaves), shmem, st, P);
    return hipGetLastError();
}

template <int NW, int K>
static hipError_t launch_sliced64_nk(dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P) {
    const void* fn = P.digests ? (const void*)k_simulate_sliced64<NW, K, true> : (const void*)k_simulate_sliced64<NW, K, false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    void* args[] = {const_cast<SlicedParams*>(&P)};
    return hipLaunchKernel(fn, grid, dim3(64 * kS64Waves), args, shmem, st);
}

hipError_t launch_simulate_sliced(int nw, int k, dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P) {
    const int lut_mode = 0;        // unused by this kernel (no gather LUT)
    BSX_DISPATCH(launch_sliced_nk)
}

// K <= 3, n <= 128 (rows a multiple of 16): shmem = rows * 1024 + (4096 + 64) * 4 bytes, one workgroup per CU
hipError_t launch_simulate_sliced64(int nw, int k, dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P) {
    switch (nw * 8 + k) {
        case 1 * 8 + 1: return launch_sliced64_nk<1, 1>(grid, shmem, st, P);
        case 1 * 8 + 2: return launch_sliced64_nk<1, 2>(grid, shmem, st, P);
        case 1 * 8 + 3: return launch_sliced64_nk<1, 3>(grid, shmem, st, P);
        case 2 * 8 + 1: return launch_sliced64_nk<2, 1>(grid, shmem, st, P);
        case 2 * 8 + 2: return launch_sliced64_nk<2, 2>(grid, shmem, st, P);
        case 2 * 8 + 3: return launch_sliced64_nk<2, 3>(grid, shmem, st, P);
        case 4 * 8 + 1: return launch_sliced64_nk<4, 1>(grid, shmem, st, P);
        case 4 * 8 + 2: return launch_sliced64_nk<4, 2>(grid, shmem, st, P);
        case 4 * 8 + 3: return launch_sliced64_nk<4, 3>(grid, shmem, st, P);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace bsx
