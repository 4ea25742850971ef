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

template <int NW, int K>
static hipError_t launch_sliced_nk(bool, dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P) {
    hipError_t e = hipFuncSetAttribute((const void*)k_simulate_sliced<NW, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_simulate_sliced<NW, K>), grid, dim3(64 * kSlicedWaves), shmem, st, P);
    return hipGetLastError();
}


hipError_t launch_simulate_sliced(int nw, int k, dim3 grid, size_t shmem, hipStream_t st, const SlicedParams& P) {
    const bool lds = true;
    BSX_DISPATCH(launch_sliced_nk)
}

}  // namespace bsx
