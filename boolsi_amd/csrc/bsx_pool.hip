// Class-pool attract kernel: dispatch over the state width (the kernels are built per width in bsx_pool_nw*.hip),
// the packing kernel between the levels of a cube cascade, and the kernel's LDS footprint.
#include "bsx_kernels_common.h"

namespace bsx {

constexpr int kPoolWaves = kPoolBlockThreads / 64;

// ------------------------------------------------------------------------------------------------
// Deep cube passes: the workgroups' segments of listed classes -> one contiguous list (block g copies segment g
// to where the segments before it end), and the list's length -> the descriptor the next level's launch reads
// (that launch is already enqueued: the host does not look at anything between the levels of a cascade).
__global__ __launch_bounds__(256) void k_compact_near(const uint32_t* seg, const uint32_t* counts, uint32_t n_seg,
                                                      uint64_t cap, uint32_t nw, uint32_t* out, LevelDesc* desc) {
    __shared__ unsigned long long before;
    const uint32_t listed = counts[blockIdx.x];
    if (listed > cap && threadIdx.x == 0) atomicOr(&desc->abort, 1u);      // a segment was too small: the list is incomplete
    if (listed == 0 && blockIdx.x != n_seg - 1) return;                    // (most segments of a short list)
    if (threadIdx.x == 0) before = 0;
    __syncthreads();
    unsigned long long part = 0;
    for (uint32_t g = threadIdx.x; g < blockIdx.x; g += blockDim.x) part += counts[g] < cap ? counts[g] : cap;
    if (part) atomicAdd(&before, part);
    __syncthreads();
    const uint64_t mine = listed < cap ? listed : cap;
    const uint32_t* src = seg + (uint64_t)blockIdx.x * cap * nw;
    uint32_t* dst = out + before * nw;
    for (uint64_t i = threadIdx.x; i < mine * nw; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0 && blockIdx.x == n_seg - 1) desc->n_entries = before + mine;
}

hipError_t launch_compact_near(const uint32_t* seg, const uint32_t* counts, uint32_t n_seg, uint64_t cap, uint32_t nw,
                               uint32_t* out, LevelDesc* desc, hipStream_t stream) {
    if (!n_seg) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_compact_near, dim3(n_seg), dim3(256), 0, stream, seg, counts, n_seg, cap, nw, out, desc);
    return hipGetLastError();
}

// Last kernel of a chain: the counter blocks -> pinned host memory, then (system-scope release) the sequence number
// the host is spinning on (bsx_attract_api.cpp: fetch_counters).
__global__ __launch_bounds__(256) void k_publish(const uint32_t* src, uint32_t* host_dst, uint32_t words, uint32_t* host_flag, uint32_t seq,
                                                  unsigned int* ticket) {
    // (the blocks are multiples of 256 bytes: whole uint4s)
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(host_dst);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words / 4; i += gridDim.x * blockDim.x) d4[i] = s4[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        // the workgroup that finishes last raises the flag (and leaves the ticket at 0 for the next chain)
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_publish(const uint32_t* src, uint32_t* host_dst, uint32_t words, uint32_t* host_flag, uint32_t seq, unsigned int* ticket,
                          hipStream_t stream) {
    const uint32_t wgs = words / 4 <= 2048 ? 1u : words / 4 <= 16384 ? 8u : 32u;       // one per 16 KiB or so
    hipLaunchKernelGGL(k_publish, dim3(wgs), dim3(256), 0, stream, src, host_dst, words, host_flag, seq, ticket);
    return hipGetLastError();
}

#define BSX_POOL_DECL(TAG)                                                                                             \
    hipError_t launch_pool_nw##TAG(int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P); \
    hipError_t configure_pool_nw##TAG(int k, int lut_mode, size_t shmem, int* blocks_per_cu);                         \
    hipError_t launch_life_nw##TAG(int k, int lut_mode, size_t shmem, hipStream_t st, const LifetimeParams& P);
BSX_POOL_DECL(1) BSX_POOL_DECL(2) BSX_POOL_DECL(4a) BSX_POOL_DECL(4b) BSX_POOL_DECL(8a) BSX_POOL_DECL(8b) BSX_POOL_DECL(8c)

// (the wide states are built in several translation units each: odd / even numbers of predecessor slots, K = 6 at 8 words apart)
hipError_t launch_attract_pool(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const AttractParams& P) {
    switch (nw) {
        case 1: return launch_pool_nw1(k, lut_mode, grid, shmem, st, P);
        case 2: return launch_pool_nw2(k, lut_mode, grid, shmem, st, P);
        case 4: return (k & 1) ? launch_pool_nw4a(k, lut_mode, grid, shmem, st, P) : launch_pool_nw4b(k, lut_mode, grid, shmem, st, P);
        case 8: return k == 6 ? launch_pool_nw8c(k, lut_mode, grid, shmem, st, P) : (k & 1) ? launch_pool_nw8a(k, lut_mode, grid, shmem, st, P) : launch_pool_nw8b(k, lut_mode, grid, shmem, st, P);
        default: return hipErrorInvalidValue;
    }
}
hipError_t configure_attract_pool(int nw, int k, int lut_mode, size_t shmem, int* blocks_per_cu) {
    switch (nw) {
        case 1: return configure_pool_nw1(k, lut_mode, shmem, blocks_per_cu);
        case 2: return configure_pool_nw2(k, lut_mode, shmem, blocks_per_cu);
        case 4: return (k & 1) ? configure_pool_nw4a(k, lut_mode, shmem, blocks_per_cu) : configure_pool_nw4b(k, lut_mode, shmem, blocks_per_cu);
        case 8: return k == 6 ? configure_pool_nw8c(k, lut_mode, shmem, blocks_per_cu) : (k & 1) ? configure_pool_nw8a(k, lut_mode, shmem, blocks_per_cu) : configure_pool_nw8b(k, lut_mode, shmem, blocks_per_cu);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_digit_lifetimes(int nw, int k, int lut_mode, size_t shmem, hipStream_t st, const LifetimeParams& P) {
    switch (nw) {
        case 1: return launch_life_nw1(k, lut_mode, shmem, st, P);
        case 2: return launch_life_nw2(k, lut_mode, shmem, st, P);
        case 4: return (k & 1) ? launch_life_nw4a(k, lut_mode, shmem, st, P) : launch_life_nw4b(k, lut_mode, shmem, st, P);
        case 8: return k == 6 ? launch_life_nw8c(k, lut_mode, shmem, st, P) : (k & 1) ? launch_life_nw8a(k, lut_mode, shmem, st, P) : launch_life_nw8b(k, lut_mode, shmem, st, P);
        default: return hipErrorInvalidValue;
    }
}

// bytes of LDS behind the cache mirror: per-attractor tables + per-wave pool, accumulators and id slots
size_t pool_lower_extra_bytes(uint32_t nw) {      // the lower-level build: the same tables, no rings
    return (size_t)(kTagAcc + kLdsAcc) * (8 + 8 + 8 + 8 + 4) + (((size_t)(kTagAcc + kLdsAcc) * nw + 1) & ~size_t(1)) * 4 + 16 + 32 +
           (size_t)64 * nw * 4 + (size_t)kLowerFoundWords * 4 + (size_t)kPoolWaves * 2 * (kTagAcc + kLdsAcc) * 4;     // + per-wave outcome counts
}
size_t pool_extra_bytes(uint32_t nw) {
    const size_t tables = (size_t)(kTagAcc + kLdsAcc) * (8 + 8 + 8 + 8 + 4) + (((size_t)(kTagAcc + kLdsAcc) * nw + 1) & ~size_t(1)) * 4 + 16 +
                          32 +                      // + the workgroup's no-attractor / cap-failure / reference-step / executed sums
                          (size_t)64 * nw * 4;      // + the cube pass's mid-bit deposit table
    const size_t per_wave = ((size_t)kPoolCap * pool_rec_words(nw) + 128 + kPoolSlots / 4) * 4;
    return tables + (size_t)kPoolWaves * per_wave;
}

}  // namespace bsx
