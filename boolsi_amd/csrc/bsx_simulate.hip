// per-lane simulate kernel (trajectory / final state / digest sinks)
#include "bsx_kernels_common.h"

namespace bsx {

// ------------------------------------------------------------------------------------------------
// simulate: s(0..T) by plain stepping (simulate.py:97-131 == S11); sinks: trajectory, final state, digest.
template <int NW, int K, int LM>
__global__ __launch_bounds__(kBlock) void k_simulate(const SimParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
    const NetView<NW, K, LM> nv = stage_network<NW, K, LM>(P.net, smem, smem_free);
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
        // fold digest (include/bsx.h): X = xor of all s(t), Y = xor of the s(t) at times with digest_ybit(t)
        uint32_t dx[NW], dy[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) dx[w] = dy[w] = 0;
        for (uint64_t t = 0;; ++t) {
            const uint32_t ym = 0u - digest_ybit((uint32_t)t);
#pragma unroll
            for (int w = 0; w < NW; ++w) { dx[w] ^= s[w]; dy[w] ^= s[w] & ym; }
            if (out) {
#pragma unroll
                for (int w = 0; w < (NW + 1) / 2; ++w) {
                    uint64_t word = s[2 * w];
                    if (2 * w + 1 < NW) word |= (uint64_t)s[2 * w + 1] << 32;
                    if ((uint32_t)w < P.w64) out[t * P.w64 + w] = word;
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
        if (P.digests) {
            uint64_t dg = kDigestSeed;
            dg = digest_fold_words<NW>(dg, dx, P.w64);
            dg = digest_fold_words<NW>(dg, dy, P.w64);
            dg = digest_fold_words<NW>(dg, s, P.w64);
            P.digests[qi] = dg;
        }
    }
    wave_atomic_add(&P.ctr->steps_ref, (unsigned long long)steps, (int)(threadIdx.x & 63));
    wave_atomic_add(&P.ctr->steps_exec, (unsigned long long)steps, (int)(threadIdx.x & 63));
}


template <int NW, int K>
static hipError_t launch_simulate_nk(int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const SimParams& P) {
    const void* fn;
    BSX_KERNEL_FOR_MODE(k_simulate, NW, K, lut_mode, fn);
    if (!fn) return hipErrorInvalidValue;
    void* args[] = {const_cast<SimParams*>(&P)};
    return hipLaunchKernel(fn, grid, dim3(kBlock), args, shmem, st);
}
template <int NW, int K>
static hipError_t configure_simulate_nk(int lut_mode, dim3, size_t shmem, hipStream_t, const int&) {
    const void* fn;
    BSX_KERNEL_FOR_MODE(k_simulate, NW, K, lut_mode, fn);
    if (!fn) return hipErrorInvalidValue;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}

hipError_t launch_simulate(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const SimParams& P) {
    BSX_DISPATCH(launch_simulate_nk)
}
// Allow the instantiation used by a network to take `shmem` bytes of dynamic LDS (above 64 KiB this
// must be requested explicitly).
hipError_t configure_simulate(int nw, int k, int lut_mode, size_t shmem) {
    const dim3 grid(1);
    const hipStream_t st = nullptr;
    const int P = 0;
    BSX_DISPATCH(configure_simulate_nk)
}

}  // namespace bsx
