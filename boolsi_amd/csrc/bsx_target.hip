// target kernel, hit compaction, launchers
#include "bsx_kernels_common.h"

namespace bsx {

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
// Summary sink (bsx_run_target_summary): hits are counted by first-hit time in a per-workgroup LDS histogram
// that is added to the HBM one at the end; t_hit may then be null -- nothing per problem leaves the chip.
constexpr uint32_t kNotReached = 0xFFFFFFFFu;

template <int NW, int K, int LM>
__global__ __launch_bounds__(kBlock, NW <= 2 ? 4 : 2) void k_target(const TargetParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* smem_free;
    const NetView<NW, K, LM> nv = stage_network<NW, K, LM>(P.net, smem, smem_free);
    const int lane = threadIdx.x & 63;
    const bool simple_space = bsx::simple_space(P.sp);
    unsigned long long* hist = reinterpret_cast<unsigned long long*>(smem + (((uint32_t)(smem_free - smem) + 3u) & ~3u));   // LDS, P.hist_bins counters
    const uint32_t last_bin = P.hist_bins - 1;
    if (P.hist) {
        for (uint32_t i = threadIdx.x; i < P.hist_bins; i += blockDim.x) hist[i] = 0;
        __syncthreads();
    }

    uint32_t A[NW], B[NW], fm[NW], fv[NW], tm[NW], tc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) { A[w] = B[w] = 0; fm[w] = P.sp.fixmask[w]; fv[w] = P.sp.fixval[w]; tm[w] = P.tmask[w]; tc[w] = P.tcode[w]; }
    // t: absolute time; lam/power: Brent's counters (only to notice that the cycle closed)
    uint32_t phase = PH_IDLE, t = 0, tp = P.sp.tp_origin, lam = 0, power = 1;
    const uint32_t t_cap = (P.cap_rel_inf || P.max_t >= kStepLimit) ? kStepLimit : (uint32_t)P.max_t;
    uint64_t pv_digits = 0, my_p = 0;
    uint64_t steps_exec = 0, steps_ref = 0, n_hits = 0;
    uint64_t members = 1;       // problems this lane's trajectory stands for (cube pass: 2^cube_shift minus the t = 0 hits)
    uint32_t limit_hits = 0;
    bool skip0 = false;         // cube pass: the t = 0 check was done per member when the class was set up
    // work: every wave starts with a fixed share (chunk_first work items, 0 = none); what lies beyond the shares comes
    // from the shared cursor, P.chunk at a time (0 = nothing lies beyond).  Same-address atomics complete at some 15 ns
    // apiece, so the short cube passes, whose classes cost about the same everywhere, are split evenly and never touch it.
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t first_dyn = (uint64_t)gridDim.x * kWavesPerBlock * P.chunk_first;
    WaveQueue q{0, 0, P.chunk != 0 && P.count > first_dyn};
    if (P.chunk_first) {
        const uint64_t b = ((uint64_t)blockIdx.x * kWavesPerBlock + wave) * P.chunk_first;
        if (b < P.count) { q.next = b; q.end = (b + P.chunk_first < P.count) ? b + P.chunk_first : P.count; }
    }

    for (;;) {
        const uint32_t n_run = __popcll(__ballot(phase >= PH_WARM));
        const bool work_left = q.more || q.next < q.end;
        if (n_run == 0 && !work_left) break;
        if (work_left && (64u - n_run >= kServiceLanes || n_run == 0)) {
            // ---- refill idle lanes with the next problems of the wave's chunk
            if (q.next == q.end) {
                const uint64_t base = first_dyn + grab_chunk(&P.ctr->cursor, P.chunk, lane);
                if (base >= P.count) q.more = false;
                else { q.next = base; q.end = (base + P.chunk < P.count) ? base + P.chunk : P.count; }
            }
            const uint64_t avail = q.end - q.next;
            const uint64_t idle = __ballot(phase == PH_IDLE);
            if (avail && idle) {
                const uint32_t rank = rank_below(idle);
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
                    members = 1; skip0 = false;
                    if (P.cube) {                           // (tp == 0 in cube passes)
                        uint32_t d = 0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) d |= (A[w] & P.rep_mask[w]) ^ P.rep_code[w];
                        members = 1ull << P.cube_shift;
                        const uint64_t at0 = d == 0 ? members >> P.cube_t0_shift : 0ull;
                        if (at0) { atomicAdd(&hist[0], (unsigned long long)at0); n_hits += at0; }
                        members -= at0;
                        skip0 = true;
                        if (members == 0) phase = PH_IDLE;  // every member hit at t = 0
                    }
                }
                const uint64_t n_idle = (uint64_t)__popcll(idle);
                q.next += n_idle < avail ? n_idle : avail;
            }
        }       // no `continue`: a single back edge keeps the loop-carried registers in place

        // ---- check the current state (covers s(T_p), model.py:200), then one network update; idle lanes
        //      are masked off (their LUT reads would only add LDS bank conflicts)
        if (phase >= PH_WARM) {
            uint32_t nxt[NW];
            net_step<NW, K>(nv, A, fm, fv, nxt);
            if (phase == PH_BRENT) {
                const bool reached = target_hit<NW>(A, tm, tc) && !(skip0 && t == 0);
                const bool capped = !reached && t >= t_cap;
                const uint32_t lam1 = lam + 1;
                const bool closed = !reached && !capped && eq_words<NW>(nxt, B);     // every state has been checked
                const bool tele = !closed && lam1 == power;
                if (reached | capped | closed) {
                    if (P.t_hit) P.t_hit[my_p] = reached ? t : kNotReached;
                    if (P.hist && reached) atomicAdd(&hist[t < last_bin ? t : last_bin], (unsigned long long)members);
                    n_hits += reached ? members : 0ull;
                    limit_hits += (capped && t_cap == kStepLimit) ? 1u : 0u;
                    steps_exec += t;
                    steps_ref += members * t;
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
    }
    if (P.hist) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < P.hist_bins; i += blockDim.x)
            if (hist[i]) atomicAdd(&P.hist[i], hist[i]);
    }
    wave_atomic_add(&P.ctr->steps_ref, (unsigned long long)steps_ref, lane);
    wave_atomic_add(&P.ctr->steps_exec, (unsigned long long)steps_exec, lane);
    wave_atomic_add(&P.ctr->log_cursor, (unsigned long long)n_hits, lane);
    wave_atomic_add(&P.ctr->step_limit_hits, limit_hits, lane);
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
        uint32_t before = rank_below(m);
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        const uint32_t all = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        if (hit && out + before < hits_cap) { hits[out + before].offset = p; hits[out + before].t = t; }
        out += all;
        __syncthreads();
    }
}


template <int NW, int K>
static hipError_t launch_target_nk(int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const TargetParams& P) {
    const void* fn;
    BSX_KERNEL_FOR_MODE(k_target, NW, K, lut_mode, fn);
    if (!fn) return hipErrorInvalidValue;
    void* args[] = {const_cast<TargetParams*>(&P)};
    return hipLaunchKernel(fn, grid, dim3(kBlock), args, shmem, st);
}
template <int NW, int K>
static hipError_t configure_target_nk(int lut_mode, dim3, size_t shmem, hipStream_t, const int&) {
    const void* fn;
    BSX_KERNEL_FOR_MODE(k_target, NW, K, lut_mode, fn);
    if (!fn) return hipErrorInvalidValue;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}

hipError_t launch_target(int nw, int k, int lut_mode, dim3 grid, size_t shmem, hipStream_t st, const TargetParams& P) {
    BSX_DISPATCH(launch_target_nk)
}
// Allow the instantiation used by a network to take `shmem` bytes of dynamic LDS (above 64 KiB this
// must be requested explicitly).
hipError_t configure_target(int nw, int k, int lut_mode, size_t shmem) {
    const dim3 grid(1);
    const hipStream_t st = nullptr;
    const int P = 0;
    BSX_DISPATCH(configure_target_nk)
}

hipError_t launch_compact(const uint32_t* t_hit, uint64_t count, uint32_t* seg_counts, const uint64_t* seg_base,
                          HitRec* hits, uint64_t hits_cap, bool write_pass, hipStream_t st) {
    const uint32_t blocks = (uint32_t)((count + kCompactSegment - 1) / kCompactSegment);
    if (!write_pass) hipLaunchKernelGGL(k_compact_count, dim3(blocks), dim3(256), 0, st, t_hit, count, seg_counts);
    else hipLaunchKernelGGL(k_compact_write, dim3(blocks), dim3(256), 0, st, t_hit, count, seg_base, hits, hits_cap);
    return hipGetLastError();
}

}  // namespace bsx
