// Host side of the C-ABI (include/bsx.h): handle management, lowering of the network / problem-space
// tables into the device layout (gather LUT, bit-packed truth-table masks, dense perturbation
// schedule), launches, and the merge of the device attractor log.  No CPU compute path exists here:
// every bsx_run_* ends in gfx950 kernel launches (bsx_attract.hip, bsx_target.hip, bsx_simulate.hip, bsx_sliced.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
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
hipError_t configure_target(int nw, int k, int lut_mode, size_t shmem);
hipError_t configure_simulate(int nw, int k, int lut_mode, size_t shmem);
}  // namespace bsx

using namespace bsx;

namespace {
thread_local std::string g_create_error;
}  // namespace

extern "C" const char* bsx_status_string(int status) {
    switch (status) {
        case BSX_OK: return "ok";
        case BSX_ERR_INVALID: return "invalid argument";
        case BSX_ERR_NO_DEVICE: return "no gfx950 device";
        case BSX_ERR_HIP: return "HIP error";
        case BSX_ERR_UNSUPPORTED: return "unsupported network size";
        case BSX_ERR_TABLE_FULL: return "result table full";
        case BSX_ERR_STEP_LIMIT: return "internal step limit reached";
        case BSX_ERR_STATE: return "network / problem space not set";
        case BSX_ERR_COMM: return "RCCL communicator error";
        default: return "unknown status";
    }
}

extern "C" const char* bsx_last_error(bsx_handle h) {
    return h ? h->error.c_str() : g_create_error.c_str();
}

extern "C" int bsx_create(bsx_handle* out, int device) {
    if (!out) return BSX_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return BSX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        g_create_error = "device index out of range";
        return BSX_ERR_INVALID;
    }
    bsx_engine* h = new bsx_engine();
    h->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&h->prop, device)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice/hipGetDeviceProperties: ") + hipGetErrorString(e);
        delete h;
        return BSX_ERR_NO_DEVICE;
    }
    if (std::strncmp(h->prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + h->prop.gcnArchName + ", this engine is built for gfx950 only";
        delete h;
        return BSX_ERR_NO_DEVICE;
    }
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess ||
        (e = h->d_ctr.alloc(1)) != hipSuccess) {
        g_create_error = std::string("stream/event creation: ") + hipGetErrorString(e);
        delete h;
        return BSX_ERR_HIP;
    }
    const char* cc_env = std::getenv("BSX_CYCLE_CACHE");       // "0" disables the cycle-state cache (A/B runs, tests)
    h->cache_enabled = !(cc_env && cc_env[0] == '0');
    if ((e = h->d_cc_journal.alloc(kCycleJournalCap)) != hipSuccess ||
        (e = h->d_cc_claims.alloc(kCycleClaimSlots)) != hipSuccess || (e = h->d_cc_count.alloc(1)) != hipSuccess) {
        g_create_error = std::string("cycle cache allocation: ") + hipGetErrorString(e);
        delete h;
        return BSX_ERR_HIP;
    }
    *out = h;
    return BSX_OK;
}

extern "C" int bsx_destroy(bsx_handle h) {
    if (!h) return BSX_OK;
    (void)hipSetDevice(h->device);
    if (h->comm) (void)bsx_comm_destroy(h);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return BSX_OK;
}

extern "C" int bsx_device_info(bsx_handle h, char* name, uint32_t name_cap, uint32_t* compute_units,
                               uint64_t* global_mem_bytes) {
    if (!h) return BSX_ERR_INVALID;
    if (name && name_cap) std::snprintf(name, name_cap, "%s (%s)", h->prop.name, h->prop.gcnArchName);
    if (compute_units) *compute_units = (uint32_t)h->prop.multiProcessorCount;
    if (global_mem_bytes) *global_mem_bytes = (uint64_t)h->prop.totalGlobalMem;
    return BSX_OK;
}

extern "C" int bsx_synchronize(bsx_handle h) {
    if (!h) return BSX_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BSX_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int bsx_set_network(bsx_handle h, uint32_t n_nodes, const uint32_t* pred_offsets,
                               const uint32_t* pred_idx, const uint32_t* tt_word_offsets,
                               const uint64_t* tt_words) {
    if (!h) return BSX_ERR_INVALID;
    if (n_nodes == 0 || !pred_offsets || !tt_word_offsets || !tt_words)
        return fail(h, BSX_ERR_INVALID, "bsx_set_network: null table or zero nodes");
    if (n_nodes > BSX_MAX_NODES) return fail(h, BSX_ERR_UNSUPPORTED, "more than BSX_MAX_NODES nodes");
    HIPCHK(h, hipSetDevice(h->device));
    h->have_net = false;
    h->have_space = false;

    const uint32_t nw = n_nodes <= 32 ? 1 : n_nodes <= 64 ? 2 : n_nodes <= 128 ? 4 : 8;
    uint32_t k_mux = 1;
    std::vector<uint32_t> wide;
    for (uint32_t i = 0; i < n_nodes; ++i) {
        if (pred_offsets[i + 1] < pred_offsets[i]) return fail(h, BSX_ERR_INVALID, "pred_offsets not monotone");
        const uint32_t k = pred_offsets[i + 1] - pred_offsets[i];
        if (k > BSX_MAX_PREDECESSORS) return fail(h, BSX_ERR_UNSUPPORTED, "node with more than BSX_MAX_PREDECESSORS predecessors");
        if (k && !pred_idx) return fail(h, BSX_ERR_INVALID, "pred_idx is null");
        for (uint32_t j = pred_offsets[i]; j < pred_offsets[i + 1]; ++j) {
            if (pred_idx[j] >= n_nodes) return fail(h, BSX_ERR_INVALID, "predecessor index out of range");
            if (j > pred_offsets[i] && pred_idx[j] <= pred_idx[j - 1])
                return fail(h, BSX_ERR_INVALID, "predecessors must be strictly ascending");
        }
        const uint32_t need_words = k <= 6 ? 1u : (1u << (k - 6));
        if (tt_word_offsets[i + 1] - tt_word_offsets[i] != need_words)
            return fail(h, BSX_ERR_INVALID, "truth table of a node must have ceil(2^k / 64) words");
        if (k > (uint32_t)kMaxMuxK) wide.push_back(i);
        else k_mux = std::max(k_mux, k);
    }

    // LDS budget: masks (+pad) [+ LUT] [+ cycle-cache mirror].  The LUT stays in LDS while a workgroup
    // fits in 144 KiB: one entry per state byte if that fits, else (beyond 64 nodes) one entry per 4 state
    // bits -- a 16x smaller table for twice the lookups -- else the byte table is read through L2.
    const size_t mask_bytes = (((size_t)(1u << k_mux) * nw + 3) & ~size_t(3)) * 4;
    const size_t cache_stride = ((2 * nw + 2 + 3) & ~3u) * 4;
    uint32_t slots = 1;
    size_t cache_lds = kCycleCacheLdsBytes;
    if (const char* kb = std::getenv("BSX_CACHE_LDS_KB")) cache_lds = std::max<size_t>(1, (size_t)std::atoi(kb)) * 1024;   // tuning knob
    while ((size_t)slots * 2 * cache_stride <= cache_lds) slots *= 2;
    h->cache_lds_slots = slots;
    const size_t cache_bytes = (size_t)slots * cache_stride + 16 + 16      // + header + alignment
                               + lean_acc_bytes(nw);                         // + lean kernel's per-attractor tables
    const size_t entry_bytes = (size_t)k_mux * nw * 4;
    const size_t fixed_bytes = mask_bytes + cache_bytes + 64;
    int lut_mode = 0;                                                       // kLutGlobal
    if (fixed_bytes + (size_t)nw * 4 * 256 * entry_bytes <= 144 * 1024) lut_mode = 1;                  // kLutLdsByte
    else if (nw >= 4 && fixed_bytes + (size_t)nw * 8 * 16 * entry_bytes <= 144 * 1024) lut_mode = 2;   // kLutLdsNibble
    if (const char* m = std::getenv("BSX_LUT_MODE")) {                      // test knob: force a smaller-footprint mode
        const int want = std::atoi(m);
        if (want == 0 || (want == 2 && nw >= 4 && fixed_bytes + (size_t)nw * 8 * 16 * entry_bytes <= 144 * 1024)) lut_mode = want;
    }
    const uint32_t chunk_bits = lut_mode == 2 ? 4 : 8, chunk_entries = 1u << chunk_bits;
    const uint32_t n_chunks = nw * 32 / chunk_bits;     // zero entries for the chunks beyond n keep the gather branch-free
    std::vector<uint32_t> masks((size_t)(1u << k_mux) * nw, 0);
    std::vector<uint32_t> lut((size_t)n_chunks * chunk_entries * k_mux * nw, 0);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        const uint32_t k = pred_offsets[i + 1] - pred_offsets[i];
        if (k > (uint32_t)kMaxMuxK) continue;
        const uint64_t tt = tt_words[tt_word_offsets[i]];
        for (uint32_t idx = 0; idx < (1u << k_mux); ++idx)      // replicate over the unused high slots
            if ((tt >> (idx & ((1u << k) - 1))) & 1ull) masks[(size_t)idx * nw + (i >> 5)] |= 1u << (i & 31);
        for (uint32_t j = 0; j < k; ++j) {
            const uint32_t p = pred_idx[pred_offsets[i] + j];
            const uint32_t chunk = p / chunk_bits, bit = p % chunk_bits;
            for (uint32_t v = 0; v < chunk_entries; ++v)
                if ((v >> bit) & 1u)
                    lut[(((size_t)chunk * chunk_entries + v) * k_mux + j) * nw + (i >> 5)] |= 1u << (i & 31);
        }
    }
    std::vector<uint32_t> wdesc, wpreds, wtt;
    for (uint32_t i : wide) {
        const uint32_t k = pred_offsets[i + 1] - pred_offsets[i];
        wdesc.push_back(i); wdesc.push_back(k);
        wdesc.push_back((uint32_t)wpreds.size()); wdesc.push_back((uint32_t)wtt.size());
        for (uint32_t j = pred_offsets[i]; j < pred_offsets[i + 1]; ++j) wpreds.push_back(pred_idx[j]);
        for (uint32_t w = tt_word_offsets[i]; w < tt_word_offsets[i + 1]; ++w) {
            wtt.push_back((uint32_t)tt_words[w]);
            wtt.push_back((uint32_t)(tt_words[w] >> 32));
        }
    }

    HIPCHK(h, h->d_lut.upload(lut));
    HIPCHK(h, h->d_masks.upload(masks));
    HIPCHK(h, h->d_wide_desc.upload(wdesc));
    HIPCHK(h, h->d_wide_preds.upload(wpreds));
    HIPCHK(h, h->d_wide_tt.upload(wtt));

    h->h_pred_offsets.assign(pred_offsets, pred_offsets + n_nodes + 1);
    h->h_pred_idx.assign(pred_idx, pred_idx + pred_offsets[n_nodes]);
    h->h_tt0.resize(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) h->h_tt0[i] = tt_words[tt_word_offsets[i]];
    h->n_nodes = n_nodes;
    h->w64 = (n_nodes + 63) / 64;
    h->net.n_nodes = n_nodes;
    h->net.nw = nw;
    h->net.k_mux = k_mux;
    h->net.n_chunks = n_chunks;
    h->net.lut_words = (uint32_t)lut.size();
    h->net.n_wide = (uint32_t)wide.size();
    h->net.lut = h->d_lut.p;
    h->net.masks = h->d_masks.p;
    h->net.wide_desc = h->d_wide_desc.p;
    h->net.wide_preds = h->d_wide_preds.p;
    h->net.wide_tt = h->d_wide_tt.p;

    const size_t lut_bytes = lut.size() * 4;
    h->lut_mode = lut_mode;
    h->shmem = mask_bytes + (lut_mode ? lut_bytes : 0) + 64;
    h->shmem_attract = h->shmem + cache_bytes;
    HIPCHK(h, configure_attract((int)nw, (int)k_mux, h->lut_mode, h->shmem_attract));
    HIPCHK(h, configure_attract_fast((int)nw, (int)k_mux, h->lut_mode, h->shmem_attract, &h->lean_blocks_per_cu));
    h->cache_stride = cache_stride;
    {   // class-pool kernel: available when it fits next to the smallest useful mirror (64 slots)
        const size_t pool_max = h->shmem + (size_t)slots * cache_stride + 32 + pool_extra_bytes(nw);
        const size_t pool_min = h->shmem + (size_t)64 * cache_stride + 32 + pool_extra_bytes(nw);
        h->pool_ok = pool_min <= 160 * 1024 - 1024;
        int blocks = 0;
        if (h->pool_ok) HIPCHK(h, configure_attract_pool((int)nw, (int)k_mux, h->lut_mode, std::min<size_t>(pool_max, 160 * 1024 - 1024), &blocks));
    }
    if (std::getenv("BSX_DEBUG")) std::fprintf(stderr, "[bsx] network: nw %u k_mux %u lut mode %d (0 L2 bytes, 1 LDS bytes, 2 LDS nibbles) shmem %zu attract shmem %zu lean blocks/CU %d\n", nw, k_mux, (int)h->lut_mode, h->shmem, h->shmem_attract, h->lean_blocks_per_cu);
    HIPCHK(h, configure_target((int)nw, (int)k_mux, h->lut_mode, h->shmem));
    HIPCHK(h, configure_simulate((int)nw, (int)k_mux, h->lut_mode, h->shmem));
    h->have_net = true;
    return BSX_OK;
}

extern "C" int bsx_set_problem_space(bsx_handle h, const uint64_t* origin_state_words,
                                     const uint32_t* any_nodes, uint32_t n_any,
                                     const bsx_fixed* fixed, uint32_t n_fixed,
                                     const bsx_fixed_var* fixed_var, uint32_t n_fixed_var,
                                     const bsx_pert* sched, uint32_t n_sched,
                                     const bsx_pert_var* pert_var, uint32_t n_pert_var) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->have_net) return fail(h, BSX_ERR_STATE, "bsx_set_problem_space before bsx_set_network");
    if (!origin_state_words) return fail(h, BSX_ERR_INVALID, "origin state is null");
    if (n_pert_var > BSX_MAX_PERT_VARIATIONS) return fail(h, BSX_ERR_UNSUPPORTED, "more than BSX_MAX_PERT_VARIATIONS perturbation variations");
    if (n_any > h->n_nodes) return fail(h, BSX_ERR_INVALID, "more 'any' nodes than nodes");
    HIPCHK(h, hipSetDevice(h->device));
    h->have_space = false;
    const uint32_t n = h->n_nodes, nw = h->net.nw;
    DevSpace sp{};
    for (uint32_t w = 0; w < h->w64; ++w) {
        uint64_t word = origin_state_words[w];
        if (w == h->w64 - 1 && (n & 63)) word &= (1ull << (n & 63)) - 1;
        sp.origin[2 * w] = (uint32_t)word;
        if (2 * w + 1 < (uint32_t)kMaxW32) sp.origin[2 * w + 1] = (uint32_t)(word >> 32);
    }
    std::vector<uint32_t> any(n_any);
    bool identity = true;
    for (uint32_t j = 0; j < n_any; ++j) {
        if (any_nodes[j] >= n || (j && any_nodes[j] <= any_nodes[j - 1]))
            return fail(h, BSX_ERR_INVALID, "'any' nodes must be ascending node indices");
        any[j] = any_nodes[j];
        identity = identity && any_nodes[j] == j;
        sp.origin[any_nodes[j] >> 5] &= ~(1u << (any_nodes[j] & 31));   // digit decides
    }
    for (uint32_t j = 0; j < n_fixed; ++j) {
        if (fixed[j].node >= n || fixed[j].value > 1) return fail(h, BSX_ERR_INVALID, "bad fixed node entry");
        sp.fixmask[fixed[j].node >> 5] |= 1u << (fixed[j].node & 31);
        if (fixed[j].value) sp.fixval[fixed[j].node >> 5] |= 1u << (fixed[j].node & 31);
        else sp.fixval[fixed[j].node >> 5] &= ~(1u << (fixed[j].node & 31));
    }
    std::vector<uint32_t> fv, pv;
    for (uint32_t j = 0; j < n_fixed_var; ++j) {
        if (fixed_var[j].node >= n || fixed_var[j].range > 3) return fail(h, BSX_ERR_INVALID, "bad fixed-node variation");
        fv.push_back(fixed_var[j].node); fv.push_back(fixed_var[j].range);
    }
    uint32_t tp_origin = 0;
    for (uint32_t j = 0; j < n_sched; ++j) {
        if (sched[j].node >= n || sched[j].value > 1 || sched[j].t == 0) return fail(h, BSX_ERR_INVALID, "bad perturbation entry");
        tp_origin = std::max(tp_origin, sched[j].t);
    }
    for (uint32_t j = 0; j < n_pert_var; ++j) {
        if (pert_var[j].node >= n || pert_var[j].range > 3 || pert_var[j].t == 0) return fail(h, BSX_ERR_INVALID, "bad perturbation variation");
        pv.push_back(pert_var[j].t); pv.push_back(pert_var[j].node); pv.push_back(pert_var[j].range);
    }
    if ((uint64_t)(tp_origin + 1) * nw * 8 > (1ull << 30)) return fail(h, BSX_ERR_UNSUPPORTED, "perturbation schedule too long for the dense table");
    std::vector<uint32_t> set((size_t)(tp_origin + 1) * nw, 0), clr((size_t)(tp_origin + 1) * nw, 0);
    for (uint32_t j = 0; j < n_sched; ++j) {
        const size_t at = (size_t)sched[j].t * nw + (sched[j].node >> 5);
        const uint32_t m = 1u << (sched[j].node & 31);
        if (sched[j].value) { set[at] |= m; clr[at] &= ~m; } else { clr[at] |= m; set[at] &= ~m; }
    }
    {
        std::vector<std::array<uint32_t, 3>> ordered;
        for (uint32_t j = 0; j < n_sched; ++j) ordered.push_back({sched[j].t, sched[j].node, sched[j].value});
        std::stable_sort(ordered.begin(), ordered.end(), [](const auto& a, const auto& b) { return a[0] < b[0]; });
        h->h_sched.clear();
        for (const auto& e : ordered) { h->h_sched.push_back(e[0]); h->h_sched.push_back(e[1]); h->h_sched.push_back(e[2]); }
    }
    HIPCHK(h, h->d_any.upload(any));
    HIPCHK(h, h->d_fv.upload(fv));
    HIPCHK(h, h->d_pv.upload(pv));
    HIPCHK(h, h->d_set.upload(set));
    HIPCHK(h, h->d_clr.upload(clr));
    // cycles depend on the network and the origin fixed nodes: start the cache empty
    HIPCHK(h, hipMemset(h->d_cc_journal.p, 0, sizeof(CycleRecord) * kCycleJournalCap));
    HIPCHK(h, hipMemset(h->d_cc_claims.p, 0, sizeof(unsigned int) * kCycleClaimSlots));
    HIPCHK(h, hipMemset(h->d_cc_count.p, 0, sizeof(unsigned int)));
    h->fast_ok = true;
    h->h_journal.clear();
    h->journal_stale = true;
    h->fast_steps = 0;
    h->fast_calibrated = false;
    sp.n_any = n_any;
    sp.identity_any = identity ? 1 : 0;
    // deposit plan for scattered 'any' nodes: runs of consecutive nodes inside one 32-bit state word
    sp.n_runs = 0;
    if (!identity && n_any && n_any <= 64) {
        std::vector<uint32_t> plan;
        uint32_t j = 0;
        while (j < n_any) {
            uint32_t len = 1;
            while (j + len < n_any && any[j + len] == any[j] + len && ((any[j] + len) >> 5) == (any[j] >> 5)) ++len;
            plan.push_back(j | (any[j] >> 5) << 8 | (any[j] & 31u) << 16);
            plan.push_back(len >= 32 ? 0xFFFFFFFFu : (1u << len) - 1u);
            j += len;
        }
        if (plan.size() <= 2 * kMaxDepositRuns) {
            sp.n_runs = (uint32_t)(plan.size() / 2);
            std::copy(plan.begin(), plan.end(), sp.deposit);
        }
    }
    // number of variants = product of the variation radices (batching.py:10-46), saturating at 2^64
    {
        unsigned __int128 v = 1;
        auto times = [&](uint32_t range) { if (v <= UINT64_MAX) v *= (range == BSX_RANGE_MAYBE_TRUE_OR_FALSE ? 3u : 2u); };
        for (uint32_t j = 0; j < n_fixed_var; ++j) times(fixed_var[j].range);
        for (uint32_t j = 0; j < n_pert_var; ++j) times(pert_var[j].range);
        h->variant_count_saturated = v > UINT64_MAX;
        h->variant_count = h->variant_count_saturated ? UINT64_MAX : (uint64_t)v;
    }
    h->tp_max = tp_origin;
    for (uint32_t j = 0; j < n_pert_var; ++j) h->tp_max = std::max(h->tp_max, pert_var[j].t);
    sp.n_fv = n_fixed_var;
    sp.n_pv = n_pert_var;
    sp.tp_origin = tp_origin;
    sp.any_nodes = h->d_any.p;
    sp.fv = h->d_fv.p;
    sp.pv = h->d_pv.p;
    sp.sched_set = h->d_set.p;
    sp.sched_clr = h->d_clr.p;
    h->sp = sp;
    h->have_space = true;
    return BSX_OK;
}

// ------------------------------------------------------------------------------------------------
namespace {

struct Launch {
    dim3 grid;
    uint32_t chunk;
};

Launch plan_persistent(const bsx_engine* h, uint64_t count, size_t shmem) {
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
int check_range(bsx_handle h, const bsx_index* first, uint64_t count) {
    if (!first) return fail(h, BSX_ERR_INVALID, "first index is null");
    const uint32_t n_any = h->sp.n_any;
    for (uint32_t b = n_any; b < 64 * BSX_MAX_WORDS; ++b)
        if ((first->init_digits[b >> 6] >> (b & 63)) & 1ull)
            return fail(h, BSX_ERR_INVALID, "init_digits has bits at or above n_any");
    if (!h->variant_count_saturated && first->variant >= h->variant_count)
        return fail(h, BSX_ERR_INVALID, "variant number outside the problem space");
    if (count == 0) return BSX_OK;
    // last = init_digits + (count - 1), up to 257 bits; what lies above bit n_any carries into the variant
    uint64_t sum[5];
    unsigned __int128 carry = count - 1;
    for (int w = 0; w < 4; ++w) {
        carry += first->init_digits[w];
        sum[w] = (uint64_t)carry;
        carry >>= 64;
    }
    sum[4] = (uint64_t)carry;
    uint64_t over = 0;                                      // (sum >> n_any); fits 64 bits since count does
    for (uint32_t b = n_any; b < 320 && b < n_any + 64; ++b)
        over |= ((sum[b >> 6] >> (b & 63)) & 1ull) << (b - n_any);
    const uint64_t last_variant = first->variant + over;
    if (last_variant < over || (!h->variant_count_saturated && last_variant >= h->variant_count))
        return fail(h, BSX_ERR_INVALID, "first + count runs past the end of the problem space");
    return BSX_OK;
}

int check_max_t(bsx_handle h, uint64_t max_t) {
    if (max_t != BSX_T_INF && max_t < h->tp_max)
        return fail(h, BSX_ERR_INVALID, "max_t is below the last perturbation time (origin schedule or a variation)");
    return BSX_OK;
}

void set_first(DevSpace& sp, const bsx_index* first) {
    for (int w = 0; w < 4; ++w) sp.first_digits[w] = first->init_digits[w];
    sp.first_variant = first->variant;
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct KeyLess {
    bool operator()(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) const { return a < b; }
};

}  // namespace

namespace {

constexpr uint64_t kFastMinProblems = 8192;     // below this the general kernel alone is used
constexpr uint64_t kDiscoverySample = 65536;    // problems (sampled over the range) run through the detector when nothing is cached yet
constexpr uint64_t kLeanTile = 1ull << 28;      // problems per lean-kernel launch (straggler list: 4 B each)
constexpr uint32_t kFastSteps = 48;             // FAST phase length (steps without a cached cycle state), first guess
constexpr uint32_t kFastStepsMax = 3072;
constexpr uint64_t kProbeTile = 1ull << 22;     // lean tiles while the FAST length is being calibrated

using MergedTable = std::map<std::vector<uint32_t>, bsx_attr_rec, KeyLess>;

struct AttractRun {
    Counters ctr{};
    float ms = 0.f;
};

// One k_attract launch (general or fast) + merge of its log into `merged` unless `discard_log`.
enum PassKind { kPassGeneral = 0, kPassLean = 1, kPassPool = 2 };

// LDS mirror size for the lean / pool kernels: they fill the mirror once from the journal, so it only has
// to hold what the journal holds (4 slots per state keeps probe chains short); a smaller mirror leaves
// the LDS to more workgroups.  The general kernel inserts while it runs and keeps the full size.
int lean_mirror_slots(bsx_handle h, uint32_t* slots_out) {
    if (!h->journal_stale) { *slots_out = h->mirror_slots; return BSX_OK; }
    unsigned int known = 0;
    HIPCHK(h, hipMemcpy(&known, h->d_cc_count.p, sizeof(known), hipMemcpyDeviceToHost));
    known = std::min<unsigned int>(known, kCycleJournalCap);
    h->h_journal.resize(known);
    if (known) HIPCHK(h, hipMemcpy(h->h_journal.data(), h->d_cc_journal.p, known * sizeof(CycleRecord), hipMemcpyDeviceToHost));
    uint64_t states = 0;
    uint32_t taken = 0;
    for (const CycleRecord& r : h->h_journal) {
        if (taken >= (uint32_t)kTagAcc + kLdsAcc) break;
        if (!r.ready || r.length == 0 || r.length > kCycleCacheMaxLen) continue;
        states += r.length;
        ++taken;
    }
    uint32_t slots = 64;
    while (slots < 4 * states && slots < h->cache_lds_slots) slots *= 2;
    h->mirror_slots = *slots_out = std::min(slots, h->cache_lds_slots);
    h->journal_stale = false;
    return BSX_OK;
}

int launch_attract_pass(bsx_handle h, AttractParams& P, int kind, DevBuf<LogRec>& d_log, MergedTable* merged,
                        AttractRun& run) {
    const bool fast = kind != kPassGeneral;
    if (!fast) h->journal_stale = true;         // the detector may publish attractors
    size_t shmem = h->shmem_attract;
    if (fast) {
        uint32_t slots = h->cache_lds_slots;
        if (int rc = lean_mirror_slots(h, &slots)) return rc;
        P.cc.lds_slots = slots;
        shmem = h->shmem + (size_t)slots * h->cache_stride + 32 + (kind == kPassPool ? pool_extra_bytes(h->net.nw) : lean_acc_bytes(h->net.nw));
    }
    const Launch L = plan_persistent(h, P.count, shmem);
    P.chunk = L.chunk;
    if (const char* c = std::getenv("BSX_CHUNK")) P.chunk = (uint32_t)std::max(64, std::atoi(c));     // tuning knob
    const uint64_t waves = (uint64_t)L.grid.x * kWavesPerBlock;
    const uint64_t log_cap = waves * kTableSlots + (1u << 16);
    if (d_log.n < log_cap) HIPCHK(h, d_log.alloc(log_cap));
    P.log = d_log.p;
    P.log_cap = log_cap;
    HIPCHK(h, hipMemsetAsync(h->d_ctr.p, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if (kind == kPassPool) HIPCHK(h, launch_attract_pool((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    else if (kind == kPassLean) HIPCHK(h, launch_attract_fast((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    else HIPCHK(h, launch_attract((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipMemcpyAsync(&run.ctr, h->d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipEventElapsedTime(&run.ms, h->ev0, h->ev1));
    if (std::getenv("BSX_DEBUG"))
        std::fprintf(stderr, "[bsx] %s pass: %llu problems, %llu lane-steps, %llu stragglers, %.3f ms (BSX_DIAG build: %llu wave iterations, %llu service rounds)\n",
                     kind == kPassPool ? "pool" : fast ? "lean" : "general", (unsigned long long)P.count, (unsigned long long)run.ctr.steps_exec,
                     (unsigned long long)run.ctr.n_stragglers, run.ms, (unsigned long long)run.ctr.wave_iters,
                     (unsigned long long)run.ctr.service_rounds);
    if (run.ctr.log_overflow) return fail(h, BSX_ERR_TABLE_FULL, "device attractor log overflowed");
    if (!merged) return BSX_OK;
    const uint64_t n_log = run.ctr.log_cursor;
    std::vector<LogRec> log(n_log);
    if (n_log) HIPCHK(h, hipMemcpy(log.data(), d_log.p, n_log * sizeof(LogRec), hipMemcpyDeviceToHost));
    // merge by key (attract.py:405-455 write_aggregated_attractors_to_db, exact integers)
    const uint32_t nw = h->net.nw;
    for (const LogRec& r : log) {
        std::vector<uint32_t> key(r.key, r.key + nw);
        auto it = merged->find(key);
        if (it == merged->end()) {
            bsx_attr_rec a{};
            for (uint32_t w = 0; w < nw; ++w) a.key[w >> 1] |= (uint64_t)r.key[w] << (32 * (w & 1));
            a.length = r.length;
            it = merged->emplace(key, a).first;
        }
        bsx_attr_rec& a = it->second;
        a.count += r.count;
        a.sum_l += r.sum_l;
        const uint64_t lo = a.sum_l2_lo + r.sum_l2;
        if (lo < a.sum_l2_lo) ++a.sum_l2_hi;
        a.sum_l2_lo = lo;
    }
    return BSX_OK;
}

// first + delta for spaces whose initial-state digits fit one word (the fast path's precondition)
void advance_first(DevSpace& sp, const bsx_index* first, uint64_t delta) {
    for (int w = 0; w < 4; ++w) sp.first_digits[w] = first->init_digits[w];
    sp.first_digits[0] += delta;
    sp.first_variant = first->variant;
}

}  // namespace

extern "C" int bsx_run_attract(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                               uint64_t max_len, bsx_attr_rec* table, uint32_t cap, uint32_t* n_out,
                               uint64_t* n_no_attractor, bsx_problem_rec* per_problem, bsx_stats* stats) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->have_net || !h->have_space) return fail(h, BSX_ERR_STATE, "network / problem space not set");
    if (!table || !n_out) return fail(h, BSX_ERR_INVALID, "table / n_out is null");
    if (int rc = check_range(h, first, count)) return rc;
    if (int rc = check_max_t(h, max_t)) return rc;
    const double t_begin = now_ms();
    HIPCHK(h, hipSetDevice(h->device));
    *n_out = 0;
    if (n_no_attractor) *n_no_attractor = 0;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (count == 0) return BSX_OK;
    if (count > (1ull << 32)) return fail(h, BSX_ERR_INVALID, "at most 2^32 problems per call");

    DevBuf<LogRec>& d_log = h->d_log;
    DevBuf<ProblemRec32> d_pp;
    if (per_problem) HIPCHK(h, d_pp.alloc(count));

    AttractParams P{};
    P.net = h->net;
    P.sp = h->sp;
    set_first(P.sp, first);
    P.count = count;
    P.cap_rel_inf = max_t == BSX_T_INF ? 1 : 0;
    P.max_t = max_t;
    P.max_len = max_len;
    P.ctr = h->d_ctr.p;
    P.per_problem = per_problem ? d_pp.p : nullptr;
    P.cc.journal = h->d_cc_journal.p;
    P.cc.journal_count = h->d_cc_count.p;
    P.cc.claims = h->d_cc_claims.p;
    // cycles depend on the fixed nodes: with fixed-node variations they differ per problem
    P.cc.enabled = (h->cache_enabled && h->sp.n_fv == 0) ? 1u : 0u;
    P.cc.lds_slots = h->cache_lds_slots;
    if (!h->fast_steps) h->fast_steps = kFastSteps;
    P.fast_steps = h->fast_steps;
    if (const char* sl = std::getenv("BSX_SERVICE_LANES")) P.pad = (uint32_t)std::atoi(sl);

    MergedTable merged;
    uint64_t n_none = 0, steps_ref = 0, steps_exec = 0;
    double kernel_ms = 0.0;
    uint32_t launches = 0, limit_hits = 0;
    auto account = [&](const AttractRun& r) {
        n_none += r.ctr.n_none; steps_ref += r.ctr.steps_ref; steps_exec += r.ctr.steps_exec;
        kernel_ms += r.ms; ++launches; limit_hits += r.ctr.step_limit_hits;
    };

    // Fast path: simple enumeration (no variations, 'any' nodes = nodes 0..a-1, a <= 64), no warm-up,
    // cycle cache on.  [discovery prefix with the detector] -> lean kernel -> stragglers.
    // (a short uniform warm-up is fine; its length enters the lean kernel's 32-bit sums of trajectory_l^2)
    const bool simple = h->sp.n_any <= 64 && (h->sp.identity_any || h->sp.n_runs) && !h->sp.n_fv && !h->sp.n_pv && h->sp.tp_origin <= 200;
    bool use_fast = P.cc.enabled && simple && h->fast_ok && count >= kFastMinProblems;
    if (const char* e = std::getenv("BSX_LEAN")) use_fast = use_fast && std::atoi(e) != 0;      // tuning / test knob
    if (std::getenv("BSX_DEBUG")) std::fprintf(stderr, "[bsx] attract: count %llu cache %u identity %u n_any %u n_fv %u n_pv %u tp %u fast_ok %d -> lean path %d\n", (unsigned long long)count, P.cc.enabled, h->sp.identity_any, h->sp.n_any, h->sp.n_fv, h->sp.n_pv, h->sp.tp_origin, (int)h->fast_ok, (int)use_fast);
    uint64_t done = 0;
    if (use_fast) {
        unsigned int known = 0;
        HIPCHK(h, hipMemcpy(&known, h->d_cc_count.p, sizeof(known), hipMemcpyDeviceToHost));
        if (known == 0) {
            // Nothing cached yet: run the detector over a pseudo-random sample of the range (all digit
            // positions vary), only to fill the cycle cache; its results are discarded and every problem
            // is counted exactly once below.
            const uint64_t m = std::min<uint64_t>(count, kDiscoverySample);
            std::vector<uint32_t> sample(m);
            for (uint64_t i = 0; i < m; ++i) {
                uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;        // splitmix64 finaliser
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                sample[i] = (uint32_t)((z ^ (z >> 31)) % count);
            }
            DevBuf<uint32_t> d_sample;
            HIPCHK(h, d_sample.upload(sample));
            AttractParams Q = P;
            Q.count = m;
            Q.offsets = d_sample.p;
            Q.per_problem = nullptr;
            AttractRun r;
            if (int rc = launch_attract_pass(h, Q, kPassGeneral, d_log, nullptr, r)) return rc;
            kernel_ms += r.ms; ++launches; steps_exec += r.ctr.steps_exec;
            HIPCHK(h, hipMemcpy(&known, h->d_cc_count.p, sizeof(known), hipMemcpyDeviceToHost));
            if (known == 0) use_fast = false;           // nothing cacheable was found
        }
    }
    // Lean kernel over tiles; what it cannot resolve (attractors not cached yet, long transients) goes
    // through the detector right after each tile, which also teaches the cache for the next tile.
    // The first tiles of a space are small probes: if most of their stragglers did end on a cached
    // cycle state (just later than the FAST length), the FAST length is quadrupled for what follows.
    // BSX_MERGE: 2 (default) class-pool kernel, 1 lean kernel with the in-lane sibling merge, 0 lean kernel
    // without merging (A/B runs, tests)
    const char* merge_env = std::getenv("BSX_MERGE");
    int merge_mode = merge_env ? std::atoi(merge_env) : 2;
    if (merge_mode == 2 && !h->pool_ok) merge_mode = 1;
    const bool merge_lanes = merge_mode != 0;
    while (use_fast && h->fast_ok && done < count) {
        const uint64_t tile = std::min<uint64_t>(count - done, h->fast_calibrated ? kLeanTile : kProbeTile);
        // straggler list: one word per problem, or up to three per class (base + 64-bit member mask) from the
        // pool kernel -- probe tiles get room for every problem as a class of its own, big tiles for a third
        // (more stragglers than that and the lean path is the wrong tool anyway)
        const uint64_t strag_cap = h->fast_calibrated ? tile : 3 * tile;
        DevBuf<uint32_t>& d_strag = h->d_strag;
        if (d_strag.n < strag_cap) HIPCHK(h, d_strag.alloc(strag_cap));
        AttractParams Q = P;
        advance_first(Q.sp, first, done);
        Q.count = tile;
        Q.fast_steps = h->fast_steps;
        // the pool kernel first runs with member counts (classes of different groups merge too); that only works
        // while nothing has to go back to the general kernel, so a tile that raises the abort flag is repeated
        // with member masks.  Per-problem records need the masks from the start.
        bool counting = merge_mode == 2 && !per_problem && (h->fast_calibrated || std::getenv("BSX_FORCE_COUNTING"));     // (knob: tests)
        Q.merge = counting ? 2u : (merge_lanes ? 1u : 0u);
        Q.per_problem = per_problem ? d_pp.p + done : nullptr;
        Q.stragglers = d_strag.p;
        Q.stragglers_cap = strag_cap;
        AttractRun r;
        MergedTable tile_table;         // folded into `merged` only if the pass is accepted
        if (int rc = launch_attract_pass(h, Q, merge_mode == 2 ? kPassPool : kPassLean, d_log, &tile_table, r)) return rc;
        if (counting && (r.ctr.straggler_overflow & 2u)) {
            kernel_ms += r.ms; ++launches;              // dropped pass
            counting = false;
            Q.merge = 1u;
            tile_table.clear();
            r = AttractRun{};
            if (int rc = launch_attract_pass(h, Q, kPassPool, d_log, &tile_table, r)) return rc;
        }
        if (r.ctr.straggler_overflow) {
            // more (group, mask) pairs than the list holds: the cache does not cover this space.  Drop the
            // pass and give the rest of the range to the detector.
            kernel_ms += r.ms; ++launches;
            h->fast_ok = false;
            break;
        }
        for (auto& kv : tile_table) {
            auto it = merged.find(kv.first);
            if (it == merged.end()) { merged.emplace(kv.first, kv.second); continue; }
            bsx_attr_rec& a = it->second;
            a.count += kv.second.count;
            a.sum_l += kv.second.sum_l;
            const uint64_t lo = a.sum_l2_lo + kv.second.sum_l2_lo;
            a.sum_l2_hi += kv.second.sum_l2_hi + (lo < a.sum_l2_lo ? 1 : 0);
            a.sum_l2_lo = lo;
        }
        account(r);
        uint64_t late = 0;
        if (r.ctr.n_stragglers) {
            uint64_t n_list = r.ctr.n_stragglers;
            if (merge_lanes) {
                // (group base, member mask words) records -> problem offsets, ascending
                const size_t rec = merge_mode == 2 ? 3 : 2;     // the pool kernel's groups have 64 members
                std::vector<uint32_t> pairs(rec * r.ctr.straggler_classes);
                HIPCHK(h, hipMemcpy(pairs.data(), d_strag.p, pairs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                std::vector<uint32_t> offs;
                offs.reserve(n_list);
                for (size_t c = 0; c + rec <= pairs.size(); c += rec)
                    for (size_t wd = 1; wd < rec; ++wd)
                        for (uint32_t left = pairs[c + wd]; left; left &= left - 1)
                            offs.push_back(pairs[c] + (uint32_t)(32 * (wd - 1)) + (uint32_t)__builtin_ctz(left));
                std::sort(offs.begin(), offs.end());
                n_list = offs.size();
                HIPCHK(h, hipMemcpy(d_strag.p, offs.data(), offs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            }
            AttractParams S = Q;
            S.count = n_list;
            S.offsets = d_strag.p;
            S.stragglers = nullptr;
            S.merge = 0;
            AttractRun rs;
            if (int rc = launch_attract_pass(h, S, kPassGeneral, d_log, &merged, rs)) return rc;
            account(rs);
            late = rs.ctr.n_cache_resolved;
        }
        done += tile;
        const bool many = r.ctr.n_stragglers > tile / 32;
        if (std::getenv("BSX_DEBUG")) std::fprintf(stderr, "[bsx] tile %llu: %llu stragglers, %llu of them ended on a cached cycle state; FAST length %u\n", (unsigned long long)tile, (unsigned long long)r.ctr.n_stragglers, (unsigned long long)late, h->fast_steps);
        if (many && 2 * late >= r.ctr.n_stragglers && h->fast_steps < kFastStepsMax) {
            h->fast_steps = std::min(kFastStepsMax, h->fast_steps * 4);     // long transients: give FAST more steps
        } else {
            if (tile >= kFastMinProblems) h->fast_calibrated = true;
            if (r.ctr.n_stragglers > tile / 2) h->fast_ok = false;          // the cache does not cover this space
        }
    }
    if (done < count) {
        AttractParams Q = P;
        if (done) advance_first(Q.sp, first, done);
        Q.count = count - done;
        Q.per_problem = per_problem ? d_pp.p + done : nullptr;
        AttractRun r;
        if (int rc = launch_attract_pass(h, Q, kPassGeneral, d_log, &merged, r)) return rc;
        account(r);
    }

    if (merged.size() > cap) return fail(h, BSX_ERR_TABLE_FULL, "more distinct attractors than the caller's table capacity");
    uint32_t i = 0;
    for (auto& kv : merged) table[i++] = kv.second;
    *n_out = i;
    if (n_no_attractor) *n_no_attractor = n_none;

    if (per_problem) {
        const uint32_t nw = h->net.nw;
        std::vector<ProblemRec32> pp(count);
        HIPCHK(h, hipMemcpy(pp.data(), d_pp.p, count * sizeof(ProblemRec32), hipMemcpyDeviceToHost));
        for (uint64_t p = 0; p < count; ++p) {
            bsx_problem_rec o{};
            for (uint32_t w = 0; w < nw; ++w) o.key[w >> 1] |= (uint64_t)pp[p].key[w] << (32 * (w & 1));
            o.length = pp[p].length; o.trajectory_l = pp[p].trajectory_l; o.found = pp[p].found;
            per_problem[p] = o;
        }
    }
    if (stats) {
        stats->problems = count;
        stats->state_steps = steps_ref;
        stats->executed_steps = steps_exec;
        stats->kernel_ms = kernel_ms;
        stats->kernel_launches = launches;
        stats->total_ms = now_ms() - t_begin;
    }
    if (limit_hits) return fail(h, BSX_ERR_STEP_LIMIT, "a trajectory reached the internal step limit without closing its cycle");
    return BSX_OK;
}

extern "C" int bsx_run_target(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                              const uint64_t* mask_words, const uint64_t* code_words, bsx_hit* hits,
                              uint64_t cap, uint64_t* n_hits, bsx_stats* stats) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->have_net || !h->have_space) return fail(h, BSX_ERR_STATE, "network / problem space not set");
    if (!mask_words || !code_words || !n_hits || (cap && !hits)) return fail(h, BSX_ERR_INVALID, "null argument");
    if (int rc = check_range(h, first, count)) return rc;
    if (int rc = check_max_t(h, max_t)) return rc;
    const double t_begin = now_ms();
    HIPCHK(h, hipSetDevice(h->device));
    *n_hits = 0;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (count == 0) return BSX_OK;

    if (count > (1ull << 32)) return fail(h, BSX_ERR_INVALID, "at most 2^32 problems per call");
    const Launch L = plan_persistent(h, count, h->shmem);
    DevBuf<uint32_t> d_thit;
    HIPCHK(h, d_thit.alloc(count));
    TargetParams P{};
    P.net = h->net;
    P.sp = h->sp;
    set_first(P.sp, first);
    P.count = count;
    P.chunk = L.chunk;
    P.cap_rel_inf = max_t == BSX_T_INF ? 1 : 0;
    P.max_t = max_t;
    for (uint32_t w = 0; w < h->w64; ++w) {
        P.tmask[2 * w] = (uint32_t)mask_words[w]; P.tcode[2 * w] = (uint32_t)code_words[w];
        if (2 * w + 1 < (uint32_t)kMaxW32) { P.tmask[2 * w + 1] = (uint32_t)(mask_words[w] >> 32); P.tcode[2 * w + 1] = (uint32_t)(code_words[w] >> 32); }
    }
    P.ctr = h->d_ctr.p;
    P.t_hit = d_thit.p;

    HIPCHK(h, hipMemsetAsync(h->d_ctr.p, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    HIPCHK(h, launch_target((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, h->shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    Counters ctr{};
    HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    const uint64_t total_hits = ctr.log_cursor;
    if (total_hits > cap) return fail(h, BSX_ERR_TABLE_FULL, "more hits than the caller's capacity");
    if (total_hits) {
        // ordered compaction of t_hit[] -> hit list (index order)
        const uint32_t segs = (uint32_t)((count + 4095) / 4096);
        DevBuf<uint32_t> d_cnt;
        DevBuf<uint64_t> d_base;
        DevBuf<HitRec> d_hits;
        HIPCHK(h, d_cnt.alloc(segs));
        HIPCHK(h, d_base.alloc(segs));
        HIPCHK(h, d_hits.alloc(total_hits));
        HIPCHK(h, launch_compact(d_thit.p, count, d_cnt.p, nullptr, nullptr, 0, false, h->stream));
        std::vector<uint32_t> cnt(segs);
        HIPCHK(h, hipMemcpyAsync(cnt.data(), d_cnt.p, segs * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<uint64_t> base(segs);
        uint64_t run = 0;
        for (uint32_t i = 0; i < segs; ++i) { base[i] = run; run += cnt[i]; }
        if (run != total_hits) return fail(h, BSX_ERR_HIP, "hit compaction count mismatch");
        HIPCHK(h, hipMemcpyAsync(d_base.p, base.data(), segs * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, launch_compact(d_thit.p, count, nullptr, d_base.p, d_hits.p, total_hits, true, h->stream));
        static_assert(sizeof(HitRec) == sizeof(bsx_hit), "hit layout");
        HIPCHK(h, hipMemcpyAsync(hits, d_hits.p, total_hits * sizeof(HitRec), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    *n_hits = total_hits;
    if (stats) {
        stats->problems = count;
        stats->state_steps = ctr.steps_ref;
        stats->executed_steps = ctr.steps_exec;
        stats->kernel_ms = ms;
        stats->kernel_launches = 1;
        stats->total_ms = now_ms() - t_begin;
    }
    if (ctr.step_limit_hits) return fail(h, BSX_ERR_STEP_LIMIT, "a trajectory reached the internal step limit");
    return BSX_OK;
}

static int run_sim_common(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                          const uint64_t* offsets, const uint64_t* t_len, const uint64_t* out_offsets,
                          uint64_t traj_words, uint64_t* trajectories, uint64_t* final_states,
                          uint64_t* digests, bsx_stats* stats) {
    const double t_begin = now_ms();
    HIPCHK(h, hipSetDevice(h->device));
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (count == 0) return BSX_OK;
    if (max_t >= kStepLimit) return fail(h, BSX_ERR_UNSUPPORTED, "simulation length above the engine's step limit");
    const uint32_t W = h->w64;
    DevBuf<uint64_t> d_traj, d_final, d_dig, d_off, d_tlen, d_ooff;
    if (trajectories) HIPCHK(h, d_traj.alloc(traj_words));
    if (final_states) HIPCHK(h, d_final.alloc(count * W));
    if (digests) HIPCHK(h, d_dig.alloc(count));
    if (offsets) { HIPCHK(h, d_off.alloc(count)); HIPCHK(h, hipMemcpy(d_off.p, offsets, count * 8, hipMemcpyHostToDevice)); }
    if (t_len) { HIPCHK(h, d_tlen.alloc(count)); HIPCHK(h, hipMemcpy(d_tlen.p, t_len, count * 8, hipMemcpyHostToDevice)); }
    if (out_offsets) { HIPCHK(h, d_ooff.alloc(count)); HIPCHK(h, hipMemcpy(d_ooff.p, out_offsets, count * 8, hipMemcpyHostToDevice)); }

    SimParams P{};
    P.net = h->net;
    P.sp = h->sp;
    set_first(P.sp, first);
    P.count = count;
    P.max_t = max_t;
    P.w64 = W;
    P.offsets = offsets ? d_off.p : nullptr;
    P.t_len = t_len ? d_tlen.p : nullptr;
    P.out_offsets = out_offsets ? d_ooff.p : nullptr;
    P.traj = trajectories ? d_traj.p : nullptr;
    P.final_states = final_states ? d_final.p : nullptr;
    P.digests = digests ? d_dig.p : nullptr;
    P.ctr = h->d_ctr.p;

    const uint32_t cus = (uint32_t)h->prop.multiProcessorCount;
    const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus * 4, (count + kBlock - 1) / kBlock));
    HIPCHK(h, hipMemsetAsync(h->d_ctr.p, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    HIPCHK(h, launch_simulate((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, dim3((uint32_t)blocks), h->shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    Counters ctr{};
    HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (trajectories) HIPCHK(h, hipMemcpy(trajectories, d_traj.p, traj_words * 8, hipMemcpyDeviceToHost));
    if (final_states) HIPCHK(h, hipMemcpy(final_states, d_final.p, count * W * 8, hipMemcpyDeviceToHost));
    if (digests) HIPCHK(h, hipMemcpy(digests, d_dig.p, count * 8, hipMemcpyDeviceToHost));
    if (stats) {
        stats->problems = count;
        stats->state_steps = ctr.steps_ref;
        stats->executed_steps = ctr.steps_exec;
        stats->kernel_ms = ms;
        stats->kernel_launches = 1;
        stats->total_ms = now_ms() - t_begin;
    }
    return BSX_OK;
}

// Final states of a fixed-length run through the bit-sliced kernel (no variations, no wide rules).
static int run_sim_sliced(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                          uint64_t* final_states, bsx_stats* stats) {
    const double t_begin = now_ms();
    HIPCHK(h, hipSetDevice(h->device));
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const uint32_t n = h->n_nodes, K = h->net.k_mux, W = h->w64;
    const uint32_t rows = (n + 15) & ~15u;     // node batch (4) x waves per workgroup (4)
    std::vector<uint32_t> desc((size_t)rows * 8, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t k = h->h_pred_offsets[i + 1] - h->h_pred_offsets[i];
        for (uint32_t j = 0; j < k; ++j) desc[(size_t)i * 8 + j] = h->h_pred_idx[h->h_pred_offsets[i] + j];
        uint64_t tt = 0;
        const bool fixed = (h->sp.fixmask[i >> 5] >> (i & 31)) & 1u;
        if (fixed) tt = ((h->sp.fixval[i >> 5] >> (i & 31)) & 1u) ? ~0ull : 0ull;     // model.py:45-47
        else
            for (uint32_t idx = 0; idx < (1u << K); ++idx)
                if ((h->h_tt0[i] >> (idx & ((1u << k) - 1))) & 1ull) tt |= 1ull << idx;
        desc[(size_t)i * 8 + 6] = (uint32_t)tt;
        desc[(size_t)i * 8 + 7] = (uint32_t)(tt >> 32);
    }
    DevBuf<uint32_t> d_desc, d_sched;
    DevBuf<uint64_t> d_final;
    HIPCHK(h, d_desc.upload(desc));
    HIPCHK(h, d_sched.upload(h->h_sched));
    HIPCHK(h, d_final.alloc(count * W));

    SlicedParams P{};
    P.sp = h->sp;
    set_first(P.sp, first);
    P.n_nodes = n;
    P.n_rows = rows;
    P.n_sched = (uint32_t)(h->h_sched.size() / 3);
    P.w64 = W;
    P.desc = d_desc.p;
    P.sched = d_sched.p;
    P.count = count;
    P.max_t = max_t;
    P.final_states = d_final.p;
    P.ctr = h->d_ctr.p;

    // K <= 3 and n <= 128: second-generation kernel (8-byte rows, constants in registers); BSX_SLICED=1 keeps the first
    const char* sl_env = std::getenv("BSX_SLICED");
    const bool gen2 = K <= 3 && rows <= 128 && !(sl_env && sl_env[0] == '1');
    const size_t shmem = gen2 ? (size_t)rows * 1024 + (4096 + 64) * 4 : (size_t)rows * (8 + 128) * 4;
    const uint64_t groups = gen2 ? (count + 4095) / 4096 : (count + 2047) / 2048;
    const uint64_t per_cu = gen2 ? 1 : std::max<size_t>(1, (160 * 1024) / shmem);
    const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(groups, (uint64_t)h->prop.multiProcessorCount * per_cu));
    HIPCHK(h, hipMemsetAsync(h->d_ctr.p, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if (gen2) HIPCHK(h, launch_simulate_sliced64((int)h->net.nw, (int)K, dim3((uint32_t)blocks), shmem, h->stream, P));
    else HIPCHK(h, launch_simulate_sliced((int)h->net.nw, (int)K, dim3((uint32_t)blocks), shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    Counters ctr{};
    HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    HIPCHK(h, hipMemcpy(final_states, d_final.p, count * W * 8, hipMemcpyDeviceToHost));
    if (stats) {
        stats->problems = count;
        stats->state_steps = ctr.steps_ref;
        stats->executed_steps = ctr.steps_exec;
        stats->kernel_ms = ms;
        stats->kernel_launches = 1;
        stats->total_ms = now_ms() - t_begin;
    }
    return BSX_OK;
}

extern "C" int bsx_run_simulate(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                                uint64_t* trajectories, uint64_t* final_states, uint64_t* digests,
                                bsx_stats* stats) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->have_net || !h->have_space) return fail(h, BSX_ERR_STATE, "network / problem space not set");
    if (int rc = check_range(h, first, count)) return rc;
    if (int rc = check_max_t(h, max_t)) return rc;
    // Long fixed-length runs that only want final states go through the bit-sliced kernel
    // (BSX_SLICED=0 forces the per-lane kernel, for A/B runs and tests).
    const char* sl_env = std::getenv("BSX_SLICED");
    const bool sliced_ok = !(sl_env && sl_env[0] == '0') && final_states && !trajectories && !digests &&
                           !h->sp.n_fv && !h->sp.n_pv && !h->net.n_wide && max_t >= 64 && max_t < kStepLimit &&
                           count >= 2048 && (size_t)((h->n_nodes + 15) & ~15u) * 136 * 4 <= 160 * 1024;
    if (sliced_ok && count) return run_sim_sliced(h, first, count, max_t, final_states, stats);
    const uint64_t words = trajectories ? count * (max_t + 1) * h->w64 : 0;
    return run_sim_common(h, first, count, max_t, nullptr, nullptr, nullptr, words, trajectories, final_states,
                          digests, stats);
}

extern "C" int bsx_run_trajectories(bsx_handle h, const bsx_index* first, const uint64_t* offsets,
                                    const uint64_t* t_len, uint64_t n, uint64_t* out,
                                    const uint64_t* out_offsets, bsx_stats* stats) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->have_net || !h->have_space) return fail(h, BSX_ERR_STATE, "network / problem space not set");
    if (n && (!offsets || !t_len || !out || !out_offsets)) return fail(h, BSX_ERR_INVALID, "null argument");
    uint64_t words = 0, tmax = 0, off_max = 0;
    for (uint64_t q = 0; q < n; ++q) off_max = std::max(off_max, offsets[q]);
    if (int rc = check_range(h, first, n ? off_max + 1 : 0)) return rc;
    for (uint64_t q = 0; q < n; ++q) {
        words = std::max(words, out_offsets[q] + (t_len[q] + 1) * h->w64);
        tmax = std::max(tmax, t_len[q]);
    }
    return run_sim_common(h, first, n, tmax, offsets, t_len, out_offsets, words, out, nullptr, nullptr, stats);
}
