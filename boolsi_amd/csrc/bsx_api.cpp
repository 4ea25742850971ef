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
#include <unordered_map>
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
hipError_t launch_digit_lifetimes(int nw, int k, int lut_mode, size_t shmem, hipStream_t st, const LifetimeParams& P);
hipError_t launch_compact_near(const uint32_t* seg, const uint32_t* counts, uint32_t n_seg, uint64_t cap, uint32_t nw, uint32_t* out, uint32_t* zero, uint32_t zero_words, hipStream_t stream);
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
        (e = h->d_ctr.alloc(1)) != hipSuccess || (e = hipHostMalloc((void**)&h->h_ctr, sizeof(Counters), hipHostMallocDefault)) != hipSuccess) {
        g_create_error = std::string("stream/event creation: ") + hipGetErrorString(e);
        if (h->h_ctr) (void)hipHostFree(h->h_ctr);
        delete h;
        return BSX_ERR_HIP;
    }
    const char* cc_env = std::getenv("BSX_CYCLE_CACHE");       // "0" disables the cycle-state cache (A/B runs, tests)
    h->cache_enabled = !(cc_env && cc_env[0] == '0');
    if ((e = h->d_cc_journal.alloc(kCycleJournalCap)) != hipSuccess ||
        (e = h->d_cc_claims.alloc(kCycleClaimSlots)) != hipSuccess || (e = h->d_cc_count.alloc(1)) != hipSuccess) {
        g_create_error = std::string("cycle cache allocation: ") + hipGetErrorString(e);
        if (h->h_ctr) (void)hipHostFree(h->h_ctr);
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
    if (h->h_ctr) (void)hipHostFree(h->h_ctr);
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
    HIPCHK(h, configure_target((int)nw, (int)k_mux, h->lut_mode, h->shmem + 16 + kTargetHistBins * 8));
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
    h->h_any = any;
    h->h_fv = fv;
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
    h->cube_depth_cap = 0;
    h->life_valid = 0;
    h->image_n = ~size_t(0);
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

}  // namespace

namespace {

constexpr uint64_t kFastMinProblems = 8192;     // below this the general kernel alone is used
constexpr uint64_t kDiscoverySample = 65536;    // problems (sampled over the range) run through the detector when nothing is cached yet
constexpr uint64_t kLeanTile = 1ull << 28;      // problems per lean-kernel launch (straggler list: 4 B each)
constexpr uint32_t kFastSteps = 48;             // FAST phase length (steps without a cached cycle state), first guess
constexpr uint32_t kFastStepsMax = 3072;
constexpr uint64_t kProbeTile = 1ull << 22;     // lean tiles while the FAST length is being calibrated
constexpr uint32_t kCubeMinBits = 16;           // cube collapse: smallest aligned block handled as a cube
constexpr uint32_t kCubeMaxBits = 48;           // ... and the largest (= the per-call limit)

using MergedTable = std::unordered_map<Key8, bsx_attr_rec, Key8Hash>;

struct AttractRun {
    Counters ctr{};
    float ms = 0.f;
};

// One k_attract launch (general or fast) + merge of its log into `merged` unless `discard_log`.
enum PassKind { kPassGeneral = 0, kPassLean = 1, kPassPool = 2 };

// LDS mirror size for the lean / pool kernels: they fill the mirror once from the journal, so it only has
// to hold what the journal holds (4 slots per state keeps probe chains short); a smaller mirror leaves
// the LDS to more workgroups.  The general kernel inserts while it runs and keeps the full size.
int mirror_slots_for(bsx_handle h, uint32_t* slots_out) {
    // At least 2 slots per entry (a cube pass adds one representative entry per state), 4 where that still lets
    // two workgroups share a CU's LDS: at n = 64 a pool workgroup is 75.7 KiB + mirror, so a 256-slot mirror
    // already halves the occupancy (measured: 3 instead of 6 waves per SIMD, profiles/r02_pmc notes).
    const uint64_t entries = (h->cube_mirror ? 2 : 1) * h->journal_states;
    uint32_t slots = 64;
    while (slots < 2 * entries && slots < h->cache_lds_slots) slots *= 2;
    const size_t fixed = h->shmem + 32 + pool_extra_bytes(h->net.nw);
    while (slots < 4 * entries && slots < h->cache_lds_slots && fixed + (size_t)2 * slots * h->cache_stride <= 80 * 1024) slots *= 2;
    h->mirror_slots = *slots_out = std::min(slots, h->cache_lds_slots);
    if (std::getenv("BSX_DEBUG")) std::fprintf(stderr, "[bsx] mirror: %llu cycle states cached, %u slots\n", (unsigned long long)h->journal_states, *slots_out);
    return BSX_OK;
}

int lean_mirror_slots(bsx_handle h, uint32_t* slots_out) {
    uint32_t ignored = 0;
    if (!slots_out) slots_out = &ignored;
    if (!h->journal_stale) return mirror_slots_for(h, slots_out);
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
    h->journal_states = states;
    h->journal_stale = false;
    return mirror_slots_for(h, slots_out);
}

void merge_records(MergedTable& merged, const LogRec* recs, size_t n, uint32_t nw);

static double g_prof[6];      // BSX_PROFILE: host time per section of a pass, ms
int launch_attract_pass(bsx_handle h, AttractParams& P, int kind, DevBuf<LogRec>& d_log, MergedTable* merged,
                        AttractRun& run) {
    const double pt0 = now_ms();
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
    if (kind == kPassPool) {
        // Cube passes under 2^28 classes: even fixed shares, no cursor traffic (their classes cost about the same
        // everywhere).  Larger ones and plain tiles, whose cost per problem varies by region: every wave starts
        // with one piece and takes the rest from the cursor, 4096 at a time (measured on config 3's plain tiles:
        // fixed three-quarter shares 2.5 ms against 1.9 ms; a read of the cursor before each atomic cost 0.5 ms more).
        const uint64_t n_waves = (uint64_t)L.grid.x * (kPoolBlockThreads / 64);
        if (P.merge == 3 && P.count < (1ull << 28)) {
            P.chunk_first = ((P.count + n_waves - 1) / n_waves + 63) / 64 * 64;
            P.chunk = 0;
        } else {
            P.chunk_first = P.chunk;
        }
    }
    if (const char* c = std::getenv("BSX_CHUNK")) { P.chunk = (uint32_t)std::max(64, std::atoi(c)); P.chunk_first = P.chunk; }     // tuning knob
    const uint64_t waves = (uint64_t)L.grid.x * kWavesPerBlock;
    const uint64_t log_cap = waves * kTableSlots + (1u << 16);
    if (d_log.n < log_cap) HIPCHK(h, d_log.alloc(log_cap));
    P.log = d_log.p;
    P.log_cap = log_cap;
    // results that are kept may spill from the log into the HBM attractor table (general kernel only: the
    // lean / pool kernels write at most one record per workgroup and cached attractor)
    P.table = (merged && !fast && h->table_slots) ? h->d_table.p : nullptr;
    P.table_mask = h->table_slots ? h->table_slots - 1 : 0;
    const bool lists = kind == kPassPool && P.merge == 3 && P.cube_depth > 1;
    if (lists) {
        // classes for the level below: one segment per workgroup (P.near_cap = the caller's total, split here)
        const uint32_t nw = h->net.nw;
        P.near_cap = std::max<uint64_t>(1, P.near_cap / L.grid.x);
        if (const char* e = std::getenv("BSX_CUBE_NEAR_CAP")) P.near_cap = (uint64_t)std::max(1, std::atoi(e));    // (tests: force the shallower restart)
        HIPCHK(h, h->d_near_seg.reserve((size_t)L.grid.x * P.near_cap * nw));
        HIPCHK(h, h->d_near_counts.reserve(L.grid.x));
        P.near = h->d_near_seg.p;
        P.near_counts = h->d_near_counts.p;
    }
    if (kind == kPassPool && !(std::getenv("BSX_MIRROR_IMAGE") && std::getenv("BSX_MIRROR_IMAGE")[0] == '0')) {
        // the cache mirror as an image: rebuilt (one workgroup) only when the journal or the mirror size has changed
        const size_t words = 4 + (size_t)P.cc.lds_slots * (h->cache_stride / 4);
        if (h->image_n != h->h_journal.size() || h->image_slots != P.cc.lds_slots || h->d_mirror.n < words) {
            HIPCHK(h, h->d_mirror.reserve(words));
            AttractParams B = P;
            B.count = 0;
            B.mirror_image = nullptr;
            B.mirror_out = h->d_mirror.p;
            HIPCHK(h, launch_attract_pool((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, dim3(1), shmem, h->stream, B));
            h->image_n = h->h_journal.size();
            h->image_slots = P.cc.lds_slots;
        }
        P.mirror_image = h->d_mirror.p;
        P.mirror_out = nullptr;
    }
    const double pt1 = now_ms();
    if (h->ctr_zeroed) h->ctr_zeroed = false;               // (k_compact_near of the pass before has cleared them)
    else HIPCHK(h, hipMemsetAsync(h->d_ctr.p, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if (kind == kPassPool) HIPCHK(h, launch_attract_pool((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    else if (kind == kPassLean) HIPCHK(h, launch_attract_fast((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    else HIPCHK(h, launch_attract((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    const double pt2 = now_ms();
    HIPCHK(h, hipMemcpyAsync(h->h_ctr, h->d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    run.ctr = *h->h_ctr;
    const double pt3 = now_ms();
    HIPCHK(h, hipEventElapsedTime(&run.ms, h->ev0, h->ev1));
    g_prof[0] += pt1 - pt0; g_prof[1] += pt2 - pt1; g_prof[2] += pt3 - pt2; g_prof[3] += run.ms;
    if (lists && run.ctr.near_classes && !run.ctr.near_overflow) {
        HIPCHK(h, h->d_near_list.reserve((size_t)run.ctr.near_classes * h->net.nw));
        static_assert(sizeof(Counters) % 4 == 0, "cleared word by word");
        HIPCHK(h, launch_compact_near(h->d_near_seg.p, h->d_near_counts.p, L.grid.x, P.near_cap, h->net.nw, h->d_near_list.p,
                                      reinterpret_cast<uint32_t*>(h->d_ctr.p), (uint32_t)(sizeof(Counters) / 4), h->stream));
        h->ctr_zeroed = true;
    }
    if (std::getenv("BSX_DEBUG"))
        std::fprintf(stderr, "[bsx] %s pass: %llu problems, %llu lane-steps, %llu stragglers, %.3f ms (BSX_DIAG build: %llu wave iterations, %llu service rounds)\n",
                     kind == kPassPool ? "pool" : fast ? "lean" : "general", (unsigned long long)P.count, (unsigned long long)run.ctr.steps_exec,
                     (unsigned long long)run.ctr.n_stragglers, run.ms, (unsigned long long)run.ctr.wave_iters,
                     (unsigned long long)run.ctr.service_rounds);
    if (std::getenv("BSX_DEBUG") && run.ctr.wave_iters)
        std::fprintf(stderr, "[bsx]   diag: kept after fresh stages %llu, lanes into pool stages %llu, kept after pool stages %llu, merged away %llu\n",
                     (unsigned long long)run.ctr.diag[0], (unsigned long long)run.ctr.diag[1], (unsigned long long)run.ctr.diag[2], (unsigned long long)run.ctr.diag[3]);
    if (std::getenv("BSX_DEBUG") && run.ctr.phase_max[0])
        std::fprintf(stderr, "[bsx]   diag: %u workgroups; prologue / loop / epilogue, us: mean %.1f / %.1f / %.1f, slowest %.1f / %.1f / %.1f\n", L.grid.x,
                     run.ctr.phase_sum[0] / 100.0 / L.grid.x, run.ctr.phase_sum[1] / 100.0 / L.grid.x, run.ctr.phase_sum[2] / 100.0 / L.grid.x,
                     run.ctr.phase_max[0] / 100.0, run.ctr.phase_max[1] / 100.0, run.ctr.phase_max[2] / 100.0);
    if (!merged) return BSX_OK;                 // results discarded (discovery): a full log does not matter
    if (run.ctr.table_inserts) h->table_dirty = true;
    if (run.ctr.log_overflow) return fail(h, BSX_ERR_TABLE_FULL, "device attractor log overflowed");
    if (run.ctr.table_overflow) return fail(h, BSX_ERR_TABLE_FULL, "more distinct attractors than the caller's table capacity (device table full)");
    if (kind == kPassPool && P.merge == 3) {    // cube pass: the sums came back with the counters
        LogRec recs[64];
        size_t n_recs = 0;
        for (uint32_t a = 0; a < 64; ++a) {
            if (!run.ctr.acc_cnt[a]) continue;
            LogRec& r = recs[n_recs++];
            for (int w = 0; w < kMaxW32; ++w) r.key[w] = run.ctr.acc_key[a][w];
            r.length = run.ctr.acc_len[a]; r.pad = 0; r.count = run.ctr.acc_cnt[a]; r.sum_l = run.ctr.acc_sl[a];
            r.sum_l2_lo = run.ctr.acc_sl2_lo[a]; r.sum_l2_hi = run.ctr.acc_sl2_hi[a];
        }
        merge_records(*merged, recs, n_recs, h->net.nw);
        return BSX_OK;
    }
    const uint64_t n_log = std::min<uint64_t>(run.ctr.log_cursor, log_cap);
    std::vector<LogRec> log(n_log);
    if (n_log) HIPCHK(h, hipMemcpy(log.data(), d_log.p, n_log * sizeof(LogRec), hipMemcpyDeviceToHost));
    // merge by key (attract.py:405-455 write_aggregated_attractors_to_db, exact integers)
    const uint32_t nw = h->net.nw;
    merge_records(*merged, log.data(), log.size(), nw);
    return BSX_OK;
}

void merge_records(MergedTable& merged, const LogRec* recs, size_t n, uint32_t nw) {
    for (size_t i = 0; i < n; ++i) {
        const LogRec& r = recs[i];
        const Key8 key = key8(r.key);
        auto it = merged.find(key);
        if (it == merged.end()) {
            bsx_attr_rec a{};
            for (uint32_t w = 0; w < nw; ++w) a.key[w >> 1] |= (uint64_t)r.key[w] << (32 * (w & 1));
            a.length = r.length;
            it = merged.emplace(key, a).first;
        }
        bsx_attr_rec& a = it->second;
        a.count += r.count;
        a.sum_l += r.sum_l;
        const uint64_t lo = a.sum_l2_lo + r.sum_l2_lo;
        a.sum_l2_hi += r.sum_l2_hi + (lo < a.sum_l2_lo ? 1 : 0);
        a.sum_l2_lo = lo;
    }
}

void fold_table(MergedTable& into, const MergedTable& from) {
    for (const auto& kv : from) {
        auto it = into.find(kv.first);
        if (it == into.end()) { into.emplace(kv.first, kv.second); continue; }
        bsx_attr_rec& a = it->second;
        a.count += kv.second.count;
        a.sum_l += kv.second.sum_l;
        const uint64_t lo = a.sum_l2_lo + kv.second.sum_l2_lo;
        a.sum_l2_hi += kv.second.sum_l2_hi + (lo < a.sum_l2_lo ? 1 : 0);
        a.sum_l2_lo = lo;
    }
}

// ---- cube collapse: which of the `a` lowest initial-state digits can the FIRST update of the
// block starting at digit value d_lo depend on?  A node's rule, restricted to the block's fixed bits, depends
// on a free predecessor iff flipping it changes the output for some assignment of the rule's other free
// inputs; a digit is relevant iff its node is such a predecessor of some node (fixed nodes have constant
// rules, model.py:45-47).  f(s) is then a function of the relevant digits alone -- exactly, not heuristically.
struct Cube {
    uint64_t d_lo;              // first digit value (multiple of 2^a)
    uint32_t a;                 // log2 of the problems in the block
    std::vector<uint32_t> rel;  // relevant digits: ascending from build_cube, then in class-index bit order
    uint32_t base[kMaxW32];     // the block's fixed bits, free bits zero
    DevSpace sp;                // enumeration of the relevant digits' assignments (plan_cube)
    uint32_t umask[kMaxW32];    // node bits of the irrelevant free digits
    uint32_t free_mask[kMaxW32];
    bool ok = false;            // false: more deposit runs than the kernels take
};

void build_cube(const bsx_engine* h, uint64_t d_lo, uint32_t a, Cube& c, const uint32_t* fixmask = nullptr) {
    if (!fixmask) fixmask = h->sp.fixmask;      // (target passes: the fixed nodes of the block's fixed-node variant)
    const uint32_t n = h->n_nodes, nw = h->net.nw;
    c.d_lo = d_lo; c.a = a; c.rel.clear(); c.ok = false;
    uint32_t base[kMaxW32];     // origin bits + the block's fixed digits
    for (int w = 0; w < kMaxW32; ++w) { base[w] = h->sp.origin[w]; c.umask[w] = 0; c.free_mask[w] = 0; }
    std::vector<char> is_free(n, 0), relevant(n, 0);
    for (uint32_t j = 0; j < h->sp.n_any; ++j) {
        const uint32_t node = h->h_any[j];
        if (j < a) { is_free[node] = 1; c.free_mask[node >> 5] |= 1u << (node & 31); }
        else if ((d_lo >> j) & 1ull) base[node >> 5] |= 1u << (node & 31);
    }
    for (uint32_t i = 0; i < n; ++i) {
        if ((fixmask[i >> 5] >> (i & 31)) & 1u) continue;
        const uint32_t k = h->h_pred_offsets[i + 1] - h->h_pred_offsets[i];
        const uint32_t* preds = h->h_pred_idx.data() + h->h_pred_offsets[i];
        if (k > (uint32_t)kMaxMuxK) {                    // wide rule: every free input counts (conservative)
            for (uint32_t j = 0; j < k; ++j) if (is_free[preds[j]]) relevant[preds[j]] = 1;
            continue;
        }
        const uint64_t tt = h->h_tt0[i];
        uint32_t free_slots = 0, fixed_idx = 0;
        for (uint32_t j = 0; j < k; ++j) {
            if (is_free[preds[j]]) free_slots |= 1u << j;
            else if ((base[preds[j] >> 5] >> (preds[j] & 31)) & 1u) fixed_idx |= 1u << j;
        }
        for (uint32_t j = 0; j < k; ++j) {
            if (!((free_slots >> j) & 1u) || relevant[preds[j]]) continue;
            const uint32_t others = free_slots & ~(1u << j);
            uint32_t x = 0;
            do {                                        // all assignments of the other free inputs
                const uint32_t idx = fixed_idx | x;
                if (((tt >> idx) ^ (tt >> (idx | (1u << j)))) & 1ull) { relevant[preds[j]] = 1; break; }
                x = (x - others) & others;
            } while (x);
        }
    }
    for (uint32_t j = 0; j < a; ++j) {
        const uint32_t node = h->h_any[j];
        if (relevant[node]) c.rel.push_back(j);
        else c.umask[node >> 5] |= 1u << (node & 31);
    }
    for (uint32_t w = 0; w < (uint32_t)kMaxW32; ++w) c.base[w] = w < nw ? base[w] : 0u;
    c.ok = c.rel.size() <= kMaxDepositRuns;
}

// Enumeration space of the cube: class-index bit q -> the node of c.rel[q] (one deposit run per relevant
// digit, in the order c.rel lists them), everything else fixed.
void plan_cube(const bsx_engine* h, Cube& c) {
    DevSpace sp = h->sp;
    for (uint32_t w = 0; w < (uint32_t)kMaxW32; ++w) sp.origin[w] = c.base[w];
    sp.n_any = (uint32_t)c.rel.size();
    sp.identity_any = 0;
    for (int w = 0; w < 4; ++w) sp.first_digits[w] = 0;
    sp.first_variant = 0;
    sp.n_runs = (uint32_t)c.rel.size();
    for (uint32_t q = 0; q < c.rel.size(); ++q) {
        const uint32_t node = h->h_any[c.rel[q]];
        sp.deposit[2 * q] = q | (node >> 5) << 8 | (node & 31u) << 16;
        sp.deposit[2 * q + 1] = 1u;
    }
    c.sp = sp;
}

// Deeper collapse: the digits of the block that F^d(x) still depends on, d = 1 .. max_depth, as masks over the
// digit index (out[d - 1]; a <= 48).  Constant propagation over the block: a node's value after s updates is
// 0, 1 or "varies" with the set of free digits it may depend on; a rule is restricted to the inputs that are
// constant over the block and counts a varying input only if the restricted truth table is sensitive to it.
// An over-approximation (never misses a dependence), and out[0] is build_cube's set.  out[d] is a subset of
// out[d - 1]: the members of a depth-d class share F^d(x) and everything after it.
void cube_levels(const bsx_engine* h, const Cube& c, uint32_t max_depth, std::vector<uint64_t>& out) {
    const uint32_t n = h->n_nodes;
    const uint32_t* fixmask = h->sp.fixmask;
    std::vector<uint8_t> val(n), nval(n);       // 0 / 1 / 2 = varies
    std::vector<uint64_t> dep(n, 0), ndep(n, 0);
    for (uint32_t i = 0; i < n; ++i) val[i] = (c.base[i >> 5] >> (i & 31)) & 1u;
    for (uint32_t j = 0; j < c.a; ++j) { const uint32_t node = h->h_any[j]; val[node] = 2; dep[node] = 1ull << j; }
    out.clear();
    for (uint32_t d = 1; d <= max_depth; ++d) {
        uint64_t all = 0;
        for (uint32_t i = 0; i < n; ++i) {
            ndep[i] = 0;
            if ((fixmask[i >> 5] >> (i & 31)) & 1u) { nval[i] = (h->sp.fixval[i >> 5] >> (i & 31)) & 1u; continue; }
            const uint32_t k = h->h_pred_offsets[i + 1] - h->h_pred_offsets[i];
            const uint32_t* preds = h->h_pred_idx.data() + h->h_pred_offsets[i];
            if (k > (uint32_t)kMaxMuxK) {                // wide rule: varies with whatever its inputs vary with (conservative)
                nval[i] = 2;
                for (uint32_t j = 0; j < k; ++j) ndep[i] |= dep[preds[j]];
                continue;
            }
            const uint64_t tt = h->h_tt0[i];
            uint32_t var_slots = 0, fixed_idx = 0;
            for (uint32_t j = 0; j < k; ++j) {
                if (val[preds[j]] == 2) var_slots |= 1u << j;
                else if (val[preds[j]]) fixed_idx |= 1u << j;
            }
            uint32_t seen = 0, sens = 0, x = 0;
            do {                                        // all assignments of the varying inputs
                const uint32_t idx = fixed_idx | x;
                seen |= 1u << ((tt >> idx) & 1ull);
                for (uint32_t j = 0; j < k; ++j)
                    if (((var_slots >> j) & 1u) && (((tt >> idx) ^ (tt >> (idx ^ (1u << j)))) & 1ull)) sens |= 1u << j;
                x = (x - var_slots) & var_slots;
            } while (x);
            if (seen != 3u) { nval[i] = seen >> 1; continue; }
            nval[i] = 2;
            for (uint32_t j = 0; j < k; ++j) if ((sens >> j) & 1u) ndep[i] |= dep[preds[j]];
        }
        // the origin's perturbation schedule overrides the rules at time d (model.py:68-71): constants for every member
        for (size_t e = 0; e + 2 < h->h_sched.size(); e += 3)
            if (h->h_sched[e] == d) { nval[h->h_sched[e + 1]] = (uint8_t)h->h_sched[e + 2]; ndep[h->h_sched[e + 1]] = 0; }
        all = 0;
        for (uint32_t i = 0; i < n; ++i) all |= ndep[i];
        out.push_back(all);
        val.swap(nval);
        dep.swap(ndep);
    }
}

// Relevant digits whose influence dies out first become the lowest class-index bits (k_digit_lifetimes):
// the classes that merge after a step or two then sit in the same batch.  A heuristic for speed only.
int order_cube_digits(bsx_handle h, Cube& c) {
    const uint32_t r = (uint32_t)c.rel.size();
    if (r < 2 || r > 64 || (std::getenv("BSX_CUBE_ORDER") && std::getenv("BSX_CUBE_ORDER")[0] == '0')) return BSX_OK;
    uint64_t need = 0;
    for (uint32_t q = 0; q < r; ++q) need |= 1ull << c.rel[q];
    // (a pass of 2^26 classes or more takes milliseconds: worth the 30 us of measuring on this very block)
    if ((need & ~h->life_valid) || r >= 26) {
        LifetimeParams L{};
        L.net = h->net;
        for (int w = 0; w < kMaxW32; ++w) { L.fixmask[w] = h->sp.fixmask[w]; L.fixval[w] = h->sp.fixval[w]; L.base[w] = c.base[w]; L.free_mask[w] = c.free_mask[w]; }
        L.n_digits = r;
        for (uint32_t q = 0; q < r; ++q) L.node[q] = h->h_any[c.rel[q]];
        HIPCHK(h, h->d_life.reserve(64));
        HIPCHK(h, hipMemsetAsync(h->d_life.p, 0, 64 * sizeof(uint32_t), h->stream));
        L.out = h->d_life.p;
        HIPCHK(h, launch_digit_lifetimes((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, h->shmem, h->stream, L));
        uint32_t measured[64];
        HIPCHK(h, hipMemcpyAsync(measured, h->d_life.p, sizeof(measured), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (uint32_t q = 0; q < r; ++q) h->life_cache[c.rel[q]] = measured[q];
        h->life_valid |= need;
    }
    uint32_t life[64];
    for (uint32_t q = 0; q < r; ++q) life[q] = h->life_cache[c.rel[q]];
    std::vector<uint32_t> idx(r);
    for (uint32_t q = 0; q < r; ++q) idx[q] = q;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return life[x] < life[y]; });
    std::vector<uint32_t> rel(r);
    for (uint32_t q = 0; q < r; ++q) rel[q] = c.rel[idx[q]];
    c.rel = rel;
    return BSX_OK;
}

// Entries of the HBM attractor table -> `merged`; the table is left empty for the next call.
int drain_attractor_table(bsx_handle h, MergedTable& merged) {
    if (!h->table_dirty) return BSX_OK;
    h->table_dirty = false;
    DevBuf<unsigned long long> d_cursor;
    DevBuf<LogRec> d_out;
    HIPCHK(h, d_cursor.alloc(1));
    HIPCHK(h, hipMemsetAsync(d_cursor.p, 0, sizeof(unsigned long long), h->stream));
    HIPCHK(h, d_out.alloc(h->table_slots));
    HIPCHK(h, launch_table_drain(h->d_table.p, h->table_slots, d_out.p, h->table_slots, d_cursor.p, h->stream));
    unsigned long long n = 0;
    HIPCHK(h, hipMemcpyAsync(&n, d_cursor.p, sizeof(n), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<LogRec> recs(n);
    if (n) HIPCHK(h, hipMemcpy(recs.data(), d_out.p, n * sizeof(LogRec), hipMemcpyDeviceToHost));
    merged.reserve(merged.size() + n);
    merge_records(merged, recs.data(), recs.size(), h->net.nw);
    return BSX_OK;
}

// first + delta for spaces whose initial-state digits fit one word (the fast path's precondition)
void advance_first(DevSpace& sp, const bsx_index* first, uint64_t delta) {
    for (int w = 0; w < 4; ++w) sp.first_digits[w] = first->init_digits[w];
    sp.first_digits[0] += delta;
    sp.first_variant = first->variant;
}

}  // namespace

// Spaces with more than 64 'any' nodes (e.g. a 128-node network with every node 'any'): a call covers at most
// 2^48 consecutive problems, so only the 64 lowest initial-state digits can change inside it.  The call is run as
// the space in which exactly those are 'any' and the higher digits of `first` are part of the origin state -- a
// plain space, which gets the lean / pool / cube paths.  (Same network, same fixed nodes: the cycle cache carries over.)
static int run_attract_low_digits(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t, uint64_t max_len,
                                  bsx_attr_rec* table, uint32_t cap, uint32_t* n_out, uint64_t* n_no_attractor, bsx_stats* stats) {
    struct Restore {                    // the handle describes the whole space again, whatever happens below
        bsx_handle h; DevSpace sp; std::vector<uint32_t> any;
        ~Restore() { h->sp = sp; h->h_any = any; h->in_low_digit_call = false; }
    } restore{h, h->sp, h->h_any};
    h->in_low_digit_call = true;
    DevSpace sv = h->sp;
    bool identity = true;
    for (uint32_t j = 0; j < 64; ++j) identity = identity && h->h_any[j] == j;
    for (uint32_t j = 64; j < h->sp.n_any; ++j)
        if ((first->init_digits[j >> 6] >> (j & 63)) & 1ull) sv.origin[h->h_any[j] >> 5] |= 1u << (h->h_any[j] & 31);
    sv.n_any = 64;
    sv.identity_any = identity ? 1 : 0;
    sv.n_runs = 0;
    if (!identity) {
        uint32_t j = 0, r = 0;
        while (j < 64) {
            uint32_t len = 1;
            while (j + len < 64 && h->h_any[j + len] == h->h_any[j] + len && ((h->h_any[j] + len) >> 5) == (h->h_any[j] >> 5)) ++len;
            sv.deposit[2 * r] = j | (h->h_any[j] >> 5) << 8 | (h->h_any[j] & 31u) << 16;
            sv.deposit[2 * r + 1] = len >= 32 ? 0xFFFFFFFFu : (1u << len) - 1u;
            ++r;
            j += len;
        }
        sv.n_runs = r;                  // <= 64 = kMaxDepositRuns
    }
    h->sp = sv;
    h->h_any.resize(64);
    bsx_index f{};
    f.init_digits[0] = first->init_digits[0];
    return bsx_run_attract(h, &f, count, max_t, max_len, table, cap, n_out, n_no_attractor, nullptr, stats);
}

extern "C" int bsx_run_attract(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                               uint64_t max_len, bsx_attr_rec* table, uint32_t cap, uint32_t* n_out,
                               uint64_t* n_no_attractor, bsx_problem_rec* per_problem, bsx_stats* stats) {
    if (!h) return BSX_ERR_INVALID;
    h->ctr_zeroed = false;      // (whatever an earlier call left behind: other entry points use the counters too)
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
    if (const char* fg = std::getenv("BSX_FGRAPH")) {           // knob: route eligible calls through the functional-graph mode
        if (fg[0] == '1' && !per_problem && h->n_nodes <= 32 && h->sp.n_any == h->n_nodes && h->sp.identity_any && !h->sp.n_fv &&
            !h->sp.n_pv && h->lut_mode != 2)
            return bsx_run_attract_fgraph(h, first, count, max_t, max_len, table, cap, n_out, n_no_attractor, stats);
    }
    // 2^48: sum_l (64 bits) holds count x trajectory length; ranges above 2^32 must collapse into cubes (below)
    if (count > (1ull << 48)) return fail(h, BSX_ERR_INVALID, "at most 2^48 problems per call");
    if (per_problem && count > (1ull << 32)) return fail(h, BSX_ERR_INVALID, "at most 2^32 problems per call with per-problem records");

    if (h->sp.n_any > 64 && !h->sp.n_fv && !h->sp.n_pv && h->sp.tp_origin <= 200 && !per_problem && !h->in_low_digit_call &&
        count >= (1u << 13) && h->cache_enabled && first->init_digits[0] + (count - 1) >= first->init_digits[0] &&
        !(std::getenv("BSX_LEAN") && std::atoi(std::getenv("BSX_LEAN")) == 0)) {
        const int rc = run_attract_low_digits(h, first, count, max_t, max_len, table, cap, n_out, n_no_attractor, stats);
        if (stats) stats->total_ms = now_ms() - t_begin;
        return rc;
    }

    DevBuf<LogRec>& d_log = h->d_log;
    DevBuf<ProblemRec32> d_pp;
    if (per_problem) HIPCHK(h, d_pp.alloc(count));
    if (h->table_dirty) { MergedTable stale; if (int rc = drain_attractor_table(h, stale)) return rc; }     // a failed call left entries behind
    {   // HBM attractor table behind the log: two slots per entry of the caller's table (kept zeroed between calls)
        uint64_t want = 1ull << 16;
        while (want < 2 * (uint64_t)cap) want *= 2;
        if (h->table_slots < want) {
            HIPCHK(h, h->d_table.alloc(want));
            HIPCHK(h, hipMemset(h->d_table.p, 0, want * sizeof(LogRec)));
            h->table_slots = want;
            h->table_dirty = false;
        }
    }

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
                sample[i] = (uint32_t)((z ^ (z >> 31)) % std::min<uint64_t>(count, 1ull << 32));
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
    // Lean / pool kernel over [done, seg_end) in tiles; returns with done < seg_end when the fast path gave up.
    auto run_tiles = [&](uint64_t seg_end) -> int {
    while (use_fast && h->fast_ok && done < seg_end) {
        const uint64_t tile = std::min<uint64_t>(seg_end - done, h->fast_calibrated ? kLeanTile : kProbeTile);
        if (int rc = lean_mirror_slots(h, nullptr)) return rc;          // (refreshes h->h_journal if the detector ran since)
        const unsigned int known_before_tile = (unsigned int)h->h_journal.size();
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
        fold_table(merged, tile_table);
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
            if (r.ctr.n_stragglers > tile / 2) {
                // Most of the tile went to the detector.  If that taught the cache new attractors (a region of the
                // space nobody had visited), the next tile will do better; if not -- cycles too long to cache,
                // or more attractors than the mirror holds -- the lean path is the wrong tool for this space.
                unsigned int known_now = 0;
                HIPCHK(h, hipMemcpy(&known_now, h->d_cc_count.p, sizeof(known_now), hipMemcpyDeviceToHost));
                if (known_now <= known_before_tile) h->fast_ok = false;
            }
        }
    }
    if (done < seg_end) {           // not (or no longer) a case for the lean path: the detector takes the rest
        AttractParams Q = P;
        if (done) advance_first(Q.sp, first, done);
        Q.count = seg_end - done;
        if (Q.count > (1ull << 32)) return fail(h, BSX_ERR_UNSUPPORTED, "range above 2^32 problems that neither collapses into cubes nor fits the lean path");
        Q.per_problem = per_problem ? d_pp.p + done : nullptr;
        AttractRun r;
        if (int rc = launch_attract_pass(h, Q, kPassGeneral, d_log, &merged, r)) return rc;
        account(r);
        done = seg_end;
    }
    return BSX_OK;
    };


    // ---- cube collapse: aligned blocks of >= 2^kCubeMinBits problems are enumerated by their relevant digits
    // only (see build_cube).  Everything before the first / after the last such block goes through the tiles.
    const char* cubes_env = std::getenv("BSX_CUBES");                     // "0": off (A/B runs, tests)
    // (a warm-up under origin perturbations is fine: the first update still depends on the relevant digits only)
    const bool cubes_ok = use_fast && merge_mode == 2 && !per_problem &&
                          !(cubes_env && cubes_env[0] == '0') && h->sp.n_any >= kCubeMinBits;
    auto run_cube = [&](const Cube& c1, bool& collapsed) -> int {
        collapsed = false;
        const uint32_t nw = h->net.nw, rec_words = nw + 3;
        const uint64_t tp = h->sp.tp_origin;            // the search starts at s(T_p); class times count from there
        const uint64_t cap_rel = max_t == BSX_T_INF ? BSX_T_INF : max_t - tp;
        const uint32_t cap_rel32 = (cap_rel == BSX_T_INF || cap_rel >= (kStepLimit / 4)) ? 0xFFFFFFFFu : (uint32_t)cap_rel;
        const uint64_t list_cap = 1ull << 20;                   // unresolved classes per pass
        const uint64_t near_cap = 1ull << 23;                   // classes a deep pass may hand to the level below (split over its workgroups)
        if (h->d_strag.n < list_cap * rec_words) HIPCHK(h, h->d_strag.alloc(list_cap * rec_words));
        const uint32_t fast_steps = (uint32_t)std::min<uint64_t>((uint64_t)cap_rel32 + 1, std::min<uint32_t>(kFastStepsMax, std::max(192u, 4 * h->fast_steps)));

        // ---- levels: rel_mask[d - 1] = digits F^d depends on.  The top level is the depth with the fewest digits (the
        // shallowest such: a deeper one would only add updates); BSX_CUBE_DEPTH caps it (1 = first update only) ...
        uint32_t max_depth = 8;
        if (const char* e = std::getenv("BSX_CUBE_DEPTH")) max_depth = (uint32_t)std::max(1, std::min(16, std::atoi(e)));
        if (h->cube_depth_cap) max_depth = std::min(max_depth, h->cube_depth_cap);
        // with a warm-up the search starts at s(T_p): classes that share F^d, d <= T_p, share every state that counts,
        // so no class has to be handed down -- one pass at the best such depth
        if (tp) max_depth = (uint32_t)std::min<uint64_t>(max_depth, tp);
        max_depth = std::max(1u, std::min(max_depth, fast_steps > 1 ? fast_steps - 1 : 1u));
        std::vector<uint64_t> rel_mask;
        cube_levels(h, c1, max_depth, rel_mask);
        // ... unless the block is so small that the extra launches cost more than the updates they save: estimate
        // 45 us for the first launch, half of that for each level below (not all of them find classes to run), and
        // 2.4e11 class updates per second (tools/depth_survey.py: blocks with under 2^20 depth-1 classes were up to
        // 3x slower through four levels than through one)
        // (an explicit BSX_CUBE_DEPTH keeps the plain rule: tests force levels onto small spaces with it)
        const bool forced_depth = std::getenv("BSX_CUBE_DEPTH") != nullptr;
        uint32_t top = 1;
        double best = 0;
        for (uint32_t d = 1; d <= max_depth; ++d) {
            const int r_d = __builtin_popcountll(rel_mask[d - 1]);
            const double est = forced_depth ? (double)r_d : 45.0 * (1.0 + 0.5 * (d - 1)) + std::ldexp(1.0, r_d) * (d + 0.3) / 2.4e5;
            if (d == 1 || est < best) { best = est; top = d; }
        }
        auto level_cube = [&](uint64_t digits, bool ordered, Cube& lc) -> int {
            lc = c1;
            lc.rel.clear();
            for (uint32_t j = 0; j < c1.a; ++j) if ((digits >> j) & 1ull) lc.rel.push_back(j);
            if (ordered) if (int rc = order_cube_digits(h, lc)) return rc;
            plan_cube(h, lc);
            return BSX_OK;
        };

        for (int attempt = 0; attempt < 32; ++attempt) {
            // every cached attractor must be in the mirror, or a class could sit on a cycle nobody recognises
            uint32_t slots = 0;
            if (int rc = lean_mirror_slots(h, &slots)) return rc;
            uint64_t states = 0;
            for (const CycleRecord& jr : h->h_journal) states += jr.length;
            if (h->h_journal.size() > (size_t)kTagAcc + kLdsAcc || 4 * states > h->cache_lds_slots) return BSX_OK;

            MergedTable pass_table;
            uint64_t pass_none = 0, pass_ref = 0;
            bool repeat = false, lower = false, give_up = false;
            uint64_t n_entries = 0;
            for (uint32_t d = top; d >= 1 && !repeat && !lower; --d) {
                const bool is_top = d == top;
                if (!is_top && n_entries == 0) break;
                const uint64_t here = rel_mask[d - 1];
                const uint32_t r_here = (uint32_t)__builtin_popcountll(here);
                Cube lc;
                if (int rc = level_cube(is_top ? here : here & ~rel_mask[d], is_top, lc)) return rc;
                const uint32_t k_bits = (uint32_t)lc.rel.size();
                AttractParams Q = P;
                Q.sp = lc.sp;
                Q.count = is_top ? 1ull << k_bits : n_entries << k_bits;
                Q.merge = 3;
                Q.cube_shift = c1.a - r_here;
                Q.cube_depth = d;
                Q.entry_shift = k_bits;
                Q.entries = is_top ? nullptr : h->d_near_list.p;        // (packed by the pass above)
                Q.near = nullptr;                                       // (segments: launch_attract_pass)
                Q.near_counts = nullptr;
                Q.near_cap = d > 1 ? near_cap : 0;
                for (int w = 0; w < kMaxW32; ++w) { Q.cube_umask[w] = c1.umask[w]; Q.cube_free[w] = c1.free_mask[w]; }
                Q.fast_steps = fast_steps;
                Q.per_problem = nullptr;
                Q.stragglers = h->d_strag.p;
                Q.stragglers_cap = list_cap * rec_words;
                AttractRun r;
                h->cube_mirror = true;
                const int rc = launch_attract_pass(h, Q, kPassPool, d_log, &pass_table, r);
                h->cube_mirror = false;
                if (rc) return rc;
                kernel_ms += r.ms; ++launches; steps_exec += r.ctr.steps_exec;
                if (std::getenv("BSX_DEBUG")) std::fprintf(stderr, "[bsx] cube 2^%u at digit value %llu: depth %u%s, %u digits here (%u relevant), %llu classes, %llu near a cycle, %llu unresolved\n", c1.a, (unsigned long long)c1.d_lo, d, is_top ? " (top)" : "", k_bits, r_here, (unsigned long long)Q.count, (unsigned long long)r.ctr.near_classes, (unsigned long long)r.ctr.straggler_classes);
                if (r.ctr.straggler_overflow) { give_up = true; break; }    // too many unresolved classes: not a space for cubes
                if (r.ctr.near_overflow) { top = d - 1; h->cube_depth_cap = top; lower = true; break; }     // start over, shallower
                // a level whose classes mostly sit next to a cycle only adds work: later blocks stop above it
                if (d > 1 && 2 * r.ctr.near_classes > Q.count) h->cube_depth_cap = d - 1;
                n_entries = r.ctr.near_classes;
                pass_none += r.ctr.n_none;
                pass_ref += r.ctr.steps_ref;
                const uint64_t n_unres = r.ctr.straggler_classes;
                if (!n_unres) continue;
                // the detector runs from each listed state: a class that was not on a cycle yet gets its exact
                // result (all members share the rest of the trajectory); one that sits on a cycle needs that
                // attractor in the cache -- the detector has just published it -- and the pass is repeated
                std::vector<uint32_t> recs(n_unres * rec_words);
                HIPCHK(h, hipMemcpy(recs.data(), h->d_strag.p, recs.size() * 4, hipMemcpyDeviceToHost));
                std::vector<uint32_t> st(n_unres * nw);
                for (uint64_t i = 0; i < n_unres; ++i) std::copy(recs.begin() + i * rec_words, recs.begin() + i * rec_words + nw, st.begin() + i * nw);
                DevBuf<uint32_t> d_states;
                DevBuf<ProblemRec32> d_res;
                HIPCHK(h, d_states.upload(st));
                HIPCHK(h, d_res.alloc(n_unres));
                AttractParams S = P;
                S.sp = lc.sp;
                S.sp.tp_origin = 0;                     // the listed states are past the warm-up
                S.count = n_unres;
                S.states = d_states.p;
                S.per_problem = d_res.p;
                S.max_len = BSX_T_INF;
                S.merge = 0;
                AttractRun rs;
                if (int rc2 = launch_attract_pass(h, S, kPassGeneral, d_log, nullptr, rs)) return rc2;
                kernel_ms += rs.ms; ++launches; steps_exec += rs.ctr.steps_exec; limit_hits += rs.ctr.step_limit_hits;
                std::vector<ProblemRec32> res(n_unres);
                HIPCHK(h, hipMemcpy(res.data(), d_res.p, n_unres * sizeof(ProblemRec32), hipMemcpyDeviceToHost));
                for (uint64_t i = 0; i < n_unres && !repeat; ++i) {
                    const uint32_t* rec = recs.data() + i * rec_words;
                    const uint64_t t_class = rec[nw], m = ((uint64_t)rec[nw + 2] << 32) | rec[nw + 1];
                    const ProblemRec32& pr = res[i];
                    if (!pr.found) { pass_none += m; pass_ref += m * max_t; continue; }        // (finite cap, or the step limit was hit)
                    if (pr.trajectory_l == 0) { repeat = true; break; }                          // on a cycle: members' mu unknown
                    const uint64_t mu = t_class + pr.trajectory_l, lam = pr.length, traj = tp + mu;
                    const bool found = cap_rel == BSX_T_INF || mu + lam <= cap_rel;
                    pass_ref += found ? m * (traj + lam) : m * max_t;
                    if (!found || lam > max_len) { pass_none += m; continue; }
                    const Key8 key = key8(pr.key);
                    auto it = pass_table.find(key);
                    if (it == pass_table.end()) {
                        bsx_attr_rec rec_a{};
                        for (uint32_t w = 0; w < nw; ++w) rec_a.key[w >> 1] |= (uint64_t)pr.key[w] << (32 * (w & 1));
                        rec_a.length = lam;
                        it = pass_table.emplace(key, rec_a).first;
                    }
                    bsx_attr_rec& e = it->second;
                    // (a class of up to 2^48 members that a very long transient leads to: the ABI's 64-bit sum of l must hold it)
                    const unsigned __int128 wl = (unsigned __int128)m * traj;
                    if ((wl >> 63) != 0 || e.sum_l + (uint64_t)wl < e.sum_l)
                        return fail(h, BSX_ERR_UNSUPPORTED, "sum of trajectory lengths of an attractor exceeds 64 bits: split the range into smaller calls");
                    e.count += m;
                    e.sum_l += m * traj;
                    const unsigned __int128 sq = (unsigned __int128)(m * traj) * traj + e.sum_l2_lo;
                    e.sum_l2_lo = (uint64_t)sq;
                    e.sum_l2_hi += (uint64_t)(sq >> 64);
                }
            }
            if (give_up) return BSX_OK;
            if (lower) continue;
            if (repeat) {
                unsigned int known = 0;
                HIPCHK(h, hipMemcpy(&known, h->d_cc_count.p, sizeof(known), hipMemcpyDeviceToHost));
                if (known <= h->h_journal.size()) return BSX_OK;        // the attractor cannot be cached: no cube for this block
                continue;                                               // (the detector pass marked the journal stale)
            }
            fold_table(merged, pass_table);
            n_none += pass_none;
            steps_ref += pass_ref;
            collapsed = true;
            return BSX_OK;
        }
        return BSX_OK;
    };

    if (cubes_ok) {
        // [first, first + count) in digit values; blocks are aligned in the digit value, not in the offset
        const unsigned __int128 lo = first->init_digits[0], hi = lo + count;
        const unsigned __int128 unit = (unsigned __int128)1 << kCubeMinBits;
        unsigned __int128 at = (lo + unit - 1) / unit * unit;
        const unsigned __int128 body_end = hi / unit * unit;
        if (at < body_end) {
            if (int rc = run_tiles((uint64_t)(at - lo))) return rc;
            while (at < body_end) {
                uint32_t a_bits = kCubeMaxBits;
                while (a_bits > kCubeMinBits && ((at & (((unsigned __int128)1 << a_bits) - 1)) != 0 || at + ((unsigned __int128)1 << a_bits) > body_end)) --a_bits;
                a_bits = std::min(a_bits, h->sp.n_any);
                Cube c;
                build_cube(h, (uint64_t)at, a_bits, c);
                bool collapsed = false;
                // worth it when the block shrinks at least fourfold (otherwise the tiles do as well and keep member masks)
                if (c.ok && c.rel.size() + 2 <= a_bits) {
                    if (int rc = run_cube(c, collapsed)) return rc;
                }
                const uint64_t block_end = (uint64_t)(at - lo) + (1ull << a_bits);      // (a_bits <= 48: fits)
                if (collapsed) done = block_end;
                else if (int rc = run_tiles(block_end)) return rc;
                at += (unsigned __int128)1 << a_bits;
            }
        }
    }
    if (int rc = run_tiles(count)) return rc;

    if (int rc = drain_attractor_table(h, merged)) return rc;
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
    if (std::getenv("BSX_PROFILE")) {
        std::fprintf(stderr, "[bsx] profile: call %.3f ms; passes: setup %.3f, enqueue %.3f, wait %.3f (kernels %.3f)\n",
                     now_ms() - t_begin, g_prof[0], g_prof[1], g_prof[2], g_prof[3]);
        for (double& v : g_prof) v = 0;
    }
    if (limit_hits) return fail(h, BSX_ERR_STEP_LIMIT, "a trajectory reached the internal step limit without closing its cycle");
    return BSX_OK;
}

// One k_target launch over [first, first + count): optional dense t_hit, optional histogram.
static int launch_target_pass(bsx_handle h, const bsx_index* first, uint64_t skip, uint64_t count, uint64_t max_t,
                              const uint64_t* mask_words, const uint64_t* code_words, uint32_t* d_thit,
                              unsigned long long* d_hist, uint32_t hist_bins, Counters& ctr, float& ms) {
    const size_t shmem = h->shmem + (d_hist ? 16 + (size_t)hist_bins * 8 : 0);
    const Launch L = plan_persistent(h, count, shmem);
    TargetParams P{};
    P.net = h->net;
    P.sp = h->sp;
    set_first(P.sp, first);
    if (skip) {                                 // first + skip, carrying into the variant (init_problem adds the offset the same way)
        unsigned __int128 carry = skip;
        for (int w = 0; w < 4; ++w) { carry += P.sp.first_digits[w]; P.sp.first_digits[w] = (uint64_t)carry; carry >>= 64; }
        const uint32_t n_any = h->sp.n_any;
        if (n_any < 256) {
            // digits at or above n_any belong to the variant number
            uint64_t over = 0;
            for (uint32_t b = n_any; b < 256 && b < n_any + 64; ++b) {
                over |= ((P.sp.first_digits[b >> 6] >> (b & 63)) & 1ull) << (b - n_any);
                P.sp.first_digits[b >> 6] &= ~(1ull << (b & 63));
            }
            P.sp.first_variant += over;
        }
    }
    P.count = count;
    P.chunk = L.chunk;
    P.cap_rel_inf = max_t == BSX_T_INF ? 1 : 0;
    P.max_t = max_t;
    for (uint32_t w = 0; w < h->w64; ++w) {
        P.tmask[2 * w] = (uint32_t)mask_words[w]; P.tcode[2 * w] = (uint32_t)code_words[w];
        if (2 * w + 1 < (uint32_t)kMaxW32) { P.tmask[2 * w + 1] = (uint32_t)(mask_words[w] >> 32); P.tcode[2 * w + 1] = (uint32_t)(code_words[w] >> 32); }
    }
    P.ctr = h->d_ctr.p;
    P.t_hit = d_thit;
    P.hist = d_hist;
    P.hist_bins = d_hist ? hist_bins : 1;
    HIPCHK(h, hipMemsetAsync(h->d_ctr.p, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    HIPCHK(h, launch_target((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    return BSX_OK;
}

// Ordered compaction of t_hit[0..count) -> up to `want` hit records (index order), offsets shifted by `skip`.
static int compact_hits(bsx_handle h, const uint32_t* d_thit, uint64_t count, uint64_t total_hits, uint64_t skip,
                        bsx_hit* hits, uint64_t want) {
    if (!total_hits || !want) return BSX_OK;
    const uint32_t segs = (uint32_t)((count + 4095) / 4096);
    DevBuf<uint32_t> d_cnt;
    DevBuf<uint64_t> d_base;
    DevBuf<HitRec> d_hits;
    const uint64_t n_write = std::min(total_hits, want);
    HIPCHK(h, d_cnt.alloc(segs));
    HIPCHK(h, d_base.alloc(segs));
    HIPCHK(h, d_hits.alloc(n_write));
    HIPCHK(h, launch_compact(d_thit, count, d_cnt.p, nullptr, nullptr, 0, false, h->stream));
    std::vector<uint32_t> cnt(segs);
    HIPCHK(h, hipMemcpyAsync(cnt.data(), d_cnt.p, segs * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<uint64_t> base(segs);
    uint64_t run = 0;
    for (uint32_t i = 0; i < segs; ++i) { base[i] = run; run += cnt[i]; }
    if (run != total_hits) return fail(h, BSX_ERR_HIP, "hit compaction count mismatch");
    HIPCHK(h, hipMemcpyAsync(d_base.p, base.data(), segs * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_compact(d_thit, count, nullptr, d_base.p, d_hits.p, n_write, true, h->stream));    // hits past n_write are dropped
    static_assert(sizeof(HitRec) == sizeof(bsx_hit), "hit layout");
    HIPCHK(h, hipMemcpyAsync(hits, d_hits.p, n_write * sizeof(HitRec), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (skip) for (uint64_t i = 0; i < n_write; ++i) hits[i].offset += skip;
    return BSX_OK;
}

// ------------------------------------------------------------------------------------------------
// Functional-graph mode (bsx_fgraph.hip): attract over [first, first + count) of a space whose n <= 32 nodes
// are all 'any', from N = 2^n-sized arrays.  Same results as bsx_run_attract.
extern "C" int bsx_run_attract_fgraph(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                                      uint64_t max_len, bsx_attr_rec* table, uint32_t cap, uint32_t* n_out,
                                      uint64_t* n_no_attractor, bsx_stats* stats) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->have_net || !h->have_space) return fail(h, BSX_ERR_STATE, "network / problem space not set");
    if (!table || !n_out) return fail(h, BSX_ERR_INVALID, "table / n_out is null");
    if (int rc = check_range(h, first, count)) return rc;
    if (int rc = check_max_t(h, max_t)) return rc;
    const uint32_t n = h->n_nodes;
    if (n > 32 || h->sp.n_any != n || !h->sp.identity_any || h->sp.n_fv || h->sp.n_pv || h->lut_mode == 2)
        return fail(h, BSX_ERR_UNSUPPORTED, "functional-graph mode needs n <= 32 nodes, all of them 'any', and no variations");
    const uint32_t tp = h->sp.tp_origin;                    // origin perturbations: the search starts at s(T_p)
    const double t_begin = now_ms();
    HIPCHK(h, hipSetDevice(h->device));
    *n_out = 0;
    if (n_no_attractor) *n_no_attractor = 0;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (count == 0) return BSX_OK;
    if (h->table_dirty) { MergedTable stale; if (int rc = drain_attractor_table(h, stale)) return rc; }
    {
        uint64_t want = 1ull << 16;
        while (want < 2 * (uint64_t)cap) want *= 2;
        if (h->table_slots < want) {
            HIPCHK(h, h->d_table.alloc(want));
            HIPCHK(h, hipMemset(h->d_table.p, 0, want * sizeof(LogRec)));
            h->table_slots = want;
        }
    }
    const uint64_t N = 1ull << n;
    const uint32_t cus = (uint32_t)h->prop.multiProcessorCount;
    const bool capped = max_t != BSX_T_INF;
    const uint64_t cap_rel = capped ? max_t - tp : UINT64_MAX;     // found iff mu + lambda <= max_t - T_p (S7)
    // doubling rounds: 2^rounds must reach every transient that can still be "found"; without a cap, every
    // transient (mu < N)
    uint32_t rounds = 0;
    while (rounds < n && (!capped || (1ull << rounds) <= cap_rel)) ++rounds;
    const uint64_t walk_cap = capped ? std::max<uint64_t>(cap_rel, 1) : (1ull << 22);
    const uint32_t cand_cap = 1u << 22;

    DevBuf<uint32_t>& succ = h->d_fg_a;
    DevBuf<uint32_t>& ja = h->d_fg_b;
    DevBuf<uint32_t>& jb = h->d_fg_c;
    HIPCHK(h, succ.reserve(N));
    HIPCHK(h, ja.reserve(std::max<uint64_t>(N, 1024)));         // phase D reuses ja + jb as one array of N pairs
    HIPCHK(h, jb.reserve(std::max<uint64_t>(N, 1024)));
    DevBuf<uint32_t> d_bits, d_cand;
    DevBuf<unsigned int> d_small;       // [0] candidate cursor, [1] cyclic, [2] open, [3] changed
    HIPCHK(h, d_bits.alloc((N + 31) / 32));
    HIPCHK(h, hipMemsetAsync(d_bits.p, 0, ((N + 31) / 32) * 4, h->stream));
    HIPCHK(h, d_cand.alloc(cand_cap));
    HIPCHK(h, d_small.alloc(4));
    HIPCHK(h, hipMemsetAsync(d_small.p, 0, 16, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_ctr.p, 0, sizeof(Counters), h->stream));

    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    uint32_t launches = 0;
    // A: successor array
    {
        const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus * 4, (N + kBlock - 1) / kBlock));
        HIPCHK(h, launch_fg_succ((int)h->net.k_mux, h->lut_mode, dim3((uint32_t)blocks), h->shmem, h->stream, h->net, h->sp, N, succ.p, 0));
        ++launches;
        if (tp) {
            HIPCHK(h, h->d_fg_warm.reserve(N));
            HIPCHK(h, launch_fg_succ((int)h->net.k_mux, h->lut_mode, dim3((uint32_t)blocks), h->shmem, h->stream, h->net, h->sp, N, h->d_fg_warm.p, tp));
            ++launches;
        }
    }
    // B: landing points f^(2^rounds)(s)
    const uint32_t* land = succ.p;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t* out = (r & 1) ? jb.p : ja.p;
        HIPCHK(h, launch_fg_double(land, out, N, cus, h->stream));
        land = out;
        ++launches;
    }
    // C: candidates -> cycle states
    HIPCHK(h, launch_fg_mark(land, N, d_bits.p, cus, h->stream));
    HIPCHK(h, launch_fg_collect(d_bits.p, (N + 31) / 32, d_cand.p, cand_cap, d_small.p, cus, h->stream));
    launches += 2;
    unsigned int small[4] = {0, 0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(small, d_small.p, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t n_cand = small[0];
    if (n_cand > cand_cap) return fail(h, BSX_ERR_UNSUPPORTED, "functional-graph mode: more than 2^22 distinct landing points (use the trajectory path)");
    uint32_t cyc_slots = 1024;
    while (cyc_slots < 4 * (uint64_t)n_cand) cyc_slots *= 2;
    DevBuf<unsigned char> d_cyc;
    HIPCHK(h, d_cyc.alloc((size_t)(cyc_slots + 1) * fg_cyc_entry_bytes()));
    HIPCHK(h, hipMemsetAsync(d_cyc.p, 0, (size_t)(cyc_slots + 1) * fg_cyc_entry_bytes(), h->stream));
    HIPCHK(h, launch_fg_cycles(succ.p, d_cand.p, n_cand, walk_cap, d_cyc.p, cyc_slots - 1, d_small.p + 1, d_small.p + 2, h->stream));
    ++launches;
    HIPCHK(h, hipMemcpyAsync(small, d_small.p, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!capped && small[2]) return fail(h, BSX_ERR_STEP_LIMIT, "functional-graph mode: a cycle longer than 2^22 states (no time cap given)");
    // D: (entry state, mu) by in-place pointer jumping; pairs live in ja..jb (N x 8 bytes)
    if ((const void*)(ja.p + N) != (const void*)jb.p) {
        // the two halves are separate allocations: use a dedicated pair array instead
        HIPCHK(h, h->d_fg_pair.reserve(N));
    }
    unsigned long long* pair = ((const void*)(ja.p + N) == (const void*)jb.p) ? reinterpret_cast<unsigned long long*>(ja.p) : h->d_fg_pair.p;
    HIPCHK(h, launch_fg_pair_init(succ.p, d_cyc.p, cyc_slots - 1, pair, N, cus, h->stream));
    ++launches;
    const uint32_t d_cap = capped ? (uint32_t)std::min<uint64_t>(cap_rel, 0xFFFFFFFEull) : 0xFFFFFFFEu;
    for (uint32_t r = 0; r < n + 2; ++r) {
        HIPCHK(h, hipMemsetAsync(d_small.p + 3, 0, 4, h->stream));
        HIPCHK(h, launch_fg_pair_jump(pair, N, d_cap, d_small.p + 3, cus, h->stream));
        ++launches;
        HIPCHK(h, hipMemcpyAsync(small, d_small.p, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (!small[3]) break;
    }
    // E: aggregate the requested problems
    AttractParams P{};
    P.ctr = h->d_ctr.p;
    P.table = h->d_table.p;
    P.table_mask = h->table_slots - 1;
    const uint64_t first_state = first->init_digits[0];
    HIPCHK(h, launch_fg_aggregate(pair, d_cyc.p, cyc_slots - 1, tp ? h->d_fg_warm.p : nullptr, tp, first_state, count, cap_rel, max_len,
                                  capped ? max_t : 0, P, cus, h->stream));
    ++launches;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    Counters ctr{};
    HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->table_dirty = true;
    if (ctr.table_overflow) { MergedTable junk; (void)drain_attractor_table(h, junk); return fail(h, BSX_ERR_TABLE_FULL, "more distinct attractors than the caller's table capacity"); }
    MergedTable merged;
    if (int rc = drain_attractor_table(h, merged)) return rc;
    if (merged.size() > cap) return fail(h, BSX_ERR_TABLE_FULL, "more distinct attractors than the caller's table capacity");
    uint32_t i = 0;
    for (auto& kv : merged) table[i++] = kv.second;
    *n_out = i;
    if (n_no_attractor) *n_no_attractor = ctr.n_none;
    if (stats) {
        stats->problems = count;
        stats->state_steps = ctr.steps_ref;
        stats->executed_steps = N * (1 + (uint64_t)tp);     // one network update per state of the space (+ the warm-up map)
        stats->kernel_ms = ms;
        stats->kernel_launches = launches;
        stats->total_ms = now_ms() - t_begin;
    }
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
    DevBuf<uint32_t> d_thit;
    HIPCHK(h, d_thit.alloc(count));
    Counters ctr{};
    float ms = 0.f;
    if (int rc = launch_target_pass(h, first, 0, count, max_t, mask_words, code_words, d_thit.p, nullptr, 0, ctr, ms)) return rc;
    const uint64_t total_hits = ctr.log_cursor;
    if (total_hits > cap) return fail(h, BSX_ERR_TABLE_FULL, "more hits than the caller's capacity (bsx_run_target_summary counts without listing)");
    if (int rc = compact_hits(h, d_thit.p, count, total_hits, 0, hits, cap)) return rc;
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

extern "C" int bsx_run_target_summary(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                                      const uint64_t* mask_words, const uint64_t* code_words,
                                      uint64_t* hist, uint32_t hist_bins, bsx_hit* hits, uint64_t cap,
                                      uint64_t* n_hits, uint64_t* n_listed, bsx_stats* stats) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->have_net || !h->have_space) return fail(h, BSX_ERR_STATE, "network / problem space not set");
    if (!mask_words || !code_words || !n_hits || (cap && (!hits || !n_listed)) || (hist_bins && !hist))
        return fail(h, BSX_ERR_INVALID, "null argument");
    if (hist_bins > kTargetHistBins) return fail(h, BSX_ERR_INVALID, "at most 2048 histogram bins");
    if (int rc = check_range(h, first, count)) return rc;
    if (int rc = check_max_t(h, max_t)) return rc;
    const double t_begin = now_ms();
    HIPCHK(h, hipSetDevice(h->device));
    *n_hits = 0;
    if (n_listed) *n_listed = 0;
    for (uint32_t b = 0; b < hist_bins; ++b) hist[b] = 0;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (count == 0) return BSX_OK;
    if (count > (1ull << 40)) return fail(h, BSX_ERR_INVALID, "at most 2^40 problems per call");

    DevBuf<unsigned long long> d_hist;              // (one bin even if the caller wants none: the kernels count into it)
    HIPCHK(h, d_hist.alloc(std::max<uint32_t>(hist_bins, 1)));
    HIPCHK(h, hipMemsetAsync(d_hist.p, 0, std::max<uint32_t>(hist_bins, 1) * sizeof(unsigned long long), h->stream));
    // The hit list is the first `cap` hits in index order (what -n keeps, simulate.py:163-164): problems are
    // scanned in pieces with a dense t_hit array until the list is full, the rest of the range is only counted.
    uint64_t done = 0, listed = 0, total = 0, steps_ref = 0, steps_exec = 0;
    double kernel_ms = 0.0;
    uint32_t launches = 0, limit_hits = 0;
    DevBuf<uint32_t> d_thit;
    auto plain_pass = [&](uint64_t at, uint64_t n, bool listing) -> int {
        Counters ctr{};
        float ms = 0.f;
        if (int rc = launch_target_pass(h, first, at, n, max_t, mask_words, code_words, listing ? d_thit.p : nullptr,
                                        d_hist.p, hist_bins ? hist_bins : 1, ctr, ms)) return rc;
        if (listing) {
            if (int rc = compact_hits(h, d_thit.p, n, ctr.log_cursor, at, hits + listed, cap - listed)) return rc;
            listed += std::min<uint64_t>(ctr.log_cursor, cap - listed);
        }
        total += ctr.log_cursor; steps_ref += ctr.steps_ref; steps_exec += ctr.steps_exec;
        kernel_ms += ms; ++launches; limit_hits += ctr.step_limit_hits;
        return BSX_OK;
    };
    // the listed hits first: dense pieces of the range
    while (done < count && listed < cap) {
        const uint64_t piece = std::min<uint64_t>(count - done, std::max<uint64_t>(1ull << 22, 4 * (cap - listed)));
        HIPCHK(h, d_thit.reserve(std::min<uint64_t>(piece, 1ull << 32)));
        const uint64_t n = std::min<uint64_t>(piece, d_thit.n);
        if (int rc = plain_pass(done, n, true)) return rc;
        done += n;
    }
    // the rest is only counted: cube passes over the aligned blocks of every fixed-node variant (the first update
    // of a block depends on its relevant digits only, build_cube), plain passes over what is left
    const char* cubes_env = std::getenv("BSX_CUBES");
    const bool cubes_ok = !(cubes_env && cubes_env[0] == '0') && h->sp.n_any >= kCubeMinBits && h->sp.n_any <= 64 &&
                          (h->sp.identity_any || h->sp.n_runs) && !h->sp.n_pv && !h->sp.tp_origin && !h->variant_count_saturated;
    while (done < count) {
        if (!cubes_ok) {
            const uint64_t n = std::min<uint64_t>(count - done, 1ull << 32);
            if (int rc = plain_pass(done, n, false)) return rc;
            done += n;
            continue;
        }
        // (variant, digit value) of problem first + done; the segment of the range inside this variant
        const uint32_t n_any = h->sp.n_any;
        const unsigned __int128 space = (unsigned __int128)1 << n_any;
        const unsigned __int128 pos = (unsigned __int128)first->init_digits[0] + done;
        const uint64_t variant = first->variant + (uint64_t)(pos >> n_any);
        const unsigned __int128 digit = pos & (space - 1);
        const uint64_t seg = (uint64_t)std::min<unsigned __int128>(count - done, space - digit);
        uint32_t vfm[kMaxW32], vfv[kMaxW32];
        for (int w = 0; w < kMaxW32; ++w) { vfm[w] = h->sp.fixmask[w]; vfv[w] = h->sp.fixval[w]; }
        {   // the variant's fixed nodes (batching.py:171-175, 212-229)
            uint64_t v = variant;
            for (size_t j = 0; j + 1 < h->h_fv.size(); j += 2) {
                const uint32_t node = h->h_fv[j], range = h->h_fv[j + 1];
                const uint32_t radix = range == BSX_RANGE_MAYBE_TRUE_OR_FALSE ? 3 : 2, dg = (uint32_t)(v % radix);
                v /= radix;
                int st = -1;
                if (range == BSX_RANGE_MAYBE_FALSE) st = dg ? 0 : -1;
                else if (range == BSX_RANGE_MAYBE_TRUE) st = dg ? 1 : -1;
                else if (range == BSX_RANGE_TRUE_OR_FALSE) st = dg ? 1 : 0;
                else st = dg == 0 ? -1 : (dg == 1 ? 0 : 1);
                if (st >= 0) {
                    vfm[node >> 5] |= 1u << (node & 31);
                    vfv[node >> 5] = (vfv[node >> 5] & ~(1u << (node & 31))) | ((uint32_t)st << (node & 31));
                }
            }
        }
        const unsigned __int128 unit = (unsigned __int128)1 << kCubeMinBits;
        const unsigned __int128 seg_end = digit + seg;
        unsigned __int128 at = (digit + unit - 1) / unit * unit;
        const unsigned __int128 body_end = seg_end / unit * unit;
        if (at >= body_end) { if (int rc = plain_pass(done, seg, false)) return rc; done += seg; continue; }
        if (at > digit) { if (int rc = plain_pass(done, (uint64_t)(at - digit), false)) return rc; done += (uint64_t)(at - digit); }
        while (at < body_end) {
            uint32_t a_bits = std::min<uint32_t>(kCubeMaxBits, n_any);
            while (a_bits > kCubeMinBits && ((at & (((unsigned __int128)1 << a_bits) - 1)) != 0 || at + ((unsigned __int128)1 << a_bits) > body_end)) --a_bits;
            const uint64_t block = 1ull << a_bits;
            Cube c;
            build_cube(h, (uint64_t)at, a_bits, c, vfm);
            if (c.ok && c.rel.size() + 2 <= a_bits) {
                plan_cube(h, c);
                TargetParams P{};
                P.net = h->net;
                P.sp = c.sp;
                P.sp.n_fv = 0;                              // the variant is baked into the masks
                for (int w = 0; w < kMaxW32; ++w) { P.sp.fixmask[w] = vfm[w]; P.sp.fixval[w] = vfv[w]; }
                P.count = 1ull << c.rel.size();
                P.cap_rel_inf = max_t == BSX_T_INF ? 1 : 0;
                P.max_t = max_t;
                uint32_t in_mask = 0;
                for (uint32_t w = 0; w < h->w64; ++w)
                    for (int half = 0; half < 2 && 2 * w + half < (uint32_t)kMaxW32; ++half) {
                        const uint32_t m32 = (uint32_t)(mask_words[w] >> (32 * half)), c32 = (uint32_t)(code_words[w] >> (32 * half));
                        const uint32_t idx = 2 * w + half;
                        P.tmask[idx] = m32; P.tcode[idx] = c32;
                        P.rep_mask[idx] = m32 & ~c.umask[idx];
                        P.rep_code[idx] = c32 & m32 & ~c.umask[idx];
                        in_mask += (uint32_t)__builtin_popcount(m32 & c.umask[idx]);
                    }
                P.cube = 1;
                P.cube_shift = a_bits - (uint32_t)c.rel.size();
                P.cube_t0_shift = in_mask;
                P.ctr = h->d_ctr.p;
                P.t_hit = nullptr;
                P.hist = d_hist.p;
                P.hist_bins = hist_bins ? hist_bins : 1;
                const size_t shmem = h->shmem + 16 + (size_t)P.hist_bins * 8;
                const Launch L = plan_persistent(h, P.count, shmem);
                P.chunk = L.chunk;
                Counters ctr{};
                float ms = 0.f;
                HIPCHK(h, hipMemsetAsync(h->d_ctr.p, 0, sizeof(Counters), h->stream));
                HIPCHK(h, hipEventRecord(h->ev0, h->stream));
                HIPCHK(h, launch_target((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
                HIPCHK(h, hipEventRecord(h->ev1, h->stream));
                HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
                total += ctr.log_cursor; steps_ref += ctr.steps_ref; steps_exec += ctr.steps_exec;
                kernel_ms += ms; ++launches; limit_hits += ctr.step_limit_hits;
                if (std::getenv("BSX_DEBUG")) std::fprintf(stderr, "[bsx] target cube 2^%u, variant %llu: %zu relevant digits, %.3f ms\n", a_bits, (unsigned long long)variant, c.rel.size(), ms);
            } else if (int rc = plain_pass(done, block, false)) return rc;
            done += block;
            at += block;
        }
        if (seg_end > body_end) { if (int rc = plain_pass(done, (uint64_t)(seg_end - body_end), false)) return rc; done += (uint64_t)(seg_end - body_end); }
    }
    if (hist_bins) HIPCHK(h, hipMemcpy(hist, d_hist.p, hist_bins * sizeof(uint64_t), hipMemcpyDeviceToHost));
    *n_hits = total;
    if (n_listed) *n_listed = listed;
    if (stats) {
        stats->problems = count;
        stats->state_steps = steps_ref;
        stats->executed_steps = steps_exec;
        stats->kernel_ms = kernel_ms;
        stats->kernel_launches = launches;
        stats->total_ms = now_ms() - t_begin;
    }
    if (limit_hits) return fail(h, BSX_ERR_STEP_LIMIT, "a trajectory reached the internal step limit");
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
                          uint64_t* final_states, uint64_t* digests, bsx_stats* stats) {
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
    DevBuf<uint64_t> d_final, d_dig;
    HIPCHK(h, d_desc.upload(desc));
    HIPCHK(h, d_sched.upload(h->h_sched));
    if (final_states) HIPCHK(h, d_final.alloc(count * W));
    if (digests) HIPCHK(h, d_dig.alloc(count));

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
    P.final_states = final_states ? d_final.p : nullptr;
    P.digests = digests ? d_dig.p : nullptr;
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
    // (digests: the second-generation kernel only, K <= 3 and n <= 128, which keeps them per row in registers)
    const bool gen2_shape = h->net.k_mux <= 3 && ((h->n_nodes + 15) & ~15u) <= 128 && !(sl_env && sl_env[0] == '1');
    const bool sliced_ok = !(sl_env && sl_env[0] == '0') && (final_states || digests) && !trajectories && (!digests || gen2_shape) &&
                           !h->sp.n_fv && !h->sp.n_pv && !h->net.n_wide && max_t >= 64 && max_t < kStepLimit &&
                           count >= 2048 && (size_t)((h->n_nodes + 15) & ~15u) * 136 * 4 <= 160 * 1024;
    if (sliced_ok && count) return run_sim_sliced(h, first, count, max_t, final_states, digests, stats);
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
