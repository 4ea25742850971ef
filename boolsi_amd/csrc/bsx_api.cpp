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

#include "bsx_host.h"


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
    {
        int khz = 0;        // rate of wall_clock64() on the device (100 MHz on gfx9)
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) h->wall_clock_khz = (double)khz;
    }
    if (std::strncmp(h->prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + h->prop.gcnArchName + ", this engine is built for gfx950 only";
        delete h;
        return BSX_ERR_NO_DEVICE;
    }
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess ||
        (e = hipEventCreate(&h->ev_top0)) != hipSuccess || (e = hipEventCreate(&h->ev_top1)) != hipSuccess ||
        (e = h->d_ctr_raw.alloc(kLevelDescBytes + sizeof(Counters) * kMaxChainBlocks)) != hipSuccess ||
        (e = hipHostMalloc((void**)&h->h_ctr, sizeof(Counters) * kMaxChainBlocks + 256, hipHostMallocDefault)) != hipSuccess) {
        g_create_error = std::string("stream/event creation: ") + hipGetErrorString(e);
        if (h->h_ctr) (void)hipHostFree(h->h_ctr);
        delete h;
        return BSX_ERR_HIP;
    }
    h->d_level = reinterpret_cast<LevelDesc*>(h->d_ctr_raw.p);
    h->d_ctr = reinterpret_cast<Counters*>(h->d_ctr_raw.p + kLevelDescBytes);
    h->h_flag = reinterpret_cast<volatile uint32_t*>(h->h_ctr + kMaxChainBlocks);
    *h->h_flag = 0;
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
    for (int i = 0; i < 8; ++i) {
        if (h->aux_done[i]) (void)hipEventDestroy(h->aux_done[i]);
        if (h->aux[i]) { (void)hipStreamSynchronize(h->aux[i]); (void)hipStreamDestroy(h->aux[i]); }
    }
    if (h->h_ctr_multi) (void)hipHostFree(h->h_ctr_multi);
    if (h->h_leaf) (void)hipHostFree(h->h_leaf);
    for (hipEvent_t e : h->ev_chain) (void)hipEventDestroy(e);
    for (hipStream_t st : h->side) if (st) (void)hipStreamDestroy(st);
    if (h->ev_top0) (void)hipEventDestroy(h->ev_top0);
    if (h->ev_top1) (void)hipEventDestroy(h->ev_top1);
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

extern "C" int bsx_network_info(bsx_handle h, uint32_t* state_words32, uint32_t* mux_slots, uint32_t* lut_mode) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->have_net) return fail(h, BSX_ERR_STATE, "network not set");
    if (state_words32) *state_words32 = h->net.nw;
    if (mux_slots) *mux_slots = h->net.k_mux;
    if (lut_mode) *lut_mode = (uint32_t)h->lut_mode;
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
        // entry = [slot j][word w]; 6-word entries in two planes: words 0..3 of every entry, then words 4..5 (net_step)
        const bool planar = k_mux * nw == 6 && lut_mode != 2;
        const size_t plane_b = (size_t)n_chunks * chunk_entries * 4;
        for (uint32_t j = 0; j < k; ++j) {
            const uint32_t p = pred_idx[pred_offsets[i] + j];
            const uint32_t chunk = p / chunk_bits, bit = p % chunk_bits;
            for (uint32_t v = 0; v < chunk_entries; ++v)
                if ((v >> bit) & 1u) {
                    const size_t ent = (size_t)chunk * chunk_entries + v;
                    const uint32_t word = j * nw + (i >> 5);
                    const size_t at = !planar ? ent * k_mux * nw + word : (word < 4 ? ent * 4 + word : plane_b + ent * 2 + (word - 4));
                    lut[at] |= 1u << (i & 31);
                }
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
    h->split_cache.clear();
    h->split_learned.clear();
    h->split_regrown.clear();
    std::memset(h->near_seen, 0, sizeof(h->near_seen));
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


constexpr size_t kAuxStreams = 8, kMultiCtr = 64;

// Auxiliary streams of the handle (created on first use): independent launches of one call run side by side on them.
static int ensure_aux(bsx_handle h) {
    if (h->aux[0]) return BSX_OK;
    for (size_t i = 0; i < kAuxStreams; ++i) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->aux[i], hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->aux_done[i], hipEventDisableTiming));
    }
    HIPCHK(h, h->d_ctr_multi.alloc(kMultiCtr));
    HIPCHK(h, hipHostMalloc((void**)&h->h_ctr_multi, sizeof(Counters) * kMultiCtr, hipHostMallocDefault));
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
    P.ctr = h->d_ctr;
    P.t_hit = d_thit;
    P.hist = d_hist;
    P.hist_bins = d_hist ? hist_bins : 1;
    HIPCHK(h, hipMemsetAsync(h->d_ctr, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    HIPCHK(h, launch_target((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
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
        const uint64_t piece = std::min<uint64_t>(count - done, std::max<uint64_t>(1ull << kCubeMinBits, 4 * (cap - listed)));
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
    // The cube passes of a call are independent of each other -- one per aligned block and fixed-node variant, each a
    // short launch of one-class-per-lane searches that mostly waits (profiles/r03_pmc_config4.json: 90 % of the wave
    // cycles) -- so they are enqueued side by side on the handle's auxiliary streams, every launch with its own counter
    // block, and the host waits once for all of them (config 4: eight launches of 0.34 ms each back to back before).
    struct PendingCube { TargetParams P; dim3 grid; size_t shmem; uint32_t a_bits; uint64_t variant; uint32_t rel; };
    std::vector<PendingCube> pending;
    auto flush_cubes = [&]() -> int {
        if (pending.empty()) return BSX_OK;
        const size_t n = pending.size();
        if (int rc = ensure_aux(h)) return rc;
        HIPCHK(h, hipMemsetAsync(h->d_ctr_multi.p, 0, n * sizeof(Counters), h->stream));
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        const size_t used = std::min<size_t>(n, kAuxStreams);
        for (size_t i = 0; i < used; ++i) HIPCHK(h, hipStreamWaitEvent(h->aux[i], h->ev0, 0));
        for (size_t i = 0; i < n; ++i) {
            pending[i].P.ctr = h->d_ctr_multi.p + i;
            HIPCHK(h, launch_target((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, pending[i].grid, pending[i].shmem, h->aux[i % kAuxStreams], pending[i].P));
        }
        for (size_t i = 0; i < used; ++i) {
            HIPCHK(h, hipEventRecord(h->aux_done[i], h->aux[i]));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->aux_done[i], 0));
        }
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->h_ctr_multi, h->d_ctr_multi.p, n * sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        kernel_ms += ms;
        for (size_t i = 0; i < n; ++i) {
            const Counters& ctr = h->h_ctr_multi[i];
            total += ctr.log_cursor; steps_ref += ctr.steps_ref; steps_exec += ctr.steps_exec;
            ++launches; limit_hits += ctr.step_limit_hits;
            if (std::getenv("BSX_DEBUG")) std::fprintf(stderr, "[bsx] target cube 2^%u, variant %llu: %u relevant digits (%zu launches side by side, %.3f ms together)\n", pending[i].a_bits, (unsigned long long)pending[i].variant, pending[i].rel, n, ms);
        }
        pending.clear();
        return BSX_OK;
    };
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
                P.ctr = h->d_ctr;
                P.t_hit = nullptr;
                P.hist = d_hist.p;
                P.hist_bins = hist_bins ? hist_bins : 1;
                const size_t shmem = h->shmem + 16 + (size_t)P.hist_bins * 8;
                const Launch L = plan_persistent(h, P.count, shmem);
                {   // even fixed shares, no cursor traffic (P.count <= 2^64 / ... classes of about equal cost)
                    const uint64_t n_waves = (uint64_t)L.grid.x * kWavesPerBlock;
                    if (P.count < (1ull << 28)) { P.chunk_first = ((P.count + n_waves - 1) / n_waves + 63) / 64 * 64; P.chunk = 0; }
                    else { P.chunk_first = 0; P.chunk = L.chunk; }
                }
                pending.push_back(PendingCube{P, L.grid, shmem, a_bits, variant, (uint32_t)c.rel.size()});
                if (pending.size() == kMultiCtr) if (int rc = flush_cubes()) return rc;
            } else if (int rc = plain_pass(done, block, false)) return rc;
            done += block;
            at += block;
        }
        if (seg_end > body_end) { if (int rc = plain_pass(done, (uint64_t)(seg_end - body_end), false)) return rc; done += (uint64_t)(seg_end - body_end); }
    }
    if (int rc = flush_cubes()) return rc;
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
    P.ctr = h->d_ctr;

    const uint32_t cus = (uint32_t)h->prop.multiProcessorCount;
    const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus * 4, (count + kBlock - 1) / kBlock));
    HIPCHK(h, hipMemsetAsync(h->d_ctr, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    HIPCHK(h, launch_simulate((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, dim3((uint32_t)blocks), h->shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    Counters ctr{};
    HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
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
    P.ctr = h->d_ctr;

    // K <= 3 and n <= 128: second-generation kernel (8-byte rows, constants in registers); BSX_SLICED=1 keeps the first
    const char* sl_env = std::getenv("BSX_SLICED");
    const bool gen2 = K <= 3 && rows <= 128 && !(sl_env && sl_env[0] == '1');
    const size_t shmem = gen2 ? (size_t)rows * 1024 + (4096 + 64) * 4 : (size_t)rows * (8 + 128) * 4;
    const uint64_t groups = gen2 ? (count + 4095) / 4096 : (count + 2047) / 2048;
    const uint64_t per_cu = gen2 ? 1 : std::max<size_t>(1, (160 * 1024) / shmem);
    const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(groups, (uint64_t)h->prop.multiProcessorCount * per_cu));
    HIPCHK(h, hipMemsetAsync(h->d_ctr, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if (gen2) HIPCHK(h, launch_simulate_sliced64((int)h->net.nw, (int)K, dim3((uint32_t)blocks), shmem, h->stream, P));
    else HIPCHK(h, launch_simulate_sliced((int)h->net.nw, (int)K, dim3((uint32_t)blocks), shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    Counters ctr{};
    HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
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
