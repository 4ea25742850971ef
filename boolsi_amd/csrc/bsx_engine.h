// Host-side internals shared by the translation units behind include/bsx.h (bsx_api.cpp, bsx_comm.cpp,
// bsx_fgraph.hip): the engine handle, device buffers, error plumbing.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "bsx.h"
#include "bsx_device.h"

namespace bsx {

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t reserve(size_t count) { return n >= count ? hipSuccess : alloc(count); }     // grow-only scratch
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

}  // namespace bsx

struct bsx_engine {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipDeviceProp_t prop{};
    double wall_clock_khz = 1e5;        // device clock behind wall_clock64()
    std::string error;

    // network
    bool have_net = false;
    uint32_t n_nodes = 0, w64 = 0;
    bsx::DevNet net{};
    int lut_mode = 0;           // kLutGlobal / kLutLdsByte / kLutLdsNibble (bsx_kernels_common.h)
    size_t shmem = 0;           // masks (+ LUT) : target / simulate kernels
    size_t shmem_attract = 0;   // + LDS mirror of the cycle-state cache

    // cycle-state cache (valid for the current network + origin fixed nodes)
    bool cache_enabled = true;
    int lean_blocks_per_cu = 0;  // occupancy of the lean attract kernel for the current network
    bool fast_ok = true;        // cleared when the lean kernel's straggler list overflowed for this space
    bool pool_ok = false;       // the class-pool kernel fits the LDS for this network
    size_t cache_stride = 0;    // bytes per slot of the LDS cache mirror
    std::vector<bsx::CycleRecord> h_journal;
    bool journal_stale = true;  // a general-kernel pass (the only writer of the journal) ran since h_journal was read
    uint32_t mirror_slots = 64;
    uint64_t journal_states = 0;    // cycle states of the journal records the lean / pool mirror takes
    bool cube_mirror = false;       // the next pool pass is a cube pass (mirror sized for representative entries)
    uint32_t fast_steps = 0;    // lean kernel: steps without a cached cycle state before a problem is handed over (0 = default)
    bool fast_calibrated = false;
    uint32_t cache_lds_slots = 0;
    bsx::DevBuf<bsx::CycleRecord> d_cc_journal;
    bsx::DevBuf<unsigned int> d_cc_claims;
    bsx::DevBuf<unsigned int> d_cc_count;

    // scratch kept across calls (grow-only): hipMalloc / hipFree per call cost ~1 ms of a 16 ms step
    bsx::DevBuf<bsx::LogRec> d_log;
    bsx::DevBuf<bsx::LogRec> d_table;   // HBM attractor table behind the log (all slots zero between calls)
    uint64_t table_slots = 0;
    bool table_dirty = false;
    bsx::DevBuf<uint32_t> d_strag;
    bsx::DevBuf<uint32_t> d_mirror;     // pool kernel: image of the LDS cache mirror, valid for (image_n journal records, image_slots slots)
    size_t image_n = ~size_t(0);
    uint32_t image_slots = 0;
    // deep cube passes, one set per side stream (the lower levels of up to kSideStreams chains run side by side):
    bsx::DevBuf<uint32_t> d_near_seg[bsx::kSideStreams];    // classes listed for the level below, one segment per workgroup,
    bsx::DevBuf<uint32_t> d_near_counts[bsx::kSideStreams]; // the segments' fill counts,
    bsx::DevBuf<uint32_t> d_near_list[bsx::kSideStreams];   // and the packed list the next level reads
    hipStream_t side[bsx::kSideStreams] = {};
    bsx::DevBuf<uint32_t> d_unres;      // cascade: unresolved classes per level (state, t, member count)
    bsx::DevBuf<bsx::LeafProgram> d_leaf;   // cascade: the depth-1 level's per-parent program (bsx_device.h)
    bsx::LeafProgram* h_leaf = nullptr;     // ... its pinned staging copy
    uint32_t life_cache[64] = {};       // cube passes: k_digit_lifetimes per digit, measured on the first block that needed it
    uint64_t life_valid = 0;            // (an ordering heuristic: later blocks of the problem reuse it)
    double near_seen[2][bsx::kMaxCubeLevels + 1][2] = {};   // cascade: [top level / below][depth] -> classes seen, of them near a cycle
    uint32_t cube_depth_cap = 0;        // 0 = no experience yet; else the deepest level that paid off on this problem
    bsx::DevBuf<uint32_t> d_life;       // cube collapse: per-digit influence lifetimes (ordering heuristic)
    bsx::DevBuf<uint32_t> d_lut, d_masks, d_wide_desc, d_wide_preds, d_wide_tt;

    // host copies for the bit-sliced simulate kernel's node descriptors
    std::vector<uint32_t> h_pred_offsets, h_pred_idx;
    std::vector<uint64_t> h_tt0;        // first table word of every node (all of it when k <= 6)
    std::vector<uint32_t> h_sched;      // origin perturbations (t, node, value), sorted by t
    std::vector<uint32_t> h_any;        // 'any' nodes in digit order (cube collapse: relevant-digit analysis)
    std::vector<uint32_t> h_fv;         // fixed-node variations (node, range) in digit order

    // problem space
    bool have_space = false;
    uint64_t variant_count = 1;         // product of the variation radices ...
    bool variant_count_saturated = false;   // ... unless it does not fit 64 bits
    uint32_t tp_max = 0;                // last perturbation time over origin schedule and variations
    bsx::DevSpace sp{};
    bsx::DevBuf<uint32_t> d_any, d_fv, d_pv, d_set, d_clr;

    // counters: one block for a single pass, one per level for a cube cascade, which is enqueued as a whole and read
    // back once.  The hand-over descriptors of the levels (k_compact_near -> next launch) sit right in front of block 0
    // so that one fill clears both.  h_ctr: pinned host buffer of the same shape; a cascade's last kernel (k_publish)
    // stores the blocks there and then the sequence number of the call into h_flag, which the host spins on -- the
    // wake-up of a blocking wait was a third of a 0.3 ms call.
    bsx::DevBuf<unsigned char> d_ctr_raw;
    bsx::LevelDesc* d_level = nullptr;      // [kMaxCubeLevels + 1]
    bsx::Counters* d_ctr = nullptr;         // [kMaxCubeLevels]
    bsx::Counters* h_ctr = nullptr;
    volatile uint32_t* h_flag = nullptr;    // (behind h_ctr in the same pinned allocation)
    uint32_t flag_seq = 0;
    hipEvent_t ev_top0 = nullptr, ev_top1 = nullptr;
    std::vector<bsx::Counters> ctr_seen;                // the counter blocks of the last batch of chains, as fetched
    std::vector<hipEvent_t> ev_chain;                   // pairs around the top-level (dominant) launch of every chain of a batch
    std::map<uint32_t, uint32_t> split_regrown;         // ... how often it was regrown because it did not fit a block
    std::map<uint32_t, double> split_learned;           // ... and how many classes' listing the handle had seen when it was grown (near_seen)
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> split_cache;    // block size -> leaves (fix mask, values) of its split tree
    // independent launches of one call side by side (target's cube passes): auxiliary streams, one counter block per launch
    hipStream_t aux[8] = {};
    hipEvent_t aux_done[8] = {};
    bsx::DevBuf<bsx::Counters> d_ctr_multi;
    bsx::Counters* h_ctr_multi = nullptr;

    // functional-graph mode (bsx_fgraph.hip): N-sized arrays, kept between calls (grow-only)
    bsx::DevBuf<uint32_t> d_fg_a, d_fg_b, d_fg_c, d_fg_warm;
    bsx::DevBuf<unsigned long long> d_fg_pair;

    // RCCL communicator of this handle's rank (bsx_comm.cpp); librccl is loaded on first use
    void* comm = nullptr;       // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    bsx::DevBuf<unsigned char> d_comm_send, d_comm_recv;
};

#define HIPCHK(h, call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (h)->error = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return BSX_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

static inline int fail(bsx_handle h, int status, const std::string& msg) {
    if (h) h->error = msg;
    return status;
}

