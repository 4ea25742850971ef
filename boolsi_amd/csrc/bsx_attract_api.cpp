// Host side of attract (include/bsx.h: bsx_run_attract, bsx_run_attract2, bsx_run_attract_fgraph): orchestration of
// the detector / lean / class-pool kernels, the cube analysis and the cascade of cube passes -- enqueued as one chain
// of launches whose levels hand their lists over on the device -- and the exact, wide-integer merge of the results.
// No CPU compute path exists here: every problem is resolved by gfx950 kernels (bsx_attract.hip, bsx_lean.hip,
// bsx_pool_kernel.h, bsx_fgraph.hip); the host only reads truth tables (which digits can matter) and adds up sums.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "bsx_host.h"

using namespace bsx;

namespace {

constexpr uint64_t kFastMinProblems = 8192;     // below this the general kernel alone is used
constexpr uint64_t kDiscoverySample = 65536;    // problems (sampled over the range) run through the detector when nothing is cached yet
constexpr uint64_t kLeanTile = 1ull << 28;      // problems per lean-kernel launch (straggler list: 4 B each)
constexpr uint32_t kFastSteps = 48;             // FAST phase length (steps without a cached cycle state), first guess
constexpr uint32_t kFastStepsMax = 3072;
constexpr uint64_t kProbeTile = 1ull << 22;     // lean tiles while the FAST length is being calibrated
constexpr uint64_t kGeneralTile = 1ull << 32;   // problems per launch of the general kernel (32-bit offsets)
constexpr double kLevelOverheadUs = 22.0;      // cascade: what one more level costs whatever its size (cost estimates)
constexpr uint64_t kUnresCap = 1ull << 16;      // cascade: unresolved classes a level may list
constexpr uint64_t kNearBytes = 1ull << 32;     // cascade: list of the classes a level hands to the level below (segments, and again
                                                // packed: a 2^63 block of the north star lists 7.5e7 classes of 12 bytes at its top)

using MergedTable = std::unordered_map<Key8, WideRec, Key8Hash>;

struct AttractRun {
    Counters ctr{};
    float ms = 0.f;
};

// What a call adds up (wide: a call may cover 2^128 problems).
struct Totals {
    MergedTable merged;
    u128 n_none = 0, steps_ref = 0;
    uint64_t steps_exec = 0;
    double kernel_ms = 0.0, dominant_ms = 0.0;
    uint64_t dominant_exec = 0;
    uint32_t launches = 0, dominant_launches = 0, limit_hits = 0, syncs = 0;
    double lower_ms = 0;                // cascades: the launches of the lower levels that had anything to do (device clock)
    uint64_t lower_exec = 0;
    uint32_t lower_launches = 0;
};

enum PassKind { kPassGeneral = 0, kPassLean = 1, kPassPool = 2 };

WideRec& slot_for(MergedTable& merged, const uint32_t* key32, uint32_t nw, uint64_t length) {
    const Key8 key = key8(key32);
    auto it = merged.find(key);
    if (it == merged.end()) {
        WideRec a;
        for (uint32_t w = 0; w < nw; ++w) a.key[w >> 1] |= (uint64_t)key32[w] << (32 * (w & 1));
        a.length = length;
        it = merged.emplace(key, a).first;
    }
    return it->second;
}

// merge by key (attract.py:405-455 write_aggregated_attractors_to_db, exact integers)
void merge_records(MergedTable& merged, const LogRec* recs, size_t n, uint32_t nw) {
    for (size_t i = 0; i < n; ++i) {
        const LogRec& r = recs[i];
        WideRec& a = slot_for(merged, r.key, nw, r.length);
        a.count += r.count;
        a.sum_l.add_at(r.sum_l, 0);
        a.sum_l2.add_shifted(r.sum_l2_lo, r.sum_l2_hi, 0);
    }
}

void fold_table(MergedTable& into, const MergedTable& from) {
    for (const auto& kv : from) {
        auto it = into.find(kv.first);
        if (it == into.end()) { into.emplace(kv.first, kv.second); continue; }
        WideRec& a = it->second;
        a.count += kv.second.count;
        a.sum_l.add(kv.second.sum_l);
        a.sum_l2.add(kv.second.sum_l2);
    }
}

// The sums a cube pass left in its Counters block (units of 2^shift problems + the absolute corrections of the
// members that are cycle states themselves, bsx_device.h) -> merged, exact.
void merge_cube_counters(MergedTable& merged, const Counters& c, uint32_t shift, uint32_t nw) {
    for (uint32_t a = 0; a < 64; ++a) {
        if (!c.acc_cnt[a] && !c.fix_cnt[a]) continue;
        WideRec& r = slot_for(merged, c.acc_key[a], nw, c.acc_len[a]);
        r.count += ((u128)c.acc_cnt[a] << shift) + (u128)(__int128)(int64_t)c.fix_cnt[a];
        r.sum_l.add_shifted(c.acc_sl[a], 0, shift);
        r.sum_l.add_signed((int64_t)c.fix_sl[a]);
        r.sum_l2.add_shifted(c.acc_sl2_lo[a], c.acc_sl2_hi[a], shift);
        r.sum_l2.add_signed((int64_t)c.fix_sl2[a]);
    }
}

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

int lean_mirror_slots(bsx_handle h, uint32_t* slots_out, Totals* tot = nullptr) {
    uint32_t ignored = 0;
    if (!slots_out) slots_out = &ignored;
    if (!h->journal_stale) return mirror_slots_for(h, slots_out);
    unsigned int known = 0;
    HIPCHK(h, hipMemcpy(&known, h->d_cc_count.p, sizeof(known), hipMemcpyDeviceToHost));
    if (tot) ++tot->syncs;
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

// The pool kernel's cache mirror as an image in HBM: rebuilt (one workgroup) only when the journal or the mirror
// size has changed; every workgroup of the passes that follow copies it instead of regenerating the cycles.
int ensure_mirror_image(bsx_handle h, AttractParams& P, size_t shmem) {
    if (std::getenv("BSX_MIRROR_IMAGE") && std::getenv("BSX_MIRROR_IMAGE")[0] == '0') { P.mirror_image = nullptr; P.mirror_out = nullptr; return BSX_OK; }
    const size_t words = 4 + (size_t)P.cc.lds_slots * (h->cache_stride / 4);
    if (h->image_n != h->h_journal.size() || h->image_slots != P.cc.lds_slots || h->d_mirror.n < words) {
        HIPCHK(h, h->d_mirror.reserve(words));
        AttractParams B = P;
        B.count = 0;
        B.level_in = nullptr;
        B.mirror_image = nullptr;
        B.mirror_out = h->d_mirror.p;
        HIPCHK(h, launch_attract_pool((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, dim3(1), shmem, h->stream, B));
        h->image_n = h->h_journal.size();
        h->image_slots = P.cc.lds_slots;
    }
    P.mirror_image = h->d_mirror.p;
    P.mirror_out = nullptr;
    return BSX_OK;
}

double g_prof[6];       // BSX_PROFILE: host time per section of a pass, ms

// One launch of the general / lean / pool kernel over P.count work items + merge of its log into `merged` (null:
// results discarded).  The host waits for it: these passes decide what runs next (stragglers, calibration).
int launch_attract_pass(bsx_handle h, AttractParams& P, int kind, DevBuf<LogRec>& d_log, MergedTable* merged,
                        AttractRun& run, Totals& tot) {
    const double pt0 = now_ms();
    const bool fast = kind != kPassGeneral;
    if (!fast) h->journal_stale = true;         // the detector may publish attractors
    size_t shmem = h->shmem_attract;
    if (fast) {
        uint32_t slots = h->cache_lds_slots;
        if (int rc = lean_mirror_slots(h, &slots, &tot)) return rc;
        P.cc.lds_slots = slots;
        shmem = h->shmem + (size_t)slots * h->cache_stride + 32 + (kind == kPassPool ? pool_extra_bytes(h->net.nw) : lean_acc_bytes(h->net.nw));
    }
    const Launch L = plan_persistent(h, P.count, shmem);
    P.chunk = L.chunk;
    // plain tiles, whose cost per problem varies by region: every wave starts with one piece and takes the rest from the
    // cursor (measured on config 3's plain tiles: fixed three-quarter shares 2.5 ms against 1.9 ms)
    if (kind == kPassPool) P.chunk_first = P.chunk;
    if (const char* c = std::getenv("BSX_CHUNK")) { P.chunk = (uint32_t)std::max(64, std::atoi(c)); P.chunk_first = P.chunk; }     // tuning knob
    const uint64_t waves = (uint64_t)L.grid.x * kWavesPerBlock;
    const uint64_t log_cap = waves * kTableSlots + (1u << 16);
    if (d_log.n < log_cap) HIPCHK(h, d_log.alloc(log_cap));
    P.log = d_log.p;
    P.log_cap = log_cap;
    P.ctr = h->d_ctr;
    P.level_in = nullptr;
    // results that are kept may spill from the log into the HBM attractor table (general kernel only: the
    // lean / pool kernels write at most one record per workgroup and cached attractor)
    P.table = (merged && !fast && h->table_slots) ? h->d_table.p : nullptr;
    P.table_mask = h->table_slots ? h->table_slots - 1 : 0;
    if (kind == kPassPool) if (int rc = ensure_mirror_image(h, P, shmem)) return rc;
    const double pt1 = now_ms();
    HIPCHK(h, hipMemsetAsync(h->d_ctr, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if (kind == kPassPool) HIPCHK(h, launch_attract_pool((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    else if (kind == kPassLean) HIPCHK(h, launch_attract_fast((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    else HIPCHK(h, launch_attract((int)h->net.nw, (int)h->net.k_mux, h->lut_mode, L.grid, shmem, h->stream, P));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    const double pt2 = now_ms();
    HIPCHK(h, hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    ++tot.syncs;
    run.ctr = *h->h_ctr;
    const double pt3 = now_ms();
    HIPCHK(h, hipEventElapsedTime(&run.ms, h->ev0, h->ev1));
    g_prof[0] += pt1 - pt0; g_prof[1] += pt2 - pt1; g_prof[2] += pt3 - pt2; g_prof[3] += run.ms;
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
    const uint64_t n_log = std::min<uint64_t>(run.ctr.log_cursor, log_cap);
    std::vector<LogRec> log(n_log);
    if (n_log) { HIPCHK(h, hipMemcpy(log.data(), d_log.p, n_log * sizeof(LogRec), hipMemcpyDeviceToHost)); ++tot.syncs; }
    merge_records(*merged, log.data(), log.size(), h->net.nw);
    return BSX_OK;
}

}  // namespace

namespace bsx {

// ---- cube collapse: which of the `a` lowest initial-state digits can the FIRST update of the
// block starting at digit value d_lo depend on?  A node's rule, restricted to the block's fixed bits, depends
// on a free predecessor iff flipping it changes the output for some assignment of the rule's other free
// inputs; a digit is relevant iff its node is such a predecessor of some node (fixed nodes have constant
// rules, model.py:45-47).  f(s) is then a function of the relevant digits alone -- exactly, not heuristically.
void build_cube(const bsx_engine* h, uint64_t d_lo, uint32_t a, Cube& c, const uint32_t* fixmask, uint64_t fix_mask, uint64_t fix_vals) {
    if (!fixmask) fixmask = h->sp.fixmask;      // (target passes: the fixed nodes of the block's fixed-node variant)
    const uint32_t n = h->n_nodes, nw = h->net.nw;
    c.d_lo = d_lo; c.a = a; c.rel.clear(); c.ok = false;
    const uint64_t low = a >= 64 ? ~0ull : (1ull << a) - 1ull;
    c.fix_mask = fix_mask & low; c.fix_vals = fix_vals & c.fix_mask;
    c.free_digits = low & ~c.fix_mask;
    c.n_free = (uint32_t)__builtin_popcountll(c.free_digits);
    uint32_t base[kMaxW32];     // origin bits + the block's fixed digits
    for (int w = 0; w < kMaxW32; ++w) { base[w] = h->sp.origin[w]; c.umask[w] = 0; c.free_mask[w] = 0; }
    std::vector<char> is_free(n, 0), relevant(n, 0);
    for (uint32_t j = 0; j < h->sp.n_any; ++j) {
        const uint32_t node = h->h_any[j];
        if (j < a && ((c.free_digits >> j) & 1ull)) { is_free[node] = 1; c.free_mask[node >> 5] |= 1u << (node & 31); }
        else if (j < a ? ((c.fix_vals >> j) & 1ull) != 0 : ((d_lo >> j) & 1ull) != 0) base[node >> 5] |= 1u << (node & 31);
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
        if (!((c.free_digits >> j) & 1ull)) continue;
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

}  // namespace bsx

namespace {

// Deeper collapse: the digits of the block that F^d(x) still depends on, d = 1 .. max_depth, as masks over the
// digit index (out[d - 1]; a <= 63).  Constant propagation over the block: a node's value after s updates is
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
    for (uint32_t j = 0; j < c.a; ++j)
        if ((c.free_digits >> j) & 1ull) { const uint32_t node = h->h_any[j]; val[node] = 2; dep[node] = 1ull << j; }
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
    // (measured once per digit and problem space: the launch + copy + wait would otherwise sit inside every call)
    if (need & ~h->life_valid) {
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

int ensure_attractor_table(bsx_handle h, uint32_t cap) {
    if (h->table_dirty) { MergedTable stale; if (int rc = drain_attractor_table(h, stale)) return rc; }     // a failed call left entries behind
    // HBM attractor table behind the log: two slots per entry of the caller's table (kept zeroed between calls)
    uint64_t want = 1ull << 16;
    while (want < 2 * (uint64_t)cap) want *= 2;
    if (h->table_slots < want) {
        HIPCHK(h, h->d_table.alloc(want));
        HIPCHK(h, hipMemset(h->d_table.p, 0, want * sizeof(LogRec)));
        h->table_slots = want;
        h->table_dirty = false;
    }
    return BSX_OK;
}

// first + delta for spaces whose initial-state digits fit one word (the fast path's precondition)
void advance_first(DevSpace& sp, const bsx_index& first, uint64_t delta) {
    for (int w = 0; w < 4; ++w) sp.first_digits[w] = first.init_digits[w];
    sp.first_digits[0] += delta;
    sp.first_variant = first.variant;
}

// The counter blocks of a finished chain -> h->h_ctr, and the one wait of the chain.  k_publish, the chain's last
// kernel, stores the blocks into the pinned host buffer and then the call's sequence number into h->h_flag; the host
// spins on that word (asking the stream now and then whether it has failed) instead of sleeping in
// hipStreamSynchronize behind a DMA copy, whose wake-up cost tens of microseconds per call.  BSX_SPIN_WAIT=0: the
// plain copy + wait.
int fetch_counters(bsx_handle h, uint32_t n_blocks) {
    const char* spin_env = std::getenv("BSX_SPIN_WAIT");
    const bool spin = !(spin_env && spin_env[0] == '0');
    if (!spin) {
        HIPCHK(h, hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters) * n_blocks, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return BSX_OK;
    }
    const uint32_t seq = ++h->flag_seq ? h->flag_seq : ++h->flag_seq;       // never 0
    HIPCHK(h, launch_publish(reinterpret_cast<const uint32_t*>(h->d_ctr), reinterpret_cast<uint32_t*>(h->h_ctr),
                             (uint32_t)(sizeof(Counters) / 4 * n_blocks), const_cast<uint32_t*>(h->h_flag), seq,
                             reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(h->d_level) + kPublishTicketOffset), h->stream));
    uint32_t polls = 0;
    while (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) != seq) {
        __builtin_ia32_pause();
        if ((++polls & 0xFFFFu) == 0) {                     // every few hundred microseconds: is the stream still alive?
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) {                          // drained; the flag store is visible by now, or never will be
                if (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) == seq) break;
                return fail(h, BSX_ERR_HIP, "the cascade finished without publishing its counters");
            }
            if (q != hipErrorNotReady) { h->error = std::string("hipStreamQuery: ") + hipGetErrorString(q); return BSX_ERR_HIP; }
        }
    }
    return BSX_OK;
}

// The depth-1 level's program (bsx_device.h: LeafProgram) for the block `c1` and the digits `added` that level adds: which
// nodes' rules read an added digit, with which inputs.  False if the level does not qualify (too many digits or dependent
// nodes, a dependent rule with more than kLeafMaxK inputs): the per-child pass takes it then.
bool build_leaf_program(const bsx_engine* h, const std::vector<uint32_t>& added_digits, LeafProgram& L) {
    const uint32_t n = h->n_nodes;
    if (added_digits.empty() || added_digits.size() > kLeafMaxBits) return false;
    std::memset(&L, 0, sizeof(L));
    L.kb = (uint32_t)added_digits.size();
    std::vector<int> digit_of(n, -1);
    for (uint32_t q = 0; q < L.kb; ++q) {
        const uint32_t node = h->h_any[added_digits[q]];
        digit_of[node] = (int)q;
        L.added[node >> 5] |= 1u << (node & 31);
    }
    for (uint32_t i = 0; i < n; ++i) {
        bool dependent = false;
        const uint32_t k = h->h_pred_offsets[i + 1] - h->h_pred_offsets[i];
        const uint32_t* preds = h->h_pred_idx.data() + h->h_pred_offsets[i];
        if (!((h->sp.fixmask[i >> 5] >> (i & 31)) & 1u))                 // (a fixed node's rule is a constant, model.py:45-47)
            for (uint32_t j = 0; j < k; ++j) dependent = dependent || digit_of[preds[j]] >= 0;
        if (!dependent) { L.indep[i >> 5] |= 1u << (i & 31); continue; }
        if (k > kLeafMaxK || L.n_dep == kLeafMaxDeps) return false;
        LeafDep& d = L.dep[L.n_dep++];
        d.node = (uint16_t)i;
        d.k = (uint16_t)k;
        for (uint32_t j = 0; j < k; ++j) d.in[j] = digit_of[preds[j]] >= 0 ? (uint16_t)(0x8000u | (uint32_t)digit_of[preds[j]]) : (uint16_t)preds[j];
        d.tt = (uint32_t)(h->h_tt0[i] & ((1ull << (1u << k)) - 1ull));   // inputs beyond k: their selectors are 0 (the low half)
    }
    return true;
}

struct CascadeEnv {
    const AttractParams& P;         // the call's template (network, caps, cache)
    uint64_t max_t, max_len;
    Totals& tot;
    DevBuf<LogRec>& d_log;
};

// What every cascade of a call shares: the time caps in the units the kernels count in, the FAST length, the deepest level.
struct CascadeShape {
    uint64_t tp, cap_rel;
    uint32_t cap_rel32, fast_steps, max_depth;
    bool forced_depth;
};

CascadeShape cascade_shape(bsx_handle h, const CascadeEnv& env) {
    CascadeShape sh{};
    sh.tp = h->sp.tp_origin;            // the search starts at s(T_p); class times count from there
    sh.cap_rel = env.max_t == BSX_T_INF ? BSX_T_INF : env.max_t - sh.tp;
    sh.cap_rel32 = (sh.cap_rel == BSX_T_INF || sh.cap_rel >= (kStepLimit / 4)) ? 0xFFFFFFFFu : (uint32_t)sh.cap_rel;
    sh.fast_steps = (uint32_t)std::min<uint64_t>((uint64_t)sh.cap_rel32 + 1, std::min<uint32_t>(kFastStepsMax, std::max(192u, 4 * h->fast_steps)));
    // BSX_CUBE_DEPTH caps the top level (1 = first update only)
    uint32_t max_depth = 8;
    if (const char* e = std::getenv("BSX_CUBE_DEPTH")) max_depth = (uint32_t)std::max(1, std::min((int)kMaxCubeLevels, std::atoi(e)));
    if (h->cube_depth_cap) max_depth = std::min(max_depth, h->cube_depth_cap);
    // with a warm-up the search starts at s(T_p): classes that share F^d, d <= T_p, share every state that counts,
    // so no class has to be handed down -- one pass at the best such depth
    if (sh.tp) max_depth = (uint32_t)std::min<uint64_t>(max_depth, sh.tp);
    sh.max_depth = std::max(1u, std::min(max_depth, sh.fast_steps > 1 ? sh.fast_steps - 1 : 1u));
    // (an explicit BSX_CUBE_DEPTH keeps the plain rule "fewest digits": tests force levels onto small spaces with it)
    sh.forced_depth = std::getenv("BSX_CUBE_DEPTH") != nullptr;
    return sh;
}

// Estimated device time of a cascade in microseconds, so that a block is not pushed through levels that cost more than they
// save and so that sub-blocks can be compared (plan_split).  rel_mask[d - 1] = digits F^d depends on.  The top level `top`
// enumerates 2^r_top classes at (top + 0.3) updates each; of the classes of level d + 1 the fraction f(d + 1) is listed, and
// each listed class has 2^(r_d - r_(d + 1)) children at level d.  Rates as measured on the north star (profiles/r03_levels.md):
// 4.5e11 class updates per second in the per-child passes, 1.5e12 children per second in the depth-1 level per parent,
// kLevelOverheadUs for every level that has anything to do.  (A least-squares fit over 107 per-parent levels says 95 us +
// 1.75e12 children/s; pricing that latency in made the trees worse -- more, smaller chains -- on every block size tried.)  f is what this handle has seen at that depth so far
// (bsx_engine::near_seen; the top level and the levels below it apart: children of listed classes are far more often near a
// cycle than classes at large), else a guess that grows with the depth.
double near_fraction(const bsx_engine* h, uint32_t d, bool is_top) {
    d = std::min<uint32_t>(d, kMaxCubeLevels);
    const auto& seen = h->near_seen[is_top ? 0 : 1];
    if (seen[d][0] >= 1024.0) return std::min(1.0, seen[d][1] / seen[d][0]);
    // nothing seen at this depth: the nearest depth that has been, a factor of two per level (deeper = nearer to the cycles)
    for (uint32_t off = 1; off <= kMaxCubeLevels; ++off) {
        if (d > off && seen[d - off][0] >= 1024.0) return std::min(1.0, seen[d - off][1] / seen[d - off][0] * std::ldexp(1.0, (int)off));
        if (d + off <= kMaxCubeLevels && seen[d + off][0] >= 1024.0) return std::min(1.0, seen[d + off][1] / seen[d + off][0] * std::ldexp(1.0, -(int)off));
    }
    return is_top ? std::min(1.0, 0.0025 * std::ldexp(1.0, (int)d - 2)) : 0.1;
}

double chain_cost_us(const bsx_engine* h, const std::vector<uint64_t>& rel_mask, uint32_t top) {
    int r_above = __builtin_popcountll(rel_mask[top - 1]);
    double n = std::ldexp(1.0, r_above);
    double cost = kLevelOverheadUs + n * (top + 0.3) / 4.5e5;
    for (uint32_t d = top - 1; d >= 1; --d) {
        const double parents = n * near_fraction(h, d + 1, d + 1 == top);
        if (parents < 1.0) { cost += 5.0 * d; break; }                  // (launches that find an empty list)
        const int r_d = __builtin_popcountll(rel_mask[d - 1]), kb = r_d - r_above;
        n = parents * std::ldexp(1.0, kb);
        cost += kLevelOverheadUs + ((d == 1 && kb >= 1 && kb <= (int)kLeafMaxBits) ? n / 1.5e6 + parents / 2.0e4 : n * (d + 0.3) / 4.5e5);
        r_above = r_d;
    }
    return cost;
}

// -> the top level (depth) that minimises the estimate, and the estimate
uint32_t choose_top(const bsx_engine* h, const CascadeShape& sh, const std::vector<uint64_t>& rel_mask, uint32_t max_depth, double* est_out = nullptr) {
    uint32_t top = 1;
    double best = 0;
    for (uint32_t d = 1; d <= max_depth && d <= rel_mask.size(); ++d) {
        const double est = sh.forced_depth ? (double)__builtin_popcountll(rel_mask[d - 1]) : chain_cost_us(h, rel_mask, d);
        if (d == 1 || est < best) { best = est; top = d; }
    }
    if (est_out) *est_out = sh.forced_depth ? chain_cost_us(h, rel_mask, top) : best;
    return top;
}

// ---- one cube: the whole cascade as ONE chain of launches -----------------------------------------------------------
// Level d of a block enumerates the assignments of the digits F^d still depends on (top level) or, below it, the digits
// level d adds on top of every class the level above has listed as "near a cycle" (DESIGN.md "Deeper collapse").  How many
// classes a level lists is only known on the device, so the chain is enqueued blind: k_compact_near packs the list and
// writes its length into a LevelDesc, the next level's launch (full persistent grid) reads it there and sizes its own
// work split.  Every level counts into its own Counters block; the host waits once -- for one chain, or for the chains of
// all sub-blocks of a split block -- reads the blocks, and only then looks at what happened: a segment overflow (-> the
// cube is redone from a shallower top), unresolved classes (attractors nobody has cached yet -> the detector runs from the
// listed states; if one of them sat on a cycle the cube is repeated with the richer cache).  Passes are accepted or
// discarded whole.
struct ChainLevel {
    uint32_t depth = 0, k_bits = 0, r_here = 0, unit_shift = 0;
    bool per_parent = false;        // depth 1, evaluated per listed class (LeafProgram) instead of per child
    Cube cube;
};
struct Chain {
    Cube c1;
    std::vector<uint64_t> rel_mask;
    std::vector<ChainLevel> lv;     // index 0 = top (depth `top`) .. top - 1 (depth 1); empty = not eligible
    uint32_t top = 1;
    uint32_t ctr_base = 0;          // its levels count into counter blocks ctr_base .. ctr_base + top - 1
    uint32_t desc_base = 0;         // ... and hand over through descriptors desc_base .. desc_base + top
    uint32_t index = 0;             // which chain of the batch (leaf program, events)
    dim3 top_grid;
};
enum ChainVerdict { kChainOk = 0, kChainLower = 1, kChainRepeat = 2, kChainGiveUp = 3 };

// Levels of the cascade for cube c1 from the top `top` (0: chosen by the estimate).  ch.lv stays empty if the cube does not
// qualify (more classes at the top than the 49-bit member counts, in units of one fresh class, can add up).
int plan_chain(bsx_handle h, const CascadeShape& sh, const Cube& c1, uint32_t top, Chain& ch) {
    ch.c1 = c1;
    ch.lv.clear();
    cube_levels(h, c1, sh.max_depth, ch.rel_mask);
    ch.top = top ? std::min<uint32_t>(top, (uint32_t)ch.rel_mask.size()) : choose_top(h, sh, ch.rel_mask, sh.max_depth);
    if (__builtin_popcountll(ch.rel_mask[ch.top - 1]) > 47) return BSX_OK;
    ch.lv.resize(ch.top);
    for (uint32_t i = 0; i < ch.top; ++i) {
        const uint32_t d = ch.top - i;
        const uint64_t here = ch.rel_mask[d - 1], digits = i == 0 ? here : here & ~ch.rel_mask[d];
        ChainLevel& l = ch.lv[i];
        l.depth = d;
        l.r_here = (uint32_t)__builtin_popcountll(here);
        l.cube = c1;
        l.cube.rel.clear();
        for (uint32_t j = 0; j < c1.a; ++j) if ((digits >> j) & 1ull) l.cube.rel.push_back(j);
        if (i == 0) if (int rc = order_cube_digits(h, l.cube)) return rc;
        plan_cube(h, l.cube);
        l.k_bits = (uint32_t)l.cube.rel.size();
        l.unit_shift = c1.n_free - l.r_here;            // members of one fresh class = the unit of this level's counts
    }
    return BSX_OK;
}

// What a batch of chains shares on the device: mirror size, grid, segment size, buffers.
struct ChainBatch {
    uint32_t slots = 0;
    size_t shmem = 0;
    Launch full{};
    uint64_t seg_cap = 0;
    uint32_t n_side = 0;            // sets of list buffers in use; > 1: lower levels on that many side streams
};

// Mirror check + buffers for a batch of chains.  ok = false: the cached attractors do not fit the mirror (no cubes then).
int prepare_batch(bsx_handle h, const CascadeEnv& env, const std::vector<Chain*>& chains, ChainBatch& B, bool& ok) {
    ok = false;
    const uint32_t nw = h->net.nw, rec_words = nw + 3;
    // every cached attractor must be in the mirror, or a class could sit on a cycle nobody recognises
    h->cube_mirror = true;
    const int rc_m = lean_mirror_slots(h, &B.slots, &env.tot);
    h->cube_mirror = false;
    if (rc_m) return rc_m;
    uint64_t states = 0;
    for (const CycleRecord& jr : h->h_journal) states += jr.length;
    if (h->h_journal.size() > (size_t)kTagAcc + kLdsAcc || 4 * states > h->cache_lds_slots) return BSX_OK;
    B.shmem = h->shmem + (size_t)B.slots * h->cache_stride + 32 + pool_extra_bytes(nw);
    B.full = plan_persistent(h, ~0ull >> 8, B.shmem);           // the persistent grid (lower levels: size unknown here)
    // classes a level may hand down: as many as the largest top level has (a level that lists more than that is not worth its
    // launch: the cube is redone shallower), at most what 4 GiB hold; split evenly over the workgroups' segments
    uint32_t top_bits = 16, blocks = 0;
    bool lists = false;
    for (const Chain* ch : chains) {
        if (ch->lv.empty()) continue;
        top_bits = std::max(top_bits, ch->lv[0].k_bits);
        lists = lists || ch->top > 1;
        blocks += ch->top;
    }
    const uint64_t list_cap = std::min<uint64_t>(kNearBytes / (4 * (nw + 1)), 1ull << top_bits);
    B.seg_cap = std::getenv("BSX_CUBE_NEAR_CAP") ? (uint64_t)std::max(1, std::atoi(std::getenv("BSX_CUBE_NEAR_CAP")))     // (tests: force the shallower restart)
                                                 : std::max<uint64_t>(64, list_cap / B.full.grid.x);
    // the lower levels of consecutive chains run on side streams (BSX_CUBE_STREAMS=1: everything on the handle's stream)
    uint32_t n_lists = 0;
    for (const Chain* ch : chains) n_lists += (!ch->lv.empty() && ch->top > 1) ? 1u : 0u;
    const char* st_env = std::getenv("BSX_CUBE_STREAMS");
    B.n_side = std::min<uint32_t>(n_lists, (uint32_t)std::max(1, std::min((int)kSideStreams, st_env ? std::atoi(st_env) : (int)kSideStreams)));
    if (B.n_side < 2) B.n_side = lists ? 1 : 0;
    for (uint32_t sl = 0; sl < B.n_side; ++sl) {
        HIPCHK(h, h->d_near_seg[sl].reserve((size_t)B.full.grid.x * B.seg_cap * (nw + 1)));      // (state + the tag of its cycle)
        HIPCHK(h, h->d_near_counts[sl].reserve(B.full.grid.x));
        HIPCHK(h, h->d_near_list[sl].reserve((size_t)B.full.grid.x * B.seg_cap * (nw + 1)));
        if (B.n_side > 1 && !h->side[sl]) HIPCHK(h, hipStreamCreateWithFlags(&h->side[sl], hipStreamNonBlocking));
    }
    HIPCHK(h, h->d_unres.reserve((size_t)std::max(blocks, 1u) * kUnresCap * rec_words));
    ok = true;
    return BSX_OK;
}

// The launches of one chain, enqueued on the handle's stream (nothing is waited for).
// With side streams (B.n_side > 1) only the top level and the packing of its list run on the handle's stream; the lower levels
// -- short launches that mostly wait on memory -- follow on side stream `slot`, next to the following chains' top levels.
// slot_busy[slot] = the event behind the last chain that used the slot's list buffers.
int enqueue_chain(bsx_handle h, const CascadeEnv& env, const CascadeShape& sh, const ChainBatch& B, Chain& ch, uint32_t slot,
                  std::vector<hipEvent_t>& slot_busy) {
    const uint32_t nw = h->net.nw, rec_words = nw + 3;
    const bool side = B.n_side > 1 && ch.top > 1;
    hipStream_t const main_st = h->stream, tail_st = side ? h->side[slot] : h->stream;
    hipEvent_t* const ev = h->ev_chain.data() + 4 * (size_t)ch.index;       // top in, top out, hand-over, chain done
    if (side && slot_busy[slot]) HIPCHK(h, hipStreamWaitEvent(main_st, slot_busy[slot], 0));   // (the buffers' previous user has finished)
    AttractParams Q0 = env.P;
    Q0.cc.lds_slots = B.slots;
    Q0.merge = 3;
    Q0.fast_steps = sh.fast_steps;
    Q0.per_problem = nullptr;
    Q0.offsets = nullptr;
    Q0.states = nullptr;
    Q0.log = nullptr; Q0.log_cap = 0; Q0.table = nullptr; Q0.table_mask = 0;
    for (int w = 0; w < kMaxW32; ++w) { Q0.cube_umask[w] = ch.c1.umask[w]; Q0.cube_free[w] = ch.c1.free_mask[w]; }
    if (int rc = ensure_mirror_image(h, Q0, B.shmem)) return rc;
    for (uint32_t i = 0; i < ch.top; ++i) {
        ChainLevel& l = ch.lv[i];
        AttractParams Q = Q0;
        Q.sp = l.cube.sp;
        Q.ctr = h->d_ctr + ch.ctr_base + i;
        Q.cube_shift = 0;                               // counts in units of one fresh class (2^unit_shift problems)
        Q.cube_depth = l.depth;
        Q.entry_shift = l.k_bits;
        Q.stragglers = h->d_unres.p + (size_t)(ch.ctr_base + i) * kUnresCap * rec_words;
        Q.stragglers_cap = kUnresCap * rec_words;
        Q.near = l.depth > 1 ? h->d_near_seg[slot].p : nullptr;
        Q.near_counts = l.depth > 1 ? h->d_near_counts[slot].p : nullptr;
        Q.near_cap = l.depth > 1 ? B.seg_cap : 0;
        dim3 grid = B.full.grid;
        if (i == 0) {
            Q.count = 1ull << l.k_bits;
            Q.entries = nullptr;
            Q.level_in = nullptr;
            const Launch L = plan_persistent(h, Q.count, B.shmem);
            grid = L.grid;
            const uint64_t n_waves = (uint64_t)grid.x * (kPoolBlockThreads / 64);
            // passes under 2^28 classes: even fixed shares, no traffic on the cursor's one address (their classes
            // cost about the same everywhere); larger ones: one piece each, the rest from the cursor
            if (Q.count < (1ull << 28)) { Q.chunk_first = ((Q.count + n_waves - 1) / n_waves + 63) / 64 * 64; Q.chunk = 0; }
            else { Q.chunk_first = L.chunk; Q.chunk = L.chunk; }
            if (const char* c = std::getenv("BSX_CHUNK")) { Q.chunk = (uint32_t)std::max(64, std::atoi(c)); Q.chunk_first = Q.chunk; }
            ch.top_grid = grid;
            HIPCHK(h, hipEventRecord(ev[0], main_st));
        } else {
            Q.count = 0;
            Q.entries = h->d_near_list[slot].p;         // (packed by the k_compact_near before this launch)
            Q.level_in = h->d_level + ch.desc_base + i;
            Q.chunk = 0; Q.chunk_first = 0;
            // the lower-level build of the kernel: no pool, no rings (its LDS is the tables alone)
            Q.lower_build = (Q.mirror_image && !(std::getenv("BSX_CUBE_LOWER") && std::getenv("BSX_CUBE_LOWER")[0] == '0')) ? 1u : 0u;
            // ... and at depth 1, where it qualifies, per parent instead of per child (BSX_CUBE_LEAF=0: per child)
            if (Q.lower_build && l.depth == 1 && !(std::getenv("BSX_CUBE_LEAF") && std::getenv("BSX_CUBE_LEAF")[0] == '0')) {
                LeafProgram& prog = h->h_leaf[ch.index];
                if (build_leaf_program(h, l.cube.rel, prog)) {
                    HIPCHK(h, hipMemcpyAsync(h->d_leaf.p + ch.index, &prog, sizeof(LeafProgram), hipMemcpyHostToDevice, tail_st));
                    Q.leaf = h->d_leaf.p + ch.index;
                    Q.entry_shift = 0;                  // work items = the listed entries themselves
                    l.per_parent = true;
                }
            }
        }
        const size_t shmem_here = Q.lower_build ? h->shmem + (size_t)B.slots * h->cache_stride + 32 + pool_lower_extra_bytes(nw) : B.shmem;
        hipStream_t const st = i == 0 ? main_st : tail_st;
        HIPCHK(h, launch_attract_pool((int)nw, (int)h->net.k_mux, h->lut_mode, grid, shmem_here, st, Q));
        if (i == 0) HIPCHK(h, hipEventRecord(ev[1], main_st));
        if (l.depth > 1)
            HIPCHK(h, launch_compact_near(h->d_near_seg[slot].p, h->d_near_counts[slot].p, grid.x, B.seg_cap, nw + 1, h->d_near_list[slot].p,
                                          h->d_level + ch.desc_base + i + 1, st));
        if (i == 0 && side) {                           // the rest of the chain: on the side stream, behind the list
            HIPCHK(h, hipEventRecord(ev[2], main_st));
            HIPCHK(h, hipStreamWaitEvent(tail_st, ev[2], 0));
        }
    }
    if (side) {
        HIPCHK(h, hipEventRecord(ev[3], tail_st));
        slot_busy[slot] = ev[3];
    }
    return BSX_OK;
}

// What a finished chain's counter blocks (in h->h_ctr) say: its sums into pass_* (only meaningful for kChainOk).
int evaluate_chain(bsx_handle h, const CascadeEnv& env, const CascadeShape& sh, const Chain& ch, MergedTable& pass_table,
                   u128& pass_none, u128& pass_ref, int& verdict, uint32_t& lower_to) {
    const AttractParams& P = env.P;
    Totals& tot = env.tot;
    const uint64_t max_t = env.max_t, max_len = env.max_len;
    const uint32_t nw = h->net.nw, rec_words = nw + 3;
    verdict = kChainOk;
    uint64_t n_entries = 0;
    for (uint32_t i = 0; i < ch.top; ++i) {
        const ChainLevel& l = ch.lv[i];
        const Counters& c = h->ctr_seen[ch.ctr_base + i];  // (the batch's blocks as fetched: the detector pass below reuses h_ctr's first)
        const uint64_t classes = i == 0 ? 1ull << l.k_bits : n_entries << l.k_bits;
        if (i > 0 && n_entries == 0) break;
        tot.steps_exec += c.steps_exec;
        if (std::getenv("BSX_DEBUG"))
            std::fprintf(stderr, "[bsx] cube 2^%u (%u digits free) at digit value %llu%s: depth %u%s, %u digits here (%u relevant), %llu classes, %llu near a cycle, %llu unresolved\n",
                         ch.c1.a, ch.c1.n_free, (unsigned long long)ch.c1.d_lo, ch.c1.fix_mask ? " [sub-block]" : "", l.depth, i == 0 ? " (top)" : l.per_parent ? " (per parent)" : "", l.k_bits, l.r_here,
                         (unsigned long long)classes, (unsigned long long)c.near_classes, (unsigned long long)c.straggler_classes);
        if (c.straggler_overflow) { verdict = kChainGiveUp; return BSX_OK; }      // too many unresolved classes: not a space for cubes
        if (c.near_overflow) { verdict = kChainLower; lower_to = l.depth - 1; return BSX_OK; }     // start over, shallower
        // (how many classes a level lists feeds the estimate that chooses later chains' tops: near_seen below)
        n_entries = c.near_classes;
        {
            double* seen = h->near_seen[i == 0 ? 0 : 1][std::min<uint32_t>(l.depth, kMaxCubeLevels)];
            seen[0] += (double)classes; seen[1] += (double)c.near_classes;
        }
        const uint32_t us = l.unit_shift;
        merge_cube_counters(pass_table, c, us, nw);
        pass_none += ((u128)c.n_none << us) + (u128)(__int128)(int64_t)c.fix_none;
        pass_ref += ((u128)c.steps_ref << us) + (u128)(__int128)(int64_t)c.fix_ref +
                    (max_t == BSX_T_INF ? (u128)0 : (u128)((__int128)(int64_t)c.fix_capfail * (__int128)max_t));
        const uint64_t n_unres = c.straggler_classes;
        if (!n_unres) continue;
        // the detector runs from each listed state: a class that was not on a cycle yet gets its exact
        // result (all members share the rest of the trajectory); one that sits on a cycle needs that
        // attractor in the cache -- the detector has just published it -- and the pass is repeated
        if (n_unres > kUnresCap) { verdict = kChainGiveUp; return BSX_OK; }
        std::vector<uint32_t> recs(n_unres * rec_words);
        HIPCHK(h, hipMemcpy(recs.data(), h->d_unres.p + (size_t)(ch.ctr_base + i) * kUnresCap * rec_words, recs.size() * 4, hipMemcpyDeviceToHost));
        ++tot.syncs;
        std::vector<uint32_t> st(n_unres * nw);
        for (uint64_t q = 0; q < n_unres; ++q) std::copy(recs.begin() + q * rec_words, recs.begin() + q * rec_words + nw, st.begin() + q * nw);
        DevBuf<uint32_t> d_states;
        DevBuf<ProblemRec32> d_res;
        HIPCHK(h, d_states.upload(st));
        HIPCHK(h, d_res.alloc(n_unres));
        AttractParams S = P;
        S.sp = l.cube.sp;
        S.sp.tp_origin = 0;                     // the listed states are past the warm-up
        S.count = n_unres;
        S.states = d_states.p;
        S.per_problem = d_res.p;
        S.max_len = BSX_T_INF;
        S.merge = 0;
        AttractRun rs;
        if (int rc2 = launch_attract_pass(h, S, kPassGeneral, env.d_log, nullptr, rs, tot)) return rc2;
        tot.kernel_ms += rs.ms; ++tot.launches; tot.steps_exec += rs.ctr.steps_exec; tot.limit_hits += rs.ctr.step_limit_hits;
        std::vector<ProblemRec32> res(n_unres);
        HIPCHK(h, hipMemcpy(res.data(), d_res.p, n_unres * sizeof(ProblemRec32), hipMemcpyDeviceToHost));
        ++tot.syncs;
        for (uint64_t q = 0; q < n_unres; ++q) {
            const uint32_t* rec = recs.data() + q * rec_words;
            const uint64_t t_class = rec[nw];
            const u128 m = (u128)(((uint64_t)rec[nw + 2] << 32) | rec[nw + 1]) << us;
            const ProblemRec32& pr = res[q];
            if (!pr.found) { pass_none += m; pass_ref += m * max_t; continue; }        // (finite cap, or the step limit was hit)
            if (pr.trajectory_l == 0) { verdict = kChainRepeat; return BSX_OK; }        // on a cycle: members' mu unknown
            const uint64_t mu = t_class + pr.trajectory_l, lam = pr.length, traj = sh.tp + mu;
            const bool found = sh.cap_rel == BSX_T_INF || mu + lam <= sh.cap_rel;
            pass_ref += found ? m * (traj + lam) : m * max_t;
            if (!found || lam > max_len) { pass_none += m; continue; }
            WideRec& e = slot_for(pass_table, pr.key, nw, lam);
            e.count += m;
            e.sum_l.add_mul(m, traj);
            e.sum_l2.add_mul(m, traj * traj);               // traj < 2^31 here (32-bit device counters)
        }
    }
    return BSX_OK;
}

// Enqueue the chains (counter blocks and descriptors laid out one after the other), wait once, account the device time.
int run_batch(bsx_handle h, const CascadeEnv& env, const CascadeShape& sh, const ChainBatch& B, std::vector<Chain*>& chains) {
    Totals& tot = env.tot;
    uint32_t blocks = 0, n_live = 0;
    for (Chain* ch : chains) {
        if (ch->lv.empty()) continue;
        ch->ctr_base = blocks;
        ch->desc_base = blocks + n_live;
        ch->index = n_live++;
        blocks += ch->top;
    }
    if (!n_live) return BSX_OK;
    if (blocks > kMaxChainBlocks || n_live > kMaxChains) return fail(h, BSX_ERR_INVALID, "internal: too many chains in one batch");
    while (h->ev_chain.size() < 4 * (size_t)n_live) {
        hipEvent_t e = nullptr;
        HIPCHK(h, hipEventCreate(&e));
        h->ev_chain.push_back(e);
    }
    if (!h->h_leaf) HIPCHK(h, hipHostMalloc((void**)&h->h_leaf, sizeof(LeafProgram) * kMaxChains, hipHostMallocDefault));
    HIPCHK(h, h->d_leaf.reserve(kMaxChains));
    const double pt0 = now_ms();
    // (descriptors and the counter blocks are one stretch of memory: one fill)
    HIPCHK(h, hipMemsetAsync(h->d_level, 0, kLevelDescBytes + sizeof(Counters) * blocks, h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    {
        std::vector<hipEvent_t> slot_busy(std::max(1u, B.n_side), nullptr);
        uint32_t n_listing = 0;                         // chains with lower levels so far: they take the slots in turn
        for (Chain* ch : chains) {
            if (ch->lv.empty()) continue;
            const uint32_t slot = (B.n_side > 1 && ch->top > 1) ? n_listing++ % B.n_side : 0u;
            if (int rc = enqueue_chain(h, env, sh, B, *ch, slot, slot_busy)) return rc;
        }
        for (hipEvent_t e : slot_busy) if (e) HIPCHK(h, hipStreamWaitEvent(h->stream, e, 0));      // join
    }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    const double pt1 = now_ms();
    if (int rc = fetch_counters(h, blocks)) return rc;
    ++tot.syncs;
    const double pt2 = now_ms();
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) {      // (events precede k_publish: complete by now, but ask nicely)
        HIPCHK(h, hipEventSynchronize(h->ev1));
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    }
    g_prof[1] += pt1 - pt0; g_prof[2] += pt2 - pt1; g_prof[3] += ms;
    tot.kernel_ms += ms;
    // evaluate_chain reads this copy: a detector pass started for one chain's unresolved classes counts into h_ctr's first
    // block again, which belongs to whichever chain was enqueued first -- not necessarily the one evaluated first
    h->ctr_seen.assign(h->h_ctr, h->h_ctr + blocks);
    for (Chain* ch : chains) {
        if (ch->lv.empty()) continue;
        float ms_top = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms_top, h->ev_chain[4 * ch->index], h->ev_chain[4 * ch->index + 1]));
        tot.launches += 2 * ch->top - 1;
        tot.dominant_ms += ms_top;
        tot.dominant_exec += h->h_ctr[ch->ctr_base].steps_exec;
        ++tot.dominant_launches;
        if (std::getenv("BSX_DEBUG")) {             // the estimate against what the chain took, level by level
            std::string line;
            double sum = ms_top * 1e3;
            char buf[96];
            std::snprintf(buf, sizeof buf, "d%u %.0f", ch->top, ms_top * 1e3);
            line += buf;
            for (uint32_t i = 1; i < ch->top; ++i) {
                const Counters& c = h->h_ctr[ch->ctr_base + i];
                const double us = (c.t_last && c.t_first_not) ? (double)(c.t_last - ~c.t_first_not) / h->wall_clock_khz * 1e3 : 0.0;
                sum += us;
                std::snprintf(buf, sizeof buf, " | d%u %.0f", ch->top - i, us);
                line += buf;
            }
            std::fprintf(stderr, "[bsx] chain %u: estimated %.0f us, took %.0f us (%s)\n", ch->index, chain_cost_us(h, ch->rel_mask, ch->top), sum, line.c_str());
        }
        for (uint32_t i = 1; i < ch->top; ++i) {
            const Counters& c = h->h_ctr[ch->ctr_base + i];
            if (!c.t_last || !c.t_first_not) continue;      // (an empty list: every workgroup left at once)
            tot.lower_ms += (double)(c.t_last - ~c.t_first_not) / h->wall_clock_khz;     // ticks -> ms
            tot.lower_exec += c.steps_exec;
            ++tot.lower_launches;
        }
    }
    ++tot.launches;         // (k_publish)
    return BSX_OK;
}

// A set of cubes (one block, or the sub-blocks of a split block): plan, enqueue all their chains, wait once, look -- and
// again, for those whose counters ask for it, from a shallower top / with the richer cache.  The results are kept aside until
// every cube is in; collapsed = they are in env.tot (all of them, or none).
int run_cubes(bsx_handle h, const CascadeEnv& env, const std::vector<Cube>& cubes, bool& collapsed) {
    collapsed = false;
    Totals& tot = env.tot;
    const CascadeShape sh = cascade_shape(h, env);
    std::vector<uint32_t> top(cubes.size(), 0);             // 0: the estimate chooses
    std::vector<char> done(cubes.size(), 0);
    Totals part;
    const CascadeEnv env_part{env.P, env.max_t, env.max_len, part, env.d_log};
    auto book_device_time = [&]() {                         // (device time and launches count whether or not the results are kept)
        tot.steps_exec += part.steps_exec; tot.kernel_ms += part.kernel_ms; tot.launches += part.launches;
        tot.dominant_ms += part.dominant_ms; tot.dominant_exec += part.dominant_exec; tot.dominant_launches += part.dominant_launches;
        tot.lower_ms += part.lower_ms; tot.lower_exec += part.lower_exec; tot.lower_launches += part.lower_launches;
        tot.syncs += part.syncs; tot.limit_hits += part.limit_hits;
    };
    for (int attempt = 0; attempt < 32; ++attempt) {
        const double pt_plan = now_ms();
        std::vector<Chain> chains;
        std::vector<size_t> who;
        chains.reserve(cubes.size());
        for (size_t i = 0; i < cubes.size(); ++i) {
            if (done[i]) continue;
            chains.emplace_back();
            who.push_back(i);
            if (int rc = plan_chain(h, sh, cubes[i], top[i], chains.back())) return rc;
            if (chains.back().lv.empty()) { book_device_time(); return BSX_OK; }
        }
        if (chains.empty()) break;
        std::vector<Chain*> ptrs;
        uint32_t blocks = 0;
        for (Chain& ch : chains) { ptrs.push_back(&ch); blocks += ch.top; }
        if (blocks > kMaxChainBlocks || chains.size() > kMaxChains) { book_device_time(); return BSX_OK; }
        // the chains whose lower levels are estimated to take longest go first: their tails run on the side streams while the
        // others' top levels still keep the handle's stream busy, instead of being what the batch ends on
        if (ptrs.size() > 2 && !(std::getenv("BSX_CUBE_ORDER_TAILS") && std::getenv("BSX_CUBE_ORDER_TAILS")[0] == '0')) {
            auto tail_us = [&](const Chain* ch) {
                const double top_us = kLevelOverheadUs + std::ldexp(1.0, __builtin_popcountll(ch->rel_mask[ch->top - 1])) * (ch->top + 0.3) / 4.5e5;
                return chain_cost_us(h, ch->rel_mask, ch->top) - top_us;
            };
            std::vector<std::pair<double, Chain*>> keyed;
            for (Chain* ch : ptrs) keyed.emplace_back(-tail_us(ch), ch);
            std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<double, Chain*>& a, const std::pair<double, Chain*>& b) { return a.first < b.first; });
            for (size_t i = 0; i < ptrs.size(); ++i) ptrs[i] = keyed[i].second;
        }
        ChainBatch B;
        bool ok = false;
        if (int rc = prepare_batch(h, env_part, ptrs, B, ok)) return rc;
        if (!ok) { book_device_time(); return BSX_OK; }
        g_prof[0] += now_ms() - pt_plan;
        if (int rc = run_batch(h, env_part, sh, B, ptrs)) return rc;
        bool repeat = false;
        const double pt_eval = now_ms();
        for (size_t q = 0; q < chains.size(); ++q) {
            MergedTable pass_table;
            u128 pass_none = 0, pass_ref = 0;
            int verdict = kChainOk;
            uint32_t lower_to = 0;
            if (int rc = evaluate_chain(h, env_part, sh, chains[q], pass_table, pass_none, pass_ref, verdict, lower_to)) return rc;
            if (verdict == kChainOk) {
                fold_table(part.merged, pass_table);
                part.n_none += pass_none;
                part.steps_ref += pass_ref;
                done[who[q]] = 1;
            } else if (verdict == kChainLower && lower_to >= 1) {
                if (std::getenv("BSX_DEBUG"))
                    std::fprintf(stderr, "[bsx] chain %u: a list overflowed, again from depth %u\n", chains[q].index, lower_to);
                top[who[q]] = lower_to;
                // (a whole block remembers that for the rest of the problem; one sub-block of a dozen, whose lists are anybody's
                // guess while the tree is grown on guesses, does not cap the others)
                if (cubes.size() == 1) h->cube_depth_cap = lower_to;
            } else if (verdict == kChainRepeat) {
                repeat = true;
            } else {                                        // not a space for cubes
                book_device_time();
                return BSX_OK;
            }
        }
        g_prof[5] += now_ms() - pt_eval;
        if (repeat) {
            unsigned int known = 0;
            HIPCHK(h, hipMemcpy(&known, h->d_cc_count.p, sizeof(known), hipMemcpyDeviceToHost));
            ++part.syncs;
            // the attractor cannot be cached: no cube for this block (else: the detector pass marked the journal stale)
            if (known <= h->h_journal.size()) { book_device_time(); return BSX_OK; }
        }
    }
    book_device_time();
    for (char d : done) if (!d) return BSX_OK;
    fold_table(tot.merged, part.merged);
    tot.n_none += part.n_none;
    tot.steps_ref += part.steps_ref;
    collapsed = true;
    return BSX_OK;
}

// ---- splitting a block into sub-blocks ---------------------------------------------------------------------------------
// The digits F^d depends on over a whole block are the union over everything the block contains.  Fix one well-chosen digit
// and, in a network of canalizing rules, whole sub-trees of dependence disappear in each half: the two sub-blocks together
// have fewer classes than the block (north star, 2^63 problems: 2^32 classes at depth 4; after a dozen greedy splits
// 2^25.4).  plan_split grows that tree greedily on the cost estimate -- at each node the digit whose two halves are cheapest
// together, as long as that saves at least 15 % and the node is worth more than a few launches -- and returns the leaves
// as (fix_mask, fix_vals) over the digit index.  Any tree is correct (the leaves partition the block); only speed depends
// on it, so the tree found for the first block of a size is reused for the other blocks of that size in the space.
struct SplitLeaf { uint64_t mask, vals; };
constexpr uint32_t kSplitMinBits = 52;      // smaller blocks finish in less time than the extra launches of their sub-blocks take

// the estimate for the (sub-)block with the digits `mask` fixed at `vals`; top_rel = the digits its top level enumerates
double cube_cost_us(bsx_handle h, const CascadeShape& sh, uint64_t d_lo, uint32_t a_bits, uint64_t mask, uint64_t vals, uint64_t* top_rel = nullptr) {
    Cube c;
    std::vector<uint64_t> rel_mask;
    build_cube(h, d_lo, a_bits, c, nullptr, mask, vals);
    cube_levels(h, c, sh.max_depth, rel_mask);
    double est = 0;
    const uint32_t top = choose_top(h, sh, rel_mask, sh.max_depth, &est);
    if (top_rel) *top_rel = rel_mask[top - 1];
    return est;
}

void plan_split(bsx_handle h, const CascadeShape& sh, uint64_t d_lo, uint32_t a_bits, bool forced, std::vector<SplitLeaf>& leaves) {
    leaves.clear();
    const double min_cost_us = 3 * kLevelOverheadUs;        // below this a node is a handful of launches: not worth halving
    std::vector<SplitLeaf> todo{{0, 0}};
    while (!todo.empty()) {
        const SplitLeaf nd = todo.back();
        todo.pop_back();
        uint64_t top_rel = 0;
        const double here = cube_cost_us(h, sh, d_lo, a_bits, nd.mask, nd.vals, &top_rel);
        bool split = false;
        // (forced -- BSX_CUBE_SPLIT=1, tests: a tree of eight leaves whatever the estimates say)
        if (forced ? leaves.size() + todo.size() + 2 <= 8 : (here > min_cost_us && leaves.size() + todo.size() + 2 <= kMaxChains)) {
            double best = 0;
            int best_digit = -1;
            for (uint64_t left = top_rel; left; left &= left - 1) {
                const int j = __builtin_ctzll(left);
                const double both = cube_cost_us(h, sh, d_lo, a_bits, nd.mask | (1ull << j), nd.vals) +
                                    cube_cost_us(h, sh, d_lo, a_bits, nd.mask | (1ull << j), nd.vals | (1ull << j));
                if (best_digit < 0 || both < best) { best = both; best_digit = j; }
            }
            if (best_digit >= 0 && (forced || best < 0.85 * here)) {
                todo.push_back(SplitLeaf{nd.mask | (1ull << best_digit), nd.vals});
                todo.push_back(SplitLeaf{nd.mask | (1ull << best_digit), nd.vals | (1ull << best_digit)});
                split = true;
            }
        }
        if (!split) leaves.push_back(nd);
    }
}

// One aligned block: as the sub-blocks of its split tree where that pays (all their chains enqueued one after the other, one
// wait), else as one cube.  collapsed = false: the caller runs the block through the plain tiles.
int run_block(bsx_handle h, const CascadeEnv& env, uint64_t d_lo, uint32_t a_bits, bool& collapsed) {
    collapsed = false;
    const CascadeShape sh = cascade_shape(h, env);
    const char* split_env = std::getenv("BSX_CUBE_SPLIT");              // "0": no sub-blocks (A/B runs, tests); "1": whatever the size
    const bool forced = split_env && split_env[0] == '1';
    if (!(split_env && split_env[0] == '0') && sh.tp == 0 && (a_bits >= kSplitMinBits || forced)) {
        // the tree grown for the first block of a size serves the others of that size too, as long as the estimate says it
        // helps there (the high digits differ, so the dependence may); a large block that it does not help gets its own
        std::vector<std::pair<uint64_t, uint64_t>>& tree = h->split_cache[a_bits];
        auto estimate = [&]() {
            double est = 0;
            for (const auto& l : tree) est += cube_cost_us(h, sh, d_lo, a_bits, l.first, l.second);
            return est;
        };
        double experience = 0;                      // top-level classes whose listing the handle has seen so far
        for (uint32_t d = 0; d <= kMaxCubeLevels; ++d) experience += h->near_seen[0][d][0];
        auto grow = [&]() {
            const double t0 = now_ms();
            std::vector<SplitLeaf> fresh;
            plan_split(h, sh, d_lo, a_bits, forced, fresh);
            tree.clear();
            for (const SplitLeaf& l : fresh) tree.emplace_back(l.mask, l.vals);
            h->split_learned[a_bits] = experience;
            if (std::getenv("BSX_DEBUG"))
                std::fprintf(stderr, "[bsx] split tree for blocks of 2^%u: %zu leaves, planned in %.2f ms (list fractions from %.3g classes seen)\n", a_bits,
                             tree.size(), now_ms() - t0, experience);
        };
        const double pt_est = now_ms();
        const double whole = cube_cost_us(h, sh, d_lo, a_bits, 0, 0);
        bool use = false;
        // (a tree grown on guesses, or on what small blocks showed, is grown again when the handle has seen 64 times more)
        if (tree.empty() || experience > 64.0 * (h->split_learned[a_bits] + 1024.0)) { grow(); use = tree.size() > 1; }
        else if (tree.size() > 1) {
            use = forced || estimate() < 0.8 * whole;
            // (a large block the tree does not help gets its own -- a few times per block size, not for every block of a sweep)
            if (!use && a_bits >= 60 && h->split_regrown[a_bits] < 2) { ++h->split_regrown[a_bits]; grow(); use = tree.size() > 1; }
        }
        g_prof[4] += now_ms() - pt_est;
        if (use) {
            std::vector<Cube> cubes(tree.size());
            bool eligible = true;
            for (size_t i = 0; i < tree.size() && eligible; ++i) {
                build_cube(h, d_lo, a_bits, cubes[i], nullptr, tree[i].first, tree[i].second);
                // (a sub-block that does not shrink at least fourfold has no cube path of its own: the block goes unsplit)
                if (!cubes[i].ok || cubes[i].rel.size() + 2 > cubes[i].n_free) eligible = false;
            }
            if (eligible) {
                if (int rc = run_cubes(h, env, cubes, collapsed)) return rc;
                if (collapsed) return BSX_OK;
            }
        }
    }
    Cube c;
    build_cube(h, d_lo, a_bits, c);
    // worth it when the block shrinks at least fourfold (otherwise the tiles do as well and keep member masks)
    if (c.ok && c.rel.size() + 2 <= a_bits) return run_cubes(h, env, std::vector<Cube>{c}, collapsed);
    return BSX_OK;
}

// ---- one segment: [first, first + count) inside the space the handle currently describes ------------------------------
// (for a plain space -- no variations, at most 64 'any' nodes -- the segment lies within its 2^n_any initial states,
// count <= 2^64; with variations the index carries into the variant number)
int attract_segment(bsx_handle h, const bsx_index& first, u128 count, uint64_t max_t, uint64_t max_len,
                    bsx_problem_rec* per_problem, Totals& tot) {
    DevBuf<LogRec>& d_log = h->d_log;
    DevBuf<ProblemRec32> d_pp;
    if (per_problem) HIPCHK(h, d_pp.alloc((size_t)count));

    AttractParams P{};
    P.net = h->net;
    P.sp = h->sp;
    set_first(P.sp, &first);
    P.count = 0;
    P.cap_rel_inf = max_t == BSX_T_INF ? 1 : 0;
    P.max_t = max_t;
    P.max_len = max_len;
    P.ctr = h->d_ctr;
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

    MergedTable& merged = tot.merged;
    auto account = [&](const AttractRun& r) {
        tot.n_none += r.ctr.n_none; tot.steps_ref += r.ctr.steps_ref; tot.steps_exec += r.ctr.steps_exec;
        tot.kernel_ms += r.ms; ++tot.launches; tot.limit_hits += r.ctr.step_limit_hits;
    };

    // Fast path: simple enumeration (no variations, 'any' nodes = nodes 0..a-1 or a few runs, a <= 64), cycle cache on.
    // [discovery prefix with the detector] -> lean / pool kernel -> stragglers.
    // (a short uniform warm-up is fine; its length enters the lean kernel's 32-bit sums of trajectory_l^2)
    const bool simple = h->sp.n_any <= 64 && (h->sp.identity_any || h->sp.n_runs) && !h->sp.n_fv && !h->sp.n_pv && h->sp.tp_origin <= 200;
    bool use_fast = P.cc.enabled && simple && h->fast_ok && count >= kFastMinProblems;
    if (const char* e = std::getenv("BSX_LEAN")) use_fast = use_fast && std::atoi(e) != 0;      // tuning / test knob
    if (std::getenv("BSX_DEBUG")) std::fprintf(stderr, "[bsx] attract: count %llu%s cache %u identity %u n_any %u n_fv %u n_pv %u tp %u fast_ok %d -> lean path %d\n", (unsigned long long)count, (count >> 64) ? " (+2^64)" : "", P.cc.enabled, h->sp.identity_any, h->sp.n_any, h->sp.n_fv, h->sp.n_pv, h->sp.tp_origin, (int)h->fast_ok, (int)use_fast);
    u128 done = 0;
    if (use_fast) {
        unsigned int known = 0;
        if (h->journal_stale || h->h_journal.empty()) {
            HIPCHK(h, hipMemcpy(&known, h->d_cc_count.p, sizeof(known), hipMemcpyDeviceToHost));
            ++tot.syncs;
        } else known = (unsigned int)h->h_journal.size();
        if (known == 0) {
            // Nothing cached yet: run the detector over a pseudo-random sample of the range (all digit
            // positions vary), only to fill the cycle cache; its results are discarded and every problem
            // is counted exactly once below.
            const uint64_t m = (uint64_t)std::min<u128>(count, kDiscoverySample);
            std::vector<uint32_t> sample(m);
            for (uint64_t i = 0; i < m; ++i) {
                uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;        // splitmix64 finaliser
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                sample[i] = (uint32_t)((z ^ (z >> 31)) % (uint64_t)std::min<u128>(count, (u128)1 << 32));
            }
            DevBuf<uint32_t> d_sample;
            HIPCHK(h, d_sample.upload(sample));
            AttractParams Q = P;
            Q.count = m;
            Q.offsets = d_sample.p;
            Q.per_problem = nullptr;
            AttractRun r;
            if (int rc = launch_attract_pass(h, Q, kPassGeneral, d_log, nullptr, r, tot)) return rc;
            tot.kernel_ms += r.ms; ++tot.launches; tot.steps_exec += r.ctr.steps_exec;
            HIPCHK(h, hipMemcpy(&known, h->d_cc_count.p, sizeof(known), hipMemcpyDeviceToHost));
            ++tot.syncs;
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
    // Lean / pool kernel over [done, seg_end) in tiles, then the detector over whatever the fast path gave up on.
    auto run_tiles = [&](u128 seg_end) -> int {
    while (use_fast && h->fast_ok && done < seg_end) {
        const uint64_t tile = (uint64_t)std::min<u128>(seg_end - done, h->fast_calibrated ? kLeanTile : kProbeTile);
        if (int rc = lean_mirror_slots(h, nullptr, &tot)) return rc;          // (refreshes h->h_journal if the detector ran since)
        const unsigned int known_before_tile = (unsigned int)h->h_journal.size();
        // straggler list: one word per problem, or up to three per class (base + 64-bit member mask) from the
        // pool kernel -- probe tiles get room for every problem as a class of its own, big tiles for a third
        // (more stragglers than that and the lean path is the wrong tool anyway)
        const uint64_t strag_cap = h->fast_calibrated ? tile : 3 * tile;
        DevBuf<uint32_t>& d_strag = h->d_strag;
        if (d_strag.n < strag_cap) HIPCHK(h, d_strag.alloc(strag_cap));
        AttractParams Q = P;
        advance_first(Q.sp, first, (uint64_t)done);
        Q.count = tile;
        Q.fast_steps = h->fast_steps;
        // the pool kernel first runs with member counts (classes of different groups merge too); that only works
        // while nothing has to go back to the general kernel, so a tile that raises the abort flag is repeated
        // with member masks.  Per-problem records need the masks from the start.
        bool counting = merge_mode == 2 && !per_problem && (h->fast_calibrated || std::getenv("BSX_FORCE_COUNTING"));     // (knob: tests)
        Q.merge = counting ? 2u : (merge_lanes ? 1u : 0u);
        Q.per_problem = per_problem ? d_pp.p + (uint64_t)done : nullptr;
        Q.stragglers = d_strag.p;
        Q.stragglers_cap = strag_cap;
        AttractRun r;
        MergedTable tile_table;         // folded into `merged` only if the pass is accepted
        if (int rc = launch_attract_pass(h, Q, merge_mode == 2 ? kPassPool : kPassLean, d_log, &tile_table, r, tot)) return rc;
        if (counting && (r.ctr.straggler_overflow & 2u)) {
            tot.kernel_ms += r.ms; ++tot.launches;              // dropped pass
            counting = false;
            Q.merge = 1u;
            tile_table.clear();
            r = AttractRun{};
            if (int rc = launch_attract_pass(h, Q, kPassPool, d_log, &tile_table, r, tot)) return rc;
        }
        if (r.ctr.straggler_overflow) {
            // more (group, mask) pairs than the list holds: the cache does not cover this space.  Drop the
            // pass and give the rest of the range to the detector.
            tot.kernel_ms += r.ms; ++tot.launches;
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
                ++tot.syncs;
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
            if (int rc = launch_attract_pass(h, S, kPassGeneral, d_log, &merged, rs, tot)) return rc;
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
                ++tot.syncs;
                if (known_now <= known_before_tile) h->fast_ok = false;
            }
        }
    }
    // not (or no longer) a case for the lean path: the detector takes the rest, 2^32 problems per launch
    while (done < seg_end) {
        const uint64_t tile = (uint64_t)std::min<u128>(seg_end - done, kGeneralTile);
        AttractParams Q = P;
        const bsx_index at = index_plus(first, done, h->sp.n_any);
        set_first(Q.sp, &at);
        Q.count = tile;
        Q.per_problem = per_problem ? d_pp.p + (uint64_t)done : nullptr;
        AttractRun r;
        if (int rc = launch_attract_pass(h, Q, kPassGeneral, d_log, &merged, r, tot)) return rc;
        account(r);
        done += tile;
    }
    return BSX_OK;
    };

    // ---- cube collapse: aligned blocks of >= 2^kCubeMinBits problems are enumerated by their relevant digits
    // only (see build_cube).  Everything before the first / after the last such block goes through the tiles.
    const char* cubes_env = std::getenv("BSX_CUBES");                     // "0": off (A/B runs, tests)
    // (a warm-up under origin perturbations is fine: the first update still depends on the relevant digits only)
    const bool cubes_ok = use_fast && merge_mode == 2 && !per_problem &&
                          !(cubes_env && cubes_env[0] == '0') && h->sp.n_any >= kCubeMinBits;
    if (cubes_ok) {
        // [first, first + count) in digit values; blocks are aligned in the digit value, not in the offset
        const u128 lo = first.init_digits[0], hi = lo + count;
        const u128 unit = (u128)1 << kCubeMinBits;
        u128 at = (lo + unit - 1) / unit * unit;
        const u128 body_end = hi / unit * unit;
        const CascadeEnv env{P, max_t, max_len, tot, d_log};
        if (at < body_end) {
            if (int rc = run_tiles(at - lo)) return rc;
            while (at < body_end) {
                uint32_t a_bits = kCubeMaxBits;
                while (a_bits > kCubeMinBits && ((at & (((u128)1 << a_bits) - 1)) != 0 || at + ((u128)1 << a_bits) > body_end)) --a_bits;
                a_bits = std::min(a_bits, h->sp.n_any);
                bool collapsed = false;
                // a block that does not collapse is tried again in halves down to 2^32 problems, below that it is the tiles' turn
                for (;;) {
                    if (int rc = run_block(h, env, (uint64_t)at, a_bits, collapsed)) return rc;
                    if (collapsed || a_bits <= 32) break;
                    --a_bits;
                }
                const u128 block_end = (at - lo) + ((u128)1 << a_bits);
                if (collapsed) done = block_end;
                else if (int rc = run_tiles(block_end)) return rc;
                at += (u128)1 << a_bits;
            }
        }
    }
    if (int rc = run_tiles(count)) return rc;

    if (per_problem) {
        const uint32_t nw = h->net.nw;
        const uint64_t n = (uint64_t)count;
        std::vector<ProblemRec32> pp(n);
        HIPCHK(h, hipMemcpy(pp.data(), d_pp.p, n * sizeof(ProblemRec32), hipMemcpyDeviceToHost));
        for (uint64_t p = 0; p < n; ++p) {
            bsx_problem_rec o{};
            for (uint32_t w = 0; w < nw; ++w) o.key[w >> 1] |= (uint64_t)pp[p].key[w] << (32 * (w & 1));
            o.length = pp[p].length; o.trajectory_l = pp[p].trajectory_l; o.found = pp[p].found;
            per_problem[p] = o;
        }
    }
    return BSX_OK;
}

// Spaces with more than 64 'any' nodes (e.g. a 128-node network with every node 'any'): only the 64 lowest initial-state
// digits change inside 2^64 consecutive problems.  Such a stretch is run as the space in which exactly those are 'any'
// and the higher digits belong to the origin state -- a plain space, which gets the lean / pool / cube paths.  (Same
// network, same fixed nodes: the cycle cache carries over from stretch to stretch.)
int attract_high_digits(bsx_handle h, const bsx_index& first, u128 count, uint64_t max_t, uint64_t max_len, Totals& tot) {
    struct Restore {                    // the handle describes the whole space again, whatever happens below
        bsx_handle h; DevSpace sp; std::vector<uint32_t> any;
        ~Restore() { h->sp = sp; h->h_any = any; }
    } restore{h, h->sp, h->h_any};
    const DevSpace whole = h->sp;
    const std::vector<uint32_t> any = h->h_any;
    bool identity = true;
    for (uint32_t j = 0; j < 64; ++j) identity = identity && any[j] == j;
    bsx_index at = first;
    while (count) {
        const u128 room = ((u128)1 << 64) - at.init_digits[0];
        const u128 seg = std::min(count, room);
        DevSpace sv = whole;
        for (uint32_t j = 64; j < whole.n_any; ++j)
            if ((at.init_digits[j >> 6] >> (j & 63)) & 1ull) sv.origin[any[j] >> 5] |= 1u << (any[j] & 31);
        sv.n_any = 64;
        sv.identity_any = identity ? 1 : 0;
        sv.n_runs = 0;
        if (!identity) {
            uint32_t j = 0, r = 0;
            while (j < 64) {
                uint32_t len = 1;
                while (j + len < 64 && any[j + len] == any[j] + len && ((any[j] + len) >> 5) == (any[j] >> 5)) ++len;
                sv.deposit[2 * r] = j | (any[j] >> 5) << 8 | (any[j] & 31u) << 16;
                sv.deposit[2 * r + 1] = len >= 32 ? 0xFFFFFFFFu : (1u << len) - 1u;
                ++r;
                j += len;
            }
            sv.n_runs = r;                  // <= 64 = kMaxDepositRuns
        }
        h->sp = sv;
        h->h_any.assign(any.begin(), any.begin() + 64);
        bsx_index f{};
        f.init_digits[0] = at.init_digits[0];
        const int rc = attract_segment(h, f, seg, max_t, max_len, nullptr, tot);
        h->sp = whole;
        h->h_any = any;
        if (rc) return rc;
        at = index_plus(at, seg, whole.n_any);
        count -= seg;
    }
    return BSX_OK;
}

// attract.py:262-302 semantics for every problem of [first, first + count): the body of both entry points.
int attract_core(bsx_handle h, const bsx_index& first, u128 count, uint64_t max_t, uint64_t max_len, uint32_t cap,
                 bsx_problem_rec* per_problem, Totals& tot) {
    if (int rc = ensure_attractor_table(h, cap)) return rc;
    const bool lean_off = std::getenv("BSX_LEAN") && std::atoi(std::getenv("BSX_LEAN")) == 0;
    int rc;
    if (h->sp.n_any > 64 && !h->sp.n_fv && !h->sp.n_pv && h->sp.tp_origin <= 200 && !per_problem && count >= (1u << 13) &&
        h->cache_enabled && !lean_off)
        rc = attract_high_digits(h, first, count, max_t, max_len, tot);
    else
        rc = attract_segment(h, first, count, max_t, max_len, per_problem, tot);
    if (rc) return rc;
    if (int rc2 = drain_attractor_table(h, tot.merged)) return rc2;
    if (tot.merged.size() > cap) return fail(h, BSX_ERR_TABLE_FULL, "more distinct attractors than the caller's table capacity");
    if (std::getenv("BSX_PROFILE")) {
        std::fprintf(stderr, "[bsx] profile: passes: setup %.3f, enqueue %.3f, wait %.3f (kernels %.3f), split estimates %.3f, reading the counters %.3f; %u host syncs\n",
                     g_prof[0], g_prof[1], g_prof[2], g_prof[3], g_prof[4], g_prof[5], tot.syncs);
        for (double& v : g_prof) v = 0;
    }
    return BSX_OK;
}

int attract_preamble(bsx_handle h, const bsx_index* first, u128 count, uint64_t max_t) {
    if (!h->have_net || !h->have_space) return fail(h, BSX_ERR_STATE, "network / problem space not set");
    if (int rc = check_range(h, first, count)) return rc;
    if (int rc = check_max_t(h, max_t)) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    return BSX_OK;
}

bool fgraph_knob(bsx_handle h) {            // knob: route eligible calls through the functional-graph mode
    const char* fg = std::getenv("BSX_FGRAPH");
    return fg && fg[0] == '1' && h->n_nodes <= 32 && h->sp.n_any == h->n_nodes && h->sp.identity_any && !h->sp.n_fv &&
           !h->sp.n_pv && h->lut_mode != 2;
}

}  // namespace

extern "C" int bsx_run_attract(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                               uint64_t max_len, bsx_attr_rec* table, uint32_t cap, uint32_t* n_out,
                               uint64_t* n_no_attractor, bsx_problem_rec* per_problem, bsx_stats* stats) {
    if (!h) return BSX_ERR_INVALID;
    if (!table || !n_out) return fail(h, BSX_ERR_INVALID, "table / n_out is null");
    if (int rc = attract_preamble(h, first, count, max_t)) return rc;
    const double t_begin = now_ms();
    *n_out = 0;
    if (n_no_attractor) *n_no_attractor = 0;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (count == 0) return BSX_OK;
    if (!per_problem && fgraph_knob(h))
        return bsx_run_attract_fgraph(h, first, count, max_t, max_len, table, cap, n_out, n_no_attractor, stats);
    if (per_problem && count > (1ull << 32)) return fail(h, BSX_ERR_INVALID, "at most 2^32 problems per call with per-problem records");

    Totals tot;
    if (int rc = attract_core(h, *first, count, max_t, max_len, cap, per_problem, tot)) return rc;
    uint32_t i = 0;
    for (auto& kv : tot.merged) {
        const WideRec& w = kv.second;
        if ((w.count >> 64) != 0 || !w.sum_l.fits(1) || !w.sum_l2.fits(2))
            return fail(h, BSX_ERR_RANGE_TOO_LARGE, "an attractor's count / sum of trajectory lengths exceeds the 64-bit fields of bsx_attr_rec: use bsx_run_attract2");
        bsx_attr_rec& a = table[i++];
        for (int k = 0; k < BSX_MAX_WORDS; ++k) a.key[k] = w.key[k];
        a.length = w.length; a.count = (uint64_t)w.count; a.sum_l = w.sum_l.w[0];
        a.sum_l2_lo = w.sum_l2.w[0]; a.sum_l2_hi = w.sum_l2.w[1];
    }
    *n_out = i;
    if (n_no_attractor) *n_no_attractor = (uint64_t)tot.n_none;
    if (stats) {
        stats->problems = count;
        stats->state_steps = (uint64_t)tot.steps_ref;           // (count < 2^64 and trajectories < 2^31: may wrap only beyond 2^33 x ... problems; bsx_run_attract2 is wide)
        stats->executed_steps = tot.steps_exec;
        stats->kernel_ms = tot.kernel_ms;
        stats->kernel_launches = tot.launches;
        stats->total_ms = now_ms() - t_begin;
    }
    if (tot.limit_hits) return fail(h, BSX_ERR_STEP_LIMIT, "a trajectory reached the internal step limit without closing its cycle");
    return BSX_OK;
}

extern "C" int bsx_run_attract2(bsx_handle h, bsx_u128 first_flat, bsx_u128 count_flat, uint64_t max_t, uint64_t max_len,
                                bsx_attr_rec2* table, uint32_t cap, uint32_t* n_out, bsx_u128* n_no_attractor,
                                bsx_stats2* stats) {
    if (!h) return BSX_ERR_INVALID;
    if (!table || !n_out) return fail(h, BSX_ERR_INVALID, "table / n_out is null");
    if (!h->have_net || !h->have_space) return fail(h, BSX_ERR_STATE, "network / problem space not set");
    // flat index I = init_digits + variant * 2^n_any (batching.py:212-229)
    const u128 I = ((u128)first_flat.hi << 64) | first_flat.lo, count = ((u128)count_flat.hi << 64) | count_flat.lo;
    const uint32_t n_any = h->sp.n_any;
    bsx_index first{};
    if (n_any >= 128) { first.init_digits[0] = (uint64_t)I; first.init_digits[1] = (uint64_t)(I >> 64); }
    else {
        const u128 low = I & (((u128)1 << n_any) - 1), variant = I >> n_any;
        if ((variant >> 64) != 0) return fail(h, BSX_ERR_UNSUPPORTED, "variant part of the problem index exceeds 64 bits");
        first.init_digits[0] = (uint64_t)low; first.init_digits[1] = (uint64_t)(low >> 64);
        first.variant = (uint64_t)variant;
    }
    if (int rc = attract_preamble(h, &first, count, max_t)) return rc;
    const double t_begin = now_ms();
    *n_out = 0;
    if (n_no_attractor) *n_no_attractor = bsx_u128{0, 0};
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (count == 0) return BSX_OK;

    Totals tot;
    if (int rc = attract_core(h, first, count, max_t, max_len, cap, nullptr, tot)) return rc;
    uint32_t i = 0;
    for (auto& kv : tot.merged) {
        const WideRec& w = kv.second;
        bsx_attr_rec2& a = table[i++];
        for (int k = 0; k < BSX_MAX_WORDS; ++k) a.key[k] = w.key[k];
        a.length = w.length;
        a.count.lo = (uint64_t)w.count; a.count.hi = (uint64_t)(w.count >> 64);
        for (int k = 0; k < 3; ++k) a.sum_l[k] = w.sum_l.w[k];
        for (int k = 0; k < 4; ++k) a.sum_l2[k] = w.sum_l2.w[k];
    }
    *n_out = i;
    if (n_no_attractor) { n_no_attractor->lo = (uint64_t)tot.n_none; n_no_attractor->hi = (uint64_t)(tot.n_none >> 64); }
    if (stats) {
        stats->problems = count_flat;
        stats->state_steps.lo = (uint64_t)tot.steps_ref; stats->state_steps.hi = (uint64_t)(tot.steps_ref >> 64);
        stats->executed_steps = tot.steps_exec;
        stats->kernel_ms = tot.kernel_ms;
        stats->dominant_ms = tot.dominant_ms;
        stats->dominant_executed_steps = tot.dominant_exec;
        stats->dominant_launches = tot.dominant_launches;
        stats->lower_ms = tot.lower_ms;
        stats->lower_executed_steps = tot.lower_exec;
        stats->lower_launches = tot.lower_launches;
        stats->kernel_launches = tot.launches;
        stats->host_syncs = tot.syncs;
        stats->total_ms = now_ms() - t_begin;
    }
    if (tot.limit_hits) return fail(h, BSX_ERR_STEP_LIMIT, "a trajectory reached the internal step limit without closing its cycle");
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
    HIPCHK(h, hipMemsetAsync(h->d_ctr, 0, sizeof(Counters), h->stream));

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
    P.ctr = h->d_ctr;
    P.table = h->d_table.p;
    P.table_mask = h->table_slots - 1;
    const uint64_t first_state = first->init_digits[0];
    HIPCHK(h, launch_fg_aggregate(pair, d_cyc.p, cyc_slots - 1, tp ? h->d_fg_warm.p : nullptr, tp, first_state, count, cap_rel, max_len,
                                  capped ? max_t : 0, P, cus, h->stream));
    ++launches;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    Counters ctr{};
    HIPCHK(h, hipMemcpyAsync(&ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->table_dirty = true;
    if (ctr.table_overflow) { MergedTable junk; (void)drain_attractor_table(h, junk); return fail(h, BSX_ERR_TABLE_FULL, "more distinct attractors than the caller's table capacity"); }
    MergedTable merged;
    if (int rc = drain_attractor_table(h, merged)) return rc;
    if (merged.size() > cap) return fail(h, BSX_ERR_TABLE_FULL, "more distinct attractors than the caller's table capacity");
    uint32_t i = 0;
    for (auto& kv : merged) {                       // (at most 2^32 problems: every sum fits the record)
        const WideRec& w = kv.second;
        bsx_attr_rec& a = table[i++];
        for (int k = 0; k < BSX_MAX_WORDS; ++k) a.key[k] = w.key[k];
        a.length = w.length; a.count = (uint64_t)w.count; a.sum_l = w.sum_l.w[0];
        a.sum_l2_lo = w.sum_l2.w[0]; a.sum_l2_hi = w.sum_l2.w[1];
    }
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
