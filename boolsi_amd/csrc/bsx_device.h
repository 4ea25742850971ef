// Shared host/device structures of the gfx950 engine (kernel parameters).  See DESIGN.md.
#pragma once
#include <stdint.h>

namespace bsx {

constexpr int kWave = 64;
constexpr int kBlock = 512;                 // 8 waves per workgroup share one LUT + cache mirror in LDS
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kPoolBlockThreads = 768;      // the pool kernel's workgroup (bsx_pool_kernel.h)
constexpr uint32_t kPoolCap = 112;          // ... classes per wave (ring buffer; > 64 + what a fresh stage leaves)
constexpr uint32_t kPoolSlots = 128;        // ... merge slots per wave (one-byte lane ids; 256 slots: 2 % fewer updates, not worth 1.5 KiB of LDS)
constexpr uint32_t pool_rec_words(uint32_t nw) { return nw + 4; }      // ... ring record: state, group base, members lo/hi, time
constexpr uint32_t kLowerFoundWords = 256;  // ... lower-level build: room for the list of cycle states inside the block
constexpr int kMaxW32 = 8;                  // 32-bit words per state (n <= 256)
constexpr int kMaxMuxK = 6;                 // nodes with more predecessors take the "wide" path
constexpr int kTableSlots = 64;             // per-wave attractor table: one slot per lane (registers)
constexpr uint32_t kStepLimit = 1u << 30;   // internal per-trajectory step limit (u32 counters)
constexpr uint64_t kDigestSeed = 0xCBF29CE484222325ull;
constexpr uint64_t kDigestPrime = 0x100000001B3ull;
constexpr uint32_t kMaxDepositRuns = 64;

// Network tables in HBM (staged into LDS by each workgroup where they fit).
struct DevNet {
    uint32_t n_nodes;
    uint32_t nw;            // 32-bit words per state
    uint32_t k_mux;         // gathered predecessor slots (max degree over mux nodes, >= 1)
    uint32_t n_chunks;      // 8-bit chunks of the state that feed the gather LUT = ceil(n/8)
    uint32_t lut_words;     // n_chunks * 256 * k_mux * nw
    uint32_t n_wide;
    const uint32_t* lut;    // [chunk][byte value][slot j][word w]: bits i with pred_j(i) in chunk and set in value
    const uint32_t* masks;  // [1 << k_mux][nw]: bit i = TT_i(idx)   (0 for wide nodes)
    const uint32_t* wide_desc;   // per wide node: node, k, first pred, first tt word (u32)
    const uint32_t* wide_preds;
    const uint32_t* wide_tt;     // 32-bit words of the wide truth tables
};

// Problem space (enumeration) tables.
struct DevSpace {
    uint32_t origin[kMaxW32];
    uint32_t fixmask[kMaxW32];      // origin fixed nodes
    uint32_t fixval[kMaxW32];
    uint32_t n_any;
    uint32_t identity_any;          // any-nodes are exactly nodes 0..n_any-1: digits deposit = OR
    // n_any <= 64 and the 'any' nodes form at most kMaxDepositRuns runs of consecutive nodes within one
    // 32-bit state word: the digits are deposited run by run (shift, mask, or) instead of bit by bit.
    // run r: digits [src, src + len) -> bits [shift, shift + len) of word `word`;
    // deposit[2r] = src | word << 8 | shift << 16, deposit[2r + 1] = ((1 << len) - 1)
    uint32_t n_runs;                // 0 = no run plan (identity spaces do not need one)
    uint32_t deposit[2 * kMaxDepositRuns];
    uint32_t n_fv;
    uint32_t n_pv;
    uint32_t tp_origin;             // last origin perturbation time (0 = none)
    uint32_t pad;
    const uint32_t* any_nodes;      // [n_any]
    const uint32_t* fv;             // [n_fv][2]  node, range
    const uint32_t* pv;             // [n_pv][3]  t, node, range
    const uint32_t* sched_set;      // [tp_origin + 1][nw]  bits forced to 1 at time t
    const uint32_t* sched_clr;      // [tp_origin + 1][nw]  bits forced to 0 at time t
    uint64_t first_digits[4];
    uint64_t first_variant;
};

// One record of the device-side attractor log (merged on the host).
struct LogRec {
    uint32_t key[kMaxW32];
    uint32_t length;
    uint32_t pad;
    uint64_t count;             // 64 bits: one class of a cube pass can stand for 2^40 problems
    uint64_t sum_l;
    uint64_t sum_l2_lo, sum_l2_hi;
};
static_assert(sizeof(LogRec) == 72, "LogRec layout");

struct ProblemRec32 {           // per-problem output, device layout
    uint32_t key[kMaxW32];
    uint32_t length;
    uint32_t trajectory_l;
    uint32_t found;
    uint32_t pad;
};

struct alignas(256) Counters {   // zeroed before every launch (a multiple of 256 bytes: one fill kernel, not two)
    unsigned long long cursor;          // next chunk of problems
    unsigned long long steps_ref;       // reference-equivalent steps
    unsigned long long steps_exec;      // executed network updates
    unsigned long long n_none;          // problems without attractor
    unsigned long long log_cursor;      // records appended to the log / hits
    unsigned int log_overflow;
    unsigned int step_limit_hits;
    unsigned long long n_stragglers;    // fast kernel: problems handed to the general kernel
    unsigned long long n_cache_resolved;// general kernel: problems that ended on a cached cycle state
    unsigned long long straggler_classes;   // lean kernel with merging: (group, member mask) pairs in the straggler list
    unsigned int straggler_overflow;
    unsigned int pad;
    unsigned long long table_inserts;   // records that went to the HBM attractor table (the log was full)
    unsigned int table_overflow;        // ... and did not find a slot there
    unsigned int near_overflow;         // deep cube pass: some workgroup listed more classes than its segment holds
    unsigned long long phase_sum[3];    // diagnostic (BSX_DIAG builds): pool kernel, 100 MHz ticks summed over workgroups: prologue, loop, epilogue
    unsigned long long phase_max[3];    // ... and the slowest workgroup's
    // cube passes of the pool kernel: the workgroups add their LDS accumulators up here (slot = the attractor's
    // position in the cache mirror, the same in every workgroup of a launch) instead of writing log records:
    // the results come back with this struct
    unsigned long long acc_cnt[64], acc_sl[64], acc_sl2_lo[64], acc_sl2_hi[64];
    unsigned int acc_len[64];
    unsigned int acc_key[64][kMaxW32];
    // ... in units of 2^unit_shift problems (the host scales them).  A member that is a cycle state itself (mu = 0, at most one
    // per class) is not a whole unit: its class is booked at mu = 1 and these signed, ABSOLUTE sums move the one problem
    unsigned long long fix_cnt[64], fix_sl[64], fix_sl2[64];   // two's complement
    unsigned long long fix_none, fix_ref, fix_capfail;
    unsigned long long near_classes;    // deep cube pass: classes whose common state F^depth is a cycle state (listed, see AttractParams::near)
    // cube passes: when the first workgroup with work started (stored complemented, so that zero = nobody did) and the last one
    // ended, 100 MHz ticks of the device's constant clock: the launch's own duration without a pair of events around it
    unsigned long long t_first_not, t_last;
    unsigned long long wave_iters;      // diagnostic: loop iterations summed over waves
    unsigned long long service_rounds;  // diagnostic
    unsigned long long diag[4];         // diagnostic (BSX_DIAG builds): pool kernel: classes kept after fresh stages,
                                        // lanes entering pool stages, classes kept after pool stages, lanes merged away
};

// Cache of known cycle states (DESIGN.md "cycle-state cache").  A trajectory enters its attractor at
// time T_p + mu, which is the first time its state is a cycle state; once ALL states of an attractor
// are known, a lookup per step ends the search there (mu steps) instead of running Brent's detector
// and the mu pass (about 2.5 x (mu + lambda) steps).  The first lane to find an attractor appends
// (key, length) to a journal in HBM; every workgroup regenerates the cycle from the key and mirrors
// its states into LDS, making them visible all at once (a partially visible cycle would let a
// trajectory slip past its entry point and report a larger mu).  The cache never changes a result.
constexpr uint32_t kCycleCacheMaxLen = 64;        // longer cycles are not cached
constexpr uint32_t kCycleCacheLdsBytes = 16 * 1024;
constexpr uint32_t kCycleJournalCap = 4096;       // attractors
constexpr uint32_t kCycleClaimSlots = 8192;       // fingerprints of published keys (dedupe)
// Lean attract kernel: results of cached attractors 1..kTagAcc are summed in registers per lane; all
// of 1..kTagAcc+kLdsAcc have per-workgroup LDS accumulators (sum l^2 u64, sum l u64, count u32) next
// to their length and key (for the log records / per-problem records).  Later attractors are left to
// the general kernel.
constexpr int kTagAcc = 3;
constexpr uint32_t kLdsAcc = 61;
// + the per-wave tables of the sibling merge (64 lane ids + 64 member-mask accumulators per wave)
constexpr uint32_t kMergeGroup = 32;            // consecutive problems loaded together; members tracked as a 32-bit mask
constexpr uint32_t kMergeSlots = 256;          // per wave: one-byte lane ids, picked by (group & 3, 6 hash bits)
constexpr uint32_t lean_acc_bytes(uint32_t nw) { return (kTagAcc + kLdsAcc) * (8 + 8 + 4 + 4 + 4 * nw) + 16 + kWavesPerBlock * (kMergeSlots + 256); }

struct CycleRecord {
    uint32_t key[kMaxW32];
    uint32_t length;
    uint32_t ready;             // written last (release)
};
static_assert(sizeof(CycleRecord) == 40, "CycleRecord layout");

struct CycleCache {
    CycleRecord* journal;
    unsigned int* journal_count;
    unsigned int* claims;       // kCycleClaimSlots fingerprints, 0 = free
    uint32_t enabled;
    uint32_t lds_slots;         // power of two; the LDS mirror holds at most lds_slots / 2 states
};

// Hand-over between two levels of a cube cascade, written by k_compact_near, read by the next level's launch.
struct LevelDesc {
    unsigned long long n_entries;   // classes listed by the level above (packed list)
    unsigned int abort;             // a segment of the level above overflowed: the list is incomplete, the block is redone shallower
    unsigned int pad;
};
constexpr uint32_t kMaxCubeLevels = 16;     // levels of one cascade
constexpr uint32_t kSideStreams = 4;        // streams the lower levels of a batch's chains are spread over
constexpr uint32_t kMaxChains = 48;         // cascades (sub-blocks of a split block) enqueued behind one wait
constexpr uint32_t kMaxChainBlocks = 160;   // ... and their levels together = Counters blocks per wait
constexpr size_t kLevelDescBytes = 3584;    // the descriptors' share of the counter buffer (a multiple of the Counters alignment)
constexpr size_t kPublishTicketOffset = kLevelDescBytes - 16;   // k_publish's ticket counter lives at the end of that share
static_assert(sizeof(LevelDesc) * (kMaxChainBlocks + kMaxChains + 1) <= kPublishTicketOffset && kLevelDescBytes % 256 == 0, "descriptor block");

// Depth-1 level of a cascade, evaluated per PARENT instead of per child.  A listed class (parent) expands into 2^kb
// children by the digits this level adds; all children share the parent's state except for those digits, and whether a
// child's F(x) is a cycle state c is decided by the few nodes whose rule reads an added digit (`dep`): every other node has
// the same value for all children -- f(parent)[i] -- which must equal c[i] or no child maps to c.  So a lane takes one
// parent: one update of its representative, a scan of the cached cycle states for those that agree on the independent
// nodes, and for each such state the dependent nodes' truth tables evaluated bit-sliced over the children (32 per word:
// an added digit is a constant bit pattern, a parent bit a broadcast) -- the children that hit are a popcount.  The others
// enter the parent's cycle with the next update (mu = 2), as in the per-child pass.
constexpr uint32_t kLeafMaxDeps = 64;       // dependent nodes the program holds (more: the per-child pass)
constexpr uint32_t kLeafMaxBits = 14;       // digits the level may add (the children are evaluated 512 at a time)
constexpr uint32_t kLeafMaxK = 4;           // inputs of a dependent node's rule
struct LeafDep {
    uint16_t in[kLeafMaxK];     // input j: 0x8000 | q = the level's added digit q (child-index bit q), else the node whose PARENT bit it is
    uint16_t node;              // the node this rule computes
    uint16_t k;                 // inputs, 1 .. kLeafMaxK
    uint32_t tt;                // truth table, bit idx = output when input j is (idx >> j) & 1
};
struct LeafProgram {
    uint32_t kb, n_dep;
    uint32_t indep[kMaxW32];    // node bits whose first update does not depend on an added digit
    uint32_t added[kMaxW32];    // node bits of the added digits
    LeafDep dep[kLeafMaxDeps];
};

struct AttractParams {
    DevNet net;
    DevSpace sp;
    uint64_t count;
    uint32_t chunk;             // problems per dequeue
                                // (pool kernel: per dequeue AFTER the wave's fixed first share, 0 = there is nothing after it)
    uint32_t cap_rel_inf;       // 1: max_t is infinite
    uint64_t max_t;             // absolute cap (valid when !cap_rel_inf)
    uint64_t max_len;           // attractor length cap (UINT64_MAX = none)
    Counters* ctr;
    LogRec* log;
    uint64_t log_cap;
    // HBM attractor table behind the log (store_attractor's dict, attract.py:374-402, for spaces with very
    // many attractors): open addressing over LogRec slots, LogRec::pad = slot state (0 empty, 1 being
    // written, 2 ready).  Null in passes whose results are discarded.
    LogRec* table;
    uint64_t table_mask;        // slots - 1 (power of two)
    ProblemRec32* per_problem;  // nullable, indexed by problem offset
    CycleCache cc;
    const uint32_t* offsets;    // nullable: work item i is problem offsets[i] (straggler pass)
    uint32_t* stragglers;       // fast kernel: offsets of problems that hit no cached cycle state
    uint64_t stragglers_cap;
    uint32_t fast_steps;        // FAST phase length
    uint32_t pad;               // lean kernel: service-lane override (0 = default)
    uint32_t merge;             // lean kernel: merge sibling trajectories that reach the same state (stragglers become pairs)
                                // pool kernel: 1 member masks, 2 member counts, 3 member counts over a cube (below)
    // Cube pass of the pool kernel (DESIGN.md "cube collapse"): the work items are not problems but the
    // 2^r assignments of the RELEVANT digits of an aligned block of 2^a problems -- digits the first update
    // does not depend on are left at 0 and every class starts with 2^(a-r) members (cube_shift = a - r).
    // cube_umask = node bits of those irrelevant free digits: a member whose s(0) is itself a cycle state
    // differs from its class representative only there.  cube_base/cube_free: the block's fixed bits and the
    // mask of all its free node bits (which cached cycle states lie inside the block).
    uint32_t cube_shift;
    uint32_t cube_umask[kMaxW32];
    uint32_t cube_free[kMaxW32];
    // Deep cube pass (DESIGN.md "deeper collapse"): the class digits are those F^depth still depends on, the
    // fresh stage makes `cube_depth` updates before the first lookup, and a class whose common state F^depth(x)
    // is a cached cycle state -- its members enter the cycle at different times <= depth -- is not accounted but
    // listed by its representative's initial state: workgroup g appends to its own segment near[g * near_cap ..]
    // (near_cap states of nw words, a counter in LDS, no global atomics) and leaves its count in near_counts[g];
    // k_compact_near then packs the segments into one list.  Counters::near_classes = the total, near_overflow =
    // a segment was too small.  The host runs the listed classes again one level down:
    // entries != null: work item i is sub-assignment i & (2^entry_shift - 1) of the digits this level adds,
    // on top of the state entries[(i >> entry_shift) * nw ..].
    uint32_t cube_depth;        // >= 1 (1 = one update, then lookups: the plain cube pass)
    uint32_t entry_shift;
    const uint32_t* entries;
    uint32_t* near;
    uint32_t* near_counts;
    uint64_t near_cap;
    // Work distribution of the pool kernel: wave w of the grid starts with [w * chunk_first, (w + 1) * chunk_first);
    // the shared cursor hands out what lies beyond n_waves * chunk_first, `chunk` at a time (chunk == 0: nothing
    // lies beyond).  Same-address atomics complete at some 15 ns apiece however many waves wait, so a pass must
    // not take more than a few hundred of them per 100 us: small cube passes are split evenly up front.
    uint64_t chunk_first;
    // Pool kernel: the LDS cache mirror (header + lds_slots entries) as a ready-made image.  Regenerating the cached
    // cycles from the journal is serial work of one thread per workgroup (8 us for 32 cycle states, far more for a
    // few hundred) that every launch of a cascade repeated; the image is built once per journal state by a
    // one-workgroup launch with mirror_out set (which does nothing else) and copied by all threads afterwards.
    const uint32_t* mirror_image;
    uint32_t* mirror_out;
    // general kernel, discovery from explicit states: work item i starts at states[i * nw ..] (no enumeration)
    const uint32_t* states;
    // Chained cascade (DESIGN.md "levels chained on the device"): a lower level's launch is enqueued before the level
    // above has run, so its number of work items comes from the device: count = level_in->n_entries << entry_shift
    // (nothing to do when that is 0 or level_in->abort is set), and the split over the waves is derived from it there.
    const LevelDesc* level_in;
    uint32_t lower_build;       // cube pass: launch the lower-level build of the kernel (entries != null, the mirror as an image)
    uint32_t pad2;
    const LeafProgram* leaf;    // lower-level build at depth 1: the children of an entry are evaluated bit-sliced (below); else null
};

// Cube collapse, ordering heuristic: how long does a flip of each relevant digit stay visible?  One thread per
// (digit, trial) steps the block's base state with a pseudo-random assignment of the free digits, once with
// the digit clear and once set, until the two trajectories meet (bsx_pool.hip: k_digit_lifetimes).  Digits
// whose influence dies first become the lowest class-index bits, so that the classes that will merge sit
// next to each other.  Only the enumeration ORDER depends on this, never a result.
constexpr uint32_t kLifeTrials = 16;
constexpr uint32_t kLifeSteps = 48;
struct LifetimeParams {
    DevNet net;
    uint32_t fixmask[kMaxW32], fixval[kMaxW32];
    uint32_t base[kMaxW32];         // the block's fixed bits (free bits zero)
    uint32_t free_mask[kMaxW32];    // node bits of its free digits
    uint32_t n_digits;              // <= 64
    uint32_t pad;
    uint32_t node[64];              // node of each relevant digit
    uint32_t* out;                  // [n_digits] sum over the trials of the steps until the flip has died out
};

struct HitRec { uint64_t offset; uint64_t t; };

struct TargetParams {
    DevNet net;
    DevSpace sp;
    uint64_t count;
    uint32_t chunk;
    uint32_t cap_rel_inf;
    uint64_t max_t;
    uint32_t tmask[kMaxW32];
    uint32_t tcode[kMaxW32];
    Counters* ctr;
    uint32_t* t_hit;            // nullable: [count] first hit time per problem, 0xFFFFFFFF = target not reached
    unsigned long long* hist;   // nullable: [hist_bins] hits by first-hit time, last bin = that time or later
    uint32_t hist_bins;         // <= kTargetHistBins (the workgroups count in LDS first, 64-bit counters)
    // Cube pass (summary sink only; DESIGN.md "cube collapse"): work items are the assignments of the relevant
    // digits of an aligned block, each standing for 2^cube_shift problems that share s(1), s(2), ...  Only s(0)
    // differs between them: a member matches the target at t = 0 iff the representative matches on the target
    // bits outside the irrelevant free digits (rep_mask / rep_code) and its own irrelevant digits that lie in
    // the target mask equal the code there -- 2^-cube_t0_shift of the members.
    uint32_t cube;
    uint32_t cube_shift;
    uint32_t cube_t0_shift;
    uint32_t pad;
    uint64_t chunk_first;       // work items every wave starts with (its fixed share; `chunk` at a time from the cursor after that)
    uint32_t rep_mask[kMaxW32];
    uint32_t rep_code[kMaxW32];
};
constexpr uint32_t kTargetHistBins = 2048;

struct SimParams {
    DevNet net;
    DevSpace sp;
    uint64_t count;
    uint64_t max_t;             // used when t_len == nullptr
    uint32_t w64;               // uint64 words per state in the outputs
    uint32_t pad;
    const uint64_t* offsets;    // nullable: problem q = first + offsets[q]  (else first + q)
    const uint64_t* t_len;      // nullable: per-problem length
    const uint64_t* out_offsets;// nullable: word offset of problem q's trajectory
    uint64_t* traj;             // nullable
    uint64_t* final_states;     // nullable
    uint64_t* digests;          // nullable
    Counters* ctr;
};

// Bit-sliced simulate kernel (fixed-length runs, no per-problem variations): one lane owns 32
// trajectories, the state is an n x 64-lane matrix of 32-bit words in LDS, double buffered.
struct SlicedParams {
    DevSpace sp;
    uint32_t n_nodes;
    uint32_t n_rows;            // n_nodes rounded up to the node batch
    uint32_t n_sched;
    uint32_t w64;
    const uint32_t* desc;       // [n_rows][8]: 6 predecessor rows, truth table (64 bits, replicated to 2^K)
    const uint32_t* sched;      // [n_sched][3] (t, node, value) sorted by t
    uint64_t count;
    uint64_t max_t;
    uint64_t* final_states;     // nullable: [count][w64]
    uint64_t* digests;          // nullable: [count] fold digests (second-generation kernel only)
    Counters* ctr;
};

}  // namespace bsx
