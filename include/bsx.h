/*
 * bsx.h -- C-ABI of the MI355X (gfx950) Boolean-network state-update engine.
 *
 * This is the drop-in boundary for BoolSi's simulate / attract / target hot path.  The reference
 * (pure Python, /root/reference) has no FFI; the seam it offers is the per-batch function
 *     execute_task(task, solve, store, empty_results, n_to_find)        boolsi/mpi.py:498-544
 * with task = (batch seed, batch index, predecessor_node_lists, truth_tables)
 *                                                                       boolsi/batching.py:313-316
 * and the solver bound by configure_solve_simulation_problem            boolsi/mpi.py:351-376.
 * One bsx_run_* call does the work of execute_task over a contiguous range of the problem
 * index: enumerate -> adjust rules for fixed nodes -> solve -> store/aggregate.
 *
 * Conventions
 *   - plain C, caller-allocated host buffers, no callbacks, no torch / numpy types;
 *   - every function returns BSX_OK (0) or a negative bsx_status; bsx_last_error() gives text;
 *   - one handle per device; a handle is not thread-safe, distinct handles are independent;
 *   - there is NO CPU implementation behind this API: bsx_create fails without a gfx950 device.
 *
 * State layout: a state is W = ceil(n/64) uint64 words, node i = bit (i % 64) of
 * word i / 64, so that the reference's state code (model.py:131-149) is the little-endian
 * big integer formed by the words.
 */
#ifndef BSX_H
#define BSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSX_MAX_NODES        256
#define BSX_MAX_WORDS        4      /* uint64 words per state / key */
#define BSX_MAX_PREDECESSORS 24
#define BSX_MAX_PERT_VARIATIONS 32
#define BSX_T_INF            UINT64_MAX   /* "no cap" for max_t / max_len (the CLI's inf) */

typedef enum {
    BSX_OK = 0,
    BSX_ERR_INVALID = -1,        /* bad argument / inconsistent tables */
    BSX_ERR_NO_DEVICE = -2,      /* no gfx950 GPU, or HIP runtime failure at create */
    BSX_ERR_HIP = -3,            /* HIP call failed (text in bsx_last_error) */
    BSX_ERR_UNSUPPORTED = -4,    /* network / problem space exceeds an engine limit (BSX_MAX_*, variant >= 2^64) */
    BSX_ERR_TABLE_FULL = -5,     /* more distinct attractors / hits than the caller's capacity */
    BSX_ERR_STEP_LIMIT = -6,     /* a trajectory ran into the engine's internal step limit
                                    (only possible when max_t is BSX_T_INF or above that limit) */
    BSX_ERR_STATE = -7,          /* call order: network / problem space not set */
    BSX_ERR_COMM = -8,           /* RCCL: library not loadable, or an ncclXxx call failed */
    BSX_ERR_RANGE_TOO_LARGE = -9 /* the call is valid but its result does not fit THIS entry point's record (a 64-bit
                                    count / sum of a bsx_attr_rec overflowed): use bsx_run_attract2, or a smaller range */
} bsx_status;

/* Variation ranges = boolsi.constants.NodeStateRange (constants.py:23-30), digit -> state as in
 * batching.py:171-175. */
enum {
    BSX_RANGE_MAYBE_FALSE = 0,          /* '0?'   digit 0: absent, 1: 0           radix 2 */
    BSX_RANGE_MAYBE_TRUE = 1,           /* '1?'   digit 0: absent, 1: 1           radix 2 */
    BSX_RANGE_TRUE_OR_FALSE = 2,        /* 'any'  digit 0: 0,      1: 1           radix 2 */
    BSX_RANGE_MAYBE_TRUE_OR_FALSE = 3   /* 'any?' digit 0: absent, 1: 0, 2: 1     radix 3 */
};

typedef struct bsx_engine* bsx_handle;

/* Problem index I of the reference's mixed-radix enumeration (batching.py:10-46, 212-229), split
 * at the initial-state digits:  I = init_digits + variant * 2^n_any.
 *   init_digits: n_any binary digits, digit j = state of the j-th 'any' initial node (node order),
 *                packed little-endian into uint64 words (so I mod 2^n_any);
 *   variant:     the remaining digits (fixed-node variations, then perturbation variations) as one
 *                number, I >> n_any.  Must fit 64 bits. */
typedef struct {
    uint64_t init_digits[BSX_MAX_WORDS];
    uint64_t variant;
} bsx_index;

typedef struct { uint32_t node; uint32_t value; } bsx_fixed;              /* origin fixed node (model.py:31-49) */
typedef struct { uint32_t node; uint32_t range; } bsx_fixed_var;          /* fixed-node variation */
typedef struct { uint32_t t; uint32_t node; uint32_t value; } bsx_pert;   /* origin perturbation at time t >= 1 */
typedef struct { uint32_t t; uint32_t node; uint32_t range; } bsx_pert_var;

/* One aggregated attractor = reference AggregatedAttractor (attract.py:18-45) with the float
 * mean / M2 replaced by exact integer sums: mean = sum_l / count, M2 = sum_l2 - sum_l^2 / count. */
typedef struct {
    uint64_t key[BSX_MAX_WORDS];   /* min state code over the cycle (attract.py:296) */
    uint64_t length;               /* attractor length */
    uint64_t count;                /* frequency */
    uint64_t sum_l;                /* sum of trajectory_l = T_p + mu (attract.py:291-298) */
    uint64_t sum_l2_lo, sum_l2_hi; /* 128-bit sum of trajectory_l^2 */
} bsx_attr_rec;

/* Optional per-problem result of attract (what store_attractor receives, attract.py:374-402). */
typedef struct {
    uint64_t key[BSX_MAX_WORDS];
    uint64_t length;
    uint64_t trajectory_l;
    uint32_t found;                /* 0: no attractor within max_t / max_len */
    uint32_t pad;
} bsx_problem_rec;

/* target: one simulation that reached the target substate (target.py:109-133). */
typedef struct {
    uint64_t offset;               /* problem = first + offset */
    uint64_t t;                    /* first t >= T_p with state & mask == code */
} bsx_hit;

typedef struct {
    uint64_t problems;             /* problems processed */
    uint64_t state_steps;          /* steps the reference algorithm performs for them (S5-S7):
                                      t_stop per problem; simulate: max_t per problem */
    uint64_t executed_steps;       /* network updates the kernels actually executed */
    double   kernel_ms;            /* device time of the hot kernels (HIP events on the engine's stream) */
    double   total_ms;             /* wall time of the call incl. uploads, merge, downloads */
    uint32_t kernel_launches;
    uint32_t pad;
} bsx_stats;

int  bsx_create(bsx_handle* out, int device);
int  bsx_destroy(bsx_handle h);
const char* bsx_last_error(bsx_handle h);       /* h may be NULL: error of the last failed bsx_create */
const char* bsx_status_string(int status);
int  bsx_device_info(bsx_handle h, char* name, uint32_t name_cap, uint32_t* compute_units,
                     uint64_t* global_mem_bytes);

/* How the current network was lowered (valid after bsx_set_network): 32-bit words per state, gathered predecessor
 * slots of the mux tree, and where the gather table lives (0 HBM / L2, 1 LDS per state byte, 2 LDS per 4 state bits).
 * Together they name the kernel instantiations a run launches, e.g. k_attract_pool<words, slots, lut_mode, cube>. */
int  bsx_network_info(bsx_handle h, uint32_t* state_words32, uint32_t* mux_slots, uint32_t* lut_mode);

/* Network = (predecessor_node_lists, truth_tables) of the task tuple (batching.py:313-316).
 * pred_idx: predecessors of node i are pred_idx[pred_offsets[i] .. pred_offsets[i+1]) ascending
 * (input.py:796).  tt_words: table of node i starts at word tt_word_offsets[i] and has 2^k bits;
 * bit idx = output when predecessor j has state (idx >> j) & 1  (SURVEY.md S2). */
int  bsx_set_network(bsx_handle h, uint32_t n_nodes,
                     const uint32_t* pred_offsets, const uint32_t* pred_idx,
                     const uint32_t* tt_word_offsets, const uint64_t* tt_words);

/* Problem space = (origin_simulation_problem, simulation_problem_variations) of the batch seed
 * (batching.py:258-282, input.py:148-157); variation arrays in the reference's list order. */
int  bsx_set_problem_space(bsx_handle h, const uint64_t* origin_state_words,
                           const uint32_t* any_nodes, uint32_t n_any,
                           const bsx_fixed* fixed, uint32_t n_fixed,
                           const bsx_fixed_var* fixed_var, uint32_t n_fixed_var,
                           const bsx_pert* sched, uint32_t n_sched,
                           const bsx_pert_var* pert_var, uint32_t n_pert_var);

/* attract (attract.py:262-302 semantics for every problem of [first, first + count)):
 * aggregated table (unordered) into table[0..*n_out), problems without attractor counted in
 * *n_no_attractor.  per_problem (count entries) may be NULL.
 * A record's 64-bit count / sum_l and 128-bit sum_l2 must hold the result: BSX_ERR_RANGE_TOO_LARGE otherwise
 * (bsx_run_attract2 has wide records); count <= 2^32 with per_problem.  Large aligned ranges are not enumerated
 * problem by problem: the engine steps classes of problems that provably share their trajectory from some update
 * on (DESIGN.md "Cube collapse"). */
int  bsx_run_attract(bsx_handle h, const bsx_index* first, uint64_t count,
                     uint64_t max_t, uint64_t max_len,
                     bsx_attr_rec* table, uint32_t cap, uint32_t* n_out,
                     uint64_t* n_no_attractor, bsx_problem_rec* per_problem, bsx_stats* stats);

/* ---- attract over ranges of any size ("bsx_u128 first, count: N = 2^a * 3^b can exceed 2^64") ----------
 * The reference's N is a Python int (input.py:899-928) and ONE attract_master run covers it (attract.py:67-230);
 * bsx_run_attract2 is that call.  first / count are the flat problem index I of batching.py:212-229 and a number of
 * consecutive indices, 128 bits each; every sum of the record is wide enough for count = 2^128 problems of
 * trajectory_l < 2^64 each.  Words are little-endian (word 0 least significant). */
typedef struct { uint64_t lo, hi; } bsx_u128;

typedef struct {
    uint64_t key[BSX_MAX_WORDS];   /* min state code over the cycle (attract.py:296) */
    uint64_t length;               /* attractor length */
    bsx_u128 count;                /* frequency */
    uint64_t sum_l[3];             /* 192-bit sum of trajectory_l */
    uint64_t sum_l2[4];            /* 256-bit sum of trajectory_l^2 */
} bsx_attr_rec2;

typedef struct {
    bsx_u128 problems;             /* as bsx_stats, wide */
    bsx_u128 state_steps;
    uint64_t executed_steps;
    double   kernel_ms;            /* device time of all kernels of the call (HIP events on the engine's stream) */
    double   total_ms;
    double   dominant_ms;          /* ... of the dominant launches alone: the top level of every cube cascade */
    uint64_t dominant_executed_steps;  /* network updates those launches executed (the roofline's numerator) */
    uint32_t dominant_launches;
    uint32_t kernel_launches;
    uint32_t host_syncs;           /* times the host waited for the device inside the call */
    uint32_t lower_launches;       /* launches of the cascades' lower levels that had classes to work on ... */
    double   lower_ms;             /* ... their device time (first workgroup in to last one out, the device's 100 MHz clock) */
    uint64_t lower_executed_steps; /* ... and the network updates they executed */
} bsx_stats2;

/* Same semantics and table contents as bsx_run_attract (which stays, for ranges whose sums fit 64 bits).
 * A flat 128-bit index reaches every problem of spaces up to 2^128 problems; larger ones (more than 128 'any'
 * nodes) are reached through bsx_index with bsx_run_attract.  BSX_ERR_INVALID if first + count leaves the space. */
int  bsx_run_attract2(bsx_handle h, bsx_u128 first, bsx_u128 count,
                      uint64_t max_t, uint64_t max_len,
                      bsx_attr_rec2* table, uint32_t cap, uint32_t* n_out,
                      bsx_u128* n_no_attractor, bsx_stats2* stats);

/* attract through the functional graph (no reference analogue: the reference re-simulates every trajectory,
 * mpi.py:521-538): for spaces whose n <= 32 nodes are all 'any' (no variations, no perturbations) the network
 * is a function on 2^n states; successor array, pointer doubling and pointer jumping over 2^n-sized arrays in
 * HBM give every problem's (key, length, trajectory_l) without stepping trajectories.  Same results as
 * bsx_run_attract; other spaces return BSX_ERR_UNSUPPORTED. */
int  bsx_run_attract_fgraph(bsx_handle h, const bsx_index* first, uint64_t count,
                            uint64_t max_t, uint64_t max_len,
                            bsx_attr_rec* table, uint32_t cap, uint32_t* n_out,
                            uint64_t* n_no_attractor, bsx_stats* stats);

/* target (target.py:109-133): hits in ascending offset order into hits[0..*n_hits); mask/code are W words
 * (target node set / target substate code, input.py:580-661). */
int  bsx_run_target(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                    const uint64_t* mask_words, const uint64_t* code_words,
                    bsx_hit* hits, uint64_t cap, uint64_t* n_hits, bsx_stats* stats);

/* target with an on-device summary instead of the full hit list, for sweeps whose hits cannot (and need
 * not) all travel to the host: *n_hits = number of problems that reach the target; hist[b] = those whose
 * first hit is at t == b for b < hist_bins - 1, hist[hist_bins - 1] = at that time or later (hist_bins may
 * be 0, at most 2048); hits[0..*n_listed) = the FIRST min(*n_hits, cap) hits in ascending offset order --
 * what the reference's -n option keeps (simulate.py:163-164, mpi.py:537-538); cap may be 0. */
int  bsx_run_target_summary(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                            const uint64_t* mask_words, const uint64_t* code_words,
                            uint64_t* hist, uint32_t hist_bins, bsx_hit* hits, uint64_t cap,
                            uint64_t* n_hits, uint64_t* n_listed, bsx_stats* stats);

/* simulate (simulate.py:97-131 == s(0..max_t) by plain stepping): any of the three sinks may
 * be NULL.  trajectories[(p * (max_t + 1) + t) * W + w], final_states[p * W + w], digests[p].
 * The digest is a checksum of the whole trajectory for runs too long to store (no reference counterpart):
 * FNV-1a (seed 0xCBF29CE484222325, prime 0x100000001B3) over the W words of X, then of Y, then of s(max_t),
 * where X = xor of s(0..max_t) and Y = xor of the s(t) with ((uint32_t)t * 0x9E3779B1) >> 31 == 1.  XORs of
 * whole states, so the bit-sliced kernels keep it per node row. */
int  bsx_run_simulate(bsx_handle h, const bsx_index* first, uint64_t count, uint64_t max_t,
                      uint64_t* trajectories, uint64_t* final_states, uint64_t* digests,
                      bsx_stats* stats);

/* trajectories s(0..t_len[q]) of listed problems (first + offsets[q]); used to materialise the
 * simulations of target hits.  out[q] starts at out_offsets[q] words. */
int  bsx_run_trajectories(bsx_handle h, const bsx_index* first, const uint64_t* offsets,
                          const uint64_t* t_len, uint64_t n, uint64_t* out,
                          const uint64_t* out_offsets, bsx_stats* stats);

/* Blocks until all work of the handle's stream is done (bench.py's timing fence). */
int  bsx_synchronize(bsx_handle h);

/* Multi-GPU merge step: one process per GPU, the problem index range-partitioned by the caller
 * ([r*N/G, (r+1)*N/G) for rank r), and ONE all-gather of the per-rank result tables at the end --
 * the counterpart of the reference master collecting its workers' batch results (mpi.py:290-330).
 * The collective is ncclAllGather (RCCL over xGMI) on the handle's stream.  The caller moves the
 * BSX_COMM_ID_BYTES unique id from rank 0 to the other ranks (any side channel: file, socket);
 * librccl.so is loaded on the first of these calls.
 *   bsx_comm_unique_id   rank 0 only: a fresh id (ncclGetUniqueId)
 *   bsx_comm_init        every rank, collectively: communicator of `world` ranks on the handle's device
 *   bsx_comm_allgather   every rank, collectively: recv[r * bytes_per_rank ..) = rank r's send buffer
 *                        (host buffers; staged through HBM by the library)
 *   bsx_comm_destroy     releases the communicator (bsx_destroy does it too) */
#define BSX_COMM_ID_BYTES 128
int  bsx_comm_unique_id(bsx_handle h, void* out, uint32_t cap);
int  bsx_comm_init(bsx_handle h, const void* unique_id, uint32_t id_bytes, int rank, int world);
int  bsx_comm_allgather(bsx_handle h, const void* send, uint64_t bytes_per_rank, void* recv);
int  bsx_comm_destroy(bsx_handle h);

#ifdef __cplusplus
}
#endif
#endif /* BSX_H */
