/*
 * TEST INFRASTRUCTURE -- CPU restatement ("oracle") of BoolSi's simulate / attract / target
 * state-update path.  Not part of the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it (as the checker / the timed CPU baseline).
 *
 * It follows the reference's algorithm function by function (file:line in /root/reference):
 *   orc_problem_from_index ........ boolsi/batching.py:160-229 (digits -> problem), 10-46 (radices)
 *   step (apply_update_rules) ..... boolsi/model.py:16-28, 52-73 (perturbation override)
 *   fixed nodes ................... boolsi/model.py:31-49
 *   warm-up ....................... boolsi/model.py:76-128
 *   detection loop ................ boolsi/model.py:152-236 (both detectors)
 *   attract solvers ............... boolsi/attract.py:262-302 (all states), 305-371 (reference points)
 *   simulate / target solvers ..... boolsi/simulate.py:97-131, boolsi/target.py:109-133
 *   aggregation ................... boolsi/attract.py:374-402, 35-45 (kept as exact integer sums)
 * Parity is pinned by tests/test_oracle_golden.py against vectors generated from the reference
 * itself (oracle/gen_golden.py -> tests/golden/ (JSON)) and against the reference's own
 * known-answer tests and example outputs.
 */
#ifndef BSX_ORACLE_H
#define BSX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_WORDS 4          /* n <= 256 */
#define ORC_T_INF UINT64_MAX

typedef struct {
    uint32_t n_nodes;
    const uint32_t* pred_offsets;     /* n+1 */
    const uint32_t* pred_idx;         /* ascending per node */
    const uint32_t* tt_word_offsets;  /* n+1 */
    const uint64_t* tt_words;         /* bit idx of node's table: predecessor j has state (idx>>j)&1 */
} orc_network;

typedef struct {
    const uint64_t* origin_state;     /* W words */
    const uint32_t* any_nodes;  uint32_t n_any;
    const uint32_t* fixed;      uint32_t n_fixed;   /* (node, value) */
    const uint32_t* fixed_var;  uint32_t n_fv;      /* (node, range)  range: 0 '0?' 1 '1?' 2 'any' 3 'any?' */
    const uint32_t* sched;      uint32_t n_sched;   /* (t, node, value) */
    const uint32_t* pert_var;   uint32_t n_pv;      /* (t, node, range) */
} orc_space;

typedef struct {
    uint64_t key[ORC_MAX_WORDS];      /* min state code over the cycle */
    uint64_t length;                  /* attractor length (lambda) */
    uint64_t trajectory_l;            /* T_p + mu */
    uint64_t t_stop;                  /* time at which the reference's loop stopped */
    uint32_t found;                   /* 1 = attractor returned by the solver */
    uint32_t pad;
} orc_attr_result;

typedef struct {
    uint64_t key[ORC_MAX_WORDS];
    uint64_t length;
    uint64_t count;
    uint64_t sum_l;
    uint64_t sum_l2_lo, sum_l2_hi;    /* 128-bit sum of squares */
} orc_attr_agg;

typedef struct {
    uint64_t t_stop;
    uint32_t reached;
    uint32_t pad;
} orc_target_result;

/* Problem index I (batching.py mixed-radix number) split at the initial-state digits:
 *   I = init_digits + variant * 2^n_any,  init_digits < 2^n_any (n_any <= 256 binary digits, one per
 *   'any' initial node, least significant first), variant = the fixed-node / perturbation digits. */
typedef struct {
    uint64_t init_digits[ORC_MAX_WORDS];
    uint64_t variant;
} orc_index;

/* Decode one problem: initial_state[W], fixed_mask[W], fixed_val[W]; returns number of perturbation
 * entries written to pert (t,node,value triples, capacity n_sched + n_pv) sorted by t. */
int orc_problem_from_index(const orc_network* net, const orc_space* sp, const orc_index* index,
                           uint64_t* initial_state, uint64_t* fixed_mask, uint64_t* fixed_val,
                           uint32_t* pert, uint32_t* n_pert);

/* One synchronous update (no fixed nodes, no perturbations): model.py:16-28. */
void orc_step(const orc_network* net, const uint64_t* state, uint64_t* next);

/* attract over [first, first+count): per_problem may be NULL; table has capacity cap.
 * storing_all_states = 1: attract.py:262-302;  0: attract.py:305-371 (including its behaviour
 * under a finite max_t).  Returns 0, or -1 if the table overflowed. */
int orc_run_attract(const orc_network* net, const orc_space* sp, const orc_index* first,
                    uint64_t count, uint64_t max_t, uint64_t max_len, int storing_all_states,
                    orc_attr_result* per_problem, orc_attr_agg* table, uint32_t cap, uint32_t* n_out,
                    uint64_t* n_no_attractor, uint64_t* state_steps, int n_threads);

/* target: per problem, reached flag and stop time (target.py:109-133 over model.py:152-236). */
int orc_run_target(const orc_network* net, const orc_space* sp, const orc_index* first,
                   uint64_t count, uint64_t max_t, const uint64_t* mask, const uint64_t* code,
                   orc_target_result* per_problem, uint64_t* state_steps, int n_threads);

/* simulate: states s(0..max_t) of each problem -> traj[(p*(max_t+1) + t)*W + w] (may be NULL),
 * final[p*W + w], digest[p] = fold digest of s(0..max_t): FNV-1a over the words of X = xor of all s(t),
 * Y = xor of the s(t) with ((uint32_t)t * 0x9E3779B1) >> 31 set, and s(max_t). */
int orc_run_simulate(const orc_network* net, const orc_space* sp, const orc_index* first,
                     uint64_t count, uint64_t max_t, uint64_t* traj, uint64_t* final_state,
                     uint64_t* digest, uint64_t* state_steps, int n_threads);

/* trajectory s(0..t_len) of single problems given by index list (for target hits). */
int orc_trajectory(const orc_network* net, const orc_space* sp, const orc_index* index,
                   uint64_t t_len, uint64_t* traj);

#ifdef __cplusplus
}
#endif
#endif
