/*
 * TEST INFRASTRUCTURE -- see bsx_oracle.h.  Plain C restatement of the reference algorithm,
 * written for clarity and checkability, not speed: one truth-table lookup per node per step,
 * the full state history for the default detector, exactly as the Python does it.
 */
#include "bsx_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define W_OF(n) (((n) + 63u) >> 6)

static inline int get_bit(const uint64_t* s, uint32_t node) { return (int)((s[node >> 6] >> (node & 63)) & 1u); }
static inline void put_bit(uint64_t* s, uint32_t node, int v) {
    uint64_t m = (uint64_t)1 << (node & 63);
    if (v) s[node >> 6] |= m; else s[node >> 6] &= ~m;
}
/* compare state codes as big integers (word W-1 most significant) */
static inline int code_cmp(const uint64_t* a, const uint64_t* b, uint32_t W) {
    for (int w = (int)W - 1; w >= 0; --w) {
        if (a[w] < b[w]) return -1;
        if (a[w] > b[w]) return 1;
    }
    return 0;
}
static inline int code_eq(const uint64_t* a, const uint64_t* b, uint32_t W) { return memcmp(a, b, W * 8) == 0; }

/* ---- model.py:16-28  apply_update_rules: next[i] = TT_i[(state[p] for p in preds_i)] ---- */
void orc_step(const orc_network* net, const uint64_t* state, uint64_t* next) {
    uint32_t W = W_OF(net->n_nodes);
    uint64_t out[ORC_MAX_WORDS] = {0, 0, 0, 0};
    for (uint32_t node = 0; node < net->n_nodes; ++node) {
        uint32_t idx = 0;
        uint32_t b = net->pred_offsets[node], e = net->pred_offsets[node + 1];
        for (uint32_t j = b; j < e; ++j) idx |= (uint32_t)get_bit(state, net->pred_idx[j]) << (j - b);
        const uint64_t* tt = net->tt_words + net->tt_word_offsets[node];
        if ((tt[idx >> 6] >> (idx & 63)) & 1u) out[node >> 6] |= (uint64_t)1 << (node & 63);
    }
    memcpy(next, out, W * 8);
}

/* ---- model.py:31-49 + 52-73: rules, fixed nodes as constant rules, then perturbation override ---- */
static void step_problem(const orc_network* net, const uint64_t* fixed_mask, const uint64_t* fixed_val,
                         const uint32_t* pert, uint32_t n_pert, uint32_t* pert_pos, uint64_t t_next,
                         uint64_t* state) {
    uint32_t W = W_OF(net->n_nodes);
    uint64_t nx[ORC_MAX_WORDS];
    orc_step(net, state, nx);
    for (uint32_t w = 0; w < W; ++w) nx[w] = (nx[w] & ~fixed_mask[w]) | (fixed_val[w] & fixed_mask[w]);
    while (*pert_pos < n_pert && pert[3 * *pert_pos] < t_next) ++*pert_pos;
    while (*pert_pos < n_pert && pert[3 * *pert_pos] == t_next) {
        put_bit(nx, pert[3 * *pert_pos + 1], (int)pert[3 * *pert_pos + 2]);
        ++*pert_pos;
    }
    memcpy(state, nx, W * 8);
}

/* ---- batching.py:212-229 (number -> digits), 160-209 (digits -> problem) ---- */
static int cmp_pert(const void* a, const void* b) {
    const uint32_t* x = (const uint32_t*)a; const uint32_t* y = (const uint32_t*)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
    return 0;
}

/* digit -> node state, batching.py:171-175: -1 = absent */
static int digit_state(uint32_t range, uint32_t digit) {
    switch (range) {
        case 0: return digit == 0 ? -1 : 0;           /* '0?'   {None, False} */
        case 1: return digit == 0 ? -1 : 1;           /* '1?'   {None, True} */
        case 2: return digit == 0 ? 0 : 1;            /* 'any'  {False, True} */
        default: return digit == 0 ? -1 : (digit == 1 ? 0 : 1);  /* 'any?' {None, False, True} */
    }
}

int orc_problem_from_index(const orc_network* net, const orc_space* sp, const orc_index* index,
                           uint64_t* initial_state, uint64_t* fixed_mask, uint64_t* fixed_val,
                           uint32_t* pert, uint32_t* n_pert) {
    uint32_t W = W_OF(net->n_nodes);
    uint64_t idx = index->variant;     /* digits above the initial-state digits */
    memcpy(initial_state, sp->origin_state, W * 8);
    memset(fixed_mask, 0, W * 8);
    memset(fixed_val, 0, W * 8);
    /* least significant digits first: one binary digit per varied initial node */
    for (uint32_t j = 0; j < sp->n_any; ++j)
        put_bit(initial_state, sp->any_nodes[j], (int)((index->init_digits[j >> 6] >> (j & 63)) & 1));
    for (uint32_t j = sp->n_any; j < 64 * ORC_MAX_WORDS; ++j)
        if ((index->init_digits[j >> 6] >> (j & 63)) & 1) return -1;
    for (uint32_t j = 0; j < sp->n_fixed; ++j) {
        put_bit(fixed_mask, sp->fixed[2 * j], 1);
        put_bit(fixed_val, sp->fixed[2 * j], (int)sp->fixed[2 * j + 1]);
    }
    for (uint32_t j = 0; j < sp->n_fv; ++j) {
        uint32_t node = sp->fixed_var[2 * j], range = sp->fixed_var[2 * j + 1];
        uint32_t radix = range == 3 ? 3 : 2;
        int st = digit_state(range, (uint32_t)(idx % radix));
        idx /= radix;
        if (st >= 0) { put_bit(fixed_mask, node, 1); put_bit(fixed_val, node, st); }
    }
    uint32_t np = 0;
    for (uint32_t j = 0; j < sp->n_sched; ++j) {
        pert[3 * np] = sp->sched[3 * j]; pert[3 * np + 1] = sp->sched[3 * j + 1]; pert[3 * np + 2] = sp->sched[3 * j + 2];
        ++np;
    }
    for (uint32_t j = 0; j < sp->n_pv; ++j) {
        uint32_t t = sp->pert_var[3 * j], node = sp->pert_var[3 * j + 1], range = sp->pert_var[3 * j + 2];
        uint32_t radix = range == 3 ? 3 : 2;
        int st = digit_state(range, (uint32_t)(idx % radix));
        idx /= radix;
        if (st < 0) continue;
        /* perturbed_nodes_by_t[t][node] = state: overrides an origin entry of the same (t, node) */
        uint32_t q;
        for (q = 0; q < np; ++q) if (pert[3 * q] == t && pert[3 * q + 1] == node) break;
        pert[3 * q] = t; pert[3 * q + 1] = node; pert[3 * q + 2] = (uint32_t)st;
        if (q == np) ++np;
    }
    qsort(pert, np, 3 * sizeof(uint32_t), cmp_pert);
    *n_pert = np;
    return idx == 0 ? 0 : -1;   /* index beyond the problem space */
}

/* ------------------------------------------------------------------ per-thread scratch */
typedef struct {
    uint64_t* hist;      /* codes since last perturbation, W words each */
    uint64_t hist_cap;
    uint64_t* set;       /* open-addressing table of indices into hist (+1), 0 = empty */
    uint64_t set_cap;
    uint32_t* pert;
} scratch;

static void scratch_init(scratch* s, const orc_space* sp) {
    s->hist_cap = 1024; s->hist = (uint64_t*)malloc(s->hist_cap * ORC_MAX_WORDS * 8);
    s->set_cap = 4096; s->set = (uint64_t*)calloc(s->set_cap, 8);
    s->pert = (uint32_t*)malloc((size_t)(sp->n_sched + sp->n_pv + 1) * 3 * sizeof(uint32_t));
}
static void scratch_free(scratch* s) { free(s->hist); free(s->set); free(s->pert); }

static uint64_t hash_code(const uint64_t* c, uint32_t W) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (uint32_t w = 0; w < W; ++w) { h ^= c[w]; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 31; }
    return h;
}
/* returns index of an equal earlier code, or -1 after inserting position `pos` */
static int64_t set_find_or_add(scratch* s, uint32_t W, uint64_t pos) {
    if ((pos + 1) * 2 > s->set_cap) {          /* grow + rehash */
        uint64_t ncap = s->set_cap * 4;
        uint64_t* nset = (uint64_t*)calloc(ncap, 8);
        for (uint64_t i = 0; i < pos; ++i) {
            uint64_t h = hash_code(s->hist + i * W, W) & (ncap - 1);
            while (nset[h]) h = (h + 1) & (ncap - 1);
            nset[h] = i + 1;
        }
        free(s->set); s->set = nset; s->set_cap = ncap;
    }
    const uint64_t* c = s->hist + pos * W;
    uint64_t h = hash_code(c, W) & (s->set_cap - 1);
    while (s->set[h]) {
        if (code_eq(s->hist + (s->set[h] - 1) * W, c, W)) return (int64_t)(s->set[h] - 1);
        h = (h + 1) & (s->set_cap - 1);
    }
    s->set[h] = pos + 1;
    return -1;
}
static void hist_push(scratch* s, uint32_t W, uint64_t pos, const uint64_t* code) {
    if (pos >= s->hist_cap) {
        s->hist_cap *= 2;
        s->hist = (uint64_t*)realloc(s->hist, s->hist_cap * ORC_MAX_WORDS * 8);
    }
    memcpy(s->hist + pos * W, code, W * 8);
}

/* ------------------------------------------------------------------ model.py:152-236 */
typedef struct {
    uint64_t t;              /* time when stopped */
    uint64_t last_pert_t;    /* T_p */
    int attractor_found;
    int target_reached;
    uint64_t n_hist;         /* all-states mode: number of codes since T_p (t - T_p + 1) */
    uint64_t last_ref_t;     /* reference-point mode: time of last reference point */
    uint64_t state[ORC_MAX_WORDS];   /* s(t) */
    uint64_t steps;
} loop_result;

typedef struct { uint64_t t; uint64_t s[ORC_MAX_WORDS]; } ref_point;

static int target_hit(const uint64_t* s, const uint64_t* mask, const uint64_t* code, uint32_t W) {
    if (!mask) return 0;     /* attract / simulate: target_substate_code is None */
    for (uint32_t w = 0; w < W; ++w) if ((s[w] & mask[w]) != code[w]) return 0;
    return 1;
}

static void simulate_until(const orc_network* net, const uint64_t* init, const uint64_t* fmask,
                           const uint64_t* fval, const uint32_t* pert, uint32_t n_pert,
                           int storing_all_states, uint64_t max_t, const uint64_t* tmask,
                           const uint64_t* tcode, scratch* sc, ref_point** refs, uint32_t* n_refs,
                           uint32_t* refs_cap, loop_result* r) {
    uint32_t W = W_OF(net->n_nodes);
    uint64_t s[ORC_MAX_WORDS];
    memcpy(s, init, W * 8);
    uint32_t ppos = 0;
    uint64_t Tp = n_pert ? pert[3 * (n_pert - 1)] : 0;   /* model.py:125 */
    r->steps = 0;
    for (uint64_t t = 1; t <= Tp; ++t) { step_problem(net, fmask, fval, pert, n_pert, &ppos, t, s); ++r->steps; }
    r->last_pert_t = Tp;

    uint64_t L = 0, next_ref_t = 0;
    uint64_t refcode[ORC_MAX_WORDS];
    if (storing_all_states) {
        memset(sc->set, 0, sc->set_cap * 8);
        hist_push(sc, W, 0, s);
        set_find_or_add(sc, W, 0);
    } else {
        *n_refs = 0;
        (*refs)[0].t = Tp; memcpy((*refs)[0].s, s, W * 8); *n_refs = 1;
        memcpy(refcode, s, W * 8);
        L = (net->n_nodes + 1) / 2;                      /* model.py:195 */
        next_ref_t = Tp + L;
    }
    uint64_t t = Tp, n_hist = 1;
    int found = 0, reached = target_hit(s, tmask, tcode, W);   /* model.py:200 */
    while (t < max_t && !found && !reached) {
        ++t;
        step_problem(net, fmask, fval, pert, n_pert, &ppos, t, s); ++r->steps;
        reached = target_hit(s, tmask, tcode, W);
        if (storing_all_states) {
            hist_push(sc, W, n_hist, s);
            found = set_find_or_add(sc, W, n_hist) >= 0;
            ++n_hist;
        } else {
            found = code_eq(s, refcode, W);
            if (t == next_ref_t && !found) {             /* model.py:223-228 */
                if (*n_refs == *refs_cap) { *refs_cap *= 2; *refs = (ref_point*)realloc(*refs, *refs_cap * sizeof(ref_point)); }
                (*refs)[*n_refs].t = t; memcpy((*refs)[*n_refs].s, s, W * 8); ++*n_refs;
                memcpy(refcode, s, W * 8);
                L *= 2;
                next_ref_t = Tp + L;
            }
        }
    }
    r->t = t; r->attractor_found = found; r->target_reached = reached; r->n_hist = n_hist;
    r->last_ref_t = storing_all_states ? 0 : (*refs)[*n_refs - 1].t;
    memcpy(r->state, s, W * 8);
}

/* ------------------------------------------------------------------ attract.py:262-302 */
static void solve_all_states(const orc_network* net, scratch* sc, const loop_result* lr, uint64_t max_len,
                             orc_attr_result* out) {
    uint32_t W = W_OF(net->n_nodes);
    memset(out, 0, sizeof(*out));
    out->t_stop = lr->t;
    if (!lr->attractor_found) return;
    const uint64_t* last = sc->hist + (lr->n_hist - 1) * W;
    uint64_t mu = 0;                                   /* codes.index(last code) */
    while (!code_eq(sc->hist + mu * W, last, W)) ++mu;
    uint64_t lam = lr->n_hist - (mu + 1);
    if (lam > max_len) return;
    const uint64_t* best = sc->hist + (lr->n_hist - lam) * W;
    for (uint64_t i = lr->n_hist - lam + 1; i < lr->n_hist; ++i)
        if (code_cmp(sc->hist + i * W, best, W) < 0) best = sc->hist + i * W;
    memcpy(out->key, best, W * 8);
    out->length = lam;
    out->trajectory_l = lr->t - lam;                   /* len(states) - (lam + 1), absolute time */
    out->found = 1;
}

/* ------------------------------------------------------------------ attract.py:305-371 */
static void solve_reference_points(const orc_network* net, const uint64_t* fmask, const uint64_t* fval,
                                   const loop_result* lr, const ref_point* refs, uint32_t n_refs,
                                   uint64_t max_t, uint64_t max_len, scratch* sc, orc_attr_result* out,
                                   uint64_t* steps) {
    uint32_t W = W_OF(net->n_nodes);
    uint32_t dummy_pos = 0;
    memset(out, 0, sizeof(*out));
    out->t_stop = lr->t;
    uint64_t lam = lr->t - refs[n_refs - 1].t;          /* attract.py:332-333, flag not consulted */
    if (lam > max_len) return;
    uint64_t s[ORC_MAX_WORDS];
    memcpy(s, lr->state, W * 8);
    uint64_t n_codes = 1;
    hist_push(sc, W, 0, s);
    for (uint64_t i = 1; i < lam; ++i) {                /* range(attractor_l - 1) */
        step_problem(net, fmask, fval, NULL, 0, &dummy_pos, 0, s); ++*steps;
        hist_push(sc, W, n_codes++, s);
    }
    /* last reference point whose code is not among the attractor codes */
    int64_t pick = -1;
    for (int64_t q = (int64_t)n_refs - 1; q >= 0 && pick < 0; --q) {
        int in_set = 0;
        for (uint64_t i = 0; i < n_codes && !in_set; ++i) in_set = code_eq(sc->hist + i * W, refs[q].s, W);
        if (!in_set) pick = q;
    }
    uint64_t traj_l;
    if (pick < 0) {
        traj_l = refs[0].t;
    } else {
        uint64_t t = refs[pick].t;
        memcpy(s, refs[pick].s, W * 8);
        for (;;) {
            int in_set = 0;
            for (uint64_t i = 0; i < n_codes && !in_set; ++i) in_set = code_eq(sc->hist + i * W, s, W);
            if (in_set) break;
            ++t;
            step_problem(net, fmask, fval, NULL, 0, &dummy_pos, 0, s); ++*steps;
        }
        traj_l = t;
    }
    if (max_t != ORC_T_INF && traj_l + lam > max_t) return;
    const uint64_t* best = sc->hist;
    for (uint64_t i = 1; i < n_codes; ++i) if (code_cmp(sc->hist + i * W, best, W) < 0) best = sc->hist + i * W;
    memcpy(out->key, best, W * 8);
    out->length = n_codes;        /* len(attractor_states); equals lam except for lam == 0 */
    out->trajectory_l = traj_l;
    out->found = 1;
}

/* ------------------------------------------------------------------ aggregation (attract.py:374-402) */
typedef struct { orc_attr_agg* e; uint32_t cap; uint32_t n; int overflow; } agg_table;

static void agg_add(agg_table* T, uint32_t W, const uint64_t* key, uint64_t length, uint64_t count,
                    uint64_t sum_l, unsigned __int128 sum_l2) {
    for (uint32_t i = 0; i < T->n; ++i) {
        if (code_eq(T->e[i].key, key, W)) {
            T->e[i].count += count; T->e[i].sum_l += sum_l;
            unsigned __int128 q = ((unsigned __int128)T->e[i].sum_l2_hi << 64) | T->e[i].sum_l2_lo;
            q += sum_l2;
            T->e[i].sum_l2_lo = (uint64_t)q; T->e[i].sum_l2_hi = (uint64_t)(q >> 64);
            return;
        }
    }
    if (T->n == T->cap) { T->overflow = 1; return; }
    orc_attr_agg* a = &T->e[T->n++];
    memset(a, 0, sizeof(*a));
    memcpy(a->key, key, W * 8);
    a->length = length; a->count = count; a->sum_l = sum_l;
    a->sum_l2_lo = (uint64_t)sum_l2; a->sum_l2_hi = (uint64_t)(sum_l2 >> 64);
}

/* index + d, where the low n_any bits are init_digits and the rest is `variant` */
static orc_index index_add(const orc_index* first, uint64_t d, uint32_t n_any) {
    uint64_t v[ORC_MAX_WORDS + 2] = {0, 0, 0, 0, 0, 0};
    uint32_t sw = n_any >> 6, sb = n_any & 63;
    for (uint32_t w = 0; w < ORC_MAX_WORDS; ++w) v[w] = first->init_digits[w];
    v[sw] |= first->variant << sb;
    if (sb) v[sw + 1] |= first->variant >> (64 - sb);
    unsigned __int128 carry = d;
    for (uint32_t w = 0; w < ORC_MAX_WORDS + 2 && carry; ++w) {
        carry += v[w];
        v[w] = (uint64_t)carry;
        carry >>= 64;
    }
    orc_index r;
    r.variant = (v[sw] >> sb) | (sb ? (v[sw + 1] << (64 - sb)) : 0);
    for (uint32_t w = 0; w < ORC_MAX_WORDS; ++w) {
        if (w < sw) r.init_digits[w] = v[w];
        else if (w == sw && sb) r.init_digits[w] = v[w] & (((uint64_t)1 << sb) - 1);
        else r.init_digits[w] = 0;
    }
    return r;
}

int orc_run_attract(const orc_network* net, const orc_space* sp, const orc_index* first,
                    uint64_t count, uint64_t max_t, uint64_t max_len, int storing_all_states,
                    orc_attr_result* per_problem, orc_attr_agg* table, uint32_t cap, uint32_t* n_out,
                    uint64_t* n_no_attractor, uint64_t* state_steps, int n_threads) {
    uint32_t W = W_OF(net->n_nodes);
    agg_table G = {table, cap, 0, 0};
    uint64_t none_total = 0, steps_total = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        scratch sc; scratch_init(&sc, sp);
        uint32_t refs_cap = 64, n_refs = 0;
        ref_point* refs = (ref_point*)malloc(refs_cap * sizeof(ref_point));
        agg_table L = {(orc_attr_agg*)malloc((size_t)cap * sizeof(orc_attr_agg)), cap, 0, 0};
        uint64_t none = 0, steps = 0;
#pragma omp for schedule(dynamic, 256)
        for (uint64_t p = 0; p < count; ++p) {
            uint64_t init[ORC_MAX_WORDS], fm[ORC_MAX_WORDS], fv[ORC_MAX_WORDS];
            uint32_t n_pert;
            orc_index ix = index_add(first, p, sp->n_any);
            orc_problem_from_index(net, sp, &ix, init, fm, fv, sc.pert, &n_pert);
            loop_result lr;
            simulate_until(net, init, fm, fv, sc.pert, n_pert, storing_all_states, max_t, NULL, NULL,
                           &sc, &refs, &n_refs, &refs_cap, &lr);
            steps += lr.steps;
            orc_attr_result res;
            if (storing_all_states) solve_all_states(net, &sc, &lr, max_len, &res);
            else solve_reference_points(net, fm, fv, &lr, refs, n_refs, max_t, max_len, &sc, &res, &steps);
            if (per_problem) per_problem[p] = res;
            if (res.found)
                agg_add(&L, W, res.key, res.length, 1, res.trajectory_l,
                        (unsigned __int128)res.trajectory_l * res.trajectory_l);
            else
                ++none;
        }
#pragma omp critical
        {
            for (uint32_t i = 0; i < L.n; ++i)
                agg_add(&G, W, L.e[i].key, L.e[i].length, L.e[i].count, L.e[i].sum_l,
                        ((unsigned __int128)L.e[i].sum_l2_hi << 64) | L.e[i].sum_l2_lo);
            if (L.overflow) G.overflow = 1;
            none_total += none; steps_total += steps;
        }
        free(L.e); free(refs); scratch_free(&sc);
    }
    *n_out = G.n;
    if (n_no_attractor) *n_no_attractor = none_total;
    if (state_steps) *state_steps = steps_total;
    return G.overflow ? -1 : 0;
}

int orc_run_target(const orc_network* net, const orc_space* sp, const orc_index* first,
                   uint64_t count, uint64_t max_t, const uint64_t* mask, const uint64_t* code,
                   orc_target_result* per_problem, uint64_t* state_steps, int n_threads) {
    uint64_t steps_total = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        scratch sc; scratch_init(&sc, sp);
        uint32_t refs_cap = 4, n_refs = 0;
        ref_point* refs = (ref_point*)malloc(refs_cap * sizeof(ref_point));
        uint64_t steps = 0;
#pragma omp for schedule(dynamic, 256)
        for (uint64_t p = 0; p < count; ++p) {
            uint64_t init[ORC_MAX_WORDS], fm[ORC_MAX_WORDS], fv[ORC_MAX_WORDS];
            uint32_t n_pert;
            orc_index ix = index_add(first, p, sp->n_any);
            orc_problem_from_index(net, sp, &ix, init, fm, fv, sc.pert, &n_pert);
            loop_result lr;   /* target.py:69 -> always storing all states */
            simulate_until(net, init, fm, fv, sc.pert, n_pert, 1, max_t, mask, code, &sc, &refs, &n_refs,
                           &refs_cap, &lr);
            steps += lr.steps;
            per_problem[p].reached = (uint32_t)lr.target_reached;   /* target.py:130-133 */
            per_problem[p].t_stop = lr.t;
            per_problem[p].pad = 0;
        }
#pragma omp critical
        steps_total += steps;
        free(refs); scratch_free(&sc);
    }
    if (state_steps) *state_steps = steps_total;
    return 0;
}

static inline uint64_t digest_step(uint64_t d, uint64_t word) { return (d ^ word) * 0x100000001B3ull; }
#define DIGEST_SEED 0xCBF29CE484222325ull

int orc_run_simulate(const orc_network* net, const orc_space* sp, const orc_index* first,
                     uint64_t count, uint64_t max_t, uint64_t* traj, uint64_t* final_state,
                     uint64_t* digest, uint64_t* state_steps, int n_threads) {
    uint32_t W = W_OF(net->n_nodes);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        uint32_t* pert = (uint32_t*)malloc((size_t)(sp->n_sched + sp->n_pv + 1) * 3 * sizeof(uint32_t));
#pragma omp for schedule(dynamic, 64)
        for (uint64_t p = 0; p < count; ++p) {
            uint64_t s[ORC_MAX_WORDS], fm[ORC_MAX_WORDS], fv[ORC_MAX_WORDS];
            uint32_t n_pert, ppos = 0;
            orc_index ix = index_add(first, p, sp->n_any);
            orc_problem_from_index(net, sp, &ix, s, fm, fv, pert, &n_pert);
            /* fold digest (not a reference quantity; include/bsx.h): X = xor of all s(t), Y = xor of the s(t)
             * with ybit(t), then FNV-1a over the words of X, Y and s(max_t) */
            uint64_t x[ORC_MAX_WORDS] = {0}, y[ORC_MAX_WORDS] = {0};
            /* simulate.py:97-131 == plain stepping s(0..max_t) */
            for (uint64_t t = 0;; ++t) {
                if (traj) memcpy(traj + (p * (max_t + 1) + t) * W, s, W * 8);
                const int yb = (int)(((uint32_t)t * 0x9E3779B1u) >> 31);
                for (uint32_t w = 0; w < W; ++w) { x[w] ^= s[w]; if (yb) y[w] ^= s[w]; }
                if (t == max_t) break;
                step_problem(net, fm, fv, pert, n_pert, &ppos, t + 1, s);
            }
            if (final_state) memcpy(final_state + p * W, s, W * 8);
            if (digest) {
                uint64_t d = DIGEST_SEED;
                for (uint32_t w = 0; w < W; ++w) d = digest_step(d, x[w]);
                for (uint32_t w = 0; w < W; ++w) d = digest_step(d, y[w]);
                for (uint32_t w = 0; w < W; ++w) d = digest_step(d, s[w]);
                digest[p] = d;
            }
        }
        free(pert);
    }
    if (state_steps) *state_steps = count * max_t;
    return 0;
}

int orc_trajectory(const orc_network* net, const orc_space* sp, const orc_index* index,
                   uint64_t t_len, uint64_t* traj) {
    uint32_t W = W_OF(net->n_nodes);
    uint32_t* pert = (uint32_t*)malloc((size_t)(sp->n_sched + sp->n_pv + 1) * 3 * sizeof(uint32_t));
    uint64_t s[ORC_MAX_WORDS], fm[ORC_MAX_WORDS], fv[ORC_MAX_WORDS];
    uint32_t n_pert, ppos = 0;
    int rc = orc_problem_from_index(net, sp, index, s, fm, fv, pert, &n_pert);
    for (uint64_t t = 0;; ++t) {
        memcpy(traj + t * W, s, W * 8);
        if (t == t_len) break;
        step_problem(net, fm, fv, pert, n_pert, &ppos, t + 1, s);
    }
    free(pert);
    return rc;
}
