"""
GPU parity: the HIP path (through the C-ABI, boolsi_amd.engine) against
  (1) the golden vectors produced by the reference itself (tests/golden/*.json),
  (2) the CPU oracle on larger seeded slices,
  (3) size-independent properties and the reference's published full-size result
      (cambium2: 39 attractors with exact basin sizes over 2^30 initial states).
Bit-exact everywhere (integer / bitwise path); the only float comparison is the derived
trajectory-length mean against the reference's CSV (rel 1e-9).
"""
import csv
import math
import os
import random

import numpy as np
import pytest

from boolsi_amd import synth
from boolsi_amd.compile import code_to_words, words_to_code, compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.input import parse_input_text
from util import load, t_of, compile_case, contiguous_runs, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['cycle_cache', 'lean_merge', 'lean_plain', 'no_cycle_cache'])
def eng(request):
    """Every test runs four times: with the cycle-state cache and the class-pool kernel (trajectories end
    at mu on a cached cycle state; sibling trajectories that reach the same state are stepped once), with
    the lean kernel and its in-lane merge, with the lean kernel without merging, and without the cache
    (every trajectory runs Brent's detector + the mu pass)."""
    from boolsi_amd.engine import Engine
    os.environ['BSX_CYCLE_CACHE'] = '0' if request.param == 'no_cycle_cache' else '1'
    e = Engine(0)
    e.cycle_cache = request.param != 'no_cycle_cache'
    os.environ.pop('BSX_CYCLE_CACHE')
    if request.param in ('lean_merge', 'lean_plain'):     # read per attract call; default = class-pool kernel
        os.environ['BSX_MERGE'] = '1' if request.param == 'lean_merge' else '0' 
    yield e
    os.environ.pop('BSX_MERGE', None)
    e.close()


def key_int(k):
    from boolsi_amd.engine import key_to_int
    return key_to_int(k)


ATTRACT_CASES = [c for f in ('attract_toy.json', 'attract_examples.json', 'attract_synth.json') for c in load(f)]
# the reference's -r solver under a finite -t is defective: parity for -r is
# claimed only without a time cap, where it equals the default detector
ATTRACT_CASES = [c for c in ATTRACT_CASES if c['storing_all_states'] or c['max_t'] is None]


def merge_tables(tables):
    merged = {}
    for table in tables:
        for a in table:
            e = merged.setdefault(str(key_int(a['key'])), [int(a['length']), 0, 0, 0])
            e[1] += int(a['count'])
            e[2] += int(a['sum_l'])
            e[3] += int(a['sum_l2_lo']) + (int(a['sum_l2_hi']) << 64)
    order = sorted(merged.items(), key=lambda kv: (-kv[1][1], int(kv[0])))
    return [[k, v[0], v[1], v[2], str(v[3])] for k, v in order]


@pytest.mark.parametrize('case', ATTRACT_CASES, ids=lambda c: c['name'])
def test_attract_golden(eng, case):
    _, net, space = compile_case(case)
    eng.set_problem(net, space)
    idx = [int(i) for i in case['indices']]
    rows = [None] * len(idx)
    tables, none, steps = [], 0, 0
    for first, count, off in contiguous_runs(idx):
        r = eng.attract(first, count, t_of(case['max_t']), t_of(case['max_len']), per_problem=True)
        tables.append(r.table)
        none += r.n_no_attractor
        steps += r.stats['state_steps']
        for q in range(count):
            p = r.per_problem[q]
            rows[off + q] = [int(p['found']), str(key_int(p['key'])), int(p['length']), int(p['trajectory_l'])]
    assert rows == [r[:4] for r in case['per_problem']]
    assert merge_tables(tables) == case['aggregate']
    assert none == sum(1 for r in case['per_problem'] if not r[0])
    if case['storing_all_states']:
        assert steps == sum(r[4] for r in case['per_problem'])      # reference loop stop times


@pytest.mark.parametrize('case', load('target.json'), ids=lambda c: c['name'])
def test_target_golden(eng, case):
    cfg, net, space = compile_case(case)
    eng.set_problem(net, space)
    mask = code_to_words(sum(1 << n for n in cfg['target node set']), net.n_words)
    code = code_to_words(cfg['target substate code'], net.n_words)
    idx = [int(i) for i in case['indices']]
    rows = [[0, None] for _ in idx]
    for first, count, off in contiguous_runs(idx):
        hits, _ = eng.target(first, count, t_of(case['max_t']), mask, code)
        for h in hits:
            rows[off + int(h['offset'])] = [1, int(h['t'])]
    for got, ref in zip(rows, case['per_problem']):
        assert got[0] == ref[0]
        if ref[0]:
            assert got[1] == ref[1]
    for i, states in case['trajectories'].items():
        trajs, _ = eng.trajectories(int(i), [0], [len(states) - 1])
        assert [str(words_to_code(s)) for s in trajs[0]] == states


@pytest.mark.parametrize('case', load('simulate.json'), ids=lambda c: c['name'])
def test_simulate_golden(eng, case):
    _, net, space = compile_case(case)
    eng.set_problem(net, space)
    idx = [int(i) for i in case['indices']]
    for first, count, off in contiguous_runs(idx):
        traj, final, digest, st = eng.simulate(first, count, case['max_t'])
        assert st['state_steps'] == count * case['max_t']
        for q in range(count):
            assert str(words_to_code(final[q])) == case['final'][off + q]
            assert str(int(digest[q])) == case['digest'][off + q]
            key = str(idx[off + q])
            if key in case['trajectories']:
                assert [str(words_to_code(s)) for s in traj[q]] == case['trajectories'][key]


# ---------------------------------------------------------------------------------------------------
# larger seeded slices against the CPU oracle

def _setup(eng, text, mode, max_t):
    from oracle.cpu_oracle import Oracle
    cfg = parse_input_text(text, max_t, mode)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    return cfg, net, space, Oracle(net, space)


def _same_attract(eng, orc, first, count, max_t, max_len=None):
    from oracle.cpu_oracle import key_int as okey
    r = eng.attract(first, count, t_of(max_t), t_of(max_len), per_problem=True)
    pp, table, none, steps = orc.attract(first, count, max_t, max_len, True, n_threads=8)
    assert np.array_equal(r.per_problem['found'], pp['found'])
    assert np.array_equal(r.per_problem['key'], pp['key'])
    assert np.array_equal(r.per_problem['length'], pp['length'])
    assert np.array_equal(r.per_problem['trajectory_l'], pp['trajectory_l'])
    assert r.n_no_attractor == none
    assert r.stats['state_steps'] == steps
    ours = sorted((key_int(a['key']), int(a['length']), int(a['count']), int(a['sum_l']), int(a['sum_l2_lo'])) for a in r.table)
    ref = sorted((okey(a['key']), int(a['length']), int(a['count']), int(a['sum_l']), int(a['sum_l2_lo'])) for a in table)
    assert ours == ref
    return r


SLICE_CASES = [
    ('northstar_n64', synth.north_star_yaml(), 64, 14),
    ('config3_n32', synth.config3_yaml(), 32, 14),
    ('cambium2', open(os.path.join(GOLDEN, 'cambium2.yaml')).read(), 30, 14),
    ('synth_n128', synth.network_yaml(128, 2, 129), 128, 14),
    ('synth_n256_k3', synth.network_yaml(256, 3, 256), 256, 12),      # (chaotic: transients of thousands of steps)
    ('synth_n48_k6', synth.network_yaml(48, 6, 48), 48, 13),
]


@pytest.mark.parametrize('name,text,space_bits,log2n', SLICE_CASES, ids=[c[0] for c in SLICE_CASES])
def test_attract_vs_oracle_slices(eng, name, text, space_bits, log2n):
    cfg, net, space, orc = _setup(eng, text, Mode.ATTRACT, 4096)
    rng = random.Random(sum(map(ord, name)))
    _same_attract(eng, orc, 0, 1 << log2n, 4096)
    first = rng.randrange((1 << space_bits) - (1 << 14))
    _same_attract(eng, orc, first, (1 << log2n) + 77, 4096)       # ragged count, unaligned base
    _same_attract(eng, orc, first, 1000, 9, 3)                    # tight caps: many "no attractor"
    _same_attract(eng, orc, first, 1, 4096)                       # single problem
    r = eng.attract(first, 0, 4096)                               # empty range
    assert len(r.table) == 0 and r.n_no_attractor == 0


def test_attract_wide_rules_vs_oracle(eng):
    # nodes with 9 predecessors take the explicit-lookup path (k > 6)
    text = synth.network_yaml(24, 9, 924)
    cfg, net, space, orc = _setup(eng, text, Mode.ATTRACT, 2000)
    _same_attract(eng, orc, 0, 1 << 12, 2000)
    _same_attract(eng, orc, 12345, 3000, 50, 4)


def test_target_vs_oracle_config4(eng):
    cfg, net, space, orc = _setup(eng, synth.config4_yaml(), Mode.TARGET, 1024)
    mask = code_to_words(sum(1 << n for n in cfg['target node set']), net.n_words)
    code = code_to_words(cfg['target substate code'], net.n_words)
    rng = random.Random(44)
    for first, count in ((0, 1 << 14), (rng.randrange(1 << 31 - 1), 20000), ((1 << 28) * 5 - 5000, 10000)):
        count = min(count, space.n_problems - first)
        hits, _ = eng.target(first, count, 1024, mask, code)
        pp, _ = orc.target(first, count, 1024, mask, code, n_threads=8)
        ref = [(q, int(pp[q]['t_stop'])) for q in range(count) if pp[q]['reached']]
        assert [(int(h['offset']), int(h['t'])) for h in hits] == ref


def test_simulate_vs_oracle_config5(eng):
    cfg, net, space, orc = _setup(eng, synth.config5_yaml(max_t=600, n_any=12), Mode.SIMULATE, 600)
    traj, final, digest, _ = eng.simulate(100, 1500, 600, trajectories=False)
    _, ofinal, odigest, _ = orc.simulate(100, 1500, 600, want_traj=False, n_threads=8)
    assert np.array_equal(final, ofinal)
    assert np.array_equal(digest, odigest)
    traj, final, digest, _ = eng.simulate(7, 9, 600)
    otraj, _, _, _ = orc.simulate(7, 9, 600)
    assert np.array_equal(traj, otraj)


# ---------------------------------------------------------------------------------------------------
# full-size runs: published result and invariants

def test_cambium2_full_sweep_matches_published_output(eng):
    """examples/output8_cambium2/attractor_summaries.csv: 39 attractors, basin sizes = rel_freq * 2^30."""
    cfg, net, space, _ = _setup(eng, open(os.path.join(GOLDEN, 'cambium2.yaml')).read(), Mode.ATTRACT, math.inf)
    assert space.n_problems == 1 << 30
    r = eng.attract(0, 1 << 30)
    assert r.n_no_attractor == 0
    ours = sorted(r.table, key=lambda a: (-int(a['count']), key_int(a['key'])))
    with open(os.path.join(GOLDEN, 'cambium2_attractor_summaries.csv')) as f:
        ref = list(csv.DictReader(f))
    with open(os.path.join(GOLDEN, 'cambium2_attractors.csv')) as f:
        states = list(csv.reader(f))
    assert len(ours) == len(ref) == 39
    assert sum(int(a['count']) for a in ours) == 1 << 30
    first_state = {row[0]: row[2:] for row in states[1:] if row[1] == 't'}
    for rank, (a, row) in enumerate(zip(ours, ref), 1):
        assert int(a['length']) == int(row['length'])
        assert int(a['count']) == round(float(row['relative_frequency']) * (1 << 30))
        assert int(a['count']) / (1 << 30) == float(row['relative_frequency'])
        mean = int(a['sum_l']) / int(a['count'])
        assert math.isclose(mean, float(row['trajectory_length_mean']), rel_tol=1e-9)
        c, s1, s2 = int(a['count']), int(a['sum_l']), int(a['sum_l2_lo']) + (int(a['sum_l2_hi']) << 64)
        sd = math.sqrt((s2 - s1 * s1 / c) / (c - 1))
        assert math.isclose(sd, float(row['trajectory_length_SD']), rel_tol=1e-7)
        # key state = first printed state of the attractor ('0_' marks the fixed node)
        bits = [int(x.rstrip('_')) for x in first_state['attractor{}'.format(rank)]]
        assert key_int(a['key']) == sum(b << i for i, b in enumerate(bits))


def test_partition_invariance_and_conservation(eng):
    """Range partitioning (the multi-GPU scheme) must not change the merged table."""
    cfg, net, space, _ = _setup(eng, synth.north_star_yaml(), Mode.ATTRACT, 4096)
    first, count = 0x123456789ABCDEF0, 1 << 22
    whole = eng.attract(first, count, 4096)
    parts = [eng.attract(first + i * (count // 4), count // 4, 4096) for i in range(4)]
    assert merge_tables([whole.table]) == merge_tables([p.table for p in parts])
    assert whole.n_no_attractor == sum(p.n_no_attractor for p in parts)
    assert sum(int(a['count']) for a in whole.table) + whole.n_no_attractor == count
    assert whole.stats['state_steps'] == sum(p.stats['state_steps'] for p in parts)
    # every reported key is a state of a cycle of the reported length (idempotence of the result)
    from oracle.cpu_oracle import Oracle
    orc = Oracle(net, space)
    for a in whole.table:
        s0 = np.array(a['key'][:net.n_words], np.uint64)
        s = s0.copy()
        seen_min = words_to_code(s0)
        for _ in range(int(a['length'])):
            s = orc.step(s)
            seen_min = min(seen_min, words_to_code(s))
        assert np.array_equal(s, s0)
        assert seen_min == words_to_code(s0)


# ---------------------------------------------------------------------------------------------------
# paths of the attract orchestration that the synthetic sweeps do not reach

def _chain_yaml(n=64):
    """x0' = 0, x_i' = x_(i-1): the transient is the position of the highest set bit + 1, the only
    attractor is the all-off fixed point.  Long transients = stragglers of the lean kernel."""
    lines = ['nodes:'] + ['    - x{}'.format(i) for i in range(n)]
    lines += ['update rules:', "    x0: '0'"] + ['    x{}: x{}'.format(i, i - 1) for i in range(1, n)]
    lines += ['initial state:'] + ['    x{}: any'.format(i) for i in range(n)]
    return '\n'.join(lines) + '\n'


def test_lean_kernel_stragglers_and_fallback(eng):
    cfg, net, space, orc = _setup(eng, _chain_yaml(), Mode.ATTRACT, math.inf)
    _same_attract(eng, orc, 0, 1 << 16, None)                       # discovery + lean kernel, no stragglers
    first = (1 << 48) - 58982
    r = _same_attract(eng, orc, first, 1 << 16, None)               # ~10 % of the problems need > 48 steps
    assert r.stats['kernel_launches'] in (1, 2, 3)
    _same_attract(eng, orc, (1 << 48) - (1 << 15), 1 << 16, None)   # 50 % stragglers: list overflows, fallback
    _same_attract(eng, orc, (1 << 60), 1 << 14, None)               # after the fallback the detector is used
    _same_attract(eng, orc, (1 << 40), 1 << 14, 30)                 # time cap below the transient: none found
    _same_attract(eng, orc, (1 << 20), 1 << 14, 30, 1)


def test_counting_pass_aborts_and_is_repeated(eng):
    """A counting pass (member counts, no masks) that meets classes it cannot resolve is dropped and the
    tile repeated with member masks; forced here on a space with far more attractors than the mirror holds."""
    from oracle.cpu_oracle import key_int as okey
    cfg, net, space, orc = _setup(eng, _identity_yaml(14, period2=[(2, 11)]), Mode.ATTRACT, math.inf)
    os.environ['BSX_FORCE_COUNTING'] = '1'
    try:
        for first, count in ((0, 1 << 14), (100, 12000)):
            r = eng.attract(first, count)
            pp, table, none, steps = orc.attract(first, count, None, None, True, n_threads=8)
            assert _table(r.table, key_int) == _table(table, okey)
            assert r.n_no_attractor == none and r.stats['state_steps'] == steps
            if first == 0 and eng.cycle_cache and os.environ.get('BSX_MERGE', '2') == '2':
                # discovery sample, counting pass, repeat with masks, detector on the stragglers (after that the
                # space is marked as not covered by the cache and the detector serves it alone)
                assert r.stats['kernel_launches'] >= 4
    finally:
        os.environ.pop('BSX_FORCE_COUNTING')


def test_long_transients_raise_the_fast_length(eng):
    """Production call (no per-problem records) on a network whose transients outlast the FAST length: the
    first probe tile comes back as stragglers that all end on a cached cycle state, the FAST length is raised,
    and later calls are resolved by the pool kernel alone.  Tables must equal the oracle's throughout."""
    from oracle.cpu_oracle import key_int as okey
    cfg, net, space, orc = _setup(eng, _chain_yaml(), Mode.ATTRACT, math.inf)
    for first, count in ((0, 1 << 16),                       # short transients: calibrates the lean path
                         ((1 << 48) - 58982, 1 << 16),       # ~10 % of the problems need more than 48 steps
                         ((1 << 56) - 40000, 1 << 16),       # again, after the FAST length has been raised
                         ((1 << 40), 1 << 15)):
        r = eng.attract(first, count)
        pp, table, none, steps = orc.attract(first, count, None, None, True, n_threads=8)
        assert _table(r.table, key_int) == _table(table, okey)
        assert r.n_no_attractor == none and r.stats['state_steps'] == steps


def _identity_yaml(n, period2=()):
    """x_i' = x_i (every state a fixed point), except the pairs in `period2`, which swap (cycles of length 2)."""
    rule = {i: i for i in range(n)}
    for a, b in period2:
        rule[a], rule[b] = b, a
    lines = ['nodes:'] + ['    - x{}'.format(i) for i in range(n)]
    lines += ['update rules:'] + ['    x{}: x{}'.format(i, rule[i]) for i in range(n)]
    lines += ['initial state:'] + ['    x{}: any'.format(i) for i in range(n)]
    return '\n'.join(lines) + '\n'


def test_more_attractors_than_the_lean_kernel_resolves(eng):
    # 2^14 attractors: the cycle cache holds a few hundred states, the lean kernel's accumulators 64
    # attractors; everything else must come back through the straggler list / the detector
    cfg, net, space, orc = _setup(eng, _identity_yaml(14), Mode.ATTRACT, math.inf)
    r = _same_attract(eng, orc, 0, 1 << 14, None)
    assert len(r.table) == 1 << 14
    # 2^12 fixed points + (2^14 - 2^12) / 2 two-cycles: keys are the smaller state of each pair
    cfg, net, space, orc = _setup(eng, _identity_yaml(14, period2=[(3, 9)]), Mode.ATTRACT, math.inf)
    r = _same_attract(eng, orc, 0, 1 << 14, None)
    assert len(r.table) == (1 << 13) + (1 << 12)
    _same_attract(eng, orc, 0, 1 << 14, None)          # again, with whatever the cache learnt


def test_scattered_any_nodes_take_the_lean_path(eng):
    # 'any' nodes in five runs (one crossing the 32-bit word boundary, so six deposit runs), the rest
    # constants: digits are deposited run by run; results must equal the oracle's bit-by-bit enumeration
    any_nodes = set(range(2, 10)) | set(range(26, 37)) | {40} | set(range(44, 52)) | set(range(60, 64))
    bits = synth.seeded_bits(64, 6401)
    text = synth.network_yaml(64, 2, 64, initial={i: str(bits[i]) for i in range(64) if i not in any_nodes})
    cfg, net, space, orc = _setup(eng, text, Mode.ATTRACT, 4096)
    assert space.n_problems == 1 << len(any_nodes)
    r = _same_attract(eng, orc, 0, 1 << 15, 4096)
    if eng.cycle_cache:
        assert r.stats['kernel_launches'] >= 2          # discovery sample + lean kernel (+ stragglers)
    _same_attract(eng, orc, (1 << 31) + 777, 1 << 15, 4096)
    _same_attract(eng, orc, space.n_problems - 20000, 20000, 4096)
    # too many runs for a deposit plan (every other node): generic enumeration, same answers
    text = synth.network_yaml(64, 2, 64, initial={i: str(bits[i]) for i in range(64) if i % 2})
    cfg, net, space, orc = _setup(eng, text, Mode.ATTRACT, 4096)
    _same_attract(eng, orc, 12345, 1 << 14, 4096)
    # target mode shares the enumeration
    tm, tc = code_to_words(0b1011 << 20, net.n_words), code_to_words(0b1001 << 20, net.n_words)
    hits, _ = eng.target(999, 1 << 14, 200, tm, tc)
    ref, _ = orc.target(999, 1 << 14, 200, tm, tc, n_threads=8)
    want = np.nonzero(ref['reached'])[0]
    assert np.array_equal(hits['offset'], want) and np.array_equal(hits['t'], ref['t_stop'][want])


def _table(rows, key_fn):
    return sorted((key_fn(a['key']), int(a['length']), int(a['count']), int(a['sum_l']), int(a['sum_l2_lo'])) for a in rows)


@pytest.mark.parametrize('seed', range(int(os.environ.get('BSX_FUZZ_SEEDS', '20'))))
def test_random_networks_aggregate_tables(eng, seed):
    """The production call (no per-problem records) on random networks: ordered and chaotic rules, scattered
    'any' nodes, constant fixed nodes, origin perturbations; ranges that are no multiples of the 64-problem
    groups of the pool kernel.  Tables, no-attractor counts and reference step counts must equal the oracle's."""
    from oracle.cpu_oracle import key_int as okey
    rng = random.Random(9000 + seed)
    n = rng.choice([12, 20, 31, 32, 40, 64, 64, 70, 100])
    k = rng.choice([1, 2, 2, 2, 3])
    n_any = min(n, rng.randint(14, 22))
    any_nodes = set(rng.sample(range(n), n_any)) if seed % 3 else set(range(n_any))
    bits = synth.seeded_bits(n, 500 + seed)
    fixed = {i: str(rng.getrandbits(1)) for i in rng.sample(range(n), rng.randint(0, 3))} if seed % 2 else None
    pert = {rng.randrange(n): {str(rng.getrandbits(1)): '2, 4'}} if seed % 4 == 1 else None
    text = synth.network_yaml(n, k, 7000 + seed, initial={i: str(bits[i]) for i in range(n) if i not in any_nodes},
                              fixed=fixed, perturbations=pert)
    # no time cap only where cycles are sure to be short (a chaotic network can have astronomically long ones:
    # the reference, the oracle and the detector would all run until their step limits)
    max_t = rng.choice([math.inf, 300, 40]) if (k == 1 or (k == 2 and n <= 40)) else rng.choice([300, 40])
    cfg, net, space, orc = _setup(eng, text, Mode.ATTRACT, max_t)
    total = space.n_problems
    for first, count in ((0, min(total, (1 << 14) + 37)), (max(0, total - 30011), min(total, 30011))):
        for _ in range(2):          # second pass: whatever the cache learnt is in use
            r = eng.attract(first, count, t_of(None if max_t is math.inf else max_t))
            pp, table, none, steps = orc.attract(first, count, None if max_t is math.inf else max_t, None, True, n_threads=8)
            assert _table(r.table, key_int) == _table(table, okey)
            assert r.n_no_attractor == none and r.stats['state_steps'] == steps


def test_attract_with_fixed_node_variations_uses_detector_only(eng):
    # not reachable through the YAML front end (attract forbids variations) but allowed by the C-ABI:
    # cycles differ per fixed-node variant, so the cycle cache must stay out of it
    text = synth.network_yaml(40, 2, 77, initial={i: str(i & 1) for i in range(14, 40)},
                              fixed={3: 'any?', 17: '0?', 21: 'any'},
                              perturbations={5: {'1': '2, 6-7', 'any?': '9'}, 30: {'0?': '3'}})
    cfg, net, space, orc = _setup(eng, text, Mode.SIMULATE, 4096)   # parsed in a mode that allows variations
    assert space.n_problems == (1 << 14) * 3 * 2 * 2 * 3 * 2
    _same_attract(eng, orc, 0, 1 << 15, 4096)
    _same_attract(eng, orc, space.n_problems - 40000, 40000, 4096)
    _same_attract(eng, orc, 123456, 5000, 12, 2)


def test_largest_network_and_rule_width(eng):
    # n = 256 (8 words, LUT read through L2), K = 6 mux tree, plus a 12-input rule on the wide path
    text = synth.network_yaml(256, 6, 2566)
    cfg, net, space, orc = _setup(eng, text, Mode.ATTRACT, 64)
    _same_attract(eng, orc, (1 << 255) + 12345, 3000, 64)
    text = synth.network_yaml(20, 12, 2012)
    cfg, net, space, orc = _setup(eng, text, Mode.ATTRACT, 5000)
    _same_attract(eng, orc, 0, 1 << 13, 5000)


@pytest.mark.parametrize('lut_mode', ['0', '1', '2'])
def test_lut_modes_agree_with_the_oracle(eng, lut_mode):
    """The gather LUT read through L2 (0), from LDS per state byte (1) and from LDS per 4 state bits (2)."""
    bits = synth.seeded_bits(100, 1003)
    text = synth.network_yaml(100, 2, 1002, initial={i: str(bits[i]) for i in range(36, 100)})
    os.environ['BSX_LUT_MODE'] = lut_mode
    try:
        cfg, net, space, orc = _setup(eng, text, Mode.ATTRACT, 3000)
    finally:
        os.environ.pop('BSX_LUT_MODE')
    _same_attract(eng, orc, 0, 1 << 14, 3000)               # discovery + lean kernel
    _same_attract(eng, orc, (1 << 35) + 99, 20000, 3000)
    _same_attract(eng, orc, 5, 3000, 40, 3)                 # general kernel only (below the lean threshold), caps
    tm, tc = code_to_words(0b111 << 40, net.n_words), code_to_words(0b101 << 40, net.n_words)
    hits, _ = eng.target(77, 1 << 13, 300, tm, tc)
    ref, _ = orc.target(77, 1 << 13, 300, tm, tc, n_threads=8)
    want = np.nonzero(ref['reached'])[0]
    assert np.array_equal(hits['offset'], want) and np.array_equal(hits['t'], ref['t_stop'][want])
    traj, fin, dig, _ = eng.simulate(123, 300, 50)
    otraj, ofin, odig, _ = orc.simulate(123, 300, 50, n_threads=8)
    assert np.array_equal(traj, otraj) and np.array_equal(fin, ofin) and np.array_equal(dig, odig)


# ---------------------------------------------------------------------------------------------------
# bit-sliced simulate kernel (final states of long fixed-length runs)

@pytest.mark.parametrize('name,text,max_t,first,count', [
    ('config5_like', synth.config5_yaml(max_t=700, n_any=14), 700, 100, 5000),          # ragged last group
    ('n200_k2', synth.network_yaml(200, 2, 2002), 100, (1 << 150) + 7, 4096),
    ('n40_k5_fixed_pert', synth.network_yaml(40, 5, 40, initial={i: ('any' if i % 3 else str(i & 1)) for i in range(40)},
                                             fixed={3: '1', 17: '0'},
                                             perturbations={5: {'1': '2, 6-7, 90'}, 30: {'0': '3, 64'}}), 128, 12345, 6000),
    ('n64_k1', synth.network_yaml(64, 1, 641), 64, 0, 2048),
    ('n48_k3_odd_t', synth.network_yaml(48, 3, 483, fixed={7: '1'}, perturbations={9: {'0': '5, 33'}}), 101, 3, 9000),
    ('n20_k2', synth.network_yaml(20, 2, 202), 33, 1000, 4097),
    ('n128_k3', synth.network_yaml(128, 3, 1283), 257, (1 << 100) + 1, 8192),
    ('n250_k4', synth.network_yaml(250, 4, 2504), 70, (1 << 249) - 5000, 3000),
], ids=lambda v: v if isinstance(v, str) and len(v) < 30 else None)
def test_sliced_simulate_matches_oracle_and_per_lane_kernel(eng, name, text, max_t, first, count):
    cfg, net, space, orc = _setup(eng, text, Mode.SIMULATE, max_t)
    _, final, _, st = eng.simulate(first, count, max_t, trajectories=False, digest=False)     # sliced kernel
    _, ofinal, _, _ = orc.simulate(first, count, max_t, want_traj=False, n_threads=8)
    assert np.array_equal(final, ofinal)
    assert st['state_steps'] == count * max_t
    os.environ['BSX_SLICED'] = '0'
    try:
        _, final2, _, _ = eng.simulate(first, count, max_t, trajectories=False, digest=False)  # per-lane kernel
    finally:
        os.environ.pop('BSX_SLICED')
    assert np.array_equal(final, final2)
    os.environ['BSX_SLICED'] = '1'            # first-generation sliced kernel (what n > 128 or K > 3 get anyway)
    try:
        _, final3, _, _ = eng.simulate(first, count, max_t, trajectories=False, digest=False)
    finally:
        os.environ.pop('BSX_SLICED')
    assert np.array_equal(final, final3)
    # digests: per-lane kernel and (where the shape allows: K <= 3, n <= 128) the bit-sliced kernel against the oracle
    _, ofinal, odigest, _ = orc.simulate(first, count, max_t, want_traj=False, n_threads=8)
    _, dfinal, ddigest, _ = eng.simulate(first, count, max_t, trajectories=False, digest=True)
    assert np.array_equal(dfinal, ofinal) and np.array_equal(ddigest, odigest)
    _, _, ddigest2, _ = eng.simulate(first, count, max_t, trajectories=False, final=False, digest=True)   # digest sink alone
    assert np.array_equal(ddigest2, odigest)
    os.environ['BSX_SLICED'] = '0'
    try:
        _, _, ddigest3, _ = eng.simulate(first, count, max_t, trajectories=False, digest=True)
    finally:
        os.environ.pop('BSX_SLICED')
    assert np.array_equal(ddigest3, odigest)
