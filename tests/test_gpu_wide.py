"""
bsx_run_attract2 (128-bit first / count, wide sums) and the production regime it exists for:
blocks of 2^48 .. 2^63 problems run as ONE chain of launches.  Exactness is pinned without the oracle's 2^25 ceiling
by networks whose answer is known in closed form at n = 64 (counts, sum of trajectory lengths and of their squares to
the last digit), through the engine's own depth choice and forced depths; the north star by agreement between the
one-call sweep, the old entry point block by block, and per-problem oracle records on sampled sub-ranges.
"""
import os
import random

import numpy as np
import pytest

from boolsi_amd import synth
from boolsi_amd.attract import merge_tables, record_ints
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.input import parse_input_text

pytestmark = pytest.mark.gpu
CORES = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
KNOBS = ('BSX_CUBES', 'BSX_CUBE_DEPTH', 'BSX_CUBE_NEAR_CAP', 'BSX_CUBE_LOWER', 'BSX_CUBE_LEAF', 'BSX_SPIN_WAIT', 'BSX_CUBE_SPLIT', 'BSX_CUBE_STREAMS')


@pytest.fixture()
def eng():
    from boolsi_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    for k in KNOBS:
        os.environ.pop(k, None)


def setup(eng, text, max_t=np.inf):
    cfg = parse_input_text(text, max_t, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    return net, space


def rows(table):
    return sorted(record_ints(a) for a in table)


def yaml_of(names, rules):
    lines = ['nodes:'] + ['    - ' + v for v in names] + ['', 'update rules:']
    lines += ['    {}: {}'.format(v, rules[v]) for v in names]
    lines += ['', 'initial state:'] + ['    {}: any'.format(v) for v in names]
    return '\n'.join(lines) + '\n'


ZERO = '{0} and not {0}'


# ---- closed forms at n = 64 ---------------------------------------------------------------------------------------

def chain_yaml(n=64, chain=16):
    """Nodes 0..chain-1: x_i <- x_(i+1), the chain's last node and every other node <- 0.  One fixed point (0).
    mu(s) = 1 + highest set chain bit if a chain bit is set, else 1 if any other bit is set, else 0."""
    names = ['x{}'.format(i) for i in range(n)]
    rules = {v: ZERO.format(v) for v in names}
    for i in range(chain - 1):
        rules[names[i]] = names[i + 1]
    return yaml_of(names, rules)


def chain_sums(first, log2_count, chain=16):
    """(count, sum mu, sum mu^2) over the aligned block [first, first + 2^log2_count) of the chain network."""
    a = log2_count
    assert first % (1 << a) == 0 and a >= chain
    high_nonzero = (first >> a) != 0
    rest_bits = a - chain                       # free non-chain digits
    per_c = 1 << rest_bits                      # states per chain value
    s1 = sum((b + 1) << b for b in range(chain)) * per_c
    s2 = sum(((b + 1) ** 2) << b for b in range(chain)) * per_c
    # chain == 0: mu = 1 unless every other bit is 0 as well
    ones = per_c if high_nonzero else per_c - 1
    return 1 << a, s1 + ones, s2 + ones


@pytest.mark.parametrize('depth', [None, '1', '4', '8'])
def test_chain_network_blocks_of_2p48_and_the_whole_space(eng, depth):
    if depth:
        os.environ['BSX_CUBE_DEPTH'] = depth
    setup(eng, chain_yaml())
    for first in (0, 5 << 48, (1 << 64) - (1 << 48)):
        got = eng.attract(first, 1 << 48)
        assert got.n_no_attractor == 0 and len(got.table) == 1
        key, length, count, s1, s2 = rows(got.table)[0]
        assert (key, length) == (0, 1)
        assert (count, s1, s2) == chain_sums(first, 48)
    # 2^56 and the whole 2^64 space, one call each
    for first, a in ((3 << 56, 56), (0, 63), (1 << 63, 63)):
        got = eng.attract2(first, 1 << a)
        assert rows(got.table) == [(0, 1) + chain_sums(first, a)]
    whole = eng.attract2(0, 1 << 64)
    c0, c1 = chain_sums(0, 63), chain_sums(1 << 63, 63)
    assert rows(whole.table) == [(0, 1, c0[0] + c1[0], c0[1] + c1[1], c0[2] + c1[2])]
    assert whole.n_no_attractor == 0 and whole.stats['problems'] == 1 << 64
    assert whole.stats['state_steps'] == c0[1] + c1[1] + (1 << 64)          # t_stop = mu + lambda per problem
    # a time cap inside the transients: found iff mu + 1 <= max_t
    setup(eng, chain_yaml(), 9)
    got = eng.attract2(0, 1 << 64, 9)
    found = 1 + sum(1 << b for b in range(8)) * (1 << 48) + ((1 << 48) - 1)    # chain bits below 8 only (mu <= 8)
    assert sum(r[2] for r in rows(got.table)) == found and got.n_no_attractor == (1 << 64) - found


def keep_yaml(n=64, k=5):
    """Nodes 0..k-1 keep their state, every other node <- 0: 2^k fixed points, each with basin 2^(n-k); mu = 0 for the
    fixed point itself, 1 for everybody else."""
    names = ['x{}'.format(i) for i in range(n)]
    rules = {v: ZERO.format(v) for v in names}
    for i in range(k):
        rules[names[i]] = names[i]
    return yaml_of(names, rules)


@pytest.mark.parametrize('depth', [None, '1', '4'])
def test_identity_plus_constants_2p48_blocks_and_whole_space(eng, depth):
    if depth:
        os.environ['BSX_CUBE_DEPTH'] = depth
    k = 5
    setup(eng, keep_yaml(k=k))
    for first in (0, 0xABCD << 48):
        got = eng.attract(first, 1 << 48)
        m = 1 << (48 - k)
        ones = m - 1 if first == 0 else m
        assert rows(got.table) == [(key, 1, m, ones, ones) for key in range(1 << k)]
    whole = eng.attract2(0, 1 << 64)
    m = 1 << (64 - k)
    assert rows(whole.table) == [(key, 1, m, m - 1, m - 1) for key in range(1 << k)]
    ragged = eng.attract2((1 << 63) - 12345, (1 << 50) + 99999)            # tiles + cubes + tiles across the 2^63 boundary
    assert sum(r[2] for r in rows(ragged.table)) == (1 << 50) + 99999
    assert all(r[3] == r[2] and r[4] == r[2] for r in rows(ragged.table))  # nobody in this range is a fixed point


def ring_yaml(n=64, ring=6):
    """Nodes 0..ring-1 rotate (x_i <- x_(i+1 mod ring)), every other node <- 0: the attractors are the binary necklaces
    of length `ring` (cycle lengths divide it), every ring value's basin is 2^(n-ring), mu = 0 or 1."""
    names = ['x{}'.format(i) for i in range(n)]
    rules = {v: ZERO.format(v) for v in names}
    for i in range(ring):
        rules[names[i]] = names[(i + 1) % ring]
    return yaml_of(names, rules)


def necklaces(ring):
    out = {}
    for v in range(1 << ring):
        orbit = {((v >> r) | (v << (ring - r))) & ((1 << ring) - 1) for r in range(ring)}
        out.setdefault(min(orbit), len(orbit))
    return out          # key -> cycle length


@pytest.mark.parametrize('depth', [None, '1', '3'])
def test_rotation_ring_necklace_counts(eng, depth):
    if depth:
        os.environ['BSX_CUBE_DEPTH'] = depth
    ring = 6
    setup(eng, ring_yaml(ring=ring))
    neck = necklaces(ring)
    assert len(neck) == 14
    for first, a, wide in ((7 << 48, 48, False), (0, 48, False), (0, 63, True), (1 << 63, 63, True)):
        got = (eng.attract2 if wide else eng.attract)(first, 1 << a)
        per_value = 1 << (a - ring)
        expect = []
        for key, lam in sorted(neck.items()):
            count = lam * per_value
            ones = count - (lam if first == 0 else 0)                       # the cycle states themselves have mu = 0
            expect.append((key, lam, count, ones, ones))
        assert rows(got.table) == expect
        assert got.stats['state_steps'] == sum(e[3] + e[1] * e[2] for e in expect)


# ---- the north star ----------------------------------------------------------------------------------------------

def test_attract2_equals_attract_block_by_block_and_counts_syncs(eng):
    setup(eng, synth.north_star_yaml(), 4096)
    base = (0x0123456789ABCDEF >> 52) << 52
    eng.attract(base, 1 << 30, 4096)                                        # discovery, calibration
    parts = [eng.attract(base + (i << 48), 1 << 48, 4096) for i in range(16)]
    one = eng.attract2(base, 1 << 52, 4096)
    assert merge_tables([one.table]) == merge_tables([p.table for p in parts])
    assert one.n_no_attractor == sum(p.n_no_attractor for p in parts) == 0
    assert one.stats['state_steps'] == sum(p.stats['state_steps'] for p in parts)
    # one aligned block = one chain of launches, read back once (twice if a level's list overflowed and the chain was redone)
    assert one.stats['host_syncs'] <= 2 and one.stats['dominant_launches'] <= 2
    ragged = eng.attract2(base + 777, (1 << 52) - 777 + (3 << 40) + 5, 4096)
    tail = eng.attract2(base + (1 << 52), (3 << 40) + 5, 4096)
    head = eng.attract2(base, 777, 4096)
    assert merge_tables([ragged.table, head.table]) == merge_tables([one.table, tail.table])
    with pytest.raises(Exception):                                          # past the end of the 2^64 space
        eng.attract2((1 << 64) - 5, 6, 4096)


@pytest.mark.parametrize('knob', ['BSX_CUBE_LOWER', 'BSX_CUBE_LEAF', 'BSX_SPIN_WAIT'])
def test_the_general_build_and_the_plain_wait_give_the_same_tables(eng, knob):
    """The lower levels through the general cube build (BSX_CUBE_LOWER=0), the depth-1 level per child instead of per
    parent (BSX_CUBE_LEAF=0) and the counters by copy + wait instead of k_publish + spinning (BSX_SPIN_WAIT=0): same
    tables, same reference step counts."""
    cases = [(synth.north_star_yaml(), 4096, (0x0123456789ABCDEF >> 52) << 52, 1 << 52), (chain_yaml(), np.inf, 0, 1 << 63),
             (ring_yaml(), np.inf, 0, 1 << 56)]
    for text, max_t, first, count in cases:
        setup(eng, text, max_t)
        eng.attract2(first, min(count, 1 << 30), max_t)
        a = eng.attract2(first, count, max_t)
        os.environ[knob] = '0'
        b = eng.attract2(first, count, max_t)
        os.environ.pop(knob)
        assert merge_tables([a.table]) == merge_tables([b.table])
        assert a.n_no_attractor == b.n_no_attractor and a.stats['state_steps'] == b.stats['state_steps']


@pytest.mark.parametrize('seed,k,log2', [(305, 2, 48), (310, 2, 48), (307, 2, 48), (309, 1, 56), (314, 2, 56), (316, 2, 56), (319, 2, 56)])
def test_random_networks_at_scale_every_variant_of_the_cascade_agrees(eng, seed, k, log2):
    """Sizes the oracle cannot reach, networks without a closed form: the engine as it comes (sub-blocks from 2^52 problems,
    per-parent level, side streams) against forced sub-blocks, no sub-blocks, the per-child depth-1 level on one stream and
    a forced depth on two -- identical tables, no-attractor counts and reference step counts (tools/fuzz_scale.py runs more
    networks)."""
    net, space = setup(eng, synth.network_yaml(64, k, seed), 4096)
    first = ((0x9E3779B97F4A7C15 * (seed + 1)) % (1 << 64)) & ~((1 << log2) - 1)
    ref = None
    for env in ({}, {'BSX_CUBE_SPLIT': '1'}, {'BSX_CUBE_SPLIT': '0'}, {'BSX_CUBE_SPLIT': '1', 'BSX_CUBE_LEAF': '0', 'BSX_CUBE_STREAMS': '1'},
                {'BSX_CUBE_SPLIT': '1', 'BSX_CUBE_DEPTH': '3', 'BSX_CUBE_STREAMS': '2'}):
        for key in KNOBS:
            os.environ.pop(key, None)
        os.environ.update(env)
        eng.set_problem(net, space)                      # (fresh experience for every variant)
        r = eng.attract2(first, 1 << log2, 4096)
        got = (merge_tables([r.table]), r.n_no_attractor, r.stats['state_steps'])
        assert sum(e[1] for e in got[0].values()) + r.n_no_attractor == 1 << log2
        if ref is None:
            ref = got
        assert got == ref, env


@pytest.mark.parametrize('k,seed', [(1, 71), (2, 72), (3, 73), (4, 74), (4, 75)])
@pytest.mark.parametrize('depth', ['2', '3'])
def test_per_parent_leaf_level_on_rules_of_one_to_four_inputs(eng, k, seed, depth):
    """The depth-1 level evaluated per parent (LeafProgram: dependent rules bit-sliced over the children) on random
    networks whose rules have 1 .. 4 inputs, forced levels: against the per-child pass and the plain enumeration, with
    and without tight caps (a cap between mu = 0 and mu = 1 splits a class)."""
    text = synth.network_yaml(26, k, seed, fixed={5: '1'} if seed & 1 else None)
    for max_t, max_len in ((np.inf, np.inf), (7, 2), (2, np.inf)):
        setup(eng, text, max_t)
        total = 1 << 26
        os.environ['BSX_CUBE_DEPTH'] = depth
        a = eng.attract(0, total, max_t, max_len)
        os.environ['BSX_CUBE_LEAF'] = '0'
        b = eng.attract(0, total, max_t, max_len)
        os.environ.pop('BSX_CUBE_LEAF')
        os.environ['BSX_CUBES'] = '0'
        c = eng.attract(0, total, max_t, max_len)
        os.environ.pop('BSX_CUBES')
        os.environ.pop('BSX_CUBE_DEPTH')
        assert merge_tables([a.table]) == merge_tables([b.table]) == merge_tables([c.table]), text
        assert a.n_no_attractor == b.n_no_attractor == c.n_no_attractor
        assert a.stats['state_steps'] == b.stats['state_steps'] == c.stats['state_steps']


def canalizing_yaml(n, seed):
    """Rules of exactly four inputs that are strongly canalizing (and / or chains, and-or forms, with negations): ordered
    dynamics, and inside a block whose fixed digits feed them many free digits stop mattering -- cubes with levels."""
    rng = random.Random(seed)
    names = ['x{}'.format(i) for i in range(n)]
    rules = {}
    for i, v in enumerate(names):
        a, b, c, d = [('not ' if rng.random() < 0.3 else '') + names[p] for p in rng.sample(range(n), 4)]
        form = rng.choice(('{} and {} and {} and {}', '{} or {} or {} or {}', '({} and {}) or ({} and {})', '{} and ({} or {} or {})',
                           '({} or {}) and ({} or {})'))
        rules[v] = form.format(a, b, c, d)
    return yaml_of(names, rules)


@pytest.mark.parametrize('seed', range(6))
def test_per_parent_leaf_level_on_four_input_rules_that_collapse(eng, seed):
    text = canalizing_yaml(40, 400 + seed)
    for max_t, max_len in ((np.inf, np.inf), (3, np.inf)):
        setup(eng, text, max_t)
        for first in (0, 0x5A << 24):
            res = []
            for env in ({'BSX_CUBE_DEPTH': '3'}, {'BSX_CUBE_DEPTH': '3', 'BSX_CUBE_LEAF': '0'}, {'BSX_CUBES': '0'}):
                os.environ.update(env)
                res.append(eng.attract(first, 1 << 24, max_t, max_len))
                for k in env:
                    os.environ.pop(k)
            a, b, c = res
            assert merge_tables([a.table]) == merge_tables([b.table]) == merge_tables([c.table]), text
            assert a.n_no_attractor == b.n_no_attractor == c.n_no_attractor
            assert a.stats['state_steps'] == b.stats['state_steps'] == c.stats['state_steps']


def test_north_star_whole_space_in_one_call(eng):
    """All 2^64 initial states: the basins found by 65 536 calls of 2^48 problems in round 2 (profiles/r02_full_space.json),
    to the last digit, now in one call of two blocks."""
    setup(eng, synth.north_star_yaml(), 4096)
    eng.attract(0, 1 << 30, 4096)
    got = eng.attract2(0, 1 << 64, 4096)
    table = rows(got.table)
    assert sorted((r[1], r[2]) for r in table) == [(16, 6370653934217854976), (16, 12076090139491696640)]
    assert got.n_no_attractor == 0 and sum(r[2] for r in table) == 1 << 64
    means = sorted(r[3] / r[2] for r in table)
    assert abs(means[0] - 10.18) < 0.01 and abs(means[1] - 10.73) < 0.01    # (r02: mean transient 10.18 / 10.73)
    assert got.stats['host_syncs'] <= 4 and got.stats['kernel_ms'] < 2000   # (one wait per block; one more where a level's list overflowed)
    halves = [eng.attract2(h << 63, 1 << 63, 4096) for h in range(2)]
    assert merge_tables([got.table]) == merge_tables([h.table for h in halves])
    # forced shallower tops give the same table
    os.environ['BSX_CUBE_DEPTH'] = '3'
    q = eng.attract2(5 << 58, 1 << 56, 4096)
    os.environ.pop('BSX_CUBE_DEPTH')
    assert merge_tables([q.table]) == merge_tables([eng.attract2(5 << 58, 1 << 56, 4096).table])


def test_sampled_sub_ranges_of_a_bench_block_against_the_oracle(eng):
    """The block the bench times, sampled: 1024 random runs of 1024 consecutive problems (2^20 problems) through the
    per-problem path against the oracle's per-problem records (exact per problem); every sampled attractor must be
    in the block's table with the same length and at least the sampled count."""
    from oracle.cpu_oracle import Oracle
    net, space = setup(eng, synth.north_star_yaml(), 4096)
    batch = 1 << 56
    base = 0x0123456789ABCDEF & ~(batch - 1)
    block = merge_tables([eng.attract2(base, batch, 4096).table])
    rng = random.Random(56)
    orc = Oracle(net, space)
    seen = {}
    for _ in range(1024):
        first = base + (rng.randrange(batch >> 10) << 10)
        got = eng.attract(first, 1024, 4096, per_problem=True)
        pp, _, none, _ = orc.attract(first, 1024, 4096, n_threads=min(CORES, 8))
        for f in ('key', 'length', 'trajectory_l', 'found'):
            assert np.array_equal(got.per_problem[f], pp[f])
        for key, (length, count, _, _) in merge_tables([got.table]).items():
            e = seen.setdefault(key, [length, 0])
            e[1] += count
    for key, (length, count) in seen.items():
        assert key in block and block[key][0] == length and block[key][1] >= count


# ---- spaces beyond 64 'any' nodes, variations ---------------------------------------------------------------------

def test_ranges_that_cross_the_64_digit_boundary_of_a_128_node_space(eng):
    net, space = setup(eng, synth.network_yaml(128, 2, 129), 4096)
    first = (7 << 64) - (1 << 40) + 12345
    count = (1 << 41) + 999                                                  # the 65th digit changes inside the range
    eng.attract(first, 1 << 22, 4096)
    one = eng.attract2(first, count, 4096)
    cut = (7 << 64) - first
    parts = [eng.attract(first, cut, 4096), eng.attract(first + cut, count - cut, 4096)]
    assert merge_tables([one.table]) == merge_tables([p.table for p in parts])
    assert sum(r[2] for r in rows(one.table)) + one.n_no_attractor == count


def test_attract2_on_a_space_with_variations(eng):
    """Fixed-node and perturbation variations (not reachable through the YAML front end in attract mode, but allowed by
    the C-ABI): no cube path, the flat index carries into the variant number; one call = detector tiles inside."""
    from oracle.cpu_oracle import Oracle
    text = synth.network_yaml(14, 2, 141, initial={i: str(i & 1) for i in range(8, 14)}, fixed={5: 'any?', 2: '0?'},
                              perturbations={3: {'1?': '2'}, 7: {'0?': '1, 4'}})
    cfg = parse_input_text(text, 64, Mode.SIMULATE)             # parsed in a mode that allows variations
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    total = space.n_problems
    assert total == (1 << 8) * 3 * 2 * 2 * 2 * 2
    orc = Oracle(net, space)
    for first, count in ((0, total), (total // 3, total // 2), (total - 777, 777)):
        got = eng.attract2(first, count, 64)
        _, table, none, steps = orc.attract(first, count, 64, per_problem=False, n_threads=CORES)
        assert rows(got.table) == rows(table) and got.n_no_attractor == none and got.stats['state_steps'] == steps
    with pytest.raises(Exception):
        eng.attract2(total - 5, 6, 64)


def test_old_entry_point_reports_sums_that_do_not_fit(eng):
    """bsx_run_attract keeps its 64-bit record: a range whose sum of trajectory lengths overflows it is an error that
    names the wide entry point, not a wrapped number."""
    from boolsi_amd import _lib
    from boolsi_amd.engine import EngineError
    setup(eng, chain_yaml())
    ok = eng.attract(0, 1 << 58)                                             # sum_l ~ 2^58 x 15: fits
    assert rows(ok.table) == [(0, 1) + chain_sums(0, 58)]
    with pytest.raises(EngineError) as e:
        eng.attract(0, 1 << 63)                                              # sum_l ~ 2^63 x 15: does not
    assert e.value.status == _lib.ERR_RANGE_TOO_LARGE
    assert rows(eng.attract2(0, 1 << 63).table) == [(0, 1) + chain_sums(0, 63)]
