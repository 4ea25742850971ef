"""
Cube collapse: aligned blocks of problems enumerated by the digits their first update depends
on, each class standing for 2^(a-r) problems.  An exact shortcut, so: identical tables, no-attractor counts
and reference step counts as the plain enumeration (BSX_CUBES=0) and as the CPU oracle, on aligned and ragged
ranges, tight caps, members that are cycle states themselves (mu = 0), and attractors met first by a cube pass.
"""
import os

import numpy as np
import pytest

from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import key_to_int
from boolsi_amd.input import parse_input_text

pytestmark = pytest.mark.gpu
CORES = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)


@pytest.fixture()
def eng():
    from boolsi_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    for k in ('BSX_CUBES', 'BSX_CUBE_DEPTH', 'BSX_CUBE_NEAR_CAP', 'BSX_CUBE_SPLIT', 'BSX_CUBE_STREAMS'):
        os.environ.pop(k, None)


def rows(table):
    return sorted((key_to_int(a['key']), int(a['length']), int(a['count']), int(a['sum_l']),
                   int(a['sum_l2_lo']) + (int(a['sum_l2_hi']) << 64)) for a in table)


def merged_rows(tables):
    acc = {}
    for t in tables:
        for k, length, c, s1, s2 in rows(t):
            e = acc.setdefault((k, length), [0, 0, 0])
            e[0] += c; e[1] += s1; e[2] += s2
    return sorted((k, length, *v) for (k, length), v in acc.items())


def setup(eng, text, max_t=4096):
    cfg = parse_input_text(text, max_t, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    return net, space


def same_as_oracle(eng, net, space, first, count, max_t=4096, max_len=None):
    from oracle.cpu_oracle import Oracle
    got = eng.attract(first, count, max_t, np.inf if max_len is None else max_len)
    _, table, none, steps = Oracle(net, space).attract(first, count, max_t, max_len, per_problem=False, n_threads=CORES)
    assert rows(got.table) == rows(table)
    assert got.n_no_attractor == none
    assert got.stats['state_steps'] == steps
    return got


def test_cubes_vs_oracle_aligned_ragged_and_capped(eng):
    net, space = setup(eng, synth.north_star_yaml())
    base = 0x0123456789ABCDEF & ~((1 << 28) - 1)
    same_as_oracle(eng, net, space, base, 1 << 24)                           # one cube (+ the discovery sample)
    g = same_as_oracle(eng, net, space, base + (1 << 24), 1 << 24)
    assert g.stats['executed_steps'] < (1 << 24) // 4                        # ... and it did collapse
    same_as_oracle(eng, net, space, base + 12345, (1 << 24) + 777)           # ragged head and tail around cubes
    same_as_oracle(eng, net, space, base - (1 << 20) - 3, (1 << 22) + 5)     # crosses a 2^28 boundary
    same_as_oracle(eng, net, space, base, 1 << 22, max_t=9, max_len=3)       # tight caps: classes run past the cap
    same_as_oracle(eng, net, space, base, 1 << 22, max_t=30)
    same_as_oracle(eng, net, space, 0, 1 << 23)                              # the block at digit value 0


@pytest.mark.parametrize('name,text,bits,log2n', [('config3', synth.config3_yaml(), 32, 22), ('n128_k2', synth.network_yaml(128, 2, 129), 128, 22),
                                                  ('n48_k3', synth.network_yaml(48, 3, 481), 48, 18)],      # (K = 3: long transients, slow oracle)
                         ids=['config3', 'n128_k2', 'n48_k3'])
def test_cubes_vs_oracle_other_networks(eng, name, text, bits, log2n):
    net, space = setup(eng, text)
    same_as_oracle(eng, net, space, 0, 1 << log2n)
    same_as_oracle(eng, net, space, (0x5DEECE66D << 7) % (1 << min(bits, 40)) | 1, (1 << (log2n - 1)) + 99)


def test_spaces_with_more_than_64_any_nodes_run_as_their_low_digits(eng):
    """n = 128 / 200, every node 'any': a call can only change the 64 lowest digits, so it runs as that plain
    space (higher digits of `first` folded into the origin) and gets the lean / pool / cube paths."""
    for n, seed, lg in ((128, 129, 22), (200, 2001, 20)):           # (the n = 200 oracle is the slow part: smaller ranges)
        net, space = setup(eng, synth.network_yaml(n, 2, seed))
        for first, count in (((0x9E3779B97F4A7C15 << 60) | (1 << 22), 1 << lg), ((1 << (n - 1)) + (5 << 64) + 12345, (1 << (lg - 1)) + 99)):
            same_as_oracle(eng, net, space, first, count)
            if n == 128:
                again = eng.attract(first, count, 4096)     # attractors of this region cached now: pool / cube passes only
                assert again.stats['executed_steps'] < count


def test_first_contact_with_an_attractor_in_a_cube_pass(eng):
    """Fresh engine, nothing cached beyond what the discovery sample finds: whatever attractor shows up first
    inside a cube pass must be learnt (detector from the listed class states) and the pass repeated."""
    from oracle.cpu_oracle import Oracle
    net, space = setup(eng, open(os.path.join(os.path.dirname(__file__), 'golden', 'cambium2.yaml')).read(), np.inf)
    got = eng.attract(0, 1 << 30)                    # 39 attractors, some with basins of a few hundred states
    assert len(got.table) == 39 and int(got.table['count'].sum()) == 1 << 30 and got.n_no_attractor == 0
    os.environ['BSX_CUBES'] = '0'
    plain = eng.attract(0, 1 << 30)
    assert rows(got.table) == rows(plain.table) and got.stats['state_steps'] == plain.stats['state_steps']


def test_cubes_with_a_warm_up_under_perturbations(eng):
    """Origin perturbations (cambium2: ETHL := 1 at t = 2..5; a synthetic schedule): the first update still
    depends on the relevant digits only and the search starts at s(T_p), which all members of a class share."""
    text = synth.network_yaml(40, 2, 401, perturbations={3: {'1': '1-3'}, 17: {'0': '2, 5'}}, fixed={9: '1'})
    for max_t in (4096, 12):
        net, space = setup(eng, text, max_t)
        same_as_oracle(eng, net, space, 0, 1 << 22, max_t=max_t)
        same_as_oracle(eng, net, space, (1 << 30) + 777, (1 << 21) + 5, max_t=max_t)


def test_members_that_are_cycle_states_themselves(eng):
    """Rules x0..x2 keep their state, x5 is constant 1, the rest constant 0: eight fixed points, each the ONLY
    mu = 0 member of its class, and (bit 5 set) not its class representative."""
    n = 20
    names = ['x{}'.format(i) for i in range(n)]
    rule = {i: names[i] for i in range(3)}
    rule[5] = '{0} or not {0}'.format(names[5])
    lines = ['nodes:'] + ['    - ' + v for v in names] + ['', 'update rules:']
    lines += ['    {}: {}'.format(names[i], rule.get(i, '{0} and not {0}'.format(names[i]))) for i in range(n)]
    lines += ['', 'initial state:'] + ['    {}: any'.format(v) for v in names]
    net, space = setup(eng, '\n'.join(lines) + '\n', np.inf)
    got = same_as_oracle(eng, net, space, 0, 1 << n, max_t=np.inf)
    assert sorted(int(a['count']) for a in got.table) == [1 << 17] * 8
    assert all(int(a['sum_l']) == (1 << 17) - 1 for a in got.table)          # one member with l = 0, the others l = 1
    same_as_oracle(eng, net, space, 1 << 16, 3 << 16, max_t=np.inf)


def test_large_ranges_cubes_equal_plain_enumeration_and_partition(eng):
    net, space = setup(eng, synth.north_star_yaml())
    base = (0x0123456789ABCDEF >> 40) << 40
    whole = eng.attract(base + 5, (1 << 34) - 9, 4096)                       # 2^34 problems in one call
    os.environ['BSX_CUBES'] = '0'
    parts = []
    at, end = base + 5, base + 5 + (1 << 34) - 9
    while at < end:
        n = min(1 << 32, end - at)
        parts.append(eng.attract(at, n, 4096))
        at += n
    os.environ.pop('BSX_CUBES')
    assert merged_rows([whole.table]) == merged_rows([p.table for p in parts])
    assert whole.n_no_attractor == sum(p.n_no_attractor for p in parts)
    assert whole.stats['state_steps'] == sum(p.stats['state_steps'] for p in parts)
    # 2^44 problems: conservation and partition invariance (what the multi-GPU split relies on)
    n = 1 << 44
    big = eng.attract(base, n, 4096)
    assert int(big.table['count'].sum()) + big.n_no_attractor == n
    quarters = [eng.attract(base + q * (n // 4), n // 4, 4096) for q in range(4)]
    assert merged_rows([big.table]) == merged_rows([q.table for q in quarters])
    assert big.stats['state_steps'] == sum(q.stats['state_steps'] for q in quarters)


@pytest.mark.parametrize('seed', range(96))
def test_cubes_equal_plain_enumeration_on_random_spaces(eng, seed):
    """Differential fuzz: random small networks (fixed nodes, warm-ups under perturbations, tight caps, ragged
    ranges) through the cube passes and through the plain enumeration -- tables, no-attractor counts and
    reference step counts must be identical.  (The plain path is pinned against the oracle elsewhere.)"""
    import random
    rng = random.Random(1000 + seed)
    n = rng.choice((18, 20, 22))
    k = rng.choice((1, 2, 2, 3))
    fixed = {rng.randrange(n): rng.choice('01')} if rng.random() < 0.4 else None
    pert = None
    if rng.random() < 0.4:
        pert = {rng.randrange(n): {rng.choice('01'): ', '.join(str(t) for t in sorted(rng.sample(range(1, 7), rng.randrange(1, 4))))}}
    initial = None
    if rng.random() < 0.3:          # some nodes not 'any': the digits are deposited run by run
        initial = {i: rng.choice('01') for i in rng.sample(range(n), 2)}
    max_t = rng.choice((np.inf, 4096, 4096, 9, 7))
    max_len = rng.choice((np.inf, np.inf, 1, 2))
    text = synth.network_yaml(n, k, 5000 + seed, initial=initial, fixed=fixed, perturbations=pert)
    net, space = setup(eng, text, max_t)
    total = space.n_problems
    count = rng.randrange(min(1 << 17, total // 2), total + 1)
    first = rng.randrange(0, total - count + 1)
    if rng.random() < 0.5:
        first &= ~0xFFFF
    os.environ.pop('BSX_CUBES', None)
    a = eng.attract(first, count, max_t, max_len)
    os.environ['BSX_CUBES'] = '0'
    try:
        b = eng.attract(first, count, max_t, max_len)
    finally:
        os.environ.pop('BSX_CUBES')
    assert rows(a.table) == rows(b.table), text
    assert a.n_no_attractor == b.n_no_attractor and a.stats['state_steps'] == b.stats['state_steps']
    assert int(a.table['count'].sum()) + a.n_no_attractor == count


# ---- deeper collapse: classes by the digits F^d still depends on, near-cycle classes handed down level by level

def shift_register_yaml(n, tail='0'):
    """x_i <- x_(i+1), the last node constant: every state falls to one fixed point, mu = 1 + its highest set bit.
    Depth d leaves n - d relevant digits and exactly one class per level sits next to the cycle."""
    names = ['x{}'.format(i) for i in range(n)]
    const = '{0} and not {0}' if tail == '0' else '{0} or not {0}'
    lines = ['nodes:'] + ['    - ' + v for v in names] + ['', 'update rules:']
    lines += ['    {}: {}'.format(names[i], names[i + 1]) for i in range(n - 1)]
    lines += ['    {}: {}'.format(names[-1], const.format(names[-1]))]
    lines += ['', 'initial state:'] + ['    {}: any'.format(v) for v in names]
    return '\n'.join(lines) + '\n'


@pytest.mark.parametrize('depth', [1, 2, 3, 5, 8, 16])
def test_deep_levels_on_a_shift_register(eng, depth):
    os.environ['BSX_CUBE_DEPTH'] = str(depth)
    n = 20
    net, space = setup(eng, shift_register_yaml(n), np.inf)
    got = same_as_oracle(eng, net, space, 0, 1 << n, max_t=np.inf)
    assert len(got.table) == 1 and int(got.table[0]['count']) == 1 << n
    # mu = 0 once, mu = b + 1 for the 2^b states with highest set bit b
    assert int(got.table[0]['sum_l']) == sum((b + 1) << b for b in range(n))
    same_as_oracle(eng, net, space, 3 << 16, 5 << 16, max_t=np.inf)          # blocks that do not hold the fixed point
    same_as_oracle(eng, net, space, 0, 1 << n, max_t=7)                      # cap inside the levels
    same_as_oracle(eng, net, space, 0, 1 << n, max_t=np.inf, max_len=0)
    net, space = setup(eng, shift_register_yaml(n, tail='1'), np.inf)        # the fixed point is all ones: not a representative
    same_as_oracle(eng, net, space, 0, 1 << n, max_t=np.inf)


@pytest.mark.parametrize('depth', [2, 4, 8])
def test_deep_levels_vs_oracle_on_the_north_star(eng, depth):
    os.environ['BSX_CUBE_DEPTH'] = str(depth)
    net, space = setup(eng, synth.north_star_yaml())
    base = 0x0123456789ABCDEF & ~((1 << 28) - 1)
    same_as_oracle(eng, net, space, base, 1 << 24)                           # (with the discovery sample)
    g = same_as_oracle(eng, net, space, base + (1 << 24), 1 << 24)
    if depth >= 3:
        assert g.stats['executed_steps'] < (1 << 24) // 64                   # 2^18 classes or fewer, a few updates each
    same_as_oracle(eng, net, space, base + (1 << 25) + 999, (1 << 23) + 12345)
    same_as_oracle(eng, net, space, 0, 1 << 23, max_t=12)


def test_a_full_near_cycle_list_restarts_shallower(eng):
    os.environ['BSX_CUBE_NEAR_CAP'] = '3'
    n = 20
    names = ['x{}'.format(i) for i in range(n)]
    # two layers feeding eight self-loops: everything is at a fixed point after two updates, every deep class is listed
    lines = ['nodes:'] + ['    - ' + v for v in names] + ['', 'update rules:']
    for i in range(n):
        if i < 8: rule = '{} or {}'.format(names[i], names[8 + i % 6])
        elif i < 14: rule = names[14 + i % 6]
        else: rule = '{0} and not {0}'.format(names[i])
        lines.append('    {}: {}'.format(names[i], rule))
    lines += ['', 'initial state:'] + ['    {}: any'.format(v) for v in names]
    net, space = setup(eng, '\n'.join(lines) + '\n', np.inf)
    same_as_oracle(eng, net, space, 0, 1 << n, max_t=np.inf)
    os.environ.pop('BSX_CUBE_NEAR_CAP')
    net, space = setup(eng, '\n'.join(lines) + '\n', np.inf)                  # (fresh engine state: no depth cap remembered)
    same_as_oracle(eng, net, space, 0, 1 << n, max_t=np.inf)


@pytest.mark.parametrize('seed', range(64))
def test_deep_levels_equal_plain_enumeration_on_random_spaces(eng, seed):
    """Differential fuzz of the level passes: sparse networks (K = 1 and 2, where influence dies out over a few
    updates), fixed nodes, tight caps, forced depths; against the plain enumeration."""
    import random
    rng = random.Random(7000 + seed)
    n = rng.choice((18, 20, 22))
    k = rng.choice((1, 1, 2, 2))
    fixed = {rng.randrange(n): rng.choice('01')} if rng.random() < 0.3 else None
    initial = {i: rng.choice('01') for i in rng.sample(range(n), 2)} if rng.random() < 0.3 else None
    max_t = rng.choice((np.inf, 4096, 4096, 9, 5, 3))
    max_len = rng.choice((np.inf, np.inf, 1, 2))
    text = synth.network_yaml(n, k, 9000 + seed, initial=initial, fixed=fixed)
    net, space = setup(eng, text, max_t)
    total = space.n_problems
    count = rng.randrange(min(1 << 17, total // 2), total + 1)
    first = rng.randrange(0, total - count + 1)
    if rng.random() < 0.5:
        first &= ~0xFFFF
    os.environ['BSX_CUBE_DEPTH'] = str(rng.choice((2, 3, 4, 8, 8, 16)))
    if rng.random() < 0.25:
        os.environ['BSX_CUBE_NEAR_CAP'] = str(rng.choice((1, 7, 100)))
    a = eng.attract(first, count, max_t, max_len)
    os.environ['BSX_CUBES'] = '0'
    b = eng.attract(first, count, max_t, max_len)
    assert rows(a.table) == rows(b.table), text
    assert a.n_no_attractor == b.n_no_attractor and a.stats['state_steps'] == b.stats['state_steps']
    assert int(a.table['count'].sum()) + a.n_no_attractor == count


@pytest.mark.parametrize('depth', ['2', '8'])
@pytest.mark.parametrize('name,text,bits,log2n', [('config3', synth.config3_yaml(), 32, 22), ('n128_k2', synth.network_yaml(128, 2, 129), 128, 22),
                                                  ('n200_k2', synth.network_yaml(200, 2, 2001), 200, 20),
                                                  ('n48_k3', synth.network_yaml(48, 3, 481), 48, 18)],
                         ids=['config3', 'n128_k2', 'n200_k2', 'n48_k3'])
def test_forced_levels_vs_oracle_other_word_counts(eng, name, text, bits, log2n, depth):
    """The level passes on states of 1, 4 and 8 words (the engine's own choice keeps blocks this small at one level)."""
    if depth == '2' and name != 'config3':
        pytest.skip('slow oracle: the deepest setting covers these networks (depth 2: config3 and the fuzz)')
    os.environ['BSX_CUBE_DEPTH'] = depth
    net, space = setup(eng, text)
    same_as_oracle(eng, net, space, 0, 1 << log2n)
    same_as_oracle(eng, net, space, (0x5DEECE66D << 7) % (1 << min(bits, 40)) | 1, (1 << (log2n - 1)) + 99)


@pytest.mark.parametrize('depth', ['2', '8'])
def test_forced_levels_with_a_warm_up(eng, depth):
    """Warm-up under the origin's perturbation schedule: one pass at the best depth <= T_p, nothing listed."""
    os.environ['BSX_CUBE_DEPTH'] = depth
    text = synth.network_yaml(40, 2, 401, perturbations={3: {'1': '1-3'}, 17: {'0': '2, 5'}}, fixed={9: '1'})
    for max_t in (4096, 12):
        net, space = setup(eng, text, max_t)
        same_as_oracle(eng, net, space, 0, 1 << 22, max_t=max_t)
        same_as_oracle(eng, net, space, (1 << 30) + 777, (1 << 21) + 5, max_t=max_t)
    net, space = setup(eng, open(os.path.join(os.path.dirname(__file__), 'golden', 'cambium2.yaml')).read(), np.inf)
    got = eng.attract(0, 1 << 30)
    os.environ['BSX_CUBES'] = '0'
    plain = eng.attract(0, 1 << 30)
    assert len(got.table) == 39 and rows(got.table) == rows(plain.table) and got.stats['state_steps'] == plain.stats['state_steps']


# ---- sub-blocks: a block split along well-chosen relevant digits (BSX_CUBE_SPLIT=1 forces an eight-leaf tree on any block) ----

@pytest.mark.parametrize('depth,streams', [(None, None), ('2', None), ('4', None), (None, '1'), ('4', '2')])
def test_sub_blocks_vs_oracle_on_the_north_star(eng, depth, streams):
    os.environ['BSX_CUBE_SPLIT'] = '1'
    if streams:
        os.environ['BSX_CUBE_STREAMS'] = streams         # (default: the lower levels of the chains on four side streams)
    if depth:
        os.environ['BSX_CUBE_DEPTH'] = depth
    net, space = setup(eng, synth.north_star_yaml())
    base = 0x0123456789ABCDEF & ~((1 << 28) - 1)
    g = same_as_oracle(eng, net, space, base, 1 << 24)                       # (first contact with the attractors happens in sub-blocks)
    assert g.stats['kernel_launches'] >= 8                                          # ... and there were several chains
    if streams or depth:
        return                                                                   # (more blocks, ragged ranges and caps: the engine's own depth, four streams)
    g = same_as_oracle(eng, net, space, base + (1 << 24), 1 << 24)
    assert g.stats['kernel_launches'] >= 8
    same_as_oracle(eng, net, space, base + (1 << 25) + 999, (1 << 23) + 12345)
    same_as_oracle(eng, net, space, 0, 1 << 23, max_t=12)


@pytest.mark.parametrize('name,text,bits,log2n', [('config3', synth.config3_yaml(), 32, 22), ('n128_k2', synth.network_yaml(128, 2, 129), 128, 22),
                                                  ('n200_k2', synth.network_yaml(200, 2, 2001), 200, 20)],
                         ids=['config3', 'n128_k2', 'n200_k2'])
def test_sub_blocks_vs_oracle_other_word_counts(eng, name, text, bits, log2n):
    os.environ['BSX_CUBE_SPLIT'] = '1'
    os.environ['BSX_CUBE_DEPTH'] = '8'
    net, space = setup(eng, text)
    same_as_oracle(eng, net, space, (0x5DEECE66D << 7) % (1 << min(bits, 40)) | 1, (1 << log2n) + 99)      # (ragged on both sides)


@pytest.mark.parametrize('seed', range(32))
def test_sub_blocks_equal_plain_enumeration_on_random_spaces(eng, seed):
    """Differential fuzz of split blocks: sparse networks, fixed nodes, tight caps, forced or free depths, short
    near-cycle lists (a sub-block restarts shallower on its own); against the plain enumeration."""
    import random
    rng = random.Random(8100 + seed)
    n = rng.choice((18, 20, 22))
    k = rng.choice((1, 2, 2, 3))
    fixed = {rng.randrange(n): rng.choice('01')} if rng.random() < 0.3 else None
    initial = {i: rng.choice('01') for i in rng.sample(range(n), 2)} if rng.random() < 0.3 else None
    max_t = rng.choice((np.inf, 4096, 4096, 9, 5, 3, 1))
    max_len = rng.choice((np.inf, np.inf, 1, 2))
    text = synth.network_yaml(n, k, 9500 + seed, initial=initial, fixed=fixed)
    net, space = setup(eng, text, max_t)
    total = space.n_problems
    count = rng.randrange(min(1 << 17, total // 2), total + 1)
    first = rng.randrange(0, total - count + 1)
    if rng.random() < 0.5:
        first &= ~0xFFFF
    os.environ['BSX_CUBE_SPLIT'] = '1'
    depth = rng.choice((None, None, 2, 3, 4, 8))
    if depth:
        os.environ['BSX_CUBE_DEPTH'] = str(depth)
    if rng.random() < 0.25:
        os.environ['BSX_CUBE_NEAR_CAP'] = str(rng.choice((1, 7, 100)))
    a = eng.attract(first, count, max_t, max_len)
    os.environ['BSX_CUBES'] = '0'
    b = eng.attract(first, count, max_t, max_len)
    assert rows(a.table) == rows(b.table), text
    assert a.n_no_attractor == b.n_no_attractor and a.stats['state_steps'] == b.stats['state_steps']
    assert int(a.table['count'].sum()) + a.n_no_attractor == count


def test_sub_blocks_members_that_are_cycle_states_themselves(eng):
    """The whole space of small networks, where every cycle state is a member of some sub-block's class."""
    os.environ['BSX_CUBE_SPLIT'] = '1'
    for seed in (3, 11, 29):
        net, space = setup(eng, synth.network_yaml(20, 2, 600 + seed), np.inf)
        same_as_oracle(eng, net, space, 0, 1 << 20, max_t=np.inf)
