"""
Parity at production size: the configuration bench.py times -- big tiles of the class-pool
kernel in COUNTING mode, no per-problem records -- compared with the CPU oracle's aggregated table, and the
BASELINE configs at their full sizes through size-independent properties (conservation, partition
invariance).  Default engine only (the kernel variants are covered at slice size in test_gpu_parity.py).
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


@pytest.fixture(scope='module')
def eng():
    from boolsi_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


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


@pytest.mark.parametrize('cubes', ['0', '1'])
def test_north_star_counting_mode_tile_equals_oracle_table(eng, cubes, monkeypatch):
    """cubes = 0: 2^26 consecutive north-star problems in one call = one probe tile (member masks) + one big tile in
    counting mode, exactly what bench.py runs; table, no-attractor count and reference step count must be
    the oracle's (which walks every trajectory: ~20 s on the box's host cores).
    cubes = 1: the same range through the default path, the cube collapse."""
    from oracle.cpu_oracle import Oracle
    monkeypatch.setenv('BSX_CUBES', cubes)
    cfg = parse_input_text(synth.north_star_yaml(), 4096, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    first, count = 0x0123456789ABCDEF & ~((1 << 28) - 1), 1 << (26 if cubes == '0' else 25)      # bench.py's base
    got = eng.attract(first, count, 4096)
    assert got.stats['kernel_launches'] >= 2
    _, table, none, steps = Oracle(net, space).attract(first, count, 4096, per_problem=False, n_threads=CORES)
    assert rows(got.table) == rows(table)
    assert got.n_no_attractor == none
    assert got.stats['state_steps'] == steps
    # and a second call (space calibrated: a single counting-mode tile) on the neighbouring range, tight cap
    got2 = eng.attract(first + count, 1 << 24, 9)
    _, table2, none2, steps2 = Oracle(net, space).attract(first + count, 1 << 24, 9, per_problem=False, n_threads=CORES)
    assert rows(got2.table) == rows(table2) and got2.n_no_attractor == none2 and got2.stats['state_steps'] == steps2


def test_config3_full_sweep_conservation_and_partition(eng):
    """BASELINE config 3: n = 32, all 2^32 initial states, attract -t 4096.  Sum of basin sizes + no-attractor
    = 2^32, and the table of the whole sweep equals the merged tables of a 4-way range partition (the
    multi-GPU scheme, SURVEY 8d gate)."""
    cfg = parse_input_text(synth.config3_yaml(), 4096, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    n = 1 << 32
    assert space.n_problems == n
    whole = eng.attract(0, n, 4096)
    assert int(whole.table['count'].sum()) + whole.n_no_attractor == n
    parts = [eng.attract(r * (n // 4), n // 4, 4096) for r in range(4)]
    assert merged_rows([whole.table]) == merged_rows([p.table for p in parts])
    assert whole.n_no_attractor == sum(p.n_no_attractor for p in parts)
    assert whole.stats['state_steps'] == sum(p.stats['state_steps'] for p in parts)
    # a slice of it against the oracle (counting-mode tiles, unaligned start)
    from oracle.cpu_oracle import Oracle
    first, count = 0x9E3779B9 & ~63 | 5, 1 << 24
    got = eng.attract(first, count, 4096)
    _, table, none, steps = Oracle(net, space).attract(first, count, 4096, per_problem=False, n_threads=CORES)
    assert rows(got.table) == rows(table) and got.n_no_attractor == none and got.stats['state_steps'] == steps


def test_config5_full_size_with_digests(eng):
    """BASELINE config 5 at its full size: n = 128, K = 3, perturbation schedule, simulate -t 10000 over all 2^26
    problems, fold digests from the bit-sliced kernel (2^22 problems per call).  Oracle on two slices of 2^10,
    and partition invariance: a window that straddles the call boundaries gives the digests the whole run gave."""
    from oracle.cpu_oracle import Oracle
    cfg = parse_input_text(synth.config5_yaml(), 10000, Mode.SIMULATE)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    n = space.n_problems
    assert n == 1 << 26
    digests = np.zeros(n, np.uint64)
    steps = 0
    for first in range(0, n, 1 << 22):
        _, _, dig, st = eng.simulate(first, 1 << 22, 10000, trajectories=False, final=False, digest=True)
        digests[first:first + (1 << 22)] = dig
        steps += st['state_steps']
    assert steps == n * 10000
    orc = Oracle(net, space)
    for first in (0, (37 << 20) + 12345):
        _, _, odig, _ = orc.simulate(first, 1 << 10, 10000, want_traj=False, n_threads=CORES)
        assert np.array_equal(digests[first:first + (1 << 10)], odig)
    first = (3 << 22) - (1 << 19) + 777                      # straddles a call boundary, unaligned
    _, fin, dig, _ = eng.simulate(first, 1 << 20, 10000, trajectories=False, final=True, digest=True)
    assert np.array_equal(dig, digests[first:first + (1 << 20)])
    _, ofin, _, _ = orc.simulate(first, 1 << 9, 10000, want_traj=False, n_threads=CORES)
    assert np.array_equal(fin[:1 << 9], ofin)


def test_north_star_2p52_problems_three_decompositions_agree(eng):
    """The bench's regime (2^48 problems per call, a cascade of cube passes): 2^52 problems as 16 calls of 2^48, as 64
    calls of 2^46 and as ragged calls -- different blocks, different relevant-digit sets and level structures, the
    same exact table; every problem accounted for.  (tools/full_space.py does the same over all 2^64.)"""
    cfg = parse_input_text(synth.north_star_yaml(), 4096, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    base, total = 0xABC << 52, 1 << 52

    def sweep(pieces):
        tables, none, steps, at = [], 0, 0, base
        for n in pieces:
            r = eng.attract(at, n, 4096)
            assert int(r.table['count'].sum()) + r.n_no_attractor == n
            tables.append(r.table); none += r.n_no_attractor; steps += r.stats['state_steps']; at += n
        assert at == base + total
        return merged_rows(tables), none, steps

    a = sweep([1 << 48] * 16)
    b = sweep([1 << 46] * 64)
    ragged, sizes = [], [(1 << 48) - 12345, (1 << 47) + 999, 1 << 48, (1 << 46) + 77, (1 << 48) - 1]
    while sum(ragged) < total:
        ragged.append(min(sizes[len(ragged) % len(sizes)], total - sum(ragged)))
    assert all(0 < n <= 1 << 48 for n in ragged) and sum(ragged) == total
    c = sweep(ragged)
    assert a == b == c
    assert sum(r[2] for r in a[0]) + a[1] == total
    os.environ['BSX_CUBE_DEPTH'] = '1'              # ... and the first-update-only cube pass on a quarter of the first block
    try:
        d1 = eng.attract(base, 1 << 42, 4096)
    finally:
        os.environ.pop('BSX_CUBE_DEPTH')
    d8 = eng.attract(base, 1 << 42, 4096)
    assert rows(d1.table) == rows(d8.table) and d1.stats['state_steps'] == d8.stats['state_steps']
