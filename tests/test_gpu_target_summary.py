"""
Target mode's on-device summary sink (bsx_run_target_summary): hit count, histogram of first-hit times and the
first n hits, against the full hit list of bsx_run_target and the CPU oracle; BASELINE config 4 at its full
size (2^31 problems, 8 knock-out variants) through conservation, partition invariance and oracle-checked slices.
"""
import os
import random

import numpy as np
import pytest

from boolsi_amd import synth
from boolsi_amd.compile import code_to_words, compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.input import parse_input_text
from util import load, t_of, compile_case

pytestmark = pytest.mark.gpu
CORES = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)


@pytest.fixture(scope='module')
def eng():
    from boolsi_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def target_words(cfg, net):
    return (code_to_words(sum(1 << n for n in cfg['target node set']), net.n_words),
            code_to_words(cfg['target substate code'], net.n_words))


@pytest.mark.parametrize('cubes', ['0', '1'])
@pytest.mark.parametrize('case', load('target.json'), ids=lambda c: c['name'])
def test_summary_equals_the_hit_list(eng, case, cubes, monkeypatch):
    """cubes = 1 (default): aligned blocks of >= 2^16 problems are counted through cube passes (classes of problems
    that share s(1), s(2), ...); the step statistic then counts a class's steps once per member."""
    monkeypatch.setenv('BSX_CUBES', cubes)
    cfg, net, space = compile_case(case)
    eng.set_problem(net, space)
    mask, code = target_words(cfg, net)
    n = min(space.n_problems, 1 << 20)          # (config 4 has 2^31 problems: its full size is the next test)
    max_t = t_of(case['max_t'])
    hits, st = eng.target(0, n, max_t, mask, code)
    bins = 16
    for cap in (0, 3, len(hits), len(hits) + 5):
        total, hist, first, st2 = eng.target_summary(0, n, max_t, mask, code, hist_bins=bins, cap=cap)
        assert total == len(hits) and int(hist.sum()) == total
        expect = np.bincount(np.minimum(hits['t'], bins - 1).astype(np.int64), minlength=bins)
        assert hist.tolist() == expect.tolist()
        assert first.tobytes() == hits[:cap].tobytes()
        if cubes == '0':
            assert st2['state_steps'] == st['state_steps']
    # a sub-range that starts inside the space (offsets are relative to `first`)
    lo = n // 3
    part, _ = eng.target(lo, n - lo, max_t, mask, code)
    total, _, first, _ = eng.target_summary(lo, n - lo, max_t, mask, code, cap=4)
    assert total == len(part) and first.tobytes() == part[:4].tobytes()


def test_config4_full_size(eng):
    from oracle.cpu_oracle import Oracle
    cfg = parse_input_text(synth.config4_yaml(), 1024, Mode.TARGET)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    mask, code = target_words(cfg, net)
    n = space.n_problems
    assert n == 1 << 31
    total, hist, first, st = eng.target_summary(0, n, 1024, mask, code, hist_bins=1026, cap=1000)
    assert int(hist.sum()) == total and hist[1025] == 0 and len(first) == min(total, 1000)
    # partition invariance: the 8 knock-out variants (2^28 problems each) separately
    parts = [eng.target_summary(v << 28, 1 << 28, 1024, mask, code, hist_bins=1026) for v in range(8)]
    assert sum(p[0] for p in parts) == total
    assert sum(p[1] for p in parts).tolist() == hist.tolist()
    assert sum(p[3]['state_steps'] for p in parts) == st['state_steps']
    # the listed hits are the first ones in index order
    lst, _ = eng.target(0, 1 << 12, 1024, mask, code)
    assert first[:len(lst)].tobytes() == lst[:1000].tobytes()
    # oracle on slices of every variant (and one across a variant boundary); the last two are aligned blocks
    # of 2^20 problems, i.e. cube passes
    orc = Oracle(net, space)
    rng = random.Random(4)
    starts = [((v << 28) + rng.randrange((1 << 28) - (1 << 16)), 1 << 16) for v in range(8)] + [((3 << 28) - 30000, 1 << 16)]
    starts += [((5 << 28) + (37 << 20), 1 << 20), ((2 << 28) + (200 << 20) + (1 << 19), 1 << 20)]
    for s, cnt in starts:
        got, h, _, _ = eng.target_summary(s, cnt, 1024, mask, code, hist_bins=1026)
        pp, _ = orc.target(s, cnt, 1024, mask, code, n_threads=CORES)
        reached = pp['reached'] != 0
        assert got == int(reached.sum())
        assert h.tolist() == np.bincount(pp['t_stop'][reached].astype(np.int64), minlength=1026).tolist()


@pytest.mark.parametrize('seed', range(48))
def test_target_cube_passes_equal_plain_counting_on_random_spaces(eng, seed, monkeypatch):
    """Differential fuzz of the summary sink: random networks, targets, caps and fixed-node variations; cube
    passes (default) against plain counting (BSX_CUBES=0): hit counts and histograms must be identical."""
    rng = random.Random(7000 + seed)
    n = rng.choice((18, 20, 22))
    k = rng.choice((1, 2, 2, 3))
    n_const = rng.choice((0, 0, 2))
    initial = {i: rng.choice('01') for i in rng.sample(range(n), n_const)} or None
    fixed = None
    if rng.random() < 0.6:
        fixed = {i: rng.choice(('0', '1', '0?', '1?', 'any', 'any?')) for i in rng.sample(range(n), rng.randrange(1, 3))}
    tnodes = rng.sample(range(n), rng.randrange(1, 6))
    target = {i: (rng.choice('01') if i in tnodes else 'any') for i in range(n)}
    max_t = rng.choice((np.inf, 1024, 6, 2))
    text = synth.network_yaml(n, k, 9000 + seed, initial=initial, fixed=fixed, target=target)
    cfg = parse_input_text(text, max_t, Mode.TARGET)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    mask, code = target_words(cfg, net)
    total = space.n_problems
    count = rng.randrange(min(1 << 17, total // 2), total + 1)
    first = rng.randrange(0, total - count + 1)
    if rng.random() < 0.5:
        first &= ~0xFFFF
    cap = rng.choice((0, 0, 5))
    monkeypatch.setenv('BSX_CUBES', '1')
    a = eng.target_summary(first, count, max_t, mask, code, hist_bins=64, cap=cap)
    monkeypatch.setenv('BSX_CUBES', '0')
    b = eng.target_summary(first, count, max_t, mask, code, hist_bins=64, cap=cap)
    assert a[0] == b[0] and a[1].tolist() == b[1].tolist() and a[2].tobytes() == b[2].tobytes(), text
