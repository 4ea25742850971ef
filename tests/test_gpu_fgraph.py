"""
Functional-graph mode (bsx_run_attract_fgraph): successor array + pointer doubling + pointer
jumping over 2^n-sized arrays must give exactly the tables of the trajectory path and of the CPU oracle.
"""
import os

import numpy as np
import pytest

from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import key_to_int
from boolsi_amd.input import parse_input_text
from util import GOLDEN

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


def setup(eng, text, max_t):
    cfg = parse_input_text(text, max_t, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    return net, space


@pytest.mark.parametrize('n,k,seed', [(10, 2, 1), (16, 2, 7), (18, 3, 3), (20, 2, 20), (12, 5, 5)])
def test_fgraph_equals_oracle_on_small_spaces(eng, n, k, seed):
    from oracle.cpu_oracle import Oracle
    for max_t, max_len in ((np.inf, None), (4096, None), (7, None), (40, 2), (1, None)):
        net, space = setup(eng, synth.network_yaml(n, k, seed), max_t)
        orc = Oracle(net, space)
        total = 1 << n
        for first, count in ((0, total), (total // 3, total // 2 + 7), (total - 1, 1)):
            got = eng.attract_fgraph(first, count, max_t, np.inf if max_len is None else max_len)
            _, table, none, steps = orc.attract(first, count, None if max_t == np.inf else max_t, max_len,
                                                per_problem=False, n_threads=CORES)
            assert rows(got.table) == rows(table), (max_t, max_len, first, count)
            assert got.n_no_attractor == none
            assert got.stats['state_steps'] == steps


def test_fgraph_with_origin_perturbations(eng):
    """A warm-up under perturbations only moves the start of the search (warm map)."""
    from oracle.cpu_oracle import Oracle
    text = synth.network_yaml(14, 2, 141, perturbations={3: {'1': '2-4'}, 9: {'0': '1, 6'}}, fixed={5: '1'})
    for max_t in (np.inf, 50, 8):
        net, space = setup(eng, text, max_t)
        got = eng.attract_fgraph(0, 1 << 14, max_t)
        _, table, none, steps = Oracle(net, space).attract(0, 1 << 14, None if max_t == np.inf else max_t, None,
                                                            per_problem=False, n_threads=CORES)
        assert rows(got.table) == rows(table) and got.n_no_attractor == none and got.stats['state_steps'] == steps


def test_fgraph_needs_a_full_any_space(eng):
    from boolsi_amd.engine import EngineError
    setup(eng, synth.north_star_yaml(), 4096)                   # 64 nodes
    with pytest.raises(EngineError) as e:
        eng.attract_fgraph(0, 1 << 20, 4096)
    assert e.value.status == -4


def test_fgraph_cambium2_full_sweep(eng):
    """2^30 states: the published 39 attractors and basin sizes, and the trajectory path's exact sums."""
    net, space = setup(eng, open(os.path.join(GOLDEN, 'cambium2.yaml')).read(), np.inf)
    fg = eng.attract_fgraph(0, 1 << 30)
    traj = eng.attract(0, 1 << 30)
    assert len(fg.table) == 39 and fg.n_no_attractor == 0
    assert rows(fg.table) == rows(traj.table)
    assert fg.stats['state_steps'] == traj.stats['state_steps']


def test_fgraph_config3_full_sweep(eng):
    """BASELINE config 3 (n = 32, all 2^32 states, -t 4096): 48 GB of arrays in HBM; table equal to the trajectory path."""
    net, space = setup(eng, synth.config3_yaml(), 4096)
    fg = eng.attract_fgraph(0, 1 << 32, 4096)
    traj = eng.attract(0, 1 << 32, 4096)
    assert rows(fg.table) == rows(traj.table)
    assert fg.n_no_attractor == traj.n_no_attractor
    assert fg.stats['state_steps'] == traj.stats['state_steps']
    print('fgraph config3: {:.1f} ms in {} launches vs trajectory path {:.1f} ms'.format(
        fg.stats['kernel_ms'], fg.stats['kernel_launches'], traj.stats['kernel_ms']))
