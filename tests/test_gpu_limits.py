"""Argument checks of the run entry points on the GPU box (ranges past the end of the space)."""
import numpy as np
import pytest

from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.input import parse_input_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from boolsi_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_range_past_the_end_of_the_space_is_rejected(eng):
    from boolsi_amd.engine import EngineError
    from util import load
    cfg = parse_input_text(synth.config5_yaml(max_t=40, n_any=12), 40, Mode.SIMULATE)     # 2^12 problems, n = 128
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    n = space.n_problems
    eng.simulate(n - 10, 10, 40, trajectories=False)                                       # up to the last problem: fine
    for first, count in ((n - 10, 11), (0, n + 1), (n, 1)):
        with pytest.raises(EngineError) as e:
            eng.simulate(first, count, 40, trajectories=False)
        assert e.value.status == -1
    with pytest.raises(EngineError):
        eng.attract(n - 4096, 8192, 4096)                       # the lean path's enumeration must not spill either
    with pytest.raises(EngineError):
        eng.target(n - 1, 2, 40, np.zeros(2, np.uint64), np.zeros(2, np.uint64))
    with pytest.raises(EngineError):
        eng.simulate(0, 8, 3, trajectories=False)               # max_t below the last perturbation time
    # spaces with variations: the variant digits bound the range
    case = next(c for c in load('simulate.json') if c['name'] == 'toyB_variations')
    from util import compile_case
    _, net2, space2 = compile_case(case)
    eng.set_problem(net2, space2)
    n2 = space2.n_problems
    assert n2 == len(case['indices'])
    _, fin, _, _ = eng.simulate(0, n2, case['max_t'], trajectories=False)
    assert len(fin) == n2
    with pytest.raises(EngineError):
        eng.simulate(0, n2 + 1, case['max_t'], trajectories=False)
    with pytest.raises(EngineError):
        eng.simulate(n2, 1, case['max_t'], trajectories=False)


def _identity_yaml(n):
    names = ['x{}'.format(i) for i in range(n)]
    lines = ['nodes:'] + ['    - ' + v for v in names] + ['', 'update rules:']
    lines += ['    {0}: {0}'.format(v) for v in names]
    lines += ['', 'initial state:'] + ['    {}: any'.format(v) for v in names]
    return '\n'.join(lines) + '\n'


def test_sixteen_million_attractors_fit_the_device_table(eng):
    """Identity network, n = 24: every state is its own fixed point -> 2^24 distinct attractors.  They overflow
    the per-wave tables and the log into the HBM attractor table; the result must be exact."""
    from boolsi_amd.engine import EngineError
    n = 24
    cfg = parse_input_text(_identity_yaml(n), np.inf, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    total = 1 << n
    r = eng.attract(0, total, np.inf, cap=total + 16)
    assert len(r.table) == total and r.n_no_attractor == 0
    assert np.array_equal(np.sort(r.table['key'][:, 0]), np.arange(total, dtype=np.uint64))
    assert (r.table['count'] == 1).all() and (r.table['length'] == 1).all()
    assert not r.table['sum_l'].any() and not r.table['sum_l2_lo'].any()
    assert r.stats['state_steps'] == total                                     # mu = 0, lambda = 1 each
    # a second, smaller run on the same handle starts from an empty table
    r2 = eng.attract(12345, 1 << 18, np.inf, cap=1 << 19)
    assert np.array_equal(np.sort(r2.table['key'][:, 0]), np.arange(12345, 12345 + (1 << 18), dtype=np.uint64))
    # the caller's capacity is the limit, loudly
    with pytest.raises(EngineError) as e:
        eng.attract(0, 1 << 20, np.inf, cap=1 << 16)
    assert e.value.status == -5
    r3 = eng.attract(0, 1 << 16, np.inf, cap=1 << 17)                          # and the handle is still usable
    assert len(r3.table) == 1 << 16
