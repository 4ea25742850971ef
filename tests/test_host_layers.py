"""
Host layers around the engine (no GPU): problem enumeration, CSV writers, node correlations, CLI
surface.  Golden data: the reference's own example outputs (tests/golden/examples/, copied data
files) and vectors generated from the reference (tests/golden/*.json).
"""
import os

import numpy as np
import pytest
from click.testing import CliRunner

from boolsi_amd import batching
from boolsi_amd.attract import AggregatedAttractor
from boolsi_amd.attractor_analysis import compute_frequency_spearmanrho, find_node_correlations, weighted_ranks
from boolsi_amd.cli import cli
from boolsi_amd.model import decode_state, encode_state
from boolsi_amd.output import format_state, list_texts, output_attractors, output_node_correlations, \
    output_simulations, write_states
from boolsi_amd.simulate import Simulation
from util import GOLDEN, load, compile_case

EXAMPLES = os.path.join(GOLDEN, 'examples')


def read(path):
    with open(path, 'rb') as f:
        return f.read()


def test_enumeration_matches_reference():
    g = load('enumeration.json')
    g['mode'] = 'simulate'
    cfg, _, _ = compile_case(g)
    origin, variations = cfg['origin simulation problem'], cfg['simulation problem variations']
    system = batching.create_numeral_system_from_variations(variations)
    for i, ref in zip(g['indices'], g['problems']):
        init, fixed, pert = batching.problem_from_index(int(i), origin, variations, system)
        assert str(encode_state(set(), init)[0]) == ref['initial_code']
        assert sorted([n, int(v)] for n, v in fixed.items()) == ref['fixed']
        assert sorted([t, n, int(v)] for t, d in pert.items() for n, v in d.items()) == ref['pert']


def test_chunking_increment_is_coprime_to_2_and_3():
    # property pinned by the reference's tests/batching_tests.py:12-68
    for n_chunks in range(1, 200):
        inc = batching.calculate_increment_for_chunking_simulation_problems(10 ** 9 + 7, n_chunks)
        assert inc % 2 and inc % 3
        assert abs(inc - n_chunks) <= 2


@pytest.mark.parametrize('example,case_name', [
    ('output2_example1', 'example1'),
    ('output5_example1_fixed_nodes', 'example1_fixed_nodes'),
    ('output6_example1_perturbations', 'example1_perturbations'),
])
def test_simulation_csv_byte_exact(tmp_path, example, case_name):
    """States from the reference (simulate.json) through OUR writers == the reference's example CSVs."""
    case = next(c for c in load('simulate.json') if c['name'] == case_name)
    n = len(case['cfg']['node_names'])
    sims = []
    for i, problem in zip(case['indices'], case['problems']):
        states = [decode_state(int(code), n) for code in case['trajectories'][i]]
        pert = {}
        for t, node, v in problem['pert']:
            pert.setdefault(t, {})[node] = bool(v)
        sims.append(Simulation(states, {node: bool(v) for node, v in problem['fixed']}, pert))
    output_simulations(sims, case['cfg']['node_names'], str(tmp_path))
    for name in ('simulation_summaries.csv', 'simulations.csv'):
        assert read(tmp_path / name) == read(os.path.join(EXAMPLES, example, name)), name


def test_attractor_csv_format(tmp_path):
    case = next(c for c in load('attract_examples.json') if c['name'] == 'example2')
    rows = case['master']['rows']
    attractors = []
    for r in rows:
        f = r['frequency']
        s1 = round(r['mean'] * f)
        s2 = round(r['m2'] + s1 * s1 / f)
        attractors.append(AggregatedAttractor(int(r['key']), r['length'], f, s1, s2,
                                              [decode_state(int(c), 3) for c in r['states']]))
    output_attractors(attractors, 8, {}, ['A', 'B', 'C'], 8, float('inf'), float('inf'), str(tmp_path))
    text = read(tmp_path / 'attractor_summaries.csv').decode().split('\r\n')
    assert text[0] == 'attractor_id,length,trajectory_length_mean,trajectory_length_SD,relative_frequency'
    cells = text[1].split(',')
    assert cells[:3] == ['attractor1', '3', '0.7142857142857143'] and cells[4] == '0.875'
    # SD is derived from exact integer sums; the reference's float accumulation differs in the last digit
    assert abs(float(cells[3]) - 0.7559289460184545) < 1e-12
    assert text[2] == 'attractor2,1,0.0,nan,0.125'
    # states file: same rows as the reference's example output (ids there predate the current naming)
    ours = read(tmp_path / 'attractors.csv').decode().split('\r\n')
    ref = read(os.path.join(EXAMPLES, 'output3_example2', 'attractors.csv')).decode().split('\r\n')
    assert ours == ref
    # problems without attractor: first summary row, '<= inf' for unset caps (cli passes inf)
    output_attractors(attractors[:1], 7, {0: True}, ['A', 'B', 'C'], 8, float('inf'), 5, str(tmp_path))
    text = read(tmp_path / 'attractor_summaries.csv').decode().split('\r\n')
    assert text[1] == 'no_attractor,<= inf,<= 5,,0.125'
    assert read(tmp_path / 'attractors.csv').decode().split('\r\n')[1] == 'attractor1,t,1_,1,0'


def test_state_formatting():
    assert format_state([True, False, True], 2, {1: False}, {2: {0: True, 1: False}}) == ['1*', '0_*', '1']
    assert write_states([[False], [True]], {}, {}, 'x', None) == [['x', '0', '0'], ['x', '1', '1']]
    assert write_states([[False]], {}, {}, '', ['t']) == [['t', '0']]
    assert list_texts(['a']) == 'a' and list_texts(['a', 'b']) == 'a and b' and list_texts(['a', 'b', 'c']) == 'a, b, and c'


def test_weighted_spearman_equals_expanded_data():
    from scipy import stats
    rng = np.random.default_rng(3)
    data = rng.integers(0, 4, size=(7, 4)).astype(float) / 3
    freq = rng.integers(1, 6, size=7)
    expanded = np.repeat(data, freq, axis=0)
    rho, p = compute_frequency_spearmanrho(data, freq)
    for a in range(4):
        assert np.allclose(weighted_ranks(data[:, a], freq), np.asarray(stats.rankdata(expanded[:, a]))[np.cumsum(freq) - 1])
        for b in range(a + 1, 4):
            ref = stats.spearmanr(expanded[:, a], expanded[:, b])
            assert np.isclose(rho[a, b], ref.statistic) and np.isclose(p[a, b], ref.pvalue)


def test_node_correlations_csv_matches_example(tmp_path):
    case = next(c for c in load('attract_examples.json') if c['name'] == 'example2')
    attractors = [AggregatedAttractor(int(r['key']), r['length'], r['frequency'], 0, 0,
                                      [decode_state(int(c), 3) for c in r['states']]) for r in case['master']['rows']]
    rho, p = find_node_correlations(attractors)
    output_node_correlations(rho, p, 0.05, ['A', 'B', 'C'], str(tmp_path))
    assert read(tmp_path / 'node_correlations.csv') == read(os.path.join(EXAMPLES, 'output3_example2', 'node_correlations.csv'))
    assert find_node_correlations(attractors[:1]) is None


def test_cli_surface():
    runner = CliRunner()
    res = runner.invoke(cli, ['--help'])
    assert res.exit_code == 0 and all(c in res.output for c in ('simulate', 'attract', 'target'))
    for cmd, flags in (('simulate', ['-t', '--simulation-time']),
                       ('attract', ['-t', '-a', '-r', '-k', '-c', '-x', '-p']),
                       ('target', ['-t', '-n'])):
        out = runner.invoke(cli, [cmd, '--help']).output
        for flag in flags + ['-o', '-b', '-d', '--no-pdf', '--no-csv', '--print-png', '--print-svg']:
            assert flag in out, (cmd, flag)
    assert runner.invoke(cli, ['simulate', os.path.join(EXAMPLES, 'output2_example1', 'example1.yaml')]).exit_code != 0


# ----------------------------------------------------------------------------- reference batch order (SURVEY f-2)

def _batch_cases():
    return load("batch_seeds.json")


def test_batch_seeds_match_reference():
    from boolsi_amd import batching
    from boolsi_amd.input import parse_input_text
    from boolsi_amd.constants import Mode
    g = _batch_cases()
    for case in g['cases']:
        cfg = parse_input_text(g['yaml'][case['space']], g['max_t'], Mode.SIMULATE)
        n = cfg['total combination count']
        variations = cfg['simulation problem variations']
        seeds = list(batching.generate_simulation_problem_batch_seeds(
            variations, case['n_chunks'], n, case['batches_per_chunk']))
        assert len(seeds) == len(case['batches']) == case['n_batches_counted'] == \
            batching.count_simulation_problem_batches(case['n_chunks'], n, case['batches_per_chunk'])
        assert batching.calculate_increment_for_chunking_simulation_problems(n, case['n_chunks']) == case['increment']
        for (first, size, inc, radices), want in zip(seeds, case['batches']):
            assert (first, size, inc, radices) == (want['first'], want['size'], want['increment'], want['radices'])
        layout = batching.BatchLayout(n, case['n_chunks'], case['batches_per_chunk'])
        order = batching.reference_order(layout)
        assert order == [i for b in case['batches'] for i in b['order']]
        assert sorted(order) == list(range(n))
        # closed-form position == place in the listing
        keys = [layout.position(i) for i in order]
        assert keys == sorted(keys)
        assert keys == [(bi, k) for bi, b in enumerate(case['batches']) for k in range(b['size'])]


def test_reference_listing_labels():
    """Problems of every batch (initial state, fixed nodes, perturbations) as the reference generates them."""
    from boolsi_amd import batching
    from boolsi_amd.input import parse_input_text
    from boolsi_amd.constants import Mode
    from boolsi_amd.model import encode_state
    g = _batch_cases()
    want = g['listed']
    cfg = parse_input_text(g['yaml'][want['space']], g['max_t'], Mode.SIMULATE)
    layout = batching.BatchLayout(cfg['total combination count'], want['n_chunks'], want['batches_per_chunk'])
    got = list(batching.reference_order(layout))
    flat = [p for b in want['batches'] for p in b]
    assert len(got) == len(flat)
    for index, p in zip(got, flat):
        init, fixed, pert = batching.problem_from_index(index, cfg['origin simulation problem'],
                                                        cfg['simulation problem variations'])
        assert str(encode_state(set(), init)[0]) == p['initial_code']
        assert sorted([n, int(v)] for n, v in fixed.items()) == p['fixed']
        assert sorted([t, n, int(v)] for t, d in pert.items() for n, v in d.items()) == p['pert']


def test_wide_records_pack_merge_and_refuse_to_overflow():
    """bsx_attr_rec2 on the host side: 128-bit counts, 192 / 256-bit sums through table_from_merged -> merge_tables, the
    narrow record's overflow check, and the flat 128-bit index of the binding."""
    import numpy as np
    import pytest
    from boolsi_amd import _lib
    from boolsi_amd.attract import merge_tables, record_ints, table_from_merged
    merged = {(1 << 200) + 5: [16, (1 << 100) + 3, (1 << 150) + 1, (1 << 250) + 9], 7: [1, 1, 0, 0]}
    wide = table_from_merged(merged, _lib.ATTR_REC2)
    assert wide.dtype == _lib.ATTR_REC2 and len(wide) == 2
    assert merge_tables([wide]) == merged
    assert merge_tables([wide, wide]) == {k: [v[0], 2 * v[1], 2 * v[2], 2 * v[3]] for k, v in merged.items()}
    assert sorted(record_ints(a) for a in wide) == sorted((k, *v) for k, v in merged.items())
    with pytest.raises(OverflowError):
        table_from_merged(merged, _lib.ATTR_REC)                    # a count of 2^100 is not a bsx_attr_rec
    with pytest.raises(OverflowError):
        table_from_merged({1: [1, 1 << 128, 0, 0]}, _lib.ATTR_REC2)
    narrow = table_from_merged({7: [3, (1 << 64) - 1, (1 << 64) - 2, (1 << 128) - 1]}, _lib.ATTR_REC)
    assert merge_tables([narrow])[7] == [3, (1 << 64) - 1, (1 << 64) - 2, (1 << 128) - 1]
    with pytest.raises(RuntimeError):
        merge_tables([narrow, table_from_merged({7: [4, 1, 0, 0]}, _lib.ATTR_REC2)])      # one key, two lengths
    assert int(_lib.U128.of((1 << 127) + 12345)) == (1 << 127) + 12345
    for bad in (-1, 1 << 128):
        with pytest.raises(ValueError):
            _lib.U128.of(bad)
