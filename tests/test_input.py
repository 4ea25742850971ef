"""
Front-end parity: boolsi_amd.input must accept / reject exactly what the reference's
boolsi/input.py does and produce identical tables (node order, sorted predecessor lists,
truth tables in S2 bit order, origin problem, heap-ordered variation lists, N).
Vectors: tests/golden/input.json (generated from the reference by oracle/gen_golden.py).
"""
import pytest

from boolsi_amd.input import parse_input_text, DuplicateKeyError, parse_raw_input_time_steps, \
    generate_safe_node_names, count_simulation_problems
from boolsi_amd.constants import NodeStateRange
from util import load, t_of, MODES, cfg_summary

GOLD = load('input.json')


@pytest.mark.parametrize('entry', GOLD['parsed'], ids=lambda e: '{}-{}'.format(e['name'], e['mode']))
def test_example_inputs_parse_identically(entry):
    if entry.get('raises'):
        with pytest.raises((ValueError, KeyError)):
            parse_input_text(entry['yaml'], t_of(entry['max_t']), MODES[entry['mode']])
        return
    cfg = parse_input_text(entry['yaml'], t_of(entry['max_t']), MODES[entry['mode']])
    assert cfg_summary(cfg) == entry['cfg']


@pytest.mark.parametrize('entry', GOLD['verdicts'], ids=lambda e: e['name'])
def test_malformed_inputs_get_the_reference_verdict(entry):
    args = (entry['yaml'], t_of(entry['max_t']), MODES[entry['mode']])
    if entry['raises'] is None:
        assert cfg_summary(parse_input_text(*args)) == entry['cfg']
    elif entry['raises'] == 'DuplicateKeyError':
        with pytest.raises(DuplicateKeyError):
            parse_input_text(*args)
    elif entry['raises'] == 'ValueError':
        with pytest.raises(ValueError):
            parse_input_text(*args)
    else:   # the reference fails with a non-validation error (e.g. AttributeError): so must we, loosely
        with pytest.raises(Exception):
            parse_input_text(*args)


def test_time_steps():
    # reference tests/input_tests.py:1096-1181 cover the same grammar
    assert list(parse_raw_input_time_steps('1-3, 5,7')) == [1, 2, 3, 5, 7]
    assert list(parse_raw_input_time_steps(' 4 , 6 ,')) == [4, 6]
    assert list(parse_raw_input_time_steps('10')) == [10]
    assert list(parse_raw_input_time_steps('2 - 2')) == [2]
    for bad in ('1;2', '3-1', 'a', '1-', '-1', '1--2'):
        with pytest.raises(ValueError):
            list(parse_raw_input_time_steps(bad))


def test_safe_node_names():
    assert generate_safe_node_names(['A', 'B']) == ['node0', 'node1']
    assert generate_safe_node_names(['node0', 'B']) == ['node_0', 'node_1']
    assert generate_safe_node_names(['node_1', 'node0', 'x']) == ['node__0', 'node__1', 'node__2']


def test_count_simulation_problems():
    r = NodeStateRange
    assert count_simulation_problems([], [], []) == 1
    assert count_simulation_problems([0, 1, 2], [], []) == 8
    assert count_simulation_problems([0], [(1, r.MAYBE_TRUE_OR_FALSE), (2, r.MAYBE_FALSE)],
                                     [(3, 0, r.TRUE_OR_FALSE), (4, 1, r.MAYBE_TRUE_OR_FALSE)]) == 2 * 3 * 2 * 2 * 3
    assert count_simulation_problems(list(range(70)), [], []) == 2 ** 70
