"""
End-to-end on the GPU through the CLI: the reference's example commands
(examples/output*/boolsi.log "Run parameters") must reproduce its committed CSV outputs byte for
byte (simulate / target), and its attractor tables (attract; float columns within 1e-9).
"""
import csv
import math
import os
import subprocess
import sys

import pytest

from util import GOLDEN, load

pytestmark = pytest.mark.gpu
EXAMPLES = os.path.join(GOLDEN, 'examples')
ROOT = os.path.dirname(GOLDEN.rstrip('/')).rsplit('/tests', 1)[0]


def run_cli(args, out_dir):
    cmd = [sys.executable, '-m', 'boolsi_amd'] + args + ['-o', out_dir]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    return res.stdout


def read(path):
    with open(path, 'rb') as f:
        return f.read()


@pytest.mark.parametrize('example,args', [
    ('output2_example1', ['simulate', 'example1.yaml', '-t', '5']),
    ('output4_example3', ['target', 'example3.yaml']),
    ('output5_example1_fixed_nodes', ['simulate', 'example1_fixed_nodes.yaml', '-t', '5']),
    ('output6_example1_perturbations', ['simulate', 'example1_perturbations.yaml', '-t', '5']),
    ('output7_cambium1', ['simulate', 'cambium1.yaml', '-t', '10']),
])
def test_example_commands_reproduce_reference_csvs(tmp_path, example, args):
    src = os.path.join(EXAMPLES, example)
    out = run_cli([args[0], os.path.join(src, args[1])] + args[2:], str(tmp_path))
    for name in ('simulation_summaries.csv', 'simulations.csv'):
        assert read(tmp_path / name) == read(os.path.join(src, name)), name
    assert 'Read Boolean network of' in out and 'Bye!' in out
    assert os.path.exists(tmp_path / 'boolsi.log') and os.path.exists(tmp_path / args[1])


def test_attract_example2(tmp_path):
    src = os.path.join(EXAMPLES, 'output3_example2')
    out = run_cli(['attract', os.path.join(src, 'example2.yaml')], str(tmp_path))
    assert 'Single process will be used to find attractors from 8 initial conditions...' in out
    assert 'Found 2 attractors.' in out
    assert read(tmp_path / 'attractors.csv') == read(os.path.join(src, 'attractors.csv'))
    assert read(tmp_path / 'node_correlations.csv') == read(os.path.join(src, 'node_correlations.csv'))
    rows = list(csv.reader(open(tmp_path / 'attractor_summaries.csv')))
    assert rows[1][:2] == ['attractor1', '3'] and rows[1][4] == '0.875'
    assert math.isclose(float(rows[1][2]), 0.7142857142857143, rel_tol=1e-12)
    assert math.isclose(float(rows[1][3]), 0.7559289460184545, rel_tol=1e-9)
    assert rows[2] == ['attractor2', '1', '0.0', 'nan', '0.125']


def test_attract_caps_and_no_attractor_row(tmp_path):
    case = next(c for c in load('attract_toy.json') if c['name'] == 'toyB_sas1_t3_ainf')
    path = tmp_path / 'toy.yaml'
    path.write_text(case['yaml'])
    out = run_cli(['attract', str(path), '-t', '3', '-c'], str(tmp_path / 'out'))
    found = sum(1 for r in case['per_problem'] if r[0])
    assert 'No attractor can be detected in 3 or less time steps from {:.2%} initial conditions.'.format(
        1 - found / len(case['per_problem'])) in out
    rows = list(csv.reader(open(tmp_path / 'out' / 'attractor_summaries.csv')))
    assert rows[1][:4] == ['no_attractor', '<= inf', '<= 3', '']
    assert float(rows[1][4]) == 1 - found / len(case['per_problem'])
    agg = case['aggregate']
    assert [r[1] for r in rows[2:]] == [str(a[1]) for a in agg]
    assert [float(r[4]) for r in rows[2:]] == [a[2] / len(case['per_problem']) for a in agg]


def test_target_n_limit(tmp_path):
    case = next(c for c in load('target.json') if c['name'] == 'toyB_target_variations')
    path = tmp_path / 'toy.yaml'
    path.write_text(case['yaml'])
    out = run_cli(['target', str(path), '-t', '12', '-n', '5'], str(tmp_path / 'out'))
    assert 'find 5 that reach target states' in out and 'At least 5 simulations reach target state.' in out
    rows = list(csv.reader(open(tmp_path / 'out' / 'simulation_summaries.csv')))
    first5 = [r for r in case['per_problem'] if r[0]][:5]
    assert [int(r[1]) for r in rows[1:]] == [r[1] for r in first5]


def test_input_error_is_logged_not_raised(tmp_path):
    path = tmp_path / 'bad.yaml'
    path.write_text('nodes: [A]\nupdate rules: {A: B}\ninitial state: {A: any}\n')
    out = run_cli(['simulate', str(path), '-t', '3'], str(tmp_path / 'out'))
    assert 'ERROR Input validation failed' in out and "Unknown expression 'B'" in out and 'Bye!' in out
