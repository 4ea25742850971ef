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


# ----------------------------------------------------------------------------- multi-process listing order (f-2)

def _sim_rows(sims):
    from boolsi_amd.model import encode_state
    return [{'states': [str(encode_state(set(), s)[0]) for s in sim.states],
             'fixed': sorted([int(k), int(v)] for k, v in sim.fixed_nodes.items()),
             'pert': sorted([int(t), int(k), int(v)] for t, d in sim.perturbed_nodes_by_t.items()
                            for k, v in d.items())} for sim in sims]


@pytest.mark.parametrize('case', load('listing.json'), ids=lambda c: '{}_np{}_b{}'.format(c['mode'], c['np'], c['b']))
def test_listing_order_of_multi_process_reference_run(case):
    """simulate_master / target_master with a BatchLayout list the simulations exactly as the reference's
    task farm stores them (tests/golden/listing.json)."""
    from math import inf
    from boolsi_amd.batching import batch_layout
    from boolsi_amd.constants import Mode
    from boolsi_amd.engine import Engine
    from boolsi_amd.input import parse_input_text
    from boolsi_amd.simulate import simulate_master
    from boolsi_amd.target import target_master
    mode = {'simulate': Mode.SIMULATE, 'target': Mode.TARGET}[case['mode']]
    cfg = parse_input_text(case['yaml'], case['max_t'], mode)
    n = cfg['total combination count']
    assert n == case['n_problems']
    listing = batch_layout(n, case['np'], case['b']) if case['np'] > 2 else None
    engine = Engine(0)
    try:
        if case['mode'] == 'simulate':
            sims = simulate_master(engine, cfg['origin simulation problem'], cfg['simulation problem variations'],
                                   cfg['incoming node lists'], cfg['truth tables'], case['max_t'], n,
                                   listing=listing)
        else:
            sims = target_master(engine, cfg['origin simulation problem'], cfg['simulation problem variations'],
                                 cfg['target substate code'], cfg['target node set'], cfg['incoming node lists'],
                                 cfg['truth tables'], inf, case['max_t'], n, listing=listing)
    finally:
        engine.close()
    assert _sim_rows(sims) == case['simulations']


def test_cli_reference_np_changes_order_only(tmp_path):
    case = next(c for c in load('listing.json') if c['mode'] == 'simulate' and c['np'] == 4)
    path = tmp_path / 'net.yaml'
    path.write_text(case['yaml'])
    run_cli(['simulate', str(path), '-t', str(case['max_t'])], str(tmp_path / 'a'))
    run_cli(['simulate', str(path), '-t', str(case['max_t']), '--reference-np', '4', '-b', '5'], str(tmp_path / 'b'))
    a = read(tmp_path / 'a' / 'simulations.csv').split(b'\n')
    b = read(tmp_path / 'b' / 'simulations.csv').split(b'\n')
    assert a != b and len(a) == len(b)
    # same simulations, renumbered: compare the multiset of per-simulation row blocks without the name column
    def blocks(lines):
        out = {}
        for line in lines[1:]:
            if line:
                name, rest = line.split(b',', 1)
                out.setdefault(name, []).append(rest)
        return sorted(map(tuple, out.values()))
    assert blocks(a) == blocks(b)


# ----------------------------------------------------------------------------- two ranks (one GPU shared, socket data plane)

def run_cli_world(args, out_dir, world=2, cwd=ROOT):
    """The CLI as `torch.distributed.run` starts it, `world` ranks sharing GPU 0 (BSX_DIST_BACKEND=socket for
    the data collectives: RCCL wants one device per rank)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world),
                   RANK=str(rank), LOCAL_RANK=str(rank), BSX_DIST_BACKEND='socket')
        env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
        cmd = [sys.executable, '-m', 'boolsi_amd'] + args + (['-o', out_dir] if out_dir else []) + ['--device', '0']
        procs.append(subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and not any('Exception caught' in o for o in outs), '\n'.join(outs)
    return outs


def test_two_ranks_write_the_same_files_as_one(tmp_path):
    sim = next(c for c in load('listing.json') if c['mode'] == 'simulate' and c['np'] == 4)
    tgt = next(c for c in load('listing.json') if c['mode'] == 'target' and c['np'] == 5)
    (tmp_path / 'sim.yaml').write_text(sim['yaml'])
    (tmp_path / 'tgt.yaml').write_text(tgt['yaml'])
    (tmp_path / 'att.yaml').write_text(load('attract_toy.json')[0]['yaml'])
    runs = [('sim', ['simulate', str(tmp_path / 'sim.yaml'), '-t', str(sim['max_t'])]),
            ('simr', ['simulate', str(tmp_path / 'sim.yaml'), '-t', str(sim['max_t']), '--reference-np', '4', '-b', '5']),
            ('tgt', ['target', str(tmp_path / 'tgt.yaml'), '-t', str(tgt['max_t'])]),
            ('tgtn', ['target', str(tmp_path / 'tgt.yaml'), '-t', str(tgt['max_t']), '-n', '7']),
            ('att', ['attract', str(tmp_path / 'att.yaml')])]
    for name, args in runs:
        run_cli(args, str(tmp_path / (name + '1')))
        outs = run_cli_world(args, str(tmp_path / (name + '2')))
        assert any('2 GPU' in o for o in outs), '\n'.join(outs)
        files = sorted(f for f in os.listdir(tmp_path / (name + '1')) if f.endswith('.csv'))
        assert files and files == sorted(f for f in os.listdir(tmp_path / (name + '2')) if f.endswith('.csv'))
        for f in files:
            assert read(tmp_path / (name + '1') / f) == read(tmp_path / (name + '2') / f), (name, f)


def test_ranks_agree_on_the_default_output_directory(tmp_path):
    """Without -o every process would name its own `output_<timestamp>`; rank 0's name must be the one, created
    before anybody writes, and no stray file may appear next to it (cli.py race)."""
    (tmp_path / 'att.yaml').write_text(load('attract_toy.json')[0]['yaml'])
    work = tmp_path / 'cwd'
    work.mkdir()
    outs = run_cli_world(['attract', str(tmp_path / 'att.yaml')], None, world=3, cwd=str(work))
    made = sorted(os.listdir(work))
    assert len(made) == 1 and made[0].startswith('output_') and os.path.isdir(work / made[0]), (made, outs)
    files = os.listdir(work / made[0])
    assert 'att.yaml' in files and 'attractor_summaries.csv' in files


def test_failing_rank_ends_the_job_with_a_nonzero_status(tmp_path):
    """One rank cannot open its GPU: it must not exit 0 while its peers wait in the merge."""
    (tmp_path / 'att.yaml').write_text(load('attract_toy.json')[0]['yaml'])
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', RANK=str(rank),
                   LOCAL_RANK=str(rank), BSX_DIST_BACKEND='socket')
        cmd = [sys.executable, '-m', 'boolsi_amd', 'attract', str(tmp_path / 'att.yaml'), '-o', str(tmp_path / 'out'),
               '--device', '0' if rank == 0 else '99']
        procs.append(subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert [p.returncode for p in procs] == [1, 1], outs


@pytest.mark.parametrize('fgraph', ['0', '1'])
def test_attract_cambium2_cli_reproduces_the_published_output(tmp_path, fgraph, monkeypatch):
    monkeypatch.setenv('BSX_FGRAPH', fgraph)       # 1: the same command through the functional-graph mode
    _cambium2_cli(tmp_path)


def _cambium2_cli(tmp_path):
    """`boolsi attract examples/cambium2.yaml` took the reference 45 h on 63 MPI workers (examples/output8_cambium2/
    boolsi.log); here the whole command -- 2^30 initial conditions, CSV output -- runs in seconds.  attractors.csv must
    be byte-identical; the summaries agree to 1e-9 (the reference accumulates mean / M2 in floating point, batch by
    batch) and in the order of the attractors."""
    out = run_cli(['attract', os.path.join(GOLDEN, 'cambium2.yaml'), '-c'], str(tmp_path))
    assert 'Found 39 attractors.' in out
    assert read(tmp_path / 'attractors.csv') == read(os.path.join(GOLDEN, 'cambium2_attractors.csv'))
    ours = list(csv.reader(open(tmp_path / 'attractor_summaries.csv')))
    ref = list(csv.reader(open(os.path.join(GOLDEN, 'cambium2_attractor_summaries.csv'))))
    assert len(ours) == len(ref) == 40 and ours[0] == ref[0]
    for a, b in zip(ours[1:], ref[1:]):
        assert a[:2] == b[:2] and float(a[4]) == float(b[4])
        assert math.isclose(float(a[2]), float(b[2]), rel_tol=1e-9) and math.isclose(float(a[3]), float(b[3]), rel_tol=1e-7)
