"""Shared helpers of the test-suite: golden vector loading and table comparison."""
import os
import json
from math import inf

from boolsi_amd.constants import Mode, RANGE_CODE
from boolsi_amd.input import parse_input_text
from boolsi_amd.compile import compile_problem, truth_table_to_mask

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MODES = {'simulate': Mode.SIMULATE, 'attract': Mode.ATTRACT, 'target': Mode.TARGET}


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def t_of(v):
    return inf if v is None else v


def cfg_summary(cfg):
    """Same flat description of a parsed input as oracle/gen_golden.py:cfg_json, from OUR parser."""
    init, fixed, pert = cfg['origin simulation problem']
    iv, fv, pv = cfg['simulation problem variations']
    out = {
        'node_names': cfg['node names'],
        'preds': cfg['incoming node lists'],
        'tt': [str(truth_table_to_mask(t, len(p))) for t, p in zip(cfg['truth tables'], cfg['incoming node lists'])],
        'origin_state': [int(b) for b in init],
        'origin_fixed': sorted([int(n), int(v)] for n, v in fixed.items()),
        'origin_pert': sorted([int(t), int(n), int(v)] for t, d in pert.items() for n, v in d.items()),
        'any_nodes': list(iv),
        'fixed_var': [[int(n), RANGE_CODE[r]] for n, r in fv],
        'pert_var': [[int(t), int(n), RANGE_CODE[r]] for t, n, r in pv],
        'n_problems': str(cfg['total combination count']),
    }
    if cfg.get('target node set') is not None:
        out['target_nodes'] = sorted(cfg['target node set'])
        out['target_code'] = str(cfg['target substate code'])
    return out


def compile_case(case):
    """Parse the case's YAML with OUR front end and lower it to engine tables."""
    cfg = parse_input_text(case['yaml'], t_of(case.get('max_t')), MODES[case['mode']])
    net, space = compile_problem(cfg)
    return cfg, net, space


def contiguous_runs(indices):
    """[(first, count, offset_in_list)] of maximal runs of consecutive indices."""
    runs = []
    i = 0
    while i < len(indices):
        j = i
        while j + 1 < len(indices) and indices[j + 1] == indices[j] + 1:
            j += 1
        runs.append((indices[i], j - i + 1, i))
        i = j + 1
    return runs
