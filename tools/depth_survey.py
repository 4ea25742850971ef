#!/usr/bin/env python
"""Does the cascade of cube passes pay on networks other than the north star?  Random n = 64 networks: K = 2 (eight
seeds, 2^32 problems each) and K = 3 (chaotic: four seeds, 2^26 problems), default (levels) against BSX_CUBE_DEPTH=1
(first update only) -- same table, times.  (K = 1 is left out: loops of copy / invert rules give state cycles far
longer than -t 4096, so every trajectory runs into the cap and 2^36 of them take minutes whatever the path.)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine, key_to_int
from boolsi_amd.input import parse_input_text


def rows(table):
    return sorted((key_to_int(a['key']), int(a['length']), int(a['count']), int(a['sum_l'])) for a in table)


eng = Engine(0)
for k, seeds, count in ((2, range(11, 19), 1 << 32), (3, range(11, 15), 1 << 26)):
    for seed in seeds:
        cfg = parse_input_text(synth.network_yaml(64, k, seed), 4096, Mode.ATTRACT)
        net, space = compile_problem(cfg)
        res = {}
        for depth in ('8', '1'):
            os.environ.pop('BSX_CUBE_DEPTH', None)  # '8': the default (levels, top level chosen by the engine's cost estimate)
            if depth == '1':
                os.environ['BSX_CUBE_DEPTH'] = depth
            eng.set_problem(net, space)
            try:
                eng.attract(0, count, 4096)         # discovery, buffers
                t0 = time.perf_counter()
                r = eng.attract(count, count, 4096)
                res[depth] = (time.perf_counter() - t0, r)
            except Exception as e:                  # e.g. too many attractors for the caller's table
                res[depth] = (None, str(e)[:80])
        a, b = res['8'], res['1']
        if a[0] is None or b[0] is None:
            print(json.dumps({'k': k, 'seed': seed, 'error': [str(a[1])[:80], str(b[1])[:80]]}), flush=True)
            continue
        print(json.dumps({'k': k, 'seed': seed, 'attractors': len(a[1].table), 'same_table': rows(a[1].table) == rows(b[1].table),
                          'levels_ms': a[0] * 1e3, 'depth1_ms': b[0] * 1e3, 'levels_launches': a[1].stats['kernel_launches'],
                          'levels_executed': a[1].stats['executed_steps'], 'depth1_executed': b[1].stats['executed_steps']}), flush=True)
eng.close()
