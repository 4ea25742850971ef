#!/usr/bin/env python
"""Plain pool tiles (BSX_CUBES=0) of the north-star network: kernel ms per 2^28 problems (regression probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['BSX_CUBES'] = '0'
from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text
eng = Engine(0)
for name, text, base in (('north star', synth.north_star_yaml(), 0x0123456789ABCDEF & ~((1 << 28) - 1)), ('config3', synth.config3_yaml(), 0)):
    cfg = parse_input_text(text, 4096, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    for i in range(4):
        r = eng.attract(base + i * (1 << 28), 1 << 28, 4096)
        print('{} tile {}: kernels {:.3f} ms / {} launches, executed {}'.format(name, i, r.stats['kernel_ms'], r.stats['kernel_launches'], r.stats['executed_steps']))
eng.close()
