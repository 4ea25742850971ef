#!/usr/bin/env python
"""One attract call of config 3 (n = 32, all 2^32 problems), twice: wall vs kernel time (debug aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text
eng = Engine(0)
cfg = parse_input_text(synth.config3_yaml(), 4096, Mode.ATTRACT)
net, space = compile_problem(cfg)
eng.set_problem(net, space)
for i in range(3):
    t0 = time.perf_counter()
    r = eng.attract(0, 1 << 32, 4096)
    print('call {}: wall {:.3f} ms, C {:.3f} ms, kernels {:.3f} ms / {} launches'.format(
        i, (time.perf_counter() - t0) * 1e3, r.stats['total_ms'], r.stats['kernel_ms'], r.stats['kernel_launches']), file=sys.stderr)
eng.close()
