#!/usr/bin/env python
"""Where the time of one bsx_run_attract call goes besides the kernel (north-star network, 2^28 problems)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text
eng = Engine(0)
cfg = parse_input_text(synth.north_star_yaml(), 4096, Mode.ATTRACT)
net, space = compile_problem(cfg)
eng.set_problem(net, space)
base = 0x0123456789ABCDEF & ~((1 << 28) - 1)
for s in range(6):
    t0 = time.perf_counter()
    r = eng.attract(base + s * (1 << 28), 1 << 28, 4096)
    dt = (time.perf_counter() - t0) * 1e3
    print('call %d: python wall %.3f ms, C call %.3f ms, kernels %.3f ms, launches %d' % (s, dt, r.stats['total_ms'], r.stats['kernel_ms'], r.stats['kernel_launches']))
