#!/usr/bin/env python
"""Where does a bench step's wall time go?  kernel time (HIP events) / C-ABI call / Python around it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 48
eng = Engine(0)
cfg = parse_input_text(synth.north_star_yaml(), 4096, Mode.ATTRACT)
net, space = compile_problem(cfg)
eng.set_problem(net, space)
batch = 1 << log2
base = 0x0123456789ABCDEF & ~(batch - 1)
for s in range(8):
    t0 = time.perf_counter()
    r = eng.attract(base + s * batch, batch, 4096)
    wall = (time.perf_counter() - t0) * 1e3
    print('step {}: wall {:.3f} ms, C call {:.3f} ms, kernels {:.3f} ms in {} launches, executed {}'.format(
        s, wall, r.stats['total_ms'], r.stats['kernel_ms'], r.stats['kernel_launches'], r.stats['executed_steps']))
eng.close()
