#!/usr/bin/env python
"""North-star blocks of 2^a problems through bsx_run_attract2, one line per call (wall / kernel / dominant launch, executed
updates, host syncs); BSX_DEBUG=1 adds the library's per-level lines.    python tools/cascade_probe.py 48 52 56 60 63"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text

MAX_T = 4096
cfg = parse_input_text(synth.north_star_yaml(), MAX_T, Mode.ATTRACT)
net, space = compile_problem(cfg)
eng = Engine(0)
eng.set_problem(net, space)
eng.attract(0, 1 << 30, MAX_T)
reps = int(os.environ.get('REPS', '3'))
for a in [int(x) for x in sys.argv[1:]] or [48, 56]:
    batch = 1 << a
    base = 0x0123456789ABCDEF & ~(batch - 1)
    for i in range(reps):
        first = (base + i * batch) % (1 << 64)
        t0 = time.perf_counter()
        r = eng.attract2(first, batch, MAX_T)
        dt = (time.perf_counter() - t0) * 1e3
        s = r.stats
        print('2^{} at {:#x}: wall {:.3f} ms, call {:.3f}, kernels {:.3f}, dominant {:.3f}; executed {:.3e} (dominant {:.3e}), '
              '{} launches, {} syncs, {} attractors'.format(a, first, dt, s['total_ms'], s['kernel_ms'], s['dominant_ms'],
                                                           s['executed_steps'], s['dominant_executed_steps'], s['kernel_launches'],
                                                           s['host_syncs'], len(r.table)), flush=True)
eng.close()
