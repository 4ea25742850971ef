#!/usr/bin/env python
"""Config 4 (target, n = 64, 8 knock-out variants x 2^28) through the summary sink with and without the list of first hits;
BSX_DEBUG=1 adds the library's per-pass lines.    python tools/target_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.compile import compile_problem, code_to_words
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text

cfg = parse_input_text(synth.config4_yaml(), 1024, Mode.TARGET)
net, space = compile_problem(cfg)
eng = Engine(0)
eng.set_problem(net, space)
mask = code_to_words(sum(1 << n for n in cfg['target node set']), net.n_words)
code = code_to_words(cfg['target substate code'], net.n_words)
count = 1 << 31
for cap in (0, 1000, 0, 1000):
    t0 = time.perf_counter()
    n_hits, hist, hits, st = eng.target_summary(0, count, 1024, mask, code, hist_bins=1026, cap=cap)
    dt = (time.perf_counter() - t0) * 1e3
    print('cap {}: wall {:.3f} ms, kernels {:.3f} ms, {} launches, {} hits, executed {:.3e}'.format(
        cap, dt, st['kernel_ms'], st['kernel_launches'], n_hits, st['executed_steps']), flush=True)
eng.close()
