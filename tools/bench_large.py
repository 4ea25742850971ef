#!/usr/bin/env python
"""Attract on networks beyond 64 nodes (per-lane kernels, LUT in LDS or read through L2): kernel time per problem."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth  # noqa: E402
from boolsi_amd.compile import compile_problem  # noqa: E402
from boolsi_amd.constants import Mode  # noqa: E402
from boolsi_amd.engine import Engine  # noqa: E402
from boolsi_amd.input import parse_input_text  # noqa: E402

eng = Engine(0)
for n, k, free in ((100, 3, 40), (128, 2, 40), (128, 4, 40), (200, 3, 40), (256, 3, 40)):
    bits = synth.seeded_bits(n, n * 7 + k)
    text = synth.network_yaml(n, k, n * 10 + k, initial={i: str(bits[i]) for i in range(free, n)})
    cfg = parse_input_text(text, 2000, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    count = 1 << 22
    eng.attract(0, count, 2000)
    r = eng.attract(count, count, 2000)
    st = r.stats
    print('n %3d K %d: %d attractors, none %d, %.2f ms per 2^22 problems, %d launches, %.1f ref steps / problem, %.1f executed' % (
        n, k, len(r.table), r.n_no_attractor, st['kernel_ms'], st['kernel_launches'], st['state_steps'] / count,
        st['executed_steps'] / count), flush=True)
