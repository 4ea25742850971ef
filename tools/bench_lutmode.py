#!/usr/bin/env python
"""A/B of the LUT modes on a 128-node K = 2 network (byte table 128 KiB = one workgroup per CU, nibble table 16 KiB)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth  # noqa: E402
from boolsi_amd.compile import compile_problem  # noqa: E402
from boolsi_amd.constants import Mode  # noqa: E402
from boolsi_amd.engine import Engine  # noqa: E402
from boolsi_amd.input import parse_input_text  # noqa: E402

eng = Engine(0)
for seed in (1280, 1281, 1282):
    bits = synth.seeded_bits(128, seed)
    text = synth.network_yaml(128, 2, seed, initial={i: str(bits[i]) for i in range(40, 128)})
    cfg = parse_input_text(text, 4096, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    for mode in ('1', '2'):
        os.environ['BSX_LUT_MODE'] = mode
        eng.set_problem(net, space)
        os.environ.pop('BSX_LUT_MODE')
        count = 1 << 24
        eng.attract(0, count, 4096)
        r = eng.attract(count, count, 4096)
        st = r.stats
        print('seed %d mode %s: %d attractors, none %d, %.2f ms per 2^24 problems, %d launches, %.1f executed steps / problem' % (
            seed, mode, len(r.table), r.n_no_attractor, st['kernel_ms'], st['kernel_launches'], st['executed_steps'] / count), flush=True)
