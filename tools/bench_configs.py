#!/usr/bin/env python
"""
Secondary measurements: the BASELINE.json configs other than the headline one, on one GPU.
    python tools/bench_configs.py [--quick]
Prints one JSON line per config (kernel time from HIP events inside the engine, wall time around
the call).  Not the driver's bench (that is /bench.py).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from boolsi_amd import synth  # noqa: E402
from boolsi_amd.attract import run_attract_range  # noqa: E402
from boolsi_amd.compile import compile_problem, code_to_words  # noqa: E402
from boolsi_amd.constants import Mode  # noqa: E402
from boolsi_amd.engine import Engine  # noqa: E402
from boolsi_amd.input import parse_input_text  # noqa: E402


def main():
    quick = '--quick' in sys.argv
    eng = Engine(0)
    out = []

    def attract(name, text, first, count, max_t):
        cfg = parse_input_text(text, max_t, Mode.ATTRACT)
        net, space = compile_problem(cfg)
        eng.set_problem(net, space)
        run_attract_range(eng, first, min(count, 1 << 20), max_t)            # warm the cycle cache
        t0 = time.perf_counter()
        merged, none, st = run_attract_range(eng, first, count, max_t)
        dt = time.perf_counter() - t0
        out.append({'config': name, 'mode': 'attract', 'n': net.n_nodes, 'problems': count, 'attractors': len(merged),
                    'no_attractor': none, 'wall_s': dt, 'kernel_ms': st['kernel_ms'],
                    'node_updates_per_s': st['state_steps'] * net.n_nodes / dt,
                    'executed_node_updates_per_s': st['executed_steps'] * net.n_nodes / dt,
                    'problems_per_s': count / dt})
        print(json.dumps(out[-1]), flush=True)

    cambium2 = open(os.path.join(ROOT, 'tests', 'golden', 'cambium2.yaml')).read()
    attract('config2 cambium2 full sweep', cambium2, 0, 1 << 30, float('inf'))
    attract('config3 synthetic n=32 K=2', synth.config3_yaml(), 0, 1 << (28 if quick else 32), 4096)

    # config 4: target, n = 64, 8 knock-out variants x 2^28 initial states
    cfg = parse_input_text(synth.config4_yaml(), 1024, Mode.TARGET)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    mask = code_to_words(sum(1 << n for n in cfg['target node set']), net.n_words)
    code = code_to_words(cfg['target substate code'], net.n_words)
    count = 1 << (26 if quick else 31)
    t0 = time.perf_counter()
    n_hits, kms, steps = 0, 0.0, 0
    for first in range(0, count, 1 << 26):
        hits, st = eng.target(first, min(1 << 26, count - first), 1024, mask, code, cap=1 << 26)
        n_hits += len(hits); kms += st['kernel_ms']; steps += st['executed_steps']
    dt = time.perf_counter() - t0
    out.append({'config': 'config4 synthetic n=64 target, 8 variants', 'mode': 'target', 'n': 64, 'problems': count,
                'hits': n_hits, 'wall_s': dt, 'kernel_ms': kms, 'executed_node_updates_per_s': steps * 64 / dt,
                'problems_per_s': count / dt})
    print(json.dumps(out[-1]), flush=True)

    # config 5: simulate -t 10000, n = 128, K = 3, perturbation schedule; digest sink
    cfg = parse_input_text(synth.config5_yaml(), 10000, Mode.SIMULATE)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    count = 1 << (16 if quick else 20)
    for label, kw in (('final states + digests (per-lane kernel)', dict(digest=True)),
                      ('final states (bit-sliced kernel)', dict(digest=False))):
        eng.simulate(0, min(count, 1 << 14), 10000, trajectories=False, **kw)       # first launch of the kernel: untimed
        t0 = time.perf_counter()
        _, fin, dig, st = eng.simulate(0, count, 10000, trajectories=False, **kw)
        dt = time.perf_counter() - t0
        out.append({'config': 'config5 synthetic n=128 simulate -t 10000 (slice of 2^26), ' + label, 'mode': 'simulate',
                    'n': 128, 'problems': count, 'wall_s': dt, 'kernel_ms': st['kernel_ms'],
                    'node_updates_per_s': st['state_steps'] * 128 / dt,
                    'kernel_node_updates_per_s': st['state_steps'] * 128 / (st['kernel_ms'] * 1e-3),
                    'seconds_for_2^26': dt * (1 << 26) / count})
        print(json.dumps(out[-1]), flush=True)
    eng.close()


if __name__ == '__main__':
    main()
