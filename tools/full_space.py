#!/usr/bin/env python
"""
The WHOLE north-star configuration, uncapped: `attract -t 4096` over all 2^64 initial states of the synthetic
n = 64, K = 2, seed 64 network (BASELINE.json caps the space to an index range because no stepping implementation
can enumerate it).  65 536 calls of 2^48 problems each, merged exactly (Python ints).

    python tools/full_space.py [log2_of_the_blocks_to_run = 16] [engines = 1] > profiles/r02_full_space.json
engines > 1: that many engine handles on the same GPU (one stream each), driven by one thread each over interleaved
blocks -- the short launches of one call overlap with those of the others.
Checks: every problem accounted for (sum of basin sizes + no-attractor count = 2^64 for the full run).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.attract import merge_tables
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text

MAX_T = 4096


def main():
    log2_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n_engines = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n_blocks = 1 << log2_blocks
    cfg = parse_input_text(synth.north_star_yaml(), MAX_T, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    block = 1 << 48
    engines = []
    for _ in range(n_engines):
        e = Engine(0)
        e.set_problem(net, space)
        e.attract(0, block, MAX_T)                              # discovery, scratch buffers
        engines.append(e)
    eng = engines[0]

    def sweep(e, blocks, out):
        merged, none, steps_ref, steps_exec, kernel_ms, launches, pending = {}, 0, 0, 0, 0.0, 0, []
        for i, b in enumerate(blocks):
            r = e.attract(b * block, block, MAX_T)
            pending.append(r.table)
            none += r.n_no_attractor
            steps_ref += r.stats['state_steps']
            steps_exec += r.stats['executed_steps']
            kernel_ms += r.stats['kernel_ms']
            launches += r.stats['kernel_launches']
            if len(pending) == 1024:
                merged = merge_tables([_as_table(merged)] + pending) if merged else merge_tables(pending)
                pending = []
                if e is eng:
                    print('block {} of {}: {:.1f} s'.format((i + 1) * n_engines, n_blocks, time.perf_counter() - t0), file=sys.stderr, flush=True)
        if pending:
            merged = merge_tables([_as_table(merged)] + pending) if merged else merge_tables(pending)
        out.append((merged, none, steps_ref, steps_exec, kernel_ms, launches))

    import threading
    t0 = time.perf_counter()
    parts = []
    threads = [threading.Thread(target=sweep, args=(e, range(i, n_blocks, n_engines), parts)) for i, e in enumerate(engines)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert len(parts) == n_engines
    merged = merge_tables([_as_table(p[0]) for p in parts])
    none, steps_ref, steps_exec = sum(p[1] for p in parts), sum(p[2] for p in parts), sum(p[3] for p in parts)
    kernel_ms, launches = sum(p[4] for p in parts), sum(p[5] for p in parts)
    dt = time.perf_counter() - t0
    problems = n_blocks * block
    total = sum(e[1] for e in merged.values()) + none
    assert total == problems, (total, problems)
    out = {
        'what': 'attract -t 4096 over {} of the 2^64 initial states of the north-star network (n = 64, K = 2, seed 64), '
                '{} calls of 2^48 problems'.format('ALL' if log2_blocks == 16 else '2^{}'.format(48 + log2_blocks), n_blocks),
        'problems': problems, 'engines_on_the_gpu': n_engines, 'wall_s': dt, 'kernel_s': kernel_ms / 1e3, 'kernel_launches': launches,
        'attractors_per_s': problems / dt, 'executed_updates': steps_exec, 'reference_equivalent_updates': steps_ref,
        'no_attractor': none,
        'attractors': [{'key_hex': '{:016x}'.format(k), 'length': e[0], 'basin': e[1], 'basin_share': e[1] / problems,
                        'mean_trajectory_l': e[2] / e[1]} for k, e in sorted(merged.items(), key=lambda kv: -kv[1][1])],
    }
    print(json.dumps(out, indent=1))
    for e in engines:
        e.close()


class _Rows(list):
    pass


def _as_table(merged):
    """A merged dict as a list of record-like dicts for merge_tables (counts may exceed 64 bits: no numpy)."""
    rows = _Rows()
    for k, (length, count, s1, s2) in merged.items():
        rows.append({'key': [(k >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)], 'length': length, 'count': count,
                     'sum_l': s1, 'sum_l2_lo': s2, 'sum_l2_hi': 0})
    return rows


if __name__ == '__main__':
    main()
