#!/usr/bin/env python
"""Differential check of the cascade's machinery at sizes the oracle cannot reach (GPU box):
random n = 64 networks, aligned blocks of 2^LOG2 problems -- default engine (sub-blocks where it chooses them, per-parent
level, side streams) against forced sub-blocks, no sub-blocks, per-child depth-1 level, one stream: identical tables,
no-attractor counts and reference step counts.      python tools/fuzz_scale.py [log2 = 40] [n_networks = 6] [first_seed = 300]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boolsi_amd import synth  # noqa: E402
from boolsi_amd.compile import compile_problem  # noqa: E402
from boolsi_amd.constants import Mode  # noqa: E402
from boolsi_amd.engine import Engine  # noqa: E402
from boolsi_amd.input import parse_input_text  # noqa: E402

KNOBS = ('BSX_CUBE_SPLIT', 'BSX_CUBE_LEAF', 'BSX_CUBE_STREAMS', 'BSX_CUBE_DEPTH')
VARIANTS = [{}, {'BSX_CUBE_SPLIT': '1'}, {'BSX_CUBE_SPLIT': '0'}, {'BSX_CUBE_SPLIT': '1', 'BSX_CUBE_LEAF': '0', 'BSX_CUBE_STREAMS': '1'},
            {'BSX_CUBE_SPLIT': '1', 'BSX_CUBE_DEPTH': '3', 'BSX_CUBE_STREAMS': '2'}]


def rows(r):
    return sorted((tuple(int(x) for x in a['key']), int(a['length']), tuple(int(x) for x in a['count']),
                   tuple(int(x) for x in a['sum_l']), tuple(int(x) for x in a['sum_l2'])) for a in r.table)


def main():
    log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n_nets = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    bad = 0
    for seed in range(seed0, seed0 + n_nets):
        k = 2 if seed % 3 else 1
        cfg = parse_input_text(synth.network_yaml(64, k, seed), 4096, Mode.ATTRACT)
        net, space = compile_problem(cfg)
        first = ((0x9E3779B97F4A7C15 * (seed + 1)) % (1 << 64)) & ~((1 << log2) - 1)
        # only networks whose blocks collapse: anything else steps 2^LOG2 problems one by one
        for key in KNOBS:
            os.environ.pop(key, None)
        eng = Engine(0)
        eng.set_problem(net, space)
        probe = eng.attract2(first, 1 << 28, 4096)
        eng.close()
        if probe.stats['executed_steps'] > (1 << 28) // 8 or probe.stats['kernel_ms'] > 50:
            print('seed {} K={}: does not collapse ({:.3e} updates for 2^28 problems), skipped'.format(seed, k, probe.stats['executed_steps']), flush=True)
            continue
        if log2 > 48:       # (and a block 256 times smaller must be quick, or the block itself takes minutes)
            eng = Engine(0)
            eng.set_problem(net, space)
            t0 = time.perf_counter()
            eng.attract2(first, 1 << (log2 - 8), 4096)
            dt = time.perf_counter() - t0
            eng.close()
            if dt > 0.25:
                print('seed {} K={}: 2^{} problems already take {:.0f} ms, skipped'.format(seed, k, log2 - 8, dt * 1e3), flush=True)
                continue
        ref = None
        for env in VARIANTS:
            for key in KNOBS:
                os.environ.pop(key, None)
            os.environ.update(env)
            eng = Engine(0)
            eng.set_problem(net, space)
            t0 = time.perf_counter()
            r = eng.attract2(first, 1 << log2, 4096)
            dt = time.perf_counter() - t0
            eng.close()
            got = (rows(r), r.n_no_attractor, r.stats['state_steps'])
            print('seed {} K={} 2^{} {}: {} attractors, {:.1f} ms, {} launches, executed {:.3e}'.format(
                seed, k, log2, env or 'default', len(r.table), dt * 1e3, r.stats['kernel_launches'], r.stats['executed_steps']), flush=True)
            if ref is None:
                ref = got
                if dt > 0.3:        # (a block this network makes expensive: one variant is enough)
                    print('seed {}: slow, the other variants are skipped'.format(seed), flush=True)
                    break
            elif got != ref:
                bad += 1
                print('MISMATCH seed', seed, env, flush=True)
    print('{} mismatches'.format(bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
