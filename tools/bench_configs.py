#!/usr/bin/env python
"""
Secondary measurements: the BASELINE.json configs other than the headline one, on one GPU.
    python tools/bench_configs.py [--quick] [--no-cpu] [--only config4]       (--only: one config, e.g. under rocprofv3 --pmc)
One JSON line per config and path, each with
  roofline      SURVEY 8(d) basis (0.25 B per executed node update / kernel time / 8 TB/s; for the functional-graph
                mode the bytes its passes stream through HBM), <= 1 by construction;
  cpu_baseline  the CPU oracle (kind "port", OpenMP on all host cores) on a bounded sample of the same input.
Kernel time = HIP events inside the engine; wall time = around the call (uploads, merges, downloads).
Not the driver's bench (that is /bench.py).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from boolsi_amd import synth  # noqa: E402
from boolsi_amd.attract import run_attract_range, merge_tables  # noqa: E402
from boolsi_amd.compile import compile_problem, code_to_words  # noqa: E402
from boolsi_amd.constants import Mode  # noqa: E402
from boolsi_amd.engine import Engine  # noqa: E402
from boolsi_amd.input import parse_input_text  # noqa: E402

HBM_PEAK_GBS = 8000.0
CORES = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)


def pmc_of(key):
    """What the separate rocprofv3 --pmc passes of this config measured (tools/profile_configs.sh -> profiles/r03_pmc_<key>.json)."""
    path = os.path.join(ROOT, 'profiles', 'r03_pmc_{}.json'.format(key))
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def roofline(bytes_moved, kernel_ms, bound='hbm-normalised', note=None, pmc_key=None):
    achieved = bytes_moved / (kernel_ms * 1e-3) / 1e9
    r = {'bound': bound, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
         'traffic': None}
    if note:
        r['basis'] = note
    pmc = pmc_of(pmc_key) if pmc_key else None
    if pmc:
        # the unit the counters show busiest names the bound; the normalised HBM figure stays next to it
        ib = pmc.get('issue_bound') or {}
        fr = {k[:-10]: v for k, v in ib.items() if k.endswith('_busy_frac')}
        r['traffic'] = pmc.get('hbm_bytes_per_launch')
        r['traffic_source'] = 'profiles/r03_pmc_{}.json (separate --pmc passes, kernel {})'.format(pmc_key, pmc.get('kernel'))
        if fr and max(fr.values()) < 0.5 and ib.get('wave_wait_frac', 0) > 0.5:
            # no unit is even half busy and the waves mostly wait: a latency-bound launch (short, one dependent chain per lane)
            r['hbm_normalised'] = {'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS}
            r.update({'bound': 'latency (waves waiting {:.0%} of their cycles)'.format(ib['wave_wait_frac']), 'frac': max(fr.values()),
                      'achieved': max(fr.values()), 'peak': 1.0, 'unit': 'busiest unit\'s busy fraction of the launch (PMC)', 'busy_fractions': fr})
        elif fr:
            unit = max(fr, key=fr.get)
            r['hbm_normalised'] = {'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS}
            r.update({'bound': {'valu': 'valu-issue', 'salu': 'salu-issue', 'lds': 'lds-pipeline'}[unit], 'frac': min(fr[unit], 1.0),
                      'achieved': min(fr[unit], 1.0), 'peak': 1.0, 'unit': 'busy fraction of the launch (PMC)', 'busy_fractions': fr})
    return r


def main():
    quick = '--quick' in sys.argv
    with_cpu = '--no-cpu' not in sys.argv
    only = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else None

    def want(key):
        return only is None or only == key
    from oracle.cpu_oracle import Oracle            # CPU baseline leg only
    eng = Engine(0)

    def emit(rec):
        print(json.dumps(rec), flush=True)

    def cpu_attract(net, space, first, sample, max_t):
        if not with_cpu:
            return None
        t0 = time.perf_counter()
        _, _, _, steps = Oracle(net, space).attract(first, sample, max_t, None, True, per_problem=False, n_threads=CORES)
        dt = time.perf_counter() - t0
        return {'value': steps * net.n_nodes / dt, 'unit': 'node-state-updates/s', 'attractors_per_s': sample / dt, 'cores': CORES,
                'kind': 'port', 'sample': 'first {} problems, CPU oracle (C, OpenMP), {:.1f} s'.format(sample, dt)}

    def attract(name, text, first, count, max_t, cpu_sample, paths=('default',), key=None):
        if not want(key or name.split()[0]):
            return
        cfg = parse_input_text(text, max_t, Mode.ATTRACT)
        net, space = compile_problem(cfg)
        eng.set_problem(net, space)
        cpu = cpu_attract(net, space, first, min(cpu_sample, count), None if max_t == float('inf') else max_t)
        if cpu:
            time.sleep(0.5)         # the oracle's OpenMP threads spin for a while after their last loop: not under the GPU timings
        for path in paths:
            os.environ.pop('BSX_CUBES', None)
            if path == 'plain enumeration (BSX_CUBES=0)':
                os.environ['BSX_CUBES'] = '0'
            if path == 'functional graph':
                eng.attract_fgraph(first, count, max_t)                             # first run allocates the arrays
                t0 = time.perf_counter()
                r = eng.attract_fgraph(first, count, max_t, cap=1 << 20)
                dt = time.perf_counter() - t0
                merged, none, st = merge_tables([r.table]), r.n_no_attractor, r.stats
                n_states = 1 << net.n_nodes
                # bytes the passes stream through HBM (a random gather counted at its 4 or 8 useful bytes):
                # succ 4N (+ warm map 4N); doubling round = read 4N + gather 4N + write 4N; mark 4N; pair init 4N + 8N;
                # jump round = read 8N + gather 8N; aggregate 8N (+ 4N warm map)
                warm = 1 if space.sched is not None and len(space.sched) else 0
                cap_rel = None if max_t == float('inf') else max_t - (int(space.sched[:, 0].max()) if warm else 0)
                doubling = net.n_nodes if cap_rel is None else min(net.n_nodes, max(cap_rel, 1).bit_length())
                jumps = st['kernel_launches'] - doubling - 6 - warm
                fg_bytes = n_states * (4 + 4 * warm + 12 * doubling + 4 + 12 + 16 * jumps + 8 + 4 * warm)
                roof = roofline(fg_bytes, st['kernel_ms'], note='array bytes the {} passes stream through HBM ({} doubling + {} jump rounds), '
                                'a random gather counted at its 4 or 8 useful bytes'.format(st['kernel_launches'], doubling, jumps))
            else:
                run_attract_range(eng, first, count, max_t)     # warms the cycle cache and the scratch buffers
                t0 = time.perf_counter()
                merged, none, st = run_attract_range(eng, first, count, max_t)
                dt = time.perf_counter() - t0
                roof = roofline(st['executed_steps'] * net.n_nodes * 0.25, st['kernel_ms'],
                                note='0.25 B per executed node update (SURVEY 8d); states stay in registers/LDS',
                                pmc_key=(key or name.split()[0]) if path == 'default' else None)
            os.environ.pop('BSX_CUBES', None)
            emit({'config': name, 'path': path, 'mode': 'attract', 'n': net.n_nodes, 'problems': count, 'attractors': len(merged),
                  'no_attractor': none, 'wall_s': dt, 'kernel_ms': st['kernel_ms'], 'kernel_launches': st['kernel_launches'],
                  'attractors_per_s': count / dt,
                  'executed_node_updates_per_s': st['executed_steps'] * net.n_nodes / dt,
                  'reference_equivalent_node_updates_per_s': st['state_steps'] * net.n_nodes / dt,
                  'roofline': roof, 'cpu_baseline': cpu})

    example2 = open(os.path.join(ROOT, 'tests', 'golden', 'examples', 'output3_example2', 'example2.yaml')).read()
    attract('config1 examples/example2.yaml (3 nodes, 8 problems: plumbing)', example2, 0, 8, float('inf'), 8)
    cambium2 = open(os.path.join(ROOT, 'tests', 'golden', 'cambium2.yaml')).read()
    attract('config2 cambium2 full sweep, 2^30 problems', cambium2, 0, 1 << 30, float('inf'), 1 << 24,
            paths=('default', 'functional graph'))
    attract('config3 synthetic n=32 K=2, all 2^32 problems', synth.config3_yaml(), 0, 1 << (28 if quick else 32), 4096, 1 << 25,
            paths=('default', 'plain enumeration (BSX_CUBES=0)') + (() if quick else ('functional graph',)))
    attract('north-star n=64 K=2, 2^40 problems', synth.north_star_yaml(), 0x0123456789ABCDEF & ~((1 << 40) - 1),
            1 << (34 if quick else 40), 4096, 1 << 25, key='northstar40')
    # a network outside the ordered regime: K = 3, p = 1/2 is chaotic -- nothing collapses, cycles are longer than the cache
    # takes, every trajectory runs the detector (k_attract<2,3,1>); most of them run into the -t cap
    # (seed 12 of tools/depth_survey.py's K = 3 rows: 5.7 s per 2^26 problems in round 2)
    attract('chaotic n=64 K=3 seed 12, 2^{} problems, -t 4096'.format(18 if quick else 22), synth.network_yaml(64, 3, 12), 0,
            1 << (18 if quick else 22), 4096, 1 << 14, key='chaotic')
    if not (want('config4') or want('config5')):
        eng.close()
        return

    if want('config4'):
        config4(eng, emit, quick, with_cpu)
    if want('config5'):
        config5(eng, emit, quick, with_cpu)
    eng.close()


def config4(eng, emit, quick, with_cpu):
    from oracle.cpu_oracle import Oracle
    # config 4: target, n = 64, 8 knock-out variants x 2^28 initial states; summary sink (count + histogram + first 1000 hits)
    cfg = parse_input_text(synth.config4_yaml(), 1024, Mode.TARGET)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    mask = code_to_words(sum(1 << n for n in cfg['target node set']), net.n_words)
    code = code_to_words(cfg['target substate code'], net.n_words)
    count = 1 << (26 if quick else 31)
    cpu = None
    if with_cpu:
        t0 = time.perf_counter()
        pp, steps = Oracle(net, space).target(0, 1 << 22, 1024, mask, code, n_threads=CORES)
        dtc = time.perf_counter() - t0
        cpu = {'value': steps * 64 / dtc, 'unit': 'node-state-updates/s', 'problems_per_s': (1 << 22) / dtc, 'cores': CORES,
               'kind': 'port', 'sample': 'first 2^22 problems, CPU oracle, {:.1f} s'.format(dtc)}
    eng.target_summary(0, count, 1024, mask, code, hist_bins=1026, cap=1000)        # identical warm call (also lets the CPU baseline's OpenMP threads go to sleep)
    t0 = time.perf_counter()
    n_hits, hist, first_hits, st = eng.target_summary(0, count, 1024, mask, code, hist_bins=1026, cap=1000)
    dt = time.perf_counter() - t0
    emit({'config': 'config4 synthetic n=64 target -t 1024, 8 knock-out variants x 2^28', 'path': 'summary sink', 'mode': 'target', 'n': 64,
          'problems': count, 'hits': int(n_hits), 'listed': len(first_hits), 'mean_first_hit_t': float((hist * np.arange(len(hist))).sum() / max(n_hits, 1)),
          'wall_s': dt, 'kernel_ms': st['kernel_ms'], 'kernel_launches': st['kernel_launches'],
          'executed_node_updates_per_s': st['executed_steps'] * 64 / dt, 'problems_per_s': count / dt,
          'roofline': roofline(st['executed_steps'] * 64 * 0.25, st['kernel_ms'], note='0.25 B per executed node update (SURVEY 8d)',
                               pmc_key='config4'),
          'cpu_baseline': cpu})


def config5(eng, emit, quick, with_cpu):
    from oracle.cpu_oracle import Oracle
    # config 5: simulate -t 10000, n = 128, K = 3, perturbation schedule; final states + fold digests
    cfg = parse_input_text(synth.config5_yaml(), 10000, Mode.SIMULATE)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    cpu = None
    if with_cpu:
        t0 = time.perf_counter()
        Oracle(net, space).simulate(0, 1 << 13, 10000, want_traj=False, n_threads=CORES)
        dtc = time.perf_counter() - t0
        cpu = {'value': (1 << 13) * 10000 * 128 / dtc, 'unit': 'node-state-updates/s', 'cores': CORES, 'kind': 'port',
               'sample': 'first 2^13 problems x 10000 steps, CPU oracle, {:.1f} s'.format(dtc)}
    def sim_roofline(label, st):
        rate = st['executed_steps'] * 128 / (st['kernel_ms'] * 1e-3)
        if label.startswith('bit-sliced'):
            # The state matrix stays in LDS for all 10 000 steps (temporal blocking factor T = 10 000), so SURVEY 8(d)'s
            # streaming basis would give a figure above 1.  What bounds the kernel is VALU issue: one v_bfi per mux and
            # 32 trajectories, 2^K - 1 = 7 muxes per node update -> 7/32 lane-instructions per node update; the chip
            # issues 1024 SIMDs x 64 lanes x clock / 4 cycles per wave-instruction.
            peak = 1024 * 64 * 2.1e9 / 4 / (7 / 32)
            pmc = pmc_of('config5') or {}
            return {'bound': 'valu-issue', 'achieved': rate, 'peak': peak, 'unit': 'node-updates/s', 'frac': rate / peak,
                    'traffic': pmc.get('hbm_bytes_per_launch'), 'pmc_busy_fractions': pmc.get('issue_bound'),
                    'basis': 'VALU issue bound of the mux tree (7 v_bfi per 32 node updates at K = 3, 2.1 GHz); the state matrix never '
                             'leaves LDS (T = 10000), HBM traffic is the initial / final states only',
                    'survey_8d_streaming_equivalent_GBps': rate * 0.25 / 1e9}
        return roofline(st['executed_steps'] * 128 * 0.25, st['kernel_ms'], note='0.25 B per node update (SURVEY 8d)')

    count = 1 << (16 if quick else 20)
    for label, kw, env in (('bit-sliced kernel, final states + digests', dict(digest=True), None),
                           ('bit-sliced kernel, final states only', dict(digest=False), None),
                           ('per-lane kernel, final states + digests (BSX_SLICED=0)', dict(digest=True), '0')):
        if env is not None:
            os.environ['BSX_SLICED'] = env
        eng.simulate(0, min(count, 1 << 14), 10000, trajectories=False, **kw)       # first launch of the kernel: untimed
        t0 = time.perf_counter()
        _, fin, dig, st = eng.simulate(0, count, 10000, trajectories=False, **kw)
        dt = time.perf_counter() - t0
        os.environ.pop('BSX_SLICED', None)
        emit({'config': 'config5 synthetic n=128 K=3 simulate -t 10000, 2^{} of 2^26 problems'.format(count.bit_length() - 1), 'path': label,
              'mode': 'simulate', 'n': 128, 'problems': count, 'wall_s': dt, 'kernel_ms': st['kernel_ms'],
              'node_updates_per_s': st['state_steps'] * 128 / dt,
              'kernel_node_updates_per_s': st['state_steps'] * 128 / (st['kernel_ms'] * 1e-3),
              'seconds_for_2^26': dt * (1 << 26) / count,
              'roofline': sim_roofline(label, st),
              'cpu_baseline': cpu})
    if '--full-config5' in sys.argv:
        # the whole of config 5 on one GPU: 2^26 problems x 10000 steps, digests only, in slices of 2^22
        t0 = time.perf_counter()
        acc = np.uint64(0)
        kms = 0.0
        for first in range(0, 1 << 26, 1 << 22):
            _, _, dig, st = eng.simulate(first, 1 << 22, 10000, trajectories=False, final=False, digest=True)
            acc ^= np.bitwise_xor.reduce(dig)
            kms += st['kernel_ms']
        dt = time.perf_counter() - t0
        emit({'config': 'config5 FULL: 2^26 problems x 10000 steps, digests', 'mode': 'simulate', 'n': 128, 'problems': 1 << 26, 'wall_s': dt,
              'kernel_ms': kms, 'xor_of_all_digests': int(acc), 'node_updates_per_s': (1 << 26) * 10000 * 128 / dt,
              'roofline': sim_roofline('bit-sliced', {'executed_steps': (1 << 26) * 10000, 'kernel_ms': kms})})


if __name__ == '__main__':
    main()
