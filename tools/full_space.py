#!/usr/bin/env python
"""
The WHOLE north-star configuration, uncapped: `attract -t 4096` over all 2^64 initial states of the synthetic
n = 64, K = 2, seed 64 network (BASELINE.json caps the space to an index range because no stepping implementation
can enumerate it).  Round 2 needed 65 536 calls of 2^48 problems (31.5 s); with 128-bit counts and the levels of the
cube cascade chained on the device it is one bsx_run_attract2 call (two blocks of 2^63 problems).

    python tools/full_space.py [--log2-blocks B] > profiles/r03_full_space.json
    python -m torch.distributed.run --nproc-per-node 8 tools/full_space.py --log2-blocks 3     # one block per GPU

--log2-blocks B cuts the space into 2^B equal blocks (default 0: one call for everything); with several ranks
(RANK / WORLD_SIZE from the launcher) rank r takes blocks r, r + W, ... and the tables are merged with the job's one
all-gather (RCCL; BSX_DIST_BACKEND=socket for ranks that share a GPU).
Checks: every problem accounted for (sum of basin sizes + no-attractor count = 2^64), basins equal to round 2's.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import _lib, synth
from boolsi_amd.attract import merge_tables, table_from_merged
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.dist import Comm
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text

MAX_T = 4096
R02_BASINS = {6370653934217854976, 12076090139491696640}       # profiles/r02_full_space.json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--log2-blocks', type=int, default=0)
    args = ap.parse_args()
    comm = Comm.from_env()
    eng = Engine(int(os.environ.get('BSX_BENCH_DEVICE', comm.local_rank)))
    if comm.world > 1:
        comm.attach_engine(eng)
    cfg = parse_input_text(synth.north_star_yaml(), MAX_T, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    n_blocks = 1 << args.log2_blocks
    block = (1 << 64) // n_blocks
    eng.attract2(0, 1 << 30, MAX_T)                             # discovery, scratch buffers
    comm.barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    tables, none, stats = [], 0, {'state_steps': 0, 'executed_steps': 0, 'kernel_ms': 0.0, 'kernel_launches': 0, 'host_syncs': 0}
    calls = 0
    for b in range(comm.rank, n_blocks, comm.world):
        r = eng.attract2(b * block, block, MAX_T)
        tables.append(r.table)
        none += r.n_no_attractor
        for k in stats:
            stats[k] += r.stats[k]
        calls += 1
    merged = merge_tables(tables)
    if comm.world > 1:
        merged = merge_tables(comm.allgather_records(table_from_merged(merged, _lib.ATTR_REC2)))
    eng.synchronize()
    comm.barrier()
    dt = comm.allreduce_max(time.perf_counter() - t0)
    none, ref, execd, launches, syncs, calls = comm.allreduce_sum_int(
        [none, stats['state_steps'], stats['executed_steps'], stats['kernel_launches'], stats['host_syncs'], calls])
    kernel_s = comm.allreduce_max(stats['kernel_ms'] / 1e3)
    if comm.rank == 0:
        problems = 1 << 64
        total = sum(e[1] for e in merged.values()) + none
        assert total == problems, (total, problems)
        assert {e[1] for e in merged.values()} == R02_BASINS, 'basins differ from round 2\'s 65 536-call sweep'
        out = {
            'what': 'attract -t 4096 over ALL 2^64 initial states of the north-star network (n = 64, K = 2, seed 64): {} '
                    'bsx_run_attract2 call(s) of 2^{} problems on {} GPU(s)'.format(calls, 64 - args.log2_blocks, comm.world),
            'problems': problems, 'n_gpus': comm.world, 'calls': calls, 'wall_s': dt, 'kernel_s_slowest_rank': kernel_s,
            'kernel_launches': launches, 'host_syncs': syncs, 'attractors_per_s': problems / dt,
            'executed_updates': execd, 'reference_equivalent_updates': ref, 'no_attractor': none,
            'round_2': '65 536 calls, 31.5 s wall, 22.8 s kernels, 262 144 launches (profiles/r02_full_space.json): same basins',
            'attractors': [{'key_hex': '{:016x}'.format(k), 'length': e[0], 'basin': e[1], 'basin_share': e[1] / problems,
                            'mean_trajectory_l': e[2] / e[1]} for k, e in sorted(merged.items(), key=lambda kv: -kv[1][1])],
        }
        print(json.dumps(out, indent=1))
    comm.barrier()
    comm.shutdown()
    eng.close()


if __name__ == '__main__':
    main()
