#!/usr/bin/env python
"""
Headline benchmark (BASELINE.json): node-state-updates/s + attractors/s of an `attract` sweep over
a synthetic 64-node network (K = 2, seed 64, every node 'any'; the 2^64 space is capped to an
index range), at 1/2/4/8 GPUs.

One "step" = one pass of the hot path (bsx_run_attract: enumerate -> step -> detect -> aggregate)
over one batch of 2^LOG2_BATCH consecutive problem indices per GPU, followed by the merge of the
per-rank attractor tables (all-gather over RCCL when N > 1).  Weak scaling: every rank gets its own
batch each step.  Network tables live in HBM before the timed region; initial states are generated
on the device from the index, so nothing crosses PCIe inside a step except the attractor table.

`value` counts the updates the REFERENCE algorithm performs for the same problems
(n x sum over problems of its stop time, model.py:201), not the extra steps Brent's detector and the
mu pass execute on the device (reported as `executed_node_updates_per_s`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2-batch B]
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_NODE_UPDATE = 0.25    # read + write of the n-bit state per step = n/4 B
MAX_T = 4096


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--log2-batch', type=int, default=28, help='log2 of problems per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-log2-sample', type=int, default=25, help='log2 of the problems the CPU baseline runs (2^25: ~12 s on the box)')
    args = ap.parse_args()

    from boolsi_amd import synth, _lib
    from boolsi_amd.attract import merge_tables, table_from_merged
    from boolsi_amd.compile import compile_problem
    from boolsi_amd.constants import Mode
    from boolsi_amd.dist import Comm
    from boolsi_amd.input import parse_input_text

    comm = Comm.from_env()          # imports torch only when WORLD_SIZE > 1
    if comm.world != max(args.gpus, 1):
        raise SystemExit('--gpus {} but WORLD_SIZE is {}: launch N > 1 through torch.distributed.run'.format(
            args.gpus, comm.world))
    from boolsi_amd.engine import Engine
    # BSX_BENCH_DEVICE: rehearse several ranks on one GPU (with BSX_DIST_BACKEND=gloo); normally rank = GPU
    eng = Engine(int(os.environ.get('BSX_BENCH_DEVICE', comm.local_rank)))

    cfg = parse_input_text(synth.north_star_yaml(), MAX_T, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    n = net.n_nodes
    batch = 1 << args.log2_batch
    base = 0x0123456789ABCDEF & ~(batch - 1)     # somewhere inside the 2^64 space, batch aligned

    def step(s):
        first = base + (s * comm.world + comm.rank) * batch
        r = eng.attract(first, batch, MAX_T)
        merged = merge_tables(comm.allgather_records(table_from_merged(merge_tables([r.table]), _lib.ATTR_REC))) \
            if comm.world > 1 else merge_tables([r.table])
        assert sum(e[1] for e in merge_tables([r.table]).values()) + r.n_no_attractor == batch
        return r, merged

    for s in range(args.warmup):
        step(s)

    comm.barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    steps_ref = steps_exec = 0
    kernel_ms = 0.0
    launches = 0
    n_attractors = 0
    for s in range(args.warmup, args.warmup + args.steps):
        r, merged = step(s)
        steps_ref += r.stats['state_steps']
        steps_exec += r.stats['executed_steps']
        kernel_ms += r.stats['kernel_ms']
        launches += r.stats['kernel_launches']
        n_attractors = max(n_attractors, len(merged))
    eng.synchronize()
    comm.barrier()
    elapsed = comm.allreduce_max(time.perf_counter() - t0)
    tot_ref, tot_exec = comm.allreduce_sum_int([steps_ref, steps_exec])

    if comm.rank == 0:
        value = tot_ref * n / elapsed
        problems = batch * args.steps * comm.world
        # roofline of the dominant kernel (k_attract_pool), rank 0: algorithmic bytes per launch / avg duration
        avg_launch_s = kernel_ms / 1e3 / launches
        alg_bytes_per_launch = steps_ref / launches * n * BYTES_PER_NODE_UPDATE
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                t = json.load(f)
            if t.get('log2_batch') == args.log2_batch:
                traffic = t.get('hbm_bytes_per_launch')
        out = {
            'metric': 'node-state-updates/s',
            'value': value,
            'unit': 'node-state-updates/s',
            'n_gpus': comm.world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'u32',
            'data': 'synthetic',
            'config': {'workload': 'north-star attract sweep: synthetic n=64 K=2 seed=64, all nodes any, '
                                   '-t 4096, 2^{} consecutive problem indices per GPU per step'.format(args.log2_batch),
                       'n_nodes': n, 'problems_per_gpu_per_step': batch, 'max_t': MAX_T,
                       'partition': 'range x{}'.format(comm.world)},
            'attractors_per_s': problems / elapsed,
            'executed_node_updates_per_s': tot_exec * n / elapsed,
            'state_steps_per_problem': tot_ref / problems,
            'n_attractors': n_attractors,
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'k_attract_pool<NW=2,K=2,LDS>', 'avg_launch_ms': avg_launch_s * 1e3,
                         'alg_bytes_per_launch': alg_bytes_per_launch},
        }
        if comm.world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_oracle import Oracle        # timed CPU baseline only (kind "port")
            cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
            sample = 1 << args.cpu_log2_sample
            orc = Oracle(net, space)
            t1 = time.perf_counter()
            _, _, _, csteps = orc.attract(base, sample, MAX_T, None, True, per_problem=False, n_threads=cores)
            dt = time.perf_counter() - t1
            out['cpu_baseline'] = {'value': csteps * n / dt, 'unit': 'node-state-updates/s', 'cores': cores,
                                   'kind': 'port',
                                   'sample': 'first 2^{} problems of the same index range, CPU oracle '
                                             '(C, OpenMP), {:.1f} s'.format(args.cpu_log2_sample, dt)}
        print(json.dumps(out))
    comm.barrier()
    eng.close()
    comm.shutdown()


if __name__ == '__main__':
    main()
