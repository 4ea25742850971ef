#!/usr/bin/env python
"""
Headline benchmark (BASELINE.json): node-state-updates/s + attractors/s of an `attract` sweep over
a synthetic 64-node network (K = 2, seed 64, every node 'any'; the 2^64 space is capped to an
index range), at 1/2/4/8 GPUs.

One "step" = one pass of the hot path (bsx_run_attract: enumerate -> step -> detect -> aggregate)
over one batch of 2^LOG2_BATCH consecutive problem indices per GPU (default 2^48, the most one call takes),
folded into the rank's running attractor table.  The engine runs such a batch as a cascade of cube passes
(DESIGN.md "Deeper collapse"): a few launches of k_attract_pool per step, the first one dominant.  Weak scaling: every rank gets its own batch each step.  After the LAST step the
per-rank tables are merged with one RCCL all-gather (inside the timed region; N > 1 only).  Network
tables live in HBM before the timed region; initial states are generated on the device from the
index, so nothing crosses PCIe inside a step except the (< 1 MB) attractor log.

Accounting (what each number counts):
  value / executed_node_updates_per_s   network updates the kernels really EXECUTED x n nodes / s.
                                        The cycle-state cache and class pooling execute far fewer
                                        updates than the reference algorithm performs for the same
                                        problems; that saving is NOT counted as work.
  reference_equivalent_node_updates_per_s   n x (sum of the reference loop's stop times, model.py:201) / s:
                                        what a stepping implementation would have had to do.
  attractors_per_s                      problems resolved per second (BASELINE.json's second metric).
  roofline                              SURVEY 8(d) basis: 0.25 B per EXECUTED node update, per launch of the
                                        dominant kernel (averaged over its launches, all levels of the cascade)
                                        / its HIP-event duration, vs 8 TB/s.  <= 1 by
                                        construction.  The kernel keeps states in registers/LDS, so its real
                                        HBM traffic (`traffic`, from a separate PMC pass) is ~1e-4 of that and
                                        HBM is not what limits it: `issue_bound` gives what the PMC counters show
                                        (VALU, SALU and LDS busy fractions), from the profile named in its `source`.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2-batch B]
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU); this program itself
uses no torch: RANK / WORLD_SIZE / MASTER_* are read by boolsi_amd.dist (TCP bootstrap + RCCL via the C-ABI).
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
PMC_FILE = os.path.join('profiles', 'r02_pmc.json')     # written by tools/pmc_read.py from the --pmc passes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--log2-batch', type=int, default=48, help='log2 of problems per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-log2-sample', type=int, default=25, help='log2 of the problems the CPU baseline runs (2^25: ~12 s on the box)')
    args = ap.parse_args()

    from boolsi_amd import synth, _lib
    from boolsi_amd.attract import merge_tables, table_from_merged
    from boolsi_amd.compile import compile_problem
    from boolsi_amd.constants import Mode
    from boolsi_amd.dist import Comm
    from boolsi_amd.input import parse_input_text

    comm = Comm.from_env()
    if comm.world != max(args.gpus, 1):
        raise SystemExit('--gpus {} but WORLD_SIZE is {}: launch N > 1 with one process per GPU '
                         '(python -m torch.distributed.run --nproc-per-node N ...)'.format(args.gpus, comm.world))
    from boolsi_amd.engine import Engine
    # BSX_BENCH_DEVICE: rehearse several ranks on one GPU (with BSX_DIST_BACKEND=socket); normally rank = GPU
    eng = Engine(int(os.environ.get('BSX_BENCH_DEVICE', comm.local_rank)))
    # RCCL communicator on the engine's device (N > 1).  This is a measurement harness: if RCCL cannot be set up on
    # this node the final merge goes over the TCP control plane instead, and the line says so (`config.merge`).
    rccl_problem = None

    def agree_on_data_plane(err):
        nonlocal rccl_problem
        errs = [e for e in comm.allgather_obj(err) if e]
        if errs and comm.backend == 'rccl':
            comm.backend = 'socket'
            rccl_problem = errs[0]

    if comm.world > 1:
        err = None
        try:
            comm.attach_engine(eng)
        except Exception as e:      # noqa: BLE001
            err = '{}: {}'.format(type(e).__name__, str(e)[:200])
        agree_on_data_plane(err)

    cfg = parse_input_text(synth.north_star_yaml(), MAX_T, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    n = net.n_nodes
    batch = 1 << args.log2_batch
    base = 0x0123456789ABCDEF & ~(batch - 1)     # somewhere inside the 2^64 space, batch aligned

    def step(s, running):
        first = base + (s * comm.world + comm.rank) * batch
        r = eng.attract(first, batch, MAX_T)
        assert int(r.table['count'].sum()) + r.n_no_attractor == batch      # every problem accounted for
        running.append(r.table)
        return r

    warm = []
    for s in range(args.warmup):
        step(s, warm)
    if comm.world > 1 and warm:
        # the collective's first call sets up its connections: part of the warm-up, like the first launches
        err = None
        try:
            comm.allgather_records(table_from_merged(merge_tables(warm), _lib.ATTR_REC))
        except Exception as e:      # noqa: BLE001
            err = '{}: {}'.format(type(e).__name__, str(e)[:200])
        agree_on_data_plane(err)

    comm.barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    steps_ref = steps_exec = 0
    kernel_ms = 0.0
    launches = 0
    tables = []
    for s in range(args.warmup, args.warmup + args.steps):
        r = step(s, tables)
        steps_ref += r.stats['state_steps']
        steps_exec += r.stats['executed_steps']
        kernel_ms += r.stats['kernel_ms']
        launches += r.stats['kernel_launches']
    mine = merge_tables(tables)
    # the job's one data collective: per-rank tables -> every rank (RCCL all-gather over xGMI)
    merged = merge_tables(comm.allgather_records(table_from_merged(mine, _lib.ATTR_REC))) if comm.world > 1 else mine
    eng.synchronize()
    comm.barrier()
    elapsed = comm.allreduce_max(time.perf_counter() - t0)
    tot_ref, tot_exec = comm.allreduce_sum_int([steps_ref, steps_exec])

    if comm.rank == 0:
        problems = batch * args.steps * comm.world
        assert sum(e[1] for e in merged.values()) <= problems
        # roofline of the dominant kernel (k_attract_pool), rank 0: executed algorithmic bytes per launch / avg duration
        avg_launch_s = kernel_ms / 1e3 / launches
        alg_bytes_per_launch = steps_exec / launches * n * BYTES_PER_NODE_UPDATE
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9
        traffic = traffic_source = None
        issue = None
        ppath = os.path.join(ROOT, PMC_FILE)
        if os.path.exists(ppath):
            with open(ppath) as f:
                pmc = json.load(f)
            if pmc.get('log2_batch') == args.log2_batch:
                traffic = pmc.get('hbm_bytes_per_launch')
                traffic_source = PMC_FILE + ' (separate rocprofv3 --pmc passes of this command, not measured in this run)'
                issue = dict(pmc.get('issue_bound') or {}, source=PMC_FILE)
        out = {
            'metric': 'node-state-updates/s',
            'value': tot_exec * n / elapsed,
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
                       'partition': 'range x{}'.format(comm.world),
                       'merge': ('one all-gather after the last step, data plane: ' + str(comm.backend) +
                                 (' (RCCL could not be used: ' + rccl_problem + ')' if rccl_problem else '')) if comm.world > 1 else 'none (1 GPU)'},
            'value_counts': 'executed network updates x n nodes (work skipped by the cycle cache / class pooling is not counted)',
            'attractors_per_s': problems / elapsed,
            'executed_node_updates_per_s': tot_exec * n / elapsed,
            'reference_equivalent_node_updates_per_s': tot_ref * n / elapsed,
            'executed_updates_per_problem': tot_exec / problems,
            'state_steps_per_problem': tot_ref / problems,
            'n_attractors': len(merged),
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': 'k_attract_pool<NW=2,K=2,LDS>', 'avg_launch_ms': avg_launch_s * 1e3,
                         'alg_bytes_per_launch': alg_bytes_per_launch,
                         'basis': '0.25 B per EXECUTED node update (SURVEY 8d) x executed updates per launch / HIP-event '
                                  'launch time; states stay in registers/LDS, so this is a normalised rate, not HBM utilisation',
                         'issue_bound': issue},
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
                                   'kind': 'port', 'attractors_per_s': sample / dt,
                                   'sample': 'first 2^{} problems of the same index range, CPU oracle '
                                             '(C, OpenMP; executes every update of the reference loop), {:.1f} s'.format(
                                                 args.cpu_log2_sample, dt)}
        print(json.dumps(out))
    comm.barrier()
    comm.shutdown()
    eng.close()


if __name__ == '__main__':
    main()
