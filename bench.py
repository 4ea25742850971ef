#!/usr/bin/env python
"""
Headline benchmark (BASELINE.json): node-state-updates/s + attractors/s of an `attract` sweep over
a synthetic 64-node network (K = 2, seed 64, every node 'any'; the 2^64 space is capped to an
index range), at 1/2/4/8 GPUs.

One "step" = one pass of the hot path (bsx_run_attract2: enumerate -> step -> detect -> aggregate) over one batch of
2^LOG2_BATCH consecutive problem indices per GPU (default 2^60: 1/16 of the whole space, so the default 16 steps are one
sweep's worth of problems), folded into the rank's running attractor table.  The engine runs such a batch as chains of
launches behind ONE wait (DESIGN.md "Deeper collapse" / "levels chained on the device" / "sub-blocks"): k_attract_pool per
level of each sub-block's cascade with a packing kernel in between, the top levels dominant; the host enqueues the chains,
waits once and adds up wide integers.  Weak scaling: every rank gets its own batch each step.
After the LAST step the per-rank tables are merged with one RCCL all-gather (inside the timed region; N > 1 only).
Network tables live in HBM before the timed region; initial states are generated on the device from the index, so
nothing crosses PCIe inside a step except a few KB of counters.

Accounting (what each number counts):
  value / executed_node_updates_per_s   network updates the kernels really EXECUTED x n nodes / s.
                                        The cycle-state cache and class pooling execute far fewer
                                        updates than the reference algorithm performs for the same
                                        problems; that saving is NOT counted as work.
  reference_equivalent_node_updates_per_s   n x (sum of the reference loop's stop times, model.py:201) / s:
                                        what a stepping implementation would have had to do.
  attractors_per_s                      problems resolved per second (BASELINE.json's second metric).
  roofline                              what bounds the dominant kernel -- whichever of the pool kernel's two builds took more device
                                        time in this run, the top level of the cascades (<..,true,false>) or their lower levels
                                        (<..,true,true>); the other one is under `other_build`:
                                        `bound` names the busiest unit of the PMC profile of THIS build (profiles/r03_pmc.json,
                                        r03_pmc_lower.json: VALU issue / LDS pipeline / SALU issue), `achieved` = that unit's busy
                                        cycles per second = its per-executed-update cost from the profile x the updates these
                                        launches executed in this run / their duration (measured live: HIP events around the top
                                        levels, the launches' own device clock for the lower levels), `peak` = one busy cycle per
                                        cycle of the 2.4 GHz maximum clock per SIMD (VALU, SALU) or per CU (LDS).
                                        `device_level`: the lower levels run on side streams next to the next chains' top levels,
                                        so the two builds' launch durations overlap; this adds both builds' busy cycles over the
                                        steps' kernel time (BSX_CUBE_STREAMS=1 serialises: profiles/r03_bench_serial.json).
                                        `hbm_normalised`: SURVEY 8(d)'s figure, 0.25 B per EXECUTED node update / launch time
                                        vs 8 TB/s -- a normalised rate, not HBM utilisation: states stay in registers / LDS,
                                        the measured HBM traffic (`traffic`) is orders of magnitude below it.

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
MAX_CLOCK_HZ = 2.4e9            # ... maximum shader clock
BYTES_PER_NODE_UPDATE = 0.25    # read + write of the n-bit state per step = n/4 B
MAX_T = 4096
PMC_FILE = os.path.join('profiles', 'r03_pmc.json')     # written by tools/pmc_read.py from the --pmc passes: top-level build
PMC_FILE_LOWER = os.path.join('profiles', 'r03_pmc_lower.json')     # ... lower-level build


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--log2-batch', type=int, default=60, help='log2 of problems per GPU per step (<= 63)')
    ap.add_argument('--allow-socket-merge', action='store_true', help='N > 1: if RCCL cannot be set up, merge over the TCP control plane instead of failing')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dump-table', metavar='PATH', help='rank 0 writes the merged attractor table of the timed steps there (JSON; tests)')
    ap.add_argument('--cpu-log2-sample', type=int, default=25, help='log2 of the problems the CPU baseline runs (2^25: ~12 s on the box)')
    args = ap.parse_args()

    from boolsi_amd import synth, _lib
    from boolsi_amd.attract import merge_tables, record_ints, table_from_merged
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
    # RCCL communicator on the engine's device (N > 1).  A node whose RCCL cannot be set up fails the run (non-zero exit on
    # every rank) unless the socket data plane was asked for: BSX_DIST_BACKEND=socket, or --allow-socket-merge to downgrade
    # after a failed attempt -- the line then says so in `config.merge`.
    rccl_problem = None

    def agree_on_data_plane(err):
        nonlocal rccl_problem
        errs = [e for e in comm.allgather_obj(err) if e]
        if errs and comm.backend == 'rccl':
            if not args.allow_socket_merge:
                comm.abort()
                raise SystemExit('RCCL could not be set up ({}); rerun with --allow-socket-merge or BSX_DIST_BACKEND=socket '
                                 'to merge over TCP instead'.format(errs[0]))
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
    info = eng.network_info()
    batch = 1 << args.log2_batch
    base = 0x0123456789ABCDEF & ~(batch - 1)     # somewhere inside the 2^64 space, batch aligned

    def step(s, running):
        # (batch number s x world + rank, around the space: a run longer than 2^64 / batch steps x ranks meets batches again)
        first = (base + (s * comm.world + comm.rank) * batch) % (1 << 64)
        r = eng.attract2(first, batch, MAX_T)
        running.append(r)
        return r

    def checked(results):
        """every problem of every step accounted for (exact, wide); -> the step tables"""
        for r in results:
            assert sum(c[0] | c[1] << 64 for c in r.table['count'].tolist()) + r.n_no_attractor == batch
        return [r.table for r in results]

    warm = []
    for s in range(args.warmup):
        step(s, warm)
    if comm.world > 1 and warm:
        # the collective's first call sets up its connections: part of the warm-up, like the first launches
        err = None
        try:
            comm.allgather_records(table_from_merged(merge_tables(checked(warm)), _lib.ATTR_REC2))
        except Exception as e:      # noqa: BLE001
            err = '{}: {}'.format(type(e).__name__, str(e)[:200])
        agree_on_data_plane(err)

    comm.barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    steps_ref = steps_exec = dom_exec = dom_launches = launches = syncs = low_exec = low_launches = 0
    kernel_ms = dom_ms = low_ms = 0.0
    results = []
    for s in range(args.warmup, args.warmup + args.steps):
        st = step(s, results).stats
        steps_ref += st['state_steps']
        steps_exec += st['executed_steps']
        kernel_ms += st['kernel_ms']
        launches += st['kernel_launches']
        dom_ms += st['dominant_ms']
        dom_exec += st['dominant_executed_steps']
        dom_launches += st['dominant_launches']
        low_ms += st['lower_ms']
        low_exec += st['lower_executed_steps']
        low_launches += st['lower_launches']
        syncs += st['host_syncs']
    mine = merge_tables(checked(results))
    # the job's one data collective: per-rank tables -> every rank (RCCL all-gather over xGMI)
    merged = merge_tables(comm.allgather_records(table_from_merged(mine, _lib.ATTR_REC2))) if comm.world > 1 else mine
    eng.synchronize()
    comm.barrier()
    elapsed = comm.allreduce_max(time.perf_counter() - t0)
    tot_ref, tot_exec = comm.allreduce_sum_int([steps_ref, steps_exec])

    if comm.rank == 0:
        problems = batch * args.steps * comm.world
        assert sum(e[1] for e in merged.values()) <= problems
        # roofline of the two builds of the pool kernel a step launches, rank 0, measured live: the top level of every
        # cascade (HIP events around those launches) and the lower levels (the launches' own first-in / last-out device
        # clock), the updates they executed from their own counters.  `roofline` is the one that took more device time.
        busy_totals = []        # per build with a PMC profile: unit -> (busy unit-cycles of this run's launches, unit-cycles per second of the device)

        def roofline_of(kernel, ms, n_launches, executed, pmc_file, timing):
            avg_launch_s = ms / 1e3 / n_launches
            upd_per_launch = executed / n_launches
            alg_bytes_per_launch = upd_per_launch * n * BYTES_PER_NODE_UPDATE
            hbm_norm = alg_bytes_per_launch / avg_launch_s / 1e9
            roof = {'bound': 'hbm-normalised', 'achieved': hbm_norm, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm_norm / HBM_PEAK_GBS,
                    'traffic': None, 'note': 'no PMC profile of this build at ' + pmc_file + ': only the normalised figure'}
            ppath = os.path.join(ROOT, pmc_file)
            if os.path.exists(ppath):
                with open(ppath) as f:
                    pmc = json.load(f)
                per_upd = pmc.get('per_executed_update') or {}
                if pmc.get('log2_batch') == args.log2_batch and per_upd:
                    # busy cycles of a unit, summed over its instances on the device (1024 SIMDs issue VALU / SALU, 256 CUs have an LDS)
                    units = {'valu': ('valu_busy_cycles', 1024), 'salu': ('salu_busy_cycles', 1024), 'lds': ('lds_busy_cycles', 256)}
                    # peak = one busy cycle per unit per cycle of the 2.4 GHz maximum clock (MI355X_MICROARCH.md); the clock the chip
                    # held is not measured here, so the fraction is a lower bound of the busy share (the PMC files' *_busy_frac
                    # are against the cycles the launch really took)
                    clock_hz = MAX_CLOCK_HZ
                    rates = {}
                    for unit, (key, lanes) in units.items():
                        if key in per_upd:
                            rates[unit] = (per_upd[key] * upd_per_launch / avg_launch_s, lanes * clock_hz)
                    busy_totals.append({u: (per_upd[units[u][0]] * executed, units[u][1] * clock_hz) for u in rates})
                    if rates:
                        unit = max(rates, key=lambda u: rates[u][0] / rates[u][1])
                        ach, peak = rates[unit]
                        roof = {'bound': {'valu': 'valu-issue', 'salu': 'salu-issue', 'lds': 'lds-pipeline'}[unit],
                                'achieved': ach / 1e9, 'peak': peak / 1e9, 'unit': 'G busy unit-cycles/s',
                                'frac': ach / peak,
                                'traffic': pmc.get('hbm_bytes_per_launch'),
                                'traffic_source': pmc_file + ' (separate rocprofv3 --pmc passes of this command, not measured in this run)',
                                'all_units_frac': {u: r[0] / r[1] for u, r in rates.items()},
                                'basis': 'busy cycles of the unit per executed update (PMC passes of this build, ' + pmc_file + ') x updates these '
                                         'launches executed in THIS run / their duration; peak = instances of the unit x the 2.4 GHz maximum clock',
                                'pmc_busy_fractions': pmc.get('issue_bound'), 'source': pmc_file}
            roof.update({'kernel': kernel, 'avg_launch_ms': avg_launch_s * 1e3, 'launches_timed': n_launches, 'device_ms_per_step': ms / args.steps,
                         'timing': timing, 'executed_updates_per_launch': upd_per_launch,
                         'hbm_normalised': {'achieved': hbm_norm, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm_norm / HBM_PEAK_GBS,
                                            'alg_bytes_per_launch': alg_bytes_per_launch,
                                            'basis': '0.25 B per EXECUTED node update (SURVEY 8d) x executed updates per launch / '
                                                     'launch time: a normalised rate, states stay in registers / LDS'}})
            return roof

        shape = (info['state_words32'], info['mux_slots'], info['lut_mode'])
        roof_top = roofline_of('k_attract_pool<{},{},{},true,false>'.format(*shape), dom_ms, dom_launches, dom_exec, PMC_FILE,
                               'HIP events around every top-level launch')
        roof_low = None
        if low_launches:
            roof_low = roofline_of('k_attract_pool<{},{},{},true,true>'.format(*shape), low_ms, low_launches, low_exec, PMC_FILE_LOWER,
                                   "every launch's own first-workgroup-in to last-workgroup-out time on the device's constant clock "
                                   '(Counters::t_first_not / t_last); launches that found an empty list are not counted')
            roof_low['unit_of_work'] = ('network updates the lower levels executed; the depth-1 level evaluates up to 2^14 children of a listed '
                                        'class bit-sliced behind ONE update of the parent, so its work per update is not that of a per-child pass')
        if roof_low and low_ms > dom_ms:
            roof, other = roof_low, roof_top
        else:
            roof, other = roof_top, roof_low
        if other:
            roof['other_build'] = other
        if roof_low:
            # The lower levels run on side streams next to the following chains' top levels, so the per-launch durations of
            # the two builds overlap (they add up to more than the kernel time, and each launch shares the chip).  What the
            # device as a whole did: busy cycles of both builds (PMC figures per executed update x the updates of this
            # run) over the HIP-event kernel time of the steps.
            dev = {}
            for u in ('valu', 'salu', 'lds'):
                if busy_totals and all(u in b for b in busy_totals):
                    dev[u] = sum(b[u][0] for b in busy_totals) / (busy_totals[0][u][1] * kernel_ms / 1e3)
            roof['device_level'] = {'busy_fraction_over_kernel_time': dev, 'kernel_ms_per_step': kernel_ms / args.steps,
                                    'launch_ms_per_step_by_build': {'top': dom_ms / args.steps, 'lower': low_ms / args.steps},
                                    'note': 'the two builds overlap on side streams (BSX_CUBE_STREAMS=1 serialises them: '
                                            'profiles/r03_bench_serial.json); per-build launch durations include that sharing'}
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
                                   '-t 4096, 2^{} consecutive problem indices per GPU per step (one bsx_run_attract2 call)'.format(args.log2_batch),
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
            'kernel_ms_per_step': kernel_ms / args.steps,
            'host_ms_per_step': elapsed * 1e3 / args.steps - kernel_ms / args.steps,
            'kernel_launches_per_step': launches / args.steps,
            'host_syncs_per_step': syncs / args.steps,
            'roofline': roof,
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
        if args.dump_table:
            with open(args.dump_table, 'w') as f:
                json.dump({'{:x}'.format(k): [str(v) for v in e] for k, e in sorted(merged.items())}, f)
        print(json.dumps(out))
    comm.barrier()
    comm.shutdown()
    eng.close()


if __name__ == '__main__':
    main()
