"""
Command line: `python -m boolsi_amd simulate|attract|target FILE [options]`, same commands and
option names as the reference's `boolsi` script (`boolsi/cli.py:35-63, 184-328`).

Differences, all outside the hot path: results are kept in memory instead of a ZODB spill
(`-d`, `-k` accepted and ignored); `-b` does not schedule anything (the GPU dequeues its own chunks),
it only shapes the listing order under `--reference-np`; graphic outputs are not produced (`--no-pdf`
is implied, `--print-*` warn).  Started once per GPU by a launcher that sets RANK / WORLD_SIZE / MASTER_*
(the elastic launcher the benchmark driver uses does), every rank drives one GPU and all three commands range-partition
the problems (boolsi_amd/dist.py); rank 0 owns the output directory: it creates it, copies the input there
and writes the results, and its path is the one every rank uses.
`--reference-np P` lists the simulations of simulate / target in the order a reference run under
`mpiexec -np P` (P - 1 workers, `-b` batches each) writes them (batching.BatchLayout); the default is
the single-process order = problem-index order.
Error policy as the reference (cli.py:138-151,181): input / engine errors are logged and the
process terminates normally.  In a multi-rank job a rank that fails closes its control connection, which
ends its peers' next collective with PeerLost instead of leaving them blocked; those ranks exit with status 1.
"""
import logging
import os
import sys
from datetime import datetime
from math import inf

import click

from .constants import Mode
from .input import process_input, InputValidationException
from .log import configure_logging

_timestamp = datetime.now().strftime('%Y%m%dT%H%M%S.%f')[:-3]


def _common(f):
    options = [
        click.argument('input_file', type=click.Path(exists=True)),
        click.option('-o', '--output-directory', type=click.Path(file_okay=False), default='output_' + _timestamp,
                     help='Directory to print output to. Defaults to "<current directory>/output_<timestamp>".'),
        click.option('-b', '--batches-per-process', type=click.IntRange(min=1, max=2 ** 31), default=100,
                     help='Batches per process of the reference run whose listing order --reference-np reproduces; '
                          'the GPU engine schedules its own chunks.'),
        click.option('--reference-np', type=click.IntRange(min=1), default=1,
                     help='List simulations in the order of a reference run under "mpiexec -np N" (default 1).'),
        click.option('-d', '--tmp-database-directory', 'db_dir', type=click.Path(file_okay=False), default='tmp_db',
                     help='Accepted for compatibility; results are kept in memory.'),
        click.option('--no-pdf', is_flag=True, help='PDF output is not available in this build (always off).'),
        click.option('--pdf-page-limit', type=click.IntRange(min=1), default=500, help='Ignored.'),
        click.option('--no-csv', is_flag=True, help='Disable CSV output. CSV output is enabled by default.'),
        click.option('--print-png', is_flag=True, help='Not available in this build.'),
        click.option('--png-dpi', type=click.IntRange(min=1), default=300, help='Ignored.'),
        click.option('--print-tiff', is_flag=True, help='Not available in this build.'),
        click.option('--tiff-dpi', type=click.IntRange(min=1), default=150, help='Ignored.'),
        click.option('--print-svg', is_flag=True, help='Not available in this build.'),
        click.option('--device', type=int, default=None, help='GPU index (default: LOCAL_RANK or 0).'),
    ]
    for option in reversed(options):
        f = option(f)
    return f


class _Run:
    """Shared bootstrap / teardown of the three commands (reference cli.py:85-181)."""

    def __init__(self, kw):
        from .dist import Comm
        self.kw = kw
        self.failed = False
        self.comm = Comm.from_env()
        # rank 0's directory is THE directory (the default name carries a per-process timestamp); it exists
        # before any other rank goes on
        self.out = self.comm.broadcast_obj(kw['output_directory'])
        if self.comm.rank == 0:
            configure_logging(self.out)
            log = logging.getLogger()
            log.info('Hi! All BoolSi output (including this log) will appear in "{}".'.format(
                os.path.join(os.path.abspath(self.out), '')))
            log.info('Run parameters: "{}".'.format(' '.join(sys.argv[1:])))
            if kw['print_png'] or kw['print_tiff'] or kw['print_svg']:
                log.warning('Graphic output (PDF/PNG/TIFF/SVG) is not available in this build; CSV only.')
        else:
            logging.getLogger().setLevel(logging.CRITICAL)
        self.comm.barrier()
        self.engine = None

    def read_input(self, max_t, mode):
        return process_input(self.kw['input_file'], self.out, max_t, mode, copy_input=self.comm.rank == 0)

    def open_engine(self):
        from .engine import Engine
        device = self.kw['device'] if self.kw['device'] is not None else self.comm.local_rank
        self.engine = Engine(device)
        self.comm.attach_engine(self.engine)        # RCCL communicator on this engine's device
        if self.comm.rank == 0:
            logging.getLogger().info('Using GPU {}: {}.'.format(device, self.engine.device_info()['name']))
        return self.engine

    def finish(self):
        log = logging.getLogger()
        log.info('Terminating...')
        self.comm.shutdown()
        if self.engine is not None:
            self.engine.close()
        log.info('All BoolSi output is located in "{}". Bye!'.format(os.path.join(os.path.abspath(self.out), '')))
        if self.failed and self.comm.world > 1:
            sys.exit(1)


def _guarded(kw, body):
    run = _Run(kw)
    if kw['no_csv']:
        logging.getLogger().warning('Cannot proceed, all output formats are disabled.')
    else:
        try:
            body(run)
        except InputValidationException:
            pass
        except KeyboardInterrupt:
            logging.getLogger().error('Interrupted by user.')
        except Exception as e:   # noqa: BLE001  (reference policy: log and terminate normally)
            run.failed = True
            run.comm.abort()     # peers blocked on this rank see the closed connection
            logging.getLogger().exception('Exception caught: {}. See stacktrace below.'.format(e))
    run.finish()


def _listing(kw, cfg):
    """Listing order of a multi-process reference run, or None for index order."""
    if kw['reference_np'] <= 2:      # no workers, or one worker: one chunk, stride 1
        return None
    from .batching import batch_layout
    return batch_layout(cfg['total combination count'], kw['reference_np'], kw['batches_per_process'])


@click.group()
def cli():
    """BoolSi-compatible simulations of synchronous Boolean networks on AMD MI355X GPUs."""


@cli.command(help='Simulate for a number of time steps.')
@_common
@click.option('-t', '--simulation-time', type=click.IntRange(min=1), required=True,
              help='(required) Number of time steps to simulate for.')
def simulate(**kw):
    def body(run):
        from .simulate import simulate_master
        from .output import output_simulations
        cfg = run.read_input(kw['simulation_time'], Mode.SIMULATE)
        sims = simulate_master(run.open_engine(), cfg['origin simulation problem'],
                               cfg['simulation problem variations'], cfg['incoming node lists'],
                               cfg['truth tables'], kw['simulation_time'], cfg['total combination count'],
                               comm=run.comm, listing=_listing(kw, cfg))
        if run.comm.rank == 0:
            output_simulations(sims, cfg['node names'], run.out)
    _guarded(kw, body)


@cli.command(help='Find and analyze attractors for correlations between the nodes.')
@_common
@click.option('-t', '--max-simulation-time', type=click.IntRange(min=1),
              help='Maximum simulation time. If set, simulation stops after this time step even if '
                   'attractor was not found.')
@click.option('-a', '--max-attractor-length', type=click.IntRange(min=1),
              help='Maximum length of attractor to look for. If set, attractors longer than this value are discarded.')
@click.option('-r', '--reduce-memory-usage', 'reduce_memory_usage', is_flag=True,
              help='Accepted for compatibility: the GPU detector always runs in O(1) memory.')
@click.option('-k', '--keep-stale-db-items', 'keep_stale_db_items', is_flag=True, help='Accepted for compatibility.')
@click.option('-c', '--no-node-correlations', 'no_node_correlations', is_flag=True,
              help="Turn off computing Spearman's correlations between node states in attractors.")
@click.option('-x', '--no-attractor-output', 'no_attractor_output', is_flag=True,
              help='Turn off outputting attractors.')
@click.option('-p', '--p-value', 'p_value', type=click.FLOAT, default=0.05,
              help='p-value threshold for statistical significance of node correlations. Defaults to 0.05.')
def attract(**kw):
    def body(run):
        from .attract import attract_master
        from .attractor_analysis import find_node_correlations
        from .output import output_attractors, output_node_correlations
        max_t = kw['max_simulation_time'] or inf
        max_len = kw['max_attractor_length'] or inf
        cfg = run.read_input(max_t, Mode.ATTRACT)
        if kw['no_node_correlations'] and kw['no_attractor_output']:
            logging.getLogger().info("Cannot proceed, both attractors' and node correlations' output is disabled.")
            return
        attractors, n_none, total_frequency, stats = attract_master(
            run.open_engine(), cfg['origin simulation problem'], cfg['simulation problem variations'],
            cfg['incoming node lists'], cfg['truth tables'], max_t, max_len,
            cfg['total combination count'], comm=run.comm)
        if run.comm.rank != 0:
            return
        logging.getLogger().info(
            'Engine: {:.3e} node-state-updates ({} state-steps) in {:.1f} ms of kernels.'.format(
                stats['state_steps'] * len(cfg['node names']), stats['state_steps'], stats['kernel_ms']))
        if not attractors:
            return
        if not kw['no_node_correlations']:
            correlations = find_node_correlations(attractors)
            if correlations:
                output_node_correlations(correlations[0], correlations[1], kw['p_value'], cfg['node names'], run.out)
        if not kw['no_attractor_output']:
            _, fixed_nodes, _ = cfg['origin simulation problem']
            output_attractors(attractors, total_frequency, fixed_nodes, cfg['node names'],
                              cfg['total combination count'], max_len, max_t, run.out)
    _guarded(kw, body)


@cli.command(help='Find conditions leading to specific states of the network.')
@_common
@click.option('-t', '--max-simulation-time', type=click.IntRange(min=1),
              help='Maximum simulation time. If set, simulation stops after this time step even if a target '
                   'state was not reached.')
@click.option('-n', '--n-simulations-reaching-target', 'n_simulations_reaching_target', type=click.IntRange(min=1),
              help='Stop after this many simulations have reached target state.')
def target(**kw):
    def body(run):
        from .target import target_master
        from .output import output_simulations
        max_t = kw['max_simulation_time'] or inf
        n_to_find = kw['n_simulations_reaching_target'] or inf
        cfg = run.read_input(max_t, Mode.TARGET)
        sims = target_master(run.open_engine(), cfg['origin simulation problem'],
                             cfg['simulation problem variations'], cfg['target substate code'],
                             cfg['target node set'], cfg['incoming node lists'], cfg['truth tables'],
                             n_to_find, max_t, cfg['total combination count'],
                             comm=run.comm, listing=_listing(kw, cfg))
        if sims and run.comm.rank == 0:
            output_simulations(sims, cfg['node names'], run.out)
    _guarded(kw, body)


def main():
    cli()


if __name__ == '__main__':
    main()
