"""
Target mode: simulations that reach the target substate (after all perturbations), reference
`boolsi/target.py` (target_master 13-88, simulate_until_target_substate_or_max_t 109-133).

The search runs on the GPU (bsx_run_target: first t >= T_p with state & mask == code, none if the
trajectory closes its cycle or max_t comes first); the states of the hits are then materialised
with bsx_run_trajectories.  `-n` keeps the first n hits in enumeration (= index) order, exactly
what the reference's single process finds before it stops (simulate.py:163-164, mpi.py:537-538):
index tiles are processed in order and the search stops after the tile that completes n.
"""
import logging
from math import inf

import numpy as np

from . import _lib
from .batching import create_numeral_system_from_variations, problem_from_index
from .compile import compile_network, compile_space, code_to_words
from .dist import Comm, partition
from .simulate import Simulation, states_from_words, MAX_STATE_CELLS

TILE = 1 << 24


def find_hits(engine, first, count, max_t, mask, code, n_to_find=inf):
    """(offset from `first`, t) of the hits in [first, first + count), index order, stopping after the
    tile that completes n_to_find.  Returns (hits array of _lib.HIT, problems scanned)."""
    found, scanned, n = [], 0, 0
    while scanned < count and n < n_to_find:
        tile = min(TILE, count - scanned)
        hits, _ = engine.target(first + scanned, tile, max_t, mask, code)
        if len(hits):
            hits = hits.copy()
            hits['offset'] += np.uint64(scanned)
            found.append(hits)
            n += len(hits)
        scanned += tile
    out = np.concatenate(found) if found else np.zeros(0, _lib.HIT)
    return out, scanned


def target_hits_partitioned(engine, max_t, mask, code, n_simulation_problems, comm=None):
    """Range-partitioned search over all problems (one rank per GPU); every rank gets all hits, in
    index order (rank order = index order).  Problem indices here must fit 64 bits."""
    comm = comm or Comm()
    first, count = partition(n_simulation_problems, comm.world, comm.rank)
    hits, _ = find_hits(engine, first, count, max_t, mask, code)
    hits = hits.copy()
    hits['offset'] += np.uint64(first)
    return comm.gather_concat(hits)


def target_master(engine, origin_simulation_problem, simulation_problem_variations,
                  target_substate_code, target_node_set, predecessor_node_lists, truth_tables,
                  n_simulations_to_reach_target_substate, max_t, n_simulation_problems, comm=None, listing=None):
    """
    -> list of Simulation (states s(0..t_hit)) on rank 0, [] elsewhere; log lines as target.py:45-86.
    Order: index order = single-process reference order; with `listing` (batching.BatchLayout) the
    order of a multi-process reference run.  `-n` keeps the first n in that order (a reference run with
    workers keeps whichever n arrive first -- not reproducible).
    Single rank, index order: tiles are searched in order and the search stops after the tile that
    completes n.  Several ranks or a listing: the whole space is searched (range-partitioned), hits
    all-gathered, ordered, cut to n, and materialised by rank 0.
    """
    log = logging.getLogger()
    comm = comm or Comm()
    n_to_find = n_simulations_to_reach_target_substate
    if n_to_find is not inf and n_to_find > n_simulation_problems:
        log.warning('Requested {} simulations that reach target state, but only {} simulation '
                    'problems provided. Will look for all simulations that reach target state.'.format(
                        n_to_find, n_simulation_problems))
        n_to_find = inf
    log.info('{} will be used to perform {} simulations and find {} that reach target states.'.format(
        'Single process' if comm.world == 1 else '{} GPUs'.format(comm.world),
        n_simulation_problems, 'all' if n_to_find is inf else n_to_find))

    n_nodes = len(predecessor_node_lists)
    net = compile_network(predecessor_node_lists, truth_tables)
    space = compile_space(origin_simulation_problem, simulation_problem_variations)
    engine.set_problem(net, space)
    mask = code_to_words(sum(1 << n for n in target_node_set), net.n_words)
    code = code_to_words(target_substate_code, net.n_words)

    scanned = n_simulation_problems
    if comm.world == 1 and listing is None:
        hits, scanned = find_hits(engine, 0, n_simulation_problems, max_t, mask, code, n_to_find)
    else:
        hits = target_hits_partitioned(engine, max_t, mask, code, n_simulation_problems, comm)
        if listing is not None and len(hits):
            keys = [listing.position(int(i)) for i in hits['offset']]
            hits = hits[sorted(range(len(hits)), key=keys.__getitem__)]
    if comm.rank != 0:
        return []
    if n_to_find is not inf:
        hits = hits[:int(n_to_find)]
    cells = int((hits['t'].astype(np.float64) + 1).sum()) * n_nodes if len(hits) else 0
    if cells > MAX_STATE_CELLS:
        raise ValueError('{} simulations reach the target state; their {} node states do not fit in host '
                         'memory -- use -n, or boolsi_amd.target.find_hits for (problem, t) pairs only'.format(
                             len(hits), cells))

    numeral_system = create_numeral_system_from_variations(simulation_problem_variations)
    simulations = []
    for lo in range(0, len(hits), 1 << 16):
        part = hits[lo:lo + (1 << 16)]
        trajs, _ = engine.trajectories(0, part['offset'], part['t'])
        for h, traj in zip(part, trajs):
            _, fixed_nodes, perturbed_nodes_by_t = problem_from_index(
                int(h['offset']), origin_simulation_problem, simulation_problem_variations, numeral_system)
            simulations.append(Simulation(states_from_words(traj, n_nodes), fixed_nodes, perturbed_nodes_by_t))

    # (no "Goal reached at ..." line: the reference's single process never prints it, because
    #  write_simulations_to_db always reports 0 added simulations, simulate.py:160,178 / mpi.py:173-184)
    wanted = n_simulation_problems if n_to_find is inf else n_to_find
    if simulations:
        if len(simulations) < wanted:
            word = 'Only'
        elif wanted < n_simulation_problems:
            word = 'At least'
        else:
            word = 'All'
        text = '{} {} simulations'.format(word, len(simulations))
    else:
        text = 'No simulations'
    log.info('{} reach target state.'.format(text))
    return simulations
