"""
Simulate mode: every combination of initial state, fixed nodes and perturbations is simulated
for `max_t` steps (reference `boolsi/simulate.py`: Simulation 16-28, simulate_master 46-78,
simulate_until_max_t 97-131, store_simulation 134-146).

The trajectories are computed on the GPU (bsx_run_simulate, plain stepping == S11); this module
labels them with the problem's fixed nodes and perturbations (host-side enumeration) and keeps
them in memory in problem-index order, where the reference spills them to ZODB.
"""
import logging

import numpy as np

from .batching import create_numeral_system_from_variations, problem_from_index
from .compile import compile_network, compile_space, words_to_code
from .model import count_perturbations, decode_state

TILE = 1 << 14                    # problems per engine call
MAX_STATE_CELLS = 1 << 31         # refuse to materialise more than this many node states on the host


class Simulation:
    """One simulated trajectory with the fixed nodes / perturbations it ran under."""

    def __init__(self, states, fixed_nodes, perturbed_nodes_by_t):
        self.states = states
        self.fixed_nodes = fixed_nodes
        self.perturbed_nodes_by_t = perturbed_nodes_by_t
        self.n_perturbations = count_perturbations(perturbed_nodes_by_t)

    def __eq__(self, other):
        return (self.states, self.fixed_nodes, self.perturbed_nodes_by_t) == \
               (other.states, other.fixed_nodes, other.perturbed_nodes_by_t)


def states_from_words(traj, n_nodes):
    """(T + 1, W) uint64 array -> list of lists of bool."""
    return [decode_state(words_to_code(s), n_nodes) for s in traj]


def simulate_master(engine, origin_simulation_problem, simulation_problem_variations,
                    predecessor_node_lists, truth_tables, max_t, n_simulation_problems, comm=None, listing=None):
    """
    -> list of Simulation (on rank 0; other ranks get []).  Order: problem-index order, which is the
    order of a single-process reference run; with `listing` (batching.BatchLayout) the order in which a
    multi-process reference run with that layout lists them (batch number, place in batch).
    With more than one rank the index range is partitioned and the trajectories all-gathered.
    """
    from .dist import Comm, partition
    comm = comm or Comm()
    log = logging.getLogger()
    if comm.world == 1:
        log.info('Single process will be used to perform {} simulations...'.format(n_simulation_problems))
    else:
        log.info('{} GPUs will be used to perform {} simulations...'.format(comm.world, n_simulation_problems))
    n_nodes = len(predecessor_node_lists)
    if n_simulation_problems * (max_t + 1) * n_nodes > MAX_STATE_CELLS:
        raise ValueError(
            '{} simulations of {} steps do not fit in host memory as full trajectories; use the '
            'final-state / digest sinks of boolsi_amd.engine.Engine.simulate instead'.format(
                n_simulation_problems, max_t))
    net = compile_network(predecessor_node_lists, truth_tables)
    space = compile_space(origin_simulation_problem, simulation_problem_variations)
    engine.set_problem(net, space)
    lo, mine = partition(n_simulation_problems, comm.world, comm.rank)
    parts = []
    for first in range(lo, lo + mine, TILE):
        count = min(TILE, lo + mine - first)
        traj, _, _, _ = engine.simulate(first, count, max_t, trajectories=True, final=False, digest=False)
        parts.append(traj)
    local = np.concatenate(parts) if parts else np.zeros((0, max_t + 1, net.n_words), np.uint64)
    trajs = comm.gather_concat(local)
    if comm.rank != 0:
        return []
    numeral_system = create_numeral_system_from_variations(simulation_problem_variations)
    from .batching import reference_order
    order = reference_order(listing) if listing is not None else range(n_simulation_problems)
    simulations = []
    for index in order:
        _, fixed_nodes, perturbed_nodes_by_t = problem_from_index(
            index, origin_simulation_problem, simulation_problem_variations, numeral_system)
        simulations.append(Simulation(states_from_words(trajs[index], n_nodes), fixed_nodes, perturbed_nodes_by_t))
    return simulations


def simulate_digests(engine, origin_simulation_problem, simulation_problem_variations,
                     predecessor_node_lists, truth_tables, max_t, first, count):
    """Large runs: final states and FNV-1a digests of s(0..max_t) instead of trajectories."""
    net = compile_network(predecessor_node_lists, truth_tables)
    space = compile_space(origin_simulation_problem, simulation_problem_variations)
    engine.set_problem(net, space)
    finals, digests = [], []
    done = 0
    while done < count:
        tile = min(1 << 24, count - done)
        _, fin, dig, _ = engine.simulate(first + done, tile, max_t, trajectories=False)
        finals.append(fin)
        digests.append(dig)
        done += tile
    return np.concatenate(finals), np.concatenate(digests)


def simulate_digests_partitioned(engine, origin_simulation_problem, simulation_problem_variations,
                                 predecessor_node_lists, truth_tables, max_t, n_simulation_problems, comm=None):
    """Whole problem space, range-partitioned over the ranks of `comm` (one GPU each); every rank gets
    the final states and digests of all problems in index order (one all-gather each)."""
    from .dist import Comm, partition
    comm = comm or Comm()
    first, count = partition(n_simulation_problems, comm.world, comm.rank)
    finals, digests = simulate_digests(engine, origin_simulation_problem, simulation_problem_variations,
                                       predecessor_node_lists, truth_tables, max_t, first, count)
    return comm.gather_concat(finals), comm.gather_concat(digests)
