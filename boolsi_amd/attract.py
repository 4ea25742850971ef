"""
Attract mode: finds and aggregates attractors over a range of simulation problems.

Mirrors the reference's `boolsi/attract.py`:
  AggregatedAttractor (18-45) ........ same fields; mean / M2 derived once from exact integer sums
  attract_master (67-230) ............ `attract_master` below: engine runs over index tiles, merge,
                                       final order = ascending (-frequency, key) (170-173)
  store_attractor / write_aggregated_attractors_to_db (374-455) ... merge_tables
The per-problem solvers (262-371) run on the GPU (bsx_run_attract).  The reduce-memory flag (-r)
is accepted and ignored: the device detector is already O(1) memory and yields the default
detector's result (S7); see DESIGN.md for the reference's -r/-t defect that is not reproduced.
"""
import logging
from math import inf

import numpy as np

from .compile import compile_network, compile_space, code_to_words, words_to_code
from .dist import Comm, partition
from .engine import key_to_int
from .model import decode_state



class AggregatedAttractor:
    """Attractor with its basin statistics (reference attract.py:18-45)."""

    def __init__(self, key, length, frequency, sum_l, sum_l2, states=None):
        self.key = key
        self.length = length
        self.frequency = frequency
        self.sum_l = sum_l
        self.sum_l2 = sum_l2
        self.states = states        # list of lists of bool, first state = key state

    @property
    def trajectory_l_mean(self):
        return self.sum_l / self.frequency

    @property
    def trajectory_l_variation_sum(self):
        # M2 = sum (l - mean)^2, from integers: (c*S2 - S1^2) / c
        return (self.frequency * self.sum_l2 - self.sum_l * self.sum_l) / self.frequency


def _words(a):
    """little-endian 64-bit words (numpy array, list, or a plain int) -> Python int"""
    if np.ndim(a) == 0:
        return int(a)
    v = 0
    for w, x in enumerate(a):
        v |= int(x) << (64 * w)
    return v


def record_ints(a):
    """One bsx_attr_rec / bsx_attr_rec2 (numpy record or dict-like) -> (key, length, count, sum_l, sum_l2) as ints."""
    names = a.dtype.names if hasattr(a, 'dtype') else a.keys()
    if 'sum_l2' in names:
        return key_to_int(a['key']), int(a['length']), _words(a['count']), _words(a['sum_l']), _words(a['sum_l2'])
    return (key_to_int(a['key']), int(a['length']), int(a['count']), int(a['sum_l']),
            int(a['sum_l2_lo']) + (int(a['sum_l2_hi']) << 64))


def merge_tables(tables):
    """List of bsx_attr_rec / bsx_attr_rec2 arrays -> dict key(int) -> [length, count, sum_l, sum_l2] (exact ints)."""
    merged = {}
    for table in tables:
        names = getattr(getattr(table, 'dtype', None), 'names', None)
        if names and len(table):
            # numpy records -> plain Python numbers in one go (per-field access costs microseconds per record)
            wide = 'sum_l2' in names
            rows = zip(*(table[n].tolist() for n in (('key', 'length', 'count', 'sum_l', 'sum_l2') if wide else
                                                      ('key', 'length', 'count', 'sum_l', 'sum_l2_lo', 'sum_l2_hi'))))
            recs = []
            for r in rows:
                key = r[0][0] | r[0][1] << 64 | r[0][2] << 128 | r[0][3] << 192
                if wide:
                    recs.append((key, r[1], r[2][0] | r[2][1] << 64, r[3][0] | r[3][1] << 64 | r[3][2] << 128,
                                 r[4][0] | r[4][1] << 64 | r[4][2] << 128 | r[4][3] << 192))
                else:
                    recs.append((key, r[1], r[2], r[3], r[4] | r[5] << 64))
        else:
            recs = [record_ints(a) for a in table]
        for key, length, count, s1, s2 in recs:
            e = merged.get(key)
            if e is None:
                merged[key] = [length, count, s1, s2]
            else:
                if e[0] != length:
                    raise RuntimeError('attractor {} reported with two lengths'.format(key))
                e[1] += count
                e[2] += s1
                e[3] += s2
    return merged


def table_from_merged(merged, dtype):
    """The merged dict as an array of `dtype` records (ATTR_REC: the sums must fit its 64 / 128-bit fields)."""
    m64 = 0xFFFFFFFFFFFFFFFF
    out = np.zeros(len(merged), dtype)
    wide = 'sum_l2' in dtype.names
    for i, (key, (length, count, s1, s2)) in enumerate(sorted(merged.items())):
        for w in range(4):
            out[i]['key'][w] = (key >> (64 * w)) & m64
        out[i]['length'] = length
        if wide:
            for name, v, nwords in (('count', count, 2), ('sum_l', s1, 3), ('sum_l2', s2, 4)):
                if v >> (64 * nwords):
                    raise OverflowError('{} does not fit {} bits'.format(name, 64 * nwords))
                for w in range(nwords):
                    out[i][name][w] = (v >> (64 * w)) & m64
        else:
            if count >> 64 or s1 >> 64 or s2 >> 128:
                raise OverflowError('sums exceed bsx_attr_rec: use ATTR_REC2')
            out[i]['count'], out[i]['sum_l'] = count, s1
            out[i]['sum_l2_lo'], out[i]['sum_l2_hi'] = s2 & m64, s2 >> 64
    return out


def run_attract_range(engine, first, count, max_t=inf, max_attractor_l=inf, cap=1 << 20):
    """Engine over [first, first+count) -> (merged dict, n_no_attractor, stats dict).  ONE engine call however large
    the range (bsx_run_attract2: 128-bit index and count, wide sums), as one attract_master run covers the
    reference's whole N (attract.py:67-230); ranges beyond 2^128 problems are cut into calls of 2^127."""
    merged_tables, none = [], 0
    stats = {'problems': 0, 'state_steps': 0, 'executed_steps': 0, 'kernel_ms': 0.0, 'total_ms': 0.0,
             'kernel_launches': 0}
    done = 0
    while done < count:
        piece = min(count - done, 1 << 127)
        r = engine.attract2(first + done, piece, max_t, max_attractor_l, cap=cap)
        merged_tables.append(r.table)
        none += r.n_no_attractor
        for k in stats:
            stats[k] += r.stats[k]
        done += piece
    return merge_tables(merged_tables), none, stats


def attract_master(engine, origin_simulation_problem, simulation_problem_variations,
                   predecessor_node_lists, truth_tables, max_t, max_attractor_l,
                   n_simulation_problems, comm=None, with_states=True):
    """
    Attract over the whole problem space, range-partitioned over `comm` (one rank per GPU).
    Returns (list of AggregatedAttractor in final order, n_no_attractor, total_frequency, stats);
    identical on every rank.  Log lines follow attract.py:95-138.
    """
    comm = comm or Comm()
    log = logging.getLogger()
    n_processes_text = '{} GPU processes'.format(comm.world) if comm.world > 1 else 'Single process'
    if comm.rank == 0:
        log.info('{} will be used to find attractors from {} initial conditions...'.format(
            n_processes_text, n_simulation_problems))

    net = compile_network(predecessor_node_lists, truth_tables)
    space = compile_space(origin_simulation_problem, simulation_problem_variations)
    engine.set_problem(net, space)
    first, count = partition(n_simulation_problems, comm.world, comm.rank)
    merged, none, stats = run_attract_range(engine, first, count, max_t, max_attractor_l)

    if comm.active:
        from . import _lib
        tables = comm.allgather_records(table_from_merged(merged, _lib.ATTR_REC2))
        merged = merge_tables(tables)
        none, steps, execd = comm.allreduce_sum_int([none, stats['state_steps'], stats['executed_steps']])
        stats['state_steps'], stats['executed_steps'] = steps, execd
        stats['problems'] = n_simulation_problems

    total_frequency = sum(e[1] for e in merged.values())
    assert total_frequency + none == n_simulation_problems

    if comm.rank == 0:
        parts = ['Found {} attractors.'.format(len(merged))]
        if total_frequency < n_simulation_problems:
            p = ['No attractor']
            if max_attractor_l < inf:
                p.append('of length {} or less'.format(max_attractor_l))
            p.append('can be')
            if max_t < inf:
                p.append('detected in')
                p.append('1 time step' if max_t == 1 else '{} or less time steps'.format(max_t))
            else:
                p.append('reached')
            p.append('from {:.2%} initial conditions.'.format(1 - total_frequency / n_simulation_problems))
            parts.append(' '.join(p))
        log.info(' '.join(parts))

    order = sorted(merged.items(), key=lambda kv: (-kv[1][1], kv[0]))     # attract.py:170-173
    attractors = []
    for key, (length, freq, s1, s2) in order:
        states = None
        if with_states:
            states = cycle_states(engine, key, length)
        attractors.append(AggregatedAttractor(key, length, freq, s1, s2, states))
    return attractors, none, total_frequency, stats


def cycle_states(engine, key, length):
    """
    States of the attractor, starting at its key state (attract.py:22-25 rotation).  Regenerated on
    the device by stepping from the key under the origin problem's fixed nodes (attract mode allows
    no fixed-node variations, input.py:392-397, so every problem shares them).
    """
    states = engine.states_from(key, length - 1)
    return [decode_state(words_to_code(s), engine.net.n_nodes) for s in states]
