"""
Lowers the front end's configuration dict (reference layout: lists of bools, dict truth tables,
heap-ordered variation lists) to the flat arrays that cross the C-ABI (include/bsx.h).

Conventions:
  S1  state word w holds nodes 64w .. 64w+63, node i = bit (i % 64); code = sum state[i] << i
  S2  truth table of node i is a 2^k-bit mask; bit `idx` is the output for the predecessor
      assignment where predecessor j (ascending node order) has state (idx >> j) & 1
  S13 problem index digits, least significant first: one binary digit per `any` initial node
      (node order), then fixed-node variations, then perturbation variations (list order =
      heapq array order of the front end), radix 3 for 'any?' else 2
"""
from dataclasses import dataclass, field

import numpy as np

from .constants import RANGE_CODE

MAX_NODES = 256           # BSX_MAX_NODES
MAX_PREDECESSORS = 24     # BSX_MAX_PREDECESSORS (2^24-bit table = 2 MiB per node)


def state_to_words(state, n_words=None):
    n = len(state)
    n_words = n_words or max(1, (n + 63) // 64)
    words = np.zeros(n_words, dtype=np.uint64)
    for i, on in enumerate(state):
        if on:
            words[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    return words


def words_to_code(words):
    code = 0
    for w, v in enumerate(np.asarray(words, dtype=np.uint64).tolist()):
        code |= int(v) << (64 * w)
    return code


def code_to_words(code, n_words):
    return np.array([(code >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(n_words)], dtype=np.uint64)


def truth_table_to_mask(truth_table, k):
    """dict {tuple(bool)*k -> bool} -> int bitmask in S2 order."""
    mask = 0
    for row, value in truth_table.items():
        if value:
            idx = 0
            for j, bit in enumerate(row):
                if bit:
                    idx |= 1 << j
            mask |= 1 << idx
    assert len(truth_table) == 1 << k
    return mask


@dataclass
class CompiledNetwork:
    n_nodes: int
    n_words: int
    pred_offsets: np.ndarray     # u32[n+1]
    pred_idx: np.ndarray         # u32[sum k]
    tt_word_offsets: np.ndarray  # u32[n+1]
    tt_words: np.ndarray         # u64[sum ceil(2^k/64)]
    tt_masks: list = field(default_factory=list)   # python ints, for tests / golden comparison

    @property
    def predecessor_lists(self):
        return [self.pred_idx[self.pred_offsets[i]:self.pred_offsets[i + 1]].tolist()
                for i in range(self.n_nodes)]


@dataclass
class CompiledSpace:
    n_nodes: int
    origin_state: np.ndarray     # u64[W]
    any_nodes: np.ndarray        # u32[n_any]
    fixed: np.ndarray            # u32[n_fixed, 2]  (node, value)
    fixed_var: np.ndarray        # u32[n_fv, 2]     (node, range code)
    sched: np.ndarray            # u32[n_sched, 3]  (t, node, value)   sorted by (t, node)
    pert_var: np.ndarray         # u32[n_pv, 3]     (t, node, range code)
    n_problems: int              # python int, may exceed 2^64
    radices: list

    @property
    def last_perturbation_t(self):
        t = 0
        if len(self.sched):
            t = max(t, int(self.sched[:, 0].max()))
        if len(self.pert_var):
            t = max(t, int(self.pert_var[:, 0].max()))
        return t


def compile_network(predecessor_node_lists, truth_tables):
    n = len(predecessor_node_lists)
    if n > MAX_NODES:
        raise ValueError('networks of more than {} nodes are not supported by the engine'.format(MAX_NODES))
    pred_offsets = np.zeros(n + 1, dtype=np.uint32)
    tt_word_offsets = np.zeros(n + 1, dtype=np.uint32)
    pred_idx, tt_words, masks = [], [], []
    for i, (preds, table) in enumerate(zip(predecessor_node_lists, truth_tables)):
        k = len(preds)
        if k > MAX_PREDECESSORS:
            raise ValueError('node {} has {} predecessors; the engine supports at most {}'.format(
                i, k, MAX_PREDECESSORS))
        if list(preds) != sorted(preds):
            raise ValueError('predecessor lists must be sorted ascending')
        mask = truth_table_to_mask(table, k) if isinstance(table, dict) else int(table)
        masks.append(mask)
        n_tt_words = max(1, (1 << k) >> 6)
        for w in range(n_tt_words):
            tt_words.append((mask >> (64 * w)) & 0xFFFFFFFFFFFFFFFF)
        pred_idx.extend(preds)
        pred_offsets[i + 1] = len(pred_idx)
        tt_word_offsets[i + 1] = len(tt_words)
    return CompiledNetwork(
        n_nodes=n, n_words=max(1, (n + 63) // 64), pred_offsets=pred_offsets,
        pred_idx=np.array(pred_idx, dtype=np.uint32), tt_word_offsets=tt_word_offsets,
        tt_words=np.array(tt_words, dtype=np.uint64), tt_masks=masks)


def compile_space(origin_simulation_problem, simulation_problem_variations):
    initial_state, fixed_nodes, perturbed_nodes_by_t = origin_simulation_problem
    initial_state_variations, fixed_nodes_variations, perturbation_variations = \
        simulation_problem_variations
    n = len(initial_state)

    def rc(r):
        return RANGE_CODE[r] if not isinstance(r, int) else r

    fixed = np.array(sorted((int(node), int(bool(v))) for node, v in fixed_nodes.items()),
                     dtype=np.uint32).reshape(-1, 2)
    fixed_var = np.array([(int(node), rc(r)) for node, r in fixed_nodes_variations],
                         dtype=np.uint32).reshape(-1, 2)
    sched = np.array(sorted((int(t), int(node), int(bool(v)))
                            for t, nodes in perturbed_nodes_by_t.items() for node, v in nodes.items()),
                     dtype=np.uint32).reshape(-1, 3)
    pert_var = np.array([(int(t), int(node), rc(r)) for t, node, r in perturbation_variations],
                        dtype=np.uint32).reshape(-1, 3)
    radices = [2] * len(initial_state_variations) + \
        [3 if c == 3 else 2 for c in fixed_var[:, 1].tolist()] + \
        [3 if c == 3 else 2 for c in pert_var[:, 2].tolist()]
    n_problems = 1
    for r in radices:
        n_problems *= r
    return CompiledSpace(
        n_nodes=n, origin_state=state_to_words(initial_state),
        any_nodes=np.array(list(initial_state_variations), dtype=np.uint32),
        fixed=fixed, fixed_var=fixed_var, sched=sched, pert_var=pert_var,
        n_problems=n_problems, radices=radices)


def compile_problem(cfg):
    """cfg = dict returned by boolsi_amd.input.parse_input -> (CompiledNetwork, CompiledSpace)."""
    net = compile_network(cfg['incoming node lists'], cfg['truth tables'])
    space = compile_space(cfg['origin simulation problem'], cfg['simulation problem variations'])
    assert space.n_problems == cfg['total combination count']
    return net, space
