"""
Seeded synthetic Boolean networks, emitted as BoolSi YAML text.

The same text feeds the reference (when golden vectors are generated), the CPU oracle and the
engine, so the three can only differ in how they execute it.  RNG = random.Random(seed);
every node draws `k` distinct predecessors uniformly and 2^k truth-table bits i.i.d. p = 1/2;
the rule is written as a disjunction of the true rows in the reference's rule syntax.
"""
import random


def node_name(i):
    return 'x{}'.format(i)


def rule_text(preds, tt_mask):
    """DNF of a truth table given in SURVEY S2 bit order (bit j of the row index = preds[j])."""
    k = len(preds)
    rows = [idx for idx in range(1 << k) if (tt_mask >> idx) & 1]
    if not rows:
        return '0'
    if len(rows) == 1 << k:
        return '1'
    terms = []
    for idx in rows:
        lits = [('' if (idx >> j) & 1 else 'not ') + node_name(p) for j, p in enumerate(preds)]
        terms.append('(' + ' and '.join(lits) + ')')
    return ' or '.join(terms)


def random_network(n, k, seed):
    """-> (predecessor lists, truth-table masks) as drawn; constants keep their drawn operands
    out of the text, so the parsed predecessor list of an all-0 / all-1 rule is empty."""
    rng = random.Random(seed)
    preds, masks = [], []
    for _ in range(n):
        preds.append(sorted(rng.sample(range(n), k)))
        masks.append(rng.getrandbits(1 << k))
    return preds, masks


def network_yaml(n, k, seed, initial=None, fixed=None, perturbations=None, target=None):
    """
    :param initial: dict node -> '0' | '1' | 'any' (default: every node 'any')
    :param fixed: dict node -> state text ('0', '1', '0?', '1?', 'any', 'any?')
    :param perturbations: dict node -> dict state text -> times text (e.g. {'1': '7, 14-16'})
    :param target: dict node -> '0' | '1' | 'any' (all nodes must be present if given)
    """
    preds, masks = random_network(n, k, seed)
    out = ['nodes:']
    out += ['    - {}'.format(node_name(i)) for i in range(n)]
    out += ['', 'update rules:']
    out += ['    {}: {}'.format(node_name(i), rule_text(preds[i], masks[i])) for i in range(n)]
    out += ['', 'initial state:']
    initial = initial or {}
    out += ['    {}: {}'.format(node_name(i), initial.get(i, 'any')) for i in range(n)]
    if fixed:
        out += ['', 'fixed nodes:']
        out += ["    {}: '{}'".format(node_name(i), s) for i, s in fixed.items()]
    if perturbations:
        out += ['', 'perturbations:']
        for i, by_state in perturbations.items():
            out.append('    {}:'.format(node_name(i)))
            out += ["        '{}': '{}'".format(s, times) for s, times in by_state.items()]
    if target:
        out += ['', 'target state:']
        out += ['    {}: {}'.format(node_name(i), target[i]) for i in range(n)]
    return '\n'.join(out) + '\n'


def seeded_bits(n, seed):
    rng = random.Random(seed)
    return [rng.getrandbits(1) for _ in range(n)]


# ----------------------------------------------------------------------------- BASELINE.json configs

def config3_yaml():
    """n = 32, K = 2, all nodes 'any' -> 2^32 problems, attract."""
    return network_yaml(32, 2, 32)


def config4_yaml():
    """n = 64, K = 2; nodes 0-27 'any', rest seeded constants; three '0?' knock-outs; target on 8 nodes
    whose values are taken from a state the network reaches (tools/make_config4.py wrote the file)."""
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'config4.yaml')) as f:
        return f.read()


def config5_yaml(max_t=10000, n_any=26):
    """n = 128, K = 3; nodes 0..n_any-1 'any'; 4 nodes perturbed every 7th/11th/13th/17th step."""
    n = 128
    bits = seeded_bits(n, 1280)
    initial = {i: str(bits[i]) for i in range(n_any, n)}
    rng = random.Random(1281)
    pnodes = sorted(rng.sample(range(n), 4))
    pbits = seeded_bits(4, 1282)
    perturbations = {}
    for node, b, period in zip(pnodes, pbits, (7, 11, 13, 17)):
        times = ', '.join(str(t) for t in range(period, max_t + 1, period))
        perturbations[node] = {str(b): times}
    return network_yaml(n, 3, 128, initial=initial, perturbations=perturbations)


def north_star_yaml():
    """n = 64, K = 2, all nodes 'any' (2^64 problems; runs take an index range of it)."""
    return network_yaml(64, 2, 64)
