"""
Node correlations across attractors: frequency-weighted Spearman's rho and two-sided p-values
(reference `boolsi/attractor_analysis.py:29-145`; SURVEY f-3).  Host-side numpy on at most a few
thousand attractors; not on the GPU path.
"""
import logging

import numpy as np
from scipy import stats


def weighted_ranks(values, weights):
    """Average ranks of `values` when value i occurs weights[i] times."""
    order = np.argsort(values, kind='stable')
    sorted_values = values[order]
    uniq, start, inverse = np.unique(sorted_values, return_index=True, return_inverse=True)
    group_weight = np.add.reduceat(weights[order], start)
    before = np.concatenate(([0], np.cumsum(group_weight)[:-1]))
    group_rank = before + 0.5 * (group_weight + 1)
    ranks = np.empty(len(values), dtype=float)
    ranks[order] = group_rank[inverse]
    return ranks


def weighted_pearson(data, weights):
    """Pearson r between columns of `data` with integer row weights, and t-test p-values."""
    dof = weights.sum() - 2
    cov = np.cov(data.T, fweights=weights)
    var = cov.diagonal()
    with np.errstate(invalid='ignore', divide='ignore'):
        r = cov / np.sqrt(np.multiply.outer(var, var))
        t = r / np.sqrt((1 - r * r) / dof)
        p = 2 * stats.t.sf(np.abs(t), dof)
    return r, p


def compute_frequency_spearmanrho(data, frequencies):
    ranks = np.column_stack([weighted_ranks(data[:, j], frequencies) for j in range(data.shape[1])])
    return weighted_pearson(ranks, frequencies)


def find_node_correlations(attractors):
    """attractors: list of AggregatedAttractor with .states and .frequency -> (Rho, P) or None."""
    total = sum(a.frequency for a in attractors)
    if len(attractors) == 1 or total <= 2:
        logging.getLogger().info('Not enough attractors to infer node correlations.')
        return None
    observations = np.array([np.mean(np.array(a.states, dtype=float), axis=0) for a in attractors])
    frequencies = np.array([a.frequency for a in attractors])
    logging.getLogger().info('Computing node correlations...')
    return compute_frequency_spearmanrho(observations, frequencies)
