"""
CSV output of simulations, attractors and node correlations, in the reference's formats
(`boolsi/output.py`: summaries + states files 381-454, simulations 460-537, attractors 540-689,
write_states / format_state 845-865, 1092-1110, node correlations 868-1089).

Graphic output (PDF / SVG / PNG / TIFF through matplotlib + seaborn) is outside this engine's
scope: the corresponding switches are accepted by the CLI and ignored with a
warning.  Files are written with csv.writer defaults (CRLF rows), as the reference does.
"""
import csv
import logging
import os

import numpy as np

from .constants import simulation_name, aggregated_attractor_name


def list_texts(texts):
    """'a' / 'a and b' / 'a, b, and c'."""
    if len(texts) == 1:
        return texts[0]
    joiner = ', and ' if len(texts) > 2 else ' and '
    return joiner.join([', '.join(texts[:-1]), texts[-1]])


def format_state(state, t, fixed_nodes, perturbations):
    """'0'/'1' per node, '_' appended for fixed nodes, '*' for nodes perturbed at time t."""
    cells = [str(int(v)) for v in state]
    for node in fixed_nodes:
        cells[node] += '_'
    for node in perturbations.get(t, ()):
        cells[node] += '*'
    return cells


def write_states(states, fixed_nodes, perturbed_nodes_by_t, states_id, time_labels):
    prefix = [states_id] if states_id else []
    return [prefix + [time_labels[t] if time_labels else str(t)] +
            format_state(state, t, fixed_nodes, perturbed_nodes_by_t)
            for t, state in enumerate(states)]


def _open_csv(stack, output_dirpath, filename):
    f = open(os.path.join(output_dirpath, filename), 'w', newline='', encoding='utf-8')
    stack.append(f)
    return csv.writer(f)


def _output_results(results, result_name, summaries_header, display_info, summaries_row, node_names,
                    output_dirpath, preamble_row=None):
    os.makedirs(output_dirpath, exist_ok=True)
    summaries_filename = '{}_summaries.csv'.format(result_name)
    states_filename = '{}s.csv'.format(result_name)
    logging.getLogger().info('Printing {}s to {}...'.format(result_name, list_texts(
        ['"{}"'.format(os.path.join(summaries_filename, '')), '"{}"'.format(os.path.join(states_filename, ''))])))
    files = []
    try:
        summaries = _open_csv(files, output_dirpath, summaries_filename)
        states_writer = _open_csv(files, output_dirpath, states_filename)
        summaries.writerow(summaries_header)
        states_writer.writerow(['{}_id'.format(result_name), 'time'] + list(node_names))
        if preamble_row is not None:
            summaries.writerow(preamble_row)
        for index, result in enumerate(results):
            states, fixed_nodes, perturbed_nodes_by_t, time_labels = display_info(result)
            row = summaries_row(result, index)
            summaries.writerow(row)
            states_writer.writerows(write_states(states, fixed_nodes, perturbed_nodes_by_t, row[0], time_labels))
    finally:
        for f in files:
            f.close()


def output_simulations(simulations, node_names, output_dirpath):
    """simulation_summaries.csv + simulations.csv (reference output.py:460-537)."""
    n = len(node_names)

    def display_info(sim):
        return sim.states, sim.fixed_nodes, sim.perturbed_nodes_by_t, list(range(len(sim.states)))

    def summaries_row(sim, index):
        per_node = [0] * n
        for nodes in sim.perturbed_nodes_by_t.values():
            for node in nodes:
                per_node[node] += 1
        return ['{}{}'.format(simulation_name, index + 1), len(sim.states) - 1] + \
               [int(node in sim.fixed_nodes) for node in range(n)] + per_node

    header = [simulation_name + '_id', 'length'] + ['{}_is_fixed'.format(x) for x in node_names] + \
             ['n_{}_perturbations'.format(x) for x in node_names]
    _output_results(simulations, simulation_name, header, display_info, summaries_row, node_names, output_dirpath)


def output_attractors(attractors, total_frequency, fixed_nodes, node_names, n_simulation_problems,
                      max_attractor_l, max_t, output_dirpath):
    """attractor_summaries.csv + attractors.csv (reference output.py:540-689)."""
    def display_info(a):
        labels = ['t'] + ['t+{}'.format(t) for t in range(1, len(a.states))]
        return a.states, fixed_nodes, {}, labels

    def summaries_row(a, index):
        sd = np.sqrt(a.trajectory_l_variation_sum / (a.frequency - 1)) if a.frequency > 1 else np.nan
        return ['{}{}'.format(aggregated_attractor_name, index + 1), len(a.states), a.trajectory_l_mean, sd,
                a.frequency / n_simulation_problems]

    preamble = None
    if total_frequency < n_simulation_problems:
        # the reference's CLI passes inf (never None) for unset caps, which prints as '<= inf'
        preamble = ['no_' + aggregated_attractor_name,
                    '' if max_attractor_l is None else '<= {}'.format(max_attractor_l),
                    '' if max_t is None else '<= {}'.format(max_t), '',
                    1 - total_frequency / n_simulation_problems]
    header = [aggregated_attractor_name + '_id', 'length', 'trajectory_length_mean', 'trajectory_length_SD',
              'relative_frequency']
    _output_results(attractors, aggregated_attractor_name, header, display_info, summaries_row, node_names,
                    output_dirpath, preamble)


def output_node_correlations(Rho, P, p_value, node_names, output_dirpath):
    """node_correlations.csv: significant, then nonsignificant, then absent pairs (output.py:1077-1089)."""
    os.makedirs(output_dirpath, exist_ok=True)
    logging.getLogger().info('Printing node correlations to {}...'.format(list_texts(['"node_correlations.csv"'])))
    na = np.isnan(Rho)
    with np.errstate(invalid='ignore'):
        significant = P < p_value
    nonsignificant = ~na & ~significant
    groups = [(significant, lambda c: (1 - abs(c[1]), c[0][0], c[0][1])),
              (nonsignificant, lambda c: (c[2], 1 - abs(c[1]), c[0][0], c[0][1])),
              (na, lambda c: (c[0][0], c[0][1]))]
    with open(os.path.join(output_dirpath, 'node_correlations.csv'), 'w', newline='', encoding='utf-8') as f:
        writer = csv.writer(f)
        writer.writerow(['node_1', 'node_2', 'rho', 'p_value'])
        for mask, key in groups:
            upper = np.triu(mask, 1)
            pairs = sorted(zip(np.argwhere(upper).tolist(), Rho[upper], P[upper]), key=key)
            for (a, b), rho, p in pairs:
                writer.writerow([node_names[a], node_names[b], rho, p])
