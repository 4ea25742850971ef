"""
Host-side problem enumeration: the same mixed-radix numbering of simulation problems as the
reference's `boolsi/batching.py` (numeral system 10-46, number -> digits 212-229, digits ->
problem 160-209).  The device decodes indices itself (bsx_kernels.hip:init_problem); this module
is what labels results on the host (which fixed nodes / perturbations a simulation had) and what
the tests compare with the reference's enumeration.

The reference's strided chunking over MPI ranks (49-157) is replaced by contiguous ranges
(dist.partition): the set of problems is identical and a single process enumerates in index order
either way; `increment_for_chunking` is kept for SURVEY f-2.
"""
from .constants import NodeStateRange

_TERNARY = NodeStateRange.MAYBE_TRUE_OR_FALSE

# digit -> node state per range (batching.py:171-175); None = variation absent
_STATE_OF_DIGIT = {
    NodeStateRange.MAYBE_FALSE: (None, False),
    NodeStateRange.MAYBE_TRUE: (None, True),
    NodeStateRange.TRUE_OR_FALSE: (False, True),
    NodeStateRange.MAYBE_TRUE_OR_FALSE: (None, False, True),
}


def create_numeral_system_from_variations(simulation_problem_variations):
    """-> (radices, unit place values), least significant position first."""
    initial_state_variations, fixed_nodes_variations, perturbation_variations = simulation_problem_variations
    radices = [2] * len(initial_state_variations)
    radices += [3 if r == _TERNARY else 2 for _, r in fixed_nodes_variations]
    radices += [3 if r == _TERNARY else 2 for _, _, r in perturbation_variations]
    places, value = [], 1
    for radix in radices:
        places.append(value)
        value *= radix
    return radices, places


def convert_number_to_variational_representation(number, radices, unit_place_values):
    digits = [0] * len(radices)
    for pos in range(len(radices) - 1, -1, -1):
        digits[pos], number = divmod(number, unit_place_values[pos])
    return digits


def convert_variational_representation_to_simulation_problem(digits, origin_simulation_problem,
                                                             simulation_problem_variations):
    origin_state, origin_fixed, origin_pert = origin_simulation_problem
    state_vars, fixed_vars, pert_vars = simulation_problem_variations
    a, b = len(state_vars), len(state_vars) + len(fixed_vars)

    initial_state = list(origin_state)
    for node, digit in zip(state_vars, digits[:a]):
        initial_state[node] = bool(digit)

    fixed_nodes = dict(origin_fixed)
    for (node, rng), digit in zip(fixed_vars, digits[a:b]):
        state = _STATE_OF_DIGIT[rng][digit]
        if state is not None:
            fixed_nodes[node] = state

    perturbed_nodes_by_t = {t: dict(nodes) for t, nodes in origin_pert.items()}
    for (t, node, rng), digit in zip(pert_vars, digits[b:]):
        state = _STATE_OF_DIGIT[rng][digit]
        if state is not None:
            perturbed_nodes_by_t.setdefault(t, {})[node] = state
    return initial_state, fixed_nodes, perturbed_nodes_by_t


def problem_from_index(index, origin_simulation_problem, simulation_problem_variations, numeral_system=None):
    radices, places = numeral_system or create_numeral_system_from_variations(simulation_problem_variations)
    digits = convert_number_to_variational_representation(index, radices, places)
    return convert_variational_representation_to_simulation_problem(
        digits, origin_simulation_problem, simulation_problem_variations)


def calculate_increment_for_chunking_simulation_problems(n_simulation_problems, n_chunks):
    """Stride coprime to 2 and 3 closest to n_chunks (batching.py:49-76)."""
    if n_chunks % 2 and n_chunks % 3:
        increment = n_chunks
    elif n_chunks % 2:
        increment = n_chunks - 2
    elif n_chunks % 3 == 2:
        increment = n_chunks - 1
    else:
        increment = n_chunks + 1
    return increment % n_simulation_problems
