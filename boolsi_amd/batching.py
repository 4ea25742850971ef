"""
Host-side problem enumeration: the same mixed-radix numbering of simulation problems as the
reference's `boolsi/batching.py` (numeral system 10-46, number -> digits 212-229, digits ->
problem 160-209).  The device decodes indices itself (bsx_kernels.hip:init_problem); this module
is what labels results on the host (which fixed nodes / perturbations a simulation had) and what
the tests compare with the reference's enumeration.

The engine itself works on contiguous index ranges (dist.partition).  The reference's strided
chunking over MPI workers (49-157) only decides the ORDER in which a multi-process run lists its
simulations; `batch_layout` / `reference_order` / `reference_position` restate that order in closed
form so that `--reference-np P` can reproduce the listing of `mpiexec -np P boolsi ...`.
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


class BatchLayout:
    """
    How the reference cuts N problems into chunks (one per worker process) and batches
    (batching.py:79-157): problems are visited in the sequence j -> (increment * j) mod N; chunk c owns a
    contiguous run of j (the first N mod n_chunks chunks are one longer), every chunk is cut into
    `batches_per_chunk` runs (the first size mod batches_per_chunk of them one longer), and batch
    number = batch_in_chunk * n_chunks + chunk.  Enumeration of batches stops at the first empty one.
    """

    def __init__(self, n_simulation_problems, n_chunks, batches_per_chunk):
        self.n = n_simulation_problems
        self.n_chunks = n_chunks
        self.batches_per_chunk = batches_per_chunk
        self.increment = calculate_increment_for_chunking_simulation_problems(n_simulation_problems, n_chunks)
        self.small_chunk, self.n_big_chunks = divmod(n_simulation_problems, n_chunks)
        self.small_batch, self.n_big_batches_small_chunk = divmod(self.small_chunk, batches_per_chunk)

    def chunk_start(self, chunk):
        return chunk * self.small_chunk + min(chunk, self.n_big_chunks)

    def n_big_batches(self, chunk):
        return self.n_big_batches_small_chunk + (1 if chunk < self.n_big_chunks else 0)

    def batch_run(self, batch_in_chunk, chunk):
        """-> (offset of the batch within its chunk, batch size)."""
        big = self.n_big_batches(chunk)
        if batch_in_chunk < big:
            return batch_in_chunk * (self.small_batch + 1), self.small_batch + 1
        return big * (self.small_batch + 1) + (batch_in_chunk - big) * self.small_batch, self.small_batch

    def batches(self):
        """(first problem index, size) of every batch, in batch-number order."""
        for batch_in_chunk in range(self.batches_per_chunk):
            for chunk in range(self.n_chunks):
                offset, size = self.batch_run(batch_in_chunk, chunk)
                if size == 0:
                    return
                yield (self.increment * (self.chunk_start(chunk) + offset)) % self.n, size

    def position(self, index):
        """Sort key (batch number, place in batch) of a problem index: where the reference lists it."""
        if self.n == 1:
            return 0, 0
        j = (index * pow(self.increment, -1, self.n)) % self.n
        big_span = self.n_big_chunks * (self.small_chunk + 1)
        if j < big_span:
            chunk = j // (self.small_chunk + 1)
        else:
            chunk = self.n_big_chunks + (j - big_span) // self.small_chunk
        offset = j - self.chunk_start(chunk)
        big = self.n_big_batches(chunk)
        if offset < big * (self.small_batch + 1):
            batch_in_chunk, place = divmod(offset, self.small_batch + 1)
        else:
            batch_in_chunk, place = divmod(offset - big * (self.small_batch + 1), self.small_batch)
            batch_in_chunk += big
        return batch_in_chunk * self.n_chunks + chunk, place


def batch_layout(n_simulation_problems, n_processes, batches_per_process):
    """Layout of a reference run with `n_processes` MPI processes (rank 0 only coordinates when there
    are workers, mpi.py:108-116)."""
    return BatchLayout(n_simulation_problems, max(n_processes - 1, 1), batches_per_process)


def count_simulation_problem_batches(n_chunks, n_simulation_problems, batches_per_chunk):
    return min(n_chunks * batches_per_chunk, n_simulation_problems)


def generate_simulation_problem_batch_seeds(simulation_problem_variations, n_chunks, n_simulation_problems,
                                            batches_per_chunk):
    """Batch seeds in the reference's form: (digits of the first problem, batch size, digits of the
    increment, radices)."""
    radices, places = create_numeral_system_from_variations(simulation_problem_variations)
    layout = BatchLayout(n_simulation_problems, n_chunks, batches_per_chunk)
    increment_digits = convert_number_to_variational_representation(layout.increment, radices, places)
    for first, size in layout.batches():
        yield convert_number_to_variational_representation(first, radices, places), size, increment_digits, radices


def reference_order(layout):
    """Problem indices in the order the reference lists them: batch after batch, stride `increment`."""
    order = []
    for first, size in layout.batches():
        order.extend((first + k * layout.increment) % layout.n for k in range(size))
    return order
