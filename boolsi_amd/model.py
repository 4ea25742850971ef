"""
Host-side pieces of the network model that are not state stepping.

`majority` and `encode_state` are needed by the YAML front end (rule evaluation, target
substate code) exactly as in the reference (`boolsi/model.py:6-13, 131-149`).  State stepping
itself (`apply_update_rules`, `simulate_step`, `simulate_n_steps`, the detection loop
`model.py:16-128,152-236`) runs on the GPU: see `boolsi_amd/engine.py` for the drop-in
wrappers that keep the reference's list-of-bools signatures.
"""


def majority(*args):
    """True iff more than half of the arguments are true; a tie is False (model.py:6-13)."""
    return sum(args) > len(args) / 2


def encode_state(substate_node_set, state):
    """(code, code restricted to `substate_node_set`) with node i weighted 2**i (model.py:131-149)."""
    code = 0
    sub = 0
    for node, on in enumerate(state):
        if on:
            code |= 1 << node
            if node in substate_node_set:
                sub |= 1 << node
    return code, sub


def decode_state(code, n_nodes):
    """Inverse of encode_state: list of n bools."""
    return [bool((code >> i) & 1) for i in range(n_nodes)]


def count_perturbations(perturbed_nodes_by_t):
    """Number of (t, node) perturbation entries (model.py:239-247)."""
    return sum(len(nodes) for nodes in perturbed_nodes_by_t.values())
