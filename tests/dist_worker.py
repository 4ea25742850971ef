"""Worker of tests/test_dist_ranks.py: one rank of a world_size-N job on CPU (socket data plane)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402

from boolsi_amd import synth, _lib  # noqa: E402
from boolsi_amd.attract import merge_tables, table_from_merged  # noqa: E402
from boolsi_amd.compile import compile_problem  # noqa: E402
from boolsi_amd.constants import Mode  # noqa: E402
from boolsi_amd.dist import Comm, partition  # noqa: E402
from boolsi_amd.input import parse_input_text  # noqa: E402
from oracle.cpu_oracle import Oracle  # noqa: E402   (stands in for the GPU engine on CPU-only hosts)


def main():
    out_path = sys.argv[1]
    comm = Comm.from_env(backend='socket')
    cfg = parse_input_text(synth.config3_yaml(), 4096, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    n_total = 20000 + 13                      # ragged on purpose
    first, count = partition(n_total, comm.world, comm.rank)
    _, table, none, steps = Oracle(net, space).attract(1000 + first, count, 4096)
    mine = np.zeros(len(table), _lib.ATTR_REC)
    for name in ('key', 'length', 'count', 'sum_l', 'sum_l2_lo', 'sum_l2_hi'):
        mine[name] = table[name]
    gathered = comm.allgather_records(table_from_merged(merge_tables([mine]), _lib.ATTR_REC))
    merged = merge_tables(gathered)
    none_all, steps_all = comm.allreduce_sum_int([none, steps])
    slowest = comm.allreduce_max(float(comm.rank))
    # ragged 2-D gather in rank order (what target / simulate use to join per-rank outputs)
    mine2d = np.arange(first, first + count, dtype=np.uint64).reshape(-1, 1).repeat(3, axis=1)[: (comm.rank + 1) * 5]
    joined = comm.gather_concat(mine2d)
    empty = comm.gather_concat(np.zeros((0, 2), np.uint32))
    # one rank with more rows than the single-collective fast path carries: second collective
    big = comm.gather_concat(np.arange(comm.rank * 1000, comm.rank * 1000 + (300 if comm.rank == 1 else 3), dtype=np.uint64))
    # more records on one rank than the single collective's slots: every rank sees it in the headers and the collective is repeated
    huge = comm.gather_concat(np.arange(comm.rank * 10000, comm.rank * 10000 + (3000 if comm.rank == comm.world - 1 else 2), dtype=np.uint64))
    # wide records (bsx_attr_rec2) and sums beyond 64 bits over the control plane (JSON carries Python ints of any size)
    wide = table_from_merged({5 + comm.rank: [16, (1 << 70) + comm.rank, (1 << 130) + 1, (1 << 200) + 3], 99: [2, 1 << 64, 7, 9]}, _lib.ATTR_REC2)
    wide_merged = merge_tables(comm.allgather_records(wide, slots=1))
    big_sum = comm.allreduce_sum_int([(1 << 130) + comm.rank, 1])
    comm.barrier()
    with open('{}.{}'.format(out_path, comm.rank), 'w') as f:
        json.dump({'rank': comm.rank, 'world': comm.world, 'first': first, 'count': count,
                   'merged': {str(k): v for k, v in merged.items()}, 'none': none_all, 'steps': steps_all,
                   'slowest': slowest, 'joined': joined.tolist(), 'empty_shape': list(empty.shape), 'big': big.tolist(), 'huge_len': len(huge), 'huge_tail': huge[-3:].tolist(),
                   'wide_merged': {str(k): [str(x) for x in v] for k, v in wide_merged.items()}, 'big_sum': [str(x) for x in big_sum], 'per_rank_counts': [len(g) for g in gathered]}, f)
    comm.shutdown()


if __name__ == '__main__':
    main()
