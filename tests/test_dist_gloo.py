"""
Multi-process path on CPU (gloo, world_size 2 and 3): range partition + all-gather merge of the
attractor tables must reproduce the single-process table exactly.  The ranks use
the CPU oracle in place of the GPU engine; everything else (partition, record packing,
collectives, integer merge) is the product code of boolsi_amd/dist.py and attract.py.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from boolsi_amd import synth
from boolsi_amd.attract import merge_tables
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.dist import partition
from boolsi_amd.input import parse_input_text

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_partition_covers_range_exactly():
    for n in (0, 1, 7, 1000, 2 ** 64 + 5, 2 ** 130):
        for world in (1, 2, 3, 8):
            spans = [partition(n, world, r) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == n
            for (a, ca), (b, _) in zip(spans, spans[1:]):
                assert a + ca == b
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


@pytest.mark.parametrize('world', [2, 3])
def test_gloo_allgather_merge_equals_single_process(tmp_path, world):
    port = free_port()
    out = str(tmp_path / 'result.json')
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world),
                   RANK=str(rank), LOCAL_RANK=str(rank), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    results = [json.load(open('{}.{}'.format(out, r))) for r in range(world)]

    from oracle.cpu_oracle import Oracle
    import numpy as np
    from boolsi_amd import _lib
    cfg = parse_input_text(synth.config3_yaml(), 4096, Mode.ATTRACT)
    net, space = compile_problem(cfg)
    _, table, none, steps = Oracle(net, space).attract(1000, 20013, 4096)
    ref = np.zeros(len(table), _lib.ATTR_REC)
    for name in ('key', 'length', 'count', 'sum_l', 'sum_l2_lo', 'sum_l2_hi'):
        ref[name] = table[name]
    expect = {str(k): v for k, v in merge_tables([ref]).items()}
    for r in results:
        assert r['world'] == world
        assert r['merged'] == expect            # identical on every rank
        assert r['none'] == none and r['steps'] == steps
        assert r['slowest'] == world - 1
        expect_joined = []
        for q in range(world):
            lo, cnt = partition(20013, world, q)
            expect_joined += [[v, v, v] for v in range(lo, lo + cnt)][: (q + 1) * 5]
        assert r['joined'] == expect_joined and r['empty_shape'] == [0, 2]
        assert r['big'] == [v for q in range(world) for v in range(q * 1000, q * 1000 + (300 if q == 1 else 3))]
    assert sum(r['count'] for r in results) == 20013
