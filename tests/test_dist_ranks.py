"""
Multi-process path on CPU (world_size 2 and 3): range partition + all-gather merge of the
attractor tables must reproduce the single-process table exactly.  The ranks use
the CPU oracle in place of the GPU engine and BSX_DIST_BACKEND=socket in place of RCCL (the data
collective then rides the product's own TCP control plane -- no GPU, no torch); everything else
(rendezvous, partition, record packing, count exchange, integer merge) is the product code of
boolsi_amd/dist.py and attract.py.  The RCCL data plane itself is covered on the GPU box
(tests/test_gpu_comm.py).
"""
import time
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
def test_allgather_merge_equals_single_process(tmp_path, world):
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
        # a rank with more records than the collective's slots (the collective is repeated once, with the largest count)
        assert r['huge_len'] == 2 * (world - 1) + 3000 and r['huge_tail'] == [(world - 1) * 10000 + v for v in (2997, 2998, 2999)]
        # wide records: counts beyond 2^64 and sums beyond 2^128 survive packing, the collective and the merge
        expect_wide = {str(5 + q): ['16', str((1 << 70) + q), str((1 << 130) + 1), str((1 << 200) + 3)] for q in range(world)}
        expect_wide['99'] = ['2', str(world << 64), str(7 * world), str(9 * world)]
        assert r['wide_merged'] == expect_wide
        assert r['big_sum'] == [str(world * (1 << 130) + sum(range(world))), str(world)]
    assert sum(r['count'] for r in results) == 20013


def _spawn(world, script, tmp_path, extra_env=None):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world),
                   RANK=str(rank), LOCAL_RANK=str(rank), BSX_RDZV_DIR=str(tmp_path), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, cwd=os.path.dirname(HERE),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    return procs


def test_a_rank_that_dies_ends_the_job_instead_of_hanging_it(tmp_path):
    """Rank 1 leaves before the collective: the others must get PeerLost promptly (a failing rank
    used to leave its peers blocked in the all-gather)."""
    script = (
        "import os, sys, time\n"
        "from boolsi_amd.dist import Comm, PeerLost\n"
        "c = Comm.from_env(backend='socket')\n"
        "c.barrier()\n"
        "if c.rank == 1:\n"
        "    c.abort(); sys.exit(3)\n"
        "try:\n"
        "    c.allgather_obj(c.rank); c.allgather_obj(c.rank)\n"
        "except PeerLost as e:\n"
        "    print('PEERLOST', e); sys.exit(4)\n"
        "sys.exit(0)\n")
    t0 = time.time()
    procs = _spawn(3, script, tmp_path)
    outs = [p.communicate(timeout=60)[0] for p in procs]
    assert time.time() - t0 < 30
    assert [p.returncode for p in procs] == [4, 3, 4], outs
    assert not [f for f in os.listdir(tmp_path) if f.startswith('bsx_rdzv_')]       # rendezvous file removed


def test_data_collective_needs_rccl_or_an_explicit_socket_backend(tmp_path):
    """Default backend = RCCL: without an attached engine the data plane refuses instead of falling back."""
    script = (
        "import numpy as np, sys\n"
        "from boolsi_amd.dist import Comm\n"
        "c = Comm.from_env()\n"
        "assert c.backend == 'rccl'\n"
        "assert c.allreduce_sum_int([c.rank + 1, 10]) == [3, 20]        # control plane works without a GPU\n"
        "assert c.broadcast_obj('dir-of-rank-%d' % c.rank) == 'dir-of-rank-0'\n"
        "try:\n"
        "    c.allgather_records(np.zeros(2, np.uint64))\n"
        "except RuntimeError as e:\n"
        "    assert 'attach_engine' in str(e); sys.exit(0)\n"
        "sys.exit(1)\n")
    procs = _spawn(2, script, tmp_path)
    outs = [p.communicate(timeout=60)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs


def test_stale_rendezvous_file_is_ignored(tmp_path):
    """A file left by an earlier job with the same key (wrong port / nonce) must not break the rendezvous."""
    import hashlib
    import struct
    port = free_port()
    key = 'fixed-key-for-test'
    digest = hashlib.sha256(key.encode()).digest()
    stale = tmp_path / 'bsx_rdzv_{}_{}'.format(os.getuid(), digest[:8].hex())
    stale.write_bytes(struct.pack('<I', free_port()) + b'stale!!!')
    script = ("from boolsi_amd.dist import Comm\n"
              "c = Comm.from_env(backend='socket')\n"
              "assert c.allreduce_max(c.rank) == 1.0\n"
              "c.shutdown()\n")
    procs = []
    for rank in (1, 0):         # rank 1 first: it meets the stale file
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', RANK=str(rank),
                   BSX_RDZV_DIR=str(tmp_path), BSX_RDZV_KEY=key)
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, cwd=os.path.dirname(HERE)))
        time.sleep(0.5)
    assert [p.wait(timeout=60) for p in procs] == [0, 0]
