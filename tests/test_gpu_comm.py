"""
RCCL data plane through the C-ABI (bsx_comm_*, include/bsx.h) on the GPU box.  One GPU means a world of
one rank -- RCCL refuses two ranks on one device -- so this pins: librccl loads, the communicator comes
up on the engine's device, ncclAllGather runs on the engine's stream and returns the bytes it was
given, and boolsi_amd.dist.Comm drives exactly that path (BSX_FORCE_DIST=1 keeps the whole machinery
on for a single rank).  World sizes 2 and 3 are covered on CPU in tests/test_dist_ranks.py.
"""
import os
import socket

import numpy as np
import pytest

from boolsi_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng():
    from boolsi_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_rccl_world_of_one_allgather_roundtrip(eng):
    from boolsi_amd.engine import EngineError
    uid = eng.comm_unique_id()
    assert len(uid) == _lib.COMM_ID_BYTES and any(uid)
    with pytest.raises(EngineError):
        eng.comm_allgather(np.arange(4, dtype=np.uint8), 1)         # no communicator yet
    eng.comm_init(uid, 0, 1)
    with pytest.raises(EngineError):
        eng.comm_init(uid, 0, 1)                                    # one communicator per handle
    for size in (1, 72, 8 + 64 * 72, 1 << 20):
        send = np.random.default_rng(size).integers(0, 256, size, dtype=np.uint8)
        assert np.array_equal(eng.comm_allgather(send, 1), send)
    eng.comm_destroy()
    eng.comm_destroy()                                              # idempotent
    with pytest.raises(EngineError):
        eng.comm_init(uid, 1, 1)                                    # rank outside the world


def test_comm_layer_merges_through_rccl(eng, tmp_path, monkeypatch):
    from boolsi_amd.attract import merge_tables, table_from_merged
    from boolsi_amd.dist import Comm
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    for k, v in dict(BSX_FORCE_DIST='1', WORLD_SIZE='1', RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                     BSX_RDZV_DIR=str(tmp_path)).items():
        monkeypatch.setenv(k, v)
    monkeypatch.delenv('BSX_DIST_BACKEND', raising=False)
    comm = Comm.from_env()
    assert comm.active and comm.backend == 'rccl' and comm.world == 1
    with pytest.raises(RuntimeError):
        comm.allgather_records(np.zeros(1, _lib.ATTR_REC))          # not attached: refuses, no silent fallback
    comm.attach_engine(eng)
    recs = np.zeros(5, _lib.ATTR_REC)
    recs['count'] = np.arange(5) + 1
    recs['key'][:, 0] = np.arange(5) * 7
    (back,) = comm.allgather_records(recs)
    assert back.tobytes() == recs.tobytes()
    (none,) = comm.allgather_records(recs[:0])
    assert len(none) == 0
    rows = np.arange(12, dtype=np.uint64).reshape(4, 3)
    assert np.array_equal(comm.gather_concat(rows), rows)

    # attract over the whole space of a small network, with the RCCL merge in the loop
    from boolsi_amd.attract import attract_master
    from boolsi_amd.constants import Mode
    from boolsi_amd.input import parse_input_text
    from util import load
    case = load('attract_toy.json')[0]
    cfg = parse_input_text(case['yaml'], np.inf, Mode.ATTRACT)
    args = (cfg['origin simulation problem'], cfg['simulation problem variations'], cfg['incoming node lists'],
            cfg['truth tables'], np.inf, np.inf, cfg['total combination count'])
    with_rccl = attract_master(eng, *args, comm=comm, with_states=False)
    alone = attract_master(eng, *args, with_states=False)
    as_rows = lambda res: [(a.key, a.length, a.frequency, a.sum_l, a.sum_l2) for a in res[0]]
    assert as_rows(with_rccl) == as_rows(alone) and with_rccl[1:3] == alone[1:3]
    comm.shutdown()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_bench_two_ranks_as_child_processes_equal_one_rank(tmp_path):
    """bench.py --gpus 2 as the driver launches it -- one fresh process per rank, rendezvous from RANK / WORLD_SIZE /
    MASTER_* -- here with both ranks on GPU 0 and the socket data plane (RCCL refuses two ranks per device): the merged
    table of the timed steps must equal the table one rank computes over the same blocks, and the line must say which
    data plane carried the merge.  Without BSX_DIST_BACKEND=socket or --allow-socket-merge the run must fail instead of
    downgrading by itself."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(world, steps, warmup, extra_env, out_name, expect_ok=True):
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port), BSX_BENCH_DEVICE='0', BSX_RDZV_TIMEOUT='120', **extra_env)
            if world == 1:
                for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
                    env.pop(k)
            cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(world), '--steps', str(steps), '--warmup', str(warmup),
                   '--log2-batch', '48', '--no-cpu-baseline', '--dump-table', str(tmp_path / out_name)]
            procs.append(subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=600) for p in procs]
        codes = [p.returncode for p in procs]
        if not expect_ok:
            return codes, outs
        assert codes == [0] * world, outs
        line = json.loads(outs[0][0].strip().splitlines()[-1])
        return line, json.load(open(tmp_path / out_name))

    two, table2 = run(2, 3, 1, {'BSX_DIST_BACKEND': 'socket'}, 'two.json')          # timed blocks 2 .. 7
    one, table1 = run(1, 6, 2, {}, 'one.json')                                       # the same blocks on one rank
    assert two['n_gpus'] == 2 and 'socket' in two['config']['merge'] and one['n_gpus'] == 1
    assert table2 == table1 and len(table1) == 2
    assert two['roofline']['kernel'] == 'k_attract_pool<2,2,1,true,false>'
    assert abs(two['attractors_per_s'] / (6 * 2 ** 48 / (two['ms_per_step'] * 3e-3)) - 1) < 1e-6
    # RCCL cannot serve two ranks on one device: no flag, no downgrade -- every rank exits non-zero
    codes, outs = run(2, 1, 1, {}, 'never.json', expect_ok=False)
    assert all(c != 0 for c in codes), outs
