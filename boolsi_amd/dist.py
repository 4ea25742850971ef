"""
Multi-GPU layer: one process per GPU, problems range-partitioned, one all-gather at the end.

Replaces the reference's mpi4py master/worker task farm (boolsi/mpi.py:193-340, 379-495): the
simulation problems are independent, so rank r simply takes the contiguous index range
[r*N/G, (r+1)*N/G) and the only data exchange is the merge of the per-rank result tables -- ONE
`ncclAllGather` (RCCL over xGMI) of fixed-size integer records, issued through the engine's C-ABI
(`bsx_comm_allgather`, include/bsx.h) on the engine's own device and stream.

Two planes:
  control  `Bootstrap`: a TCP star on the launcher's MASTER_ADDR (rank 0 is the hub).  It carries the
           128-byte ncclUniqueId from rank 0 to the others, barriers, scalar reductions of run statistics
           (JSON: ints of any size, floats, bools, strings -- nothing executable is ever deserialised), and it
           is how a rank learns that a peer has died (closed socket) instead of blocking in a collective for
           ever.  Pure Python, no MPI, no tensor framework.
           Rendezvous, ONE NODE (the default, what torch.distributed.run / the driver gives): rank 0 listens on
           an ephemeral port and publishes it, with a 32-byte secret, in a 0600 file in the local temp directory
           named after MASTER_ADDR / MASTER_PORT / the launcher's run id (the port itself belongs to the
           launcher's store).  SEVERAL NODES or ranks without a common temp directory (mpirun, srun, separate
           shells): set BSX_CONTROL_PORT (rank 0 listens there, no file) and BSX_RDZV_KEY (the shared secret)
           on every rank.  Either way a peer must answer an HMAC-SHA256 challenge keyed by the secret before
           anything it sends is looked at, and frames are capped at 1 GiB.
  data     RCCL, once `attach_engine` has built the communicator.  Without an engine the data collectives
           refuse to run unless BSX_DIST_BACKEND=socket explicitly routes them through the control
           plane -- that is for CPU-only tests and for rehearsing several ranks on ONE GPU (RCCL wants
           one device per rank); it is never picked silently.
"""
import hashlib
import hmac
import json
import os
import socket
import struct
import tempfile
import time

import numpy as np

from . import _lib

_MAGIC = b'BSX2'
_MAX_FRAME = 1 << 30


class PeerLost(RuntimeError):
    """A rank of the job went away (its control connection closed) or reported a failure."""


def _recv_exact(sock, n):
    chunks, got = [], 0
    while got < n:
        part = sock.recv(min(n - got, 1 << 20))
        if not part:
            raise PeerLost('a rank of the job closed its control connection')
        chunks.append(part)
        got += len(part)
    return b''.join(chunks)


def _send_frame(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_frame(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    if n > _MAX_FRAME:
        raise PeerLost('control frame of {} bytes exceeds the limit'.format(n))
    return _recv_exact(sock, n)


def _mac(secret, *parts):
    return hmac.new(secret, b'|'.join(parts), hashlib.sha256).digest()


class Bootstrap:
    """Control plane of a job: all-gather of small byte strings over a TCP star (rank 0 = hub)."""

    def __init__(self, rank, world, addr, key, timeout=300.0, port=None):
        self.rank, self.world = rank, world
        self._peers = {}            # hub: rank -> socket
        self._hub = None            # others: socket to rank 0
        self._listener = None
        digest = hashlib.sha256(key.encode()).digest()
        directory = os.environ.get('BSX_RDZV_DIR', tempfile.gettempdir())
        self._path = None if port else os.path.join(directory, 'bsx_rdzv_{}_{}'.format(os.getuid(), digest[:8].hex()))
        deadline = time.monotonic() + timeout
        who = struct.pack('<QQ', rank, world)
        if rank == 0:
            self._listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            self._listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            self._listener.bind((addr, port or 0))
            self._listener.listen(world)
            # the secret: random and handed over through the 0600 file (one node), or derived from the shared key
            secret = digest if port else os.urandom(32)
            if not port:
                tmp = '{}.{}'.format(self._path, os.getpid())
                fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
                with os.fdopen(fd, 'wb') as f:
                    f.write(struct.pack('<I', self._listener.getsockname()[1]) + secret)
                os.replace(tmp, self._path)     # atomic: readers see the old file or the whole new one
            while len(self._peers) < world - 1:
                self._listener.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = self._listener.accept()
                except socket.timeout:
                    raise PeerLost('only {} of {} ranks joined the job within {} s'.format(
                        len(self._peers) + 1, world, timeout))
                conn.settimeout(10.0)
                try:
                    hello = _recv_exact(conn, 4 + 16)
                    peer, pworld = struct.unpack('<QQ', hello[4:])
                    if hello[:4] != _MAGIC or pworld != world or not 0 < peer < world or peer in self._peers:
                        raise ValueError
                    challenge = os.urandom(16)
                    conn.sendall(challenge)
                    # the peer proves that it holds the secret (it read the file / knows the key) ...
                    if not hmac.compare_digest(_recv_exact(conn, 32), _mac(secret, b'peer', challenge, hello[4:])):
                        raise ValueError
                    conn.sendall(_mac(secret, b'hub', challenge))       # ... and so does the hub
                except (ValueError, PeerLost, OSError, struct.error):
                    conn.close()                # not one of ours (stale client, port scanner)
                    continue
                conn.settimeout(None)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self._peers[peer] = conn
        else:
            while self._hub is None:
                if time.monotonic() > deadline:
                    raise PeerLost('rank 0 did not {} within {} s'.format(
                        'accept on control port {}'.format(port) if port else 'publish its control port ({})'.format(self._path), timeout))
                conn = None
                try:
                    if port:
                        to, secret = port, digest
                    else:
                        with open(self._path, 'rb') as f:
                            blob = f.read()
                        (to,), secret = struct.unpack('<I', blob[:4]), blob[4:36]
                        if len(secret) != 32:
                            raise ValueError
                    conn = socket.create_connection((addr, to), timeout=5.0)
                    conn.sendall(_MAGIC + who)
                    challenge = _recv_exact(conn, 16)
                    conn.sendall(_mac(secret, b'peer', challenge, who))
                    if not hmac.compare_digest(_recv_exact(conn, 32), _mac(secret, b'hub', challenge)):
                        raise ValueError
                    conn.settimeout(None)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._hub = conn
                except (OSError, ValueError, struct.error, PeerLost):
                    if conn is not None:
                        conn.close()
                    time.sleep(0.05)            # file not there yet, or left over from an earlier job
        if rank == 0 and self._path:
            try:
                os.unlink(self._path)           # everyone is connected; the file has done its job
            except OSError:
                pass

    def allgather(self, payload):
        """payload (bytes) of every rank, in rank order.  Collective: every rank calls it in the same order."""
        if self.world == 1:
            return [payload]
        try:
            if self.rank == 0:
                parts = [payload] + [_recv_frame(self._peers[r]) for r in range(1, self.world)]
                joined = b''.join(struct.pack('<Q', len(p)) + p for p in parts)
                for r in range(1, self.world):
                    _send_frame(self._peers[r], joined)
                return parts
            _send_frame(self._hub, payload)
            joined = _recv_frame(self._hub)
        except OSError as e:
            raise PeerLost('control connection failed: {}'.format(e))
        parts, at = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from('<Q', joined, at)
            parts.append(joined[at + 8:at + 8 + n])
            at += 8 + n
        return parts

    def close(self):
        for s in list(self._peers.values()) + [self._hub, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._hub, self._listener = {}, None, None


def _dumps(obj):
    """Control-plane values: None, bools, ints of any size, floats, strings and lists of those -- as JSON."""
    return json.dumps(obj).encode()


def _loads(raw):
    return json.loads(raw.decode())


class Comm:
    """World of size 1 unless initialised from the launcher's environment (RANK / WORLD_SIZE / MASTER_*)."""

    def __init__(self):
        self.rank = 0
        self.world = 1
        self.local_rank = 0
        self.backend = None         # 'rccl' | 'socket' (data plane), None for a world of 1
        self._boot = None
        self._engine = None         # engine whose handle owns the RCCL communicator

    @classmethod
    def from_env(cls, backend=None):
        c = cls()
        world = max(1, int(os.environ.get('WORLD_SIZE', '1')))
        if world == 1 and not os.environ.get('BSX_FORCE_DIST'):     # (knob: tests run the full machinery with one rank)
            return c
        c.rank = int(os.environ.get('RANK', '0'))
        c.world = world
        c.local_rank = int(os.environ.get('LOCAL_RANK', c.rank))
        c.backend = backend or os.environ.get('BSX_DIST_BACKEND') or 'rccl'
        if c.backend not in ('rccl', 'socket'):
            raise ValueError('BSX_DIST_BACKEND must be "rccl" or "socket", not "{}"'.format(c.backend))
        addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
        key = os.environ.get('BSX_RDZV_KEY') or '{}|{}|{}|{}'.format(
            addr, os.environ.get('MASTER_PORT', ''), os.environ.get('TORCHELASTIC_RUN_ID', ''),
            os.environ.get('TORCHELASTIC_RESTART_COUNT', ''))
        port = int(os.environ.get('BSX_CONTROL_PORT', '0')) or None
        if port and not os.environ.get('BSX_RDZV_KEY'):
            raise ValueError('BSX_CONTROL_PORT needs BSX_RDZV_KEY (the secret the ranks share) as well')
        c._boot = Bootstrap(c.rank, world, addr, key, float(os.environ.get('BSX_RDZV_TIMEOUT', '300')), port)
        return c

    @property
    def active(self):
        """True when collectives really run (a job of several ranks, or the forced one-rank test mode)."""
        return self._boot is not None

    def attach_engine(self, engine):
        """Build this job's RCCL communicator on the engine's device (collective).  No-op for a world of 1
        and for the socket backend."""
        if self._boot is None or self.backend != 'rccl' or self._engine is engine:
            return
        # every rank asks its library for an id first (only rank 0's is used): a rank whose librccl does not
        # load says so here, over the control plane, instead of leaving the others inside ncclCommInitRank
        uid, err = b'', None
        try:
            uid = engine.comm_unique_id()
        except Exception as e:      # noqa: BLE001
            err = e
        flags = self._boot.allgather(b'' if err else uid)
        if err:
            raise err
        bad = [r for r, f in enumerate(flags) if not f]
        if bad:
            raise PeerLost('RCCL is not usable on rank(s) {}'.format(bad))
        engine.comm_init(flags[0], self.rank, self.world)
        self._engine = engine

    # -- control plane ------------------------------------------------------------------------
    def barrier(self):
        if self._boot is not None:
            self._boot.allgather(b'')

    def broadcast_obj(self, obj, root=0):
        """Python object of `root` on every rank (control plane: small things such as the output directory)."""
        if self._boot is None:
            return obj
        return _loads(self._boot.allgather(_dumps(obj) if self.rank == root else b'')[root])

    def allgather_obj(self, obj):
        if self._boot is None:
            return [obj]
        return [_loads(p) for p in self._boot.allgather(_dumps(obj))]

    def allreduce_max(self, value):
        return max(float(v) for v in self.allgather_obj(float(value)))

    def allreduce_sum_int(self, values):
        """Element-wise sum of a list of python ints over ranks (exact, any size)."""
        rows = self.allgather_obj([int(v) for v in values])
        return [sum(col) for col in zip(*rows)]

    def all_ok(self, ok):
        """True iff every rank passes True.  Lets a rank-local failure end the whole job together."""
        return all(self.allgather_obj(bool(ok)))

    # -- data plane ---------------------------------------------------------------------------
    def _allgather_bytes(self, raw, per_rank):
        """raw: uint8 array of exactly per_rank bytes -> (world, per_rank) uint8 array."""
        if self.backend == 'rccl':
            if self._engine is None:
                raise RuntimeError('no RCCL communicator: call Comm.attach_engine(engine) first '
                                   '(BSX_DIST_BACKEND=socket routes the merge over TCP for CPU-only tests)')
            return self._engine.comm_allgather(raw, self.world).reshape(self.world, per_rank)
        parts = self._boot.allgather(raw.tobytes())
        return np.frombuffer(b''.join(parts), np.uint8).reshape(self.world, per_rank)

    def allgather_records(self, records, slots=None):
        """
        All-gather a 1-D numpy structured array (e.g. _lib.ATTR_REC2) -> list of per-rank arrays, in ONE data-plane
        collective with nothing on the control plane: every rank sends a 16-byte header (its record count) followed
        by `slots` record slots (default 1024, the size is the same on every rank by construction).  Only if some
        rank had more records than slots -- every rank sees that in the headers -- the collective is repeated with
        the largest count.
        """
        if self._boot is None:
            return [records]
        item = records.dtype.itemsize
        raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
        cap = max(1, int(slots or 1024))
        for _ in range(2):
            buf = np.zeros(16 + cap * item, np.uint8)
            buf[:16] = np.frombuffer(struct.pack('<QQ', len(records), item), np.uint8)
            buf[16:16 + min(raw.size, cap * item)] = raw[:cap * item]
            host = self._allgather_bytes(buf, buf.size)
            counts = []
            for r in range(self.world):
                n, it = struct.unpack('<QQ', host[r, :16].tobytes())
                if it != item:
                    raise RuntimeError('rank {} sent records of {} bytes, expected {}'.format(r, it, item))
                counts.append(n)
            if max(counts) <= cap:
                return [np.frombuffer(host[r, 16:16 + c * item].tobytes(), dtype=records.dtype) for r, c in enumerate(counts)]
            cap = max(counts)
        raise RuntimeError('record counts changed between two collectives')

    def gather_concat(self, array):
        """All-gather a numpy array along axis 0 (rank order = index order for range partitions)."""
        if self._boot is None:
            return array
        flat = np.ascontiguousarray(array)
        row = flat.dtype.itemsize * int(np.prod(flat.shape[1:], dtype=np.int64))
        as_rows = flat.view(np.uint8).reshape(flat.shape[0], row) if flat.shape[0] else np.zeros((0, row), np.uint8)
        rec = np.dtype([('b', 'u1', (row,))])
        parts = self.allgather_records(as_rows.view(rec).reshape(-1))
        joined = np.concatenate([p.view(np.uint8).reshape(-1, row) for p in parts])
        return joined.view(flat.dtype).reshape((-1,) + flat.shape[1:])

    def shutdown(self):
        if self._engine is not None:
            try:
                self._engine.comm_destroy()
            except Exception:   # noqa: BLE001  (engine already closed)
                pass
            self._engine = None
        if self._boot is not None:
            self._boot.close()
            self._boot = None

    def abort(self):
        """A rank that failed: the control connection goes FIRST -- that is what wakes the peers (their next control
        operation raises PeerLost) -- and only then the communicator, whose teardown may block behind a collective
        that will never complete."""
        if self._boot is not None:
            self._boot.close()
            self._boot = None
        if self._engine is not None:
            try:
                self._engine.comm_destroy()
            except Exception:   # noqa: BLE001
                pass
            self._engine = None


def partition(n_problems, world, rank):
    """Contiguous index range of `rank`: [r*N/G, (r+1)*N/G)."""
    lo = (n_problems * rank) // world
    hi = (n_problems * (rank + 1)) // world
    return lo, hi - lo


ATTR_REC = _lib.ATTR_REC
