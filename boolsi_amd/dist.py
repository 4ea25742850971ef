"""
Multi-GPU layer: one process per GPU, problems range-partitioned, one all-gather at the end.

Replaces the reference's mpi4py master/worker task farm (boolsi/mpi.py:193-340, 379-495): the
simulation problems are independent, so rank r simply takes the contiguous index range
[r*N/G, (r+1)*N/G) and the only communication is the merge of the per-rank attractor tables --
`torch.distributed.all_gather` of fixed-size integer records (backend "nccl" = RCCL over xGMI on
GPUs, "gloo" on CPU for the tests).  torch is used for this control/collective plumbing only and is
imported lazily, so single-GPU runs never load it.
"""
import os

import numpy as np

from . import _lib


class Comm:
    """World of size 1 unless initialised from the torchrun environment."""

    def __init__(self):
        self.rank = 0
        self.world = 1
        self.local_rank = 0
        self.backend = None
        self._dist = None
        self._torch = None

    @classmethod
    def from_env(cls, backend=None):
        c = cls()
        world = int(os.environ.get('WORLD_SIZE', '1'))
        if world <= 1 and not os.environ.get('BSX_FORCE_DIST'):
            return c
        import torch
        import torch.distributed as dist
        c.rank = int(os.environ.get('RANK', '0'))
        c.world = world
        c.local_rank = int(os.environ.get('LOCAL_RANK', c.rank))
        c.backend = backend or os.environ.get('BSX_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if c.backend == 'nccl':
            torch.cuda.set_device(c.local_rank)
        if not dist.is_initialized():
            dist.init_process_group(backend=c.backend, rank=c.rank, world_size=world)
        c._dist, c._torch = dist, torch
        return c

    # -- helpers ----------------------------------------------------------------------------
    def _device(self):
        return self._torch.device('cuda', self.local_rank) if self.backend == 'nccl' else self._torch.device('cpu')

    def barrier(self):
        if self.world > 1:
            self._dist.barrier()
            if self.backend == 'nccl':
                self._torch.cuda.synchronize()

    def allreduce_max(self, value):
        if self.world == 1:
            return float(value)
        t = self._torch.tensor([float(value)], dtype=self._torch.float64, device=self._device())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def allreduce_sum_int(self, values):
        """Element-wise sum of a list of non-negative python ints (< 2^62 each) over ranks."""
        if self.world == 1:
            return [int(v) for v in values]
        t = self._torch.tensor([int(v) for v in values], dtype=self._torch.int64, device=self._device())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    def allgather_records(self, records, expect=64):
        """
        All-gather a 1-D numpy structured array (e.g. _lib.ATTR_REC) -> list of per-rank arrays.
        One collective when every rank has at most `expect` records (the usual case: a handful of
        attractors): each rank sends an 8-byte count followed by `expect` record slots.  If some rank has
        more, every rank sees that in the counts and a second collective moves the records padded to the
        common maximum.
        """
        if self.world == 1:
            return [records]
        torch, dist = self._torch, self._dist
        dev = self._device()
        item = records.dtype.itemsize
        raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)

        def exchange(cap):
            buf = np.zeros(8 + cap * item, np.uint8)
            buf[:8] = np.frombuffer(np.uint64(len(records)).tobytes(), np.uint8)
            n = min(len(records), cap) * item
            buf[8:8 + n] = raw[:n]
            mine = torch.from_numpy(buf).to(dev)
            gathered = torch.empty(self.world * buf.size, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(gathered, mine)
            host = gathered.cpu().numpy().reshape(self.world, buf.size)
            counts = [int(np.frombuffer(host[r, :8].tobytes(), np.uint64)[0]) for r in range(self.world)]
            return host, counts

        host, counts = exchange(expect)
        if max(counts) > expect:
            host, counts = exchange(max(counts))
        return [np.frombuffer(host[r, 8:8 + c * item].tobytes(), dtype=records.dtype) for r, c in enumerate(counts)]

    def gather_concat(self, array):
        """All-gather a numpy array along axis 0 (rank order = index order for range partitions)."""
        if self.world == 1:
            return array
        flat = np.ascontiguousarray(array)
        row = flat.dtype.itemsize * int(np.prod(flat.shape[1:], dtype=np.int64))
        as_rows = flat.view(np.uint8).reshape(flat.shape[0], row) if flat.shape[0] else np.zeros((0, row), np.uint8)
        rec = np.dtype([('b', 'u1', (row,))])
        parts = self.allgather_records(as_rows.view(rec).reshape(-1))
        joined = np.concatenate([p.view(np.uint8).reshape(-1, row) for p in parts])
        return joined.view(flat.dtype).reshape((-1,) + flat.shape[1:])

    def shutdown(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()


def partition(n_problems, world, rank):
    """Contiguous index range of `rank`: [r*N/G, (r+1)*N/G)."""
    lo = (n_problems * rank) // world
    hi = (n_problems * (rank + 1)) // world
    return lo, hi - lo


ATTR_REC = _lib.ATTR_REC
