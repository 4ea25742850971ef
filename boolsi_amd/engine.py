"""
Python face of the gfx950 engine: one `Engine` = one bsx handle = one GPU.

The methods are thin: they marshal numpy buffers through the C-ABI (include/bsx.h) and hand back
numpy records.  All arithmetic of the hot path happens in the HIP kernels; nothing here steps a
network.  Mode drivers that mirror the reference's `*_master` functions live in attract.py /
simulate.py / target.py.
"""
import ctypes as C
from dataclasses import dataclass
from math import inf

import numpy as np

from . import _lib
from ._lib import EngineError, EngineUnavailable, Index, Stats, T_INF, ptr  # noqa: F401

_M64 = 2 ** 64 - 1


def _cap(v):
    return T_INF if v is None or v == inf else int(v)


@dataclass
class AttractResult:
    table: np.ndarray            # _lib.ATTR_REC, unordered
    n_no_attractor: int
    per_problem: np.ndarray      # _lib.PROBLEM_REC or None
    stats: dict


def key_to_int(key_words):
    v = 0
    for w, x in enumerate(np.asarray(key_words, dtype=np.uint64).tolist()):
        v |= int(x) << (64 * w)
    return v


class Engine:
    def __init__(self, device=0):
        self._lib = _lib.load()
        handle = C.c_void_p()
        rc = self._lib.bsx_create(C.byref(handle), int(device))
        if rc != 0:
            msg = self._lib.bsx_last_error(None).decode()
            raise EngineUnavailable('bsx_create(device={}) failed: {} ({})'.format(
                device, msg, self._lib.bsx_status_string(rc).decode()))
        self._h = handle
        self.device = device
        self.net = None
        self.space = None
        self._keep = []

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None):
            self._lib.bsx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self._lib.bsx_last_error(self._h).decode() or
                              self._lib.bsx_status_string(rc).decode())

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_uint32()
        mem = C.c_uint64()
        self._check(self._lib.bsx_device_info(self._h, name, 256, C.byref(cus), C.byref(mem)))
        return {'name': name.value.decode(), 'compute_units': cus.value, 'global_mem_bytes': mem.value}

    def network_info(self):
        """How the network was lowered: names the kernel instantiations a run launches (bsx_network_info)."""
        nw, k, lm = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self._lib.bsx_network_info(self._h, C.byref(nw), C.byref(k), C.byref(lm)))
        return {'state_words32': nw.value, 'mux_slots': k.value, 'lut_mode': lm.value}

    def synchronize(self):
        self._check(self._lib.bsx_synchronize(self._h))

    # -- multi-GPU merge step (RCCL through the C-ABI; driven by boolsi_amd.dist.Comm) -----------
    def comm_unique_id(self):
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        self._check(self._lib.bsx_comm_unique_id(self._h, buf, _lib.COMM_ID_BYTES))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        self._check(self._lib.bsx_comm_init(self._h, unique_id, len(unique_id), int(rank), int(world)))

    def comm_allgather(self, send, world):
        """send: contiguous uint8 array, same size on every rank -> uint8 array of world * size bytes."""
        send = np.ascontiguousarray(send, np.uint8)
        recv = np.zeros(world * send.size, np.uint8)
        self._check(self._lib.bsx_comm_allgather(self._h, ptr(send), send.size, ptr(recv)))
        return recv

    def comm_destroy(self):
        if getattr(self, '_h', None):
            self._check(self._lib.bsx_comm_destroy(self._h))

    # -- problem definition -----------------------------------------------------------------
    def set_problem(self, net, space):
        """net, space: boolsi_amd.compile.CompiledNetwork / CompiledSpace."""
        a = [np.ascontiguousarray(x) for x in (net.pred_offsets, net.pred_idx, net.tt_word_offsets,
                                               net.tt_words)]
        self._check(self._lib.bsx_set_network(self._h, net.n_nodes, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3])))
        self.net = net
        self._keep_net = a
        self.set_space(space)

    def set_space(self, space):
        b = [np.ascontiguousarray(space.origin_state, np.uint64),
             np.ascontiguousarray(space.any_nodes, np.uint32),
             np.ascontiguousarray(space.fixed, np.uint32), np.ascontiguousarray(space.fixed_var, np.uint32),
             np.ascontiguousarray(space.sched, np.uint32), np.ascontiguousarray(space.pert_var, np.uint32)]
        self._check(self._lib.bsx_set_problem_space(
            self._h, ptr(b[0]), ptr(b[1]), len(b[1]), ptr(b[2]), len(b[2]), ptr(b[3]), len(b[3]),
            ptr(b[4]), len(b[4]), ptr(b[5]), len(b[5])))
        self.space = space
        self._keep = b

    def states_from(self, state_code, n_steps):
        """
        s, f(s), ..., f^n_steps(s) for the state with code `state_code`, under the origin problem's
        constant fixed nodes and without perturbations -> (n_steps + 1, W) uint64 array.
        Used to list the states of an attractor from its key (reference attract.py:22-25 keeps
        them in memory instead).
        """
        from .compile import CompiledSpace, code_to_words
        saved = self.space
        empty2 = np.zeros((0, 2), np.uint32)
        empty3 = np.zeros((0, 3), np.uint32)
        tmp = CompiledSpace(n_nodes=self.net.n_nodes, origin_state=code_to_words(state_code, self.net.n_words),
                            any_nodes=np.zeros(0, np.uint32), fixed=saved.fixed, fixed_var=empty2, sched=empty3,
                            pert_var=empty3, n_problems=1, radices=[])
        self.set_space(tmp)
        try:
            trajs, _ = self.trajectories(0, [0], [n_steps])
        finally:
            self.set_space(saved)
        return trajs[0]

    def index(self, i):
        """python int problem index -> bsx_index (split at the initial-state digits)."""
        n_any = len(self.space.any_nodes)
        low = i & ((1 << n_any) - 1)
        ix = Index()
        for w in range(_lib.MAX_WORDS):
            ix.init_digits[w] = (low >> (64 * w)) & _M64
        variant = i >> n_any
        if variant > _M64:
            raise ValueError('variant part of the problem index exceeds 64 bits')
        ix.variant = variant
        return ix

    # -- runs ---------------------------------------------------------------------------------
    def attract_fgraph(self, first, count, max_t=inf, max_len=inf, cap=65536):
        """attract through the functional graph (spaces with n <= 32 nodes, all 'any'); same result as attract."""
        table = np.zeros(cap, _lib.ATTR_REC)
        n_out, none, st = C.c_uint32(), C.c_uint64(), Stats()
        self._check(self._lib.bsx_run_attract_fgraph(self._h, C.byref(self.index(first)), count, _cap(max_t), _cap(max_len),
                                                     ptr(table), cap, C.byref(n_out), C.byref(none), C.byref(st)))
        return AttractResult(table[:n_out.value].copy(), none.value, None, st.as_dict())

    def attract(self, first, count, max_t=inf, max_len=inf, per_problem=False, cap=65536):
        # the output table is reused between calls (the library fills table[0..n_out), which is copied out)
        table = getattr(self, '_attr_table', None)
        if table is None or len(table) < cap:
            table = self._attr_table = np.zeros(cap, _lib.ATTR_REC)
        pp = np.zeros(count, _lib.PROBLEM_REC) if per_problem else None
        n_out = C.c_uint32()
        none = C.c_uint64()
        st = Stats()
        rc = self._lib.bsx_run_attract(self._h, C.byref(self.index(first)), count, _cap(max_t), _cap(max_len),
                                       ptr(table), cap, C.byref(n_out), C.byref(none), ptr(pp) if per_problem else None,
                                       C.byref(st))
        self._check(rc)
        return AttractResult(table[:n_out.value].copy(), none.value, pp, st.as_dict())

    def attract2(self, first, count, max_t=inf, max_len=inf, cap=65536):
        """attract over [first, first + count) for flat problem indices / counts of up to 128 bits (bsx_run_attract2):
        one call for the whole range however large -> AttractResult with a _lib.ATTR_REC2 table (wide sums)."""
        table = getattr(self, '_attr_table2', None)
        if table is None or len(table) < cap:
            table = self._attr_table2 = np.zeros(cap, _lib.ATTR_REC2)
        n_out, none, st = C.c_uint32(), _lib.U128(), _lib.Stats2()
        self._check(self._lib.bsx_run_attract2(self._h, _lib.U128.of(first), _lib.U128.of(count), _cap(max_t), _cap(max_len),
                                               ptr(table), cap, C.byref(n_out), C.byref(none), C.byref(st)))
        return AttractResult(table[:n_out.value].copy(), int(none), None, st.as_dict())

    def target(self, first, count, max_t, mask_words, code_words, cap=None):
        cap = count if cap is None else cap
        hits = np.zeros(max(cap, 1), _lib.HIT)
        m = np.ascontiguousarray(mask_words, np.uint64)
        c = np.ascontiguousarray(code_words, np.uint64)
        n_hits = C.c_uint64()
        st = Stats()
        self._check(self._lib.bsx_run_target(self._h, C.byref(self.index(first)), count, _cap(max_t), ptr(m), ptr(c),
                                             ptr(hits), cap, C.byref(n_hits), C.byref(st)))
        return hits[:n_hits.value], st.as_dict()      # ascending offset order (ABI contract)

    def target_summary(self, first, count, max_t, mask_words, code_words, hist_bins=0, cap=0):
        """Hits counted on the device: -> (n_hits, histogram of first-hit times (last bin = that time or later),
        the first min(n_hits, cap) hits in index order, stats).  Nothing per problem crosses PCIe."""
        hist = np.zeros(max(hist_bins, 1), np.uint64)
        hits = np.zeros(max(cap, 1), _lib.HIT)
        m = np.ascontiguousarray(mask_words, np.uint64)
        c = np.ascontiguousarray(code_words, np.uint64)
        n_hits, n_listed = C.c_uint64(), C.c_uint64()
        st = Stats()
        self._check(self._lib.bsx_run_target_summary(
            self._h, C.byref(self.index(first)), count, _cap(max_t), ptr(m), ptr(c), ptr(hist) if hist_bins else None,
            hist_bins, ptr(hits) if cap else None, cap, C.byref(n_hits), C.byref(n_listed), C.byref(st)))
        return n_hits.value, hist[:hist_bins], hits[:n_listed.value], st.as_dict()

    def simulate(self, first, count, max_t, trajectories=True, final=True, digest=True):
        W = self.net.n_words
        traj = np.zeros((count, max_t + 1, W), np.uint64) if trajectories else None
        fin = np.zeros((count, W), np.uint64) if final else None
        dig = np.zeros(count, np.uint64) if digest else None
        st = Stats()
        self._check(self._lib.bsx_run_simulate(self._h, C.byref(self.index(first)), count, int(max_t),
                                               ptr(traj) if trajectories else None, ptr(fin) if final else None,
                                               ptr(dig) if digest else None, C.byref(st)))
        return traj, fin, dig, st.as_dict()

    def trajectories(self, first, offsets, t_len):
        """States s(0..t_len[q]) of problems first + offsets[q] -> list of (t_len[q] + 1, W) arrays."""
        W = self.net.n_words
        offsets = np.ascontiguousarray(offsets, np.uint64)
        t_len = np.ascontiguousarray(t_len, np.uint64)
        sizes = (t_len + np.uint64(1)) * np.uint64(W)
        out_off = np.zeros(len(offsets), np.uint64)
        if len(offsets) > 1:
            out_off[1:] = np.cumsum(sizes[:-1])
        out = np.zeros(int(sizes.sum()), np.uint64)
        st = Stats()
        self._check(self._lib.bsx_run_trajectories(self._h, C.byref(self.index(first)), ptr(offsets), ptr(t_len),
                                                   len(offsets), ptr(out), ptr(out_off), C.byref(st)))
        return [out[int(o):int(o) + int(s)].reshape(-1, W) for o, s in zip(out_off, sizes)], st.as_dict()
