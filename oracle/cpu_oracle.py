"""
TEST INFRASTRUCTURE -- ctypes wrapper of the CPU oracle (oracle/bsx_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module, as the
checker / timed CPU baseline.  The product package never does (it fails loudly without the HIP
library instead of falling back).
"""
import os
import ctypes as C
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'liboracle.so')
MAX_WORDS = 4
T_INF = 2 ** 64 - 1
_M64 = 2 ** 64 - 1


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ('bsx_oracle.c', 'bsx_oracle.h')]
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])
    return _LIB_PATH


class _Net(C.Structure):
    _fields_ = [('n_nodes', C.c_uint32), ('pred_offsets', C.c_void_p), ('pred_idx', C.c_void_p),
                ('tt_word_offsets', C.c_void_p), ('tt_words', C.c_void_p)]


class _Space(C.Structure):
    _fields_ = [('origin_state', C.c_void_p),
                ('any_nodes', C.c_void_p), ('n_any', C.c_uint32),
                ('fixed', C.c_void_p), ('n_fixed', C.c_uint32),
                ('fixed_var', C.c_void_p), ('n_fv', C.c_uint32),
                ('sched', C.c_void_p), ('n_sched', C.c_uint32),
                ('pert_var', C.c_void_p), ('n_pv', C.c_uint32)]


ATTR_RESULT = np.dtype([('key', '<u8', (MAX_WORDS,)), ('length', '<u8'), ('trajectory_l', '<u8'),
                        ('t_stop', '<u8'), ('found', '<u4'), ('pad', '<u4')])
ATTR_AGG = np.dtype([('key', '<u8', (MAX_WORDS,)), ('length', '<u8'), ('count', '<u8'),
                     ('sum_l', '<u8'), ('sum_l2_lo', '<u8'), ('sum_l2_hi', '<u8')])
TARGET_RESULT = np.dtype([('t_stop', '<u8'), ('reached', '<u4'), ('pad', '<u4')])



class _Index(C.Structure):
    _fields_ = [('init_digits', C.c_uint64 * MAX_WORDS), ('variant', C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _t(v):
    return T_INF if v is None or v == float('inf') else int(v)


class Oracle:
    """Holds the flat arrays of one compiled problem (boolsi_amd.compile) alive for the C side."""

    def __init__(self, net, space):
        self.net, self.space = net, space
        self.W = net.n_words
        self._keep = [np.ascontiguousarray(a) for a in (
            net.pred_offsets, net.pred_idx, net.tt_word_offsets, net.tt_words, space.origin_state,
            space.any_nodes, space.fixed, space.fixed_var, space.sched, space.pert_var)]
        k = self._keep
        self.cnet = _Net(net.n_nodes, _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(k[3]))
        self.cspace = _Space(_ptr(k[4]), _ptr(k[5]), len(k[5]), _ptr(k[6]), len(k[6]), _ptr(k[7]), len(k[7]),
                             _ptr(k[8]), len(k[8]), _ptr(k[9]), len(k[9]))

    def index(self, i):
        """python int problem index -> orc_index (split at the initial-state digits)."""
        n_any = len(self.space.any_nodes)
        low = i & ((1 << n_any) - 1)
        ix = _Index()
        for w in range(MAX_WORDS):
            ix.init_digits[w] = (low >> (64 * w)) & _M64
        variant = i >> n_any
        if variant > _M64:
            raise ValueError('variant part of the problem index exceeds 64 bits')
        ix.variant = variant
        return ix

    def step(self, words):
        s = np.ascontiguousarray(words, dtype=np.uint64)
        out = np.zeros(self.W, dtype=np.uint64)
        lib().orc_step(C.byref(self.cnet), _ptr(s), _ptr(out))
        return out

    def problem(self, index):
        init = np.zeros(self.W, np.uint64)
        fm = np.zeros(self.W, np.uint64)
        fv = np.zeros(self.W, np.uint64)
        pert = np.zeros((len(self.space.sched) + len(self.space.pert_var) + 1, 3), np.uint32)
        n_pert = C.c_uint32(0)
        rc = lib().orc_problem_from_index(C.byref(self.cnet), C.byref(self.cspace),
                                          C.byref(self.index(index)),
                                          _ptr(init), _ptr(fm), _ptr(fv), _ptr(pert), C.byref(n_pert))
        return rc, init, fm, fv, pert[:n_pert.value]

    def attract(self, first, count, max_t=None, max_len=None, storing_all_states=True, cap=65536,
                per_problem=True, n_threads=1):
        pp = np.zeros(count, ATTR_RESULT) if per_problem else None
        table = np.zeros(cap, ATTR_AGG)
        n_out = C.c_uint32(0)
        none = C.c_uint64(0)
        steps = C.c_uint64(0)
        rc = lib().orc_run_attract(
            C.byref(self.cnet), C.byref(self.cspace), C.byref(self.index(first)),
            C.c_uint64(count), C.c_uint64(_t(max_t)), C.c_uint64(_t(max_len)),
            C.c_int(1 if storing_all_states else 0), _ptr(pp) if per_problem else None, _ptr(table),
            C.c_uint32(cap), C.byref(n_out), C.byref(none), C.byref(steps), C.c_int(n_threads))
        if rc != 0:
            raise RuntimeError('oracle attractor table overflow')
        return pp, table[:n_out.value], none.value, steps.value

    def target(self, first, count, max_t, mask_words, code_words, n_threads=1):
        pp = np.zeros(count, TARGET_RESULT)
        steps = C.c_uint64(0)
        m = np.ascontiguousarray(mask_words, np.uint64)
        c = np.ascontiguousarray(code_words, np.uint64)
        lib().orc_run_target(C.byref(self.cnet), C.byref(self.cspace), C.byref(self.index(first)),
                             C.c_uint64(count), C.c_uint64(_t(max_t)), _ptr(m), _ptr(c),
                             _ptr(pp), C.byref(steps), C.c_int(n_threads))
        return pp, steps.value

    def simulate(self, first, count, max_t, want_traj=True, n_threads=1):
        traj = np.zeros((count, max_t + 1, self.W), np.uint64) if want_traj else None
        final = np.zeros((count, self.W), np.uint64)
        digest = np.zeros(count, np.uint64)
        steps = C.c_uint64(0)
        lib().orc_run_simulate(C.byref(self.cnet), C.byref(self.cspace), C.byref(self.index(first)),
                               C.c_uint64(count), C.c_uint64(max_t),
                               _ptr(traj) if want_traj else None, _ptr(final), _ptr(digest),
                               C.byref(steps), C.c_int(n_threads))
        return traj, final, digest, steps.value

    def trajectory(self, index, t_len):
        traj = np.zeros((t_len + 1, self.W), np.uint64)
        lib().orc_trajectory(C.byref(self.cnet), C.byref(self.cspace), C.byref(self.index(index)),
                             C.c_uint64(t_len), _ptr(traj))
        return traj


def key_int(key_words):
    v = 0
    for w, x in enumerate(np.asarray(key_words, dtype=np.uint64).tolist()):
        v |= int(x) << (64 * w)
    return v
