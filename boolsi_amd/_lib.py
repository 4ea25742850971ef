"""
ctypes binding of the C-ABI in include/bsx.h (boolsi_amd/libbsx_hip.so).

This is the only door to compute in the package.  There is deliberately no fallback: if the
shared library is missing, or no gfx950 device is present, `load()` / `Engine()` raise
`EngineUnavailable` with the reason ("the product path must fail loudly").
"""
import os
import ctypes as C

import numpy as np

MAX_WORDS = 4
T_INF = 2 ** 64 - 1

LIB_NAME = os.environ.get('BSX_LIB', 'libbsx_hip.so')      # (BSX_LIB=libbsx_hip_diag.so: diagnostic build, tools only)
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

EXPORTS = (
    'bsx_create', 'bsx_destroy', 'bsx_last_error', 'bsx_status_string', 'bsx_device_info', 'bsx_network_info',
    'bsx_set_network', 'bsx_set_problem_space', 'bsx_run_attract', 'bsx_run_attract2', 'bsx_run_attract_fgraph', 'bsx_run_target',
    'bsx_run_target_summary', 'bsx_run_simulate', 'bsx_run_trajectories', 'bsx_synchronize',
    'bsx_comm_unique_id', 'bsx_comm_init', 'bsx_comm_allgather', 'bsx_comm_destroy',
)
COMM_ID_BYTES = 128


class EngineUnavailable(RuntimeError):
    """The HIP engine cannot be used (library not built, or no MI355X / gfx950 device)."""


class EngineError(RuntimeError):
    def __init__(self, status, message):
        super().__init__('bsx error {}: {}'.format(status, message))
        self.status = status


class Index(C.Structure):
    _fields_ = [('init_digits', C.c_uint64 * MAX_WORDS), ('variant', C.c_uint64)]


# bsx_status values the host layers react to (include/bsx.h)
ERR_UNSUPPORTED = -4
ERR_RANGE_TOO_LARGE = -9


class U128(C.Structure):
    """bsx_u128: flat problem indices and counts beyond 2^64."""
    _fields_ = [('lo', C.c_uint64), ('hi', C.c_uint64)]

    @classmethod
    def of(cls, v):
        v = int(v)
        if not 0 <= v < 1 << 128:
            raise ValueError('value does not fit 128 bits')
        return cls(v & (2 ** 64 - 1), v >> 64)

    def __int__(self):
        return self.lo | (self.hi << 64)


class Stats2(C.Structure):
    _fields_ = [('problems', U128), ('state_steps', U128), ('executed_steps', C.c_uint64),
                ('kernel_ms', C.c_double), ('total_ms', C.c_double), ('dominant_ms', C.c_double),
                ('dominant_executed_steps', C.c_uint64), ('dominant_launches', C.c_uint32),
                ('kernel_launches', C.c_uint32), ('host_syncs', C.c_uint32), ('lower_launches', C.c_uint32),
                ('lower_ms', C.c_double), ('lower_executed_steps', C.c_uint64)]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_ if name != 'pad'}
        d['problems'], d['state_steps'] = int(self.problems), int(self.state_steps)
        return d


class Stats(C.Structure):
    _fields_ = [('problems', C.c_uint64), ('state_steps', C.c_uint64), ('executed_steps', C.c_uint64),
                ('kernel_ms', C.c_double), ('total_ms', C.c_double), ('kernel_launches', C.c_uint32),
                ('pad', C.c_uint32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if name != 'pad'}


ATTR_REC = np.dtype([('key', '<u8', (MAX_WORDS,)), ('length', '<u8'), ('count', '<u8'), ('sum_l', '<u8'),
                     ('sum_l2_lo', '<u8'), ('sum_l2_hi', '<u8')])
# bsx_attr_rec2: count 128, sum_l 192, sum_l2 256 bits, little-endian 64-bit words
ATTR_REC2 = np.dtype([('key', '<u8', (MAX_WORDS,)), ('length', '<u8'), ('count', '<u8', (2,)), ('sum_l', '<u8', (3,)),
                      ('sum_l2', '<u8', (4,))])
PROBLEM_REC = np.dtype([('key', '<u8', (MAX_WORDS,)), ('length', '<u8'), ('trajectory_l', '<u8'),
                        ('found', '<u4'), ('pad', '<u4')])
HIT = np.dtype([('offset', '<u8'), ('t', '<u8')])
FIXED = np.dtype([('node', '<u4'), ('value', '<u4')])
FIXED_VAR = np.dtype([('node', '<u4'), ('range', '<u4')])
PERT = np.dtype([('t', '<u4'), ('node', '<u4'), ('value', '<u4')])
PERT_VAR = np.dtype([('t', '<u4'), ('node', '<u4'), ('range', '<u4')])

_lib = None


def load():
    """Load libbsx_hip.so and declare the prototypes.  No GPU is touched by loading."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            '{} is not built: run `python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C boolsi_amd/csrc` (needs hipcc)'.format(LIB_PATH))
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise EngineUnavailable('cannot load {}: {}'.format(LIB_PATH, e))
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    lib.bsx_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.bsx_destroy.argtypes = [vp]
    lib.bsx_last_error.argtypes = [vp]
    lib.bsx_last_error.restype = C.c_char_p
    lib.bsx_status_string.argtypes = [C.c_int]
    lib.bsx_status_string.restype = C.c_char_p
    lib.bsx_device_info.argtypes = [vp, C.c_char_p, u32, C.POINTER(u32), C.POINTER(u64)]
    lib.bsx_network_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    lib.bsx_set_network.argtypes = [vp, u32, vp, vp, vp, vp]
    lib.bsx_set_problem_space.argtypes = [vp, vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32]
    lib.bsx_run_attract.argtypes = [vp, C.POINTER(Index), u64, u64, u64, vp, u32, C.POINTER(u32),
                                    C.POINTER(u64), vp, C.POINTER(Stats)]
    lib.bsx_run_attract2.argtypes = [vp, U128, U128, u64, u64, vp, u32, C.POINTER(u32), C.POINTER(U128),
                                     C.POINTER(Stats2)]
    lib.bsx_run_attract_fgraph.argtypes = [vp, C.POINTER(Index), u64, u64, u64, vp, u32, C.POINTER(u32),
                                           C.POINTER(u64), C.POINTER(Stats)]
    lib.bsx_run_target.argtypes = [vp, C.POINTER(Index), u64, u64, vp, vp, vp, u64, C.POINTER(u64),
                                   C.POINTER(Stats)]
    lib.bsx_run_target_summary.argtypes = [vp, C.POINTER(Index), u64, u64, vp, vp, vp, u32, vp, u64, C.POINTER(u64),
                                           C.POINTER(u64), C.POINTER(Stats)]
    lib.bsx_run_simulate.argtypes = [vp, C.POINTER(Index), u64, u64, vp, vp, vp, C.POINTER(Stats)]
    lib.bsx_run_trajectories.argtypes = [vp, C.POINTER(Index), vp, vp, u64, vp, vp, C.POINTER(Stats)]
    lib.bsx_synchronize.argtypes = [vp]
    lib.bsx_comm_unique_id.argtypes = [vp, vp, u32]
    lib.bsx_comm_init.argtypes = [vp, vp, u32, C.c_int, C.c_int]
    lib.bsx_comm_allgather.argtypes = [vp, vp, u64, vp]
    lib.bsx_comm_destroy.argtypes = [vp]
    for name in EXPORTS:
        getattr(lib, name)          # AttributeError here = header and library disagree
        if name not in ('bsx_last_error', 'bsx_status_string'):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None
