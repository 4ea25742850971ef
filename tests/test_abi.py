"""
C-ABI surface (no GPU needed): the shared library loads and exports exactly what include/bsx.h
declares; without a GPU the product path fails loudly instead of falling back to anything.
"""
import os
import re

import pytest

from boolsi_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'bsx.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bsx_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_functions()
    assert declared == sorted(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(_lib.Index) == 40
    assert C.sizeof(_lib.Stats) == 48
    assert _lib.ATTR_REC.itemsize == 72
    assert _lib.PROBLEM_REC.itemsize == 56
    assert _lib.HIT.itemsize == 16
    assert C.sizeof(_lib.U128) == 16 and C.sizeof(_lib.Stats2) == 104
    assert _lib.ATTR_REC2.itemsize == 32 + 8 + 16 + 24 + 32


def test_no_gpu_means_loud_failure_not_fallback():
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.bsx_create(C.byref(h), 0)
    if rc == 0:                       # running on a GPU box: nothing to assert here
        lib.bsx_destroy(h)
        pytest.skip('a GPU is present')
    assert rc < 0 and not h.value
    assert lib.bsx_last_error(None)
    from boolsi_amd.engine import Engine, EngineUnavailable
    with pytest.raises(EngineUnavailable):
        Engine(0)


def test_product_package_uses_no_tensor_framework():
    """north_star: ctypes + HIP + RCCL only.  Nothing under boolsi_amd/ may import torch (the multi-GPU merge is
    ncclAllGather behind the C-ABI, the rendezvous a TCP socket)."""
    import re
    pkg = os.path.join(ROOT, 'boolsi_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(import|from)\s+torch', text, flags=re.M), f
    import subprocess
    import sys
    code = 'import sys; import boolsi_amd.cli, boolsi_amd.dist, boolsi_amd.attract, boolsi_amd.engine; assert "torch" not in sys.modules'
    assert subprocess.run([sys.executable, '-c', code], cwd=ROOT).returncode == 0


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, 'boolsi_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'cpu_oracle' not in text and 'liboracle' not in text and 'oracle/' not in text, f
