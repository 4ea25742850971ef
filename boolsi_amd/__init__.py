"""
boolsi_amd -- MI355X-native engine for BoolSi's simulate / attract / target state-update path.

Host side in Python (YAML front end, problem enumeration, result tables, CSV, CLI); compute in
hand-written HIP for gfx950 behind the C-ABI of include/bsx.h, bound with ctypes
(boolsi_amd/_lib.py).  There is no CPU fallback: without the built library (or without a GPU)
engine calls raise.
"""
__version__ = '0.1.0'
