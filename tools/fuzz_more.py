#!/usr/bin/env python
"""More seeds of the differential fuzz tests of tests/test_gpu_cubes.py than the suite runs (GPU box):
    python tools/fuzz_more.py [first_seed] [n_seeds]
Every seed runs the cube / deep-level / sub-block fuzz bodies (cascade against the plain enumeration on a random space)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_cubes as T  # noqa: E402
from boolsi_amd.engine import Engine  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bodies = [T.test_cubes_equal_plain_enumeration_on_random_spaces, T.test_deep_levels_equal_plain_enumeration_on_random_spaces,
          T.test_sub_blocks_equal_plain_enumeration_on_random_spaces]
t0 = time.time()
bad = 0
for seed in range(first, first + count):
    for body in bodies:
        eng = Engine(0)
        try:
            body(eng, seed)
        except AssertionError as e:
            bad += 1
            print('FAILED', body.__name__, seed, str(e)[:300], flush=True)
        finally:
            eng.close()
            for k in ('BSX_CUBES', 'BSX_CUBE_DEPTH', 'BSX_CUBE_NEAR_CAP', 'BSX_CUBE_SPLIT', 'BSX_CUBE_STREAMS'):
                os.environ.pop(k, None)
    if (seed - first) % 20 == 19:
        print('seed', seed, 'done, {:.0f} s, {} failures'.format(time.time() - t0, bad), flush=True)
print('{} seeds x {} bodies, {} failures, {:.0f} s'.format(count, len(bodies), bad, time.time() - t0))
sys.exit(1 if bad else 0)
