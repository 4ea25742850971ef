#!/usr/bin/env python
"""
Writes boolsi_amd/data/config4.yaml: BASELINE config 4 (n = 64, K = 2, nodes 0-27 'any', three '0?'
knock-outs, target on 8 nodes).  The target values are read off a state that one trajectory of
the network really reaches (step 6 of problem 12345 of the un-knocked-out variant, computed with
the CPU oracle), so that `target` has something to find.  Build-container tool, run once.
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from boolsi_amd import synth  # noqa: E402
from boolsi_amd.compile import compile_problem  # noqa: E402
from boolsi_amd.constants import Mode  # noqa: E402
from boolsi_amd.input import parse_input_text  # noqa: E402
from oracle.cpu_oracle import Oracle  # noqa: E402

n = 64
bits = synth.seeded_bits(n, 640)
initial = {i: str(bits[i]) for i in range(28, n)}
rng = random.Random(641)
knock = sorted(rng.sample(range(n), 3))
tnodes = sorted(rng.sample(range(n), 8))
base = synth.network_yaml(n, 2, 64, initial=initial, fixed={i: '0?' for i in knock})
cfg = parse_input_text(base, 1024, Mode.SIMULATE)
net, space = compile_problem(cfg)
traj = Oracle(net, space).trajectory(12345, 6)
state = int(traj[6][0])
target = {i: 'any' for i in range(n)}
for node in tnodes:
    target[node] = str((state >> node) & 1)
text = synth.network_yaml(n, 2, 64, initial=initial, fixed={i: '0?' for i in knock}, target=target)
path = os.path.join(ROOT, 'boolsi_amd', 'data', 'config4.yaml')
with open(path, 'w') as f:
    f.write(text)
print('wrote', path, 'target nodes', tnodes)
