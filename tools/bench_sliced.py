import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text
eng = Engine(0)
for name, text in (('pert', synth.config5_yaml(max_t=10000, n_any=26)), ('nopert', synth.network_yaml(128, 3, 128, initial={i: '0' for i in range(26, 128)}))):
    cfg = parse_input_text(text, 10000, Mode.SIMULATE)
    net, space = compile_problem(cfg)
    eng.set_problem(net, space)
    for gen in ('2', '1'):
        if gen == '1': os.environ['BSX_SLICED'] = '1'
        else: os.environ.pop('BSX_SLICED', None)
        for count in (1 << 20,):
            _, fin, _, st = eng.simulate(0, count, 10000, trajectories=False, digest=False)
            _, fin, _, st = eng.simulate(0, count, 10000, trajectories=False, digest=False)
            print(name, 'gen', gen, count, 'kernel_ms %.2f' % st['kernel_ms'])
            if gen == '2':
                _, fin, dig, st = eng.simulate(0, count, 10000, trajectories=False, digest=True)
                _, fin, dig, st = eng.simulate(0, count, 10000, trajectories=False, digest=True)
                print(name, 'gen', gen, count, 'with digests: kernel_ms %.2f' % st['kernel_ms'])
