import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.attract import run_attract_range
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text
eng=Engine(0)
cfg=parse_input_text(synth.north_star_yaml(),4096,Mode.ATTRACT); net,space=compile_problem(cfg); eng.set_problem(net,space)
first=0x0123456789ABCDEF & ~((1<<40)-1)
t0=time.perf_counter(); m,n,st=run_attract_range(eng, first, 1<<36, 4096); print('warm 2^36 wall %.2f ms'%((time.perf_counter()-t0)*1e3), st['kernel_ms'], st['total_ms'], st['kernel_launches'])
for i in range(3):
    t0=time.perf_counter(); m,n,st=run_attract_range(eng, first, 1<<40, 4096); print('2^40 wall %.2f ms'%((time.perf_counter()-t0)*1e3), 'kernel %.2f total_ms %.2f launches %d'%(st['kernel_ms'], st['total_ms'], st['kernel_launches']))
