import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text
eng=Engine(0)
cfg=parse_input_text(synth.north_star_yaml(),4096,Mode.ATTRACT); net,space=compile_problem(cfg); eng.set_problem(net,space)
first=0x0123456789ABCDEF & ~((1<<40)-1)
for cap in (65536, 1<<20, 1<<20, 65536):
    eng.attract(first, 1<<22, 4096, cap=cap)
    t0=time.perf_counter(); r=eng.attract(first, 1<<40, 4096, cap=cap); dt=time.perf_counter()-t0
    print('cap',cap,'wall %.2f ms'%(dt*1e3),'kernel %.2f total_ms %.2f launches %d'%(r.stats['kernel_ms'],r.stats['total_ms'],r.stats['kernel_launches']))
