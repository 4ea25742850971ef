"""One functional-graph sweep of BASELINE config 3 (n = 32, 2^32 states) -- the target of the HBM PMC passes:
    rocprofv3 --pmc FETCH_SIZE -d out/fetch -- python3 tools/run_fgraph.py
    rocprofv3 --pmc WRITE_SIZE -d out/write -- python3 tools/run_fgraph.py
    python3 tools/run_fgraph.py --read out        (sums the counters per kernel against the kernel-trace durations)"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if '--read' in sys.argv:
    root = sys.argv[sys.argv.index('--read') + 1]
    kib = {}
    for name in ('FETCH_SIZE', 'WRITE_SIZE'):
        for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == name and 'k_fg_' in r['Kernel_Name']:
                    k = r['Kernel_Name'].split('(')[0].replace('bsx::', '').replace('void ', '')
                    kib.setdefault(k, {}).setdefault(name, 0.0)
                    kib[k][name] += float(r['Counter_Value'])
    dur = {}
    for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_fg_' in r['Kernel_Name']:
                k = r['Kernel_Name'].split('(')[0].replace('bsx::', '').replace('void ', '')
                dur[k] = dur.get(k, 0) + int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    out = {'note': 'one functional-graph sweep of config 3 (2^32 states); FETCH_SIZE doubled (gfx950 counts half of wide reads), '
                   'KiB -> bytes; durations from a separate --kernel-trace run of the same command', 'kernels': {}}
    tot_b = tot_t = 0
    for k, v in sorted(kib.items()):
        b = (2 * v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0)) * 1024
        t = dur.get(k, 0) * 1e-9
        out['kernels'][k] = {'hbm_bytes': b, 'seconds': t, 'GBps': b / t / 1e9 if t else None}
        tot_b += b
        tot_t += t
    out['total'] = {'hbm_bytes': tot_b, 'seconds': tot_t, 'GBps': tot_b / tot_t / 1e9 if tot_t else None, 'frac_of_8TBps': tot_b / tot_t / 8e12 if tot_t else None}
    print(json.dumps(out, indent=1))
    sys.exit(0)

from boolsi_amd import synth
from boolsi_amd.compile import compile_problem
from boolsi_amd.constants import Mode
from boolsi_amd.engine import Engine
from boolsi_amd.input import parse_input_text

eng = Engine(0)
cfg = parse_input_text(synth.config3_yaml(), 4096, Mode.ATTRACT)
net, space = compile_problem(cfg)
eng.set_problem(net, space)
r = eng.attract_fgraph(0, 1 << 32, 4096)
print(json.dumps({'attractors': len(r.table), 'kernel_ms': r.stats['kernel_ms'], 'launches': r.stats['kernel_launches']}))
eng.close()
