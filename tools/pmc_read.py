"""Sum rocprofv3 --pmc counters per kernel from the rocpd databases under a directory.
usage: python tools/pmc_read.py gpurun_out/pmc_dir [kernel-substring]"""
import glob
import sqlite3
import sys
import collections

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else 'lean'
for f in sorted(glob.glob(root + '/**/*.db', recursive=True)):
    c = sqlite3.connect(f)
    cols = [r[1] for r in c.execute('pragma table_info(counters_collection)')]
    rows = c.execute('select * from counters_collection').fetchall()
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        d = dict(zip(cols, r))
        if needle in d['kernel_name']:
            acc[d['counter_name']][d['dispatch_id']] += d['value']
    for name, by in sorted(acc.items()):
        vals = list(by.values())
        print('%-26s launches %d  per launch %.5g' % (name, len(vals), sum(vals) / len(vals)))
