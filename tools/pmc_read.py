"""Sum rocprofv3 --pmc counters per launch of one kernel from the CSVs under a directory and derive what bounds it.
usage: python tools/pmc_read.py <dir with pmc_*/ sub-directories and pmc_*.json bench lines> [kernel-substring] [log2_batch] [all]
Writes one JSON object (profiles/<tag>_pmc.json is a copy of it; bench.py reads `per_executed_update`, `shader_clock_hz`,
`hbm_bytes_per_launch` and `issue_bound` from there).  Counters are summed over the device; per launch = the LARGEST
launches of the kernel (the top level of every cascade, which is what bench.py times as the dominant kernel).
gfx950 corrections (MI355X_MICROARCH.md): FETCH_SIZE counts half of wide reads, so it is doubled; FETCH_SIZE / WRITE_SIZE
are in KiB units of the TCC; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* are quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs.
The bench lines written by the same PMC passes give the updates those launches executed and their duration (serialised
by the profiler, so only ratios to the counters of the same pass are used)."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else 'k_attract_pool'
log2_batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
# 'all': average over every launch of the kernel (the bench's PMC passes run without warm-up steps, so the launches the
# counters see are exactly the ones whose updates and durations the bench line of the same pass adds up: the top level of
# every sub-block's cascade, of different sizes); default: the largest launches only
every_launch = len(sys.argv) > 4 and sys.argv[4] == 'all'
per = collections.defaultdict(dict)      # counter -> dispatch id -> value
for f in sorted(glob.glob(root + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if needle in r['Kernel_Name']:
            d = per[r['Counter_Name']]
            key = (f, r['Dispatch_Id'])
            d[key] = d.get(key, 0.0) + float(r['Counter_Value'])
out = {'kernel': needle, 'log2_batch': log2_batch, 'counters_per_launch': {}, 'launches_seen': {}}
for name, by in per.items():
    vals = sorted(by.values())
    big = vals if every_launch else ([v for v in vals if v >= 0.5 * vals[-1]] or vals)        # the dominant launches only
    out['counters_per_launch'][name] = sum(big) / len(big)
    out['launches_seen'][name] = len(big)
c = out['counters_per_launch']
lines = []
for f in sorted(glob.glob(root + '/pmc_*.json')):
    try:
        lines.append(json.loads(open(f).read().strip().splitlines()[-1]))
    except (ValueError, IndexError):
        pass
def block(b):
    """the roofline block of a bench line that is about this kernel (bench.py: `roofline` or its `other_build`)"""
    r = b.get('roofline') or {}
    want = needle.replace(' ', '')
    for cand in (r, r.get('other_build') or {}):
        if want in cand.get('kernel', '').replace(' ', ''):
            return cand
    return r if 'kernel' not in r else None


lines = [b for b in lines if block(b)]
upd = [block(b)['executed_updates_per_launch'] for b in lines]
if upd:
    out['executed_updates_per_launch'] = sum(upd) / len(upd)
    out['launches_per_bench_line'] = [block(b).get('launches_timed') for b in lines]
if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
    out['hbm_bytes_per_launch'] = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
    out['hbm_note'] = 'FETCH_SIZE x 2 (gfx950 counts half of wide reads) + WRITE_SIZE, KiB -> bytes, separate passes'
if 'GRBM_GUI_ACTIVE' in c:
    cycles = c['GRBM_GUI_ACTIVE'] / 8.0                          # shader cycles of the launch
    simds, cus = 1024, 256
    bound = {'launch_cycles': cycles}
    busy = {}
    if 'SQ_ACTIVE_INST_VALU' in c:
        busy['valu_busy_cycles'] = 4 * c['SQ_ACTIVE_INST_VALU']
        bound['valu_busy_frac'] = busy['valu_busy_cycles'] / (simds * cycles)
    if 'SQ_ACTIVE_INST_SCA' in c:
        busy['salu_busy_cycles'] = 4 * c['SQ_ACTIVE_INST_SCA']
        bound['salu_busy_frac'] = busy['salu_busy_cycles'] / (simds * cycles)
    if 'SQ_LDS_IDX_ACTIVE' in c:
        busy['lds_busy_cycles'] = c['SQ_LDS_IDX_ACTIVE']
        bound['lds_busy_frac'] = c['SQ_LDS_IDX_ACTIVE'] / (cus * cycles)
        if bound['lds_busy_frac'] > 1.0:
            bound['lds_note'] = ('above 1: the counters come from separate runs (clock and tile mix differ by a few percent) '
                                 'and the LDS pipeline is busy for the whole launch -- read as saturated')
    if 'SQ_LDS_BANK_CONFLICT' in c and 'SQ_LDS_IDX_ACTIVE' in c:
        bound['lds_conflict_share'] = c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']
    if 'SQ_WAIT_ANY' in c and 'SQ_WAVE_CYCLES' in c:
        bound['wave_wait_frac'] = c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']
    for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS'):
        if k in c:
            bound[k.lower() + '_per_launch'] = c[k]
    bound['reading'] = 'fractions of the launch during which the unit is busy, device average; the largest one is the bound'
    out['issue_bound'] = bound
    if upd:
        u = out['executed_updates_per_launch']
        out['per_executed_update'] = {k: v / u for k, v in busy.items()}
        for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS'):
            if k in c:
                out['per_executed_update'][k.lower()] = c[k] / u
    ms = [block(b)['avg_launch_ms'] for b in lines]
    if ms:
        out['shader_clock_hz'] = cycles / (sum(ms) / len(ms) * 1e-3)
        out['clock_note'] = 'GRBM_GUI_ACTIVE / 8 XCDs / the launch duration of the same profiled runs'
print(json.dumps(out, indent=1))
