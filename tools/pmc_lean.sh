#!/bin/bash
# SQ counter passes over the lean attract kernel (one rocprofv3 run per counter group; --pmc only).
# usage (on the GPU box): bash tools/pmc_lean.sh <out-prefix>
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp -d $OUT/$tag -o run -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/$tag.log 2>&1
  python3 - "$OUT/$tag" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pool' in r['Kernel_Name'] or 'lean' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in sorted(acc.items()):
    print('%-28s launches %d  per launch %.4g' % (k, n, v / n))
PY
done
