#!/bin/bash
# Final measurements of a build, on the GPU box:  bash tools/profile_round.sh <tag>
#   1. un-profiled bench line with the CPU baseline          -> gpurun_out/<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command  -> gpurun_out/<tag>/stats/
#   3. separate --pmc passes for HBM traffic                 -> gpurun_out/<tag>/pmc_*/
set -e
TAG=${1:-round}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o run -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_$c.log 2>&1
done
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -5
