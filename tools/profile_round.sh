#!/bin/bash
# Final measurements of a build, on the GPU box:  bash tools/profile_round.sh <tag>     (writes gpurun_out/<tag>/)
#   1. un-profiled bench line with the CPU baseline                      -> bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command              -> stats/   (kernel_stats.csv)
#   3. separate --pmc passes of the same command (never mixed with traces): HBM traffic, then the SQ groups
#      that show what bounds the kernel (VALU issue, LDS, waiting)        -> pmc_<group>/
#   4. tools/pmc_read.py turns 3 into pmc.json (copy to profiles/<tag>_pmc.json: bench.py reads traffic and
#      the issue bound from there)
#   5. the other BASELINE configs: tools/bench_configs.py un-profiled -> configs.jsonl, and under
#      rocprofv3 --kernel-trace --stats -> configs_stats/
set -e
TAG=${1:-round}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
PMC_CMD="python3 bench.py --no-cpu-baseline --steps 4 --warmup 0"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '+')
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$tag" -o run -- $PMC_CMD > "$OUT/pmc_$tag.json" 2> "$OUT/pmc_$tag.err" || echo "pmc pass $tag failed"
done
python3 tools/pmc_read.py $OUT "true, false>" 60 all > $OUT/pmc.json
python3 tools/pmc_read.py $OUT "true, true>" 60 all > $OUT/pmc_lower.json
cat $OUT/pmc.json
if [ "$2" != "--bench-only" ]; then
  python3 tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs_stats -o run -- python3 tools/bench_configs.py --no-cpu > $OUT/configs_stats.log 2>&1
fi
find $OUT -name "*kernel_stats.csv" | head -3 | xargs -n1 head -6
