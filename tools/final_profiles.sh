#!/bin/bash
# Everything profiles/r03_* is made of, in dependency order, on the GPU box:   bash tools/final_profiles.sh <tag>
#   1. tools/profile_round.sh (bench + kernel stats + PMC passes of the bench)  -> pmc.json, copied to profiles/r03_pmc.json
#   2. the bench again, now naming its bound from THAT profile                   -> bench_final.json
#   3. tools/profile_configs.sh (PMC passes of config 4 / 5 / chaotic)           -> profiles/r03_pmc_<key>.json
#   4. tools/bench_configs.py with the bounds named from 3, un-profiled and under rocprofv3 --kernel-trace --stats
#   5. tools/full_space.py
set -e
TAG=${1:-r03f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG --bench-only > $OUT.round.log 2>&1 || { tail -20 $OUT.round.log; exit 1; }
cp $OUT/pmc.json profiles/r03_pmc.json
cp $OUT/pmc_lower.json profiles/r03_pmc_lower.json
python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err
BSX_CUBE_STREAMS=1 python3 bench.py --no-cpu-baseline > $OUT/bench_serial.json 2> $OUT/bench_serial.err      # (no side streams: clean per-build launch times)
bash tools/profile_configs.sh $TAG/cfg config4 config5 chaotic > $OUT.cfg.log 2>&1 || { tail -20 $OUT.cfg.log; exit 1; }
for k in config4 config5 chaotic; do cp $OUT/cfg/pmc_$k.json profiles/r03_pmc_$k.json; done
python3 tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs_stats -o run -- python3 tools/bench_configs.py --no-cpu > $OUT/configs_stats.log 2>&1
python3 tools/full_space.py > $OUT/full_space.json 2> $OUT/full_space.err
cp profiles/r03_pmc*.json $OUT/            # (gpurun only brings gpurun_out/ back)
tail -c 1500 $OUT/bench_final.json; echo; grep -E "wall_s|calls" $OUT/full_space.json
