#!/bin/bash
# PMC passes of the other timed kernels, on the GPU box:   bash tools/profile_configs.sh <tag> [key ...]
# For every key (config4 -> k_target, config5 -> k_simulate_sliced64, chaotic -> k_attract<2,3,1>, config3 / northstar40 ->
# k_attract_pool) the same separate --pmc passes as tools/profile_round.sh, of `tools/bench_configs.py --no-cpu --only <key>`
# (the program directly behind `--`, never mixed with traces), reduced by tools/pmc_read.py to
# gpurun_out/<tag>/pmc_<key>.json (copy to profiles/r03_pmc_<key>.json: tools/bench_configs.py names the bound from it).
set -e
TAG=${1:-configs}; shift || true
KEYS=${@:-config4 config5 chaotic}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for key in $KEYS; do
  case $key in
    config4) needle=k_target ;;
    config5) needle=k_simulate_sliced64 ;;
    chaotic) needle="k_attract<" ;;
    *) needle=k_attract_pool ;;
  esac
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/$key
  mkdir -p $OUT
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | tr ' ' '+')
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$tag" -o run -- python3 tools/bench_configs.py --no-cpu --only $key > "$OUT/pmc_$tag.jsonl" 2> "$OUT/pmc_$tag.err" || echo "pmc pass $key $tag failed"
  done
  python3 tools/pmc_read.py $OUT "$needle" > $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_$key.json
  echo "== $key ($needle)"; grep -E "busy_frac|wave_wait|hbm_bytes|conflict" $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_$key.json
done
