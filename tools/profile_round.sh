#!/bin/bash
# Reproduces the files under profiles/ (run on the GPU box):  bash tools/profile_round.sh <tag> [bench args...]
# One kernel-trace/stats pass and four separate PMC passes (never mixed with other trace domains), each over
# `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`; summaries land in gpurun_out/prof_<tag>/.
set -e
TAG=${1:?tag}; shift || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e $*"
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o runc -- $BENCH > $OUT/kt.log 2>&1
pass() {  # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o runc -- $BENCH > $OUT/$name.log 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass grbm GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
python3 $ROOT/tools/summarise_profiles.py $OUT $TAG "$@"
