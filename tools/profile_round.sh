#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   1. kernel trace + stats (CSV)   2. PMC pass FETCH_SIZE   3. PMC pass WRITE_SIZE   (separate passes: TCC slots)
# usage: tools/profile_round.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --render-workers 0 --unique-frames 64 --inputs resident --repeats 1 $*"
PASSES=${PASSES:-trace fetch write}
[[ $PASSES == *trace* ]] && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py $ARGS > $OUT/trace.log 2>&1
[[ $PASSES == *fetch* ]] && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
[[ $PASSES == *write* ]] && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python bench.py $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
python tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
