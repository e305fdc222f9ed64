#!/bin/bash
# Collects the whole evidence set of one round on the GPU box (run through gpurun):  tools/profile_all.sh <tag>
set -u
TAG=${1:-r02}
tools/profile_round.sh $TAG --repeats 1
tools/profile_sq.sh ${TAG}_ba python tools/bench_ba.py --windows 256 --reps 1
tools/profile_tcc.sh ${TAG}_ba python tools/bench_ba.py --windows 256 --reps 1
tools/profile_sq.sh ${TAG}_orb python tools/bench_orb.py --batch 256 --reps 1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sgbm_$TAG -o sgbm -- python tools/bench_sgbm.py --batch 32 --reps 3 > gpurun_out/prof_sgbm_$TAG.log 2>&1
tail -3 gpurun_out/prof_sgbm_$TAG.log
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -c 600 gpurun_out/bench_$TAG.json
