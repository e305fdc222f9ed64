#!/bin/bash
# One-call GPU check of a kernel family after an edit (run through gpurun): its parity tests, a fuzz slice, its micro-benchmark.
#   tools/gpu_check.sh orb|sgbm|ba|match [fuzz seconds] [fuzz seed]
cd "$GRAFT_REPO_ROOT"
FAM=${1:-orb}; SECS=${2:-60}; SEED=${3:-1}
case $FAM in
  orb)   T="tests/test_gpu_orb.py tests/test_gpu_edge_cases.py tests/test_gpu_pipeline.py"; F=orb;   B="tools/bench_orb.py --batch 256 --reps 10";;
  sgbm)  T="tests/test_gpu_sgbm.py";                                                        F=sgbm;  B="tools/bench_sgbm.py --batch 32 --reps 4";;
  ba)    T="tests/test_gpu_lm.py tests/test_gpu_pipeline.py";                                F=ba;    B="tools/bench_ba.py --windows 256 --reps 8";;
  match) T="tests/test_gpu_match.py";                                                        F=match; B="tools/bench_match.py";;
esac
( timeout 900 python -m pytest $T -q -m gpu ) 2>&1 | tail -2
( timeout $((SECS + 120)) python tests/fuzz_parity.py --seconds $SECS --seed $SEED --only $F ) 2>&1 | tail -1
( timeout 300 python $B ) 2>&1 | grep -v amdgpu.ids | tail -2
