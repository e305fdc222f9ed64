set -u
tools/profile_round.sh r01c
tools/profile_sq.sh r01c_ba python tools/bench_ba.py --windows 256 --reps 1
tools/profile_tcc.sh r01c_ba python tools/bench_ba.py --windows 256 --reps 1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sgbm -o sgbm -- python tools/bench_sgbm.py --batch 32 --reps 3 > gpurun_out/prof_sgbm.log 2>&1
tail -3 gpurun_out/prof_sgbm.log
timeout 300 python bench.py > gpurun_out/bench_r01c.json 2> gpurun_out/bench_r01c.err; tail -c 600 gpurun_out/bench_r01c.json
