cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( timeout 600 python bench.py --depth sgbm --batch 32 --unique-frames 32 --no-cpu-baseline --inputs resident ) > gpurun_out/r03/bench_sgbm_b32.json 2> gpurun_out/r03/bench_sgbm_b32.err
tail -c 300 gpurun_out/r03/bench_sgbm_b32.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_sgbm -o sgbm -- python tools/bench_sgbm.py --batch 32 --reps 4 > gpurun_out/r03/prof_sgbm.log 2>&1
ls gpurun_out/r03/prof_sgbm
