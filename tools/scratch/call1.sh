set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03/gputests1.log 2>&1
tail -5 gpurun_out/r03/gputests1.log
( time timeout 600 python bench.py ) > gpurun_out/r03/bench1.json 2> gpurun_out/r03/bench1.err
tail -c 600 gpurun_out/r03/bench1.err
timeout 120 tools/scratch/mfma_f64_rate > gpurun_out/r03/mfma_f64_rate.log 2>&1
cat gpurun_out/r03/mfma_f64_rate.log
