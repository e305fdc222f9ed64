cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03/gputests_final.log 2>&1
tail -4 gpurun_out/r03/gputests_final.log
( time timeout 600 python bench.py ) > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err
tail -c 200 gpurun_out/r03/bench_final.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r03b > gpurun_out/prof_r03b.log 2>&1
grep -E "pnp_|lm_window|orb_fast" gpurun_out/prof_r03b.log | head -8
python tools/bench_latency.py 2>&1 | grep -v amdgpu | tail -12
