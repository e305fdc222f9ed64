cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03/gputests_final.log 2>&1
tail -4 gpurun_out/r03/gputests_final.log
( time timeout 600 python bench.py ) > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err
tail -c 300 gpurun_out/r03/bench_final.err
bash tools/profile_round.sh r03b > gpurun_out/prof_r03b.log 2>&1
grep -E "orb_|lm_window|match_train" gpurun_out/prof_r03b.log | head -12
