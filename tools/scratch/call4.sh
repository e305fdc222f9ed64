cd $GRAFT_REPO_ROOT
PASSES=fetch bash tools/profile_round.sh r03a > gpurun_out/prof_r03a_fetch.log 2>&1
tail -30 gpurun_out/prof_r03a_fetch.log
