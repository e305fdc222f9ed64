set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03/gputests2.log 2>&1
tail -8 gpurun_out/r03/gputests2.log
