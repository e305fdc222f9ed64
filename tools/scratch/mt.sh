cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for d in 0 15; do
VSLAM_MATCH_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mt_$d -o m -- python tools/bench_match.py --reps 3 > gpurun_out/mt_$d.log 2>&1
grep "match_" gpurun_out/mt_$d/m_kernel_stats.csv | cut -d, -f1-8 | cut -c1-60,150-260
done
