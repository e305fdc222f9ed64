cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_fused_paths.py -q -m gpu -x 2>&1 | tail -1
for b in 32 160; do timeout 300 python tools/bench_sgbm.py --batch $b --reps 3 2>&1 | grep -E "^B=|forward" | cut -c1-130; done
VSLAM_SGBM_FWD_MIN=9999 timeout 300 python tools/bench_sgbm.py --batch 8 --reps 3 2>&1 | grep -E "^B=|down" | cut -c1-130
