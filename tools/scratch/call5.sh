cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( timeout 600 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_determinism.py -q -m gpu ) > gpurun_out/r03/sgbmtests5.log 2>&1
tail -12 gpurun_out/r03/sgbmtests5.log
( timeout 200 python tests/fuzz_parity.py --seconds 60 --seed 7 --only sgbm ) > gpurun_out/r03/fuzz_sgbm5.log 2>&1
tail -3 gpurun_out/r03/fuzz_sgbm5.log
( timeout 300 python tools/bench_sgbm.py --batch 32 --reps 4 ) > gpurun_out/r03/sgbm_fused.log 2>&1
( VSLAM_SGBM_UNFUSED=1 timeout 300 python tools/bench_sgbm.py --batch 32 --reps 4 ) > gpurun_out/r03/sgbm_unfused.log 2>&1
tail -n 6 gpurun_out/r03/sgbm_fused.log; tail -n 6 gpurun_out/r03/sgbm_unfused.log
