set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_edge_cases.py tests/test_gpu_pipeline.py -q -m gpu ) > gpurun_out/r03/orbtests3.log 2>&1
tail -15 gpurun_out/r03/orbtests3.log
( timeout 300 python tests/fuzz_parity.py --seconds 90 --seed 5 --only orb ) > gpurun_out/r03/fuzz_orb3.log 2>&1
tail -3 gpurun_out/r03/fuzz_orb3.log
( timeout 300 python tools/bench_orb.py --batch 256 --reps 10 ) > gpurun_out/r03/orb_fused.log 2>&1
( VSLAM_ORB_UNFUSED=1 timeout 300 python tools/bench_orb.py --batch 256 --reps 10 ) > gpurun_out/r03/orb_unfused.log 2>&1
tail -2 gpurun_out/r03/orb_fused.log gpurun_out/r03/orb_unfused.log
