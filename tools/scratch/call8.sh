cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_pipeline.py tests/test_gpu_determinism.py tests/test_gpu_sequence.py -q -m gpu ) 2>&1 | tail -4
( timeout 300 python tools/bench_orb.py --batch 256 --reps 10 ) 2>&1 | grep -v amdgpu.ids | tail -2
( VSLAM_ORB_NOSPLIT=1 timeout 300 python tools/bench_orb.py --batch 256 --reps 10 ) 2>&1 | grep -v amdgpu.ids | tail -2
( timeout 300 python bench.py --no-cpu-baseline --inputs resident --unique-frames 64 ) 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split', r['value'], r['ms_per_step'], r['timing']['ms_per_step_each'])"
( VSLAM_ORB_NOSPLIT=1 timeout 300 python bench.py --no-cpu-baseline --inputs resident --unique-frames 64 ) 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nosplit', r['value'], r['ms_per_step'], r['timing']['ms_per_step_each'])"
