cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_edge_cases.py -q -m gpu ) 2>&1 | tail -2
( timeout 300 python tests/fuzz_parity.py --seconds 60 --seed 11 --only orb ) 2>&1 | tail -1
( timeout 300 python tools/bench_orb.py --batch 256 --reps 10 ) 2>&1 | grep -v amdgpu.ids | tail -2
