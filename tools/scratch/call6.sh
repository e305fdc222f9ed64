cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_sgbm.py -q -m gpu ) 2>&1 | tail -2
( timeout 300 python tests/fuzz_parity.py --seconds 120 --seed 9 --only sgbm ) 2>&1 | tail -1
