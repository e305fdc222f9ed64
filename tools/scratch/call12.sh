cd $GRAFT_REPO_ROOT
( VSLAM_ORB_PROFILE=1 timeout 300 python tools/bench_orb.py --batch 256 --reps 1 ) 2>&1 | grep "orb profile" | tail -16
