cd $GRAFT_REPO_ROOT
for B in 64 128 256; do
  a=$( (timeout 200 python tools/bench_orb.py --batch $B --reps 20) 2>&1 | grep "^B=" | awk '{print $3}')
  b=$( (VSLAM_ORB_UNFUSED=1 timeout 200 python tools/bench_orb.py --batch $B --reps 20) 2>&1 | grep "^B=" | awk '{print $3}')
  echo "orb B=$B (2B images) fused $a ms unfused $b ms"
done
