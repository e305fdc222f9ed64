cd $GRAFT_REPO_ROOT
for v in pb128 pb256 pb512; do
  echo $v; ( VSLAM_LIB=build/libvslam_hip_$v.so timeout 300 python tools/bench_sgbm.py --batch 32 --reps 4 ) 2>&1 | grep -v amdgpu.ids | cut -c1-260 | head -2
done
