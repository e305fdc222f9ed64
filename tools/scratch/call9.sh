cd $GRAFT_REPO_ROOT
for v in "" prio1 prio4 prio16; do
  if [ -z "$v" ]; then L=""; else L="build/libvslam_hip_$v.so"; fi
  echo "variant: ${v:-default}"; ( VSLAM_LIB=$L timeout 300 python tools/bench_ba.py --windows 256 --reps 8 ) 2>&1 | grep -v amdgpu.ids | tail -1
done
( VSLAM_LIB=build/libvslam_hip_prio4.so timeout 600 python -m pytest tests/test_gpu_lm.py -q -m gpu ) 2>&1 | tail -2
