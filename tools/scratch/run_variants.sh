#!/bin/bash
# on the GPU box: run a command once per variant library (tools/scratch/variants/*.so): run_variants.sh "cmd" name1 name2 ...
CMD=$1; shift
cp stereo-visual-slam_amd/libvslam_hip.so /tmp/base.so
for v in base "$@"; do
  if [ $v == base ]; then cp /tmp/base.so stereo-visual-slam_amd/libvslam_hip.so; else cp tools/scratch/variants/$v.so stereo-visual-slam_amd/libvslam_hip.so; fi
  echo "== $v"; bash -c "$CMD" 2>&1 | grep -v amdgpu.ids | tail -2
done
cp /tmp/base.so stereo-visual-slam_amd/libvslam_hip.so
