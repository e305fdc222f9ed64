#!/bin/bash
# Builds a tuning variant of libvslam_hip.so: one translation unit recompiled with extra -D flags, the rest reused.
#   tools/build_variant.sh <name> <unit: lm_kernels|ba_resident|orb_kernels|match_kernels|...> "<extra flags>"
# Output: build/libvslam_hip_<name>.so (git-ignored; travels to the GPU box).  Select it with VSLAM_LIB=build/libvslam_hip_<name>.so
set -e
NAME=$1; UNIT=$2; EXTRA=$3
cd "$(dirname "$0")/../stereo-visual-slam_amd/csrc"
make -s -j8
mkdir -p ../../build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -Wno-unused-result"
case $UNIT in lm_kernels|ba_resident|geom_kernels|track_kernels) FLAGS="$FLAGS -ffp-contract=fast";; match_kernels|orb_kernels) FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form";; esac
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $UNIT.hip -o ../../build/${UNIT}_$NAME.o
OBJS=""; for u in api orb_kernels match_kernels geom_kernels lm_kernels ba_resident sgbm_kernels pnp_kernels track_kernels; do if [ $u = $UNIT ]; then OBJS="$OBJS ../../build/${UNIT}_$NAME.o"; else OBJS="$OBJS $u.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libvslam_hip_$NAME.so $OBJS -Wl,-rpath,/opt/rocm/lib
echo built build/libvslam_hip_$NAME.so
