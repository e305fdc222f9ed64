#!/bin/bash
# build a variant of libvslam_hip.so with extra -D flags on ONE source file: tools/scratch/build_variant.sh NAME SRC.hip "-DX=1 ..."
# (output tools/scratch/variants/NAME.so; on the GPU box: cp it over stereo-visual-slam_amd/libvslam_hip.so before a run)
set -e
cd "$(dirname "$0")/../../stereo-visual-slam_amd/csrc"
NAME=$1; SRC=$2; shift 2
EXTRA=""
case $SRC in
  match_kernels.hip|orb_kernels.hip) EXTRA="-mllvm -amdgpu-mfma-vgpr-form";;
  lm_kernels.hip|ba_resident.hip|geom_kernels.hip|track_kernels.hip) EXTRA="-ffp-contract=fast";;
esac
OBJ=/tmp/variant_${NAME}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w $EXTRA "$@" -c $SRC -o $OBJ
OBJS=""
for f in api orb_kernels match_kernels geom_kernels lm_kernels ba_resident sgbm_kernels pnp_kernels track_kernels; do
  if [ "$f.hip" == "$SRC" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/variants/$NAME.so $OBJS -Wl,-rpath,/opt/rocm/lib
echo built $NAME
