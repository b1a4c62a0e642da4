#!/bin/bash
# tools/build_variant.sh TAG [extra hipcc flags...] -- A/B build of the library: sac_amd/libsac_amd_TAG.so with kernels_pred.hip
# (and only it) compiled with the extra flags; select it with SACAMD_LIB_PATH.  Experiment tooling, not part of the product build.
set -e
TAG=$1; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
C=$HERE/sac_amd/csrc
mkdir -p $C/build_$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-value"
for f in ${VARIANT_SRCS:-kernels_pred}; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$f.hip -o $C/build_$TAG/$f.o & done
wait
OBJS=""
for f in kernels_pred kernels_misc kernels_coder host; do
  if [ -f $C/build_$TAG/$f.o ]; then OBJS="$OBJS $C/build_$TAG/$f.o"; else OBJS="$OBJS $C/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $HERE/sac_amd/libsac_amd_$TAG.so -L/opt/rocm/lib -lrccl
echo built $HERE/sac_amd/libsac_amd_$TAG.so
