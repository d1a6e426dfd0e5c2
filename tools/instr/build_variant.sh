#!/bin/bash
# One source of the library rebuilt with extra flags, linked with the product build's other objects:
#   tools/instr/build_variant.sh NAME raster_bwd "-DBWD_KO=1"   ->  manus_amd/libmanus_hip_NAME.so   (load: MANUS_HIP_VARIANT=NAME)
# (python -m manus_amd.build with MGR_VARIANT / MGR_EXTRA_FLAGS rebuilds all nine sources; this takes a minute per variant)
set -e
NAME=$1; SRC=$2; EXTRA=$3
cd "$(dirname "$0")/../../manus_amd"
(cd .. && python -m manus_amd.build >/dev/null)
mkdir -p build_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1"
/opt/rocm/bin/hipcc $EXTRA $FLAGS -c csrc/$SRC.hip -o build_$NAME/$SRC.o
OBJS=""
for o in build/*.o; do b=$(basename $o); if [ "$b" = "$SRC.o" ]; then OBJS="$OBJS build_$NAME/$SRC.o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmanus_hip_$NAME.so $OBJS
echo built libmanus_hip_$NAME.so
