#!/bin/bash
# Dev tool: build a second library from another git revision of one kernel file, for same-device A/B timing
# (MI355X boxes differ by up to ~10 % in sustained clock, so numbers from different gpurun calls are not comparable).
#   scripts/ab_build.sh <rev> <file.hip>   ->  scripts/_ab/libwast3d_hip_base.so
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}; f=${2:-knn_mfma.hip}
mkdir -p scripts/_ab/src
git show $rev:wast3d_amd/csrc/$f > scripts/_ab/src/$f
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=on -I/opt/rocm/include -DROCPRIM_NO_DEPRECATED_WARNINGS -fno-honor-nans \
      -Iinclude -Iwast3d_amd/csrc -c scripts/_ab/src/$f -o scripts/_ab/${f%.hip}.o
objs=$(ls wast3d_amd/_obj/*.o | grep -v "/${f%.hip}.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/_ab/libwast3d_hip_base.so $objs scripts/_ab/${f%.hip}.o
echo built scripts/_ab/libwast3d_hip_base.so from $rev:$f
