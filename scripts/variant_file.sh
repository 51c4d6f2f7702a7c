#!/bin/bash
# Dev tool: a second library in which ONE source file comes from another git revision (same-device A/B of a kernel file other than knn_mfma.hip):
#   scripts/variant_file.sh <tag> <rev> <file.hip>   ->  scripts/_ab/libwast3d_hip_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; rev=$2; f=$3
mkdir -p scripts/_ab/src
python -c "import sys; sys.path.insert(0, '.'); from wast3d_amd import build; build.build(verbose=False)"
git show $rev:wast3d_amd/csrc/$f > scripts/_ab/src/$f
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=on -I/opt/rocm/include -DROCPRIM_NO_DEPRECATED_WARNINGS -Iinclude -Iwast3d_amd/csrc \
      -c scripts/_ab/src/$f -o scripts/_ab/${f%.hip}_$tag.o
objs=$(ls wast3d_amd/_obj/*.o | grep -v "/${f%.hip}.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/_ab/libwast3d_hip_$tag.so $objs scripts/_ab/${f%.hip}_$tag.o
echo built scripts/_ab/libwast3d_hip_$tag.so
