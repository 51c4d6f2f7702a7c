#!/bin/bash
# Dev tool: a second library that differs from the in-tree one in ONE (format, width) leaf of knn_mfma.hip compiled with extra flags — same-device A/B and diagnostic
# builds (W3D_LIB_PATH selects the library).   scripts/variant_build.sh <tag> [fmt=0] [d=256] -- <extra hipcc flags>   ->  scripts/_ab/libwast3d_hip_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
fmt=0; d=256
if [ "$1" != "--" ]; then fmt=$1; shift; fi
if [ "$1" != "--" ]; then d=$1; shift; fi
shift || true
mkdir -p scripts/_ab
python -c "import sys; sys.path.insert(0, '.'); from wast3d_amd import build; build.build(verbose=False)"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=on -I/opt/rocm/include -DROCPRIM_NO_DEPRECATED_WARNINGS -fno-honor-nans \
      -DW3D_KNN_FMT=$fmt -DW3D_KNN_D=$d "$@" -c wast3d_amd/csrc/knn_mfma.hip -o scripts/_ab/knn_mfma_f${fmt}_d${d}_$tag.o
objs=$(ls wast3d_amd/_obj/*.o | grep -v "/knn_mfma_f${fmt}_d${d}.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/_ab/libwast3d_hip_$tag.so $objs scripts/_ab/knn_mfma_f${fmt}_d${d}_$tag.o
echo built scripts/_ab/libwast3d_hip_$tag.so
