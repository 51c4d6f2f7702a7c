#!/bin/bash
# Dev tool (needs `git apply scripts/experiments/bwd_pipelined.patch`: the ablation macro lives in that experiment): where the fused NN-loss backward spends its time.  Builds one library per ablation (scripts/_ab/libwast3d_bwd_<bits>.so, -DW3D_BWD_ABLATE=<bits>:
# parts of the kernel compiled out, see desc_bwd_window.hip) and times scripts/bwd_bench.py with each.  Build here, time on the GPU box:
#   scripts/bwd_ablate.sh build      |      scripts/bwd_ablate.sh run
set -e
cd "$(dirname "$0")/.."
LIST="0 1 2 4 8 32 64 5 13 47 111 127"
if [ "$1" = build ]; then
  mkdir -p scripts/_ab
  objs=$(ls wast3d_amd/_obj/*.o | grep -v "/desc_bwd_window.o")
  for a in $LIST; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=on -I/opt/rocm/include -DROCPRIM_NO_DEPRECATED_WARNINGS -DW3D_BWD_ABLATE=$a \
          -Iinclude -Iwast3d_amd/csrc -c wast3d_amd/csrc/desc_bwd_window.hip -o scripts/_ab/desc_bwd_window_$a.o
    hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/_ab/libwast3d_bwd_$a.so $objs scripts/_ab/desc_bwd_window_$a.o
  done
  echo built $LIST
else
  for a in $LIST; do
    echo -n "ablate=$a: "
    W3D_LIB_PATH=scripts/_ab/libwast3d_bwd_$a.so timeout 300 python scripts/bwd_bench.py 2>/dev/null | grep "fused window=12288\|gF given" | tr '\n' ' '; echo
  done
fi
