#!/bin/bash
# Same-device A/B: baseline library (scripts/_ab) vs the in-tree one, alternating, same configs
# usage: scripts/ab_knn.sh <cfg> [<cfg> ...]
cd "$(dirname "$0")/.."
for rep in 1 2; do
  echo "== base";  W3D_LIB_PATH=$PWD/scripts/_ab/libwast3d_hip_base.so python scripts/knn_bench.py --iters 5 --cfg "$@" 2>&1 | grep "ms"
  echo "== new";   python scripts/knn_bench.py --iters 5 --cfg "$@" 2>&1 | grep "ms"
done
