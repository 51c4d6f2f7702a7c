#!/bin/bash
# Same-device A/B of the whole style-loop step: baseline library (scripts/_ab) vs in-tree, optional W3D_KNN_MODE list
cd "$(dirname "$0")/.."
pick='import json,sys; o=json.loads(sys.stdin.readline()); print("ms_per_step=%.3f knn_ms=%.3f knn_TF=%.1f" % (o["ms_per_step"], o["roofline"]["launch_ms"], o["roofline"]["achieved"]))'
for rep in 1 2; do
  echo "== base";  W3D_LIB_PATH=$PWD/scripts/_ab/libwast3d_hip_base.so python bench.py --no-cpu-baseline --steps 20 | python -c "$pick"
  for m in "$@"; do echo "== new mode=$m"; W3D_KNN_MODE=$m python bench.py --no-cpu-baseline --steps 20 | python -c "$pick"; done
done
