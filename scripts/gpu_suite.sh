#!/bin/bash
# GPU test suite in groups, each under its own time limit and logging straight to gpurun_out/ (a hang is localised and the logs survive a kill).
# usage: scripts/gpu_suite.sh <tag> [group ...]     groups: geom raster shard scale cloud vgg parity bench
tag=$1; shift
groups=${@:-geom raster shard scale cloud vgg parity bench}
mkdir -p gpurun_out
T="python -m pytest -x -q --timeout 200 -p no:cacheprovider"
for g in $groups; do
  log=gpurun_out/${tag}_$g.log
  case $g in
    geom)   timeout 500 $T tests/test_gpu_geometry.py > $log 2>&1 ;;
    raster) timeout 500 $T tests/test_gpu_raster.py -k "not sharded and not at_scale" > $log 2>&1 ;;
    shard)  timeout 900 $T --timeout 320 tests/test_gpu_raster.py -k "sharded" > $log 2>&1 ;;
    scale)  timeout 400 $T tests/test_gpu_raster.py -k "at_scale" > $log 2>&1 ;;
    cloud)  timeout 400 $T tests/test_gpu_pointcloud.py tests/test_gpu_edges.py > $log 2>&1 ;;
    vgg)    timeout 300 $T tests/test_gpu_vgg.py > $log 2>&1 ;;
    parity) timeout 900 $T --timeout 400 -s tests/test_gpu_parity_fp32.py > $log 2>&1 ;;
    bench)  timeout 600 python bench.py --steps 20 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err ;;
  esac
  echo "== $g rc=$? $(tail -1 $log 2>/dev/null | cut -c1-150)"
done
