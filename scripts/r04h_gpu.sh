#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python3 scripts/emit_diag.py 20 > gpurun_out/r04h_emit_diag.log 2>&1
tail -16 gpurun_out/r04h_emit_diag.log
F="--steps 20 --warmup 5 --whole-run 0 --no-side-lines --no-cpu-baseline"
for cfg in '{"schedule": "early"}' '{}'; do
  timeout 300 python3 bench.py $F --loop-config "$cfg" 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = o['roofline']
print('$cfg', 'ms/step %.3f' % o['ms_per_step'], 'knn ms %.3f' % r['step_ms'], {s['stage'][:14]: (round(s['ms'], 3), round(s.get('ms_in_loop') or 0, 3)) for s in o['rooflines'][1:8]})
"
done
timeout 600 python3 -m pytest tests/test_gpu_parity_fp32.py tests/test_gpu_raster.py tests/test_gpu_vgg.py tests/test_gpu_pointcloud.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | tail -4
