#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_geometry.py -x -q -m gpu -k "emit or cached_bounds or pruned_sweep_equals or window or slot" > gpurun_out/r04d_tests.log 2>&1
tail -4 gpurun_out/r04d_tests.log
F="--steps 20 --warmup 5 --whole-run 0 --no-side-lines --no-cpu-baseline"
for cfg in '{"emit": true, "bounds_cache": false}' '{}'; do
  timeout 300 python3 bench.py $F --loop-config "$cfg" 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = o['roofline']
print('$cfg', 'ms/step %.3f' % o['ms_per_step'], 'knn ms %.3f' % r['step_ms'], 'swept %.4f' % r['swept_fraction'], 'frac %.4f' % r['frac'], {s['stage'][:14]: (round(s['ms'], 3), round(s.get('ms_in_loop') or 0, 3)) for s in o['rooflines'][1:6]})
"
done
