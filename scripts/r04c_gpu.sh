#!/bin/bash
# round 4, GPU call: full -m gpu suite, then A/B of the selection schemes on the headline workload
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04c_gpu_suite.log 2>&1
tail -4 gpurun_out/r04c_gpu_suite.log
F="--steps 20 --warmup 5 --whole-run 0 --no-side-lines --no-cpu-baseline"
for cfg in '{"emit": false, "bounds_cache": false}' '{"emit": true, "bounds_cache": false}' '{}'; do
  timeout 300 python3 bench.py $F --loop-config "$cfg" 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = o['roofline']
print('$cfg', 'ms/step %.3f' % o['ms_per_step'], 'knn ms %.3f' % r['step_ms'], 'swept %.4f' % r['swept_fraction'], 'frac %.4f' % r['frac'], {s['stage'][:14]: (round(s['ms'], 3), round(s.get('ms_in_loop') or 0, 3)) for s in o['rooflines'][1:6]})
"
done
