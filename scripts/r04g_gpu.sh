#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04g_gpu_suite.log 2>&1
tail -4 gpurun_out/r04g_gpu_suite.log
F="--steps 20 --warmup 5 --whole-run 0 --no-side-lines --no-cpu-baseline --late 400"
for cfg in '{"bound_motion_ratio": 0.0}' '{"bound_motion_ratio": 0.02}' '{"bound_motion_ratio": 0.05}'; do
  timeout 300 python3 bench.py $F --loop-config "$cfg" 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = o['roofline']; l = o.get('late', {})
print('$cfg', 'ms/step %.3f' % o['ms_per_step'], 'knn ms %.3f' % r['step_ms'], 'swept %.4f' % r['swept_fraction'], 'frac %.4f' % r['frac'], 'redone', o['config']['rows_redone_cold_max'], {s['stage'][:14]: (round(s['ms'], 3), round(s.get('ms_in_loop') or 0, 3)) for s in o['rooflines'][1:6]})
print('   late:', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in l.items() if k in ('ms_per_step', 'knn_ms', 'swept_fraction', 'frac')})
"
done
