#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_geometry.py -x -q -m gpu -k "emit or cached_bounds or pruned_sweep or hybrid" 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 900 python3 -m pytest tests/test_gpu_raster.py tests/test_gpu_parity_fp32.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | head -3
F="--steps 20 --warmup 5 --no-side-lines --no-cpu-baseline"
for cfg in '{"bounds_summaries": false}' '{}'; do
  timeout 300 python3 bench.py $F --loop-config "$cfg" 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = o['roofline']; w = o['run_1001']
print('$cfg', 'ms/step %.3f' % o['ms_per_step'], 'knn ms %.3f' % r['step_ms'], 'swept %.4f' % r['swept_fraction'], 'redone', o['config']['rows_redone_cold_max'], {s['stage'][:14]: (round(s['ms'], 3), round(s.get('ms_in_loop') or 0, 3)) for s in o['rooflines'][1:5]})
print('   run_1001 it/s %.2f ms_mean %.3f p50 %.3f' % (w['iters_per_s'], w['ms_mean'], w['ms_p50']), w['swept_fraction_at'], 'swept mean %.4f' % w['swept_fraction_mean'])
"
done
