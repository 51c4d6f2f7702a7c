#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python3 -m pytest tests/test_gpu_raster.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python3 bench.py --config raster 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('raster fwd %.3f bwd %.3f ms, R %d' % (o['forward_ms'], o['backward_ms'], o['config']['tile_instances']))"
F="--steps 20 --warmup 5 --no-side-lines --no-cpu-baseline"
for cfg in '{"row_order_refresh": 16}' '{"row_order_refresh": 32}' '{"row_order_refresh": 64}' '{"row_order_refresh": 128}'; do
  timeout 300 python3 bench.py $F --loop-config "$cfg" 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = o['run_1001']
print('$cfg', 'ms/step %.3f' % o['ms_per_step'], '| run_1001 it/s %.2f ms_mean %.3f p50 %.3f' % (r['iters_per_s'], r['ms_mean'], r['ms_p50']), r['swept_fraction_at'], 'swept mean %.4f knn mean %.3f' % (r['swept_fraction_mean'], r['knn_ms_mean']))
"
done
