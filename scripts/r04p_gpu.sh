#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_geometry.py -x -q -m gpu -k "emit or cached_bounds or pruned_sweep or hybrid or knn" 2>&1 | grep -E "passed|failed|Error|assert" | head -5
timeout 900 python3 -m pytest tests/test_gpu_parity_fp32.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | head -3
F="--steps 20 --warmup 5 --no-side-lines --no-cpu-baseline"
timeout 300 python3 bench.py $F 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = o['roofline']; w = o['run_1001']
print('ms/step %.3f' % o['ms_per_step'], 'knn ms %.3f' % r['step_ms'], 'swept %.4f' % r['swept_fraction'], 'frac %.4f' % r['frac'], 'redone', o['config']['rows_redone_cold_max'])
print('   run_1001 it/s %.2f ms_mean %.3f p50 %.3f' % (w['iters_per_s'], w['ms_mean'], w['ms_p50']), w['swept_fraction_at'], 'redone', w['rows_redone_cold_max'])
"
timeout 300 python3 bench.py --config C2 --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2: cold %.3f ms (%.3f)  warm lists %.3f ms (%.3f)  warm emit %.3f ms (%.3f)' % (o['roofline']['launch_ms'], o['roofline']['frac'], o['warm_sweep']['ms'], o['warm_sweep']['frac'], o['warm_emit_sweep']['ms'], o['warm_emit_sweep']['frac']))"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04p_stats -- python3 $R/bench.py --steps 20 --warmup 5 --whole-run 0 --no-side-lines --no-cpu-baseline > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04p_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("select", "persistent", "plan_steady", "prev_threshold4", "descriptor_bwd")):
        print(r["Name"][:70], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
find gpurun_out/r04p_stats -type f ! -name "*kernel_stats.csv" -delete
