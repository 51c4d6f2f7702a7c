#!/bin/bash
# run_1001 (the whole 1001-iteration run of a fresh loop) under LoopConfig variants, one box: scripts/ab_whole_run.sh out.log '{...}' '{...}' ...
out=$1; shift
: > "$out"
for c in '{}' "$@" '{}'; do
  python bench.py --steps 2 --warmup 1 --no-side-lines --no-cpu-baseline --whole-run 1001 --loop-config "$c" 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read()); w = o['run_1001']
print(json.dumps({'cfg': sys.argv[1], 'iters_per_s': round(w['iters_per_s'], 2), 'ms_mean': round(w['ms_mean'], 3), 'ms_p50': round(w['ms_p50'], 3), 'ms_p90': round(w['ms_p90'], 3), 'ms_max': round(w['ms_max'], 1), 'swept_at': w['swept_fraction_at'], 'swept_mean': round(w['swept_fraction_mean'], 4), 'knn_ms_mean': round(w['knn_ms_mean'], 3)}))" "$c" >> "$out"
done
cat "$out"
