#!/bin/bash
# A/B of LoopConfig variants on the headline workload, alternating runs on ONE box (boxes of the pool differ by +-3 %): ms per step, sweep ms, swept fraction, stage times.
# usage: scripts/ab_loop_configs.sh out.log '{"bound_motion_ratio": 0.01}' '{...}' ...   (the empty config {} = the default is always run first and last)
out=$1; shift
: > "$out"
run() {
  python bench.py --steps 20 --warmup 5 --no-side-lines --no-cpu-baseline --whole-run 0 --loop-config "$1" 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read())
st = {r['stage'].split(' (')[0][:28]: (round(r.get('ms_in_loop') or 0, 3), round(r['ms'], 3)) for r in o['rooflines'] if 'ms_in_loop' in r}
print(json.dumps({'cfg': sys.argv[1], 'ms_per_step': round(o['ms_per_step'], 3), 'knn_ms': round(o['roofline']['step_ms'], 3), 'swept': round(o['roofline']['swept_fraction'], 4), 'frac': round(o['roofline']['frac'], 4), 'stages[in_loop,alone]': st}))" "$1" >> "$out"
}
run '{}'
for c in "$@"; do run "$c"; done
run '{}'
cat "$out"
