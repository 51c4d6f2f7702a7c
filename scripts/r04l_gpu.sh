#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
F="--steps 20 --warmup 5 --whole-run 0 --no-side-lines --no-cpu-baseline"
for cfg in '{"adam_late": true}' '{}' '{"adam_late": true}' '{}'; do
  timeout 300 python3 bench.py $F --loop-config "$cfg" 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = o['roofline']
print('$cfg', 'ms/step %.3f' % o['ms_per_step'], 'knn ms %.3f' % r['step_ms'], {s['stage'][:14]: (round(s['ms'], 3), round(s.get('ms_in_loop') or 0, 3)) for s in o['rooflines'][1:9]})
"
done
