#!/bin/bash
# one short bench line per loop configuration: usage scripts/ab_line.sh '{"adam_late": false}' '{}' ...
for cfg in "$@"; do
python3 bench.py --steps 20 --warmup 5 --whole-run 0 --no-side-lines --no-cpu-baseline --loop-config "$cfg" 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = o['roofline']
st = {s['stage'][:28]: (round(s['ms'], 2), round(s.get('ms_in_loop', 0), 2)) for s in o['rooflines'][1:]}
print('$cfg: ms/step %.3f knn %.3f swept %.4f |' % (o['ms_per_step'], r['step_ms'], r['swept_fraction']), st)"
done
