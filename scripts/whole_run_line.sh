#!/bin/bash
# one whole 1001-iteration run of the default workload, printed short: usage scripts/whole_run_line.sh ['{"loop": "config"}']
cfg=${1:-{\}}
python3 bench.py --steps 20 --warmup 5 --no-side-lines --no-cpu-baseline --whole-run 1001 --loop-config "$cfg" 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); w = o['run_1001']
print('$cfg: 20-step %.3f ms | run_1001 %.1f it/s mean %.3f p50 %.3f p90 %.3f max %.1f first %.1f | stations' % (o['ms_per_step'], w['iters_per_s'], w['ms_mean'], w['ms_p50'], w['ms_p90'], w['ms_max'], w['ms_first']), [(s['iterations'][1], round(s['swept_fraction'], 3), round(s['ms_per_step'], 2)) for s in w['windows']], 'redone', w['rows_redone_cold_max'])"
