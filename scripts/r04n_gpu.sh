#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for w in 8 4 2; do
  timeout 600 python3 scripts/rank_sim.py $w > gpurun_out/r04n_ranksim_w$w.log 2>&1
  tail -6 gpurun_out/r04n_ranksim_w$w.log | cut -c1-300
done
timeout 600 python3 bench.py --force-distributed --steps 10 --warmup 3 --whole-run 0 --no-side-lines --no-cpu-baseline 2> gpurun_out/r04n_dist1.err | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('world-1 distributed path: ms/step %.3f' % o['ms_per_step'], o['rccl'])"
tail -3 gpurun_out/r04n_dist1.err
