#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for p in 0 2 4 5 6 7; do
  W3D_KNN_MODE=$((p << 28)) timeout 300 python3 bench.py --config C2 --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('parts $p: cold %.3f ms (%.3f)  warm lists %.3f ms (%.3f)  warm emit %.3f ms (%.3f)' % (o['roofline']['launch_ms'], o['roofline']['frac'], o['warm_sweep']['ms'], o['warm_sweep']['frac'], o['warm_emit_sweep']['ms'], o['warm_emit_sweep']['frac']))"
done
