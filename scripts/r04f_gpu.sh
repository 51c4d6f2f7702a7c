#!/bin/bash
# round 4: the whole -m gpu suite, the round's profile (scripts/profile_round.sh r04), the default bench line
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04f_gpu_suite.log 2>&1
tail -3 gpurun_out/r04f_gpu_suite.log
bash scripts/profile_round.sh r04 > gpurun_out/r04f_profile.log 2>&1
cd $R
python3 scripts/trim_profile_out.py r04
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04f_bench_line.json 2> gpurun_out/r04f_bench.err
tail -2 gpurun_out/r04f_bench.err
du -sh gpurun_out
python3 - <<'PY'
import json
o = json.loads(open("gpurun_out/r04f_bench_line.json").read().strip().splitlines()[-1])
print("value", o["value"], "ms", o["ms_per_step"], "frac", o["roofline"]["frac"], "swept", o["roofline"]["swept_fraction"])
r = o["run_1001"]; print("run_1001", r["iters_per_s"], r["ms_mean"], r["ms_p50"], r["swept_fraction_at"])
for sl in o.get("side_lines", []):
    print(json.dumps(sl)[:600])
PY
