#!/bin/bash
# round 4 final: -m gpu suite, the round's profile, the default bench line
R=$GRAFT_REPO_ROOT
cd $R
timeout 1800 python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
bash scripts/profile_round.sh r04 > gpurun_out/r04m_profile.log 2>&1
cd $R
python3 scripts/trim_profile_out.py r04
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04m_bench_line.json 2> gpurun_out/r04m_bench.err
tail -2 gpurun_out/r04m_bench.err
du -sh gpurun_out
python3 - <<'PY'
import json
o = json.loads(open("gpurun_out/r04m_bench_line.json").read().strip().splitlines()[-1])
print("value", o["value"], "ms", o["ms_per_step"], "frac", o["roofline"]["frac"], "swept", o["roofline"]["swept_fraction"], "knn", o["roofline"]["step_ms"])
r = o["run_1001"]; print("run_1001", r["iters_per_s"], r["ms_mean"], r["ms_p50"], r["swept_fraction_at"])
for st in o["rooflines"]:
    print(st["stage"][:70], st.get("ms"), st.get("ms_in_loop"), round(st.get("frac", 0), 4))
for sl in o.get("side_lines", []):
    print(json.dumps(sl)[:420])
PY
