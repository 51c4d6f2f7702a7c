#!/bin/bash
# round 4, second GPU call: the new kNN paths (emit sweep, cached lower bounds), SSIM, raster radix fallback; then the bench line
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_geometry.py tests/test_gpu_raster.py tests/test_gpu_vgg.py -x -q -m gpu -k "emit or cached_bounds or motion_bound_with_prune or ssim or pruned_sweep_equals or window or slot or style_loop" > gpurun_out/r04b_newtests.log 2>&1
tail -5 gpurun_out/r04b_newtests.log
timeout 300 python3 bench.py --config raster > gpurun_out/r04b_raster_line.json 2> gpurun_out/r04b_raster.err
cut -c1-900 gpurun_out/r04b_raster_line.json
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04b_bench_line.json 2> gpurun_out/r04b_bench.err
tail -3 gpurun_out/r04b_bench.err
python3 - <<'PY'
import json
try:
    o = json.loads(open("gpurun_out/r04b_bench_line.json").read().strip().splitlines()[-1])
    print("value", o["value"], "ms", o["ms_per_step"], "roofline frac", o["roofline"]["frac"], "swept", o["roofline"]["swept_fraction"], "knn ms", o["roofline"]["step_ms"])
    r = o.get("run_1001", {})
    print("run_1001", {k: r.get(k) for k in ("iters_per_s", "ms_mean", "ms_p50", "ms_max", "swept_fraction_at", "frac", "rows_redone_cold_max")})
    for st in o["rooflines"]:
        print(st["stage"][:60], st.get("ms"), st.get("ms_in_loop"))
    for sl in o.get("side_lines", []):
        print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("frac", "achieved")}) for k, v in sl.items() if k in ("line", "ms_per_step", "ms_per_sweep", "iters_per_s", "swept_fraction", "roofline", "value", "error", "warm_sweep")})
except Exception as e:
    print("parse failed", e)
PY
