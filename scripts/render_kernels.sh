#!/bin/bash
# Per-kernel times of the stand-alone render leg (dev tool): rocprofv3 kernel stats of scripts/render_bench.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/rb_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rb_stats -- python3 $R/scripts/render_bench.py 2>&1 | grep "render leg"
python3 - <<EOF
import csv, glob
f = glob.glob("$R/gpurun_out/rb_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["AverageNs"]) * int(r["Calls"]) > 2e4 and "knn_lists" not in r["Name"]: print(r["Name"][:60].ljust(60), r["Calls"], "%.3f ms" % (float(r["AverageNs"]) / 1e6))
EOF
rm -rf $R/gpurun_out/rb_stats
