#!/bin/bash
# Steady-state HBM traffic of the loop's kernels: FETCH_SIZE and WRITE_SIZE passes over bench.py with enough iterations for the sweeps to settle
# (the first sweeps of a run drift apart and re-read the style set).  scripts/traffic_pass.sh <tag>
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc_$grp -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_$grp.log 2>&1
done
