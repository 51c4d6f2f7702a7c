#!/bin/bash
# Round profile of bench.py on the GPU box:  scripts/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>_stats/
#   2. separate --pmc passes (kernel-trace only, one counter group each) -> gpurun_out/<tag>_pmc_<counter>/
# Copy the summaries to profiles/ afterwards (gpurun_out/ is scratch).
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py > $R/gpurun_out/${tag}_bench.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc_$t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_$t.log 2>&1
done
tail -1 $R/gpurun_out/${tag}_bench.log
