#!/bin/bash
# Effective clock and MFMA duty of the kNN kernel: GRBM_GUI_ACTIVE and SQ_VALU_MFMA_BUSY_CYCLES in separate --pmc passes
# usage: knn_clock.sh <cfg> [<cfg> ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/scripts/knn_bench.py --cfg "$@" --iters 2 > $R/gpurun_out/pmc_$tag.log 2>&1
done
