#!/bin/bash
# HBM traffic of the kNN kernel: separate --pmc passes (FETCH_SIZE, WRITE_SIZE, L2 hit/miss), kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc2_$tag -- python3 $R/scripts/knn_bench.py --cfg 1000000,65536,256,20,0 --iters 1 > $R/gpurun_out/pmc2_$tag.log 2>&1
done
