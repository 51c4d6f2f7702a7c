#!/bin/bash
# PMC passes over the kNN micro-benchmark (separate runs per counter group; kernel-trace only, as the pool requires)
# usage: knn_prof.sh <cfg>   e.g. 1000000,65536,256,20,0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/scripts/knn_bench.py --cfg $1 --iters 1 > $R/gpurun_out/pmc_$tag.log 2>&1
done
