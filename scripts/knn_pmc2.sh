#!/bin/bash
# Stall / issue-mix counters of the kNN kernel (dev tool): separate --pmc passes over scripts/knn_bench.py (kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_BRANCH SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/kpmc_$i -- python3 $R/scripts/knn_bench.py --cfg 1000000,65536,256,20,0 --iters 1 > $R/gpurun_out/kpmc_$i.log 2>&1
done
python3 - <<EOF
import csv, glob
for d in sorted(glob.glob("$R/gpurun_out/kpmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    rows = [r for r in csv.DictReader(open(f[0])) if "knn_mfma_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) > 500000]
    by = {}
    for r in rows: by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in by.items(): print(k, "%.4g" % v[-1], "(n=%d)" % len(v))
EOF
