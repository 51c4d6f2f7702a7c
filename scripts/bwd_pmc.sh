#!/bin/bash
# Stall / issue-mix / cache counters of the fused window backward (dev tool): separate --pmc passes over scripts/bwd_bench.py (kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/bpmc_$i -- python3 $R/scripts/${BWD_SCRIPT:-bwd_bench.py} > $R/gpurun_out/bpmc_$i.log 2>&1
done
python3 - <<EOF
import csv, glob
for d in sorted(glob.glob("$R/gpurun_out/bpmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no counters"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "descriptor_bwd_window_kernel<true" in r["Kernel_Name"]]
    by = {}
    for r in rows: by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in by.items(): print(k, "%.4g" % v[-1], "(n=%d)" % len(v))
EOF
