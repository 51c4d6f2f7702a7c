#!/bin/bash
# PMC passes over the fused backward alone (scripts/sum_cache_time.py): usage scripts/bwd_pmc.sh <tag>
tag=${1:-bwd}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  t=$(echo $grp | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc_$t -- python3 $R/scripts/sum_cache_time.py 8 > $R/gpurun_out/${tag}_pmc_$t.log 2>&1
done
find $R/gpurun_out -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/${tag}_pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "descriptor_bwd_window" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(k, "mean per launch %.4g over %d launches" % (sum(v) / len(v), len(v)))
PY
