#!/bin/bash
# Instruction mix of the COLD emit sweep next to the WARM one (same rows, one item per row block), BASELINE config C2's shape: separate --pmc passes (dev tool).
# usage: scripts/cold_pmc.sh <tag>      -> gpurun_out/<tag>.log
tag=${1:-cpmc}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/${tag}_$i -- python3 $R/scripts/cold_pmc_run.py > /tmp/${tag}_$i.log 2>&1
done
python3 - > $R/gpurun_out/$tag.log <<PY
import csv, glob
out = {}; durs = {}
for d in sorted(glob.glob("/tmp/${tag}_[0-9]*")):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_emit_kernel" in r["Kernel_Name"] or "knn_select" in r["Kernel_Name"]:
                durs.setdefault(r["Kernel_Name"][:100], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_emit_kernel" in r["Kernel_Name"]: out.setdefault((r["Kernel_Name"][:100], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k, v in sorted(durs.items()):
    v = sorted(v); print("%-96s median launch %.1f us (n=%d)" % (k, v[len(v) // 2], len(v)))
for k in sorted(out):
    v = sorted(out[k]); print("%-96s %-28s %.5g (n=%d)" % (k[0], k[1], v[len(v) // 2], len(v)))
PY
cat $R/gpurun_out/$tag.log
