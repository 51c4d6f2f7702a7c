#!/bin/bash
# Stall / issue-mix counters of the emit sweep in the C3 loop (dev tool): separate --pmc passes over a short bench run; per counter the median over the launches.
# usage: scripts/emit_pmc.sh <tag>
tag=${1:-epmc}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
FLAGS="--steps 12 --warmup 6 --no-cpu-baseline --whole-run 0 --no-side-lines"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_BRANCH SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_$i -- python3 $R/bench.py $FLAGS > $R/gpurun_out/${tag}_$i.log 2>&1
done
find $R/gpurun_out -name "*.db" -delete
python3 - <<PY
import csv, glob
out = {}
durs = []
for d in sorted(glob.glob("$R/gpurun_out/${tag}_[0-9]*")):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_emit_kernel" in r["Kernel_Name"]: durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        by = {}
        for r in csv.DictReader(open(f)):
            if "knn_emit_kernel" in r["Kernel_Name"]: by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in by.items():
            v = sorted(v[6:] or v); out[k] = (v[len(v) // 2], len(v))
durs = sorted(durs)
print("knn_emit_kernel: median launch %.1f us over %d launches (under the counters)" % (durs[len(durs) // 2] if durs else 0, len(durs)))
for k in sorted(out): print("%-32s %.5g  (n=%d)" % (k, out[k][0], out[k][1]))
PY
rm -rf $R/gpurun_out/${tag}_[0-9]*/ $R/gpurun_out/${tag}_[0-9]*.log
