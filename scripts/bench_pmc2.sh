#!/bin/bash
# Issue-mix / stall counters of the kNN kernel inside bench.py (dev tool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/kpmc2_${TAG:-x}_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/kpmc2_${TAG:-x}_$i.log 2>&1
done
python3 - <<EOF
import csv, glob
for d in sorted(glob.glob("$R/gpurun_out/kpmc2_${TAG:-x}_*")):
    f = [d] if d.endswith(".csv") else sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=__import__("os").path.getmtime)[-1:]
    if not f: continue
    rows = [r for r in csv.DictReader(open(f[0])) if "knn_mfma_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) > 500000]
    by = {}
    for r in rows: by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in by.items(): print(k, "%.4g" % v[-1], "(n=%d)" % len(v))
EOF
