#!/bin/bash
# Instruction mix and issue utilisation of every project kernel in the default loop (two --pmc passes over a short bench run): usage scripts/inst_mix_pmc.sh <tag>
tag=${1:-mix}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
FLAGS="--steps 12 --warmup 6 --no-cpu-baseline --whole-run 0 --no-side-lines"
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  t=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc_$t -- python3 $R/bench.py $FLAGS > $R/gpurun_out/${tag}_pmc_$t.log 2>&1
done
find $R/gpurun_out -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("$R/gpurun_out/${tag}_pmc_*")):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, v in dur.items():
    if len(v) < 10: continue
    us = sorted(v)[len(v) // 2]
    c = {n: sorted(x)[len(x) // 2] for n, x in cnt[k].items()}
    if not c.get("SQ_INSTS_VALU"): continue
    valu_busy = c["SQ_INSTS_VALU"] * 4 / (us * 1e-6 * 2.4e9 * 1024)
    rows.append((us * len(v), k, us, len(v), c, valu_busy))
for _, k, us, n, c, vb in sorted(rows, reverse=True)[:18]:
    print(f"{k:60s} {us:8.1f} us x{n:4d}  VALU {c.get('SQ_INSTS_VALU', 0):.3g} SALU {c.get('SQ_INSTS_SALU', 0):.3g} LDS {c.get('SQ_INSTS_LDS', 0):.3g} VMEM rd {c.get('SQ_INSTS_VMEM_RD', 0):.3g} wr {c.get('SQ_INSTS_VMEM_WR', 0):.3g}"
          f"  VALU issue {vb:.2f} of the SIMD clocks, waiting {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f} of the wave clocks, LDS conflict cycles {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g}")
PY
rm -rf $R/gpurun_out/${tag}_pmc_*/
