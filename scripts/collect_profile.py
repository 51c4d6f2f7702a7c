#!/usr/bin/env python3
"""Copies the judged summaries of a scripts/profile_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench command (kernel names cut at 120 chars)
   profiles/<tag>_bench_line.json    the JSON line that run printed
   profiles/<tag>_pmc/*.csv          per-dispatch counters of the project's own kernels, one file per --pmc pass
usage: scripts/collect_profile.py <tag>"""
import csv, glob, json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# --out DIR: write there instead of profiles/ (scripts/profile_round.sh collects ON the GPU box — the raw traces exceed what gpurun copies back — into gpurun_out/<tag>_profiles)
go, pr = os.path.join(root, "gpurun_out"), (sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(root, "profiles"))
os.makedirs(pr, exist_ok=True)
st = max(glob.glob(os.path.join(go, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # newest run of that tag
rows = list(csv.DictReader(open(st)))
with open(os.path.join(pr, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader()
    for r in rows:
        r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
# every launch of the dominant kernel in that run, in order (the --stats average includes the warm-up steps, the first of which sweeps with cold
# bounds; bench.py's roofline averages the timed steps only: compare with the rows after the first 2 x warm-up launches)
tr = max(glob.glob(os.path.join(go, f"{tag}_stats", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
kl = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr)) if any(n in r["Kernel_Name"] for n in ("knn_mfma_kernel", "knn_mfma_hot_kernel", "knn_mfma_persistent_kernel", "knn_emit_kernel")))
with open(os.path.join(pr, f"{tag}_knn_launches.csv"), "w") as f:
    f.write("launch,duration_ms\n")
    for i, (a, b) in enumerate(kl):
        f.write(f"{i},{(b - a) / 1e6:.3f}\n")
line = [l for l in open(os.path.join(go, f"{tag}_bench.log")) if l.startswith("{")][-1]
open(os.path.join(pr, f"{tag}_bench_line.json"), "w").write(line)
os.makedirs(os.path.join(pr, f"{tag}_pmc"), exist_ok=True)
keep = ("knn_mfma", "knn_emit", "knn_straggler", "knn_select", "knn_plan", "descriptor", "render_", "adam_kernel", "preprocess", "knn_merge", "knn_prev_threshold", "knn_tile_order", "knn_tile_prune", "knn_motion", "tile_sort", "radix_sort", "emit_", "tile_hist")
cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "Counter_Name", "Counter_Value"]
for d in sorted(glob.glob(os.path.join(go, f"{tag}_pmc_*"))):
    if not os.path.isdir(d):
        continue
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not cc:
        continue
    rows = [r for r in csv.DictReader(open(max(cc, key=os.path.getmtime))) if any(k in r["Kernel_Name"] for k in keep)]
    name = os.path.basename(d)[len(tag) + 5:].lower() + ".csv"
    with open(os.path.join(pr, f"{tag}_pmc", name), "w", newline="") as f:
        cs = [c for c in cols if c in rows[0]]
        w = csv.DictWriter(f, fieldnames=cs); w.writeheader()
        for r in rows:
            r = dict(r); r["Kernel_Name"] = r["Kernel_Name"][:100]; w.writerow({c: r[c] for c in cs})
import subprocess
try:
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "wast3d_amd", "bench.py"], capture_output=True, text=True).stdout.strip()
    open(os.path.join(pr, f"{tag}_pmc", "HEAD"), "w").write(head + (" + uncommitted changes" if dirty else "") + "\n")
except Exception:
    pass
print("ok", tag)
