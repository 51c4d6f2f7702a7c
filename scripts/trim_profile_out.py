#!/usr/bin/env python3
"""Run on the GPU box after scripts/profile_round.sh / traffic_pass.sh: gpurun returns at most 64 MiB of gpurun_out/.  Keeps, of every --pmc pass, only the counter rows of
the project's own kernels (what scripts/collect_profile.py reads) and drops the passes' kernel traces and agent tables; the --stats run is left whole."""
import csv, glob, os, sys
tag = sys.argv[1]
keep = ("knn_", "descriptor", "render_", "adam_kernel", "preprocess", "tile_", "emit_", "radix_sort", "ssim_")
for d in glob.glob(os.path.join("gpurun_out", f"{tag}*pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*"), recursive=True):
        if os.path.isdir(f):
            continue
        if f.endswith("counter_collection.csv"):
            rows = list(csv.DictReader(open(f)))
            if rows:
                cols = list(rows[0].keys())
                with open(f, "w", newline="") as o:
                    w = csv.DictWriter(o, fieldnames=cols); w.writeheader()
                    for r in rows:
                        if any(k in r["Kernel_Name"] for k in keep):
                            w.writerow(r)
        else:
            os.remove(f)

# the long passes (late phase, rasteriser): keep the --stats summary and, of the kernel trace, only the launches of the dominant kernels (start, end, name cut)
for d in glob.glob(os.path.join("gpurun_out", f"{tag}late_stats")) + glob.glob(os.path.join("gpurun_out", f"{tag}raster_stats")):
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("knn_mfma", "knn_select", "knn_tile_prune", "render_", "tile_sort", "radix_sort"))]
        with open(f, "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"]); w.writeheader()
            for r in rows:
                w.writerow({"Kernel_Name": r["Kernel_Name"][:80], "Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"]})
    for f in glob.glob(os.path.join(d, "**", "*"), recursive=True):
        if os.path.isfile(f) and not (f.endswith("kernel_trace.csv") or f.endswith("kernel_stats.csv")):
            os.remove(f)
