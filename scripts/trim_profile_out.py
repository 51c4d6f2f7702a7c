#!/usr/bin/env python3
"""Run on the GPU box after scripts/profile_round.sh / traffic_pass.sh: gpurun returns at most 64 MiB of gpurun_out/.  Keeps, of every --pmc pass, only the counter rows of
the project's own kernels (what scripts/collect_profile.py reads) and drops the passes' kernel traces and agent tables; the --stats run is left whole."""
import csv, glob, os, sys
tag = sys.argv[1]
keep = ("knn_", "descriptor", "render_", "adam_kernel", "preprocess")
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
