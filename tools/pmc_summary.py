#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (one or more pass directories) into a JSON summary:
mean counter value per launch for every kernel whose name contains --kernel.
Usage: python tools/pmc_summary.py --kernel rts_smooth_lane_kernel gpurun_out/pmcA gpurun_out/pmcB ... > profiles/x.json"""
import argparse, csv, glob, json, os, collections

ap = argparse.ArgumentParser()
ap.add_argument("--kernel", required=True)
ap.add_argument("--note", default="")
ap.add_argument("dirs", nargs="+")
a = ap.parse_args()
out = {"kernel_filter": a.kernel, "note": a.note, "counters": {}}
for d in a.dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))  # counter -> dispatch -> sum over dims
        names = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if a.kernel not in row["Kernel_Name"]:
                    continue
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
        for c, disp in per.items():
            vals = list(disp.values())
            out["counters"][c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
        if names:
            out["kernel_name"] = sorted(set(names.values()))[0][:160]
print(json.dumps(out, indent=1))
