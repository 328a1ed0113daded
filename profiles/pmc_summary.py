#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_*/<pass>/**/*_counter_collection.csv) per kernel."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_primary<false>"
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
for k, v in out.items():
    print(f"{k:28s} {v:18.1f}  (n={len(agg[k])})")
dur = []
for f in glob.glob(os.path.join(root, "*", "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if dur:
    print(f"avg duration under PMC: {sum(dur)/len(dur)/1e3:.1f} us over {len(dur)} dispatches")
