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

# derived figures (MI355X: 256 CUs x 4 SIMDs; the counters are sums over the 8 XCDs, a wave64 VALU instruction occupies its SIMD 4 cycles)
if "SQ_ACTIVE_INST_VALU" in out and "GRBM_GUI_ACTIVE" in out and out["GRBM_GUI_ACTIVE"] > 0:
    simd_cycles = out["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    print(f"derived valu_busy            {out['SQ_ACTIVE_INST_VALU'] * 4.0 / simd_cycles:18.3f}")
    if "SQ_THREAD_CYCLES_VALU" in out:
        print(f"derived valu_lanes_active    {out['SQ_THREAD_CYCLES_VALU'] / (64.0 * out['SQ_ACTIVE_INST_VALU']):18.3f}")
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    print(f"derived hbm_bytes_per_launch {(out['FETCH_SIZE'] + out['WRITE_SIZE']) * 1024.0:18.0f}")
