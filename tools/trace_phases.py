#!/usr/bin/env python3
"""Splits the launches of one kernel in a rocprofv3 --kernel-trace CSV into those that ran ALONE and those that shared
the GPU with another launch of the same kernel (frames in flight): count, average and median begin-to-end duration.
bench.py's default command contains both kinds (the timed region with frames in flight, then the same frames one at a
time), and `rocprofv3 --stats` averages over both; this is the per-kind figure bench.py's roofline.kernel_ms (alone) and
roofline.kernel_ms_in_flight must agree with.

    python tools/trace_phases.py <dir or *_kernel_trace.csv> "<kernel substring>"
"""
import csv, glob, os, statistics, sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"]]
    rows.sort()
    alone, shared = [], []
    for i, (b, e) in enumerate(rows):
        overl = any(b2 < e and e2 > b for j, (b2, e2) in enumerate(rows[max(0, i - 8):i + 9], max(0, i - 8)) if j != i)
        (shared if overl else alone).append((e - b) / 1e3)
    print(f"{path}\nkernel contains: {sub}\nlaunches: {len(rows)}")
    for name, v in (("alone (one frame at a time)", alone), ("sharing the GPU with another launch (frames in flight)", shared)):
        if v:
            print(f"  {name}: {len(v)} launches, average {statistics.mean(v):.1f} us, median {statistics.median(v):.1f} us, min {min(v):.1f}, max {max(v):.1f}")
    if rows:
        span = (rows[-1][1] - rows[0][0]) / 1e3
        print(f"  all: average {statistics.mean([(e - b) / 1e3 for b, e in rows]):.1f} us (what --stats prints); first start to last end {span:.0f} us")


if __name__ == "__main__":
    main()
