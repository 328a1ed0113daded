#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC passes of one bench command (tools/pmc_full.sh).

usage: python profiles/pmc_to_json.py <pmc dir> "<kernel name as rocprofv3 prints it>" <workload> <source note>

Records the averaged counters behind roofline.traffic / roofline.valu of bench.py together with the FINGERPRINT of the
kernel binary they were collected on (tools/kernel_resources.py: kernel symbols, registers, spills, scratch, LDS and the
hash of the device ELF inside librt_amd.so).  bench.py quotes the counters only while the library it loads still has
that fingerprint.  Run on the GPU box right after the passes, so the library is the one that was profiled."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402

root, kern, workload, note = sys.argv[1:5]
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in agg.items()}
dur = []
for f in glob.glob(os.path.join(root, "*", "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
VALU_PEAK = 0.909e9      # profiles/calib/valu_mask on this pool: 1.127 ms for 2000 x 64 x 8 FMAs per SIMD (gpurun_out -> profiles/r2/valu_issue_calib.txt)
simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0        # the counter sums the 8 XCDs; 256 CUs x 4 SIMDs
out = {
    "workload": workload, "kernel": kern,
    "pmc_source": note,
    "FETCH_SIZE_KB": round(c["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(c["WRITE_SIZE"], 1),
    "fetch_correction": 1.0,
    "correction_basis": "profiles/calib/gather_calib.hip (round 1, same pool): 64-B node / 48-B record gathers read FETCH_SIZE = bytes (factor 1.0); a 16-B/lane coalesced stream would need the guide's factor 2",
    "hbm_bytes_per_launch": int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0),
    # VALU issue against the SIMD's MEASURED capacity: profiles/calib/valu_mask (8 waves per SIMD, independent FMAs) retires
    # VALU_PEAK wave64 instructions per second and SIMD -- one per ~2.5 cycles: a SIMD-32 data path, two waves' 4-cycle issue
    # slots interleaved (MI355X guide, row 'v_fma_f32 (wave64)').  Rounds 1 and early 2 charged 4 cycles per instruction and
    # called the kernel VALU-bound at 0.91; that figure is the share of cycles in which SOME wave held an issue slot
    # (valu_issue_slots, capacity ~1.5 concurrent slots), not the pipe's utilisation.
    "valu_busy": round(c["SQ_INSTS_VALU"] / 1024.0 / (sum(dur) / len(dur) * 1e-9) / VALU_PEAK, 3) if dur else None,
    "valu_peak_wave_insts_per_s_per_simd": VALU_PEAK,
    "valu_issue_slots": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, 3),
    # where a resident wave's time goes (SQ_WAIT_ANY: parked at s_waitcnt, i.e. on memory; SQ_WAIT_INST_ANY: ready but not
    # issued; SQ_ACTIVE_INST_ANY: issuing), and how many waves a SIMD holds on average over the launch
    "wave_time": {"waiting_for_memory": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                  "issue_stalled": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                  "issuing": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)} if "SQ_WAIT_ANY" in c else None,
    "waves_per_simd_avg": round(c["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles, 2) if "SQ_WAVE_CYCLES" in c else None,
    "valu_lanes_active": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 3),
    "valu_insts_per_launch": int(c["SQ_INSTS_VALU"]),
    "ta_busy": round(c["TA_BUSY_avr"] / (c["GRBM_GUI_ACTIVE"] / 8.0), 3) if "TA_BUSY_avr" in c else None,
    "l2_hit_rate": round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 3) if "TCC_HIT_sum" in c else None,
    "avg_duration_us_under_pmc": round(sum(dur) / len(dur) / 1e3, 1) if dur else None,
    "fingerprint_kernels": "k_stream",
    "fingerprint": kernel_resources.fingerprint(kernel_resources.DEFAULT_SO, "k_stream"),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "fingerprint"}, indent=1))
