#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC passes of the bench commands (tools/pmc_full.sh, run by tools/profile_round3.sh).

usage: python profiles/pmc_to_json.py <workload> <views> <note> main=<pmc dir> [in_flight=<pmc dir>] [path_10m=<pmc dir>] [configs2=<pmc dir>]

  main      passes of the default command with --in-flight 1: k_stream<false, true>, every launch alone (what
            roofline.traffic / valu / hbm_frac quote, over the one-frame-at-a-time launch duration)
  in_flight passes of the same command with RT_TUNE_STREAM_WAVES=8 --in-flight 1: k_stream8<false, true>, the kernel the TIMED
            region launches with >= 3 frames in flight (a counter pass serialises launches, so the kernel is forced and runs
            alone); bench.py divides its per-launch instruction and byte counts by the timed region's ms_per_step
            (roofline.in_flight)
  path_10m / configs2   passes of bench.py --workload soup_10m_1080p_path_1spp_8seg / soup_1m_1080p_4spp: the path kernels'
            counters quoted under extra.path_10m / extra.configs2

Records the averaged counters together with the FINGERPRINT of the kernel binaries they were collected on
(tools/kernel_resources.py: kernel symbols, registers, spills, scratch, LDS and the hash of each kernel's machine code).
bench.py quotes the counters only while the library it loads still has that fingerprint.  Run on the GPU box right after the
passes, so the library is the one that was profiled."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402

VALU_PEAK = 0.909e9      # profiles/calib/valu_mask on this pool: 1.127 ms for 2000 x 64 x 8 FMAs per SIMD (profiles/r2/valu_issue_calib.txt)
HBM_PEAK = 8000.0e9


def collect(root, kern):
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
    return c, dur


def derived(c, dur):
    """The figures bench.py quotes, from one kernel's averaged counters."""
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0        # the counter sums the 8 XCDs; 256 CUs x 4 SIMDs
    avg_s = sum(dur) / len(dur) * 1e-9 if dur else None
    hbm = int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0)
    return {
        "FETCH_SIZE_KB": round(c["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(c["WRITE_SIZE"], 1),
        "hbm_bytes_per_launch": hbm,
        "hbm_frac_alone": round(hbm / avg_s / HBM_PEAK, 4) if avg_s else None,
        # VALU issue against the SIMD's MEASURED capacity (profiles/calib/valu_mask: one wave64 instruction per ~2.5 cycles with 8
        # waves per SIMD); valu_issue_slots is the share of cycles in which SOME wave held an issue slot (rounds 1-2's "busy")
        "valu_busy": round(c["SQ_INSTS_VALU"] / 1024.0 / avg_s / VALU_PEAK, 3) if avg_s else None,
        "valu_issue_slots": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, 3),
        "wave_time": {"waiting_for_memory": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                      "issue_stalled": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                      "issuing": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)} if "SQ_WAIT_ANY" in c else None,
        "waves_per_simd_avg": round(c["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles, 2) if "SQ_WAVE_CYCLES" in c else None,
        "valu_lanes_active": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 3),
        "valu_insts_per_launch": int(c["SQ_INSTS_VALU"]),
        "vmem_rd_insts_per_launch": int(c.get("SQ_INSTS_VMEM_RD", 0)),
        "ta_busy": round(c["TA_BUSY_avr"] / (c["GRBM_GUI_ACTIVE"] / 8.0), 3) if "TA_BUSY_avr" in c else None,
        "ta_addr_stalled_by_tc_cycles": int(c["TA_ADDR_STALLED_BY_TC_CYCLES_sum"]) if "TA_ADDR_STALLED_BY_TC_CYCLES_sum" in c else None,
        "l2_hit_rate": round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 3) if "TCC_HIT_sum" in c else None,
        "l2_requests_per_launch": int(c["TCC_REQ_sum"]) if "TCC_REQ_sum" in c else None,
        "avg_duration_us_under_pmc": round(avg_s * 1e6, 1) if avg_s else None,
        "launches_averaged": len(dur),
    }


def main():
    workload, views, note = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dirs = dict(a.split("=", 1) for a in sys.argv[4:])
    c, dur = collect(dirs["main"], "k_stream<false, true>")
    out = {"workload": workload, "kernel": "k_stream<false, true>", "views": views, "pmc_source": note,
           "fetch_correction": 1.0,
           "correction_basis": "profiles/calib/gather_calib.hip (round 1, same pool): 64-B node / 48-B record gathers read FETCH_SIZE = bytes (factor 1.0); a 16-B/lane coalesced stream would need the guide's factor 2",
           "valu_peak_wave_insts_per_s_per_simd": VALU_PEAK}
    out.update(derived(c, dur))
    if "in_flight" in dirs:
        c8, d8 = collect(dirs["in_flight"], "k_stream8<false, true>")
        blk = derived(c8, d8)
        blk["kernel"] = "k_stream8<false, true>"
        blk["collected_with"] = ("RT_TUNE_STREAM_WAVES=8 --in-flight 1: the timed region's kernel forced, every launch alone (rocprofv3 --pmc serialises "
                                 "launches); instruction and byte counts per launch carry over to the timed region, durations do not")
        out["in_flight"] = blk
    for key, kerns in (("path_10m", ("k_path_start<false>", "k_path_bounce<false>")), ("configs2", ("k_path_start<false>",))):
        if key in dirs:
            sec = {}
            for k in kerns:
                ck, dk = collect(dirs[key], k)
                if ck:
                    sec[k] = derived(ck, dk)
            sec["collected_with"] = "rocprofv3 --pmc on bench.py --workload ... --in-flight 1 (every launch alone); one run per counter group"
            out[key] = sec
    out["fingerprint_kernels"] = "k_stream"
    out["fingerprint"] = kernel_resources.fingerprint(kernel_resources.DEFAULT_SO, "k_stream")
    out["fingerprint_path_kernels"] = "k_path"
    out["fingerprint_path"] = kernel_resources.fingerprint(kernel_resources.DEFAULT_SO, "k_path")
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("fingerprint")}, indent=1))


if __name__ == "__main__":
    main()
