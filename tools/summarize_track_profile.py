#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile_track.sh: the tracking correlator kernel alone under rocprofv3) into
profiles/<tag>_pmc_summary.json, <tag>_kernel_stats.csv and <tag>_kernel_us.json.
usage: summarize_track_profile.py <tag> [channels]"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# per channel-ms 3 x 2048 + 511 + 1023 lane-ops of the reference's formulation
LANE_OPS_PER_CHANNEL_MODEL = 3 * 2048 + 511 + 1023
VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9      # 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 3.93e13


def main(tag="r03_track", channels=212992, match="k_track_epl", ms_per_launch=1):
    """match: substring of the kernel's name; ms_per_launch: milliseconds of stream one launch serves (k_track_loop: K) --
    every per-channel figure below is per channel AND millisecond"""
    channels = int(channels) * int(ms_per_launch)
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    stats_csv = os.path.join(src, "trace", "trace_kernel_stats.csv")
    shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(stats_csv)) if match in r["Name"]]
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    full = top["Name"]
    out = {"tag": tag, "command": f"tools/bench_track_kernel.py {channels} under rocprofv3 (one --pmc pass per counter group, "
                                  "--kernel-trace only)",
           "kernel": full.split("(")[0], "channel_milliseconds_per_launch": channels, "ms_per_launch": int(ms_per_launch),
           "kernel_trace_avg_ns": float(top["AverageNs"]), "kernel_trace_min_ns": float(top["MinNs"]), "kernel_trace_max_ns": float(top["MaxNs"]),
           "kernel_trace_calls": int(top["Calls"]), "counters_avg_per_launch": {}}
    for name in sorted(os.listdir(src)):
        path = os.path.join(src, name, "pmc_counter_collection.csv")
        if not name.startswith("pmc_") or not os.path.exists(path):
            continue
        agg, meta = collections.defaultdict(list), {}
        for r in csv.DictReader(open(path)):
            if r["Kernel_Name"] == full:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")}
        for k, v in agg.items():
            out["counters_avg_per_launch"][k] = sum(v) / len(v)
        if meta:
            out["dispatch"] = meta
    c = out["counters_avg_per_launch"]
    d = {}
    if "SQ_INSTS_VALU" in c:
        d["valu_instructions_per_channel"] = c["SQ_INSTS_VALU"] / channels
        d["lane_ops_per_channel_issued"] = c["SQ_INSTS_VALU"] * 64 / channels
        d["lane_ops_per_channel_reference_formulation"] = LANE_OPS_PER_CHANNEL_MODEL
        t = out["kernel_trace_avg_ns"] * 1e-9
        d["algorithmic_lane_ops_per_s"] = LANE_OPS_PER_CHANNEL_MODEL * channels / t
        d["frac_of_valu_peak_algorithmic"] = d["algorithmic_lane_ops_per_s"] / VALU_PEAK_LANE_OPS
        d["frac_of_valu_peak_issued"] = c["SQ_INSTS_VALU"] * 64 / t / VALU_PEAK_LANE_OPS
    if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VALU" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        d["gpu_cycles_per_launch"] = cycles
        d["valu_issue_utilisation_4cycle_model"] = c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cycles
        d["effective_clock_ghz"] = cycles / out["kernel_trace_avg_ns"]
    if "SQ_INSTS_LDS" in c:
        d["lds_instructions_per_channel"] = c["SQ_INSTS_LDS"] / channels
    if "SQ_INSTS_VMEM_RD" in c:
        d["vmem_read_instructions_per_channel"] = c["SQ_INSTS_VMEM_RD"] / channels
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
        if int(ms_per_launch) > 1 or "loop" in match:   # k_track_loop: 100 B of live state in and out per channel, K flag bytes, K blocks
            n_real = channels // int(ms_per_launch)
            d["algorithmic_bytes_per_launch"] = n_real * (2 * 100 + int(ms_per_launch)) + 2046 * int(ms_per_launch)
        else:
            d["algorithmic_bytes_per_launch"] = channels * (16 + 4 + 12) + 2046
    d["round2"] = {"kernel_trace_avg_ns": 234458.8, "valu_instructions_per_channel": 523.0,
                   "valu_issue_utilisation_4cycle_model": 0.84, "source": "profiles/r02f_track_pmc_summary.json"}
    out["derived"] = d
    with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    live = os.path.join(ROOT, "gpurun_out", f"{tag}_kernel_us.json")
    if os.path.exists(live):
        shutil.copy(live, os.path.join(dst, f"{tag}_kernel_us.json"))
    print(json.dumps(out["derived"], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
