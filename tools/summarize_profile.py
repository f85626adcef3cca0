#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by tools/profile_bench.sh on the GPU box) into the small,
tracked files under profiles/:
  <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats, verbatim
  <tag>_pmc_summary.json    per-launch averages of every counter for the profiled kernel (name taken from the trace)
  kernel_counters.json      one entry per (kernel, searches per launch, blocks per search): SQ_INSTS_VALU and the HBM
                            bytes per launch -- what bench.py prices `roofline.achieved` / `roofline.traffic` from

usage: summarize_profile.py <tag> [searches_per_launch [n_ms [kernel-name-substring]]]
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short_name(full):
    """'void gpsx::k_acq_poly<8, 16, 0>(gpsx::AcqParams, ...)' -> 'k_acq_poly<8,16,0>' (what gpsx_last_kernel returns)"""
    m = re.search(r"gpsx::(\w+(?:<[^>]*>)?)", full)
    return m.group(1).replace(" ", "") if m else full


def main(tag="r02", searches=256, n_ms=1, pattern=None):
    searches, n_ms = int(searches), int(n_ms)
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats_csv = os.path.join(src, "trace", "trace_kernel_stats.csv")
    shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    # the profiled kernel: the one the trace spent most time in (or the one matching `pattern`)
    rows = list(csv.DictReader(open(stats_csv)))
    rows = [r for r in rows if (pattern in r["Name"] if pattern else "gpsx::k_" in r["Name"])]
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    full_name, kernel = top["Name"], short_name(top["Name"])
    summary = {"tag": tag, "command": "kernel trace: rocprofv3 --kernel-trace --stats -- python bench.py $TRACE_ARGS (default: the driver's "
                                       "own --steps 50 --warmup 5) --no-cpu-baseline --no-tracking --no-pcie --no-native $BENCH_ARGS; counters: "
                                       "the same with --steps 3 --warmup 1, one rocprofv3 --pmc pass per counter group, --kernel-trace only",
               "kernel": kernel, "kernel_full_name": full_name, "searches_per_launch": searches, "n_ms": n_ms,
               "kernel_trace_avg_ns": float(top["AverageNs"]), "kernel_trace_calls": int(top["Calls"]),
               "kernel_trace_min_ns": float(top["MinNs"]), "kernel_trace_max_ns": float(top["MaxNs"]),
               "kernel_trace_stddev_ns": float(top["StdDev"]),
               "counters_avg_per_launch": {}}
    # the median of the launches (the mean of a 20-launch trace moves with one slow launch): from the per-dispatch trace
    trace_csv = os.path.join(src, "trace", "trace_kernel_trace.csv")
    if os.path.exists(trace_csv):
        durs = sorted(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(trace_csv))
                      if r.get("Kernel_Name") == full_name)
        if durs:
            summary["kernel_trace_median_ns"] = durs[len(durs) // 2] if len(durs) % 2 else 0.5 * (durs[len(durs) // 2 - 1] + durs[len(durs) // 2])
    for name in sorted(os.listdir(src)):
        path = os.path.join(src, name, "pmc_counter_collection.csv")
        if not name.startswith("pmc_") or not os.path.exists(path):
            continue
        agg = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(path)):
            if r["Kernel_Name"] == full_name:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                          "Accum_VGPR_Count", "SGPR_Count")}
        for k, v in agg.items():
            summary["counters_avg_per_launch"][k] = sum(v) / len(v)
        if meta:
            summary["dispatch"] = meta
    c = summary["counters_avg_per_launch"]
    entry = {"kernel": kernel, "searches_per_launch": searches, "n_ms": n_ms, "world": 1,
             "source": f"profiles/{tag}_pmc_summary.json"}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts
        # 64 B per 128 B request for wide coalesced reads -> doubled here as the guide prescribes (an upper bound for the
        # 2-byte loads of a 2 KB block); WRITE_SIZE is taken as reported.
        fetch_b = c["FETCH_SIZE"] * 1024 * 2
        write_b = c["WRITE_SIZE"] * 1024
        summary["hbm_traffic"] = {"hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes_corrected": fetch_b,
                                  "write_bytes": write_b, "fetch_kib_raw": c["FETCH_SIZE"], "write_kib_raw": c["WRITE_SIZE"]}
        entry["hbm_bytes_per_launch"] = fetch_b + write_b
    # the kernel trace's own figures: what bench.py's roofline.frac is computed from (mean launch of THIS csv)
    entry["kernel_trace_avg_ns"] = summary["kernel_trace_avg_ns"]
    entry["kernel_trace_calls"] = summary["kernel_trace_calls"]
    entry["trace_source"] = f"profiles/{tag}_kernel_stats.csv"
    for k in ("kernel_trace_median_ns", "kernel_trace_min_ns", "kernel_trace_max_ns"):
        if k in summary:
            entry[k] = summary[k]
    if "SQ_INSTS_VALU" in c:
        entry["SQ_INSTS_VALU"] = c["SQ_INSTS_VALU"]
        hyp = searches * n_ms * 32 * 21 * 16368
        summary["derived"] = {"hypotheses_per_launch": hyp, "valu_lane_ops_per_hypothesis": c["SQ_INSTS_VALU"] * 64 / hyp}
        if "GRBM_GUI_ACTIVE" in c:
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
            summary["derived"].update({
                "gpu_cycles_per_launch": cycles,
                "valu_issue_cycles_per_simd": c["SQ_INSTS_VALU"] * 4.0 / 1024.0,   # 4 cycles per wave64 int op, 1024 SIMDs
                "valu_issue_utilisation": c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cycles})
        for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA"):
            if k in c:
                entry[k] = c[k]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            summary["derived"]["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
        if "GRBM_GUI_ACTIVE" in c:
            # the clock the chip actually ran this kernel at (it follows the power budget): shader cycles / kernel time
            summary["derived"]["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / summary["kernel_trace_avg_ns"]
            entry["gpu_cycles_per_launch"] = c["GRBM_GUI_ACTIVE"] / 8.0
            entry["kernel_trace_avg_ns"] = summary["kernel_trace_avg_ns"]
            for k in ("kernel_trace_median_ns", "kernel_trace_min_ns", "kernel_trace_max_ns"):
                if k in summary:
                    entry[k] = summary[k]
    with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    kc_path = os.path.join(dst, "kernel_counters.json")
    table = json.load(open(kc_path)) if os.path.exists(kc_path) else []
    table = [e for e in table if (e["kernel"], e["searches_per_launch"], e["n_ms"]) != (kernel, searches, n_ms)]
    table.append(entry)
    with open(kc_path, "w") as f:
        json.dump(table, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(*(sys.argv[1:5] or ["r02"]))
