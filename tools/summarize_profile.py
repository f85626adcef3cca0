#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by tools/profile_bench.sh on the GPU box) into the small,
tracked files under profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
<tag>_pmc_summary.json (per-launch averages of every counter for the dominant kernel) and hbm_traffic.json
(what bench.py reports as roofline.traffic)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag="r01", searches=64, n_ms=1):
    searches, n_ms = int(searches), int(n_ms)   # n_ms > 1: a side profile, profiles/hbm_traffic.json is left alone
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
    summary = {"tag": tag, "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline (one rocprofv3 --pmc pass "
                                       "per counter group, --kernel-trace only)", "kernel": "gpsx::k_acq<8,false,ALGO> (the acquisition grid kernel of the run)",
               "counters_avg_per_launch": {}}
    for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        path = os.path.join(src, name, "pmc_counter_collection.csv")
        if not os.path.exists(path):
            continue
        agg = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(path)):
            if "k_acq<" in r["Kernel_Name"] or "k_acq_poly<" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                          "Accum_VGPR_Count", "SGPR_Count")}
        for k, v in agg.items():
            summary["counters_avg_per_launch"][k] = sum(v) / len(v)
        summary["dispatch"] = meta
    c = summary["counters_avg_per_launch"]
    with open(os.path.join(src, "trace", "trace_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            if "k_acq<" in r["Name"] or "k_acq_poly<" in r["Name"]:
                summary["kernel_trace_avg_ns"] = float(r["AverageNs"])
                summary["kernel_trace_calls"] = int(r["Calls"])
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts
        # 64 B per 128 B request for wide coalesced reads -> doubled here as the guide prescribes (upper bound for this
        # kernel, whose reads are 2-byte loads of a 2 KB block); WRITE_SIZE is taken as reported.
        fetch_b = c["FETCH_SIZE"] * 1024 * 2
        write_b = c["WRITE_SIZE"] * 1024
        tr = {"searches_per_launch": searches, "hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes_corrected": fetch_b,
              "write_bytes": write_b, "fetch_kib_raw": c["FETCH_SIZE"], "write_kib_raw": c["WRITE_SIZE"],
              "source": f"profiles/{tag}_pmc_summary.json"}
        if n_ms == 1:
            with open(os.path.join(dst, "hbm_traffic.json"), "w") as f:
                json.dump(tr, f, indent=1)
        else:
            tr["blocks_per_search"] = n_ms
        summary["hbm_traffic"] = tr
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
        summary["derived"] = {
            "gpu_cycles_per_launch": cycles,
            "valu_issue_cycles_per_simd": c["SQ_INSTS_VALU"] * 4.0 / 1024.0,   # 4 cycles per wave64 int op, 1024 SIMDs
            "valu_issue_utilisation": c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cycles,
            # main-loop instructions the formulation needs: workgroups x 4 waves x 2 halves x steps x 64
            "main_loop_wave_instructions_sad": n_ms * searches * 672 * 4 * 2 * 256 * 64,
            "main_loop_wave_instructions_dot8": n_ms * searches * 672 * 4 * 2 * 128 * 64,
            # polyphase kernel: per (search, Doppler, PRN group) 4 waves x (one direct offset of 128 x 64 dots
            # + 15 recurrence offsets of 32 words x 64 (and + bcnt) pairs)
            "main_loop_wave_instructions_poly": n_ms * searches * 21 * 4 * 4 * (128 * 64 + 15 * 32 * 128),
        }
        d = summary["derived"]
        d["sad_share_of_valu"] = d["main_loop_wave_instructions_sad"] / c["SQ_INSTS_VALU"]
        d["dot8_share_of_valu"] = d["main_loop_wave_instructions_dot8"] / c["SQ_INSTS_VALU"]
        d["poly_share_of_valu"] = d["main_loop_wave_instructions_poly"] / c["SQ_INSTS_VALU"]
    with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(*(sys.argv[1:4] or ["r01"]))   # tag [searches_per_launch [n_ms]]
