#!/usr/bin/env python3
"""tools/microbench/valu_rates (run on the GPU box) -> profiles/valu_class_rates.json: the issue rates bench.py prices the
vector-ALU grid kernel's instruction stream against (bench.py _poly_valu_roofline).

usage: valu_class_rates.py <microbench output .txt> <tag>      (copies the text to profiles/<tag>_valu_rates_microbench.txt)
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOUR_CYCLE = ("v_max_u32", "v_bfe_u32", "v_lshl_or_b32", "v_add3_u32", "v_alignbit_b32", "v_perm_b32", "v_mad_i32_i24",
              "v_cvt_f32_i32", "v_mov_b32_dpp row_shr", "v_bcnt_u32_b32 (acc)")


def main(txt, tag):
    rates = {}
    for line in open(txt):
        m = re.match(r"^(.*?)\s+([\d.]+) ms\s+([\d.]+) Gstmt-lanes/s\s+\(x(\d) ops =>\s+([\d.]+) Glane-ops/s", line)
        if m:
            rates[m.group(1).strip()] = float(m.group(5)) / 1e3          # T lane-ops/s
    pair = rates["v_and(sgpr)+v_bcnt(acc)"]
    four = [rates[k] for k in FOUR_CYCLE if k in rates]
    dst = os.path.join(ROOT, "profiles", f"{tag}_valu_rates_microbench.txt")
    if os.path.abspath(txt) != dst:
        shutil.copy(txt, dst)
    out = {"and_bcnt_pair_tlane_ops": pair,
           "four_cycle_class_tlane_ops": max(four),      # the fastest member: a peak is an upper bound
           "four_cycle_class_members": {k: rates[k] for k in FOUR_CYCLE if k in rates},
           "two_cycle_class_tlane_ops": {k: rates[k] for k in ("v_and_b32", "v_add_u32", "v_mul_f32", "v_fma_f32") if k in rates},
           "model_4_cycle_tlane_ops": 256 * 64 * 2.4e9 / 1e12,
           "source": f"profiles/{tag}_valu_rates_microbench.txt",
           "note": "tools/microbench/valu_rates.hip on one MI355X: 64 lanes per wave instruction; the rates are at the clock the "
                   "chip sustains under each stream (the 4-cycle classes land at 0.92 of the 2.4 GHz model)"}
    with open(os.path.join(ROOT, "profiles", "valu_class_rates.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
