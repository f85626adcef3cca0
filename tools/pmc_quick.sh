#!/bin/bash
# Quick PMC look at one kernel of the bench: tools/pmc_quick.sh <kernel-name-substring> [bench args...]
# (separate passes, --kernel-trace only, as the pool's rules require; prints per-launch averages)
set -u
PAT=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# ($PMC_CMD: another command than the bench, e.g. PMC_CMD="python $REPO/tools/bench_native_grid.py --steps 3 --warmup 1")
CMD=${PMC_CMD:-"python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-pcie --no-native $*"}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/a -o pmc -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU --output-format csv -d $OUT/b -o pmc -- $CMD > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/c -o pmc -- $CMD > $OUT/c.log 2>&1
tail -3 $OUT/c.log
python - "$PAT" $OUT <<'PY'
import csv, collections, glob, sys
pat, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
meta = None
for path in glob.glob(out + "/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"], r["Grid_Size"])
for k in sorted(agg):
    print(f"{k:28s} {sum(agg[k]) / len(agg[k]):.5g}")
if agg:
    print("vgpr/agpr/scratch/lds/grid", meta)
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VALU" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        print(f"gpu cycles {cyc:.4g}  valu insts x4 / simd-cycles {c['SQ_INSTS_VALU'] * 4 / 1024 / cyc:.3f}  salu/valu {c['SQ_INSTS_SALU'] / c['SQ_INSTS_VALU']:.2f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print(f"mfma busy / simd-cycles {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:.3f}")
    import json, os
    if os.environ.get("PMC_JSON"):
        json.dump({"kernel_substring": pat, "command": os.environ.get("PMC_CMD", "bench.py --steps 3 --warmup 1"),
                   "counters_avg_per_launch": c, "dispatch": dict(zip(("vgpr", "agpr", "scratch", "lds", "grid"), meta)),
                   "derived": {"gpu_cycles_per_launch": c.get("GRBM_GUI_ACTIVE", 0) / 8,
                               "mfma_busy_frac": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8))
                               if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c else None,
                               "valu_issue_utilisation": (c["SQ_INSTS_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8))
                               if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c else None}},
                  open(os.environ["PMC_JSON"], "w"), indent=1)
PY
