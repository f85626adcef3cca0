#!/usr/bin/env python3
"""Repeat the pipelined byte-phase grid (k_acq_mx<4>) on rotating data sets and compare EVERY launch's triplets and keys with
what the direct 4-bit-dot-product kernel gave for that data set: the pipeline hands LDS buffers from piece to stage by barrier
count alone, and a slip there would show as a rare mismatch, not as a failed test.  Prints one JSON line."""
import argparse
import json
import os
os.environ.setdefault("GPSX_USE_LAB_LIBRARY", "1")   # forced kernel forms ($GPSX_ACQ_*): the lab build of the library
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=2000)
    ap.add_argument("--searches", type=int, nargs="+", default=[256, 37, 9])
    ap.add_argument("--sets", type=int, default=4)
    args = ap.parse_args()
    from stm32f4_sdr_gps_amd import capi, synth
    prns = np.arange(1, 33, dtype=np.uint8)
    os.environ["GPSX_ACQ_ALGO"] = "dot8"
    ref = capi.Engine(0)
    del os.environ["GPSX_ACQ_ALGO"]
    eng = capi.Engine(0)
    t0 = time.time()
    out = {"launches": 0, "mismatching_launches": 0, "by_launch_size": {}}
    for n in args.searches:
        kw = dict(n_search=n, dopp_min_hz=-7000, dopp_step_hz=500, n_dopp=29, phase_mode=capi.PHASES_BYTE)
        sets = []
        for k in range(args.sets):
            blocks = synth.cold_start_block(n, seed=900 + 17 * k + n, amp_scale=0.25 + 0.25 * (k & 1), two_bit=bool(k & 2))
            e = ref if not (k & 2) else None
            if k & 2:   # 2-bit captures: the reference engine reads the same sign plane from the 1-bit form
                one = synth.cold_start_block(n, seed=900 + 17 * k + n, amp_scale=0.25 + 0.25 * (k & 1))
                want = ref.acq_grid(one, prns, **kw)
            else:
                want = ref.acq_grid(blocks, prns, **kw)
            sets.append((blocks, bool(k & 2), want))
        assert ref.lib.gpsx_last_kernel(ref.h).startswith(b"k_acq<8,false,dot8>")
        bad = 0
        per = max(1, args.launches // len(args.searches))
        for i in range(per):
            blocks, two, want = sets[i % len(sets)]
            eng.set_if_format(capi.IF_2BIT_SM if two else capi.IF_1BIT)
            pk, keys = eng.acq_grid(blocks, prns, **kw)
            if not (np.array_equal(pk, want[0]) and np.array_equal(keys, want[1])):
                bad += 1
        assert eng.lib.gpsx_last_kernel(eng.h) == b"k_acq_mx<4>"
        out["by_launch_size"][str(n)] = {"launches": per, "mismatching": bad}
        out["launches"] += per
        out["mismatching_launches"] += bad
    out["seconds"] = time.time() - t0
    print(json.dumps(out))
    sys.exit(1 if out["mismatching_launches"] else 0)


if __name__ == "__main__":
    main()
