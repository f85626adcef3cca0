#!/usr/bin/env python3
"""k_acq_weighted alone (the weighted two-bit acquisition extension; device-resident captures, HIP events on the engine's stream):
searches x 32 PRN x 21 Doppler x 16368 phases per launch.  usage: bench_weighted_kernel.py [searches [reps]]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from stm32f4_sdr_gps_amd import capi, synth
    searches = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    eng = capi.Engine(0)
    blocks = synth.cold_start_block(searches, seed=11, amp_scale=0.25, two_bit=True)
    prns = np.arange(1, 33, dtype=np.uint8)
    g = capi.AcqWeightedT(searches, 1, 32, prns.ctypes.data_as(C.POINTER(C.c_uint8)), -5000, 500, 21, 1)
    d_if = eng.malloc(blocks.size + 2)
    eng.h2d(d_if, np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
    d_pk = eng.malloc(searches * 32 * 21 * 16)

    def run():
        rc = eng.lib.gpsx_acq_grid_weighted_dev(eng.h, C.byref(g), C.c_void_p(d_if), searches, C.c_void_p(d_pk))
        assert rc == 0, eng.lib.gpsx_last_error(eng.h)
    for _ in range(2):
        run()
    e0, e1 = eng.event(), eng.event()
    eng.record(e0)
    for _ in range(reps):
        run()
    eng.record(e1)
    eng.synchronize()
    ms = eng.elapsed_ms(e0, e1) / reps
    hyp = searches * 32 * 21 * 16368
    dot4 = hyp * 2 * 256            # per hypothesis two streams x 256 four-chip steps
    print(json.dumps({"kernel": "gpsx::" + eng.lib.gpsx_last_kernel(eng.h).decode(), "searches": searches, "ms": round(ms, 3), "hyp_per_s": hyp / (ms * 1e-3),
                      "dot4_lane_ops_per_s": dot4 / (ms * 1e-3),
                      "frac_of_valu_issue_peak": dot4 / (ms * 1e-3) / (256 * 64 * 2.4e9),
                      "note": "algorithmic v_dot4_i32_i8 lane-ops (512 per hypothesis) against one wave64 op per 4 cycles per SIMD at 2.4 GHz"}))


if __name__ == "__main__":
    main()
