#!/usr/bin/env python3
"""The grid kernel alone (device-resident captures, HIP events on the engine's stream), no result checks: for timing
ablations ($GPSX_MX_EXPERIMENT) and A/B runs ($GPSX_ACQ_ALGO).  tools/bench_grid_kernel.py [searches [n_ms [reps]]]"""
import ctypes as C
import json
import os
os.environ.setdefault("GPSX_USE_LAB_LIBRARY", "1")   # forced kernel forms ($GPSX_ACQ_*): the lab build of the library
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from stm32f4_sdr_gps_amd import capi, synth
    searches = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n_ms = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    if os.environ.get("GPSX_LIB"):   # A/B runs against another build of the library
        capi.LIB_PATH = capi.LAB_LIB_PATH = os.environ["GPSX_LIB"]   # (tools/build_variant.sh: a lab build)
    eng = capi.Engine(0)
    blocks = synth.cold_start_block(searches * n_ms, seed=11, amp_scale=float(os.environ.get("GPSX_BENCH_AMP", "0.25")), two_bit=True)
    eng.set_if_format(capi.IF_2BIT_SM)
    prns = np.arange(1, 33, dtype=np.uint8)
    g = eng.grid_desc(prns, n_search=searches, n_ms=n_ms, search_stride_blocks=n_ms, dopp_min_hz=-5000, dopp_step_hz=500,
                      n_dopp=21)
    d_if = eng.malloc(blocks.size + 2)
    eng.h2d(d_if, np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
    d_pk = eng.malloc(searches * 32 * 21 * 8 * 16)
    d_keys = eng.malloc(searches * 32 * 21 * 8)

    def run():
        rc = eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(g), C.c_void_p(d_if), searches * n_ms, C.c_void_p(d_pk),
                                       C.c_void_p(d_keys), None, None, None)
        assert rc == 0, eng.lib.gpsx_last_error(eng.h)
    for _ in range(3):
        run()
    e0, e1 = eng.event(), eng.event()
    eng.record(e0)
    for _ in range(reps):
        run()
    eng.record(e1)
    eng.synchronize()
    ms = eng.elapsed_ms(e0, e1) / reps
    hyp = searches * n_ms * 32 * 21 * 16368
    print(json.dumps({"kernel": eng.lib.gpsx_last_kernel(eng.h).decode(), "searches": searches, "n_ms": n_ms,
                      "experiment": os.environ.get("GPSX_MX_EXPERIMENT", "0"), "ms": round(ms, 4),
                      "hyp_per_s": hyp / (ms * 1e-3)}))




if __name__ == "__main__":
    main()
