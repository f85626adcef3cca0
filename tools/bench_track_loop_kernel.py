#!/usr/bin/env python3
"""k_track_loop alone (device-resident blocks, states and flag bytes; HIP events on the engine's stream): K milliseconds per
launch for n channels.  usage: bench_track_loop_kernel.py [K] [channels ...]   ($GPSX_LOOP_MUX17=1: the 17 ms multiplex, K a multiple of 17)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from stm32f4_sdr_gps_amd import capi, synth
    if os.environ.get("GPSX_LIB"):   # A/B runs against another build of the library
        capi.LIB_PATH = os.environ["GPSX_LIB"]
    eng = capi.Engine(0)
    mux = os.environ.get("GPSX_LOOP_MUX17") == "1"
    if mux:
        eng.set_loop_schedule(capi.SCHED_MUX17)
    k = int(sys.argv[1]) if len(sys.argv) > 1 else (17 if mux else 20)
    blk = synth.default_four_sv(k, seed=7)
    d_if = eng.malloc(blk.nbytes + 2)
    eng.h2d(d_if, np.concatenate([blk.reshape(-1), np.zeros(2, np.uint8)]))
    rows = []
    for n in [int(a) for a in sys.argv[2:]] or [256, 65536, 212992, 1048576]:
        st = np.zeros(n, capi.LOOP_DTYPE)
        st["prn"] = (np.arange(n) % 32) + 1
        st["code_phase_fine"] = (61 * np.arange(n) % 16368).astype(np.float32)
        st["if_freq_offset_hz"] = (-5000 + 39 * (np.arange(n) % 256)).astype(np.float32)
        st["found_freq_offset_hz"] = st["if_freq_offset_hz"].astype(np.int16)
        st["rng"] = np.arange(n) + 1
        d_st, d_fl = eng.malloc(st.nbytes), eng.malloc(n * k)
        eng.h2d(d_st, st)
        call = lambda t: eng._chk(eng.lib.gpsx_track_loop_dev(eng.h, C.c_void_p(d_if), k, C.c_void_p(d_st), n, t, C.c_void_p(d_fl), None), "loop")
        for i in range(3):
            call(i * k)
        e0, e1 = eng.event(), eng.event()
        reps = 20
        eng.record(e0)
        for i in range(reps):
            call((3 + i) * k)
        eng.record(e1)
        eng.synchronize()
        us = eng.elapsed_ms(e0, e1) / reps * 1e3
        rows.append({"channels": n, "ms_per_launch": k, "kernel_us": us, "us_per_ms_of_stream": us / k})
        eng.free(d_st)
        eng.free(d_fl)
    print(json.dumps({"kernel": "gpsx::k_track_loop", "schedule": "GPSX_SCHED_MUX17" if mux else "GPSX_SCHED_EVERY_MS", "rows": rows}))


if __name__ == "__main__":
    main()
