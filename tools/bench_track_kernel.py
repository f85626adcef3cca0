#!/usr/bin/env python3
"""k_track_epl alone (device-resident block, states and accumulators; HIP events on the engine's stream): what part of
the per-millisecond tracking step is the kernel and what part the PCIe round trip."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from stm32f4_sdr_gps_amd import capi, synth
    eng = capi.Engine(0)
    blk = synth.default_four_sv(1, seed=7)[0]
    d_if = eng.malloc(2048)
    eng.h2d(d_if, np.concatenate([blk, np.zeros(2, np.uint8)]))
    rows = []
    for n in [int(a) for a in sys.argv[1:]] or [256, 4096, 65536, 212992]:
        st = np.zeros(n, capi.TRK_DTYPE)
        st["prn"] = (np.arange(n) % 32) + 1
        st["code_phase_fine"] = (61 * np.arange(n) % 16368).astype(np.float32)
        st["if_freq_offset_hz"] = (-5000 + 39 * (np.arange(n) % 256)).astype(np.float32)
        d_st, d_iq = eng.malloc(st.nbytes), eng.malloc(n * 12)
        eng.h2d(d_st, st)
        for _ in range(5):
            eng._chk(eng.lib.gpsx_track_epl_batch_dev(eng.h, C.c_void_p(d_if), C.c_void_p(d_st), n, C.c_void_p(d_iq)), "trk")
        e0, e1 = eng.event(), eng.event()
        reps = 50
        eng.record(e0)
        for _ in range(reps):
            eng.lib.gpsx_track_epl_batch_dev(eng.h, C.c_void_p(d_if), C.c_void_p(d_st), n, C.c_void_p(d_iq))
        eng.record(e1)
        eng.synchronize()
        rows.append({"channels": n, "kernel_us": eng.elapsed_ms(e0, e1) / reps * 1e3})
        eng.free(d_st)
        eng.free(d_iq)
    print(json.dumps({"kernel": "gpsx::k_track_epl", "rows": rows}))


if __name__ == "__main__":
    main()
