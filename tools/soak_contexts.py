#!/usr/bin/env python3
"""Create / use / destroy contexts in a loop and watch device memory: every buffer a context grows (code tables, arena,
scratch planes, multi-block scratch, capture rings, tracking graphs) must go with it."""
import ctypes as C
import os
os.environ.setdefault("GPSX_USE_LAB_LIBRARY", "1")   # forced kernel forms ($GPSX_ACQ_*): the lab build of the library
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_mb():
    hip = C.CDLL("libamdhip64.so")
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value / 2**20


def main():
    from stm32f4_sdr_gps_amd import capi, synth
    stream = synth.default_four_sv(12, seed=7)
    prns = np.arange(1, 33, dtype=np.uint8)
    base = None
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        os.environ["GPSX_ACQ_ALGO"] = ("poly", "mx", "dot8")[it % 3]
        os.environ["GPSX_ACQ_MS_MODE"] = ("walk", "blocks")[it % 2]
        e = capi.Engine(0)
        e.acq_grid(stream[:8], prns, n_search=2, n_ms=4, search_stride_blocks=4)
        e.acq_grid(stream[:4], prns, n_search=4)
        cap = capi.Capture(e, 4)
        for t in range(6):
            cap.push(stream[t])
        st = np.zeros(300, capi.TRK_DTYPE)
        st["prn"] = 5
        e.track_epl(cap.ready_view()[0], st)
        e.track_epl(stream[0], st[:4].copy())
        if it % 2:
            cap.close()          # otherwise the context closes it
        e.close()
        now = free_mb()
        if it == 6:
            base = now           # after the runtime's own one-time allocations: every (kernel, form) pair has run once
                                 # (code objects, and the queue's scratch for the kernels that spill)
        if base is not None:
            assert abs(now - base) < 64, f"iteration {it}: free memory moved by {now - base:.0f} MiB"
    print(f"ok: free device memory steady at {now:.0f} MiB")


if __name__ == "__main__":
    main()
