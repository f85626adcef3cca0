#!/usr/bin/env python3
"""Liveness soak of the batched step's worker pool (csrc/gpsx_steps.cpp StepPool): many thousands of
gps_tracking_process_batch calls at the smallest threaded channel count, back to back and with random pauses between them
(workers spinning, asleep on the futex, or on their way to sleep when the next run is posted).  A lost wake-up or a job left
unclaimed shows as a hang -- run it under `timeout`.  tools/soak_step_pool.py [steps [channels]]"""
import ctypes as C
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    steps_n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    import steps_driver as sd
    from stm32f4_sdr_gps_amd import capi, synth
    lib = capi.load_library()
    steps = sd.StepsLib(lib, False)
    lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
    lib.gps_tracking_process_batch.restype = None
    sats = [synth.Sat(p + 1, -3000.0 + 700.0 * p, 500.0 * p, 0.3, 0.1 * p) for p in range(8)]
    stream = synth.make_if(64, sats, noise_amp=1.0, seed=9)
    per = np.stack([sd.preset_channel(steps, p + 1, int(round((-3000.0 + 700.0 * p) / 500.0)) * 500, int(500.0 * p // 8) % 2046)
                    for p in range(8)])
    table = np.ascontiguousarray(per[np.arange(n) % 8])
    rng = random.Random(1)
    t0 = time.time()
    for t in range(steps_n):
        steps.set_time(t)
        blk = stream[t % 64]
        lib.gps_tracking_process_batch(table.ctypes.data, n, blk.ctypes.data, t & 3)
        phase = (t // 5000) % 3
        if phase == 1:
            time.sleep(rng.random() * 2e-4)        # around the workers' spin-to-sleep transition
        elif phase == 2 and t % 7 == 0:
            time.sleep(rng.random() * 3e-3)        # long pauses: everybody asleep
    print(f"SOAK OK: {steps_n} steps of {n} channels, {lib.gps_tracking_batch_workers()} workers, {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
