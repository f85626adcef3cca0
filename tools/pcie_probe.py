#!/usr/bin/env python3
"""Host-to-host rate of gpsx_acq_grid_async with 1..4 contexts in rotation (the pcie_inclusive leg of bench.py, opened up):
tools/pcie_probe.py [captures]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from stm32f4_sdr_gps_amd import capi, synth
    n_search = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    blocks = synth.cold_start_block(n_search, seed=11, amp_scale=0.25, two_bit=True)
    prns = np.arange(1, 33, dtype=np.uint8)
    engs = [capi.Engine(0) for _ in range(4)]
    engs[0].bind_thread_to_device()
    g = engs[0].grid_desc(prns, n_search=n_search, n_ms=1, search_stride_blocks=1, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
    pins = []
    for e in engs:
        e.set_if_format(capi.IF_2BIT_SM)
        pins.append((torch.from_numpy(blocks.reshape(-1).copy()).pin_memory(),
                     torch.zeros(n_search * 32 * 21 * 8 * 16, dtype=torch.uint8).pin_memory(),
                     torch.zeros(n_search * 32 * 21, dtype=torch.int64).pin_memory()))
    hyp = n_search * 32 * 21 * 16368
    for n_ctx in (1, 2, 3, 4):
        for want_keys in (True, False):
            reps = 12
            warm = max(6, 2 * n_ctx)    # (the first calls of a process are slow: pinned pages meet the DMA for the first time)
            for i in range(reps + warm):
                if i == warm:
                    for e in engs[:n_ctx]:
                        e.synchronize()
                    t0 = time.perf_counter()
                e = engs[i % n_ctx]
                p = pins[i % n_ctx]
                e.synchronize()
                rc = e.lib.gpsx_acq_grid_async(e.h, C.byref(g), p[0].data_ptr(), n_search, p[1].data_ptr(),
                                               p[2].data_ptr() if want_keys else None)
                assert rc == 0
            for e in engs[:n_ctx]:
                e.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"{n_ctx} contexts, keys={want_keys}: {dt * 1e3:.3f} ms per call, {hyp / dt:.3e} hyp/s")


if __name__ == "__main__":
    main()
