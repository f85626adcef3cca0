#!/usr/bin/env python3
"""Does the walk form's record stream (k_acq_mx<3>: 1 MB of running sums per workgroup between a record's write and its
re-read, 268 MB in flight on 256 CUs) hit the 256 MB memory-side cache when less of it is in flight?  Runs the 256-search
ten-block launch on a stream restricted to a fraction of the CUs (hipExtStreamCreateWithCUMask, the same share of every
group of four / two CUs) and prints ms per launch and ms x CUs / 256: if the stream cost nothing the latter would be the
arithmetic's ~22-23 ms; if the chip-wide footprint does not matter it stays at the full chip's figure.
  tools/experiments/walk_cu_mask.py [searches [reps]]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from stm32f4_sdr_gps_amd import capi, synth
    searches = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_ms = 10
    hip = C.CDLL("libamdhip64.so")
    blocks = synth.cold_start_block(searches * n_ms, seed=11, amp_scale=0.25, two_bit=True)
    prns = np.arange(1, 33, dtype=np.uint8)
    for name, word, share in (("all", 0xFFFFFFFF, 1.0), ("7 of 8", 0x7F7F7F7F, 0.875), ("3 of 4", 0x77777777, 0.75), ("5 of 8", 0x1F1F1F1F, 0.625),
                              ("1 of 2", 0x55555555, 0.5)):
        mask = (C.c_uint32 * 8)(*([word] * 8))
        stream = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(stream), 8, mask)
        assert rc == 0, rc
        eng = capi.Engine(0, stream=stream.value)
        eng.set_if_format(capi.IF_2BIT_SM)
        g = eng.grid_desc(prns, n_search=searches, n_ms=n_ms, search_stride_blocks=n_ms, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
        d_if = eng.malloc(blocks.size + 2)
        eng.h2d(d_if, np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
        d_pk = eng.malloc(searches * 32 * 21 * 8 * 16)
        d_keys = eng.malloc(searches * 32 * 21 * 8)

        def run():
            rc = eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(g), C.c_void_p(d_if), searches * n_ms, C.c_void_p(d_pk), C.c_void_p(d_keys),
                                           None, None, None)
            assert rc == 0, eng.lib.gpsx_last_error(eng.h)
        run()
        e0, e1 = eng.event(), eng.event()
        eng.record(e0)
        for _ in range(reps):
            run()
        eng.record(e1)
        eng.synchronize()
        ms = eng.elapsed_ms(e0, e1) / reps
        print(json.dumps({"cus": name, "share": share, "kernel": eng.lib.gpsx_last_kernel(eng.h).decode(), "ms_per_launch": round(ms, 3),
                          "ms_times_share": round(ms * share, 3), "records_in_flight_MB": round(268.4 * share, 1)}), flush=True)
        for d in (d_if, d_pk, d_keys):
            eng.free(d)
        eng.close()
        hip.hipStreamDestroy(stream)


if __name__ == "__main__":
    main()
