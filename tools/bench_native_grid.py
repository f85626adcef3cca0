#!/usr/bin/env python3
"""The reference-native acquisition grid (SURVEY.md 8(d), config 3 "also report"): 32 PRN x 29 Doppler bins
(-7000 .. +7000 Hz @ 500 Hz, PM/GPS/acquisition.c:285-289) x 2046 byte-granular code phases, replica bit shift 0 --
what acquisition_freq_search sweeps on the MCU, one (PRN, Doppler) per captured millisecond.

Device-resident captures, HIP events on the engine's stream around K launches of gpsx_acq_grid_dev.  The byte phases are
sample offsets t0 = 0 and 8 of the fine grid: the matrix-core kernel runs the ten passes that lead there and two
epilogues (GPSX_ACQ_ALGO=dot8: round 1's direct kernel).  Prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
os.environ.setdefault("GPSX_USE_LAB_LIBRARY", "1")   # forced kernel forms ($GPSX_ACQ_*): the lab build of the library
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--searches", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    from stm32f4_sdr_gps_amd import capi, synth
    if os.environ.get("GPSX_LIB"):   # A/B runs against another build of the library
        capi.LIB_PATH = capi.LAB_LIB_PATH = os.environ["GPSX_LIB"]   # (tools/build_variant.sh: a lab build)
    eng = capi.Engine(0)
    n_prn, n_dopp, n_phase = 32, 29, 2046
    blocks = synth.cold_start_block(args.searches, seed=11, amp_scale=0.25)
    prns = np.arange(1, n_prn + 1, dtype=np.uint8)
    g = eng.grid_desc(prns, n_search=args.searches, n_ms=1, search_stride_blocks=1, dopp_min_hz=-7000, dopp_step_hz=500,
                      n_dopp=n_dopp, phase_mode=capi.PHASES_BYTE)
    n_pk = args.searches * n_prn * n_dopp
    d_if = eng.malloc(blocks.size + 2)
    eng.h2d(d_if, np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
    d_peaks = eng.malloc(n_pk * 16)
    d_keys = eng.malloc(n_pk * 8)

    def launch():
        rc = eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(g), C.c_void_p(d_if), args.searches, C.c_void_p(d_peaks),
                                       C.c_void_p(d_keys), None, None, None)
        eng._chk(rc, "gpsx_acq_grid_dev")

    for _ in range(args.warmup):
        launch()
    eng.synchronize()
    e0, e1 = eng.event(), eng.event()
    eng.record(e0)
    for _ in range(args.steps):
        launch()
    eng.record(e1)
    eng.synchronize()
    ms = eng.elapsed_ms(e0, e1) / args.steps
    hyp = args.searches * n_prn * n_dopp * n_phase
    # spot check against what the reference would have reported: the strongest (PRN, Doppler) of capture 0
    keys = np.zeros(n_pk, np.int64)
    eng.d2h(keys, d_keys)
    k0 = keys.reshape(args.searches, n_prn, n_dopp)[0]
    p, d = np.unravel_index(int(np.argmax(k0)), k0.shape)
    name, cus, _ = eng.device_info()
    print(json.dumps({"metric": "acquisition hypotheses/sec, reference-native grid (32 PRN x 29 Doppler x 2046 byte phases)",
                      "value": hyp / (ms * 1e-3), "unit": "hypotheses/s", "ms_per_launch": ms,
                      "searches_per_launch": args.searches, "hypotheses_per_launch": hyp, "device": name,
                      "kernel": "gpsx::" + eng.lib.gpsx_last_kernel(eng.h).decode() + " (byte-phase mode)",
                      "strongest_of_capture_0": {"prn": int(prns[p]), "doppler_hz": int(-7000 + 500 * d),
                                                 "max_val": int(k0[p, d] >> 14)},
                      "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": 10000.0,
                                   # four FP4 GEMM passes (two per byte offset of a chip offset) x 2 streams x 2*32*1024*1024 flops
                                   # per (capture, Doppler) pair of 32 PRN x 2046 hypotheses
                                   "flops_per_hyp": 4 * 2 * 2.0 * 32 * 1024 * 1024 / (32 * 2046),
                                   "achieved": 4 * 2 * 2.0 * 32 * 1024 * 1024 * args.searches * n_dopp / (ms * 1e-3) / 1e12,
                                   "frac": 4 * 2 * 2.0 * 32 * 1024 * 1024 * args.searches * n_dopp / (ms * 1e-3) / 1e16},
                      "mcu_equivalent": "one (PRN, Doppler) per ~0.2 s on STM32F407 (SURVEY.md 3.2): 928 pairs = ~186 s "
                                        "per capture"}))


if __name__ == "__main__":
    main()
