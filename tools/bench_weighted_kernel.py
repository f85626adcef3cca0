#!/usr/bin/env python3
"""The weighted two-bit acquisition extension's kernels alone (device-resident captures, HIP events on the engine's stream):
searches x 32 PRN x 21 Doppler x 16368 phases per launch, first on the matrix cores (k_acq_mxw), then on the vector ALU
(k_acq_weighted), and the sign-only fine grid (k_acq_mx<0>) on the same captures' sign plane beside them.
usage: bench_weighted_kernel.py [searches [reps]]"""
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
    if os.environ.get("GPSX_LIB"):   # A/B runs against another build of the library (tools/build_variant.sh)
        capi.LIB_PATH = capi.LAB_LIB_PATH = os.environ["GPSX_LIB"]
    eng = capi.Engine(0, lab=bool(os.environ.get("GPSX_LIB")))
    blocks = synth.cold_start_block(searches, seed=11, amp_scale=0.25, two_bit=True)
    prns = np.arange(1, 33, dtype=np.uint8)
    g = capi.AcqWeightedT(searches, 1, 32, prns.ctypes.data_as(C.POINTER(C.c_uint8)), -5000, 500, 21, 1)
    d_if = eng.malloc(blocks.size + 2)
    eng.h2d(d_if, np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
    d_pk = eng.malloc(searches * 32 * 21 * 16)

    def run():
        rc = eng.lib.gpsx_acq_grid_weighted_dev(eng.h, C.byref(g), C.c_void_p(d_if), searches, C.c_void_p(d_pk))
        assert rc == 0, eng.lib.gpsx_last_error(eng.h)

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = eng.event(), eng.event()
        eng.record(e0)
        for _ in range(reps):
            fn()
        eng.record(e1)
        eng.synchronize()
        return eng.elapsed_ms(e0, e1) / reps
    hyp = searches * 32 * 21 * 16368
    for path in (capi.ACQ_PATH_MATRIX, capi.ACQ_PATH_VECTOR):
        eng.set_acq_path(path)
        ms = timed(run)
        line = {"kernel": "gpsx::" + eng.lib.gpsx_last_kernel(eng.h).decode(), "searches": searches, "ms": round(ms, 3), "hyp_per_s": hyp / (ms * 1e-3)}
        if path == capi.ACQ_PATH_MATRIX:
            # per hypothesis 2 streams x 18 passes x 1024 chips x 2 (multiply, add) on MX-FP4 operands
            flops = hyp / 16 * 2 * 18 * 1024 * 2
            line.update({"mfma_tflops": flops / (ms * 1e-3) / 1e12, "frac_of_fp4_dense_peak": flops / (ms * 1e-3) / 10.0e15,
                         "note": "MFMA flops as issued (18 passes per 16 sample offsets) against the 10 PFLOP/s dense MX-FP4 peak"})
        else:
            dot4 = hyp * 2 * 256            # per hypothesis two streams x 256 four-chip steps
            line.update({"dot4_lane_ops_per_s": dot4 / (ms * 1e-3), "frac_of_valu_issue_peak": dot4 / (ms * 1e-3) / (256 * 64 * 2.4e9),
                         "note": "algorithmic v_dot4_i32_i8 lane-ops (512 per hypothesis) against one wave64 op per 4 cycles per SIMD at 2.4 GHz"})
        print(json.dumps(line))
    eng.set_acq_path(capi.ACQ_PATH_MATRIX)
    # the sign-only fine grid on the same captures (their sign plane): the reference-parity kernel this one is built from
    eng.set_if_format(capi.IF_2BIT_SM)
    gd = eng.grid_desc(prns, n_search=searches, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
    d_keys = eng.malloc(searches * 32 * 21 * 8)
    d_pk8 = eng.malloc(searches * 32 * 21 * 8 * 16)

    def run_sign():
        rc = eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(gd), C.c_void_p(d_if), searches, C.c_void_p(d_pk8), C.c_void_p(d_keys), None, None, None)
        assert rc == 0, eng.lib.gpsx_last_error(eng.h)
    ms = timed(run_sign)
    print(json.dumps({"kernel": "gpsx::" + eng.lib.gpsx_last_kernel(eng.h).decode(), "searches": searches, "ms": round(ms, 3), "hyp_per_s": hyp / (ms * 1e-3),
                      "note": "the sign-only fine grid, same shape"}))


if __name__ == "__main__":
    main()
