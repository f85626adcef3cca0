#!/usr/bin/env python3
"""One-off evidence run (not part of the test suite: minutes of CPU): the bench's OWN headline launch -- 256 captures of 2-bit IF,
32 PRN x 21 Doppler x 16368 phases each, one gpsx_acq_grid_dev call on k_acq_mx<0> -- compared with the CPU oracle on EVERY
capture: every (max, phase, sum, avr) record of every (PRN, Doppler, bit shift) and every packed key, 2.8e9 hypotheses behind
them.  Then the same captures on the vector-ALU path (k_acq_poly) byte for byte against the matrix path, and N_TEN ten-block
searches (BASELINE configs[3], k_acq_mx<3>) of the same stream against the oracle.  The oracle is the checker, computed LIVE
here (no fixtures).  Prints one JSON summary; exits non-zero on the first mismatch.

usage (on the GPU box): python tools/full_launch_parity.py [captures [ten_block_searches [amp_scale]]]  > gpurun_out/r06_full_launch_parity.json
(amp_scale 0.25 = the bench's captures; 1.0 = the strong test signal, whose magnitudes take the exact-root path of the epilogue far
more often; the run also sweeps the reference's OWN grid -- 32 x 29 x 2046 byte phases, k_acq_mx<4> -- on every capture)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def want_keys(want):
    fine = 8 * want["phase"].astype(np.int64) + np.arange(8)[None, None, :]
    return ((want["max_val"].astype(np.int64) << 14) | (16383 - fine)).max(axis=2)


def main():
    from oracle import pyoracle
    from stm32f4_sdr_gps_amd import capi, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_ten = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    amp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
    threads = max(4, min(64, len(os.sched_getaffinity(0))))
    orc = pyoracle.Oracle()            # grid_fixtures stays None: every sweep below is computed live
    prns = np.arange(1, 33, dtype=np.uint8)
    grid = dict(dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
    two = synth.cold_start_block(n, seed=11, amp_scale=amp, two_bit=True)           # bench.py's captures (at amp_scale 0.25)
    one = synth.cold_start_block(n, seed=11, amp_scale=amp)                           # their sign plane
    eng = capi.Engine(0)
    eng.set_if_format(capi.IF_2BIT_SM)
    t0 = time.time()
    pk, keys = eng.acq_grid(two, prns, n_search=n, **grid)
    kernel = eng.lib.gpsx_last_kernel(eng.h).decode()
    out = {"captures": n, "amp_scale": amp, "kernel": "gpsx::" + kernel, "hypotheses": n * 32 * 21 * 16368, "oracle_threads": threads}
    t0 = time.time()
    for i in range(n):
        w = orc.acq_grid(one[i:i + 1], 1, prns, -5000, 500, 21, 8, n_threads=threads, live=True)
        for f in ("max_val", "phase", "sum", "avr"):
            if not np.array_equal(pk[i][f], w[f]):
                raise SystemExit(f"capture {i}: field {f} differs from the oracle")
        if not np.array_equal(keys[i], want_keys(w)):
            raise SystemExit(f"capture {i}: packed keys differ from the oracle")
    out["records_compared"] = int(pk.size)
    out["keys_compared"] = int(keys.size)
    out["oracle_seconds"] = round(time.time() - t0, 1)
    out["every_record_and_key_identical_to_the_oracle"] = True
    # the same launch on the vector ALU (north_star's letter): byte for byte the matrix path's records and keys
    vec = capi.Engine(0)
    vec.set_if_format(capi.IF_2BIT_SM)
    vec.set_acq_path(capi.ACQ_PATH_VECTOR)
    pk_v, keys_v = vec.acq_grid(two, prns, n_search=n, **grid)
    out["vector_alu_kernel"] = "gpsx::" + vec.lib.gpsx_last_kernel(vec.h).decode()
    if pk_v.tobytes() != pk.tobytes() or not np.array_equal(keys_v, keys):
        raise SystemExit("the vector-ALU path's records differ from the matrix path's")
    out["vector_alu_path_identical"] = True
    vec.close()
    # the reference's own grid on every capture: 29 Doppler bins (+-7 kHz), 2046 byte phases, bit shift 0
    pk_b, keys_b = eng.acq_grid(two, prns, n_search=n, dopp_min_hz=-7000, dopp_step_hz=500, n_dopp=29, phase_mode=capi.PHASES_BYTE)
    out["native_grid_kernel"] = "gpsx::" + eng.lib.gpsx_last_kernel(eng.h).decode()
    for i in range(n):
        w = orc.acq_grid(one[i:i + 1], 1, prns, -7000, 500, 29, 1, n_threads=threads, live=True)
        for f in ("max_val", "phase", "sum", "avr"):
            if not np.array_equal(pk_b[i][f].reshape(w[f].shape), w[f]):
                raise SystemExit(f"native grid, capture {i}: field {f} differs from the oracle")
    out["native_grid_records_compared"] = int(pk_b.size)
    out["native_grid_identical_to_the_oracle"] = True
    # BASELINE configs[3]: ten-block searches of the same stream (search s = blocks 10 s .. 10 s + 9), walk form
    if n_ten and n >= 10 * n_ten:
        pk10, keys10 = eng.acq_grid(two[:10 * n_ten], prns, n_search=n_ten, n_ms=10, search_stride_blocks=10, **grid)
        out["ten_block_kernel"] = "gpsx::" + eng.lib.gpsx_last_kernel(eng.h).decode()
        t0 = time.time()
        for s in range(n_ten):
            w = orc.acq_grid(one[10 * s:10 * s + 10], 10, prns, -5000, 500, 21, 8, n_threads=threads, live=True)
            for f in ("max_val", "phase", "sum", "avr"):
                if not np.array_equal(pk10[s][f], w[f]):
                    raise SystemExit(f"ten-block search {s}: field {f} differs from the oracle")
            if not np.array_equal(keys10[s], want_keys(w)):
                raise SystemExit(f"ten-block search {s}: packed keys differ from the oracle")
        out["ten_block_searches"] = n_ten
        out["ten_block_hypothesis_blocks"] = n_ten * 10 * 32 * 21 * 16368
        out["ten_block_oracle_seconds"] = round(time.time() - t0, 1)
        out["ten_block_identical_to_the_oracle"] = True
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
