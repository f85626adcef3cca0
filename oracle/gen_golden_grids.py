#!/usr/bin/env python3
"""oracle/gen_golden_grids.py -- TEST INFRASTRUCTURE: the committed fixtures of the CPU oracle's LARGE acquisition-grid sweeps,
tests/golden/f11_grids/<key>.npz (key = sha256 of the inputs, oracle/pyoracle.py grid_key).

A fixture holds the INPUTS of one Oracle.acq_grid call of the GPU suite (the IF blocks -- synthetic, from the tests' seeded
generators --, the PRN list, the grid arguments) and this oracle's OUTPUTS for them (the (max, phase, sum, avr) record of every
(PRN, Doppler, bit shift)).  The input cases are harvested by running the GPU suite once under $GPSX_GOLDEN_RECORD=<dir> (the
oracle then leaves every large case it was asked for in <dir>); THIS script, run in the build container, recomputes every case
from its stored inputs with oracle/liboracle.so -- and, for one-block cases when oracle/_ref/libref_pm.so is built, a sample
of (PRN, Doppler, bit shift) cells with the reference's own correlation_search -- and writes the fixture.  The outputs in the
tree are therefore what the oracle computes HERE; a harvested output that differs from the recomputation is an error.

usage: gen_golden_grids.py [harvest-dir]        regenerate tests/golden/f11_grids/ (from the harvest dir, or in place)
       gen_golden_grids.py --verify [n]          recompute n (default: all) committed fixtures and compare
"""
import glob
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
DST = os.path.join(ROOT, "tests", "golden", "f11_grids")


def recompute(orc, z, threads):
    n_ms, dmin, dstep, n_dopp, n_bits = (int(a) for a in z["args"])
    return orc.acq_grid(z["blocks"], n_ms, z["prns"], dmin, dstep, n_dopp, n_bits, n_threads=threads, live=True)


def ref_sample(ref, z, peaks, rng, n=6):
    """one-block cases: n random (PRN, Doppler, bit shift) cells with the reference's own C (gps_misc.c correlation_search)"""
    n_ms, dmin, dstep, n_dopp, n_bits = (int(a) for a in z["args"])
    if n_ms != 1:
        return 0
    blk = np.ascontiguousarray(z["blocks"][:2046])
    for _ in range(n):
        p, d, b = int(rng.integers(len(z["prns"]))), int(rng.integers(n_dopp)), int(rng.integers(n_bits))
        di, dq = ref.wipeoff(blk, float(4092000 + dmin + d * dstep))
        mx, avr, ph = ref.correlation_search(ref.replica(ref.ca_code(int(z["prns"][p])), b), di, dq, 0, 2046)
        got = peaks[p, d, b]
        assert (int(got["max_val"]), int(got["avr"]), int(got["phase"])) == (mx, avr, ph), (p, d, b)
    return n


def main(argv):
    from oracle import pyoracle
    orc = pyoracle.Oracle()
    ref = pyoracle.RefPM() if pyoracle.RefPM.available() else None
    threads = max(4, min(64, len(os.sched_getaffinity(0))))
    rng = np.random.default_rng(6)
    if argv and argv[0] == "--verify":
        files = sorted(glob.glob(os.path.join(DST, "*.npz")))
        files = files[:int(argv[1])] if len(argv) > 1 else files
        for f in files:
            with np.load(f) as z:
                assert os.path.basename(f)[:-4] == pyoracle.grid_key(z["blocks"], z["prns"], tuple(int(a) for a in z["args"]))
                assert np.array_equal(recompute(orc, z, threads), z["peaks"]), f
        print(f"{len(files)} fixtures verified against the live oracle")
        return
    src = argv[0] if argv else DST
    os.makedirs(DST, exist_ok=True)
    total, cells = 0.0, 0
    for f in sorted(glob.glob(os.path.join(src, "*.npz"))):
        with np.load(f) as z:
            args = tuple(int(a) for a in z["args"])
            key = pyoracle.grid_key(z["blocks"], z["prns"], args)
            assert key == os.path.basename(f)[:-4], f
            t0 = time.perf_counter()
            peaks = recompute(orc, z, threads)
            dt = time.perf_counter() - t0
            if not np.array_equal(peaks, z["peaks"]):
                raise SystemExit(f"{f}: the harvested outputs differ from this container's recomputation")
            if ref is not None:
                cells += ref_sample(ref, z, peaks, rng)
            harvested = json.loads(str(z["meta"])) if "meta" in z else {}
            pyoracle.save_grid_fixture(DST, key, z["blocks"], z["prns"], args, peaks, generated_by="oracle/gen_golden_grids.py",
                                       seconds_here=round(dt, 2), threads_here=threads,
                                       seconds_when_harvested=round(float(harvested.get("seconds", 0.0)), 2))
            total += dt
            print(f"{key}  n_ms={args[0]} prns={len(z['prns'])} dopp={args[3]} bits={args[4]}  {dt:.1f} s")
    print(f"{len(glob.glob(os.path.join(DST, '*.npz')))} fixtures in {DST}; {total:.0f} s of oracle sweeps on {threads} threads; "
          f"{cells} cells cross-checked with the reference's own correlation_search")


if __name__ == "__main__":
    main(sys.argv[1:])
