#!/usr/bin/env python3
"""Golden vectors for sdrobs2obsd (PM/GPS/RTK/rtklib_common.c:75-92), the step between the pseudorange calculation and the
position solver: random channel records -> what the reference's own function (oracle/_ref/libref_pvt.so, built in place from
RTK/solving.c + RTK/rtklib_common.c) writes into its observation records.  Run in the build container (needs
/root/reference); writes tests/golden/f10_obs.npz.  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pvt_types as T  # noqa: E402

PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "gpsx_compat.h"
int main(void){
 printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(gps_ch_t), offsetof(gps_ch_t, obs_data.pseudorange_m), offsetof(gps_ch_t, obs_data.tow_s),
        offsetof(gps_ch_t, eph_data.week_gpst), offsetof(gps_ch_t, prn), offsetof(gps_ch_t, tracking_data.if_freq_offset_hz),
        offsetof(gps_ch_t, tracking_data.snr_value), sizeof(obsd_t));
 return 0; }
"""


def channel_offsets():
    """(sizeof gps_ch_t, offsets of pseudorange_m, tow_s, week_gpst, prn, if_freq_offset_hz, snr_value, sizeof obsd_t) from
    include/gpsx_compat.h (layout-identical to the reference's headers: tests/test_abi_and_host.py)."""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c")
        open(src, "w").write(PROBE)
        exe = os.path.join(d, "p")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        return [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]


def random_table(rng, n, off):
    size, o_pr, o_tow, o_week, o_prn, o_freq, o_snr, _ = off
    table = np.zeros((n, size), np.uint8)
    vals = []
    for i in range(n):
        pr = float(rng.uniform(1.9e7, 2.6e7))
        tow = float(rng.choice([rng.uniform(0, 604800), 0.0, 604799.999, -5.25, 2e9, -3e9, 17.0]))
        week = int(rng.integers(1000, 3500))
        prn = int(rng.integers(1, 33))
        freq = np.float32(rng.uniform(-7000, 7000))
        snr = np.float32(rng.choice([rng.uniform(-25, 40), -20.0, 43.9, 44.0, 100.0]))
        table[i, o_pr:o_pr + 8] = np.frombuffer(np.float64(pr).tobytes(), np.uint8)
        table[i, o_tow:o_tow + 8] = np.frombuffer(np.float64(tow).tobytes(), np.uint8)
        table[i, o_week:o_week + 4] = np.frombuffer(np.int32(week).tobytes(), np.uint8)
        table[i, o_prn] = prn
        table[i, o_freq:o_freq + 4] = np.frombuffer(freq.tobytes(), np.uint8)
        table[i, o_snr:o_snr + 4] = np.frombuffer(snr.tobytes(), np.uint8)
        vals.append((pr, tow, week, prn, float(freq), float(snr)))
    return table, np.array(vals)


def run(lib, table, obsd_size):
    n = len(table)
    out = np.zeros((n, obsd_size), np.uint8)
    lib.sdrobs2obsd(C.c_void_p(table.ctypes.data), C.c_int(n), C.c_void_p(out.ctypes.data))
    return out


def main():
    ref = T.load_lazy(os.path.join(ROOT, "oracle", "_ref", "libref_pvt.so"))
    off = channel_offsets()
    rng = np.random.default_rng(10)
    table, vals = random_table(rng, 64, off)
    out = run(ref, table, off[7])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "f10_obs.npz"), inputs=vals, obsd=out,
                        offsets=np.array(off))
    print("f10_obs.npz:", out.shape)


if __name__ == "__main__":
    main()
