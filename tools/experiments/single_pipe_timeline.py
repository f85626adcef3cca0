#!/usr/bin/env python3
"""Cycle stamps of one workgroup (block 100, clusters 5 and 6) of the pipelined single-block form k_acq_mx<6> on the bench launch,
library built with -DGPSX_MX_TIMELINE: per half step the barrier, the next vector's build, role 1's pieces, the role's own half
(pass or epilogue) and role 0's pieces, for wave 0 (role 0) and wave 4 (role 1)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stm32f4_sdr_gps_amd import capi, synth   # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "stm32f4_sdr_gps_amd/lib/libgpsx_b.so")
eng = capi.Engine(0)
searches, n_prn, n_dopp = 256, 32, 21
blocks = synth.cold_start_block(searches, seed=11, amp_scale=0.25)
prns = np.arange(1, n_prn + 1, dtype=np.uint8)
g = eng.grid_desc(prns, n_search=searches, n_ms=1, search_stride_blocks=1, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=n_dopp,
                  phase_mode=capi.PHASES_FINE)
n_pk = searches * n_prn * n_dopp * 8
d_if = eng.malloc(blocks.size + 2)
eng.h2d(d_if, np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
d_peaks = eng.malloc(n_pk * 16 + 2 * 512 * 8)
d_keys = eng.malloc(searches * n_prn * n_dopp * 8)
for _ in range(3):
    eng._chk(eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(g), C.c_void_p(d_if), searches, C.c_void_p(d_peaks), C.c_void_p(d_keys),
                                       None, None, None), "grid")
eng.synchronize()
print(eng.lib.gpsx_last_kernel(eng.h))
out = np.zeros(n_pk * 16 + 2 * 512 * 8, np.uint8)
eng.d2h(out, d_peaks)
tl = out[n_pk * 16:].view(np.uint64).reshape(2, 512).astype(np.int64)
for role in range(2):
    t = tl[role]
    n = int(np.flatnonzero(t)[-1]) + 1
    t = t[:n]
    print("role", role, "stamps", n)
    # even half steps: barrier, build, role 1's pieces, own half, role 0's pieces = 5 intervals; odd ones: the last three
    i, hs, rows = 0, 0, []
    while True:
        k = 5 if hs % 2 == 0 else 3
        if i + k >= n:
            break
        d = np.diff(t[i:i + k + 1])
        rows.append((hs % 34,) + (tuple(d) if k == 5 else (0, 0) + tuple(d)))
        i += k
        hs += 1
    rows = np.array(rows)
    print("  hs  barrier  build  pieces1    own  pieces0")
    for r in rows[:68]:
        print("  %2d  %6d %6d  %6d %6d  %6d" % tuple(r))
    print("  per cluster:", rows[:68, 1:].sum() / 2.0)
