#!/usr/bin/env python3
"""Cycle stamps of one workgroup (block 1000) of k_acq_mx<0> on the bench launch (256 captures, 32 PRN x 21 Doppler x 16368
phases), library built with -DGPSX_MX_TIMELINE (VARIANT_DEFS=-DGPSX_MX_TIMELINE bash tools/build_variant.sh): per half step
what the barrier, the vector build, the pass and the epilogue took, for wave 0 (role 0) and wave 4 (role 1)."""
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
n_ms = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # 10: the walk form k_acq_mx<3>, block 5 of workgroup 1000
blocks = synth.cold_start_block(searches * n_ms, seed=11, amp_scale=0.25)
prns = np.arange(1, n_prn + 1, dtype=np.uint8)
g = eng.grid_desc(prns, n_search=searches, n_ms=n_ms, search_stride_blocks=n_ms, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=n_dopp,
                  phase_mode=capi.PHASES_FINE)
n_pk = searches * n_prn * n_dopp * 8
d_if = eng.malloc(blocks.size + 2)
eng.h2d(d_if, np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
d_peaks = eng.malloc(n_pk * 16 + 2 * 512 * 8)
d_keys = eng.malloc(searches * n_prn * n_dopp * 8)
for _ in range(3):
    eng._chk(eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(g), C.c_void_p(d_if), searches * n_ms, C.c_void_p(d_peaks), C.c_void_p(d_keys),
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
    print("role", role, "stamps", n, "loop cycles", t[-1] - t[0])
    # first stamp: before the start values; then five per half step: top, behind the barrier, behind the build, behind the pass,
    # (next top) behind the epilogue
    rows = []
    i, hs = 1, 0
    while i + 4 < n:
        rows.append((hs, t[i + 1] - t[i], t[i + 2] - t[i + 1], t[i + 3] - t[i + 2], t[i + 4] - t[i + 3]))
        i += 4
        hs += 1
    rows = np.array(rows)
    print("  hs  barrier   build    pass   epilogue(+behind)")
    for r in rows:
        print("  %2d  %6d  %6d  %6d  %6d" % tuple(r))
    mid = rows[(rows[:, 0] >= 6) & (rows[:, 0] < 32)]
    for par in range(2):
        sel = mid[mid[:, 0] % 2 == par]
        print("  hs %% 2 == %d: barrier %6.0f build %6.0f pass %6.0f epilogue %6.0f" % ((par,) + tuple(sel[:, 1:].mean(axis=0))))
    print("  per step:", mid[:, 1:].sum() / (len(mid) / 2.0))
