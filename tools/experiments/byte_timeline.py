#!/usr/bin/env python3
"""Cycle stamps of one workgroup of the pipelined byte-phase form (library built with -DGPSX_MX_TIMELINE:
VARIANT_DEFS=-DGPSX_MX_TIMELINE bash tools/build_variant.sh): per half stage, what the barrier, the all-hands piece and the
role's own half took, for wave 0 (role 0) and wave 4 (role 1)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stm32f4_sdr_gps_amd import capi, synth   # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "stm32f4_sdr_gps_amd/lib/libgpsx_b.so")
eng = capi.Engine(0)
searches, n_prn, n_dopp = 256, 32, 29
blocks = synth.cold_start_block(searches, seed=11, amp_scale=0.25)
prns = np.arange(1, n_prn + 1, dtype=np.uint8)
g = eng.grid_desc(prns, n_search=searches, n_ms=1, search_stride_blocks=1, dopp_min_hz=-7000, dopp_step_hz=500, n_dopp=n_dopp,
                  phase_mode=capi.PHASES_BYTE)
n_pk = searches * n_prn * n_dopp
d_if = eng.malloc(blocks.size + 2)
eng.h2d(d_if, np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
d_peaks = eng.malloc(n_pk * 16 + 2 * 2048 * 8)
d_keys = eng.malloc(n_pk * 8)
for _ in range(4):
    eng._chk(eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(g), C.c_void_p(d_if), searches, C.c_void_p(d_peaks), C.c_void_p(d_keys),
                                       None, None, None), "grid")
eng.synchronize()
out = np.zeros(n_pk * 16 + 2 * 2048 * 8, np.uint8)
eng.d2h(out, d_peaks)
tl = out[n_pk * 16:].view(np.uint64).reshape(2, 2048).astype(np.int64)
for role in range(2):
    t = tl[role]
    t2 = t[1024:]
    t2 = t2[:int(np.flatnonzero(t2)[-1]) + 1] if t2.any() else t2[:0]
    if np.count_nonzero(t2) >= 16:   # four stamps per piece, alternating start-of-offset-0-stage / start-of-offset-8-stage pieces
        q = t2[:len(t2) // 8 * 8].reshape(-1, 8)[2:-2]
        print("  pieces of the offset-0 stage: vectors %5.0f  wipe-off / codes %5.0f  request %5.0f" % tuple(np.diff(q[:, :4], axis=1).mean(axis=0)))
        print("  pieces of the offset-8 stage: vectors %5.0f  commit %5.0f  fold %5.0f" % tuple(np.diff(q[:, 4:], axis=1).mean(axis=0)))
    t = t[:1024]
    n = int(np.flatnonzero(t)[-1]) + 1
    t = t[:n]
    print("role", role, "stamps", n, "total cycles", t[-1] - t[0])
    # stamps per half stage: even hs: top, after barrier, after piece ; odd hs: top, before role work
    i, hs, rows = 0, 0, []
    while i + 3 < n:   # three stamps per half stage: top, behind the barrier (even ones), behind role 1's pieces
        rows.append((hs, t[i + 1] - t[i], t[i + 2] - t[i + 1], t[i + 3] - t[i + 2]))
        i += 3
        hs += 1
    rows = np.array(rows)
    steady = rows[(rows[:, 0] >= 16) & (rows[:, 0] < rows[-1, 0] - 16)]
    for h in range(4):
        sel = steady[steady[:, 0] % 4 == h]
        print("  h =", h, "barrier wait %6.0f  pieces (role 1) %6.0f  own half (+ role 0 pieces) %6.0f  (n = %d)" % (sel[:, 1].mean(), sel[:, 2].mean(), sel[:, 3].mean(), len(sel)))
    print("  per cluster:", steady[:, 1:].sum() / (len(steady) / 4.0))
