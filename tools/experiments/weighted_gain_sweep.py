#!/usr/bin/env python3
"""Processing gain of the weighted two-bit acquisition grid (gpsx_acq_grid_weighted) over the same correlator on the sign plane:
(capture, satellite) pairs acquired -- best cell on the true Doppler bin (+-1) and within 8 samples of the true code phase -- of
96 captures x 6 satellites, by amplitude scale of the synthetic satellites, and the time of each call."""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from stm32f4_sdr_gps_amd import capi, synth
eng=capi.Engine(0)
truth = {3: (-3210.0, 777.0), 5: (912.5, 1600.0), 11: (4480.0, 12001.0), 14: (4037.0, 4000.0), 20: (-1025.0, 9000.0), 30: (2018.0, 13000.0)}
prns = np.array(sorted(truth), np.uint8)
n=96
import time
for sc in (0.2,0.15,0.12,0.1,0.08,0.06):
    blocks = synth.cold_start_block(n, seed=11, amp_scale=sc, two_bit=True)
    out=[]
    for use_mag in (True, False):
        t=time.time(); pk = eng.acq_grid_weighted(blocks, prns, n, -5000, 500, 21, use_magnitude=use_mag); dt=time.time()-t
        hits=0
        for i,p in enumerate(prns):
            dopp, delay = truth[int(p)]
            best_bin = pk[:, i, :]["max_val"].argmax(axis=1)
            best = pk[np.arange(n), i, best_bin]
            bin_ok = np.abs(-5000 + 500 * best_bin - dopp) <= 500
            phase_ok = np.abs((best["phase"].astype(int) - delay + 8184) % 16368 - 8184) <= 8
            hits += int((bin_ok & phase_ok).sum())
        out.append((hits, round(dt*1e3,1)))
    print(sc, out, flush=True)
