set -x
for th in 1 16 64; do GPSX_STEP_THREADS=$th python - <<'PY'
import os, sys, time, ctypes as C, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import steps_driver as sd
from stm32f4_sdr_gps_amd import capi, synth
n, ms, n_sig = 16384, 700, 32
lib = capi.load_library()
e = capi.Engine(0); e.bind_thread_to_device(); e.close()
steps = sd.StepsLib(lib, False)
lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
lib.gps_tracking_process_batch.restype = None
sig_prn = [(i % 32) + 1 for i in range(n_sig)]
sig_dopp = [-5000.0 + 39.0 * i + 7.0 for i in range(n_sig)]
sig_delay = [(61.0 * i) % 16368 for i in range(n_sig)]
sats = [synth.Sat(sig_prn[i], sig_dopp[i], sig_delay[i], 0.12, 0.37 * i) for i in range(n_sig)]
stream = synth.make_if(ms, sats, noise_amp=1.0, seed=5)
per_sig = np.stack([sd.preset_channel(steps, sig_prn[i], int(round(sig_dopp[i] / 500.0)) * 500, int(sig_delay[i] // 8) % 2046) for i in range(n_sig)])
table = np.ascontiguousarray(per_sig[np.arange(n) % n_sig])
lat = np.zeros(ms); ntrk = np.zeros(ms, int)
for t in range(ms):
    steps.set_time(t)
    s = time.perf_counter()
    lib.gps_tracking_process_batch(table.ctypes.data, n, stream[t].ctypes.data, t & 3)
    lat[t] = time.perf_counter() - s
    ntrk[t] = int((table[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0] == sd.TRK_RUN).sum())
sp = np.nonzero(lat > 2e-3)[0]
print("threads", os.environ["GPSX_STEP_THREADS"], "p50", round(np.percentile(lat[350:], 50) * 1e6), "spikes at", [(int(i), round(lat[i] * 1e3, 1), int(ntrk[i - 1]), int(ntrk[i])) for i in sp][:40])
PY
done 2>&1 | grep threads | tee gpurun_out/r03c_spikes.txt
