#!/usr/bin/env python3
"""BASELINE.json configs[4] with the tracking loops ON THE DEVICE: N concurrent channels in closed loop on one shared IF
stream, sustained real time, the host out of the per-millisecond path.

Per launch of K milliseconds (default 20): the K blocks go to the GPU, ONE kernel (k_track_loop) runs, per channel and
millisecond, the E/P/L correlators, the reference's DLL / PLL / FLL, false-lock check, SNR and 20 ms bit synchroniser on
channel state that stays in HBM, and one flag byte per channel and millisecond comes back; the host's share is the word
layer (gps_tracking_words_batch: one call of the reference's word logic per completed navigation bit, i.e. per channel every
20 ms) and handing polarity changes back.  The stream and the channels are those of tools/bench_tracking_closed_loop.py
(channel i tracks signal i mod S; hand-over from a cold start's acquisition result); pre-tracking and the first 240 ms run in
the host mode on the S distinct channels, whose loop states are then handed to the device (gpsx_loop_state_from_channel)
and replicated.

Real time: launch j's blocks are complete at (j + 1) K ms; its results must be back before launch j + 1's are, i.e. every
launch's latency (blocks in -> flags consumed) must stay under K ms.  Reported: p50 / p99 / max of that latency over the
steady half, the same per millisecond of stream (latency / K), and how many channels hold code and carrier lock at the end.

Importable: device_loop(channels, ms, k, ...) -> dict (bench.py's `tracking.closed_loop.device_loop`)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_cache = {}
T_HAND = 240


def _prepare(n_sig, ms, amp, lib, literal=False, fast_synth=False, t_hand=T_HAND):
    """stream [ms, 2046], per-signal channel records and device loop states at tick t_hand (host mode up to there)"""
    import steps_driver as sd
    from stm32f4_sdr_gps_amd import capi, synth
    key = (n_sig, ms, amp, literal, fast_synth, t_hand)
    if key in _cache:
        return _cache[key]
    _cache.clear()
    sig_prn = [(i % 32) + 1 for i in range(n_sig)]
    sig_dopp = [-5000.0 + 39.0 * i + (0.0 if literal else 7.0) for i in range(n_sig)]
    sig_delay = [(61.0 * i) % 16368 for i in range(n_sig)]
    sats = [synth.Sat(sig_prn[i], sig_dopp[i], sig_delay[i], amp, 0.37 * i) for i in range(n_sig)]
    stream = (synth.make_if_static if literal or fast_synth else synth.make_if)(ms, sats, noise_amp=1.0, seed=5)
    steps = sd.StepsLib(lib, False)
    lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
    lib.gps_tracking_process_batch.restype = None
    table = np.stack([sd.preset_channel(steps, sig_prn[i], int(round(sig_dopp[i] / 500.0)) * 500, int(sig_delay[i] // 8) % 2046)
                      for i in range(n_sig)])
    for t in range(t_hand):
        steps.set_time(t)
        lib.gps_tracking_process_batch(table.ctypes.data, n_sig, stream[t].ctypes.data, t & 3)
    st = np.zeros(n_sig, capi.LOOP_DTYPE)
    for i in range(n_sig):
        lib.gpsx_loop_state_from_channel(table[i].ctypes.data, i + 1, st[i:i + 1].ctypes.data)
    tracking = table[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0] == sd.TRK_RUN
    _cache[key] = (stream, table, st, tracking, np.array(sig_dopp), np.array(sig_delay))
    return _cache[key]


def device_loop(channels=256, ms=1200, k=20, signals=32, amp=0.12, paced=True, bind=True, literal=False, fast_synth=False, mux17=False):
    """literal: SURVEY.md 8(d) config 5 to the letter -- one signal per channel at -5000 + 39 i Hz (no 7 Hz offset), the stream
    tests/golden/f7_steps_config5_256ch.npz was recorded on when channels = 256 and ms = 10000 (the reference's own lock count on
    it is reported beside the device loop's).
    mux17: GPSX_SCHED_MUX17 -- the channels are receivers of four in the reference's 17 ms multiplex (channel c = slot c & 3 of
    receiver c >> 2, served 4 ms of every 17); k must be a multiple of 17, the hand-over tick is 255 = 15 x 17."""
    from stm32f4_sdr_gps_amd import capi
    n = channels
    t_hand = 255 if mux17 else T_HAND
    assert not mux17 or k % 17 == 0
    n_sig = signals if 0 < signals < n and not literal else n
    eng = capi.Engine(0)
    lib = eng.lib
    affinity = os.sched_getaffinity(0)
    bound = eng.bind_thread_to_device() if bind else False
    try:
        t0 = time.time()
        if mux17:
            eng.set_loop_schedule(capi.SCHED_MUX17)
        stream, sig_table, sig_st, sig_tracking, sig_dopp, sig_delay = _prepare(n_sig, ms, amp, lib, literal, fast_synth, t_hand)
        prep_s = time.time() - t0
        idx = np.arange(n) % n_sig
        table = np.ascontiguousarray(sig_table[idx])              # the host's records: word layer state per channel
        st = np.ascontiguousarray(sig_st[idx])
        st["rng"] = np.arange(n, dtype=np.uint32) + 1
        # (a signal whose channel never left pre-tracking -- PRN 1 at code phase 0, see bench.py -- is handed over as it is and
        #  counts as not locked)
        d_state = eng.malloc(st.nbytes)
        eng.h2d(d_state, st)
        blocks = eng.host_array((k, stream.shape[1]), np.uint8)   # page-locked: what a capture driver fills
        flags = eng.host_array((k, n), np.uint8)
        changed = np.zeros(max(16, n), np.int32)
        n_launch = (ms - t_hand) // k
        lat = np.zeros(n_launch)
        gpu = np.zeros(n_launch)
        t_start = time.perf_counter()
        for j in range(n_launch):
            t = t_hand + j * k
            if paced:      # the K-th block of this launch exists (j + 1) K milliseconds after the first block of the run
                wait = t_start + (j + 1) * k * 1e-3 - time.perf_counter()
                if wait > 0:
                    time.sleep(wait)
            s = time.perf_counter()
            blocks[:] = stream[t:t + k]
            rc = lib.gpsx_track_loop(eng.h, blocks.ctypes.data, k, d_state, n, t, flags.ctypes.data, None)
            g = time.perf_counter()
            if rc != 0:
                raise RuntimeError(lib.gpsx_last_error(eng.h).decode())
            m = lib.gps_tracking_words_batch(table.ctypes.data, n, flags.ctypes.data, k, t, changed.ctypes.data, len(changed))
            if m:
                vals = np.ascontiguousarray(table[changed[:m], 212 + 13])
                lib.gpsx_loop_set_polarity(eng.h, d_state, changed.ctypes.data, vals.ctypes.data, m)
            e = time.perf_counter()
            lat[j], gpu[j] = e - s, g - s
        behind = time.perf_counter() - t_start - n_launch * k * 1e-3
        final = np.zeros_like(st)
        eng.d2h(final, d_state)
        eng.free(d_state)
        dopp, delay = sig_dopp[idx], sig_delay[idx]
        err = np.abs(((final["code_phase_fine"] - delay + 8184) % 16368) - 8184)
        locked = sig_tracking[idx] & (err < 4.0) & (np.abs(final["if_freq_offset_hz"] - dopp) < 60.0)
        steady = lat[n_launch // 2:]
        late = int((steady >= k * 1e-3).sum())
        words = int(table[:, 212 + 56:212 + 60].copy().view("<u4").sum())
        ref_locked = None
        if literal and (n, ms) == (256, 10000):
            try:
                from golden_util import load
                ref_mask = load("f7_steps_config5_256ch.npz")["locked"]
                ref_locked = {"count": int(ref_mask.sum()), "same_channels": int((ref_mask == locked).sum())}
            except Exception:
                ref_locked = None
        return {"metric": "closed-loop real-time tracking channels, loops on the device (k_track_loop: correlators + DLL / PLL / FLL + "
                          "false-lock check + SNR + bit synchroniser per channel and ms in one kernel, state in HBM; host: word "
                          "layer per completed navigation bit)",
                "channels": n, "signals_in_stream": n_sig, "ms": ms, "ms_per_launch": k, "launches": n_launch,
                "schedule": "GPSX_SCHED_MUX17 (receivers of four channels, each served 4 ms of every 17)" if mux17 else "GPSX_SCHED_EVERY_MS",
                "paced": bool(paced), "behind_at_end_ms": float(max(0.0, behind) * 1e3),
                "thread_on_gpu_numa_node": bool(bound), "host_workers": int(lib.gps_tracking_batch_workers()) if n >= 2048 else 1,
                "launch_p50_us": float(np.percentile(steady, 50) * 1e6), "launch_p99_us": float(np.percentile(steady, 99) * 1e6),
                "launch_max_us": float(steady.max() * 1e6), "deadline_us": k * 1000.0, "launches_over_deadline": late,
                "per_ms_p50_us": float(np.percentile(steady, 50) * 1e6 / k), "per_ms_max_us": float(steady.max() * 1e6 / k),
                "gpu_part_p50_us": float(np.percentile(gpu[n_launch // 2:], 50) * 1e6),
                "warmup_max_us": float(lat[:n_launch // 2].max() * 1e6) if n_launch > 1 else None,
                "real_time": bool(late == 0),
                "channels_handed_over_tracking": int(sig_tracking[idx].sum()), "code_and_carrier_lock": int(locked.sum()),
                "false_lock_jumps": int(final["reseed_count"].sum()), "good_words_on_the_host": words,
                "code_and_carrier_lock_in_the_reference_on_this_stream": ref_locked,
                "prepare_seconds": prep_s,
                "note": "latency of a launch = K blocks into page-locked memory -> H2D -> k_track_loop -> D2H of K flag bytes per "
                        "channel -> word layer on the host; real_time = every launch of the steady half (the second half of the "
                        "run) is back before the next one's blocks are complete (K ms)"}
    finally:
        eng.close()
        os.sched_setaffinity(0, affinity)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="+", default=[256])
    ap.add_argument("--ms", type=int, default=1200)
    ap.add_argument("--k", type=int, default=20, help="milliseconds per launch")
    ap.add_argument("--signals", type=int, default=32)
    ap.add_argument("--amp", type=float, default=0.12)
    ap.add_argument("--unpaced", action="store_true")
    ap.add_argument("--no-bind", action="store_true")
    ap.add_argument("--literal", action="store_true", help="SURVEY.md 8(d) config 5 to the letter (see device_loop)")
    ap.add_argument("--fast-synth", action="store_true", help="synthesise the stream with synth.make_if_static (long soaks)")
    ap.add_argument("--mux17", action="store_true", help="the reference's 17 ms multiplex (--k a multiple of 17)")
    args = ap.parse_args()
    for n in args.channels:
        print(json.dumps(device_loop(n, args.ms, args.k, args.signals, args.amp, not args.unpaced, not args.no_bind, args.literal,
                                     args.fast_synth, args.mux17)), flush=True)


if __name__ == "__main__":
    main()
