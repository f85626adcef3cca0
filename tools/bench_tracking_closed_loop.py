#!/usr/bin/env python3
"""BASELINE.json configs[4]: N concurrent tracking channels in CLOSED LOOP on one shared IF stream, sustained real time.

Every millisecond: gps_tracking_process_batch() = one pre-track job list / one E/P/L launch for all channels, then the
reference's DLL / PLL / FLL float loops per channel on the host (csrc/gpsx_steps.cpp; from 2048 channels on spread over the
calling thread's CPUs -- this script pins itself to the GPU's NUMA node first, gpsx_bind_thread_to_device).  The stream
carries one signal per channel (PRN (i mod 32) + 1, Doppler -5000 + 39 i Hz, delay 61 i samples)
or, with --signals S, S signals shared by the channels (channel i tracks signal i mod S: thousands of channels without
synthesising thousands of signals); channels start from the acquisition result a cold start would hand over (code phase
to half a chip, Doppler to the 500 Hz bin).
The steps are paced at the stream's rate (step t starts no earlier than t ms after the first; --unpaced: back to back).
Reports the per-millisecond step latency -- real time means EVERY step after the warm-up < 1 ms: p50 / p99 / max of the
steady half are reported, `real_time` is judged on the max -- and how many channels hold code and carrier lock at the end.

Importable: closed_loop(channels, ms, ...) -> dict (bench.py's `tracking.closed_loop`)."""
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

_stream_cache = {}


def _cpu_throttle():
    """(periods throttled, microseconds throttled) of this container's CPU quota so far (cgroup v2 cpu.stat), or None"""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except Exception:
        return None


def _realtime_thread():
    """What a real-time host does for its per-millisecond thread, as far as this container lets it: resident pages
    (mlockall(MCL_CURRENT)) and the FIFO scheduling class, so that another tenant's runnable thread does not take the CPU in the
    middle of a step.  Returns what took (a shared box usually refuses the scheduling class: reported, not hidden) and a
    function that undoes it."""
    out = {"mlockall_current": False, "sched": "SCHED_OTHER"}
    libc = C.CDLL("libc.so.6", use_errno=True)
    try:
        out["mlockall_current"] = libc.mlockall(1) == 0                     # MCL_CURRENT
        if not out["mlockall_current"]:
            out["mlockall_errno"] = C.get_errno()
    except Exception as exc:   # noqa: BLE001
        out["mlockall_error"] = repr(exc)
    try:
        os.sched_setscheduler(0, os.SCHED_FIFO, os.sched_param(10))
        out["sched"] = "SCHED_FIFO 10"
    except (PermissionError, OSError) as exc:
        out["sched_refused"] = repr(exc)

    def undo():
        try:
            os.sched_setscheduler(0, os.SCHED_OTHER, os.sched_param(0))
        except Exception:   # noqa: BLE001
            pass
        try:
            libc.munlockall()
        except Exception:   # noqa: BLE001
            pass
    return out, undo


def closed_loop(channels=256, ms=2000, amp=0.12, signals=0, bind=True, paced=True, literal=False, realtime_thread=False):
    """literal: SURVEY.md 8(d) config 5 to the letter -- one signal per channel at -5000 + 39 i Hz exactly (no 7 Hz offset),
    synthesised by synth.make_if_static (the stream tests/golden/f7_steps_config5_256ch.npz was recorded on when channels =
    256 and ms = 10000: the reference's own lock count on it is reported beside the engine's)."""
    import steps_driver as sd
    from stm32f4_sdr_gps_amd import capi, synth
    n = channels
    n_sig = signals if 0 < signals < n and not literal else n
    lib = capi.load_library()
    affinity = os.sched_getaffinity(0)
    bound = False
    if bind:
        e = capi.Engine(0)
        bound = e.bind_thread_to_device()      # before the first batched step: its worker pool is sized from this thread's CPUs
        e.close()
    try:
        steps = sd.StepsLib(lib, False)
        lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
        lib.gps_tracking_process_batch.restype = None
        sig_prn = [(i % 32) + 1 for i in range(n_sig)]
        sig_dopp = [-5000.0 + 39.0 * i + (0.0 if literal else 7.0) for i in range(n_sig)]
        sig_delay = [(61.0 * i) % 16368 for i in range(n_sig)]
        key = (n_sig, ms, amp, literal)
        t0 = time.time()
        if key not in _stream_cache:
            _stream_cache.clear()
            sats = [synth.Sat(sig_prn[i], sig_dopp[i], sig_delay[i], amp, 0.37 * i) for i in range(n_sig)]
            _stream_cache[key] = (synth.make_if_static if literal else synth.make_if)(ms, sats, noise_amp=1.0, seed=5)
        stream = _stream_cache[key]
        gen_s = time.time() - t0
        per_sig = np.stack([sd.preset_channel(steps, sig_prn[i], int(round(sig_dopp[i] / 500.0)) * 500,
                                              int(sig_delay[i] // 8) % 2046) for i in range(n_sig)])
        table = np.ascontiguousarray(per_sig[np.arange(n) % n_sig])
        dopp = np.array(sig_dopp)[np.arange(n) % n_sig]
        delay = np.array(sig_delay)[np.arange(n) % n_sig]
        lat = np.zeros(ms)
        n_trk = np.zeros(ms, np.int64)
        rt_setup, rt_undo = (None, None)
        if realtime_thread and n < 2048:      # (one stepping thread below 2048 channels: only that case is raised to FIFO)
            rt_setup, rt_undo = _realtime_thread()
        thr0 = _cpu_throttle()
        t_start = time.perf_counter()
        for t in range(ms):
            steps.set_time(t)
            blk = stream[t]
            if paced:      # block t of the IF stream exists t milliseconds after the start, not before
                wait = t_start + t * 1e-3 - time.perf_counter()
                if wait > 0:
                    time.sleep(wait)
            s = time.perf_counter()
            lib.gps_tracking_process_batch(table.ctypes.data, n, blk.ctypes.data, t & 3)
            lat[t] = time.perf_counter() - s
        behind = time.perf_counter() - t_start - ms * 1e-3
        if rt_undo:
            rt_undo()
        thr1 = _cpu_throttle()
        fine = table[:, 60 + 80:60 + 84].copy().view("<f4")[:, 0]
        freq = table[:, 60 + 4:60 + 8].copy().view("<f4")[:, 0]
        state = table[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0]
        err = np.abs(((fine - delay + 8184) % 16368) - 8184)
        locked = (state == sd.TRK_RUN) & (err < 4.0) & (np.abs(freq - dopp) < 60.0)
        steady = lat[ms // 2:]
        late = int((steady >= 1e-3).sum())
        del n_trk
        ref_locked = None
        if literal and (n, ms) == (256, sd.CONFIG5_LITERAL_MS):
            try:
                from golden_util import load
                ref_locked = int(load("f7_steps_config5_256ch.npz")["locked"].sum())
            except Exception:
                ref_locked = None
        return {"metric": "closed-loop real-time tracking channels (gps_tracking_process_batch per ms: work lists, one E/P/L "
                          "launch, DLL / PLL / FLL + nav-bit logic per channel on the host)",
                "channels": n, "signals_in_stream": n_sig, "ms": ms, "signal_amp": amp,
                "paced_at_1ms": bool(paced), "behind_at_end_ms": float(max(0.0, behind) * 1e3),
                "cpu_quota_throttled_ms_during_run": None if not (thr0 and thr1) else (thr1[1] - thr0[1]) / 1e3,
                "cpu_quota_throttled_periods_during_run": None if not (thr0 and thr1) else thr1[0] - thr0[0],
                "host_workers": int(lib.gps_tracking_batch_workers()) if n >= 2048 else 1,
                "thread_on_gpu_numa_node": bool(bound), "realtime_thread": rt_setup,
                "p50_us": float(np.percentile(steady, 50) * 1e6), "p99_us": float(np.percentile(steady, 99) * 1e6),
                "max_us": float(steady.max() * 1e6), "steps_over_1ms": late,
                "slowest_steady_steps_ms": [int(i) + ms // 2 for i in np.argsort(steady)[::-1][:4]],
                "warmup_max_us": float(lat[:ms // 2].max() * 1e6),
                "real_time": bool(late == 0),
                "tracking_state": int((state == sd.TRK_RUN).sum()), "code_and_carrier_lock": int(locked.sum()),
                "code_and_carrier_lock_in_the_reference_on_this_stream": ref_locked,
                "not_locked": np.flatnonzero(~locked)[:64].tolist(),
                "median_code_error_samples": float(np.median(err)), "synth_seconds": gen_s,
                "note": "latencies of the second half of the run (steady state: every channel past pre-tracking); "
                        "warmup_max_us = the worst step of the first half (graph instantiation, buffer growth, the "
                        "pre-tracking job lists); real_time = no steady step took 1 ms or more.  paced_at_1ms: step t starts "
                        "no earlier than t ms after the first (the stream's own rate -- a receiver is fed one block per "
                        "millisecond; back to back, the worker threads of small counts never sleep and the container's CPU "
                        "quota throttles the process, --unpaced)"}
    finally:
        os.sched_setaffinity(0, affinity)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="+", default=[256])
    ap.add_argument("--ms", type=int, default=2000)
    ap.add_argument("--amp", type=float, default=0.12)
    ap.add_argument("--signals", type=int, default=0,
                    help="satellites in the stream (default: one per channel); with fewer, channel i tracks signal "
                         "i mod signals -- a cheap way to load thousands of channels without synthesising thousands of signals")
    ap.add_argument("--no-bind", action="store_true")
    ap.add_argument("--unpaced", action="store_true", help="steps back to back instead of one per millisecond")
    ap.add_argument("--literal", action="store_true", help="SURVEY.md 8(d) config 5 to the letter (see closed_loop)")
    args = ap.parse_args()
    for n in args.channels:
        print(json.dumps(closed_loop(n, args.ms, args.amp, args.signals, not args.no_bind, not args.unpaced, args.literal)), flush=True)


if __name__ == "__main__":
    main()
