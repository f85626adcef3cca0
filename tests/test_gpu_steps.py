"""Step-level (Tier 2) parity on the GPU: libgpsx.so's acquisition_process / acquisition_start_* /
gps_tracking_process, driven in the firmware's own call order (tests/steps_driver.py), must take the 4-channel table
through exactly the states the reference's acquisition.c / tracking.c take it through on the same IF stream.
Golden traces: tests/golden/f7_steps_*.npz (oracle/gen_golden_steps.py, recorded from the reference's C).

Integer state (acquisition state machine, histograms, found phase/frequency, NCO accumulator, pre-track phases, PLL
check buffers, SNR sums, bit-sync counters) must match bit for bit at every millisecond.  The float loop state
(code_phase_fine, if_freq_offset_hz, dll/pll/fll memories, snr_value) is produced by the same float32 expressions
and the same libm, so it is compared bit for bit as well; the stated tolerance of SURVEY.md 8(c) (|d code_phase_fine|
<= 0.01 sample, |d if_freq_offset_hz| <= 0.5 Hz) is the fallback bar and is asserted separately.
"""
import ctypes as C
import os

import numpy as np
import pytest

import steps_driver as sd
from golden_util import fnv1a32, load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpsx_lib():
    from stm32f4_sdr_gps_amd import capi
    return capi.load_library()


def _first_mismatch(got, want, lo, hi):
    bad = np.argwhere(got[:, :, lo:hi] != want[:, :, lo:hi])
    return None if len(bad) == 0 else tuple(int(x) for x in bad[0])


@pytest.mark.parametrize("name", ["hints", "cold"])
def test_step_trace_matches_reference(gpsx_lib, name):
    from stm32f4_sdr_gps_amd import synth
    g = load(f"f7_steps_{name}.npz")
    n_ms = int(g["n_ms"])
    stream = synth.four_sv_with_nav(n_ms, seed=7)
    assert fnv1a32(stream[::97]) == int(g["stream_fnv"]), "synthetic stream differs from the one the trace was recorded on"
    C.CDLL("libc.so.6").srand(1)
    snaps = sd.run_scenario(sd.StepsLib(gpsx_lib, False), stream, g["prns"].tolist(), g["hints"].tolist(), n_ms)
    want = g["snaps"]
    # fallback bar first (so a float-only divergence is reported as such)
    fine = snaps[:, :, 60 + 80:60 + 84].copy().view("<f4")[:, :, 0]
    fine_w = want[:, :, 60 + 80:60 + 84].copy().view("<f4")[:, :, 0]
    freq = snaps[:, :, 60 + 4:60 + 8].copy().view("<f4")[:, :, 0]
    freq_w = want[:, :, 60 + 4:60 + 8].copy().view("<f4")[:, :, 0]
    assert np.abs(fine - fine_w).max() <= 0.01 and np.abs(freq - freq_w).max() <= 0.5
    # the bar: every byte of acq_data and tracking_data, and the bit-synchronisation part of nav_data
    assert _first_mismatch(snaps, want, 0, 60) is None, ("acq_data", _first_mismatch(snaps, want, 0, 60))
    assert _first_mismatch(snaps, want, 60, 212) is None, ("tracking_data", _first_mismatch(snaps, want, 60, 212))
    assert _first_mismatch(snaps, want, 212, 223) is None, ("nav_data sync", _first_mismatch(snaps, want, 212, 223))
    end = sd.summarize(snaps)
    if name == "hints":
        assert [r["found_code_phase"] for r in end] == [200, 500, 1124, 1624]
        assert all(r["trk_state"] == sd.TRK_RUN for r in end)
        assert abs(end[0]["code_phase_fine"] - 1600.0) < 1.0 and abs(end[0]["freq"] - 912.5) < 2.0


def test_step_trace_through_the_capture_interface(gpsx_lib):
    """Same scenario, blocks delivered through signal_capture_* (PM/signal_capture.h) on the engine's capture rings: the
    step calls get pinned ring pointers and read the HBM mirror.  1200 ms cover acquisition, pre-track and tracking."""
    from stm32f4_sdr_gps_amd import synth
    g = load("f7_steps_hints.npz")
    n_ms = 1200
    stream = synth.four_sv_with_nav(int(g["n_ms"]), seed=7)[:n_ms]
    C.CDLL("libc.so.6").srand(1)
    snaps = sd.run_scenario(sd.StepsLib(gpsx_lib, False), stream, g["prns"].tolist(), g["hints"].tolist(), n_ms,
                            via_capture=True)
    want = g["snaps"][:n_ms]
    assert _first_mismatch(snaps, want, 0, 223) is None, _first_mismatch(snaps, want, 0, 223)
    assert gpsx_lib.signal_capture_get_packet_cnt() == n_ms - 1      # set_time(t) each step; push made it t + 1 before
    assert {sd.summarize(snaps)[i]["trk_state"] for i in range(4)} == {sd.TRK_RUN}


def test_word_layer_on_lnav_subframes_matches_reference(gpsx_lib):
    """15 s of the 4-SV table carrying parity-correct LNAV subframes (two satellites with inverted data polarity):
    preamble search, parity, polarity detection, subframe assembly and time stamp (PM/GPS/nav_data.c:257-451) on top of
    the GPU correlators must leave every channel in exactly the reference's state after every millisecond --
    tests/golden/f7_steps_lnav.npz holds a CRC of acq_data + tracking_data + nav_data + obs_data + eph_data (the decoded
    ephemeris) per millisecond, the full state every 500 ms and the final state (oracle/gen_golden_steps.py lnav)."""
    from stm32f4_sdr_gps_amd import synth
    g = load("f7_steps_lnav.npz")
    n_ms = int(g["n_ms"])
    stream = synth.four_sv_with_lnav(n_ms, seed=7)
    assert fnv1a32(stream[::97]) == int(g["stream_fnv"])
    C.CDLL("libc.so.6").srand(1)
    crcs, checkpoints, final = sd.run_scenario(sd.StepsLib(gpsx_lib, False), stream, g["prns"].tolist(),
                                               g["hints"].tolist(), n_ms, digest=True)
    bad = np.flatnonzero(crcs != g["crcs"])
    if len(bad):   # locate the first divergence through the checkpoints
        cp = np.argwhere(checkpoints != g["checkpoints"])
        raise AssertionError(f"state diverges at ms {int(bad[0])}; first checkpoint mismatch "
                             f"{tuple(int(x) for x in cp[0]) if len(cp) else None}")
    assert np.array_equal(checkpoints, g["checkpoints"]) and np.array_equal(final, g["final"])
    nav = final[:, 212:324]
    assert final[0, 344 + 312:344 + 314].view("<u2")[0] == 1 and final[0, 344 + 315] == 2   # PRN 5: ephemeris decoder ran on a subframe 2
    assert int(nav[0, 68:70].view("<u2")[0]) == 1 and nav[0, 14] == 1          # PRN 5: a whole subframe, polarity known
    assert nav[3, 13] == 1 and nav[3, 14] == 1                                  # PRN 30: inverted polarity found and confirmed
    assert int(nav[0, 60:64].view("<u4")[0]) == 11260                           # the subframe's bit-edge time stamp


def test_time_source_is_overridable_weak_symbol(gpsx_lib):
    import subprocess
    out = subprocess.check_output(["nm", "-D", os.path.join(os.path.dirname(gpsx_lib._name), "libgpsx.so")], text=True)
    weak = {l.split()[-1] for l in out.splitlines() if " W " in l}
    assert {"signal_capture_get_packet_cnt", "gps_nav_data_analyse_new_code", "gps_nav_data_words_detection",
            "gps_nav_data_decode_subframe"} <= weak


def test_batched_tracking_step_matches_per_channel_reference(gpsx_lib):
    """gps_tracking_process_batch: four channels served EVERY millisecond by one launch each for pre-tracking and
    E/P/L; per channel the state must follow the reference's gps_tracking_process run alone on that channel
    (tests/golden/f7_steps_continuous.npz: one reference library instance per channel, index = t & 3)."""
    from stm32f4_sdr_gps_amd import synth
    g = load("f7_steps_continuous.npz")
    n_ms = int(g["n_ms"])
    stream = synth.four_sv_with_nav(n_ms, seed=7)
    assert fnv1a32(stream[::97]) == int(g["stream_fnv"])
    steps = sd.StepsLib(gpsx_lib, False)
    gpsx_lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
    gpsx_lib.gps_tracking_process_batch.restype = None
    C.CDLL("libc.so.6").srand(1)
    table = np.stack([sd.preset_channel(steps, int(p), int(f), int(c))
                      for p, f, c in zip(g["prns"], g["found_freq"], g["found_phase"])])
    snaps = np.zeros((n_ms, 4, sd.SNAP), np.uint8)
    for t in range(n_ms):
        steps.set_time(t)
        blk = np.ascontiguousarray(stream[t])
        gpsx_lib.gps_tracking_process_batch(table.ctypes.data, 4, blk.ctypes.data, t & 3)
        snaps[t] = sd.snapshot(table)
    want = g["snaps"]
    assert _first_mismatch(snaps, want, 0, 212) is None, _first_mismatch(snaps, want, 0, 212)
    # nav_data as far as the snapshots go (bit synchronisation + polarity flag, which the word layer sets when it has
    # seen two inverted preambles: on this random-bit stream that happens around t = 1980 on one channel)
    assert _first_mismatch(snaps, want, 212, sd.SNAP) is None, _first_mismatch(snaps, want, 212, sd.SNAP)
    assert want[:, :, 212 + 13].any()
    end = sd.summarize(snaps)
    assert all(r["trk_state"] == sd.TRK_RUN for r in end)
    for r, truth in zip(end, (1600.0, 4000.0, 9000.0, 13000.0)):
        assert abs(r["code_phase_fine"] - truth) < 1.5


_POOL_SCRIPT = r"""
import ctypes as C, os, sys, zlib
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import steps_driver as sd
from stm32f4_sdr_gps_amd import capi, synth
n, ms, n_sig = 4096, 400, 8
lib = capi.load_library()
steps = sd.StepsLib(lib, False)
lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
lib.gps_tracking_process_batch.restype = None
sats = [synth.Sat(i + 1, -2000.0 + 450.0 * i, (2000.0 * i + 37.0) % 16368, 0.25, 0.3 * i) for i in range(n_sig)]
stream = synth.make_if(ms, sats, noise_amp=1.0, seed=9)
per_sig = np.stack([sd.preset_channel(steps, s.prn, int(round(s.doppler_hz / 500.0)) * 500, int(s.delay_samples // 8) % 2046) for s in sats])
table = np.ascontiguousarray(per_sig[np.arange(n) % n_sig])
crcs = []
for t in range(ms):
    steps.set_time(t)
    lib.gps_tracking_process_batch(table.ctypes.data, n, stream[t].ctypes.data, t & 3)
    if t % 25 == 24:
        crcs.append(zlib.crc32(table[:, :664].tobytes()))
state = table[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0]
print("RESULT", int((state == sd.TRK_RUN).sum()), *crcs)
"""


def test_batched_step_on_worker_threads_equals_the_single_threaded_step():
    """gps_tracking_process_batch spreads its per-channel host loops over worker threads from 2048 channels on
    (gpsx_steps.cpp StepPool) and, from 65536 tracked channels on, runs them piece by piece while the GPU correlates the
    next pieces (gpsx_track_epl_batch_chunked; the threshold is lowered here).  4096 channels, 400 ms from pre-tracking into
    tracking and nav-bit synchronisation: the whole channel table (acq, tracking, nav and observation state of every
    channel, CRC every 25 ms) must be the same with 1, 3 and 7 workers, and with 5 and 7 workers in the overlapped form
    (4 pieces once 1024 channels track; 16 pieces from the first tracked channel on, i.e. with pre-tracking and tracking
    channels mixed and pieces smaller than the worker count) -- a channel's state may not depend on which thread served
    it, on how the ranges were cut or on the order its correlators went to the GPU in.  (Strong signals: the one shared
    state of the path, rand() in the PLL's false-lock reseed, is never drawn.)  Separate processes: the pool is sized
    once per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for threads, extra in (("1", {}), ("3", {}), ("7", {}),
                           ("5", {"GPSX_STEP_OVERLAP_FROM": "1024", "GPSX_STEP_CHUNKS": "4"}),
                           ("7", {"GPSX_STEP_OVERLAP_FROM": "1", "GPSX_STEP_CHUNKS": "16"})):
        env = dict(os.environ, GPSX_STEP_THREADS=threads, **extra)
        r = subprocess.run([sys.executable, "-c", _POOL_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert r.returncode == 0 and line, r.stderr[-2000:]
        outs.append(line[0])
    assert all(o == outs[0] for o in outs), outs
    assert int(outs[0].split()[1]) >= 4096 * 7 // 8      # and the channels did reach tracking


_CONFIG5_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import steps_driver as sd
from golden_util import fnv1a32, load
from stm32f4_sdr_gps_amd import capi, synth
g = load("f7_steps_config5_64ch.npz")
n_ms = int(g["n_ms"])
sats, chans, seed = sd.config5_64ch_scenario()
assert np.array_equal(np.array(chans, np.int32), g["chans"])
stream = synth.make_if(n_ms, sats, noise_amp=1.0, seed=seed)
assert fnv1a32(stream[::97]) == int(g["stream_fnv"]), "synthetic stream differs from the one the trace was recorded on"
lib = capi.load_library()
steps = sd.StepsLib(lib, False)
lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
lib.gps_tracking_process_batch.restype = None
# libc's rand() is process-global and the ROCm runtime draws from it too (libamd_comgr, hundreds of draws whenever a code
# object is loaded: tools/experiments/randshim.c shows them).  A host that wants the reference's reseed sequence seeds AFTER
# the kernels of the path have been loaded: here every kernel the step launches runs once on another context first.
warm = capi.Engine(0)
jobs = np.zeros(1, capi.JOB_DTYPE); jobs[0] = (0, 1, 5, capi.IF_HZ + 900.0, 0, 0, 2046)
warm.acq_jobs(stream[:1], jobs)
st = np.zeros(64, capi.TRK_DTYPE); st["prn"] = 1 + np.arange(64) % 32
warm.track_epl(stream[0], st); warm.rewind(st, np.full(64, 3, np.uint8))
warm.close()
lib.gps_fill_summ_table()
C.CDLL("libc.so.6").srand(1)
table = np.stack([sd.preset_channel(steps, *c) for c in chans])
for t in range(n_ms):
    steps.set_time(t)
    lib.gps_tracking_process_batch(table.ctypes.data, len(chans), stream[t].ctypes.data, t & 3)
    crc = sd.snapshot_crcs(table)
    bad = np.flatnonzero(crc != g["crcs"][t])
    if len(bad):
        print("MISMATCH ms", t, "channels", bad[:8].tolist(), "reseeds so far", [tuple(r) for r in g["reseeds"] if r[0] <= t][-3:])
        sys.exit(1)
    if (t + 1) % 100 == 0:
        assert np.array_equal(sd.snapshot(table), g["checkpoints"][(t + 1) // 100 - 1])
assert np.array_equal(sd.snapshot(table), g["final"])
state = table[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0]
print("RESULT", int((state == sd.TRK_RUN).sum()), np.flatnonzero(state != sd.TRK_RUN).tolist(), len(g["reseeds"]), len(set(g["reseeds"][:, 1].tolist())), lib.gps_tracking_batch_workers())
"""


@pytest.mark.parametrize("threads, extra", [("1", {}), ("3", {"GPSX_STEP_THREADS_FROM": "16"}),
                                             ("7", {"GPSX_STEP_THREADS_FROM": "16", "GPSX_STEP_OVERLAP_FROM": "8", "GPSX_STEP_CHUNKS": "4"})])
def test_batched_step_64_channels_follows_the_reference_incl_false_lock_reseeds(threads, extra):
    """The batched step beyond the reference's four channels, against the reference itself: 64 channels on the closed-loop
    bench's signal table (tests/steps_driver.py config5_64ch_scenario; golden: one private instance of the reference's
    tracking.c / nav_data.c per channel, called in channel order every millisecond, oracle/gen_golden_steps.py config5),
    1500 ms, every channel's 226 state bytes after every millisecond.  14 of the channels were handed over on the wrong
    Doppler bin or on weak signals and go through the PLL's false-lock reseed -- 21 draws from libc's rand()
    (tracking.c:309-326) -- and channels 0 and 32 (PRN 1 at code phase 0) never leave pre-tracking, in the reference
    and here alike (tracking.c: a settled phase of 0 reads as "none": the 1 signal in 32 of bench.py's closed-loop
    ladder that never locks).  On ONE thread, and on 3 and 7 worker threads (threshold lowered; 7: overlapped with the
    correlators in 4 pieces): workers only detect a false lock, the draws are made afterwards in channel order
    (gpsx_steps.cpp false_lock_detect / finish_deferred), so the trace is the reference's whichever thread served a
    channel."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPSX_STEP_THREADS=threads, **extra)
    r = subprocess.run([sys.executable, "-c", _CONFIG5_SCRIPT, root], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert r.returncode == 0 and line, (r.stdout[-1500:], r.stderr[-1500:])
    f = line[0].split()
    assert int(f[1]) == 62 and "[0, 32]" in line[0]
    assert line[0].split("]")[1].split() == ["21", "14", threads]


def _load_the_paths_kernels_then_seed_rand(stream):
    """libc's rand() is process-global and the ROCm runtime draws from it too (libamd_comgr, whenever a code object is
    loaded): run every kernel the batched step launches once on another context, THEN srand(1) -- from there on the only
    draws are the false-lock reseeds, in the reference's order."""
    from stm32f4_sdr_gps_amd import capi
    warm = capi.Engine(0)
    jobs = np.zeros(1, capi.JOB_DTYPE)
    jobs[0] = (0, 1, 5, capi.IF_HZ + 900.0, 0, 0, 2046)
    warm.acq_jobs(stream[:1], jobs)
    st = np.zeros(256, capi.TRK_DTYPE)
    st["prn"] = 1 + np.arange(256) % 32
    warm.track_epl(stream[0], st)
    warm.rewind(st, np.full(256, 3, np.uint8))
    warm.close()
    capi.load_library().gps_fill_summ_table()
    C.CDLL("libc.so.6").srand(1)


def test_config5_to_the_letter_256_channels_10_seconds_follow_the_reference(gpsx_lib):
    """BASELINE.json configs[4] as SURVEY.md 8(d) words it -- 256 channels on 256 distinct signals (PRN (i mod 32) + 1,
    -5000 + 39 i Hz, 61 i samples), 10 000 ms -- through gps_tracking_process_batch, against the reference run on the same
    stream (tests/golden/f7_steps_config5_256ch.npz: one private instance of its step sources per channel,
    oracle/gen_golden_steps.py config5_literal): every channel's 226 state bytes every 100 ms and at the end, and the
    reference's own lock count -- 229 of the 256 (eight signals share each PRN and the hand-over is only good to the
    byte: 27 channels settle on a wrong phase or carrier in the reference, and the same 27 here)."""
    g = load("f7_steps_config5_256ch.npz")
    n_ms = int(g["n_ms"])
    stream, chans, dopp, delay = sd.config5_literal_scenario(n_ms)
    assert fnv1a32(stream[::97]) == int(g["stream_fnv"]), "synthetic stream differs from the one the trace was recorded on"
    assert np.array_equal(np.array(chans, np.int32), g["chans"])
    steps = sd.StepsLib(gpsx_lib, False)
    gpsx_lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
    gpsx_lib.gps_tracking_process_batch.restype = None
    _load_the_paths_kernels_then_seed_rand(stream)
    table = np.stack([sd.preset_channel(steps, *c) for c in chans])
    for t in range(n_ms):
        steps.set_time(t)
        gpsx_lib.gps_tracking_process_batch(table.ctypes.data, 256, stream[t].ctypes.data, t & 3)
        if (t + 1) % 100 == 0:
            bad = np.flatnonzero(sd.snapshot_crcs(table) != g["crcs"][(t + 1) // 100 - 1])
            assert len(bad) == 0, (t, bad[:8].tolist())
    assert np.array_equal(sd.snapshot(table), g["final"])
    locked = sd.lock_mask(table, dopp, delay)
    assert np.array_equal(locked, g["locked"]) and int(locked.sum()) == 229


_HOOK_SHIM = r"""
#include <pthread.h>
#include <stdint.h>
static uint32_t tick;
static int calls, foreign;
static pthread_t first;
static int have_first;
uint32_t signal_capture_get_packet_cnt(void)        /* the host's own time source: overrides libgpsx's weak default */
{
  pthread_t me = pthread_self();
  if (!have_first) { first = me; have_first = 1; }
  else if (!pthread_equal(me, first)) __sync_fetch_and_add(&foreign, 1);
  __sync_fetch_and_add(&calls, 1);
  return tick;
}
void shim_set(uint32_t t) { tick = t; }
int shim_calls(void) { return calls; }
int shim_foreign(void) { return foreign; }
"""

_HOOK_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import steps_driver as sd
from stm32f4_sdr_gps_amd import capi, synth
shim = C.CDLL(sys.argv[2])
shim.shim_set.argtypes = [C.c_uint32]
lib = capi.load_library()
steps = sd.StepsLib(lib, False)
lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
lib.gps_tracking_process_batch.restype = None
n, ms = 4096, 260
sats = [synth.Sat(i + 1, -2000.0 + 450.0 * i, (2000.0 * i + 37.0) % 16368, 0.25, 0.3 * i) for i in range(8)]
stream = synth.make_if(ms, sats, noise_amp=1.0, seed=9)
per_sig = np.stack([sd.preset_channel(steps, s.prn, int(round(s.doppler_hz / 500.0)) * 500, int(s.delay_samples // 8) % 2046) for s in sats])
table = np.ascontiguousarray(per_sig[np.arange(n) % 8])
per_step = []
for t in range(ms):
    shim.shim_set(t)
    before = shim.shim_calls()
    lib.gps_tracking_process_batch(table.ctypes.data, n, stream[t].ctypes.data, t & 3)
    per_step.append(shim.shim_calls() - before)
stamps = table[:, 60 + 76:60 + 80].copy().view("<u4")[:, 0]     # prev_track_timestamp of channels that track
print("RESULT", max(per_step), min(per_step), shim.shim_foreign(), int(stamps.max()), lib.gps_tracking_batch_workers(),
      lib.gps_tracking_batch_last_workers())
"""


def test_batched_step_under_an_overridden_time_source_calls_it_once_from_the_calling_thread(tmp_path):
    """the weak hooks are the host's.  A host that brings its own signal_capture_get_packet_cnt (here: an LD_PRELOADed
    definition that counts its calls and remembers its first caller) must see it called from the thread that called the
    step, never from a worker and never concurrently: the batched step reads the tick ONCE per step and, seeing the symbol
    resolve outside the library, runs without workers -- at 4096 channels, where it would otherwise use its pool."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hook_shim.c"
    src.write_text(_HOOK_SHIM)
    so = tmp_path / "libhookshim.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", str(src), "-o", str(so), "-lpthread"])
    env = dict(os.environ, LD_PRELOAD=str(so), GPSX_STEP_THREADS="6")
    r = subprocess.run([sys.executable, "-c", _HOOK_SCRIPT, root, str(so)], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert r.returncode == 0 and line, r.stderr[-2000:]
    most, least, foreign, last_stamp, workers, used = (int(x) for x in line[0].split()[1:])
    assert (most, least) == (1, 1), line[0]          # one read per step, whatever the channel count
    assert foreign == 0
    assert last_stamp == 259 and workers == 6 and used == 1   # the channels saw the host's tick; the pool exists, and was not used
    # the same run without the override: the pool is used (only the worker count is looked at: the script cannot set the
    # library's own counter through the shim)
    env2 = dict(os.environ, GPSX_STEP_THREADS="6")
    r2 = subprocess.run([sys.executable, "-c", _HOOK_SCRIPT, root, str(so)], env=env2, capture_output=True, text=True, timeout=600)
    line2 = [l for l in r2.stdout.splitlines() if l.startswith("RESULT")]
    assert r2.returncode == 0 and line2, r2.stderr[-2000:]
    assert int(line2[0].split()[-1]) == 6
