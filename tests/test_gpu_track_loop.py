"""The tracking loops on the device (gpsx_track_loop: E/P/L correlators + DLL / PLL / FLL + false-lock check + SNR + the
20 ms bit synchroniser in ONE kernel, K milliseconds per launch, channel state resident in HBM) against
  (a) the reference's own closed-loop traces (tests/golden/f7_steps_continuous.npz: its gps_tracking_process served every
      millisecond), within the tolerance SURVEY.md 8(c) states for float loop outputs computed with another libm --
      |d code_phase_fine| <= 0.01 sample, |d if_freq_offset_hz| <= 0.5 Hz, over the whole trace;
  (b) the CPU oracle, teacher-forced: the six accumulators of EVERY millisecond, from the state the device itself had, bit
      for bit -- the integer part of the path has no tolerance;
  (c) itself under other launch lengths (K = 1, 3, 20, 64: bit for bit) and the host mode of this library
      (gps_tracking_process_batch) on thousands of channels."""
import ctypes as C

import numpy as np
import pytest

import steps_driver as sd
from golden_util import fnv1a32, load

pytestmark = pytest.mark.gpu

TOL_FINE, TOL_HZ = 0.01, 0.5


@pytest.fixture(scope="module")
def eng():
    from stm32f4_sdr_gps_amd import capi
    e = capi.Engine(0)
    yield e
    e.close()


def _states_from_records(lib, records, seeds=None):
    from stm32f4_sdr_gps_amd import capi
    st = np.zeros(len(records), capi.LOOP_DTYPE)
    for i, rec in enumerate(records):
        rec = np.ascontiguousarray(rec)
        lib.gpsx_loop_state_from_channel(rec.ctypes.data, int(seeds[i]) if seeds is not None else i + 1, st[i:i + 1].ctypes.data)
    return st


def _run(eng, stream, st0, t0, t1, k, want_trace=True):
    """device loop from tick t0 to t1 in launches of k milliseconds; returns (flags [t1 - t0, n], trace, final states)"""
    n = len(st0)
    d = eng.malloc(st0.nbytes)
    try:
        eng.h2d(d, st0)
        flags, trace = [], []
        t = t0
        while t < t1:
            kk = min(k, t1 - t)
            f, tr = eng.track_loop(stream[t:t + kk], d, n, t, want_trace=want_trace)
            flags.append(f)
            trace.append(tr)
            t += kk
        out = np.zeros_like(st0)
        eng.d2h(out, d)
    finally:
        eng.free(d)
    return np.concatenate(flags), (np.concatenate(trace) if want_trace else None), out


def _golden_start(g):
    """first tick t0 with t0 & 3 == 0 from which all four channels of the trace are in GPS_TRACKING_RUN, and the channel
    records (snapshot bytes + PRN) of the millisecond before it"""
    snaps = g["snaps"]
    state = snaps[:, :, 60 + 148:60 + 152].copy().view("<i4")[:, :, 0]
    first = int(np.flatnonzero((state == sd.TRK_RUN).all(axis=1))[0])
    t0 = (first + 1 + 3) // 4 * 4 + 8
    recs = np.zeros((4, sd.CH_SIZE), np.uint8)
    recs[:, :sd.SNAP] = snaps[t0 - 1]
    recs[:, 664] = g["prns"]
    return t0, recs


def test_device_loop_follows_the_reference_trace_within_the_stated_tolerance(eng, oracle):
    from stm32f4_sdr_gps_amd import synth
    g = load("f7_steps_continuous.npz")
    n_ms = int(g["n_ms"])
    stream = synth.four_sv_with_nav(n_ms, seed=7)
    assert fnv1a32(stream[::97]) == int(g["stream_fnv"])
    t0, recs = _golden_start(g)
    st0 = _states_from_records(eng.lib, recs)
    flags, trace, final = _run(eng, stream, st0, t0, n_ms, 20)
    want = g["snaps"][t0:]
    fine_w = want[:, :, 60 + 80:60 + 84].copy().view("<f4")[:, :, 0]
    freq_w = want[:, :, 60 + 4:60 + 8].copy().view("<f4")[:, :, 0]
    accum_w = want[:, :, 60 + 8:60 + 12].copy().view("<u4")[:, :, 0]
    d_fine, d_freq = np.abs(trace["code_phase_fine"] - fine_w), np.abs(trace["if_freq_offset_hz"] - freq_w)
    print("max |d fine|", d_fine.max(), "max |d freq|", d_freq.max(), "ms with identical floats",
          int(((d_fine == 0) & (d_freq == 0)).all(axis=1).sum()), "of", len(want),
          "identical NCO accumulators", int((trace["if_freq_accum"] == accum_w).all(axis=1).sum()))
    assert d_fine.max() <= TOL_FINE and d_freq.max() <= TOL_HZ
    # (b) teacher-forced: every millisecond's accumulators from the state the device had going into it
    chips = [oracle.ca_code(int(p)) for p in g["prns"]]
    for c in range(4):
        fine, freq, acc = float(st0["code_phase_fine"][c]), float(st0["if_freq_offset_hz"][c]), int(st0["if_freq_accum"][c])
        for i in range(len(trace)):
            iq, acc_out = oracle.track_epl(stream[t0 + i], chips[c], fine, freq, acc)
            assert np.array_equal(iq, trace["iq"][i, c]) and acc_out == int(trace["if_freq_accum"][i, c]), (c, t0 + i)
            fine, freq, acc = float(trace["code_phase_fine"][i, c]), float(trace["if_freq_offset_hz"][i, c]), acc_out
    # the bit synchroniser: integer logic on the prompt signs -- the reference's flags and counters at the end of the trace
    end = g["snaps"][-1]
    assert np.array_equal(final["period_sync_ok_flag"], end[:, 212]) and final["period_sync_ok_flag"].any()
    assert np.array_equal(final["old_swap_time"], end[:, 216:220].copy().view("<u4")[:, 0])
    assert np.array_equal(final["right_period_cnt"], end[:, 213])
    assert np.array_equal(flags[:, :] & 8, np.where(want[:, :, 212] > 0, 8, 0))          # period sync, every millisecond
    # prompt signs: flag bit 0 of every millisecond = sign of the traced IP
    assert np.array_equal(flags & 1, (trace["iq"][:, :, 2] > 0).astype(np.uint8))
    # one navigation bit per 20 ms while synchronised, none otherwise
    for c in range(4):
        on = want[:, c, 212] > 0
        done = (flags[:, c] & 2) > 0
        assert abs(int(done.sum()) - int(on.sum()) // 20) <= 2 + int(np.abs(np.diff(on.astype(int))).sum()), c
    # the SNR estimate, the only consumer of the logarithm
    snr_w = end[:, 60 + 132:60 + 136].copy().view("<f4")[:, 0]
    assert np.allclose(final["snr_value"], snr_w, atol=1e-3)


@pytest.mark.parametrize("k", [1, 3, 64])
def test_launch_length_does_not_change_a_bit(eng, k):
    from stm32f4_sdr_gps_amd import synth
    g = load("f7_steps_continuous.npz")
    stream = synth.four_sv_with_nav(int(g["n_ms"]), seed=7)
    t0, recs = _golden_start(g)
    st0 = _states_from_records(eng.lib, recs)
    t1 = t0 + 400
    f20, tr20, end20 = _run(eng, stream, st0, t0, t1, 20)
    f, tr, end = _run(eng, stream, st0, t0, t1, k)
    assert np.array_equal(f, f20) and tr.tobytes() == tr20.tobytes() and end.tobytes() == end20.tobytes()


def test_device_loop_on_thousands_of_channels_against_the_host_mode(eng):
    """20 000 channels on eight signals (cpw = 4 channels per wave, a ragged last wave): the host mode --
    gps_tracking_process_batch, the reference's loops on the CPU behind every correlator launch -- takes all of them through
    pre-tracking into tracking; at tick 200 the channels are handed to the device loop (gpsx_loop_state_from_channel) and both
    modes go on for 240 ms.  Same tolerance as against the reference; the NCO accumulators and the bit synchroniser's
    state are integers and must agree exactly wherever the floats did not part."""
    from stm32f4_sdr_gps_amd import capi, synth
    n, n_sig, t_hand, t_end = 20003, 8, 200, 440
    lib = eng.lib
    steps = sd.StepsLib(lib, False)
    lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
    lib.gps_tracking_process_batch.restype = None
    sats = [synth.Sat(i + 1, -2000.0 + 450.0 * i, (2000.0 * i + 37.0) % 16368, 0.25, 0.3 * i) for i in range(n_sig)]
    stream = synth.make_if(t_end, sats, noise_amp=1.0, seed=9)
    per_sig = np.stack([sd.preset_channel(steps, s.prn, int(round(s.doppler_hz / 500.0)) * 500, int(s.delay_samples // 8) % 2046)
                        for s in sats])
    table = np.ascontiguousarray(per_sig[np.arange(n) % n_sig])
    for t in range(t_hand):
        steps.set_time(t)
        lib.gps_tracking_process_batch(table.ctypes.data, n, stream[t].ctypes.data, t & 3)
    tracking = table[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0] == sd.TRK_RUN
    assert tracking.sum() >= n * 7 // 8
    st0 = _states_from_records(lib, table)
    _, _, final = _run(eng, stream, st0, t_hand, t_end, 24, want_trace=False)
    assert eng.lib.gpsx_last_kernel(eng.h) == b"k_track_loop"
    for t in range(t_hand, t_end):
        steps.set_time(t)
        lib.gps_tracking_process_batch(table.ctypes.data, n, stream[t].ctypes.data, t & 3)
    host = _states_from_records(lib, table)
    m = tracking
    d_fine = np.abs(final["code_phase_fine"][m] - host["code_phase_fine"][m])
    d_freq = np.abs(final["if_freq_offset_hz"][m] - host["if_freq_offset_hz"][m])
    same = (d_fine == 0) & (d_freq == 0)
    print("channels", int(m.sum()), "bit-identical floats", int(same.sum()), "max |d fine|", d_fine.max(), "max |d freq|", d_freq.max())
    assert d_fine.max() <= TOL_FINE and d_freq.max() <= TOL_HZ
    assert same.mean() > 0.5
    for f in ("if_freq_accum", "period_sync_ok_flag", "right_period_cnt", "old_swap_time", "pll_bad_state_cnt", "fll_old_i",
              "fll_old_q", "i_part_summ", "q_part_summ", "snr_summ_cnt", "code_filt_cnt"):
        assert np.array_equal(final[f][m][same], host[f][m][same]), f
    assert np.array_equal(final["pll_check_buf"][m][same], host["pll_check_buf"][m][same])


def test_device_loop_false_lock_jump_and_bad_prn(eng):
    """A channel handed over 300 Hz off a clean carrier flips signs inside every 4 ms group: after 10 + 81 groups the
    false-lock detector must move it to found_freq_offset_hz -+ 250 Hz, at least 200 Hz from where it stood, drawing from
    the channel's own generator (flag bit 4, reseed_count); and a PRN outside 1..210 is reported by the call."""
    from stm32f4_sdr_gps_amd import capi, synth
    sats = [synth.Sat(7, 1300.0, 4000.0, 0.8, 0.0)]
    stream = synth.make_if(600, sats, noise_amp=0.3, seed=3)
    st = np.zeros(2, capi.LOOP_DTYPE)
    st["prn"] = 7
    st["code_phase_fine"] = 4000.0
    st["if_freq_offset_hz"] = [1000.0, 1300.0]      # channel 0: 300 Hz off; channel 1: on the carrier
    st["found_freq_offset_hz"] = [1000, 1500]
    st["rng"] = [12345, 54321]
    # the FLL would pull channel 0 in; keep it where it is by handing it over again every 4 ms
    d = eng.malloc(st.nbytes)
    try:
        eng.h2d(d, st)
        moved_at = None
        for t in range(0, 600, 4):
            flags, _ = eng.track_loop(stream[t:t + 4], d, 2, t)
            cur = np.zeros_like(st)
            eng.d2h(cur, d)
            if flags[:, 0].max() & 16:
                moved_at = t
                break
            cur["if_freq_offset_hz"][0] = 1000.0     # (only the carrier is put back: the detector's counters run on)
            cur["fll_err"][0] = 0.0
            eng.h2d(d, cur)
        assert moved_at is not None and 4 * 85 <= moved_at <= 4 * 140, moved_at
        assert cur["reseed_count"][0] == 1 and cur["reseed_count"][1] == 0 and cur["rng"][0] != 12345 and cur["rng"][1] == 54321
        assert 750 <= cur["if_freq_offset_hz"][0] <= 1260 and abs(cur["if_freq_offset_hz"][0] - 1000.0) >= 190
        assert cur["pll_bad_state_master_cnt"][0] == 0
        st["prn"][1] = 211
        eng.h2d(d, st)
        with pytest.raises(capi.GpsxError):
            eng.track_loop(stream[:4], d, 2, 0)
    finally:
        eng.free(d)


_words_cache = {}


@pytest.mark.parametrize("k", [1, 20])
def test_word_layer_behind_the_device_loop_equals_the_host_mode(eng, k):
    """13 s of the 4-SV stream carrying LNAV subframes (two satellites with inverted data polarity), every channel served every
    millisecond.  Host mode: gps_tracking_process_batch all the way.  Device mode: the same until tick 600, then
    gpsx_track_loop in launches of k ms, the flag bytes through gps_tracking_words_batch, polarity changes back through
    gpsx_loop_set_polarity.  The word layer's whole state -- words, parity history, polarity, subframe images, subframe
    time stamps, decoded ephemerides -- must come out the same (k = 1: a polarity change reaches the device for the next
    millisecond exactly as in the host mode; k = 20: up to 19 ms later, which can only touch the bit being voted on while
    the word layer is still hunting for its next preamble)."""
    from stm32f4_sdr_gps_amd import capi, synth
    n_ms, t_hand = 13000, 600
    lib = eng.lib
    steps = sd.StepsLib(lib, False)
    lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
    lib.gps_tracking_process_batch.restype = None
    stream = synth.four_sv_with_lnav(n_ms, seed=7)
    presets = [(5, 900, 200), (14, 4000, 500), (20, -1000, 1124), (30, 2000, 1624)]

    def host_until(t_end):
        table = np.stack([sd.preset_channel(steps, *p) for p in presets])
        for t in range(t_end):
            steps.set_time(t)
            lib.gps_tracking_process_batch(table.ctypes.data, 4, stream[t].ctypes.data, t & 3)
        return table

    if "want" not in _words_cache:       # (the host mode's 13 s and its first 600 ms: the same for both launch lengths)
        _words_cache["want"], _words_cache["hand"] = host_until(n_ms), host_until(t_hand)
    want, table = _words_cache["want"], _words_cache["hand"].copy()
    assert (table[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0] == sd.TRK_RUN).all()
    st = _states_from_records(lib, table)
    d = eng.malloc(st.nbytes)
    try:
        eng.h2d(d, st)
        changed = np.zeros(4, np.int32)
        n_changes = 0
        for t in range(t_hand, n_ms, k):
            kk = min(k, n_ms - t)
            flags, _ = eng.track_loop(stream[t:t + kk], d, 4, t)
            n = lib.gps_tracking_words_batch(table.ctypes.data, 4, flags.ctypes.data, kk, t, changed.ctypes.data, 4)
            if n:
                vals = np.array([table[c, 212 + 13] for c in changed[:n]], np.uint8)
                assert lib.gpsx_loop_set_polarity(eng.h, d, changed.ctypes.data, vals.ctypes.data, n) == 0
                n_changes += n
        final = np.zeros_like(st)
        eng.d2h(final, d)
    finally:
        eng.free(d)
    assert n_changes >= 2                                     # PRN 14 and PRN 30 carry inverted data
    assert np.array_equal(final["inv_polarity_flag"], want[:, 212 + 13])
    nav_w, nav = want[:, 212:324], table[:, 212:324]
    # word layer: bytes 13 .. 111 of nav_data (polarity flags, word buffer and counters, parity history, time stamps, subframe
    # image); bytes 0 .. 12 are the bit synchroniser, which lives on the device
    assert np.array_equal(nav[:, 13:], nav_w[:, 13:]), np.argwhere(nav[:, 13:] != nav_w[:, 13:])[:6]
    assert np.array_equal(table[:, 344:664], want[:, 344:664])  # eph_data: the decoded subframes
    assert int(nav_w[:, 56:60].copy().view("<u4").max()) >= 1      # (parity-correct words were found: the layer did run)
    host_side = _states_from_records(lib, want)
    for f in ("period_sync_ok_flag", "right_period_cnt", "old_swap_time", "last_bit_pos_cnt", "last_bit_neg_cnt", "if_freq_accum"):
        assert np.array_equal(final[f], host_side[f]), f
    if k == 1:
        # the located bit edge (accurate_swap_time / _ok, rebuilt on the host from flag bits 5 / 6).  Only compared at k = 1: in
        # the host mode a polarity change takes effect in the MIDDLE of a 4 ms group, which reads as a sign flip at
        # position 2 and can locate an "edge" there (PRN 30 on this stream); with k > 1 the change reaches the device at a
        # launch boundary and that artefact does not happen
        assert np.array_equal(nav[:, 9:11], nav_w[:, 9:11])
        assert np.array_equal(final["accurate_swap_time"], host_side["accurate_swap_time"])


def test_device_loop_from_random_loop_states_equals_the_host_mode(eng):
    """Not only channels in lock: 6000 channels whose loop state is RANDOM -- code phases anywhere in [0, 16368] including the
    wrap region where Early / Prompt / Late are not neighbours and the DLL's wrap expressions fire, carriers anywhere in
    +-6 kHz, random loop-filter memories, false-lock counters and bit-synchroniser state (below the reseed threshold: the two
    modes draw from different generators there) -- on a stream with eight signals and on pure noise, 24 ms.  Every field of
    every channel must come out of the device loop as the host mode (the reference's loops on the CPU behind the same
    correlators) leaves it: floats within the stated tolerance and, where they are bit-identical, every integer too."""
    from stm32f4_sdr_gps_amd import capi, synth
    n, t0, t1 = 6000, 400, 424
    lib = eng.lib
    steps = sd.StepsLib(lib, False)
    lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
    lib.gps_tracking_process_batch.restype = None
    sats = [synth.Sat(i + 1, -2000.0 + 450.0 * i, (2000.0 * i + 37.0) % 16368, 0.25, 0.3 * i) for i in range(8)]
    for name, stream in (("signals", synth.make_if(t1, sats, noise_amp=1.0, seed=9)),
                         ("noise", synth.make_if(t1, [], noise_amp=1.0, seed=10))):
        rng = np.random.default_rng(77)
        table = np.zeros((n, sd.CH_SIZE), np.uint8)
        for c in range(n):
            table[c] = sd.preset_channel(steps, int(rng.integers(1, 33)), int(rng.integers(-12, 13)) * 500, int(rng.integers(0, 2046)))
        trk = table[:, 60:212]

        def put(off, arr, dt):
            a = np.ascontiguousarray(arr.astype(dt))
            trk[:, off:off + a.dtype.itemsize] = a.view(np.uint8).reshape(n, -1)

        fine = rng.uniform(0.0, 16368.0, n)
        fine[:400] = rng.choice([0.0, 0.4, 7.9, 8.0, 16360.2, 16367.9, 16368.0], 400)       # the edges
        put(80, fine, "<f4")
        put(4, rng.uniform(-6000.0, 6000.0, n), "<f4")
        put(8, rng.integers(0, 2**32, n, dtype=np.uint64), "<u4")
        put(92, rng.uniform(-1, 1, n), "<f4")             # dll_code_err
        put(96, rng.uniform(-1, 1, n), "<f4")             # pll_code_err
        put(100, rng.integers(-3000, 3000, n), "<i2")     # fll_old_i / q
        put(102, rng.integers(-3000, 3000, n), "<i2")
        put(104, rng.uniform(-1.5, 1.5, n), "<f4")        # fll_err
        for k in range(4):
            put(108 + 2 * k, rng.integers(-3000, 3000, n), "<i2")
        put(116, rng.integers(0, 11, n), "u1")            # pll_bad_state_cnt
        put(118, rng.integers(0, 70, n), "<u2")           # pll_bad_state_master_cnt: cannot reach 81 within 6 groups
        put(120, rng.integers(0, 100000, n), "<u4")
        put(124, rng.integers(0, 100000, n), "<u4")
        put(128, rng.integers(0, 202, n), "<u2")          # snr_summ_cnt: some cross kSnrLength inside the run
        put(140, rng.integers(0, 300, n), "<u2")          # code_filt_cnt
        put(144, rng.uniform(0, 1e6, n), "<f4")           # code_phase_fine_filt
        put(76, np.full(n, t0 - 1), "<u4")                # prev_track_timestamp: served last millisecond
        put(148, np.full(n, sd.TRK_RUN), "<i4")
        nav = table[:, 212:324]
        nav[:, 0] = rng.integers(0, 2, n)                 # period_sync_ok_flag
        nav[:, 1] = rng.integers(0, 11, n)                # right_period_cnt
        nav[:, 4:8] = np.ascontiguousarray(rng.integers(t0 - 40, t0, n).astype("<u4")).view(np.uint8).reshape(n, 4)   # old_swap_time
        nav[:, 8] = rng.integers(0, 20, n)                # old_reminder
        nav[:, 11] = rng.integers(0, 20, n)               # votes
        nav[:, 12] = rng.integers(0, 20, n)
        nav[:, 13] = rng.integers(0, 2, n)                # inv_polarity_flag
        st0 = _states_from_records(lib, table)
        _, _, final = _run(eng, stream, st0, t0, t1, 8, want_trace=False)
        for t in range(t0, t1):
            steps.set_time(t)
            lib.gps_tracking_process_batch(table.ctypes.data, n, stream[t].ctypes.data, t & 3)
        host = _states_from_records(lib, table)
        d_fine = np.abs(final["code_phase_fine"] - host["code_phase_fine"])
        d_freq = np.abs(final["if_freq_offset_hz"] - host["if_freq_offset_hz"])
        same = (d_fine == 0) & (d_freq == 0)
        print(name, "channels", n, "bit-identical floats", int(same.sum()), "max |d fine|", d_fine.max(), "max |d freq|", d_freq.max())
        assert d_fine.max() <= TOL_FINE and d_freq.max() <= TOL_HZ, name
        assert same.mean() > 0.98, name
        # the loop-filter memories that hold an arctangent (units of pi / radians): SURVEY.md 8(c)'s 1e-6 -- the device's
        # arctangents are the correctly rounded ones, glibc 2.35's float versions are good to an ulp: on random inputs the two
        # differ in the last bit now and then (on the reference's lock-in traces they never did)
        for f in ("pll_code_err", "fll_err", "dll_code_err"):
            assert np.abs(final[f] - host[f]).max() <= 1e-6, (name, f)
        assert np.allclose(final["code_phase_fine_filt"], host["code_phase_fine_filt"], rtol=1e-6), name
        assert np.allclose(final["snr_value"], host["snr_value"], rtol=1e-5, atol=1e-5), name
        for f in final.dtype.names:
            if f in ("rng", "reseed_count", "slot_start_ticks", "slot_ip", "slot_bits", "found_freq_offset_hz", "snr_value",
                     "snr_i_latch", "snr_q_latch",
                     "code_phase_fine", "if_freq_offset_hz", "pll_code_err", "fll_err", "dll_code_err", "code_phase_fine_filt"):
                continue       # (not part of gps_ch_t / floats, compared above)
            assert np.array_equal(final[f][same], host[f][same]), (name, f, np.flatnonzero(final[f][same] != host[f][same])[:5])
        assert (final["reseed_count"] == 0).all()


_LIBC_DRAWS_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
k = int(sys.argv[2])
import steps_driver as sd
from golden_util import fnv1a32, load
from stm32f4_sdr_gps_amd import capi, synth
g = load("f7_steps_config5_64ch.npz")
n_ms, t_hand = int(g["n_ms"]), 200
sats, chans, seed = sd.config5_64ch_scenario()
stream = synth.make_if(n_ms, sats, noise_amp=1.0, seed=seed)
assert fnv1a32(stream[::97]) == int(g["stream_fnv"])
lib = capi.load_library()
steps = sd.StepsLib(lib, False)
lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
lib.gps_tracking_process_batch.restype = None
lib.gps_tracking_words_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_int]
lib.gpsx_loop_state_from_channel.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
lib.gpsx_loop_state_to_channel.argtypes = [C.c_void_p, C.c_void_p]
# every kernel of the path once on another context, THEN srand(1): the ROCm runtime draws from rand() when code objects load
warm = capi.Engine(0)
jobs = np.zeros(1, capi.JOB_DTYPE); jobs[0] = (0, 1, 5, capi.IF_HZ + 900.0, 0, 0, 2046)
warm.acq_jobs(stream[:1], jobs)
st = np.zeros(64, capi.TRK_DTYPE); st["prn"] = 1 + np.arange(64) % 32
warm.track_epl(stream[0], st); warm.rewind(st, np.full(64, 3, np.uint8))
wl = np.zeros(4, capi.LOOP_DTYPE); wl["prn"] = 1; wl["rng"] = 1
wd = warm.malloc(wl.nbytes); warm.h2d(wd, wl)
warm.set_loop_draws(capi.DRAWS_LIBC); warm.track_loop(stream[:2], wd, 4, 0)
warm.free(wd); warm.close()
lib.gps_fill_summ_table()
C.CDLL("libc.so.6").srand(1)
table = np.stack([sd.preset_channel(steps, *c) for c in chans])
for t in range(t_hand):
    steps.set_time(t)
    lib.gps_tracking_process_batch(table.ctypes.data, len(chans), stream[t].ctypes.data, t & 3)
    assert np.array_equal(sd.snapshot_crcs(table), g["crcs"][t]), t
state = table[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0]
idx = np.flatnonzero(state == sd.TRK_RUN)
assert len(idx) == 62
sub = np.ascontiguousarray(table[idx])
n = len(idx)
eng = capi.Engine(0)
eng.set_loop_draws(capi.DRAWS_LIBC)
st = np.zeros(n, capi.LOOP_DTYPE)
for i in range(n):
    lib.gpsx_loop_state_from_channel(sub[i].ctypes.data, i + 1, st[i:i + 1].ctypes.data)
d = eng.malloc(st.nbytes); eng.h2d(d, st)
jumps = []
checked = 0
t = t_hand
while t < n_ms:
    kk = min(k, n_ms - t)
    flags, _ = eng.track_loop(stream[t:t + kk], d, n, t)
    lib.gps_tracking_words_batch(sub.ctypes.data, n, flags.ctypes.data, kk, t, None, 0)
    eng.d2h(st, d)
    for i in range(n):
        lib.gpsx_loop_state_to_channel(st[i:i + 1].ctypes.data, sub[i].ctypes.data)
    for ms, c in np.argwhere(flags & 16):
        jumps.append((t + int(ms), int(idx[c])))
    t += kk
    bad = np.flatnonzero(sd.snapshot_crcs(sub) != g["crcs"][t - 1][idx])
    if len(bad):
        print("MISMATCH after ms", t - 1, "channels", idx[bad][:8].tolist(), "jumps so far", jumps[-4:])
        sys.exit(1)
    checked += 1
assert np.array_equal(sd.snapshot(sub), g["final"][idx])
assert sorted(jumps) == sorted(map(tuple, g["reseeds"].tolist())), jumps
assert int(st["reseed_count"].sum()) == len(g["reseeds"])
print("RESULT", checked, len(jumps), len(set(c for _, c in jumps)))
eng.free(d); eng.close()
"""


@pytest.mark.parametrize("k", [1, 20, 64])
def test_device_loop_with_the_references_own_false_lock_draws(k):
    """GPSX_DRAWS_LIBC against the reference's 64-channel trace (tests/golden/f7_steps_config5_64ch.npz: 1500 ms, 21 false-lock
    jumps on 14 channels, three of them in the same millisecond, drawn from libc's rand() after srand(1)): the library's host
    mode takes the channels through pre-tracking; at tick 200 the 62 tracking channels move into the device loop (launches of
    k ms).  A channel whose detector fires reports and stands still, the host draws in (millisecond, channel) order, the
    reported channels are replayed from the launch's input state with their candidates -- and every channel's 226 state bytes
    carry the reference's CRC after every launch (k = 1: after every millisecond), the jumps are the reference's 21 at the
    reference's milliseconds.  In a process of its own: rand() is process-global."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _LIBC_DRAWS_SCRIPT, root, str(k)], capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert r.returncode == 0 and line, (r.stdout[-1500:], r.stderr[-1500:])
    f = line[0].split()
    assert int(f[1]) == -(-1300 // k) and f[2:] == ["21", "14"]
