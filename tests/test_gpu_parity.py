"""GPU parity tests: the HIP engine, called through the C ABI (libgpsx.so via ctypes), against
  (a) the committed golden vectors produced by the reference's own C (tests/golden/), and
  (b) the CPU oracle (oracle/liboracle.so) on seeded inputs.
Everything here is integer/bit work: the bar is bit-exact equality.  Nothing reads /root/reference.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import oracle_threads
from golden_util import IF_HZ, fnv1a32, known_answers, load

ORC_THREADS = oracle_threads()

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from stm32f4_sdr_gps_amd import capi
    e = capi.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def stream():
    from stm32f4_sdr_gps_amd import synth
    return synth.default_four_sv(12, seed=7)


@pytest.fixture(scope="module")
def eng_wg():
    """An engine whose tracking steps run k_track_epl (one workgroup per channel, the reference's 16-bit-word indexing) at every
    channel count ($GPSX_TRACK_WAVE_FROM): the second implementation the default kernel, k_track_epl_wave, is compared with."""
    import os
    from stm32f4_sdr_gps_amd import capi
    old = os.environ.get("GPSX_TRACK_WAVE_FROM")
    os.environ["GPSX_TRACK_WAVE_FROM"] = str(1 << 30)
    try:
        e = capi.Engine(0, lab=True)
    finally:
        if old is None:
            del os.environ["GPSX_TRACK_WAVE_FROM"]
        else:
            os.environ["GPSX_TRACK_WAVE_FROM"] = old
    yield e
    e.close()


def _peak_tuple(p):
    return int(p["max_val"]), int(p["phase"]), int(p["sum"]), int(p["avr"])


# ---------------------------------------------------------------------------------------------------------------
# K1
def test_ca_codes_golden(eng):
    g = load("f1_ca_codes.npz")
    chips = eng.ca_codes(np.arange(1, 211))
    assert np.array_equal(chips[:32], g["chips_1_32"])
    for prn in range(1, 211):
        assert fnv1a32(chips[prn - 1]) == int(g["fnv_1_210"][prn - 1]), prn


def test_ca_codes_rejects_bad_prn(eng):
    from stm32f4_sdr_gps_amd.capi import GpsxError
    with pytest.raises(GpsxError):
        eng.ca_codes([0])
    with pytest.raises(GpsxError):
        eng.ca_codes([211])


# ---------------------------------------------------------------------------------------------------------------
# per-call primitives (the device work behind the reference-named functions)
def test_wipeoff_golden(eng):
    g = load("f2_wipeoff.npz")
    for k in range(3):
        for j, d in enumerate(g["doppler_hz"]):
            di, dq, _ = eng.wipeoff(g["blocks"][k], float(IF_HZ + int(d)))
            assert fnv1a32(di[:1022]) == int(g["fnv_i"][k, j]) and fnv1a32(dq[:1022]) == int(g["fnv_q"][k, j])
    di, dq, _ = eng.wipeoff(g["blocks"][0], float(IF_HZ + 900))
    assert np.array_equal(di, g["example_i"]) and np.array_equal(dq, g["example_q"])
    # stateful NCO over 20 ms + rewind
    from stm32f4_sdr_gps_amd.capi import TRK_DTYPE
    acc = int(g["track_acc0"])
    for ms in range(20):
        off = np.float32(g["track_offsets"][ms])
        di, dq, acc = eng.wipeoff(g["blocks"][ms % 3], float(np.float32(IF_HZ) + off), acc)
        assert acc == int(g["track_accs"][ms])
        assert [fnv1a32(di[:1022]), fnv1a32(dq[:1022])] == list(map(int, g["track_fnv"][ms]))
        st = np.zeros(1, TRK_DTYPE)
        st["prn"], st["if_freq_offset_hz"], st["if_freq_accum"] = 1, off, acc
        eng.rewind(st, [int(g["rewind_steps"][ms])])
        assert int(st["if_freq_accum"][0]) == int(g["rewind_out"][ms])


def test_wipeoff_leaves_last_16_samples_alone(eng):
    rng = np.random.default_rng(5)
    pre = (rng.integers(0, 65536, 1024).astype(np.uint16), rng.integers(0, 65536, 1024).astype(np.uint16))
    sig = rng.integers(0, 256, 2046, dtype=np.uint8)
    di, dq, _ = eng.wipeoff(sig, 4092000.0 + 1234.0, 0, prefill=pre)
    assert di[1022] == pre[0][1022] and dq[1022] == pre[1][1022] and di[1023] == pre[0][1023]


def test_nco_step_float_division_matches_host(eng, oracle):
    # (uint32)(freq / 0.003810972f) must round as IEEE binary32 on the device: compare accumulators after one block
    rng = np.random.default_rng(6)
    sig = np.zeros(2046, np.uint8)
    freqs = np.concatenate([np.float32(IF_HZ) + rng.uniform(-10000, 10000, 300).astype(np.float32),
                            rng.uniform(1e3, 1.6e7, 100).astype(np.float32)])
    for f in freqs:
        _, _, acc = eng.wipeoff(sig, float(f), 0)
        want = (oracle.nco_step(float(f)) * 32 * 511) & 0xFFFFFFFF
        assert acc == want, float(f)


def test_replica_golden(eng):
    g = load("f3_replica.npz")
    codes = eng.ca_codes(g["prns"])
    for a in range(3):
        for b in range(16):
            assert np.array_equal(eng.replica(codes[a], b), g["replica"][a, b])
    # the pad word is OR-ed, never cleared (gps_misc.c:284)
    out = eng.replica(codes[1], 5, pad_in=0xA500)
    assert out[1023] == (0xA500 | int(g["replica"][1, 5][1023]))


def test_corr_offsets_golden_planes(eng):
    g = load("f4_corr.npz")
    blk = g["stream"][int(g["block_index"])]
    offsets = np.arange(2047, dtype=np.uint16)
    for a, (prn, d) in enumerate(g["full_cases"]):
        di, dq, _ = eng.wipeoff(blk, float(IF_HZ + int(d)))
        chips = eng.ca_codes([int(prn)])[0]
        for b in range(8):
            rep = eng.replica(chips, b)
            ci, cq, c8 = eng.corr_offsets(rep, di, dq, offsets)
            assert np.array_equal(ci, g["cnt_i"][a, b]) and np.array_equal(cq, g["cnt_q"][a, b])
            assert np.array_equal(c8, g["corr8"][a, b])


def test_corr_offsets_arbitrary_buffers_vs_oracle(eng, oracle):
    # buffers that are NOT chip-shaped / wiped: the per-call interface must be exact for any contents
    rng = np.random.default_rng(7)
    offsets = np.arange(2047, dtype=np.uint16)
    for _ in range(3):
        rep = rng.integers(0, 65536, 1024).astype(np.uint16)
        di = rng.integers(0, 65536, 1024).astype(np.uint16)
        dq = rng.integers(0, 65536, 1024).astype(np.uint16)
        ci, cq, c8 = eng.corr_offsets(rep, di, dq, offsets)
        want = np.array([oracle.mult_and_summ(di, dq, rep, int(o)) for o in offsets])
        assert np.array_equal(ci, want[:, 0]) and np.array_equal(cq, want[:, 1])
        assert np.array_equal(c8, [oracle.mag8(int(a), int(b)) for a, b in want])


def test_mag8_sqrt_is_correctly_rounded_everywhere_it_matters(eng):
    """(int16) sqrtf(float(I*I) + float(Q*Q)) truncates a ROUNDED root: wherever the true root sits just below an integer
    a 1-ulp-approximate hardware sqrt can land on the wrong side.  Sweep every I with Q = 0 and Q = I, the pairs around
    every integer radius, and two million random pairs, against numpy's correctly rounded float32 sqrt."""
    def want(ci, cq):
        i = np.clip(ci.astype(np.int64) - 8184, 0, None)
        q = np.clip(cq.astype(np.int64) - 8184, 0, None)
        e = (i * i).astype(np.float32) + (q * q).astype(np.float32)
        return np.sqrt(e).astype(np.int16)
    rng = np.random.default_rng(12)
    a = np.arange(0, 16369, dtype=np.int64)
    sets = [(a, np.full_like(a, 8184)), (a, a), (a, 16368 - a)]
    # pairs (i, q) with i*i + q*q within +-2 of k*k for every radius k: the truncation boundaries
    ks = np.arange(1, 11574, dtype=np.int64)
    for frac in (0.0, 0.38, 0.6, 0.71, 0.92):
        i = np.floor(ks * frac).astype(np.int64)
        for dq in (-1, 0, 1):
            q = np.floor(np.sqrt(np.maximum(ks * ks - i * i, 0))).astype(np.int64) + dq
            ok = (i <= 8184) & (q >= 0) & (q <= 8184)
            sets.append((8184 + i[ok], 8184 + q[ok]))
    sets.append((rng.integers(8184, 16369, 2_000_000), rng.integers(8184, 16369, 2_000_000)))
    # the grid kernels' small-radius shortcut (whole wave below radius 1024: bare v_sqrt_f32 of e + 1/2): EVERY pair of
    # that domain, in an order that keeps waves inside it, plus the same pairs interleaved with large ones (mixed waves
    # must take the general path)
    ii, qq = np.meshgrid(np.arange(1024, dtype=np.int64), np.arange(1024, dtype=np.int64), indexing="ij")
    small = (ii * ii + qq * qq) < (1 << 20)
    si, sq = 8184 + ii[small], 8184 + qq[small]
    sets.append((si, sq))
    mixed_i, mixed_q = si.copy(), sq.copy()
    mixed_i[::7] = 16368 - (mixed_i[::7] - 8184) % 5000
    sets.append((mixed_i, mixed_q))
    sets.append((8184 - ii[small], sq))                       # negative I: clipped to zero
    for ci, cq in sets:
        got = eng.mag8(ci, cq)
        assert np.array_equal(got, want(ci, cq))


def test_corr_search_golden(eng):
    g = load("f4_corr.npz")
    blk = g["stream"][int(g["block_index"])]
    for prn, d, b, s0, s1, mx, av, ph in g["search"]:
        di, dq, _ = eng.wipeoff(blk, float(IF_HZ + int(d)))
        rep = eng.replica(eng.ca_codes([int(prn)])[0], int(b))
        assert eng.corr_search(rep, di, dq, int(s0), int(s1)) == (int(mx), int(av), int(ph))
    z = np.zeros(1024, np.uint16)
    assert eng.corr_search(z, z, z, 300, 400) == (0, 0, 0)   # nothing above zero: phase 0 whatever the window


def test_config1_reference_self_test_through_reference_named_calls(eng):
    """BASELINE.json configs[0] / SS/main.c:59-69, written the way the reference writes it."""
    import ctypes as C
    lib = eng.lib
    g = load("f6_config1.npz")
    ka = known_answers()["f6_config1"]
    ch = np.zeros(1688, np.uint8)          # gps_ch_t
    ch[664] = 1                            # .prn = SIM_PRN_CODE
    lib.gps_fill_summ_table()
    lib.gps_channell_prepare(ch.ctypes.data)
    assert np.array_equal(ch[665:665 + 1023], load("f1_ca_codes.npz")["chips_1_32"][0])
    tmp_prn, tmp_i, tmp_q = (np.zeros(1024, np.uint16) for _ in range(3))
    for k, noise in enumerate(g["noise_levels"]):
        signal = np.ascontiguousarray(g["blocks"][k])
        lib.gps_generate_prn_data2(ch.ctypes.data, tmp_prn.ctypes.data, 0)
        lib.gps_shift_to_zero_freq(signal.ctypes.data, tmp_i.ctypes.data, tmp_q.ctypes.data, C.c_float(IF_HZ + 2000))
        avr, phase = C.c_uint16(), C.c_uint16()
        mx = lib.correlation_search(tmp_prn.ctypes.data, tmp_i.ctypes.data, tmp_q.ctypes.data, 0, 2046,
                                    C.byref(avr), C.byref(phase))
        want = ka[str(int(noise))]
        assert (mx, avr.value, phase.value) == (want["max"], want["avr"], want["phase"])
        ri, rq = C.c_int16(), C.c_int16()
        lib.gps_correlation_iq(tmp_prn.ctypes.data, tmp_i.ctypes.data, tmp_q.ctypes.data, 100, C.byref(ri), C.byref(rq))
        assert (ri.value, rq.value) == (want["i_at_100"], want["q_at_100"])
        assert lib.gps_correlation8(tmp_prn.ctypes.data, tmp_i.ctypes.data, tmp_q.ctypes.data, 100) == want["max"]
    assert ka["0"]["max"] == 7904 and ka["0"]["phase"] == 100


def test_reference_named_tracking_calls(eng, oracle):
    import ctypes as C
    lib = eng.lib
    rng = np.random.default_rng(8)
    trk = np.zeros(152, np.uint8)          # gps_tracking_t
    off, acc = np.float32(-1234.5), 0xDEADBEEF
    trk[4:8] = np.frombuffer(off.tobytes(), np.uint8)
    trk[8:12] = np.frombuffer(np.uint32(acc).tobytes(), np.uint8)
    sig = rng.integers(0, 256, 2046, dtype=np.uint8)
    di, dq = np.zeros(1024, np.uint16), np.zeros(1024, np.uint16)
    lib.gps_shift_to_zero_freq_track(trk.ctypes.data, sig.ctypes.data, di.ctypes.data, dq.ctypes.data)
    oi, oq, oacc = oracle.wipeoff(sig, float(np.float32(IF_HZ) + off), acc)
    assert np.array_equal(di, oi) and np.array_equal(dq, oq)
    assert int(trk[8:12].view(np.uint32)[0]) == oacc
    lib.gps_rewind_if_phase(trk.ctypes.data, 13)
    assert int(trk[8:12].view(np.uint32)[0]) == oracle.rewind(float(off), oacc, 13)


# ---------------------------------------------------------------------------------------------------------------
# K2+K3+K4: the acquisition grid kernel
def test_acq_grid_raw_counts_and_energy_vs_golden_planes(eng):
    g = load("f4_corr.npz")
    blk = g["stream"][int(g["block_index"])][None, :]
    for a, (prn, d) in enumerate(g["full_cases"]):
        out = eng.acq_grid_debug(blk, [int(prn)], dopp_min_hz=int(d), n_dopp=1, want_cnt=True, want_energy=True)
        for b in range(8):
            assert np.array_equal(out["cnt"][0, 0, 0, b, :, 0], g["cnt_i"][a, b, :2046]), (prn, d, b)
            assert np.array_equal(out["cnt"][0, 0, 0, b, :, 1], g["cnt_q"][a, b, :2046]), (prn, d, b)
            assert np.array_equal(out["energy"][0, 0, 0, b], g["corr8"][a, b, :2046].astype(np.uint32))


def test_acq_grid_peaks_vs_golden_search_triplets(eng):
    g = load("f4_corr.npz")
    blk = g["stream"][int(g["block_index"])][None, :]
    for prn, d, b, s0, s1, mx, av, ph in g["search"]:
        peaks, _ = eng.acq_grid(blk, [int(prn)], dopp_min_hz=int(d), n_dopp=1, win=(int(s0), int(s1)))
        p = peaks[0, 0, 0, int(b)]
        assert (int(p["max_val"]), int(p["avr"]), int(p["phase"])) == (int(mx), int(av), int(ph))
    for prn, d, b, h, mx, av, ph in g["hashed"]:
        out = eng.acq_grid_debug(blk, [int(prn)], dopp_min_hz=int(d), n_dopp=1, want_energy=True)
        assert fnv1a32(out["energy"][0, 0, 0, int(b)].astype(np.int16)) == int(h)
        p = out["peaks"][0, 0, 0, int(b)]
        assert (int(p["max_val"]), int(p["avr"]), int(p["phase"])) == (int(mx), int(av), int(ph))


def test_acq_grid_full_cold_start_grid_vs_oracle(eng, oracle):
    """BASELINE.json configs[2] at full size: 32 PRN x 21 Doppler x 16368 phases on one block, every triplet."""
    from stm32f4_sdr_gps_amd import synth
    blk = synth.cold_start_block(1, seed=11)
    prns = np.arange(1, 33, dtype=np.uint8)
    peaks, keys = eng.acq_grid(blk, prns, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
    assert eng.lib.gpsx_last_kernel(eng.h) == b"k_acq_mx<5>"       # a lone capture: 21 clusters as 42 workgroups (split form)
    want = oracle.acq_grid(blk, 1, prns, -5000, 500, 21, 8, n_threads=ORC_THREADS, live=True)
    for f in ("max_val", "phase", "sum", "avr"):
        assert np.array_equal(peaks[0][f], want[f]), f
    # packed keys: energy and lowest fine phase of the best bit shift
    fine = 8 * want["phase"].astype(np.int64) + np.arange(8)[None, None, :]
    k = (want["max_val"].astype(np.int64) << 14) | (16383 - fine)
    assert np.array_equal(keys[0], k.max(axis=2))
    # the six satellites of the synthetic block dominate their (PRN, Doppler) neighbourhood
    assert int(peaks["max_val"].max()) > 1500


def test_acq_grid_reference_native_grid_29_bins_byte_phases(eng, oracle, stream):
    """The reference's own search grid (PM/GPS/acquisition.c:280-312, PM/config.h:41-44): 2046 byte phases x 29 Doppler bins
    (+-7 kHz at 500 Hz), all 32 PRNs, two captures -- k_acq_mx<4>: sample offsets 0 and 8 of the fine grid, each started
    directly from its own block sums (four matrix passes).  Then windows that cut through the even / odd offsets of one chip
    offset, and a PRN list that is not a multiple of the cluster."""
    from stm32f4_sdr_gps_amd.capi import PHASES_BYTE
    prns = np.arange(1, 33, dtype=np.uint8)
    peaks, _ = eng.acq_grid(stream[4:6], prns, n_search=2, dopp_min_hz=-7000, dopp_step_hz=500, n_dopp=29, phase_mode=PHASES_BYTE)
    assert eng.lib.gpsx_last_kernel(eng.h) == b"k_acq_mx<4>"
    for s_ in range(2):
        want = oracle.acq_grid(stream[4 + s_:5 + s_], 1, prns, -7000, 500, 29, 1, n_threads=ORC_THREADS)
        for f in ("max_val", "phase", "sum", "avr"):
            assert np.array_equal(peaks[s_][f], want[f]), (s_, f)
    prns5 = np.array([5, 14, 20, 30, 1, 33, 210], np.uint8)
    for win in ((0, 1), (1, 2), (1, 2046), (0, 2045), (777, 778), (778, 779), (100, 1901), (2045, 2046), (3, 3)):
        peaks, _ = eng.acq_grid(stream[2:3], prns5, dopp_min_hz=-1000, dopp_step_hz=1000, n_dopp=3, phase_mode=PHASES_BYTE, win=win)
        assert eng.lib.gpsx_last_kernel(eng.h) == b"k_acq_mx<4>"
        for p, prn in enumerate(prns5):
            for d in range(3):
                pk, _, _ = oracle.search_job(stream[2:3], 1, oracle.ca_code(int(prn)), float(IF_HZ - 1000 + 1000 * d), 0, win[0], win[1])
                assert _peak_tuple(peaks[0, p, d, 0]) == (pk["max_val"], pk["phase"], pk["sum"], pk["avr"]), (win, p, d)


def test_byte_phase_grid_persistent_workgroups_two_prn_sets_and_two_bit_if(oracle, stream, monkeypatch):
    """k_acq_mx<4> runs one persistent workgroup per CU that walks the launch's clusters: more clusters than CUs (40 PRNs = two
    32-slot sets x 29 Doppler bins x 6 captures = 348: every workgroup walks at least one cluster of each set and reloads
    the set's tables in between), on 2-bit sign/magnitude captures -- against the direct 4-bit-dot-product kernel
    (GPSX_ACQ_ALGO=dot8: another algorithm, itself checked against the oracle) on every triplet and key, and against the
    oracle on a sample."""
    from stm32f4_sdr_gps_amd import capi, synth
    sats = [synth.Sat(5, 912.5, 1600.0, 0.6, 0.3), synth.Sat(30, 2018.0, 13000.0, 0.6, 4.0), synth.Sat(20, -3011.0, 7000.0, 0.5, 1.0)]
    two = synth.make_if(6, sats, seed=23, two_bit=True)
    one = synth.make_if(6, sats, seed=23)
    prns = np.concatenate([np.arange(1, 33), [33, 40, 61, 100, 120, 150, 200, 210]]).astype(np.uint8)
    kw = dict(n_search=6, dopp_min_hz=-7000, dopp_step_hz=500, n_dopp=29, phase_mode=capi.PHASES_BYTE)
    e = capi.Engine(0)
    monkeypatch.setenv("GPSX_ACQ_ALGO", "dot8")
    ref = capi.Engine(0, lab=True)
    monkeypatch.delenv("GPSX_ACQ_ALGO")
    try:
        e.set_if_format(capi.IF_2BIT_SM)
        pk, keys = e.acq_grid(two, prns, **kw)
        assert e.lib.gpsx_last_kernel(e.h) == b"k_acq_mx<4>"
        want_pk, want_keys = ref.acq_grid(one, prns, **kw)
        assert ref.lib.gpsx_last_kernel(ref.h).startswith(b"k_acq<8,false,dot8>")
        assert np.array_equal(pk, want_pk) and np.array_equal(keys, want_keys)
        for s_, p, d in ((0, 4, 15), (5, 29, 18), (3, 33, 8), (2, 39, 0), (1, 0, 28)):
            o, _, _ = oracle.search_job(one[s_:s_ + 1], 1, oracle.ca_code(int(prns[p])), float(IF_HZ - 7000 + 500 * d), 0)
            assert _peak_tuple(pk[s_, p, d, 0]) == (o["max_val"], o["phase"], o["sum"], o["avr"]), (s_, p, d)
    finally:
        e.close()
        ref.close()


def test_byte_phase_grid_pipeline_random_descriptors(monkeypatch):
    """Seeded sweep of byte-phase grids big enough for the pipeline to run several clusters per workgroup: PRN lists of 33..210
    codes (two to seven 32-slot sets: the grid of persistent workgroups is then a multiple of the set count), arbitrary Doppler
    grids, windows, shards, strides between the searches' blocks, 1-bit and 2-bit captures -- every triplet and key against the
    direct 4-bit-dot-product kernel."""
    from stm32f4_sdr_gps_amd import capi, synth
    rng = np.random.default_rng(4242)
    e = capi.Engine(0)
    monkeypatch.setenv("GPSX_ACQ_ALGO", "dot8")
    ref = capi.Engine(0, lab=True)
    monkeypatch.delenv("GPSX_ACQ_ALGO")
    try:
        for trial in range(8):
            n_prn = int(rng.integers(33, 211))
            prns = rng.permutation(np.arange(1, 211))[:n_prn].astype(np.uint8)
            n_dopp = int(rng.integers(3, 12))
            n_search = int(rng.integers(12, 40))
            stride = int(rng.integers(0, 2))
            two = bool(trial & 1)
            n_blocks = (n_search - 1) * stride + 1
            blocks = synth.cold_start_block(n_blocks, seed=300 + trial, amp_scale=0.3, two_bit=two)
            a, b = sorted(int(x) for x in rng.integers(0, 2047, 2))
            world = int(rng.integers(1, 4))
            kw = dict(n_search=n_search, search_stride_blocks=stride, dopp_min_hz=int(rng.integers(-7000, 3000)),
                      dopp_step_hz=int(rng.integers(1, 900)), n_dopp=n_dopp, phase_mode=capi.PHASES_BYTE, win=(a, b),
                      shard=(int(rng.integers(world)), world))
            for eng_ in (e, ref):
                eng_.set_if_format(capi.IF_2BIT_SM if two else capi.IF_1BIT)
            pk, keys = e.acq_grid(blocks, prns, **kw)
            assert e.lib.gpsx_last_kernel(e.h) == b"k_acq_mx<4>", trial
            want_pk, want_keys = ref.acq_grid(blocks, prns, **kw)
            assert np.array_equal(pk, want_pk) and np.array_equal(keys, want_keys), (trial, kw, n_prn, two)
    finally:
        e.close()
        ref.close()


@pytest.mark.parametrize("n_search,win,shard", [(64, None, None), (37, (5, 2001), (1, 3)), (9, (1, 2), None), (18, None, None),
                                                 (256, None, None)])
def test_byte_phase_grid_pipeline_depths(oracle, monkeypatch, n_search, win, shard):
    """k_acq_mx<4> is one software pipeline per persistent workgroup: what cluster c + 1 and c + 2 need (block, wipe-off, block
    sums' codes, vectors) is made behind cluster c's stage barriers, results are written two stages late.  Launches whose
    workgroups walk 1, 2 (18 captures x 29 bins = 522 clusters on 256 CUs: two or three), 4-5 and 7-8 clusters -- the fill, the
    short paths with missing pieces and the steady state, and the bench's own launch (256 captures: 29 clusters per
    workgroup) -- with and without a search window, every triplet and key against
    the direct 4-bit-dot-product kernel (GPSX_ACQ_ALGO=dot8, itself checked against the oracle), a sample against the oracle."""
    from stm32f4_sdr_gps_amd import capi, synth
    blocks = synth.cold_start_block(n_search, seed=31 + n_search, amp_scale=0.3)
    prns = np.arange(1, 33, dtype=np.uint8)
    kw = dict(n_search=n_search, dopp_min_hz=-7000, dopp_step_hz=500, n_dopp=29, phase_mode=capi.PHASES_BYTE)
    if win:
        kw["win"] = win
    if shard:   # (the middle third of the sharding units: the run starts and ends inside clusters)
        kw["shard"] = shard
    e = capi.Engine(0)
    monkeypatch.setenv("GPSX_ACQ_ALGO", "dot8")
    ref = capi.Engine(0, lab=True)
    monkeypatch.delenv("GPSX_ACQ_ALGO")
    try:
        pk, keys = e.acq_grid(blocks, prns, **kw)
        assert e.lib.gpsx_last_kernel(e.h) == b"k_acq_mx<4>"
        want_pk, want_keys = ref.acq_grid(blocks, prns, **kw)
        assert ref.lib.gpsx_last_kernel(ref.h).startswith(b"k_acq<8,false,dot8>")
        assert np.array_equal(pk, want_pk) and np.array_equal(keys, want_keys)
        w0, w1 = win if win else (0, 2046)
        rng = np.random.default_rng(n_search)
        owned = np.argwhere(keys != 0) if shard else None
        for _ in range(6):
            s_, p, d = (int(v) for v in owned[rng.integers(len(owned))]) if shard else \
                (int(rng.integers(n_search)), int(rng.integers(32)), int(rng.integers(29)))
            o, _, _ = oracle.search_job(blocks[s_:s_ + 1], 1, oracle.ca_code(int(prns[p])), float(IF_HZ - 7000 + 500 * d), 0, w0, w1)
            assert _peak_tuple(pk[s_, p, d, 0]) == (o["max_val"], o["phase"], o["sum"], o["avr"]), (s_, p, d)
    finally:
        e.close()
        ref.close()


def test_acq_grid_non_coherent_10ms_and_per_ms_triplets(eng, oracle, stream):
    """BASELINE.json configs[3] semantics: energy = sum over 10 blocks of the per-block magnitude."""
    prns = np.array([5, 14, 20, 30, 7, 9, 11, 13, 15], np.uint8)   # 9 PRNs: exercises a partial PRN group
    out = eng.acq_grid_debug(stream[:10], prns, n_ms=10, dopp_min_hz=-1000, dopp_step_hz=500, n_dopp=5,
                             want_per_ms=True, want_energy=True)
    want = oracle.acq_grid(stream[:10], 10, prns, -1000, 500, 5, 8, n_threads=ORC_THREADS)
    for f in ("max_val", "phase", "sum", "avr"):
        assert np.array_equal(out["peaks"][0][f], want[f]), f
    for p, d, b in [(0, 3, 0), (1, 4, 5), (8, 0, 7), (3, 2, 2)]:
        pk, energy, per_ms = oracle.search_job(stream[:10], 10, oracle.ca_code(int(prns[p])),
                                               float(IF_HZ - 1000 + 500 * d), b, want_energy=True, want_per_ms=True)
        assert np.array_equal(out["energy"][0, p, d, b], energy)
        for f in ("max_val", "phase", "sum", "avr"):
            assert np.array_equal(out["per_ms"][0, p, d, b][f], per_ms[f]), f
    # PRN 5 at +900 Hz bin (index 4 is +1000, index 3 is +500): the 10 ms sum must find delay 1600 samples = 200 bytes
    assert int(out["peaks"][0, 0, 4, 0]["phase"]) == 200 or int(out["peaks"][0, 0, 3, 0]["phase"]) == 200


def test_acq_grid_multiple_searches_and_stride(eng, oracle, stream):
    prns = np.array([5, 30], np.uint8)
    peaks, _ = eng.acq_grid(stream[:9], prns, n_search=3, n_ms=2, search_stride_blocks=3, dopp_min_hz=500,
                            dopp_step_hz=500, n_dopp=4)
    for s in range(3):
        want = oracle.acq_grid(stream[3 * s:3 * s + 2], 2, prns, 500, 500, 4, 8, n_threads=4)
        for f in ("max_val", "phase", "sum", "avr"):
            assert np.array_equal(peaks[s][f], want[f]), (s, f)


def test_acq_grid_windows(eng, oracle, stream):
    prns = np.array([5, 14], np.uint8)
    for win in [(0, 500), (250, 751), (2045, 2046), (7, 7), (1, 2)]:
        peaks, _ = eng.acq_grid(stream[6:7], prns, dopp_min_hz=1000, n_dopp=1, win=win)
        for p, prn in enumerate(prns):
            for b in range(8):
                pk, _, _ = oracle.search_job(stream[6:7], 1, oracle.ca_code(int(prn)), float(IF_HZ + 1000), b,
                                             win[0], win[1])
                assert _peak_tuple(peaks[0, p, 0, b]) == (pk["max_val"], pk["phase"], pk["sum"], pk["avr"])


def test_acq_grid_sharding_union_equals_single(eng, stream):
    """Multi-GPU contract: shards compute disjoint (search, PRN-group, Doppler) units; max over shards of the key
    tables equals the unsharded table; peaks of foreign units are zero."""
    prns = np.arange(1, 21, dtype=np.uint8)      # 20 PRNs -> 3 groups
    kw = dict(n_search=2, n_ms=1, search_stride_blocks=1, dopp_min_hz=-1000, dopp_step_hz=500, n_dopp=5)
    full_peaks, full_keys = eng.acq_grid(stream[:2], prns, **kw)
    for world in (2, 3, 8):
        acc = np.zeros_like(full_keys)
        owned = np.zeros(full_keys.shape, np.int32)
        for r in range(world):
            pk, ks = eng.acq_grid(stream[:2], prns, shard=(r, world), **kw)
            acc = np.maximum(acc, ks)
            mine = ks != 0
            owned += mine
            assert np.array_equal(pk["max_val"][mine], full_peaks["max_val"][mine])
            assert not pk["sum"][~mine].any()
        assert np.array_equal(acc, full_keys)
        assert (owned == 1).all()


def test_acq_jobs_per_channel_hints_and_windows(eng, oracle, stream):
    """acquisition_process() shape: the reference's 4-channel table, each with its own Doppler hint and window."""
    from stm32f4_sdr_gps_amd.capi import JOB_DTYPE
    jobs = np.zeros(7, JOB_DTYPE)
    rows = [(0, 1, 5, IF_HZ + 900, 0, 0, 2046), (0, 1, 14, IF_HZ + 4000, 0, 0, 2046), (1, 1, 20, IF_HZ - 1000, 0, 875, 1375),
            (2, 1, 30, IF_HZ + 2000, 0, 1595, 1655), (3, 1, 5, 4092912.5, 3, 185, 215), (0, 1, 120, IF_HZ + 33.25, 7, 0, 2046),
            (4, 1, 32, 4089000.75, 1, 2000, 2046)]
    for i, r in enumerate(rows):
        jobs[i] = r
    peaks, energy = eng.acq_jobs(stream, jobs, want_energy=True)
    for i, (blk, n_ms, prn, f, b, s0, s1) in enumerate(rows):
        pk, en, _ = oracle.search_job(stream[blk:blk + n_ms], n_ms, oracle.ca_code(prn), float(np.float32(f)), b, s0, s1,
                                      want_energy=True)
        assert _peak_tuple(peaks[i]) == (pk["max_val"], pk["phase"], pk["sum"], pk["avr"]), i
        assert np.array_equal(energy[i], en), i


def test_acq_grid_saturated_block_sums_clean_carriers(eng, oracle):
    """Clean synthetic inputs make whole 16-sample windows all ones after wipe-off (block sum 16): the 4-bit dot-product
    path must restore those exactly.  Inputs: the reference simulator's noise-free block (PRN 1, IF + 2000 Hz), an
    all-ones / all-zeros capture, a bare Fs/4 carrier, and a carrier times PRN 7."""
    g = load("f6_config1.npz")
    rng = np.random.default_rng(3)
    chips7 = oracle.ca_code(7)
    carrier = np.tile(np.array([0x99], np.uint8), 2046)
    code_bits = np.repeat(chips7, 16)[:16368]
    prn_on_carrier = np.packbits(np.unpackbits(carrier, bitorder="little")[:16368] ^ code_bits, bitorder="little")
    blocks = np.stack([g["blocks"][0], np.full(2046, 0xFF, np.uint8), np.zeros(2046, np.uint8), carrier, prn_on_carrier,
                       np.roll(prn_on_carrier, 137)])
    prns = np.array([1, 7, 16, 22], np.uint8)
    for k in range(len(blocks)):
        for dopp in (2000, 0, -500):
            out = eng.acq_grid_debug(blocks[k:k + 1], prns, dopp_min_hz=dopp, n_dopp=1, want_cnt=True)
            want = oracle.acq_grid(blocks[k:k + 1], 1, prns, dopp, 500, 1, 8, n_threads=4)
            for f in ("max_val", "phase", "sum", "avr"):
                assert np.array_equal(out["peaks"][0][f], want[f]), (k, dopp, f)
            # raw popcounts of a few (PRN, shift) planes against the oracle primitive
            di, dq, _ = oracle.wipeoff(blocks[k], float(IF_HZ + dopp))
            for p, b in [(0, 0), (1, 3), (3, 7)]:
                rep = oracle.replica(oracle.ca_code(int(prns[p])), b)
                ref_cnt = np.array([oracle.mult_and_summ(di, dq, rep, o) for o in range(0, 2046, 7)])
                assert np.array_equal(out["cnt"][0, p, 0, b, ::7, :], ref_cnt), (k, dopp, p, b)
    # config 1 through the grid kernel: same 7904 / 65 / 100 as the reference's self-test
    peaks, _ = eng.acq_grid(g["blocks"][0:1], [1], dopp_min_hz=2000, n_dopp=1)
    assert _peak_tuple(peaks[0, 0, 0, 0])[:2] == (7904, 100) and int(peaks[0, 0, 0, 0]["avr"]) == 65


def test_acq_rejects_bad_arguments(eng, stream):
    from stm32f4_sdr_gps_amd.capi import GpsxError, JOB_DTYPE
    with pytest.raises(GpsxError):
        eng.acq_grid(stream[:1], [0])
    with pytest.raises(GpsxError):
        eng.acq_grid(stream[:1], [1], n_ms=2)                 # reads past the supplied blocks
    with pytest.raises(GpsxError):
        eng.acq_grid(stream[:1], [1], win=(5, 3000))
    with pytest.raises(GpsxError):
        eng.acq_grid(stream[:1], [1], phase_mode=1234)
    jobs = np.zeros(1, JOB_DTYPE)
    jobs[0] = (0, 1, 5, float(IF_HZ), 9, 0, 2046)            # offset_bits > 7
    with pytest.raises(GpsxError):
        eng.acq_jobs(stream[:1], jobs)


# ---------------------------------------------------------------------------------------------------------------
# K2+K3+K5: tracking correlators
def test_track_epl_256_channels_vs_oracle(eng, oracle, stream):
    """BASELINE.json configs[4] shape: 256 concurrent channels on one shared IF stream, several milliseconds."""
    from stm32f4_sdr_gps_amd.capi import TRK_DTYPE
    rng = np.random.default_rng(9)
    n = 256
    st = np.zeros(n, TRK_DTYPE)
    st["prn"] = (np.arange(n) % 32) + 1
    st["code_phase_fine"] = rng.uniform(0, 16368, n).astype(np.float32)
    st["code_phase_fine"][:8] = [0.0, 0.99, 7.5, 8.0, 15.9, 16367.9, 16368.0, 16360.0]   # E/L wrap cases
    st["if_freq_offset_hz"] = (-5000 + 39 * np.arange(n)).astype(np.float32) + rng.uniform(-1, 1, n).astype(np.float32)
    st["if_freq_accum"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    ref_acc = st["if_freq_accum"].copy()
    for ms in range(4):
        fine_before = st["code_phase_fine"].copy()
        iq = eng.track_epl(stream[ms], st)
        for c in range(n):
            want, ref_acc[c] = oracle.track_epl(stream[ms], oracle.ca_code(int(st["prn"][c])), float(fine_before[c]),
                                                float(st["if_freq_offset_hz"][c]), int(ref_acc[c]))
            assert np.array_equal(iq[c], want), (ms, c)
        assert np.array_equal(st["if_freq_accum"], ref_acc)
        st["code_phase_fine"] = np.mod(st["code_phase_fine"] + rng.uniform(-3, 3, n).astype(np.float32), 16368).astype(np.float32)


def test_track_epl_wave_form_for_many_channels(eng, eng_wg, oracle, stream):
    """The step runs k_track_epl_wave (one wave per channel, Early / Prompt / Late out of one shared replica window per four
    data words) at every channel count; k_track_epl (one workgroup per channel) is the library's second implementation.
    2304 channels against the oracle on a sample, and -- including code phases the reference itself would mishandle
    (negative, beyond 16368: both kernels keep every access in range the same way) -- the two kernels against each other
    on every channel; and the second implementation against the oracle as well."""
    from stm32f4_sdr_gps_amd.capi import TRK_DTYPE
    rng = np.random.default_rng(10)
    n = 256
    st = np.zeros(n, TRK_DTYPE)
    st["prn"] = (np.arange(n) % 32) + 1
    st["code_phase_fine"] = rng.uniform(0, 16368, n).astype(np.float32)
    st["code_phase_fine"][:14] = [0.0, 0.99, 7.5, 8.0, 15.9, 16367.9, 16368.0, 16360.0, 16352.0, 16359.99, -1.0, -9.0, -20000.0,
                                  20000.0]
    st["if_freq_offset_hz"] = (-5000 + 39 * np.arange(n)).astype(np.float32)
    st["if_freq_accum"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    small = st.copy()
    iq_small = eng_wg.track_epl(stream[2], small)              # 256 channels: workgroup-per-channel kernel
    big = np.tile(st, 9)                                       # 2304 channels: wave-per-channel kernel
    iq_big = eng.track_epl(stream[2], big)
    for rep in range(9):
        assert np.array_equal(iq_big[rep * n:(rep + 1) * n], iq_small), rep
        assert np.array_equal(big[rep * n:(rep + 1) * n], small), rep
    for c in list(range(10)) + list(range(14, n, 7)):          # valid phases: against the oracle
        want, acc = oracle.track_epl(stream[2], oracle.ca_code(int(st["prn"][c])), float(st["code_phase_fine"][c]),
                                     float(st["if_freq_offset_hz"][c]), int(st["if_freq_accum"][c]))
        assert np.array_equal(iq_big[n + c], want) and int(big["if_freq_accum"][n + c]) == acc, c
        assert np.array_equal(iq_small[c], want) and int(small["if_freq_accum"][c]) == acc, c


def test_track_epl_graph_cache_alternating_shapes_and_formats(oracle, stream):
    """The per-millisecond step runs as a captured graph per (channel count, sample format), a few shapes cached: calls
    alternating between more shapes than the cache holds, and between 1-bit and 2-bit blocks, must keep giving the
    oracle's accumulators (and must not leak state from one shape's staging buffers into another's)."""
    from stm32f4_sdr_gps_amd import capi, synth
    e = capi.Engine(0)
    try:
        sats = [synth.Sat(5, 912.5, 1600.0, 0.6, 0.3), synth.Sat(30, 2018.0, 13000.0, 0.6, 4.0)]
        one = synth.make_if(2, sats, seed=21)
        two = synth.make_if(2, sats, seed=21, two_bit=True)
        rng = np.random.default_rng(3)
        for trial in range(44):
            # capacities 4, 8, .. 4096 (next power of two) x two formats = up to 20 shapes, a cache of twelve
            n = (1, 5, 9, 17, 33, 65, 300, 600, 1500, 3000, 2)[trial % 11]
            two_bit = trial % 3 == 0
            e.set_if_format(capi.IF_2BIT_SM if two_bit else capi.IF_1BIT)
            st = np.zeros(n, capi.TRK_DTYPE)
            st["prn"] = rng.choice([5, 30], n)
            st["code_phase_fine"] = rng.uniform(0, 16368, n).astype(np.float32)
            st["if_freq_offset_hz"] = rng.uniform(-4000, 4000, n).astype(np.float32)
            st["if_freq_accum"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
            before = st.copy()
            iq = e.track_epl((two if two_bit else one)[trial & 1], st)
            for c in range(0, n, max(1, n // 5)):
                want, acc = oracle.track_epl(one[trial & 1], oracle.ca_code(int(before["prn"][c])),
                                             float(before["code_phase_fine"][c]), float(before["if_freq_offset_hz"][c]),
                                             int(before["if_freq_accum"][c]))
                assert np.array_equal(iq[c], want) and int(st["if_freq_accum"][c]) == acc, (trial, n, c)
    finally:
        e.close()


def test_track_epl_four_sv_default_table_locks_on_signal(eng, stream):
    """BASELINE.json configs[1]: the reference's 4-SV table (PM/main.c:59-73).  With the true code phase and a
    Doppler within tens of Hz, the prompt correlator must dominate early/late and carry most of the power."""
    from stm32f4_sdr_gps_amd.capi import TRK_DTYPE
    st = np.zeros(4, TRK_DTYPE)
    st["prn"] = [5, 14, 20, 30]
    st["code_phase_fine"] = [1600.0, 4000.0, 9000.0, 13000.0]
    st["if_freq_offset_hz"] = [912.5, 4037.0, -1025.0, 2018.0]
    power = np.zeros((4, 3))
    for ms in range(12):
        iq = eng.track_epl(stream[ms], st).astype(np.int64)
        power += np.stack([iq[:, 0] ** 2 + iq[:, 1] ** 2, iq[:, 2] ** 2 + iq[:, 3] ** 2, iq[:, 4] ** 2 + iq[:, 5] ** 2], 1)
    assert (power[:, 1] > power[:, 0]).all() and (power[:, 1] > power[:, 2]).all()
    assert (np.sqrt(power[:, 1] / 12) > 1500).all()


# ---------------------------------------------------------------------------------------------------------------
# N3: 2-bit (MAX2769 sign + magnitude) ingest
def test_two_bit_ingest_unpack_and_sign_plane_parity(oracle, tmp_path):
    """2-bit captures are unpacked on the GPU (k_unpack2, and inside k_acq / k_track_epl on the way into LDS); the
    correlation runs on the sign plane, so every result must equal the 1-bit path on the same samples."""
    from stm32f4_sdr_gps_amd import capi, synth
    sats = [synth.Sat(5, 912.5, 1600.0, 0.6, 0.3), synth.Sat(30, 2018.0, 13000.0, 0.6, 4.0)]
    one = synth.make_if(4, sats, seed=21)
    two = synth.make_if(4, sats, seed=21, two_bit=True)
    assert two.shape == (4, 4092)
    eng = capi.Engine(0)
    try:
        sign, mag = eng.if_unpack2(two)
        assert np.array_equal(sign, one)
        pairs = np.unpackbits(two, axis=1, bitorder="little")
        assert np.array_equal(mag, np.packbits(pairs[:, 1::2], axis=1, bitorder="little"))
        # file round trip through the raw-stream reader (the replay tool's on-disk format)
        path = tmp_path / "rec_file.bin"
        two.tofile(path)
        assert np.array_equal(synth.read_if_file(str(path), two_bit=True), two)

        prns = np.array([5, 30, 7], np.uint8)
        ref_peaks, ref_keys = eng.acq_grid(one, prns, n_search=2, n_ms=2, dopp_min_hz=500, dopp_step_hz=500, n_dopp=4)
        st0 = np.zeros(2, capi.TRK_DTYPE)
        st0["prn"], st0["code_phase_fine"], st0["if_freq_offset_hz"] = [5, 30], [1600.0, 13003.0], [912.5, 2018.0]
        st_a = st0.copy()
        iq_a = eng.track_epl(one[3], st_a)
        eng.set_if_format(capi.IF_2BIT_SM)
        peaks, keys = eng.acq_grid(two, prns, n_search=2, n_ms=2, dopp_min_hz=500, dopp_step_hz=500, n_dopp=4)
        for f in ("max_val", "phase", "sum", "avr"):
            assert np.array_equal(peaks[f], ref_peaks[f])
        assert np.array_equal(keys, ref_keys)
        jobs = np.zeros(1, capi.JOB_DTYPE)
        jobs[0] = (1, 1, 5, float(IF_HZ + 900), 2, 100, 400)
        pk2, _ = eng.acq_jobs(two, jobs)
        st_b = st0.copy()
        iq_b = eng.track_epl(two[3], st_b)
        assert np.array_equal(iq_a, iq_b) and np.array_equal(st_a, st_b)
        eng.set_if_format(capi.IF_1BIT)
        pk1, _ = eng.acq_jobs(one, jobs)
        assert _peak_tuple(pk1[0]) == _peak_tuple(pk2[0])
        want, _, _ = oracle.search_job(one[1:2], 1, oracle.ca_code(5), float(IF_HZ + 900), 2, 100, 400)
        assert _peak_tuple(pk2[0]) == (want["max_val"], want["phase"], want["sum"], want["avr"])
        with pytest.raises(capi.GpsxError):
            eng.set_if_format(7)
    finally:
        eng.close()


def test_acq_grid_randomised_descriptors_vs_oracle(eng, oracle, stream):
    """Seeded sweep over the descriptor space: PRN lists of any length (1..210, repeats allowed), arbitrary Doppler
    grids, windows, 1..5 ms integration, both phase modes, several searches with overlapping strides, shards."""
    from stm32f4_sdr_gps_amd.capi import PHASES_BYTE, PHASES_FINE
    rng = np.random.default_rng(2024)
    for trial in range(14):
        n_prn = int(rng.integers(1, 12))
        prns = rng.integers(1, 211, n_prn).astype(np.uint8)
        n_dopp = int(rng.integers(1, 5))
        dmin = int(rng.integers(-7000, 6000))
        dstep = int(rng.integers(1, 900))
        n_ms = int(rng.integers(1, 6))
        n_search = int(rng.integers(1, 4))
        stride = int(rng.integers(0, 3))
        if (n_search - 1) * stride + n_ms > len(stream):
            n_search, stride = 1, 1
        a, b = sorted(int(x) for x in rng.integers(0, 2047, 2))
        mode = PHASES_FINE if trial % 3 else PHASES_BYTE
        n_bits = 8 if mode == PHASES_FINE else 1
        world = int(rng.integers(1, 4))
        merged = None
        for r in range(world):
            peaks, keys = eng.acq_grid(stream, prns, n_search=n_search, n_ms=n_ms, search_stride_blocks=stride,
                                       dopp_min_hz=dmin, dopp_step_hz=dstep, n_dopp=n_dopp, phase_mode=mode, win=(a, b),
                                       shard=(r, world))
            merged = _merge_owned(np.zeros_like(peaks) if merged is None else merged, peaks, keys)
        for s in range(n_search):
            blk = stream[s * stride:s * stride + n_ms]
            for p in range(n_prn):
                for d in range(n_dopp):
                    for bit in range(n_bits):
                        pk, _, _ = oracle.search_job(blk, n_ms, oracle.ca_code(int(prns[p])), float(IF_HZ + dmin + d * dstep),
                                                     bit, a, b)
                        assert _peak_tuple(merged[s, p, d, bit]) == (pk["max_val"], pk["phase"], pk["sum"], pk["avr"]), \
                            (trial, s, p, d, bit)


def _merge_owned(acc, peaks, keys):
    """Shards write zeros for units they do not own; owned units are those with a non-zero key."""
    out = acc.copy()
    mine = keys != 0
    out[mine] = peaks[mine]
    return out


def test_kernels_do_not_write_outside_their_outputs(eng, stream):
    """Out-of-bounds canaries (no sanitizer exists for the device side): every device output is placed
    between guard regions filled with a pattern, for descriptors with partial PRN groups and odd sizes, in every kernel
    variant the API can select; the guards must come back untouched."""
    import ctypes as C
    from stm32f4_sdr_gps_amd import capi
    GUARD = 4096
    pattern = np.frombuffer(np.random.default_rng(99).bytes(GUARD), np.uint8)

    def guarded(nbytes):
        total = nbytes + 2 * GUARD
        p = eng.malloc(total)
        eng.h2d(p, np.concatenate([pattern, np.zeros(nbytes, np.uint8), pattern]))
        return p, p + GUARD, nbytes

    def check(p, nbytes):
        back = np.zeros(nbytes + 2 * GUARD, np.uint8)
        eng.d2h(back, p)
        eng.free(p)
        assert np.array_equal(back[:GUARD], pattern) and np.array_equal(back[-GUARD:], pattern)
        return back[GUARD:-GUARD]

    blocks = np.ascontiguousarray(stream[:6])
    d_if = eng.malloc(blocks.size + 2)
    eng.h2d(d_if, np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
    for (n_prn, n_dopp, n_search, n_ms, mode, dbg) in [(13, 3, 2, 1, capi.PHASES_FINE, False),   # polyphase, 8-offset form
                                                        (9, 2, 1, 1, capi.PHASES_BYTE, False),    # dot8, byte phases
                                                        (5, 2, 2, 3, capi.PHASES_FINE, False),    # multi-block
                                                        (3, 1, 1, 2, capi.PHASES_FINE, True)]:    # inspection outputs
        prns = np.arange(1, n_prn + 1, dtype=np.uint8)
        g = eng.grid_desc(prns, n_search=n_search, n_ms=n_ms, search_stride_blocks=n_ms, dopp_min_hz=-500,
                          dopp_step_hz=500, n_dopp=n_dopp, phase_mode=mode)
        n_bits = 8 if mode == capi.PHASES_FINE else 1
        n_pk = n_search * n_prn * n_dopp * n_bits
        bufs = [guarded(n_pk * 16), guarded(n_search * n_prn * n_dopp * 8)]
        if dbg:
            bufs += [guarded(n_pk * n_ms * 16), guarded(n_pk * 2046 * 4), guarded(n_pk * 2046 * 4)]
        ptrs = [b[1] for b in bufs] + [None] * (5 - len(bufs))
        rc = eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(g), d_if, len(blocks), *ptrs)
        assert rc == 0
        eng.synchronize()
        outs = [check(b[0], b[2]) for b in bufs]
        assert outs[0].view(capi.PEAK_DTYPE)["sum"].all()      # every owned entry was produced
    # tracking
    st = np.zeros(37, capi.TRK_DTYPE)
    st["prn"] = (np.arange(37) % 32) + 1
    st["code_phase_fine"] = np.linspace(0, 16367, 37).astype(np.float32)
    p_st, d_st, n_st = guarded(st.nbytes)
    eng.h2d(d_st, st)
    p_iq, d_iq, n_iq = guarded(37 * 12)
    assert eng.lib.gpsx_track_epl_batch_dev(eng.h, d_if, d_st, 37, d_iq) == 0
    eng.synchronize()
    check(p_st, n_st)
    check(p_iq, n_iq)
    eng.free(d_if)


# ---- IF ingest: capture ring (include/gpsx.h; PM/signal_capture.c:14-24,57-82, PC_SpiLight replay) ---------------------

def test_capture_ring_mirror_windows_and_wrap(eng, stream):
    from stm32f4_sdr_gps_amd import capi
    cap = capi.Capture(eng, 4)
    try:
        assert cap.block_bytes == 2046 and cap.packet_cnt() == 0 and cap.ready_ptr() == 0
        with pytest.raises(capi.GpsxError):
            cap.window_dev(1)                                    # nothing received yet
        for t in range(6):                                       # slots 0 1 2 3 0 1: the ring has wrapped
            cap.push(stream[t])
            assert cap.packet_cnt() == t + 1
            assert np.array_equal(cap.ready_view()[0], stream[t])
        for n in (1, 3, 4):                                      # 3 and 4 straddle the wrap: contiguous in HBM anyway
            back = np.zeros((n, 2046), np.uint8)
            eng.d2h(back, cap.window_dev(n))
            assert np.array_equal(back, stream[6 - n:6]), n
        with pytest.raises(capi.GpsxError):
            cap.window_dev(5)                                    # more than the ring holds
    finally:
        cap.close()


def test_host_entry_points_read_ring_pointers_from_the_mirror(eng, stream):
    """Pointers into committed ring slots are served from HBM; results equal the plain host-buffer path.  A slot handed
    out for writing again is no longer trusted: the call then copies what the host memory holds NOW."""
    from stm32f4_sdr_gps_amd import capi
    prns = np.array([5, 14, 20, 30], np.uint8)
    kw = dict(n_search=1, n_ms=2, dopp_min_hz=500, dopp_step_hz=500, n_dopp=3)
    want_pk, want_keys = eng.acq_grid(stream[2:4], prns, **kw)
    st0 = np.zeros(4, capi.TRK_DTYPE)
    st0["prn"] = prns
    st0["code_phase_fine"] = [1600.0, 4000.5, 9003.0, 13007.9]
    st0["if_freq_offset_hz"] = [912.5, 4037.0, -1025.0, 2018.0]
    st_w = st0.copy()
    want_iq = eng.track_epl(stream[3], st_w)
    cap = capi.Capture(eng, 8)
    try:
        for t in range(4):
            cap.push(stream[t])
        pk, keys = eng.acq_grid(cap.ready_view(2), prns, **kw)                 # slots 2..3
        assert np.array_equal(pk, want_pk) and np.array_equal(keys, want_keys)
        st = st0.copy()
        assert np.array_equal(eng.track_epl(cap.ready_view()[0], st), want_iq) and np.array_equal(st, st_w)
        # staleness: take slot 4 for writing, scribble over it WITHOUT committing -> must not be read from the mirror
        slot = eng.lib.gpsx_capture_write_slot(cap.h)
        view = np.frombuffer((C.c_uint8 * 2046).from_address(slot), np.uint8)
        view[:] = stream[3]
        st = st0.copy()
        assert np.array_equal(eng.track_epl(view, st), want_iq)
        # and a committed slot whose HOST bytes are then modified keeps serving the committed block (documented: the
        # ring belongs to the producer between write_slot and commit only)
        eng._chk(eng.lib.gpsx_capture_commit(cap.h), "commit")
        assert cap.packet_cnt() == 5
    finally:
        cap.close()


def test_capture_replay_of_a_recorded_if_file(eng, stream, tmp_path):
    from stm32f4_sdr_gps_amd import capi
    path = str(tmp_path / "rec_file.bin")
    stream[:9].tofile(path)
    with open(path, "ab") as f:
        f.write(b"\x55" * 1000)                                  # trailing partial block: dropped
    prns = np.array([5, 14, 20, 30], np.uint8)
    st0 = np.zeros(4, capi.TRK_DTYPE)
    st0["prn"] = prns
    st0["code_phase_fine"] = [1600.0, 4000.5, 9003.0, 13007.9]
    st0["if_freq_offset_hz"] = [912.5, 4037.0, -1025.0, 2018.0]
    cap = capi.Capture(eng, 2)
    try:
        seen, st, iqs = [], st0.copy(), []

        def on_block(idx):
            seen.append(idx)
            iqs.append(eng.track_epl(cap.ready_view()[0], st).copy())          # per-ms processing on the fresh block
            return 1 if idx == 7 else 0                                         # stop early once

        assert cap.replay_file(path, first_block=2, on_block=on_block) == 6 and seen == [2, 3, 4, 5, 6, 7]
        st_w, want = st0.copy(), []
        for t in range(2, 8):
            want.append(eng.track_epl(stream[t], st_w).copy())
        assert np.array_equal(np.array(iqs), np.array(want)) and np.array_equal(st, st_w)
        assert cap.replay_file(path, first_block=7, max_blocks=-1) == 2         # blocks 7, 8; the partial tail is not one
        assert cap.replay_file(path, first_block=0, max_blocks=3) == 3
        assert cap.packet_cnt() == 11
        with pytest.raises(capi.GpsxError):
            cap.replay_file(str(tmp_path / "missing.bin"))
    finally:
        cap.close()


# ---- the alternative grid kernels stay bit-identical to the default ($GPSX_ACQ_ALGO, read when a context is created) --

_ALT_CASES = (dict(n_search=2, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21),
              dict(n_search=1, dopp_min_hz=-6000, dopp_step_hz=250, n_dopp=45, win=(100, 1901)),
              dict(n_search=1, dopp_min_hz=-12000, dopp_step_hz=4000, n_dopp=7, win=(0, 3)),
              dict(n_search=2, n_ms=3, search_stride_blocks=2, dopp_min_hz=-1000, dopp_step_hz=500, n_dopp=5))
_ALT_PRNS = np.array([1, 5, 7, 14, 20, 25, 30, 31, 32, 3, 12], np.uint8)
_alt_want = {}


def _alt_oracle(oracle, stream, case):
    """The oracle's triplets and keys for _ALT_CASES[case] on stream[:5] (computed once for all seven kernels)."""
    if case not in _alt_want:
        kw = _ALT_CASES[case]
        n_ms, stride = kw.get("n_ms", 1), kw.get("search_stride_blocks", kw.get("n_ms", 1))
        start, stop = kw.get("win", (0, 2046))
        pk = np.zeros((kw["n_search"], len(_ALT_PRNS), kw["n_dopp"], 8), capi_peak_dtype())
        from concurrent.futures import ThreadPoolExecutor
        codes = [oracle.ca_code(int(prn)) for prn in _ALT_PRNS]

        def job(idx):       # one windowed search of the oracle (ctypes releases the GIL: the jobs run on ORC_THREADS cores)
            s_, p, d, b = idx
            one = oracle.search_job(stream[s_ * stride:s_ * stride + n_ms], n_ms, codes[p],
                                    float(IF_HZ + kw["dopp_min_hz"] + d * kw["dopp_step_hz"]), b, start, stop)
            one = one[0] if isinstance(one, tuple) else one
            return idx, (one["max_val"], one["phase"], one["sum"], one["avr"])
        with ThreadPoolExecutor(ORC_THREADS) as ex:
            for idx, rec in ex.map(job, list(np.ndindex(kw["n_search"], len(_ALT_PRNS), kw["n_dopp"], 8))):
                pk[idx] = rec
        fine = 8 * pk["phase"].astype(np.int64) + np.arange(8)[None, None, None, :]
        keys = ((pk["max_val"].astype(np.int64) << 14) | (16383 - fine)).max(axis=3)
        _alt_want[case] = (pk, keys)
    return _alt_want[case]


def capi_peak_dtype():
    from stm32f4_sdr_gps_amd import capi
    return capi.PEAK_DTYPE


_ALT_KERNELS = {   # what gpsx_last_kernel must report per algorithm for (single-block, multi-block) fine grids
    "mx": (b"k_acq_mx<5>", b"k_acq_mx<"), "mx2": (b"k_acq_mx<5>", b"k_acq_mx<"), "mx4": (b"k_acq_mx<5>", b"k_acq_mx<"), "mx8": (b"k_acq_mx<5>", b"k_acq_mx<"), "poly": (b"k_acq_poly<", b"k_acq_poly<"), "dot8": (b"k_acq<8,false,dot8>", b"k_acq<8,true,dot8>"),
    "sad": (b"k_acq<8,false,sad>", b"k_acq<8,true,sad>"), "seg4": (b"k_acq_poly<", b"k_acq_poly<"), "seg8": (b"k_acq_poly<", b"k_acq_poly<"),
    "seg16": (b"k_acq_poly<", b"k_acq_poly<")}


@pytest.mark.parametrize("algo", ["mx", "mx2", "mx4", "mx8", "poly", "dot8", "sad", "seg4", "seg8", "seg16"])
def test_alternative_grid_kernels_match_the_oracle(oracle, stream, algo, monkeypatch):
    """Every acquisition kernel in the library against the CPU oracle (not against each other: the default IS mx, a
    comparison with it would be vacuous for mx): mx = the matrix-core kernel, poly = the polyphase VALU kernel, dot8 / sad =
    the direct forms, seg* = the polyphase kernel at a forced number of sample offsets per workgroup (GPSX_ACQ_SEG selects
    the polyphase kernel by itself); mx on launches this small is the split form k_acq_mx<5> (mx2 / mx4 / mx8: two / four /
    eight workgroups per cluster, i.e. direct starts at sample offsets 8 / 4, 8, 12 / 2, 4 .. 14 -- every quirk term as a
    start value).  Windows, a PRN count that is not a multiple of the group, odd Doppler counts and
    steps, multi-block searches; gpsx_last_kernel must name the kernel under test."""
    from stm32f4_sdr_gps_amd import capi
    var, val = ("GPSX_ACQ_SEG", algo[3:]) if algo.startswith("seg") else ("GPSX_ACQ_ALGO", algo[:2] if algo.startswith("mx") else algo)
    monkeypatch.setenv(var, val)
    if algo in ("mx2", "mx4", "mx8"):      # the split form at two / four / eight workgroups per cluster, whatever the launch size
        monkeypatch.setenv("GPSX_ACQ_SPLIT", algo[2])
    alt = capi.Engine(0, lab=True)
    monkeypatch.delenv(var)
    try:
        prns = _ALT_PRNS
        for case, kw in enumerate(_ALT_CASES):
            want_pk, want_keys = _alt_oracle(oracle, stream, case)
            pk, keys = alt.acq_grid(stream[:5], prns, **kw)
            ran = alt.lib.gpsx_last_kernel(alt.h)
            assert ran.startswith(_ALT_KERNELS[algo][kw.get("n_ms", 1) > 1]), (algo, kw, ran)
            for f in ("max_val", "phase", "sum", "avr"):
                assert np.array_equal(pk[f], want_pk[f]), (algo, kw, f)
            assert np.array_equal(keys, want_keys), (algo, kw)
        # and on 2-bit sign/magnitude captures (each kernel has its own unpack on the way into LDS)
        from stm32f4_sdr_gps_amd import synth
        sats = [synth.Sat(5, 912.5, 1600.0, 0.6, 0.3), synth.Sat(30, 2018.0, 13000.0, 0.6, 4.0)]
        two = synth.make_if(3, sats, seed=21, two_bit=True)
        one = synth.make_if(3, sats, seed=21)
        prns4 = np.array([5, 30, 1], np.uint8)
        alt.set_if_format(capi.IF_2BIT_SM)
        pk, keys = alt.acq_grid(two, prns4, n_search=3, dopp_min_hz=500, dopp_step_hz=500, n_dopp=4)
        for s_ in range(3):
            want = oracle.acq_grid(one[s_:s_ + 1], 1, prns4, 500, 500, 4, 8, n_threads=ORC_THREADS)
            for f in ("max_val", "phase", "sum", "avr"):
                assert np.array_equal(pk[s_][f], want[f]), (algo, s_, f)
    finally:
        alt.close()


@pytest.fixture(scope="module")
def eng_poly():
    """An engine whose fine grids run the polyphase VALU kernel (k_acq_poly.hip): the independent implementation the
    matrix-core kernel is compared with where the oracle would take too long."""
    import os
    from stm32f4_sdr_gps_amd import capi
    old = os.environ.get("GPSX_ACQ_ALGO")
    os.environ["GPSX_ACQ_ALGO"] = "poly"
    try:
        e = capi.Engine(0, lab=True)
    finally:
        if old is None:
            del os.environ["GPSX_ACQ_ALGO"]
        else:
            os.environ["GPSX_ACQ_ALGO"] = old
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_mx():
    """An engine whose single-block fine grids all run k_acq_mx<0> -- the bench's kernel, one workgroup per cluster -- whatever
    their size ($GPSX_ACQ_NO_SPLIT: small launches would otherwise take the split form k_acq_mx<5>, which the default
    engine's tests cover)."""
    import os
    from stm32f4_sdr_gps_amd import capi
    old = {k: os.environ.get(k) for k in ("GPSX_ACQ_ALGO", "GPSX_ACQ_NO_SPLIT")}
    os.environ["GPSX_ACQ_ALGO"] = "mx"
    os.environ["GPSX_ACQ_NO_SPLIT"] = "1"
    try:
        e = capi.Engine(0, lab=True)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    yield e
    e.close()


def test_matrix_core_grid_vs_oracle(eng_mx, oracle, stream):
    """k_acq_mx against the CPU oracle directly: 32 PRNs (one full cluster) x 3 Doppler bins x 16368 phases on two
    captures, and the 10-block non-coherent sum (BASELINE.json configs[3]'s integration) on a 5-PRN list."""
    prns = np.arange(1, 33, dtype=np.uint8)
    peaks, keys = eng_mx.acq_grid(stream[:2], prns, n_search=2, dopp_min_hz=500, dopp_step_hz=1500, n_dopp=3)
    assert eng_mx.lib.gpsx_last_kernel(eng_mx.h) == b"k_acq_mx<0>"
    for s in range(2):
        want = oracle.acq_grid(stream[s:s + 1], 1, prns, 500, 1500, 3, 8, n_threads=ORC_THREADS)
        for f in ("max_val", "phase", "sum", "avr"):
            assert np.array_equal(peaks[s][f], want[f]), (s, f)
    prns5 = np.array([5, 14, 20, 30, 7], np.uint8)
    peaks, _ = eng_mx.acq_grid(stream[:10], prns5, n_search=1, n_ms=10, dopp_min_hz=-1000, dopp_step_hz=500, n_dopp=5)
    assert eng_mx.lib.gpsx_last_kernel(eng_mx.h) == b"k_acq_mx<2>"
    want = oracle.acq_grid(stream[:10], 10, prns5, -1000, 500, 5, 8, n_threads=ORC_THREADS)
    for f in ("max_val", "phase", "sum", "avr"):
        assert np.array_equal(peaks[0][f], want[f]), f


def test_matrix_core_grid_windows_sets_and_shards(eng_mx, eng_poly, oracle, stream):
    """40 PRNs = two 32-slot clusters (the second one a single 8-PRN group), windows, and every shard of 2, 3, 5 and 8:
    a shard's run of units cuts clusters anywhere, the kernel then skips the foreign 8-PRN groups of a workgroup.  The
    expected values come from the polyphase VALU kernel (a different algorithm; itself checked against the oracle in
    test_alternative_grid_kernels_match_the_oracle) and, for the PRN-count edge cases, from the oracle."""
    prns = np.concatenate([np.arange(1, 33), [33, 40, 61, 100, 120, 150, 200, 210]]).astype(np.uint8)
    for kw in (dict(n_search=2, dopp_min_hz=-2000, dopp_step_hz=1000, n_dopp=5),
               dict(n_search=1, dopp_min_hz=250, dopp_step_hz=500, n_dopp=2, win=(100, 1901)),
               dict(n_search=1, dopp_min_hz=0, dopp_step_hz=500, n_dopp=1, win=(7, 8))):
        want_pk, want_keys = eng_poly.acq_grid(stream[:2], prns, **kw)
        assert eng_poly.lib.gpsx_last_kernel(eng_poly.h).startswith(b"k_acq_poly<")   # (an independent kernel, not mx again)
        pk, keys = eng_mx.acq_grid(stream[:2], prns, **kw)
        assert eng_mx.lib.gpsx_last_kernel(eng_mx.h) == b"k_acq_mx<0>"
        assert np.array_equal(pk, want_pk) and np.array_equal(keys, want_keys), kw
    # a single PRN (one row of the GEMM in use) and 33 (a second cluster holding one PRN), against the oracle
    for plist in (np.array([17], np.uint8), np.arange(1, 34, dtype=np.uint8)):
        pk, _ = eng_mx.acq_grid(stream[3:4], plist, n_search=1, dopp_min_hz=-750, dopp_step_hz=1500, n_dopp=2)
        want = oracle.acq_grid(stream[3:4], 1, plist, -750, 1500, 2, 8, n_threads=ORC_THREADS)
        for f in ("max_val", "phase", "sum", "avr"):
            assert np.array_equal(pk[0][f], want[f]), (len(plist), f)
    kw = dict(n_search=2, dopp_min_hz=-2000, dopp_step_hz=1000, n_dopp=5)
    want_pk, want_keys = eng_poly.acq_grid(stream[:2], prns, **kw)
    for world in (2, 3, 5, 8):
        acc = np.zeros_like(want_keys)
        owned = np.zeros(want_keys.shape, np.int32)
        for r in range(world):
            pk, ks = eng_mx.acq_grid(stream[:2], prns, shard=(r, world), **kw)
            mine = ks != 0
            owned += mine
            acc = np.maximum(acc, ks)
            for f in ("max_val", "phase", "sum", "avr"):
                assert np.array_equal(pk[f][mine], want_pk[f][mine]), (world, r, f)
            assert not pk["sum"][~mine].any()
        assert np.array_equal(acc, want_keys) and (owned == 1).all()


@pytest.mark.parametrize("amp_scale", [0.25, 1.0])
def test_bench_size_batch_equals_single_capture_launches_and_finds_the_satellites(amp_scale, oracle):
    """BASELINE.json configs[2] at the size bench.py runs it (256 captures x 32 PRN x 21 Doppler x 16368 phases per launch,
    2-bit IF: 5376 workgroups, 21 rounds of the chip; amp_scale 0.25 is bench.py's own input -- satellites below the noise
    --, 1.0 the strong test signal).  Every capture of the batch must get exactly the triplets and keys it gets when
    launched alone; three captures (first, middle, last round) are checked against the CPU oracle triplet by triplet; and
    in at least 80 % of the captures the strongest hypothesis of each present PRN sits on that satellite's Doppler bin and,
    to two samples, on its code phase."""
    from stm32f4_sdr_gps_amd import capi, synth
    e = capi.Engine(0)
    try:
        n = 256
        blocks2 = synth.cold_start_block(n, seed=11, amp_scale=amp_scale, two_bit=True)
        blocks1 = synth.cold_start_block(n, seed=11, amp_scale=amp_scale)         # the same stream's sign plane, 1 bit
        e.set_if_format(capi.IF_2BIT_SM)
        prns = np.arange(1, 33, dtype=np.uint8)
        kw = dict(dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
        pk, keys = e.acq_grid(blocks2, prns, n_search=n, **kw)
        assert e.lib.gpsx_last_kernel(e.h) == b"k_acq_mx<0>"
        for i in (0, 1, 17, 40, 63, 130, 255):
            pk1, keys1 = e.acq_grid(blocks2[i:i + 1], prns, n_search=1, **kw)
            assert np.array_equal(pk1[0], pk[i]) and np.array_equal(keys1[0], keys[i]), i
        for i in (0, 131, 255):     # (capture 131 against the oracle computed live, the other two against its committed fixtures)
            want = oracle.acq_grid(blocks1[i:i + 1], 1, prns, -5000, 500, 21, 8, n_threads=ORC_THREADS, live=(i == 131))
            for f in ("max_val", "phase", "sum", "avr"):
                assert np.array_equal(pk[i][f], want[f]), (i, f)
        assert (keys >> 14).min() > 0                              # every (capture, PRN, Doppler) search produced a peak
        if amp_scale < 1.0:
            return      # below the noise a single millisecond does not acquire: the equalities above are the whole claim
        # (PRN, Doppler Hz, delay in samples) of synth.cold_start_block; code phase = samples into the block at which the
        # code starts = delay mod 16368 (the stream is continuous: every capture sees the same alignment)
        for prn, dopp, delay in ((3, -3210.0, 777.0), (5, 912.5, 1600.0), (11, 4480.0, 12001.0), (14, 4037.0, 4000.0),
                                 (20, -1025.0, 9000.0), (30, 2018.0, 13000.0)):
            k = keys[:, prn - 1, :]                                   # [capture, Doppler bin]
            best_bin = np.argmax(k, axis=1)
            fine = 16383 - (k[np.arange(n), best_bin] & 16383)        # 8 * byte offset + bit shift = sample offset
            err = (fine.astype(np.int64) - int(delay) + 8184) % 16368 - 8184
            # (not every capture: the reference's one-sided clip, quirk Q4, blanks a satellite whenever its carrier phase
            #  puts I or Q negative over the millisecond, and a neighbouring bin or a noise peak then wins)
            hit = (np.abs(best_bin - (dopp + 5000) / 500) <= 1.6) & (np.abs(err) <= 2)
            assert hit.sum() >= 0.8 * n, (prn, int(hit.sum()))
            assert np.all((k[np.arange(n), best_bin] >> 14)[hit] > 400), prn
    finally:
        e.close()


def test_acq_grid_async_four_contexts_in_rotation_vs_oracle_and_device_path(oracle):
    """gpsx_acq_grid_async -- the entry point behind bench.py's `pcie_inclusive` (host buffers in, host buffers out, several
    contexts in rotation so that one call's transfers run under the others' sweeps) -- with page-locked buffers from
    gpsx_host_alloc, exactly as the bench drives it: (a) four contexts, three rounds, every call a different pair of
    captures, every triplet and key against the CPU oracle; (b) the bench's 256-capture batch through each of the four
    contexts, against gpsx_acq_grid_dev on the same batch resident in HBM (every byte), and three of its captures against
    the oracle."""
    import ctypes as C
    from stm32f4_sdr_gps_amd import capi, synth
    n_ctx, n_big = 4, 256
    blocks2 = synth.cold_start_block(n_big, seed=11, amp_scale=0.25, two_bit=True)
    blocks1 = synth.cold_start_block(n_big, seed=11, amp_scale=0.25)
    prns = np.arange(1, 33, dtype=np.uint8)
    engs = [capi.Engine(0) for _ in range(n_ctx)]
    try:
        for e in engs:
            e.set_if_format(capi.IF_2BIT_SM)
        # (a) small calls in rotation: call j on context j % 4 sweeps captures (2 j, 2 j + 1)
        n_calls = 3 * n_ctx
        g2 = engs[0].grid_desc(prns, n_search=2, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
        bufs = []
        for j in range(n_calls):
            e = engs[j % n_ctx]
            pin_if = e.host_array((2, capi.BYTES_PER_MS_2BIT), np.uint8)
            pin_pk = e.host_array((2, 32, 21, 8), capi.PEAK_DTYPE)
            pin_keys = e.host_array((2, 32, 21), np.int64)
            pin_if[:] = blocks2[2 * j:2 * j + 2]
            pin_pk[:] = np.zeros((), capi.PEAK_DTYPE)
            pin_keys[:] = -1
            bufs.append((pin_pk, pin_keys))
            rc = e.lib.gpsx_acq_grid_async(e.h, C.byref(g2), pin_if.ctypes.data, 2, pin_pk.ctypes.data, pin_keys.ctypes.data)
            assert rc == 0, e.lib.gpsx_last_error(e.h)
        for e in engs:
            e.synchronize()
        for j in range(n_calls):
            pin_pk, pin_keys = bufs[j]
            for s_ in range(2):
                want = oracle.acq_grid(blocks1[2 * j + s_:2 * j + s_ + 1], 1, prns, -5000, 500, 21, 8, n_threads=ORC_THREADS)
                for f in ("max_val", "phase", "sum", "avr"):
                    assert np.array_equal(pin_pk[s_][f], want[f]), (j, s_, f)
                fine = 8 * want["phase"].astype(np.int64) + np.arange(8)[None, None, :]
                k = ((want["max_val"].astype(np.int64) << 14) | (16383 - fine)).max(axis=2)
                assert np.array_equal(pin_keys[s_], k), (j, s_)
        # (b) the bench batch: device-resident reference result first
        e0 = engs[0]
        gb = e0.grid_desc(prns, n_search=n_big, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
        n_pk, n_keys = n_big * 32 * 21 * 8, n_big * 32 * 21
        d_if = e0.malloc(blocks2.nbytes + 2)
        d_pk = e0.malloc(n_pk * capi.PEAK_DTYPE.itemsize)
        d_keys = e0.malloc(n_keys * 8)
        e0.h2d(d_if, blocks2)
        rc = e0.lib.gpsx_acq_grid_dev(e0.h, C.byref(gb), d_if, n_big, d_pk, d_keys, None, None, None)
        assert rc == 0
        e0.synchronize()
        assert e0.lib.gpsx_last_kernel(e0.h) == b"k_acq_mx<0>"
        dev_pk = np.zeros((n_big, 32, 21, 8), capi.PEAK_DTYPE)
        dev_keys = np.zeros((n_big, 32, 21), np.int64)
        e0.d2h(dev_pk, d_pk)
        e0.d2h(dev_keys, d_keys)
        for p in (d_if, d_pk, d_keys):
            e0.free(p)
        big = []
        for e in engs:
            pin_if = e.host_array(blocks2.shape, np.uint8)
            pin_pk = e.host_array((n_big, 32, 21, 8), capi.PEAK_DTYPE)
            pin_keys = e.host_array((n_big, 32, 21), np.int64)
            pin_if[:] = blocks2
            big.append((pin_if, pin_pk, pin_keys))
        for rnd in range(2):                      # two rounds of the rotation: the second reuses arenas still in flight
            for e, (pin_if, pin_pk, pin_keys) in zip(engs, big):
                rc = e.lib.gpsx_acq_grid_async(e.h, C.byref(gb), pin_if.ctypes.data, n_big, pin_pk.ctypes.data, pin_keys.ctypes.data)
                assert rc == 0, e.lib.gpsx_last_error(e.h)
        for e in engs:
            e.synchronize()
        for i, (_, pin_pk, pin_keys) in enumerate(big):
            assert np.array_equal(pin_pk, dev_pk), i
            assert np.array_equal(pin_keys, dev_keys), i
        for i in (7, 128, 249):
            want = oracle.acq_grid(blocks1[i:i + 1], 1, prns, -5000, 500, 21, 8, n_threads=ORC_THREADS)
            for f in ("max_val", "phase", "sum", "avr"):
                assert np.array_equal(dev_pk[i][f], want[f]), (i, f)
    finally:
        for e in engs:
            e.close()


def test_in_process_group_sharded_sweep_over_rccl(eng, stream):
    """gpsx_group_* / gpsx_acq_grid_sharded: the C host's multi-GPU sweep (RCCL communicators inside one process).  One
    GPU here, so a group of one: the launch with shard (0, 1) followed by a real ncclAllReduce(MAX) over a 1-rank
    communicator must give the plain sweep's keys; two contexts on one device are refused."""
    from stm32f4_sdr_gps_amd import capi
    lib = eng.lib
    prns = np.array([5, 14, 20, 30, 1, 2, 3], np.uint8)
    kw = dict(n_search=2, dopp_min_hz=-1000, dopp_step_hz=500, n_dopp=5)
    want_pk, want_keys = eng.acq_grid(stream[:2], prns, **kw)
    g = eng.grid_desc(prns, **kw)
    n_pk, n_keys = want_pk.size, want_keys.size
    d_if, d_pk, d_keys = eng.malloc(2 * 2046 + 2), eng.malloc(n_pk * 16), eng.malloc(n_keys * 8)
    eng.h2d(d_if, np.concatenate([stream[:2].reshape(-1), np.zeros(2, np.uint8)]))
    grp = C.c_void_p()
    ctxs = (C.c_void_p * 1)(eng.h)
    eng._chk(lib.gpsx_group_create(ctxs, 1, C.byref(grp)), "gpsx_group_create")
    try:
        ifs, pks, ks = (C.c_void_p * 1)(d_if), (C.c_void_p * 1)(d_pk), (C.c_void_p * 1)(d_keys)
        eng._chk(lib.gpsx_acq_grid_sharded(grp, C.byref(g), ifs, 2, pks, ks), "gpsx_acq_grid_sharded")
        keys = np.zeros_like(want_keys)
        pk = np.zeros_like(want_pk)
        eng.d2h(keys, d_keys)
        eng.d2h(pk, d_pk)
        assert np.array_equal(keys, want_keys) and np.array_equal(pk, want_pk)
    finally:
        lib.gpsx_group_destroy(grp)
    other = capi.Engine(0)
    try:
        two = (C.c_void_p * 2)(eng.h, other.h)
        assert lib.gpsx_group_create(two, 2, C.byref(grp)) == -22
    finally:
        other.close()


def _device_count():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.parametrize("n_dev", [2, 4, 8])
def test_in_process_group_of_n_devices_all_reduces_over_rccl(stream, oracle, n_dev):
    """gpsx_group_create / gpsx_acq_grid_sharded with n > 1 REAL ranks: one context per device, RCCL communicators from
    ncclCommInitAll, each device sweeps its contiguous run of (search, Doppler, 8-PRN group) units, ONE ncclAllReduce(MAX,
    int64) over xGMI merges the key tables -- every device must then hold the oracle's keys for the whole grid, and its own
    units' triplets.  Skips on a box with fewer devices (the one-GPU boxes these tests usually see): it runs the moment a
    multi-GPU node collects it."""
    if _device_count() < n_dev:
        pytest.skip(f"needs {n_dev} GPUs in one node")
    from stm32f4_sdr_gps_amd import capi
    engs = [capi.Engine(d) for d in range(n_dev)]
    lib = engs[0].lib
    grp = C.c_void_p()
    bufs = []
    try:
        prns = np.arange(1, 33, dtype=np.uint8)
        kw = dict(n_search=2, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
        g = engs[0].grid_desc(prns, **kw)
        n_pk, n_keys = 2 * 32 * 21 * 8, 2 * 32 * 21
        blocks = np.concatenate([np.ascontiguousarray(stream[:2]).reshape(-1), np.zeros(2, np.uint8)])
        for e in engs:
            d_if, d_pk, d_keys = e.malloc(blocks.nbytes), e.malloc(n_pk * 16), e.malloc(n_keys * 8)
            e.h2d(d_if, blocks)
            bufs.append((e, d_if, d_pk, d_keys))
        handles = (C.c_void_p * n_dev)(*[e.h for e in engs])
        assert lib.gpsx_group_create(handles, n_dev, C.byref(grp)) == 0, lib.gpsx_last_error(engs[0].h)
        ifs = (C.c_void_p * n_dev)(*[b[1] for b in bufs])
        pks = (C.c_void_p * n_dev)(*[b[2] for b in bufs])
        ks = (C.c_void_p * n_dev)(*[b[3] for b in bufs])
        for rep in range(3):      # the communicators are reused across sweeps
            assert lib.gpsx_acq_grid_sharded(grp, C.byref(g), ifs, 2, pks, ks) == 0, lib.gpsx_last_error(engs[0].h)
        want_keys = np.zeros((2, 32, 21), np.int64)
        want_pk = []
        for s_ in range(2):
            w = oracle.acq_grid(stream[s_:s_ + 1], 1, prns, -5000, 500, 21, 8, n_threads=ORC_THREADS)
            fine = 8 * w["phase"].astype(np.int64) + np.arange(8)[None, None, :]
            want_keys[s_] = ((w["max_val"].astype(np.int64) << 14) | (16383 - fine)).max(axis=2)
            want_pk.append(w)
        n_units = 2 * 21 * 4
        owned = np.zeros((2, 32, 21), np.int32)
        for r, (e, _, d_pk, d_keys) in enumerate(bufs):
            e.synchronize()
            keys = np.zeros((2, 32, 21), np.int64)
            pk = np.zeros((2, 32, 21, 8), capi.PEAK_DTYPE)
            e.d2h(keys, d_keys)
            e.d2h(pk, d_pk)
            assert np.array_equal(keys, want_keys), r            # the merged table, on every device
            lo, hi = n_units * r // n_dev, n_units * (r + 1) // n_dev
            for u in range(lo, hi):                               # this device's own units: full triplets
                s_, d, grp8 = u // (21 * 4), (u // 4) % 21, u % 4
                owned[s_, 8 * grp8:8 * grp8 + 8, d] += 1
                for f in ("max_val", "phase", "sum", "avr"):
                    assert np.array_equal(pk[s_, 8 * grp8:8 * grp8 + 8, d][f], want_pk[s_][f][8 * grp8:8 * grp8 + 8, d]), (r, u, f)
        assert (owned == 1).all()
    finally:
        if grp:
            lib.gpsx_group_destroy(grp)
        for e, d_if, d_pk, d_keys in bufs:
            for p in (d_if, d_pk, d_keys):
                e.free(p)
        for e in engs:
            e.close()


@pytest.mark.parametrize("algo", ["mx", "poly"])
@pytest.mark.parametrize("mode", ["walk", "blocks"])
def test_both_multi_block_forms_match_the_oracle(oracle, stream, mode, algo, monkeypatch):
    """n_ms > 1 has two forms on the matrix-core kernel and on the polyphase kernel alike, picked by launch size: `walk` (a workgroup walks the blocks of its
    unit, running sums in an HBM slice) for many searches, `blocks` (a workgroup per (unit, block), all magnitudes
    through HBM as u16, k_acq_vals_search sums and searches) for few.  $GPSX_ACQ_MS_MODE forces one: both must give the
    oracle's triplets -- windows, stride, a PRN count off the group size, sharded halves included."""
    from stm32f4_sdr_gps_amd import capi
    monkeypatch.setenv("GPSX_ACQ_MS_MODE", mode)
    monkeypatch.setenv("GPSX_ACQ_ALGO", algo)
    e = capi.Engine(0, lab=True)
    monkeypatch.delenv("GPSX_ACQ_MS_MODE")
    monkeypatch.delenv("GPSX_ACQ_ALGO")
    try:
        prns = np.array([1, 5, 7, 14, 20, 25, 30, 31, 32, 3, 12], np.uint8)
        for n_search, n_ms, stride, win in ((1, 10, 10, (0, 2046)), (2, 3, 4, (101, 1900)), (3, 2, 2, (0, 1))):
            kw = dict(n_search=n_search, n_ms=n_ms, search_stride_blocks=stride, dopp_min_hz=-1000, dopp_step_hz=500,
                      n_dopp=5, win=win)
            peaks, keys = e.acq_grid(stream, prns, **kw)
            for s_ in range(n_search):
                blk = stream[s_ * stride:s_ * stride + n_ms]
                for p in (0, 4, 10):
                    for d in (0, 3):
                        for b in (0, 5):
                            pk, _, _ = oracle.search_job(blk, n_ms, oracle.ca_code(int(prns[p])),
                                                         float(IF_HZ - 1000 + 500 * d), b, win[0], win[1])
                            assert _peak_tuple(peaks[s_, p, d, b]) == (pk["max_val"], pk["phase"], pk["sum"], pk["avr"]), \
                                (mode, kw, s_, p, d, b)
            halves = [e.acq_grid(stream, prns, shard=(r, 2), **kw) for r in range(2)]
            assert np.array_equal(np.maximum(halves[0][1], halves[1][1]), keys)
            assert np.array_equal(halves[0][0]["max_val"] + halves[1][0]["max_val"], peaks["max_val"])
    finally:
        e.close()


def test_config_if_hz_moves_the_doppler_axis_and_the_tracking_reference(oracle, stream):
    """gpsx_config_t: the IF is a run-time value of the context (the reference compiles it in, PM/config.h:23).  A context
    whose IF is 1500 Hz higher, asked for Doppler bins / frequency offsets 1500 Hz lower, generates the same carriers: same
    triplets and accumulators as the oracle at the reference's IF.  The sample rate is structural and refused."""
    from stm32f4_sdr_gps_amd import capi
    e = capi.Engine(0)
    try:
        cfg = e.get_config()
        assert (cfg.sample_rate_hz, cfg.if_hz) == (16368000, capi.IF_HZ)
        with pytest.raises(capi.GpsxError):
            e.set_config(sample_rate_hz=16367600)
        with pytest.raises(capi.GpsxError):
            e.set_config(if_hz=0)
        e.set_config(if_hz=capi.IF_HZ + 1500)
        assert e.get_config().if_hz == capi.IF_HZ + 1500
        prns = np.array([1, 7, 19, 32], np.uint8)
        peaks, keys = e.acq_grid(stream[0], prns, dopp_min_hz=-2000 - 1500, dopp_step_hz=500, n_dopp=5)
        want = oracle.acq_grid(stream[0], 1, prns, -2000, 500, 5, 8, n_threads=4)
        for f in ("max_val", "phase", "sum", "avr"):
            assert np.array_equal(peaks[0][f], want[f]), f
        # tracking: one small (per-channel kernel, captured graph) and one large (wave kernel) batch
        for n in (8, 2304):
            rng = np.random.default_rng(n)
            st = np.zeros(n, capi.TRK_DTYPE)
            st["prn"] = (np.arange(n) % 32) + 1
            st["code_phase_fine"] = rng.uniform(0, 16368, n).astype(np.float32)
            off = rng.integers(-5000, 5000, n).astype(np.float32)
            st["if_freq_offset_hz"] = off - 1500.0
            st["if_freq_accum"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
            acc0 = st["if_freq_accum"].copy()
            iq = e.track_epl(stream[1], st)
            for c in range(0, n, max(1, n // 48)):
                w, acc = oracle.track_epl(stream[1], oracle.ca_code(int(st["prn"][c])), float(st["code_phase_fine"][c]),
                                          float(off[c]), int(acc0[c]))
                assert np.array_equal(iq[c], w) and int(st["if_freq_accum"][c]) == acc, (n, c)
        # back to the default: the captured tracking graphs of the other IF are gone, results follow
        e.set_config()
        st = np.zeros(8, capi.TRK_DTYPE)
        st["prn"] = np.arange(1, 9)
        st["if_freq_offset_hz"] = 250.0
        iq = e.track_epl(stream[1], st)
        w, _ = oracle.track_epl(stream[1], oracle.ca_code(3), 0.0, 250.0, 0)
        assert np.array_equal(iq[2], w)
    finally:
        e.close()


def test_walk_form_16_bit_running_sums_overflow_falls_back_exactly(oracle, stream, monkeypatch):
    """The walk form keeps its running sums as 16-bit records first; a cluster in which a sum outgrows them raises its flag
    and is done again by the 24-bit form launched behind it.  Ten copies of the reference simulator's noise-free block put
    7904 per block on (PRN 1, IF + 2000 Hz): 71136 after nine blocks -- that cluster must overflow and come out exact; the
    second search (signals in noise) never overflows and keeps its first result."""
    from stm32f4_sdr_gps_amd import capi
    monkeypatch.setenv("GPSX_ACQ_MS_MODE", "walk")
    monkeypatch.setenv("GPSX_ACQ_ALGO", "mx")
    e = capi.Engine(0, lab=True)
    monkeypatch.delenv("GPSX_ACQ_MS_MODE")
    monkeypatch.delenv("GPSX_ACQ_ALGO")
    try:
        clean = load("f6_config1.npz")["blocks"][0]
        blocks = np.concatenate([np.tile(clean, (10, 1)), stream[:10]])
        prns = np.array([1, 7, 19], np.uint8)
        peaks, keys = e.acq_grid(blocks, prns, n_search=2, n_ms=10, search_stride_blocks=10, dopp_min_hz=2000, dopp_step_hz=500,
                                 n_dopp=2)
        assert e.lib.gpsx_last_kernel(e.h) == b"k_acq_mx<3>"
        assert int(peaks[0, 0, 0]["max_val"].max()) > 70000          # nine-block partial sums above 2^16 on the way
        assert int(peaks[1]["max_val"].max()) < 30000
        for s_ in range(2):
            blk = blocks[10 * s_:10 * s_ + 10]
            for p, d, b in ((0, 0, 0), (0, 0, 4), (0, 1, 7), (1, 0, 2), (2, 1, 5)):
                pk, _, _ = oracle.search_job(blk, 10, oracle.ca_code(int(prns[p])), float(IF_HZ + 2000 + 500 * d), b, 0, 2046)
                assert _peak_tuple(peaks[s_, p, d, b]) == (pk["max_val"], pk["phase"], pk["sum"], pk["avr"]), (s_, p, d, b)
    finally:
        e.close()


def test_bind_thread_to_device_pins_to_the_gpus_numa_node():
    """gpsx_bind_thread_to_device: the calling thread's affinity becomes the GPU's local CPU list (a subset of what it had),
    or the call reports that the topology is not exposed and changes nothing."""
    import os
    from stm32f4_sdr_gps_amd import capi
    before = os.sched_getaffinity(0)
    e = capi.Engine(0)
    try:
        ok = e.bind_thread_to_device()
        after = os.sched_getaffinity(0)
        if ok:
            assert after and after <= set(range(4096)) and len(after) <= os.cpu_count()
            st = np.zeros(8, capi.TRK_DTYPE)
            st["prn"] = np.arange(1, 9)
            assert e.track_epl(np.zeros(2046, np.uint8), st).shape == (8, 6)     # the engine works from the pinned thread
        else:
            assert after == before
    finally:
        os.sched_setaffinity(0, before)
        e.close()


def test_track_epl_rejects_prns_outside_1_210_from_the_kernel(oracle, stream):
    """The PRN check of gpsx_track_epl_batch lives in the kernels (a host loop over the states is a sixth of the millisecond
    at 400 000 channels): a bad PRN makes the call fail after the step -- that channel saw the empty code --, the next call
    with good states succeeds, and the padding channels of a captured step (5 channels in a graph of 8) raise nothing."""
    from stm32f4_sdr_gps_amd import capi
    e = capi.Engine(0)
    try:
        for n in (5, 3000, 140000):          # captured graph with padding, wave kernel, two-chunk step
            st = np.zeros(n, capi.TRK_DTYPE)
            st["prn"] = (np.arange(n) % 32) + 1
            st["code_phase_fine"] = (61 * np.arange(n) % 16368).astype(np.float32)
            st["if_freq_offset_hz"] = 250.0
            good = st.copy()
            iq = e.track_epl(stream[0], st)                      # good states: no complaint, padding included
            w, _ = oracle.track_epl(stream[0], oracle.ca_code(int(good["prn"][n - 1])), float(good["code_phase_fine"][n - 1]), 250.0, 0)
            assert np.array_equal(iq[n - 1], w)
            for bad in (0, 211, -7):
                st2 = good.copy()
                st2["prn"][n // 2] = bad
                with pytest.raises(capi.GpsxError, match="prn must be 1..210"):
                    e.track_epl(stream[0], st2)
            st3 = good.copy()
            assert np.array_equal(e.track_epl(stream[0], st3), iq)   # the flag does not stick
    finally:
        e.close()


@pytest.mark.parametrize("n", [140000, 393216])
def test_track_epl_chunked_pipeline_every_chunk_boundary_vs_oracle(oracle, stream, n):
    """The three-stage copy / correlate / copy pipeline of gpsx_track_epl_batch (4 chunks from 131 072 channels, 6 from
    393 216: what bench.py's tracking ladder runs at its top counts): (a) EVERY channel's accumulators and carrier phase
    against gpsx_track_epl_batch_dev on the same states resident in HBM (one launch, no chunks); (b) against the CPU oracle
    on a sample that holds both sides (+-2 channels) of every chunk boundary, the first and last channels and a stride
    through the rest, E/L wrap cases included."""
    from stm32f4_sdr_gps_amd import capi
    e = capi.Engine(0)
    try:
        rng = np.random.default_rng(n)
        st = e.host_array((n,), capi.TRK_DTYPE)
        st["prn"] = (np.arange(n) % 32) + 1
        st["code_phase_fine"] = rng.uniform(0, 16368, n).astype(np.float32)
        st["code_phase_fine"][:10] = [0.0, 0.99, 7.5, 8.0, 15.9, 16367.9, 16368.0, 16360.0, 16352.0, 16359.99]
        st["if_freq_offset_hz"] = (-5000 + (39 * np.arange(n)) % 10000).astype(np.float32) + rng.uniform(-1, 1, n).astype(np.float32)
        st["if_freq_accum"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        before = np.array(st)
        iq = e.host_array((n, 6), np.int16)
        blk = e.host_array((2046,), np.uint8)
        blk[:] = stream[3]
        e.track_epl(blk, st, iq_out=iq)
        # (a) the unchunked device path on the same inputs
        d_blk, d_st, d_iq = e.malloc(2048), e.malloc(n * 16), e.malloc(n * 12)
        e.h2d(d_blk, np.concatenate([stream[3], np.zeros(2, np.uint8)]))
        e.h2d(d_st, before)
        rc = e.lib.gpsx_track_epl_batch_dev(e.h, d_blk, d_st, n, d_iq)
        assert rc == 0
        e.synchronize()
        st_dev, iq_dev = np.zeros(n, capi.TRK_DTYPE), np.zeros((n, 6), np.int16)
        e.d2h(st_dev, d_st)
        e.d2h(iq_dev, d_iq)
        for p in (d_blk, d_st, d_iq):
            e.free(p)
        assert np.array_equal(iq, iq_dev)
        assert np.array_equal(np.array(st), st_dev)
        # (b) the oracle on the chunk boundaries (the call's own chunking rule: gpsx_api.hip) and a stride
        chunks = 6 if n >= 393216 else 4
        per = ((n + chunks - 1) // chunks + 3) & ~3
        sample = set(range(12)) | set(range(n - 4, n)) | set(range(13, n, n // 97))
        for c in range(1, chunks):
            sample |= {c * per + d for d in (-2, -1, 0, 1, 2) if 0 <= c * per + d < n}
        codes = {p: oracle.ca_code(p) for p in range(1, 33)}
        for c in sorted(sample):
            want, acc = oracle.track_epl(stream[3], codes[int(before["prn"][c])], float(before["code_phase_fine"][c]),
                                         float(before["if_freq_offset_hz"][c]), int(before["if_freq_accum"][c]))
            assert np.array_equal(iq[c], want), c
            assert int(st["if_freq_accum"][c]) == acc, c
    finally:
        e.close()


@pytest.mark.parametrize("n,chunks", [(5, 1), (4099, 3), (70001, 4), (150000, 16)])
def test_track_epl_chunked_entry_delivers_pieces_in_order_vs_oracle(oracle, stream, n, chunks):
    """gpsx_track_epl_batch_chunked (what the batched host step overlaps its loops with): the callback sees contiguous
    pieces in channel order that cover every channel once; AT THE TIME OF THE CALLBACK the piece's accumulators and
    carrier phases are in the caller's arrays (copied out inside the callback) and equal the oracle's on both sides of
    every piece boundary, the ends and a stride; the channels of later pieces are not required to be there yet.  More
    pieces than channels / 4 leave empty tail pieces out.  A PRN outside 1..210 fails the whole call after the callbacks."""
    from stm32f4_sdr_gps_amd import capi
    e = capi.Engine(0)
    try:
        rng = np.random.default_rng(n)
        st = e.host_array((n,), capi.TRK_DTYPE)
        st["prn"] = (np.arange(n) % 32) + 1
        st["code_phase_fine"] = rng.uniform(0, 16368, n).astype(np.float32)
        st["code_phase_fine"][:5] = [0.0, 7.5, 16367.9, 16368.0, 16359.99]
        st["if_freq_offset_hz"] = (-5000 + (39 * np.arange(n)) % 10000).astype(np.float32)
        st["if_freq_accum"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        before = np.array(st)
        iq = e.host_array((n, 6), np.int16)
        iq[:] = 0x5555
        pieces, seen_iq, seen_acc = [], np.zeros((n, 6), np.int16), np.zeros(n, np.uint32)

        def on_chunk(first, cnt):
            pieces.append((first, cnt))
            seen_iq[first:first + cnt] = iq[first:first + cnt]
            seen_acc[first:first + cnt] = st["if_freq_accum"][first:first + cnt]

        out = e.track_epl_chunked(stream[3], st, chunks, on_chunk, iq_out=iq)
        assert out is iq
        per = ((n + chunks - 1) // chunks + 3) & ~3
        assert pieces == [(f, min(per, n - f)) for f in range(0, n, per)]
        assert np.array_equal(seen_iq, iq) and np.array_equal(seen_acc, st["if_freq_accum"])
        sample = set(range(min(12, n))) | set(range(max(0, n - 4), n)) | set(range(0, n, max(1, n // 61)))
        for f, _ in pieces[1:]:
            sample |= {f + d for d in (-2, -1, 0, 1) if 0 <= f + d < n}
        codes = {p: oracle.ca_code(p) for p in range(1, 33)}
        for c in sorted(sample):
            want, acc = oracle.track_epl(stream[3], codes[int(before["prn"][c])], float(before["code_phase_fine"][c]),
                                         float(before["if_freq_offset_hz"][c]), int(before["if_freq_accum"][c]))
            assert np.array_equal(seen_iq[c], want), c
            assert int(seen_acc[c]) == acc, c
        # argument checks and the PRN verdict
        for bad_chunks in (0, 17):
            with pytest.raises(capi.GpsxError, match="n_chunks"):
                e.track_epl_chunked(stream[3], st, bad_chunks, on_chunk)
        st2 = e.host_array((n,), capi.TRK_DTYPE)
        st2[:] = before
        st2["prn"][n - 1] = 211
        calls = []
        with pytest.raises(capi.GpsxError, match="prn must be 1..210"):
            e.track_epl_chunked(stream[3], st2, chunks, lambda f, c: calls.append(f))
        assert len(calls) == len(pieces)
        st2[:] = before
        assert np.array_equal(e.track_epl(stream[3], st2), np.array(iq))   # ... and the one-call form agrees on every channel
    finally:
        e.close()


def test_track_epl_wave_form_channel_counts_and_channels_per_wave(oracle, stream, eng_wg):
    """k_track_epl_wave serves every channel count, 1 to 16 channels per wave depending on the launch size (lanes 4 c + k carry
    channel c's values): a single channel, counts below one workgroup, and counts that give 1, 2, 5 and 16 channels per
    wave, none a multiple of the workgroup's share -- the last wave is ragged, the last workgroup has idle waves -- against
    the oracle on a sample and, channel by channel, against the same states in another order (a channel's result must not
    depend on its position in a wave)."""
    from stm32f4_sdr_gps_amd import capi
    e = capi.Engine(0)
    try:
        codes = {p: oracle.ca_code(p) for p in range(1, 33)}
        for n in (1, 3, 6, 255, 2049, 8197, 20491, 70003):
            rng = np.random.default_rng(n)
            st = np.zeros(n, capi.TRK_DTYPE)
            st["prn"] = rng.integers(1, 33, n)
            st["code_phase_fine"] = rng.uniform(0, 16368, n).astype(np.float32)
            edge = [0.0, 0.99, 7.5, 8.0, 15.9, 16367.9, 16368.0, 16360.0, 16352.0, 16359.99, -1.0, -9.0, -20000.0, 20000.0]
            st["code_phase_fine"][:min(14, n)] = edge[:min(14, n)]
            st["if_freq_offset_hz"] = rng.uniform(-7000, 7000, n).astype(np.float32)
            st["if_freq_accum"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
            before = st.copy()
            iq = e.track_epl(stream[1], st)
            perm = rng.permutation(n)
            st2 = before[perm].copy()
            iq2 = e.track_epl(stream[1], st2)
            assert np.array_equal(iq2, iq[perm]) and np.array_equal(st2, st[perm]), n
            for c in [c for c in range(10) if c < n] + ([int(c) for c in rng.integers(14, n, 120)] if n > 14 else []) + [n - 1, max(n - 2, 0)]:
                want, acc = oracle.track_epl(stream[1], codes[int(before["prn"][c])], float(before["code_phase_fine"][c]),
                                             float(before["if_freq_offset_hz"][c]), int(before["if_freq_accum"][c]))
                assert np.array_equal(iq[c], want) and int(st["if_freq_accum"][c]) == acc, (n, c)
            # phases outside [0, 16368) (channels 10..13): the workgroup-per-channel kernel is the reference for what "kept in
            # range" means there
            small = before[:256].copy()
            iq_small = eng_wg.track_epl(stream[1], small)
            assert np.array_equal(iq_small, iq[:256][:len(small)]) and np.array_equal(small, st[:256]), n
    finally:
        e.close()


@pytest.mark.parametrize("n", [13, 16, 70])
def test_last_partly_filled_round_of_a_launch_goes_to_the_split_form(n, oracle, monkeypatch):
    """k_acq_mx<0> runs one workgroup per (capture, Doppler) cluster and CU: a launch is rounds of 256 clusters.  When the last
    round fills at most half the chip, the full rounds go out as they are and the leftover clusters in the split form
    (k_acq_mx<5>: 2, 4 or 8 workgroups per cluster, each started directly at its own sample offset, results merged through two
    planes and converted for those clusters only).  13 captures = 273 clusters = 256 + 17 x 8, the tail beginning in the
    middle of capture 12's Doppler bins; 16 = 336 = 256 + 80 x 2; 70 = 1470 = 5 x 256 + 190: more than half a round, no tail.
    Every capture must equal its launch on a context that never splits, byte for byte; the captures around the seam
    also equal the oracle."""
    from stm32f4_sdr_gps_amd import capi, synth
    e = capi.Engine(0)
    monkeypatch.setenv("GPSX_ACQ_NO_SPLIT", "1")
    plain = capi.Engine(0, lab=True)
    monkeypatch.delenv("GPSX_ACQ_NO_SPLIT")
    try:
        assert e.device_info()[1] == 256
        blocks = synth.cold_start_block(n, seed=23, amp_scale=0.5)
        prns = np.arange(1, 33, dtype=np.uint8)
        kw = dict(dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
        pk, keys = e.acq_grid(blocks, prns, n_search=n, **kw)
        assert e.lib.gpsx_last_kernel(e.h) == b"k_acq_mx<0>"
        pk0, keys0 = plain.acq_grid(blocks, prns, n_search=n, **kw)
        assert pk.tobytes() == pk0.tobytes() and np.array_equal(keys, keys0)
        seam = {13: (11, 12), 16: (12, 13, 15), 70: (69,)}[n]
        for i in seam:
            want = oracle.acq_grid(blocks[i:i + 1], 1, prns, -5000, 500, 21, 8, n_threads=ORC_THREADS)
            for f in ("max_val", "phase", "sum", "avr"):
                assert np.array_equal(pk[i][f], want[f]), (i, f)
        # windows reach the tail too
        pkw, _ = e.acq_grid(blocks, prns, n_search=n, win=(300, 1700), **kw)
        pkw0, _ = plain.acq_grid(blocks, prns, n_search=n, win=(300, 1700), **kw)
        assert pkw.tobytes() == pkw0.tobytes()
    finally:
        e.close()
        plain.close()


def test_chunk_callback_may_not_reenter_its_context_and_its_exceptions_reach_the_caller(oracle, stream):
    """gpsx_track_epl_batch_chunked calls back while later pieces are in flight on the context's arena and side streams: an
    entry point of the SAME context called from inside the callback is refused (GPSX_EINVAL), another context works; and
    an exception raised in a Python callback is not swallowed by ctypes but re-raised by Engine.track_epl_chunked."""
    from stm32f4_sdr_gps_amd import capi
    e, other = capi.Engine(0), capi.Engine(0)
    try:
        n = 70001
        st = np.zeros(n, capi.TRK_DTYPE)
        st["prn"] = (np.arange(n) % 32) + 1
        st["code_phase_fine"] = (61 * np.arange(n) % 16368).astype(np.float32)
        small = np.zeros(8, capi.TRK_DTYPE)
        small["prn"] = np.arange(1, 9)
        seen = []

        def cb(first, cnt):
            try:
                e.track_epl(stream[1], small.copy())
                seen.append("same context accepted")
            except capi.GpsxError as exc:
                seen.append(str(exc))
            iq = other.track_epl(stream[1], small.copy())        # another context is free
            w, _ = oracle.track_epl(stream[1], oracle.ca_code(3), 0.0, 0.0, 0)
            assert np.array_equal(iq[2], w)

        iq = e.track_epl_chunked(stream[1], st, 4, cb)
        assert len(seen) == 4 and all("callback" in s_ for s_ in seen), seen
        w, _ = oracle.track_epl(stream[1], oracle.ca_code(int(st["prn"][n - 1])), float(st["code_phase_fine"][n - 1]), 0.0, 0)
        assert np.array_equal(iq[n - 1], w)
        e.track_epl(stream[1], small.copy())                       # and the context is usable again afterwards

        def boom(first, cnt):
            raise KeyError("from the callback")

        with pytest.raises(KeyError):
            e.track_epl_chunked(stream[1], st, 4, boom)
    finally:
        e.close()
        other.close()


def test_the_product_library_ignores_the_lab_knobs(stream, monkeypatch):
    """lib/libgpsx.so with $GPSX_ACQ_ALGO / $GPSX_ACQ_NO_SPLIT / $GPSX_ACQ_MS_MODE set runs what it runs without them; the public
    selector gpsx_set_acq_path is how a host chooses the vector-ALU path, and both paths give the same bytes."""
    from stm32f4_sdr_gps_amd import capi
    prns = np.arange(1, 33, dtype=np.uint8)
    kw = dict(n_search=2, dopp_min_hz=-1000, dopp_step_hz=500, n_dopp=5)
    base = capi.Engine(0)
    monkeypatch.setenv("GPSX_ACQ_ALGO", "dot8")
    monkeypatch.setenv("GPSX_ACQ_NO_SPLIT", "1")
    monkeypatch.setenv("GPSX_ACQ_MS_MODE", "blocks")
    e = capi.Engine(0)
    try:
        assert e.lib.gpsx_is_lab_build() == 0
        pk0, k0 = base.acq_grid(stream[:2], prns, **kw)
        ran0 = base.lib.gpsx_last_kernel(base.h)
        pk, k = e.acq_grid(stream[:2], prns, **kw)
        assert e.lib.gpsx_last_kernel(e.h) == ran0 and ran0.startswith(b"k_acq_mx")
        assert pk.tobytes() == pk0.tobytes() and np.array_equal(k, k0)
        e.set_acq_path(capi.ACQ_PATH_VECTOR)
        pk_v, k_v = e.acq_grid(stream[:2], prns, **kw)
        assert e.lib.gpsx_last_kernel(e.h).startswith(b"k_acq_poly")
        assert pk_v.tobytes() == pk0.tobytes() and np.array_equal(k_v, k0)
        e.set_acq_path(capi.ACQ_PATH_MATRIX)
        e.acq_grid(stream[:2], prns, **kw)
        assert e.lib.gpsx_last_kernel(e.h) == ran0
    finally:
        e.close()
        base.close()
