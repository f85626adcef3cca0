"""Differential pinning of the CPU restatement (oracle/gpsx_oracle.c) against the reference's own C, compiled in
place from /root/reference into oracle/_ref/libref_pm.so (no stand-ins: gps_misc.c + common_ram.c only).

Covers SURVEY.md 8(a) rows a1, a3-a10 and every entry of the quirk list.  Skipped where the _ref build is absent.
"""
import numpy as np
import pytest

from oracle.pyoracle import BYTES, CHIPS, IF_HZ, WORDS16


def _rng(seed):
    return np.random.default_rng(seed)


def test_ca_codes_all_210(oracle, ref_pm):
    for prn in range(1, 211):
        assert np.array_equal(oracle.ca_code(prn), ref_pm.ca_code(prn)), prn


def test_ca_code_prn_below_1_is_silent(oracle, ref_pm):
    # gps_misc.c:345 -- buffer untouched
    assert not oracle.ca_code(0).any() and not ref_pm.ca_code(0).any()


@pytest.mark.parametrize("prn", [1, 5, 32, 120])
def test_replica_all_16_shifts(oracle, ref_pm, prn):
    chips = ref_pm.ca_code(prn)
    for b in range(16):
        for pad in (0, 0xA5A5):
            a = oracle.replica(chips, b, pad)
            r = ref_pm.replica(chips, b, pad)
            assert np.array_equal(a, r), (prn, b, pad)


def test_nco_step_table(oracle):
    # per-word step is d * 4198400 for Doppler d*500 Hz; raw steps at IF and IF+900
    assert oracle.nco_step(4092000.0) == 1 << 30
    assert oracle.nco_step(4092900.0) == 1073977984
    for d in range(-14, 15):
        step32 = (oracle.nco_step(float(IF_HZ + 500 * d)) * 32) & 0xFFFFFFFF
        assert step32 == (d * 4198400) & 0xFFFFFFFF


def test_wipeoff_stateless(oracle, ref_pm):
    rng = _rng(1)
    for trial in range(40):
        sig = rng.integers(0, 256, BYTES, dtype=np.uint8)
        f = float(IF_HZ + int(rng.integers(-7000, 7001)))
        if trial % 4 == 0:
            f = float(np.float32(IF_HZ) + np.float32(rng.uniform(-7000, 7000)))
        pre = (rng.integers(0, 65536, WORDS16 + 1).astype(np.uint16), rng.integers(0, 65536, WORDS16 + 1).astype(np.uint16))
        oi, oq, _ = oracle.wipeoff(sig, f, 0, prefill=pre)
        ri, rq = ref_pm.wipeoff(sig, f, prefill=pre)
        assert np.array_equal(oi, ri) and np.array_equal(oq, rq)
        # quirk Q2: the last 16 samples (bytes 2044, 2045) are never written
        assert oi[1022] == pre[0][1022] and oq[1022] == pre[1][1022]


def test_wipeoff_stateful_and_rewind(oracle, ref_pm):
    rng = _rng(2)
    acc_o = acc_r = int(rng.integers(0, 2**32))
    for ms in range(60):
        sig = rng.integers(0, 256, BYTES, dtype=np.uint8)
        off = float(np.float32(rng.uniform(-6000, 6000)))
        oi, oq, acc_o = oracle.wipeoff(sig, float(np.float32(IF_HZ) + np.float32(off)), acc_o)
        ri, rq, acc_r = ref_pm.wipeoff_track(sig, off, acc_r)
        assert np.array_equal(oi, ri) and np.array_equal(oq, rq) and acc_o == acc_r
        if ms % 5 == 4:
            steps = int(rng.integers(0, 256))
            acc_o = oracle.rewind(off, acc_o, steps)
            acc_r = ref_pm.rewind(off, acc_r, steps)
            assert acc_o == acc_r


def test_mult_and_summ_every_offset_random_buffers(oracle, ref_pm):
    # arbitrary (non chip-shaped) buffers, all 2047 offsets incl. 2046 and the odd-offset skipping (quirk Q3)
    rng = _rng(3)
    for trial in range(3):
        di = rng.integers(0, 65536, WORDS16 + 1).astype(np.uint16)
        dq = rng.integers(0, 65536, WORDS16 + 1).astype(np.uint16)
        rep = rng.integers(0, 65536, WORDS16 + 1).astype(np.uint16)
        for o in range(0, BYTES + 1):
            assert oracle.mult_and_summ(di, dq, rep, o) == ref_pm.mult_and_summ(di, dq, rep, o), o


def test_corr8_iq_search_signal_like(oracle, ref_pm):
    rng = _rng(4)
    for prn, b in [(1, 0), (5, 3), (14, 7), (30, 12)]:
        chips = ref_pm.ca_code(prn)
        rep = ref_pm.replica(chips, b)
        sig = rng.integers(0, 256, BYTES, dtype=np.uint8)
        di, dq = ref_pm.wipeoff(sig, float(IF_HZ + 900))
        for o in list(range(0, 40)) + [1021, 1022, 1023, 1024, 2043, 2044, 2045, 2046] + list(rng.integers(0, 2046, 60)):
            o = int(o)
            assert oracle.correlation8(rep, di, dq, o) == ref_pm.correlation8(rep, di, dq, o)
            assert oracle.correlation_iq(rep, di, dq, o) == ref_pm.correlation_iq(rep, di, dq, o)
        for (a, z) in [(0, 2046), (100, 600), (2000, 2046), (7, 8), (5, 5)]:
            assert oracle.correlation_search(rep, di, dq, a, z) == ref_pm.correlation_search(rep, di, dq, a, z)


def test_search_all_zero_case(oracle, ref_pm):
    # perfectly aligned non-inverted signal -> one-sided clip gives 0 everywhere near the peak (quirk Q4);
    # an all-zero search returns phase 0 whatever the window.
    z = np.zeros(WORDS16 + 1, np.uint16)
    ones = np.full(WORDS16 + 1, 0xFFFF, np.uint16)
    assert oracle.correlation_search(z, z, z, 300, 400) == ref_pm.correlation_search(z, z, z, 300, 400)
    assert oracle.correlation_search(ones, z, z, 0, 2046) == ref_pm.correlation_search(ones, z, z, 0, 2046)


def test_mag8_matches_reference_over_sampled_plane(oracle, ref_pm):
    # drive gps_correlation8 through constructed buffers is slow; instead compare mag8 with the same expression the
    # reference evaluates, computed by numpy in float32 (sqrtf is correctly rounded on both sides).
    rng = _rng(5)
    ci = rng.integers(0, 16369, 20000)
    cq = rng.integers(0, 16369, 20000)
    i = np.clip(ci - 8184, 0, None).astype(np.int32)
    q = np.clip(cq - 8184, 0, None).astype(np.int32)
    want = np.sqrt((i * i).astype(np.float32) + (q * q).astype(np.float32)).astype(np.int16)
    got = np.array([oracle.mag8(int(a), int(b)) for a, b in zip(ci, cq)], np.int16)
    assert np.array_equal(want, got)


def test_track_epl_vs_reference_call_sequence(oracle, ref_pm):
    # tracking.c:115-138 composed from the reference primitives by hand
    rng = _rng(6)
    acc = 12345
    for k in range(50):
        prn = int(rng.integers(1, 33))
        chips = ref_pm.ca_code(prn)
        sig = rng.integers(0, 256, BYTES, dtype=np.uint8)
        fine_f = float(np.float32(rng.uniform(0, 16368))) if k > 4 else [0.0, 7.9, 8.0, 16367.5, 16368.0][k]
        off = float(np.float32(rng.uniform(-5000, 5000)))
        iq, acc_o = oracle.track_epl(sig, chips, fine_f, off, acc)
        fine = int(np.int16(np.float32(fine_f)))
        rep = ref_pm.replica(chips, fine & 7)
        di, dq, acc_r = ref_pm.wipeoff_track(sig, off, acc)
        p = (fine // 8) & 0xFFFF
        e = (p - 1) & 0xFFFF
        l = (p + 1) & 0xFFFF
        if e >= 2046:
            e = 2045
        if l >= 2046:
            l = 0
        want = []
        for o in (e, p, l):
            want += list(ref_pm.correlation_iq(rep, di, dq, o))
        assert list(iq) == want and acc_o == acc_r
        acc = acc_o
