"""The synthetic stream generator's LNAV framing (stm32f4_sdr_gps_amd/synth.py), checked against the parity masks every
GPS receiver carries (IS-GPS-200 20.3.5.2 written as six 32-bit masks over [D29* D30* D1 .. D30])."""
import numpy as np

from stm32f4_sdr_gps_amd import synth

MASKS = (0xBB1F3480, 0x5D8F9A40, 0xAEC7CD00, 0x5763E680, 0x6BB1F340, 0x8B7A89C0)


def _parity_ok(word30, d29_prev, d30_prev):
    bits = list(word30)
    if d30_prev:                                   # data bits were transmitted inverted
        bits = [b ^ 1 for b in bits[:24]] + bits[24:]
    w = (d29_prev << 31) | (d30_prev << 30)
    for i, b in enumerate(bits):
        w |= b << (29 - i)
    for k, m in enumerate(MASKS):
        if bin(w & m).count("1") & 1 != bits[24 + k]:
            return False
    return True


def test_lnav_subframes_are_parity_correct_and_framed():
    rng = np.random.Generator(np.random.PCG64(5))
    d29 = d30 = 0
    for n in range(12):
        sub_id, tow = n % 5 + 1, 1000 + n
        sf = synth.lnav_subframe(sub_id, tow, rng)
        assert len(sf) == 300 and tuple(sf[:8]) == (1, 0, 0, 0, 1, 0, 1, 1)
        for w in range(10):
            word = sf[30 * w:30 * w + 30]
            assert _parity_ok(word, d29, d30), (n, w)
            d29, d30 = word[28], word[29]
        assert sf[58:60] == [0, 0] and sf[298:300] == [0, 0]             # hand-over word and word 10 end in 00
        how = sf[30:60]                                                   # follows the TLM word, whose D30 may be 1
        src = [b ^ sf[29] for b in how[:24]]
        assert int("".join(map(str, src[:17])), 2) == tow and int("".join(map(str, src[19:22])), 2) == sub_id


def test_lnav_stream_polarity_and_offsets():
    a = synth.lnav_bits(700, 37, 2007)
    b = synth.lnav_bits(700, 0, 2007)
    assert np.array_equal(a[:600], b[37:637])
    pre = np.array([1, 0, 0, 0, 1, 0, 1, 1], np.uint8)
    assert np.array_equal(b[:8], pre) and np.array_equal(b[300:308], pre) and np.array_equal(b[600:608], pre)
