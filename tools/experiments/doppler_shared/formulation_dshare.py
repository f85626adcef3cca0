"""CPU model of the Doppler-shared form of the polyphase bit-plane correlations, checked against the
oracle's wipe-off.  This is a test of the MATH and of the boundary tables, in numpy.

For sample offset j inside a chip the kernel needs, per Doppler bin f and stream s (I, Q),

    X_j(q) = sum_k cc[(k - q) mod 1023] * d_j[k],        d_j[k] = D_{f,s}(16 k + j),  k = 0 .. 1022

(k_acq_poly.hip).  The wiped sample is the raw sample XOR a carrier bit that only depends on the NCO quadrant of the
32-sample carrier word m = k >> 1 and on the bit position 16 (k & 1) + j inside the Fs/4 pattern
(PM/GPS/gps_misc.c:211-240):  d_j[k] = x_j[k] ^ sig[k].  With T[k] = cc[(k - q) mod 1023] ^ x_j[k] -- which does not
depend on the Doppler bin at all -- the Hamming distance H(q) = sum_k T[k] ^ sig[k] is a signed sum of prefix
popcounts of T taken where the carrier bit changes:

    H(q) = const + sum_b  cP_b * P(K_b) + cE_b * PE(K_b)  +  P(1023),
    P(K) = sum_{k < K} T[k],   PE(K) = the same over even k only (needed where quirk Q1 makes even and odd k differ)

and X_j(q + 1) - X_j(q) = -(H(q + 1) - H(q)) / 2 is all the recurrence M_{j+1}(q) = M_j(q) - X_j(q) + X_j(q + 1) needs.
A +-5 kHz grid has 10.5 boundaries per millisecond on average.
"""
import numpy as np
import pytest

from golden_util import IF_HZ, load

COS = (0x09999999, 0xCCCCCCCC, 0x66666666, 0x33333333)   # PM/GPS/gps_misc.c:216-217 (the 7-nibble literal is quirk Q1)
SIN = (0x33333333, 0x09999999, 0xCCCCCCCC, 0x66666666)


def step_per_word(freq_hz):
    step = int(np.float32(freq_hz) / np.float32(0.003810972))            # (uint32)(freq / 0.003810972f)
    return (step * 32) & 0xFFFFFFFF


def carrier_bits(freq_hz, stream, j):
    """sig[k], k = 0..1022: the carrier bit the reference XORs onto sample 16 k + j (0 for the 16 unmixed samples)."""
    pat = COS if stream == 0 else SIN
    step = step_per_word(freq_hz)
    sig = np.zeros(1023, np.int64)
    for k in range(1022):
        quad = ((step * (k >> 1)) & 0xFFFFFFFF) >> 30
        sig[k] = (pat[quad] >> (16 * (k & 1) + j)) & 1
    return sig


def plane_class(j):
    return 4 if j == 12 else 5 if j == 15 else j & 3        # bit positions 28 and 31 see the truncated literal


def build_entries(freq_hz, stream, j):
    """Boundary list [(K, cP, cE)] for one (Doppler, stream, plane class) and the constant term."""
    sig = carrier_bits(freq_hz, stream, j)
    se, so = sig[0:1022:2], sig[1:1022:2]                     # carrier words m = 0 .. 510
    cP = 1 - 2 * so
    cE = 2 * (so - se)
    entries = []
    for m in range(1, 511):
        dp, de = int(cP[m - 1] - cP[m]), int(cE[m - 1] - cE[m])
        if dp or de:
            entries.append((2 * m, dp, de))
    dp, de = int(cP[510]) - 1, int(cE[510])                   # k = 1022 is never mixed: carrier bit 0 from there on
    if dp or de:
        entries.append((1022, dp, de))
    return entries, int(se.sum() + so.sum())


def hamming_by_boundaries(chips, xj, entries, const):
    """H(q) for all q from Doppler-independent prefix popcounts, word by word as the kernel does it."""
    H = np.zeros(1023, np.int64)
    cc2 = np.concatenate([chips, chips]).astype(np.int64)
    for q in range(1023):
        T = cc2[1023 - q:2046 - q] ^ xj                        # T[k] = cc[(k - q) mod 1023] ^ x[k]
        words = np.zeros(32, np.uint64)
        for w in range(32):
            seg = T[32 * w:32 * w + 32]
            words[w] = int((seg << np.arange(len(seg))).sum())
        pop = lambda v: bin(int(v)).count("1")
        pre = np.concatenate([[0], np.cumsum([pop(w) for w in words])])
        pre_e = np.concatenate([[0], np.cumsum([pop(int(w) & 0x55555555) for w in words])])
        h = const + int(pre[32])
        for K, cp, ce in entries:
            w, mask = K >> 5, (1 << (K & 31)) - 1
            if cp:
                h += cp * (int(pre[w]) + pop(int(words[w]) & mask))
            if ce:
                h += ce * (int(pre_e[w]) + pop(int(words[w]) & mask & 0x55555555))
        H[q] = h
    return H


@pytest.mark.parametrize("prn,dopp,planes", [(5, 900, (0, 3, 12, 15)), (14, -5000, (1, 12, 14, 15)), (1, 4000, (2, 15))])
def test_boundary_form_equals_the_wiped_planes(oracle, prn, dopp, planes):
    g = load("f4_corr.npz")
    blk = g["stream"][int(g["block_index"])]
    chips = oracle.ca_code(prn).astype(np.int64)
    freq = float(IF_HZ + dopp)
    di, dq, _ = oracle.wipeoff(blk, freq)
    x = np.unpackbits(np.ascontiguousarray(blk), bitorder="little").astype(np.int64)
    for stream, dwords in ((0, di), (1, dq)):
        D = np.unpackbits(np.ascontiguousarray(dwords[:1023]).view(np.uint8), bitorder="little").astype(np.int64)
        for j in planes:
            xj = x[j::16][:1023].copy()
            xj[1022] = 0                                        # samples 16352.. are never mixed and read as 0
            dj = D[j::16][:1023]
            sig = carrier_bits(freq, stream, j)
            assert np.array_equal(dj, xj ^ sig), (stream, j)    # the model of the wipe-off itself
            entries, const = build_entries(freq, stream, j)
            # every boundary sits on an even k; a class-4/5 plane has even/odd terms only inside quirk-Q1 runs
            assert all(K % 2 == 0 and cp in (-2, 0, 2) and ce in (-2, 0, 2) for K, cp, ce in entries)
            if plane_class(j) < 4:
                assert all(ce == 0 for _, _, ce in entries)
            H = hamming_by_boundaries(chips, xj, entries, const)
            cc2 = np.concatenate([chips, chips])
            want = np.array([int((cc2[1023 - q:2046 - q] ^ dj).sum()) for q in range(1023)])
            assert np.array_equal(H, want), (stream, j)
            # and the quantity the recurrence uses
            X = np.array([int((cc2[1023 - q:2046 - q] & dj).sum()) for q in range(1023)])
            assert np.array_equal(np.diff(X), -np.diff(H) // 2) and not np.any(np.diff(H) % 2)


def test_boundary_counts_on_the_cold_start_grid():
    counts = []
    for d in range(21):
        f = float(IF_HZ - 5000 + 500 * d)
        n_i = len(build_entries(f, 0, 0)[0])
        n_q = len(build_entries(f, 1, 0)[0])
        counts.append(n_i + n_q)
        assert n_i + n_q <= 4 * abs(-5000 + 500 * d) // 1000 + 4
    assert 9.0 < np.mean(counts) < 13.0                          # ~10.5 sign changes per ms + the two end terms
