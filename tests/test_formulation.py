"""CPU model of the formulation the HIP acquisition kernel uses (DESIGN.md "acq_grid kernel"), checked against the
oracle for every byte offset and replica bit shift.  This is a test of the MATH, in numpy; the kernel itself is
checked on the GPU by tests/test_gpu_*.py.

    cnt(o, b) = C0(8o + b)                                  pure circular, sample-granular correlation
              + [chip1022] * (2 * pop(D[8o .. 8o+b)) - b)   non-circular replica shift (quirk Q5)
              - [o odd] * ( w(p1) + [p1 != 1022] * w(1022) ) words the reference skips at odd offsets (quirk Q3)

    C0(16q + t0) = sum_c | S'_t0[q + c] - B[c] |,   S'_t0[k] = 1 + popcount(D[16k + t0 .. +16)),  B[c] = 17 if chip else 1
"""
import numpy as np
import pytest

from golden_util import IF_HZ, load

N = 16368


def _bits(words16):
    return np.unpackbits(np.ascontiguousarray(words16[:1023]).view(np.uint8), bitorder="little").astype(np.int64)


def model_counts(d_words, chips, b):
    """d_words: 1024 u16 wiped data (word 1022 == 0).  Returns cnt[2046] for replica shift b."""
    D = _bits(d_words)                               # 16368 samples
    Dd = np.concatenate([D, D, D[:64]])
    csum = np.concatenate([[0], np.cumsum(Dd)])
    S = (csum[16:16 + 2 * N] - csum[:2 * N])         # S[t] = ones in [t, t+16)
    B = np.where(chips > 0, 17, 1).astype(np.int64)
    c1022, c1021 = int(chips[1022]), int(chips[1021])
    low = (1 << b) - 1
    high = (0xFFFF << b) & 0xFFFF
    dbytes = np.packbits(D.astype(np.uint8), bitorder="little").astype(np.int64)   # 2046 bytes
    cnt = np.zeros(2046, np.int64)
    for t0idx in (0, 1):
        t0 = b + 8 * t0idx
        Sp = 1 + S[t0::16][:2046 + 8]                # S'_t0[k], doubled
        win = np.lib.stride_tricks.sliding_window_view(Sp[:2045], 1023)   # win[q, c] = S'[q + c]
        A = np.abs(win - B[None, :]).sum(axis=1)     # A[q], q = 0..1022
        for q in range(1023):
            o = 2 * q + t0idx
            v = int(A[q])
            if c1022 and b:
                v += 2 * bin(int(dbytes[o]) & low).count("1") - b
            if t0idx:
                p1 = 1022 - q
                cm1 = int(chips[p1 - 1]) if p1 >= 1 else 0
                R = (cm1 * low) | (int(chips[p1]) * high)
                v -= bin(((int(dbytes[0]) << 8) & 0xFFFF) ^ R).count("1")
                if q != 0:
                    R2 = (c1021 * low) | (c1022 * high)
                    v -= bin((int(dbytes[o - 2]) | (int(dbytes[o - 1]) << 8)) ^ R2).count("1")
            cnt[o] = v
    return cnt


@pytest.mark.parametrize("prn,dopp", [(5, 900), (14, 4000), (1, -5000)])
def test_sad_formulation_equals_oracle(oracle, prn, dopp):
    g = load("f4_corr.npz")
    blk = g["stream"][int(g["block_index"])]
    chips = oracle.ca_code(prn)
    di, dq, _ = oracle.wipeoff(blk, float(IF_HZ + dopp))
    case = [tuple(c) for c in g["full_cases"].tolist()].index((prn, dopp))
    for b in range(8):
        ci = model_counts(di, chips, b)
        cq = model_counts(dq, chips, b)
        assert np.array_equal(ci, g["cnt_i"][case, b, :2046].astype(np.int64)), b
        assert np.array_equal(cq, g["cnt_q"][case, b, :2046].astype(np.int64)), b


def test_sad_formulation_codes_with_both_end_chips(oracle):
    # chip 1022 / 1021 / 0 values drive the corrections: make sure both polarities are exercised
    seen = set()
    for prn in range(1, 33):
        c = oracle.ca_code(prn)
        seen.add((int(c[1021]), int(c[1022])))
    assert len(seen) == 4
    g = load("f4_corr.npz")
    blk = g["stream"][2]
    for prn in (2, 4, 6, 16):   # cover the four (c1021, c1022) combinations
        chips = oracle.ca_code(prn)
        di, dq, _ = oracle.wipeoff(blk, float(IF_HZ - 2500))
        for b in (0, 1, 7):
            rep = oracle.replica(chips, b)
            want = np.array([oracle.mult_and_summ(di, dq, rep, o) for o in range(2046)], np.int64)
            assert np.array_equal(model_counts(di, chips, b), want[:, 0])
            assert np.array_equal(model_counts(dq, chips, b), want[:, 1])


def mx_byte_phase_counts(d_words, chips):
    """The byte-phase grid (replica bit shift 0) as k_acq_mx<4> forms it on the matrix cores (csrc/k_acq_mx.hip, mx_byte_pipe):
    both sample offsets started directly from their block sums S_t0[k] = pop(D[16 k + t0, +16)), as ONE Toeplitz product each,
        even o = 2 q:      cnt - 8184 = pop(D) + 8 + sum_c chip[c] (-2 S_0[(q + c) mod 1023])
        odd  o = 2 q + 1:  cnt - 8184 = pop(D) + 8 - pop(W) + sum_c chip[c] (-2 E[q + c]) + [q > 0] (sgn S_8[q - 1] - 16 chip[1022])
    with E[i] = S_8[i mod 1023] except E[1022] = 8 (first period only: the wrap word's term -chip[1022 - q] (16 - 2 pop(W)) is that
    entry lowered by 16 - 2 pop(W), and pop(W) = pop(D[0, 8)) IS S_8[1022]), sgn = 2 chip[1022] - 1 (the tail word P = D[16 (q - 1)
    + 8, +16) acts as one more chip, "chip -1" = chip 1022) -- no per-hypothesis correction left for the vector ALU."""
    D = _bits(d_words)                                # 16368 samples, the last 16 (word 1022) zero
    Dd = np.concatenate([D, D])
    chips = chips.astype(np.int64)
    ones = int(D.sum())
    S0 = np.array([Dd[16 * k:16 * k + 16].sum() for k in range(1023)])
    S8 = np.array([Dd[16 * k + 8:16 * k + 24].sum() for k in range(1023)])
    popw = int(D[:8].sum())
    assert S8[1022] == popw
    E0 = np.concatenate([S0, S0])
    E8 = np.concatenate([S8, S8])
    E8[1022] = 8
    sgn = 2 * int(chips[1022]) - 1
    cnt = np.zeros(2046, np.int64)
    for q in range(1023):
        cnt[2 * q] = 8184 + ones + 8 - 2 * int((chips * E0[q:q + 1023]).sum())
        odd = ones + 8 - popw - 2 * int((chips * E8[q:q + 1023]).sum())
        if q > 0:
            odd += sgn * int(S8[q - 1]) - 16 * int(chips[1022])
        cnt[2 * q + 1] = 8184 + odd
    return cnt


def test_matrix_core_byte_phase_formulation_equals_oracle(oracle):
    g = load("f4_corr.npz")
    for prn, dopp, blk_i in ((2, -2500, 2), (4, 1300, 0), (6, 4975, 1), (16, -6800, 3), (31, 0, 2)):   # the four (c1021, c1022) pairs
        chips = oracle.ca_code(prn)
        di, dq, _ = oracle.wipeoff(g["stream"][blk_i], float(IF_HZ + dopp))
        rep = oracle.replica(chips, 0)
        want = np.array([oracle.mult_and_summ(di, dq, rep, o) for o in range(2046)], np.int64)
        assert np.array_equal(mx_byte_phase_counts(di, chips), want[:, 0]), (prn, "I")
        assert np.array_equal(mx_byte_phase_counts(dq, chips), want[:, 1]), (prn, "Q")
