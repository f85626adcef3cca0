"""The weighted two-bit acquisition grid (EXTENSION, not in the reference: include/gpsx.h gpsx_acq_grid_weighted,
csrc/k_acq_weighted.hip) against its own oracle (oracle/gpsx_oracle.c orc_acq_grid_weighted, pinned to the sample-by-sample
definition in tests/test_oracle_weighted.py): every (search, PRN, Doppler) record, in both weight modes, on PRN lists that do and
do not fill the kernel's groups of eight; that it leaves the one-bit path alone; and what the mode is for -- on the bench's own
sub-noise satellites the two-bit grid puts more captures' peaks on the true (Doppler bin, code phase) than the sign-only grid."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from stm32f4_sdr_gps_amd import capi
    e = capi.Engine(0)
    yield e
    e.close()


def _blocks(n_ms, amp, seed):
    from stm32f4_sdr_gps_amd import synth
    sats = [synth.Sat(7, 1310.0, 4321.0, amp, 0.4), synth.Sat(19, -2240.0, 12007.0, amp, 2.0), synth.Sat(30, 2018.0, 13000.0, amp, 4.0)]
    return synth.make_if_static(n_ms, sats, noise_amp=1.0, seed=seed, two_bit=True)


def _path(eng, path):
    from stm32f4_sdr_gps_amd import capi
    eng.set_acq_path(capi.ACQ_PATH_MATRIX if path == "matrix" else capi.ACQ_PATH_VECTOR)
    return b"k_acq_mxw" if path == "matrix" else b"k_acq_weighted"


@pytest.mark.parametrize("path", ["matrix", "vector"])
@pytest.mark.parametrize("use_mag", [True, False])
def test_weighted_grid_matches_its_oracle(eng, oracle, use_mag, path):
    from conftest import oracle_threads
    blocks = _blocks(5, 0.3, 3)
    kernel = _path(eng, path)
    try:
        for prns, kw in ((np.array([7, 19, 30, 1, 2, 3, 4, 5, 6, 8, 9], np.uint8), dict(n_search=2, dopp_min_hz=-2500, dopp_step_hz=500, n_dopp=3, stride_blocks=2)),
                         (np.array([19], np.uint8), dict(n_search=3, dopp_min_hz=-2240, dopp_step_hz=250, n_dopp=1, stride_blocks=1)),
                         (np.arange(1, 17, dtype=np.uint8), dict(n_search=1, dopp_min_hz=1000, dopp_step_hz=500, n_dopp=2, stride_blocks=1)),
                         # more than one set of 32 PRN slots (the matrix form's cluster), the second one not full
                         (np.arange(1, 41, dtype=np.uint8), dict(n_search=1, dopp_min_hz=1310, dopp_step_hz=500, n_dopp=1, stride_blocks=1))):
            got = eng.acq_grid_weighted(blocks, prns, use_magnitude=use_mag, **kw)
            assert eng.lib.gpsx_last_kernel(eng.h) == kernel
            want = oracle.acq_grid_weighted(blocks, kw["n_search"], prns, kw["dopp_min_hz"], kw["dopp_step_hz"], kw["n_dopp"], use_mag,
                                            stride_blocks=kw["stride_blocks"], n_threads=oracle_threads())
            for f in ("max_val", "phase", "sum", "avr"):
                assert np.array_equal(got[f], want[f]), (len(prns), f, np.argwhere(got[f] != want[f])[:4].tolist())
    finally:
        _path(eng, "matrix")


@pytest.mark.parametrize("path", ["matrix", "vector"])
def test_weighted_grid_on_strong_and_degenerate_captures(eng, oracle, path):
    """Magnitudes near the top of the range (a clean strong satellite: |I| in the tens of thousands, the exact-root path), an
    all-one-value capture (every window the same: the recurrence vectors are all zero but at the unmixed samples), and a capture
    whose magnitude bit is always set."""
    from conftest import oracle_threads
    from stm32f4_sdr_gps_amd import synth
    strong = synth.make_if_static(1, [synth.Sat(7, 1310.0, 4321.0, 4.0, 0.4)], noise_amp=0.05, seed=5, two_bit=True)
    flat = np.full_like(strong, 0xFF)                      # sign 1, magnitude 1 everywhere
    sign_only = strong | np.uint8(0xAA)                    # magnitude bit forced to 1: weights +-3
    kernel = _path(eng, path)
    try:
        for blocks in (strong, flat, sign_only):
            prns = np.array([7, 8], np.uint8)
            got = eng.acq_grid_weighted(blocks, prns, 1, 810, 500, 2, use_magnitude=True)
            assert eng.lib.gpsx_last_kernel(eng.h) == kernel
            want = oracle.acq_grid_weighted(blocks, 1, prns, 810, 500, 2, True, n_threads=oracle_threads())
            for f in ("max_val", "phase", "sum", "avr"):
                assert np.array_equal(got[f], want[f]), (f, got[f].tolist(), want[f].tolist())
        assert got["max_val"].max() > 2896                 # (the last capture: the exact-root path did run)
    finally:
        _path(eng, "matrix")


def test_weighted_grid_over_the_root_s_input_range(eng, oracle):
    """The matrix-core kernel's fast root (floor(sqrt(E)) from the hardware root of E + 2, E < 2^24) and its exact fallback share
    a border at |I|, |Q| = 2896: satellites from far below the noise to far above it put peak magnitudes on both sides of it and
    side-lobe magnitudes all over the fast path's range -- every sum has to match the oracle's (one root off by one shows)."""
    from conftest import oracle_threads
    from stm32f4_sdr_gps_amd import synth
    prns = np.array([7, 19, 30, 2], np.uint8)
    tops = []
    for amp in (0.02, 0.05, 0.1, 0.2, 0.4, 0.8, 1.6):
        sats = [synth.Sat(7, 1310.0, 4321.0, amp, 0.4), synth.Sat(19, 1290.0, 12007.0, amp * 0.7, 2.0), synth.Sat(30, 1350.0, 13000.0, amp * 0.4, 4.0)]
        blocks = synth.make_if_static(1, sats, noise_amp=1.0, seed=int(amp * 1000), two_bit=True)
        got = eng.acq_grid_weighted(blocks, prns, 1, 1310, 40, 2, use_magnitude=True)
        assert eng.lib.gpsx_last_kernel(eng.h) == b"k_acq_mxw"
        want = oracle.acq_grid_weighted(blocks, 1, prns, 1310, 40, 2, True, n_threads=oracle_threads())
        for f in ("max_val", "phase", "sum", "avr"):
            assert np.array_equal(got[f], want[f]), (amp, f, got[f].tolist(), want[f].tolist())
        tops.append(int(got["max_val"].max()))
    assert min(tops) < 2896 < max(tops), tops


def test_weighted_mode_argument_checks_and_the_one_bit_path_is_untouched(eng, oracle):
    from stm32f4_sdr_gps_amd import capi
    blocks = _blocks(2, 0.3, 3)
    with pytest.raises(capi.GpsxError):
        eng.acq_grid_weighted(blocks, np.array([0], np.uint8), 1, 0, 500, 1)            # PRN 0
    with pytest.raises(capi.GpsxError):
        eng.acq_grid_weighted(blocks, np.array([5], np.uint8), 3, 0, 500, 1)            # three searches, two blocks
    # the reference-parity path on the same two-bit capture (sign plane) before and after a weighted call: the same bytes
    prns = np.array([7, 19], np.uint8)
    eng.set_if_format(capi.IF_2BIT_SM)
    try:
        pk0, k0 = eng.acq_grid(blocks[:1], prns, n_search=1, dopp_min_hz=1000, dopp_step_hz=500, n_dopp=2)
        eng.acq_grid_weighted(blocks, prns, 2, 1000, 500, 2)
        pk1, k1 = eng.acq_grid(blocks[:1], prns, n_search=1, dopp_min_hz=1000, dopp_step_hz=500, n_dopp=2)
    finally:
        eng.set_if_format(capi.IF_1BIT)
    assert pk0.tobytes() == pk1.tobytes() and np.array_equal(k0, k1)


def test_two_bits_acquire_more_sub_noise_captures_than_one(eng):
    """96 captures of the bench's six satellites at amplitude scale 0.1 (the bench's 0.25 is acquired by every capture in either
    mode: tools/experiments/weighted_gain_sweep.py has the sweep -- 576 / 572 pairs at 0.2, 560 / 368 at 0.12, 497 / 212 at 0.1,
    112 / 8 at 0.06).  A capture "acquires" a
    satellite when the grid's best (Doppler bin, phase) of that PRN is the true bin (or its neighbour on the other side of the
    true Doppler) and within 8 samples of the true code phase.  Same correlator, same captures: with the magnitude bit more
    (capture, satellite) pairs acquire than on the sign plane alone, and the true cells' peak-to-mean ratio is higher."""
    from stm32f4_sdr_gps_amd import synth
    n = 96
    blocks = synth.cold_start_block(n, seed=11, amp_scale=0.1, two_bit=True)
    truth = {3: (-3210.0, 777.0), 5: (912.5, 1600.0), 11: (4480.0, 12001.0), 14: (4037.0, 4000.0), 20: (-1025.0, 9000.0), 30: (2018.0, 13000.0)}
    prns = np.array(sorted(truth), np.uint8)
    acquired, ratio = {}, {}
    for use_mag in (True, False):
        pk = eng.acq_grid_weighted(blocks, prns, n, -5000, 500, 21, use_magnitude=use_mag)
        hits, ratios = 0, []
        for i, p in enumerate(prns):
            dopp, delay = truth[int(p)]
            best_bin = pk[:, i, :]["max_val"].argmax(axis=1)                       # per capture: the PRN's strongest Doppler bin
            best = pk[np.arange(n), i, best_bin]
            bin_ok = np.abs(-5000 + 500 * best_bin - dopp) <= 500
            phase_ok = np.abs((best["phase"].astype(int) - delay + 8184) % 16368 - 8184) <= 8
            hits += int((bin_ok & phase_ok).sum())
            true_bin = int(round((dopp + 5000) / 500))
            cell = pk[:, i, true_bin]
            ratios.append(float((cell["max_val"] / np.maximum(cell["avr"], 1)).mean()))
        acquired[use_mag], ratio[use_mag] = hits, float(np.mean(ratios))
    print("acquired (capture, satellite) pairs of", n * len(prns), ": two-bit", acquired[True], "sign only", acquired[False],
          "| peak / mean of the true cells: two-bit", round(ratio[True], 2), "sign only", round(ratio[False], 2))
    assert acquired[True] > 1.5 * acquired[False] and ratio[True] > 1.03 * ratio[False]
