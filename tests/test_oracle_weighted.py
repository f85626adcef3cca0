"""The weighted two-bit correlation EXTENSION's oracle (oracle/gpsx_oracle.c orc_acq_grid_weighted; not in the reference, which
wires the MAX2769's sign bit only): the grid's chip-sum formulation against the definition itself, sample by sample
(orc_weighted_iq), on random hypotheses; its sign-only mode against the one-bit stream; and what the mode is for -- a
satellite below the noise comes out of the two-bit grid with a higher peak-to-mean ratio than out of the sign-only grid of
the same capture."""
import math

import numpy as np


def _stream(n_ms, amp, seed=5):
    from stm32f4_sdr_gps_amd import synth
    sats = [synth.Sat(7, 1310.0, 4321.0, amp, 0.4), synth.Sat(19, -2240.0, 12007.0, amp, 2.0)]
    return synth.make_if_static(n_ms, sats, noise_amp=1.0, seed=seed, two_bit=True), sats


def test_weighted_grid_equals_the_definition_on_random_hypotheses(oracle):
    blocks, _ = _stream(2, 0.3)
    prns = np.array([7, 19, 3], np.uint8)
    for use_mag in (True, False):
        pk = oracle.acq_grid_weighted(blocks, 2, prns, 1000, 500, 2, use_mag, n_threads=4)
        rng = np.random.default_rng(1)
        for s_ in range(2):
            for p in range(3):
                for d in range(2):
                    tau = int(pk[s_, p, d]["phase"])
                    i, q = oracle.weighted_iq(blocks[s_], int(prns[p]), 4092000 + 1000 + 500 * d, tau, use_mag)
                    assert math.isqrt(i * i + q * q) == int(pk[s_, p, d]["max_val"]), (s_, p, d)
                    # no other phase of a random sample beats it; an earlier phase does not even tie
                    for t in rng.integers(0, 16368, 12):
                        i2, q2 = oracle.weighted_iq(blocks[s_], int(prns[p]), 4092000 + 1000 + 500 * d, int(t), use_mag)
                        m2 = math.isqrt(i2 * i2 + q2 * q2)
                        assert m2 <= pk[s_, p, d]["max_val"] and not (m2 == pk[s_, p, d]["max_val"] and t < tau)
        assert (pk["avr"] == pk["sum"] // 16368).all()


def test_the_satellite_sits_at_its_delay_and_two_bits_beat_one(oracle):
    blocks, sats = _stream(3, 0.22, seed=9)
    prns = np.array([7, 19], np.uint8)
    ratio = {}
    for use_mag in (True, False):
        pk = oracle.acq_grid_weighted(blocks, 3, prns, 1000, 500, 2, use_mag, n_threads=4)   # bins 1000, 1500: PRN 7 at 1310
        hit = pk[:, 0, 1]                         # PRN 7, the 1500 Hz bin (190 Hz off: inside the 1 ms bin)
        assert (np.abs(hit["phase"].astype(int) - 4321) <= 8).all(), hit["phase"]
        ratio[use_mag] = float((hit["max_val"] / hit["avr"]).mean())
    print("peak / mean, two-bit", ratio[True], "sign only", ratio[False])
    assert ratio[True] > ratio[False] * 1.05
