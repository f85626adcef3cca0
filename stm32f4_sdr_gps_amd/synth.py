"""Synthetic 1-bit GPS L1 C/A IF stream generator (SURVEY.md 8(d) "Synthetic input").

Host-side tooling for tests and bench.py -- not part of the correlator hot path.  The sample format is the
reference's: fs = 16.368 MHz, IF = 4.092 MHz, one sign bit per sample, packed LSB first (sample n is bit n & 7 of
byte n >> 3), 2046 bytes per millisecond (reference: PM/config.h:23-28, PM/signal_capture.c:9-11).

    s[n] = sign( sum_sv A * d * c_sv(floor((n - tau)/16) mod 1023) * cos(2 pi (IF + fd) n / fs + phi) + w[n] )

with w ~ U(-noise, noise) from a fixed-seed PCG64 stream and a continuous sample index across milliseconds.
"""
from __future__ import annotations

from dataclasses import dataclass

import os

import numpy as np

FS_HZ = 16_368_000
IF_HZ = 4_092_000
SAMPLES_PER_MS = 16368
BYTES_PER_MS = 2046
CHIPS = 1023

# IS-GPS-200 G2 phase-selector taps for PRN 1..32
_G2_TAPS = [(2, 6), (3, 7), (4, 8), (5, 9), (1, 9), (2, 10), (1, 8), (2, 9), (3, 10), (2, 3), (3, 4), (5, 6), (6, 7),
            (7, 8), (8, 9), (9, 10), (1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9), (1, 3), (4, 6), (5, 7), (6, 8),
            (7, 9), (8, 10), (1, 6), (2, 7), (3, 8), (4, 9)]


def ca_code(prn: int) -> np.ndarray:
    """C/A code chips (0/1) for PRN 1..32."""
    if not 1 <= prn <= 32:
        raise ValueError("synth.ca_code supports PRN 1..32")
    t1, t2 = _G2_TAPS[prn - 1]
    g1 = [1] * 10
    g2 = [1] * 10
    out = np.zeros(CHIPS, np.uint8)
    for i in range(CHIPS):
        out[i] = g1[9] ^ g2[t1 - 1] ^ g2[t2 - 1]
        f1 = g1[2] ^ g1[9]
        f2 = g2[1] ^ g2[2] ^ g2[5] ^ g2[7] ^ g2[8] ^ g2[9]
        g1 = [f1] + g1[:9]
        g2 = [f2] + g2[:9]
    return out


@dataclass
class Sat:
    prn: int
    doppler_hz: float
    delay_samples: float
    amp: float = 0.6
    phase_rad: float = 0.0
    nav_bits: np.ndarray | None = None  # +-1 per 20 ms, optional


def pack_2bit(sign_bits: np.ndarray, mag_bits: np.ndarray) -> np.ndarray:
    """16368 sign bits + 16368 magnitude bits -> 4092 bytes of MAX2769-style pairs (GPSX_IF_2BIT_SM: sample n in bits
    2(n & 3) (sign) and 2(n & 3) + 1 (magnitude) of byte n >> 2)."""
    pairs = np.empty(2 * len(sign_bits), np.uint8)
    pairs[0::2] = sign_bits
    pairs[1::2] = mag_bits
    return np.packbits(pairs, bitorder="little")


def read_if_file(path: str, two_bit: bool = False, max_ms: int | None = None) -> np.ndarray:
    """Raw IF capture file as the reference's replay tool streams it (PC_SpiLight/Readme.txt: `-i rec_file.bin`): a flat
    byte stream, 2046 bytes (1-bit) or 4092 bytes (2-bit) per millisecond; a trailing partial block is dropped."""
    per_ms = 4092 if two_bit else BYTES_PER_MS
    raw = np.fromfile(path, dtype=np.uint8)
    n = len(raw) // per_ms
    if max_ms is not None:
        n = min(n, max_ms)
    return raw[:n * per_ms].reshape(n, per_ms)


def _n_threads() -> int:
    return max(1, min(16, len(os.sched_getaffinity(0))))


def _ms_chunks(n_ms: int, batch: int):
    return [(m0, min(n_ms, m0 + batch)) for m0 in range(0, n_ms, batch)]


def noise_generator(seed: int, first_sample: int) -> np.random.Generator:
    """The stream's noise generator positioned at sample `first_sample`: PCG64(seed) advanced by one draw per sample (a float64
    uniform consumes exactly one 64-bit output), so that any millisecond range can be synthesised on its own -- by any thread --
    and comes out sample-identical to a sequential run from millisecond 0."""
    bg = np.random.PCG64(seed)
    if first_sample:
        bg.advance(first_sample)
    return np.random.Generator(bg)


def make_if(n_ms: int, sats: list[Sat], noise_amp: float = 1.0, seed: int = 7, start_ms: int = 0,
            two_bit: bool = False, mag_threshold: float = 0.6) -> np.ndarray:
    """Return uint8 array [n_ms, 2046] of packed 1-bit samples, or [n_ms, 4092] of sign/magnitude pairs if two_bit
    (magnitude bit = |x| > mag_threshold).  Synthesised in batches of 8 ms on the cores this process may use; the samples do not
    depend on the batching or the thread count (elementwise float64 arithmetic, noise by position: noise_generator)."""
    out = np.zeros((n_ms, 4092 if two_bit else BYTES_PER_MS), np.uint8)
    codes = {s.prn: (1.0 - 2.0 * ca_code(s.prn).astype(np.float64)) for s in sats}
    n0 = np.arange(SAMPLES_PER_MS, dtype=np.float64)

    def chunk(rng_):
        m0, m1 = rng_
        rng = noise_generator(seed, m0 * SAMPLES_PER_MS)
        base = np.array([float((start_ms + ms) * SAMPLES_PER_MS) for ms in range(m0, m1)])
        n = n0[None, :] + base[:, None]
        x = rng.uniform(-noise_amp, noise_amp, n.shape) if noise_amp > 0 else np.zeros(n.shape)
        for s in sats:
            chip = np.floor((n - s.delay_samples) / 16.0).astype(np.int64) % CHIPS
            d = 1.0
            if s.nav_bits is not None:
                bit_idx = np.floor((n - s.delay_samples) / (16.0 * CHIPS * 20)).astype(np.int64) % len(s.nav_bits)
                d = s.nav_bits[bit_idx]
            ph = 2.0 * np.pi * ((IF_HZ + s.doppler_hz) / FS_HZ) * n + s.phase_rad
            x = x + s.amp * d * codes[s.prn][chip] * np.cos(ph)
        bits = (x >= 0).astype(np.uint8)
        if two_bit:
            mag = (np.abs(x) > mag_threshold).astype(np.uint8)
            for i in range(m1 - m0):
                out[m0 + i] = pack_2bit(bits[i], mag[i])
        else:
            out[m0:m1] = np.packbits(bits, axis=1, bitorder="little")

    chunks = _ms_chunks(n_ms, 8)
    threads = _n_threads()
    if threads == 1 or len(chunks) == 1:
        for c in chunks:
            chunk(c)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(chunk, chunks))
    return out


def make_if_static(n_ms: int, sats: list[Sat], noise_amp: float = 1.0, seed: int = 7, batch_ms: int = 250,
                   two_bit: bool = False, mag_threshold: float = 0.6) -> np.ndarray:
    """make_if for MANY satellites over MANY milliseconds (BASELINE.json configs[4] as SURVEY.md 8(d) words it: 256 signals,
    10 000 ms), for satellites without navigation data.  One millisecond is exactly 1023 chips and 4092 IF cycles, so a
    satellite's samples of millisecond m are those of millisecond 0 with the carrier turned by alpha = 2 pi fd m / 1000:

        c(n) cos(theta(n) + alpha) = [c(n) cos theta(n)] cos alpha - [c(n) sin theta(n)] sin alpha

    and the whole stream is one matrix product [n_ms, 2 S] x [2 S, 16368] plus noise.  Both factors and the noise are rounded
    to integers (2^-15, 2^-15 and 2^-30 units) first: every product and partial sum is then an integer below 2^53, the float64
    product is exact in ANY summation order, and the sign bits do not depend on the BLAS, its threading or the machine.
    (Not sample-identical to make_if, which accumulates unrounded doubles: a different, equally valid stream.)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = np.arange(SAMPLES_PER_MS, dtype=np.float64)
    basis = np.zeros((2 * len(sats), SAMPLES_PER_MS))
    for k, s in enumerate(sats):
        if s.nav_bits is not None:
            raise ValueError("make_if_static: satellites without navigation data only")
        code = 1.0 - 2.0 * ca_code(s.prn).astype(np.float64)
        chip = np.floor((n - s.delay_samples) / 16.0).astype(np.int64) % CHIPS
        theta = 2.0 * np.pi * ((IF_HZ + s.doppler_hz) / FS_HZ) * n + s.phase_rad
        basis[2 * k] = np.rint(s.amp * code[chip] * np.cos(theta) * 32768.0)
        basis[2 * k + 1] = np.rint(s.amp * code[chip] * np.sin(theta) * 32768.0)
    fd = np.array([s.doppler_hz for s in sats])
    out = np.zeros((n_ms, 4092 if two_bit else BYTES_PER_MS), np.uint8)
    for m0 in range(0, n_ms, batch_ms):
        m = np.arange(m0, min(n_ms, m0 + batch_ms), dtype=np.float64)
        alpha = 2.0 * np.pi * np.outer(m, fd) / 1000.0
        turn = np.empty((len(m), 2 * len(sats)))
        turn[:, 0::2] = np.rint(np.cos(alpha) * 32768.0)
        turn[:, 1::2] = -np.rint(np.sin(alpha) * 32768.0)
        x = turn @ basis
        if noise_amp > 0:
            x += np.rint(rng.uniform(-noise_amp, noise_amp, x.shape) * float(1 << 30))
        if two_bit:      # sign / magnitude pairs as pack_2bit lays them out: sample n in bits 2 (n & 3), 2 (n & 3) + 1 of byte n >> 2
            pairs = np.empty((len(m), 2 * SAMPLES_PER_MS), bool)
            pairs[:, 0::2] = x >= 0
            pairs[:, 1::2] = np.abs(x) > mag_threshold * float(1 << 30)
            out[m0:m0 + len(m)] = np.packbits(pairs, axis=1, bitorder="little")
        else:
            out[m0:m0 + len(m)] = np.packbits(x >= 0, axis=1, bitorder="little")
    return out


def default_four_sv(n_ms: int, seed: int = 7) -> np.ndarray:
    """The reference's default 4-satellite table (PM/main.c:59-73): PRN 5/14/20/30, hints 900/4000/-1000/2000 Hz."""
    # true Dopplers sit a few tens of Hz off the hints so the carrier phase walks through all four quadrants (the
    # reference's one-sided clip, gps_misc.c:108-111, only sees one of them at a time)
    sats = [Sat(5, 912.5, 1600.0, 0.6, 0.3), Sat(14, 4037.0, 4000.0, 0.6, 1.1), Sat(20, -1025.0, 9000.0, 0.6, 2.5),
            Sat(30, 2018.0, 13000.0, 0.6, 4.0)]
    return make_if(n_ms, sats, noise_amp=1.0, seed=seed)


def four_sv_with_nav(n_ms: int, seed: int = 7) -> np.ndarray:
    """The 4-SV table with 50 bit/s navigation data on every satellite (random bits, different per satellite), for the
    step-level tests: exercises the 20 ms bit synchronisation and the PLL's second gain set."""
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    bits = [1.0 - 2.0 * rng.integers(0, 2, 400).astype(np.float64) for _ in range(4)]
    sats = [Sat(5, 912.5, 1600.0, 0.6, 0.3, bits[0]), Sat(14, 4037.0, 4000.0, 0.6, 1.1, bits[1]),
            Sat(20, -1025.0, 9000.0, 0.6, 2.5, bits[2]), Sat(30, 2018.0, 13000.0, 0.6, 4.0, bits[3])]
    return make_if(n_ms, sats, noise_amp=1.0, seed=seed)


# ---- LNAV framing (IS-GPS-200, 20.3.3 / 20.3.5): what a receiver's word layer needs to lock on -------------------------
# Parity bits D25..D30 of a word: XOR of the listed source bits d1..d24 with D29* or D30*, the previous word's last two
# transmitted bits (table 20-XIV); the 24 data bits themselves go out XORed with D30*.
_PARITY = ((29, (1, 2, 3, 5, 6, 10, 11, 12, 13, 14, 17, 18, 20, 23)), (30, (2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24)),
           (29, (1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22)), (30, (2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23)),
           (30, (1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24)), (29, (3, 5, 6, 8, 9, 10, 11, 13, 15, 19, 22, 23, 24)))
_PREAMBLE = (1, 0, 0, 0, 1, 0, 1, 1)


def lnav_word(d: list[int], d29_prev: int, d30_prev: int, solve_tail: bool = False) -> list[int]:
    """24 source bits -> 30 transmitted bits.  solve_tail: choose source bits 23, 24 (the non-information bearing bits of
    the hand-over word and of word 10) so that the word ends in D29 = D30 = 0."""
    for tail in range(4 if solve_tail else 1):
        if solve_tail:
            d = d[:22] + [tail >> 1, tail & 1]
        par = []
        for start, taps in _PARITY:
            p = d29_prev if start == 29 else d30_prev
            for t in taps:
                p ^= d[t - 1]
            par.append(p)
        if not solve_tail or (par[4] == 0 and par[5] == 0):
            return [b ^ d30_prev for b in d] + par
    raise AssertionError("no tail bits give D29 = D30 = 0")


def lnav_subframe(sub_id: int, tow_count: int, rng: np.random.Generator, payload: list[list[int]] | None = None) -> list[int]:
    """300 transmitted bits of one subframe: TLM (preamble), HOW (time of week of the next subframe, subframe ID), eight
    words of payload -- random, or the 8 x 24 source bits given (words 3..10; the last two bits of word 10 are solved for
    parity either way).  Starts from D29* = D30* = 0, which every subframe's last word guarantees."""
    def rand(n):
        return [int(b) for b in rng.integers(0, 2, n)]
    words = [list(_PREAMBLE) + rand(16),
             [(tow_count >> (16 - i)) & 1 for i in range(17)] + [0, 0] + [(sub_id >> (2 - i)) & 1 for i in range(3)] + [0, 0]]
    words += [list(w) for w in payload] if payload is not None else [rand(24) for _ in range(8)]
    out, d29, d30 = [], 0, 0
    for i, d in enumerate(words):
        w = lnav_word(d, d29, d30, solve_tail=i in (1, 9))
        out += w
        d29, d30 = w[28], w[29]
    return out


def lnav_bits(n_bits: int, first_bit: int, seed: int) -> np.ndarray:
    """0/1 navigation bits of a satellite that is first_bit bits into subframe 1 when the stream begins."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bits, tow, sub = [], 100, 1
    while len(bits) < first_bit + n_bits:
        bits += lnav_subframe(sub, tow, rng)
        tow += 1
        sub = sub % 5 + 1
    return np.array(bits[first_bit:first_bit + n_bits], np.uint8)


_lnav_streams: dict = {}


def four_sv_with_lnav(n_ms: int, seed: int = 7) -> np.ndarray:
    """The 4-SV table carrying parity-correct LNAV subframes (different payload and subframe timing per satellite;
    satellites 2 and 4 with the opposite data polarity), for the word layer: preamble search, parity, polarity
    detection, subframe assembly and time stamp.  The last stream made is kept (read-only): the GPU suite asks for the
    same 15 s several times."""
    if (n_ms, seed) in _lnav_streams:
        return _lnav_streams[(n_ms, seed)]
    n_bits = n_ms // 20 + 2
    first = (37, 161, 250, 96)
    flip = (0, 1, 0, 1)
    bits = [(1.0 - 2.0 * (lnav_bits(n_bits, first[k], seed + 2000 + k) ^ flip[k]).astype(np.float64)) for k in range(4)]
    sats = [Sat(5, 912.5, 1600.0, 0.6, 0.3, bits[0]), Sat(14, 4037.0, 4000.0, 0.6, 1.1, bits[1]),
            Sat(20, -1025.0, 9000.0, 0.6, 2.5, bits[2]), Sat(30, 2018.0, 13000.0, 0.6, 4.0, bits[3])]
    stream = make_if(n_ms, sats, noise_amp=1.0, seed=seed)
    stream.setflags(write=False)
    _lnav_streams.clear()
    _lnav_streams[(n_ms, seed)] = stream
    return stream


def cold_start_block(n_ms: int = 1, seed: int = 11, amp_scale: float = 1.0, two_bit: bool = False) -> np.ndarray:
    """Config 3/4 input: six satellites in view.  amp_scale = 1 gives the strong test signal
    (amplitudes 0.5-0.6 against U(-1, 1) noise, i.e. well above a real sky, so that single-millisecond peaks are
    unmistakable); amp_scale ~ 0.25 is closer to a live antenna (each satellite below the noise)."""
    base = [(3, -3210.0, 777.0, 0.5, 0.7), (5, 912.5, 1600.0, 0.6, 0.3), (11, 4480.0, 12001.0, 0.5, 5.1),
            (14, 4037.0, 4000.0, 0.6, 1.1), (20, -1025.0, 9000.0, 0.6, 2.5), (30, 2018.0, 13000.0, 0.6, 4.0)]
    sats = [Sat(p, f, d, a * amp_scale, ph) for (p, f, d, a, ph) in base]
    # (make_if_static: the same signal model in exact integer arithmetic, ~20 x faster than make_if -- the 8-GPU bench synthesises
    #  20 480 blocks per rank; same samples with and without two_bit: same sign plane)
    return make_if_static(n_ms, sats, noise_amp=1.0, seed=seed, two_bit=two_bit)
