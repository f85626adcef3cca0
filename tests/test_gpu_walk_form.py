"""BASELINE.json configs[3] -- 32 PRN x 21 Doppler x 16368 phases, ten blocks summed non-coherently
(PM/GPS/acquisition.c:18, :296-311 around gps_misc.c:155-191) -- at its own shape and at the launch size `bench.py --gpus N`
runs, through the kernel those runs take: the walk form `k_acq_mx<3>` (a workgroup walks the ten blocks of its (search,
Doppler) x 32 PRNs, running sums through an HBM slice).  Every comparison here is with the CPU oracle or with a *different*
kernel form that is itself compared with the oracle; the bar is bit-exact."""
import os

import numpy as np
import pytest

from golden_util import load

pytestmark = pytest.mark.gpu

PRNS = np.arange(1, 33, dtype=np.uint8)
GRID = dict(dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21)
N_MS = 10


def _threads():
    return max(4, min(32, len(os.sched_getaffinity(0))))


def _engine(monkeypatch, ms_mode):
    """A context whose multi-block dispatch is pinned to one form ($GPSX_ACQ_MS_MODE is read at gpsx_create)."""
    from stm32f4_sdr_gps_amd import capi
    if ms_mode:
        monkeypatch.setenv("GPSX_ACQ_MS_MODE", ms_mode)
    e = capi.Engine(0, lab=bool(ms_mode))      # (the knob is the lab library's: lib/libgpsx_lab.so)
    if ms_mode:
        monkeypatch.delenv("GPSX_ACQ_MS_MODE")
    return e


def _want_keys(want):
    fine = 8 * want["phase"].astype(np.int64) + np.arange(8)[None, None, :]
    return ((want["max_val"].astype(np.int64) << 14) | (16383 - fine)).max(axis=2)


def _assert_triplets(got, want, what):
    for f in ("max_val", "phase", "sum", "avr"):
        assert np.array_equal(got[f], want[f]), (what, f)


def test_walk_form_one_full_ten_block_cold_start_search_vs_oracle(oracle, monkeypatch):
    """configs[3] as written: ONE search of the whole grid over ten blocks = 1.1e8 hypotheses, dispatch forced to the walk
    form (a lone search would take the block-parallel form): every triplet of every (PRN, Doppler, bit shift) and every
    packed key against the oracle."""
    from stm32f4_sdr_gps_amd import synth
    e = _engine(monkeypatch, "walk")
    try:
        blk = synth.cold_start_block(N_MS, seed=11, amp_scale=0.25)
        peaks, keys = e.acq_grid(blk, PRNS, n_search=1, n_ms=N_MS, search_stride_blocks=N_MS, **GRID)
        assert e.lib.gpsx_last_kernel(e.h) == b"k_acq_mx<3>"
        want = oracle.acq_grid(blk, N_MS, PRNS, -5000, 500, 21, 8, n_threads=_threads())
        _assert_triplets(peaks[0], want, "one search")
        assert np.array_equal(keys[0], _want_keys(want))
        assert int(want["max_val"].max()) > 10 * 500          # ten blocks of noise-level magnitudes were summed
    finally:
        e.close()


def test_walk_form_bench_launch_of_256_ten_block_searches(oracle, monkeypatch):
    """The launch `bench.py --gpus N` times on every rank: 256 ten-block searches of 2-bit IF (5376 walking workgroups,
    21 rounds of the chip, 1.5 MB of running-sum scratch per cluster), default dispatch.  Search 100 is replaced by ten
    copies of the reference simulator's noise-free block (7904 per block on PRN 1 at IF + 2000 Hz: 71 136 after nine
    blocks), so that the 16-bit running sums of ONE cluster overflow and the 24-bit relaunch fires inside the full
    launch.  Nine spread searches (first / last rounds of the chip, the overflowing one, its neighbours) must equal
    their lone launches -- `k_acq_mx<2>`, the block-parallel form, a different kernel, itself compared with the oracle
    in test_both_multi_block_forms_match_the_oracle -- byte for byte; three of them (one ordinary, the overflowing
    one, the last) are compared with the oracle triplet by triplet."""
    from stm32f4_sdr_gps_amd import capi, synth
    e = _engine(monkeypatch, None)
    lone = _engine(monkeypatch, "blocks")
    try:
        n = 256
        blocks2 = synth.cold_start_block(n * N_MS, seed=11, amp_scale=0.25, two_bit=True)
        clean = load("f6_config1.npz")["blocks"][0]
        clean_bits = np.unpackbits(clean, bitorder="little")
        blocks2[100 * N_MS:101 * N_MS] = synth.pack_2bit(clean_bits, np.zeros_like(clean_bits))
        for eng in (e, lone):
            eng.set_if_format(capi.IF_2BIT_SM)
        kw = dict(n_ms=N_MS, search_stride_blocks=N_MS, **GRID)
        pk, keys = e.acq_grid(blocks2, PRNS, n_search=n, **kw)
        assert e.lib.gpsx_last_kernel(e.h) == b"k_acq_mx<3>"
        assert int(pk[100, 0, 14]["max_val"].max()) > 70000      # PRN 1, bin 14 = +2000 Hz: past 16 bits on the way
        others = np.delete(np.arange(n), 100)
        assert int(pk[others]["max_val"].max()) < 30000
        for i in (0, 1, 37, 99, 100, 101, 170, 254, 255):
            pk1, keys1 = lone.acq_grid(blocks2[i * N_MS:(i + 1) * N_MS], PRNS, n_search=1, **kw)
            assert lone.lib.gpsx_last_kernel(lone.h) == b"k_acq_mx<2>"
            assert np.array_equal(pk1[0], pk[i]) and np.array_equal(keys1[0], keys[i]), i
        for i in (37, 100, 255):
            signs = np.stack([np.packbits(np.unpackbits(b, bitorder="little")[0::2], bitorder="little")
                              for b in blocks2[i * N_MS:(i + 1) * N_MS]])          # the sign plane of the same captures
            want = oracle.acq_grid(signs, N_MS, PRNS, -5000, 500, 21, 8, n_threads=_threads())
            _assert_triplets(pk[i], want, i)
            assert np.array_equal(keys[i], _want_keys(want)), i
    finally:
        e.close()
        lone.close()
