"""Index arithmetic of the multi-GPU acquisition sweep (host side; mirrors the grid kernels and k_acq_keys in csrc/).

A work UNIT is one (search, Doppler bin, group of 8 PRNs): unit = (search * n_dopp + dopp_idx) * n_groups + prn_idx // 8
(84 units for one 32 PRN x 21 Doppler search).  Rank r of `world` computes the contiguous run of units
[r * U // world, (r + 1) * U // world): balanced to within one unit (11 or 10 units per rank for one search on 8 GPUs),
and the four 8-PRN groups of a (search, Doppler) pair stay on one GPU wherever the run boundaries allow -- the
matrix-core grid kernel sweeps 32 PRNs per workgroup.  All 8 replica bit shifts of a unit stay together, so the best
fine phase of a (search, PRN, Doppler) pair is decided locally.  Every rank fills its entries of a zero-initialised int64
table key[search, prn, dopp] = (energy << 14) | (16383 - fine_phase); ONE all-reduce(MAX) merges the ranks.  The complement
makes ties resolve to the lowest fine phase, like correlation_search's strict '>' (PM/GPS/gps_misc.c:170).
"""
from __future__ import annotations

import numpy as np

GROUP = 8


def n_groups(n_prn: int) -> int:
    return (n_prn + GROUP - 1) // GROUP


def unit_table(n_search: int, n_prn: int, n_dopp: int) -> np.ndarray:
    """unit index of every (search, prn_idx, dopp_idx)."""
    s = np.arange(n_search)[:, None, None]
    p = np.arange(n_prn)[None, :, None]
    d = np.arange(n_dopp)[None, None, :]
    return (s * n_dopp + d) * n_groups(n_prn) + p // GROUP


def unit_run(n_units: int, rank: int, world: int):
    """[lo, hi) of the units rank `rank` computes."""
    return n_units * rank // world, n_units * (rank + 1) // world


def owned_mask(n_search: int, n_prn: int, n_dopp: int, rank: int, world: int) -> np.ndarray:
    lo, hi = unit_run(n_search * n_dopp * n_groups(n_prn), rank, world)
    u = unit_table(n_search, n_prn, n_dopp)
    return (u >= lo) & (u < hi)


def pack_keys(max_val: np.ndarray, phase: np.ndarray) -> np.ndarray:
    """max_val, phase: [..., n_bits] per replica bit shift -> int64 keys [...] maximised over the bit shifts."""
    n_bits = max_val.shape[-1]
    fine = 8 * phase.astype(np.int64) + np.arange(n_bits, dtype=np.int64)
    return ((max_val.astype(np.int64) << 14) | (16383 - fine)).max(axis=-1)


def unpack_keys(keys: np.ndarray):
    """-> (energy, fine_phase)"""
    return keys >> 14, 16383 - (keys & 16383)
