"""Index arithmetic of the multi-GPU acquisition sweep (host side; mirrors k_acq / k_acq_keys in csrc/k_acq_grid.hip).

A work UNIT is one (search, group of 8 PRNs, Doppler bin): unit = (search * n_groups + prn_idx // 8) * n_dopp + dopp_idx
(84 units for one 32 PRN x 21 Doppler search: 11 or 10 per rank on 8 GPUs).
Rank r of `world` computes the units with unit % world == r (all 8 replica bit shifts of a unit stay together, so the
best fine phase of a (search, PRN, Doppler) pair is decided locally).  Every rank fills its entries of a zero-initialised
int64 table key[search, prn, dopp] = (energy << 14) | (16383 - fine_phase); ONE all-reduce(MAX) merges the ranks.  The
complement makes ties resolve to the lowest fine phase, like correlation_search's strict '>' (PM/GPS/gps_misc.c:170).
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
    return (s * n_groups(n_prn) + p // GROUP) * n_dopp + d


def owned_mask(n_search: int, n_prn: int, n_dopp: int, rank: int, world: int) -> np.ndarray:
    return unit_table(n_search, n_prn, n_dopp) % world == rank


def pack_keys(max_val: np.ndarray, phase: np.ndarray) -> np.ndarray:
    """max_val, phase: [..., n_bits] per replica bit shift -> int64 keys [...] maximised over the bit shifts."""
    n_bits = max_val.shape[-1]
    fine = 8 * phase.astype(np.int64) + np.arange(n_bits, dtype=np.int64)
    return ((max_val.astype(np.int64) << 14) | (16383 - fine)).max(axis=-1)


def unpack_keys(keys: np.ndarray):
    """-> (energy, fine_phase)"""
    return keys >> 14, 16383 - (keys & 16383)
