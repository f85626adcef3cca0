"""world_size-2 (and 3) multi-process test of the sharded acquisition sweep's exchange step on CPU with gloo:
each rank fills the key-table entries of the grid units it owns (computed here by the CPU oracle in place of the GPU),
ONE all_reduce(MAX) merges them, and the result equals the unsharded table.  This is bench.py's N>1 data path with
the engine swapped for the oracle; the ownership rule is the same host module the bench and the kernels follow."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    from stm32f4_sdr_gps_amd import sharding, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = pyoracle.Oracle()
    n_search, n_dopp = 2, 3
    prns = np.array([5, 14, 20, 30, 1, 2, 3, 4, 6, 7], np.uint8)     # 10 PRNs -> 2 groups of 8: both terms of the ownership rule are exercised
    blocks = synth.default_four_sv(n_search, seed=7)
    keys = np.zeros((n_search, len(prns), n_dopp), np.int64)
    mine = sharding.owned_mask(n_search, len(prns), n_dopp, rank, world)
    for s in range(n_search):
        for p in range(len(prns)):
            for d in range(n_dopp):
                if mine[s, p, d]:
                    pk = orc.acq_grid(blocks[s:s + 1], 1, prns[p:p + 1], 500 + 500 * d, 500, 1, 8)
                    keys[s, p, d] = sharding.pack_keys(pk["max_val"][0, 0], pk["phase"][0, 0])
    t = torch.from_numpy(keys)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the one collective of the path
    if rank == 0:
        np.save(os.path.join(out_dir, f"keys_w{world}.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sweep_merges_to_unsharded_table(world, tmp_path, oracle):
    import torch.multiprocessing as mp
    from stm32f4_sdr_gps_amd import sharding, synth
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    merged = np.load(tmp_path / f"keys_w{world}.npy")
    prns = np.array([5, 14, 20, 30, 1, 2, 3, 4, 6, 7], np.uint8)
    blocks = synth.default_four_sv(2, seed=7)
    want = np.zeros_like(merged)
    for s in range(2):
        pk = oracle.acq_grid(blocks[s:s + 1], 1, prns, 500, 500, 3, 8, n_threads=4)
        want[s] = sharding.pack_keys(pk["max_val"], pk["phase"])
    assert np.array_equal(merged, want)
    energy, fine = sharding.unpack_keys(merged)
    assert (energy > 0).all() and (fine < 16368).all()
