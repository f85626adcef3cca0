"""bench.py's N > 1 data path on a one-GPU box: two ranks share device 0 and exchange over gloo (RCCL refuses two ranks
on one device); everything else -- per-rank shard descriptors, the engine launches, the single all_reduce(MAX) of the
packed key table, the timing protocol, the JSON line -- is the code the driver runs with RCCL on 2/4/8 GPUs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_ms, searches, port, ms_mode, algo", [(10, 1, 29571, "blocks", ""), (1, 4, 29572, "", ""),
                                                                (10, 4, 29573, "walk", "poly"), (10, 5, 29574, "", "")])
def test_two_ranks_sharded_sweep_equals_unsharded(n_ms, searches, port, ms_mode, algo):
    """n_ms = 10 is what the driver's N > 1 runs execute (BASELINE.json configs[3]; bench.py's default there), in both
    forms: the polyphase kernel with a workgroup per (unit, block) (what a lone search takes), the polyphase kernel walking
    its blocks, and -- what the 256-searches-per-GPU runs take -- the matrix-core kernel walking them (default dispatch,
    5 searches per rank); n_ms = 1 is the coherent grid on the matrix cores.  GPSX_BENCH_VERIFY compares the merged key table with
    an unsharded sweep of the same captures."""
    env = dict(os.environ, GPSX_BENCH_SHARE_DEVICE="1", GPSX_BENCH_BACKEND="gloo", GPSX_BENCH_VERIFY="1")
    if ms_mode:
        env["GPSX_ACQ_MS_MODE"] = ms_mode
    if algo:
        env["GPSX_ACQ_ALGO"] = algo
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--searches", str(searches), "--no-cpu-baseline"] + (["--n-ms", "1"] if n_ms == 1 else [])
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "VERIFY sharded == unsharded" in res.stdout
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 1e9
    assert line["config"]["blocks_per_search"] == n_ms
    assert line["config"]["hypotheses_per_step"] == 2 * searches * n_ms * 32 * 21 * 16368
    assert 0 < line["roofline"]["frac"] <= 1.0
    assert line["per_gpu_unsharded"]["value"] > 1e9   # one GPU's share at the same configuration, no sharding
    if n_ms > 1:
        assert line["single_search"]["ms_per_search"] > 0
