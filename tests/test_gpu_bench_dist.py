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


def test_two_ranks_sharded_sweep_equals_unsharded():
    env = dict(os.environ, GPSX_BENCH_SHARE_DEVICE="1", GPSX_BENCH_BACKEND="gloo", GPSX_BENCH_VERIFY="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--searches", "4", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "VERIFY sharded == unsharded" in res.stdout
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 1e9
    assert line["config"]["hypotheses_per_step"] == 2 * 4 * 32 * 21 * 16368
