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


def _free_port(preferred):
    """`preferred` if nobody listens there, else any free port (a leftover rendezvous of an earlier, killed run must not fail a test)"""
    import socket
    for port in (preferred, 0):
        with socket.socket() as sock:
            try:
                sock.bind(("127.0.0.1", port))
                return sock.getsockname()[1]
            except OSError:
                continue
    return preferred


@pytest.mark.parametrize("n_ms, searches, port, ms_mode, algo", [(10, 1, 29571, "blocks", ""), (1, 4, 29572, "", ""),
                                                                (10, 4, 29573, "walk", "poly"), (10, 5, 29574, "", "")])
def test_two_ranks_sharded_sweep_equals_unsharded(n_ms, searches, port, ms_mode, algo, tmp_path, oracle):
    """n_ms = 10 is BASELINE.json configs[3] (`--n-ms 10`; the default N > 1 run reports it as `configs3_sharded` beside the
    one-block headline: test_default_multi_gpu_run_is_the_single_gpu_workload_sharded), in both
    forms: the polyphase kernel with a workgroup per (unit, block) (what a lone search takes), the polyphase kernel walking
    its blocks, and -- what the 256-searches-per-GPU runs take -- the matrix-core kernel walking them (default dispatch,
    5 searches per rank); n_ms = 1 is the coherent grid on the matrix cores.  GPSX_BENCH_VERIFY compares the merged key table with
    an unsharded sweep of the same captures; GPSX_BENCH_DUMP_KEYS hands the merged table to this test, which compares
    searches of it with the CPU oracle (one per rank's half of the searches where the table is the ten-block walk's: the
    first multi-GPU run is then also a parity run)."""
    dump = str(tmp_path / "merged_keys.npy")
    env = dict(os.environ, GPSX_BENCH_SHARE_DEVICE="1", GPSX_BENCH_BACKEND="gloo", GPSX_BENCH_VERIFY="1",
               GPSX_BENCH_DUMP_KEYS=dump)
    if ms_mode:
        env["GPSX_ACQ_MS_MODE"] = ms_mode
    if algo:
        env["GPSX_ACQ_ALGO"] = algo
    if ms_mode or algo:
        env["GPSX_USE_LAB_LIBRARY"] = "1"       # forced kernel forms: the lab build of the library reads those knobs
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port(port)), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--searches", str(searches), "--no-cpu-baseline", "--n-ms", str(n_ms)]
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
    # the merged table against the oracle: bench.py's own input (synth.cold_start_block, seed 11, amplitudes x 0.25), sign plane
    import numpy as np
    from stm32f4_sdr_gps_amd import synth
    keys = np.load(dump)
    assert keys.shape == (2 * searches, 32, 21)
    blocks = synth.cold_start_block(2 * searches * n_ms, seed=11, amp_scale=0.25)
    prns = np.arange(1, 33, dtype=np.uint8)
    threads = max(4, min(32, len(os.sched_getaffinity(0))))
    for s_ in ((1, 2 * searches - 2) if (n_ms, algo) == (10, "") else (2 * searches - 1,)):
        want = oracle.acq_grid(blocks[s_ * n_ms:(s_ + 1) * n_ms], n_ms, prns, -5000, 500, 21, 8, n_threads=threads)
        fine = 8 * want["phase"].astype(np.int64) + np.arange(8)[None, None, :]
        assert np.array_equal(keys[s_], ((want["max_val"].astype(np.int64) << 14) | (16383 - fine)).max(axis=2)), s_


@pytest.mark.parametrize("scaling, port", [("strong", 0), ("weak", 0)])
def test_plain_python_launch_two_ranks(scaling, port, tmp_path, oracle):
    """`python bench.py --gpus 2` with no launcher around it: the script starts its own two ranks under
    torch.distributed.run on a free port.  strong: the SAME six ten-block searches dealt to the two ranks (units as two
    contiguous runs); weak: six per rank.  Inside the run rank 0 compares cells of the merged table with the CPU oracle
    (`parity` in the line); here the merged table is compared with an unsharded sweep (GPSX_BENCH_VERIFY) and other cells of it
    with the oracle."""
    import numpy as np
    from stm32f4_sdr_gps_amd import synth
    dump = str(tmp_path / "merged_keys.npy")
    env = dict(os.environ, GPSX_BENCH_SHARE_DEVICE="1", GPSX_BENCH_BACKEND="gloo", GPSX_BENCH_VERIFY="1", GPSX_BENCH_DUMP_KEYS=dump)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--searches", "6",
           "--scaling", scaling, "--no-cpu-baseline", "--n-ms", "10"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "VERIFY sharded == unsharded" in res.stdout
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    n_search = 6 if scaling == "strong" else 12
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["config"]["blocks_per_search"] == 10
    assert line["config"]["hypotheses_per_step"] == n_search * 10 * 32 * 21 * 16368
    assert line["config"]["searches_per_gpu_per_step"] == n_search / 2
    assert line["parity"]["parity_checked"] is True and line["parity"]["hypotheses_checked"] == 2 * 16 * 16368 * 10
    assert line["communicator"]["rccl_ranks"] == 2 and line["single_search"]["ms_per_search"] > 0
    keys = np.load(dump)
    assert keys.shape == (n_search, 32, 21)
    s_ = n_search // 2
    blocks = synth.cold_start_block(n_search * 10, seed=11, amp_scale=0.25)[s_ * 10:(s_ + 1) * 10]
    prns = np.array([5, 14, 27], np.uint8)
    want = oracle.acq_grid(blocks, 10, prns, -5000, 10000, 2, 8, n_threads=max(4, min(32, len(os.sched_getaffinity(0)))))
    fine = 8 * want["phase"].astype(np.int64) + np.arange(8)[None, None, :]
    assert np.array_equal(keys[s_][np.ix_(prns.astype(int) - 1, [0, 20])], ((want["max_val"].astype(np.int64) << 14) | (16383 - fine)).max(axis=2))


def test_plain_python_launch_refuses_more_ranks_than_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPSX_BENCH_SHARE_DEVICE")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode != 0 and "GPU(s) visible" in res.stderr


def test_eight_ranks_on_one_device(tmp_path, oracle):
    """The driver's largest launch shape -- eight ranks -- on the one-GPU box: every rank on device 0, gloo instead of RCCL, the
    plain-`python` launch.  One ten-block search per rank (weak scaling): 8 x 84 units as eight contiguous runs of 84; and the
    strong form of ONE search (north_star's literal point): its 84 units as runs of 10 or 11.  Merged table = unsharded sweep
    (GPSX_BENCH_VERIFY), rank 0's oracle sample inside the run, a full row of the merged table against the oracle here."""
    import numpy as np
    from stm32f4_sdr_gps_amd import synth
    for scaling, searches, n_search in (("weak", 1, 8), ("strong", 1, 1)):
        dump = str(tmp_path / f"merged_{scaling}.npy")
        env = dict(os.environ, GPSX_BENCH_SHARE_DEVICE="1", GPSX_BENCH_BACKEND="gloo", GPSX_BENCH_VERIFY="1", GPSX_BENCH_DUMP_KEYS=dump)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--searches", str(searches),
               "--scaling", scaling, "--no-cpu-baseline", "--n-ms", "10"]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert res.returncode == 0, res.stderr[-2000:]
        assert "VERIFY sharded == unsharded" in res.stdout
        out_lines = [l for l in res.stdout.splitlines() if l.strip()]
        line = json.loads(out_lines[-1])                      # the JSON line is the LAST thing on stdout
        assert line["n_gpus"] == 8 and line["scaling"] == scaling and line["communicator"]["rccl_ranks"] == 8
        assert line["config"]["hypotheses_per_step"] == n_search * 10 * 32 * 21 * 16368
        assert line["parity"]["parity_checked"] is True
        keys = np.load(dump)
        assert keys.shape == (n_search, 32, 21)
        s_ = n_search - 1
        blocks = synth.cold_start_block(n_search * 10, seed=11, amp_scale=0.25)[s_ * 10:(s_ + 1) * 10]
        prns = np.array([9, 17, 25], np.uint8)                 # one PRN of each of three 8-PRN groups, a Doppler bin per rank's run
        want = oracle.acq_grid(blocks, 10, prns, -5000, 2500, 5, 8, n_threads=max(4, min(32, len(os.sched_getaffinity(0)))))
        fine = 8 * want["phase"].astype(np.int64) + np.arange(8)[None, None, :]
        assert np.array_equal(keys[s_][np.ix_(prns.astype(int) - 1, [0, 5, 10, 15, 20])],
                              ((want["max_val"].astype(np.int64) << 14) | (16383 - fine)).max(axis=2)), scaling


def test_single_gpu_line_is_compact_strict_and_carries_every_leg(tmp_path):
    """`python bench.py` as the driver runs it at N = 1 (smaller batch, fewer steps, tracking ladders off): the LAST stdout line is
    the compact record -- under 4 KB, strict JSON, the contract's keys, `roofline` and `cpu_baseline`, one entry per secondary
    leg, none of them an error -- and the full record is in bench_detail.json beside the script."""
    from stm32f4_sdr_gps_amd import benchline
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--searches", "16",
                          "--no-tracking", "--cpu-budget-s", "2"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    last = [l for l in res.stdout.splitlines() if l.strip()][-1]
    line = benchline.check(last)                       # size, strict JSON, contract keys, roofline keys
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["config"]["blocks_per_search"] == 1
    assert line["config"]["hypotheses_per_step"] == 16 * 32 * 21 * 16368 and line["value"] > 1e9
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] <= 1 and line["roofline"]["kernel"].startswith("gpsx::k_acq_mx")
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["value"] > 1e5 and line["cpu_baseline"]["cores"] == 1
    assert line["value_pcie_inclusive"] > 1e9
    for leg in ("native_grid", "configs3_one_gpu", "letter_compliant", "weighted_2bit_extension"):
        assert "error" not in line[leg] and line[leg]["value"] > 1e9, (leg, line[leg])
    assert line["configs3_one_gpu"]["bound"] == "hbm" and line["letter_compliant"]["keys_identical_to_the_matrix_core_path"] is True
    assert line["letter_compliant"]["kernel"].startswith("gpsx::k_acq_poly")
    # (priced only where profiles/kernel_counters.json holds this kernel instance's issued instruction count: at 16 captures the
    #  library picks another instance of the polyphase kernel than at the bench's 256)
    assert 0 < line["letter_compliant"].get("frac", 0.5) <= 1
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    assert detail["value"] == pytest.approx(line["value"], rel=1e-5) and "note" in detail["roofline"]
    assert detail["configs3_one_gpu"]["roofline"]["algorithmic_bytes"] > 0 and detail["pcie_inclusive"]["serial"] > 1e9


def test_default_multi_gpu_run_is_the_single_gpu_workload_sharded(tmp_path, oracle):
    """`bench.py --gpus 2` with no --n-ms, as the driver launches it: the SAME workload as N = 1 (BASELINE.json configs[2], one block
    per search, `searches` captures PER GPU: per-GPU work fixed as N grows), its units sharded, one all-reduce(MAX) of the keys per
    step; configs[3] (ten-block searches, sharded, all-reduced) rides beside it as `configs3_sharded` and north_star's literal
    point (ONE ten-block search over the N ranks) as `single_search`.  The merged table = the unsharded sweep (GPSX_BENCH_VERIFY),
    and one search of each rank's half against the oracle, every key."""
    import numpy as np
    from stm32f4_sdr_gps_amd import benchline, synth
    dump = str(tmp_path / "merged_keys.npy")
    env = dict(os.environ, GPSX_BENCH_SHARE_DEVICE="1", GPSX_BENCH_BACKEND="gloo", GPSX_BENCH_VERIFY="1", GPSX_BENCH_DUMP_KEYS=dump)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--searches", "12", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "VERIFY sharded == unsharded" in res.stdout
    line = benchline.check([l for l in res.stdout.splitlines() if l.strip()][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["blocks_per_search"] == 1
    assert "configs[2]" in line["config"]["workload"]                       # the N = 1 line's workload, word for word
    assert line["config"]["searches_per_gpu_per_step"] == 12 and line["config"]["hypotheses_per_step"] == 24 * 32 * 21 * 16368
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] <= 1
    assert line["communicator"]["rccl_ranks"] == 2 and line["parity"]["parity_checked"] is True
    assert line["per_gpu_unsharded"]["value"] > 1e9
    ten = line["configs3_sharded"]
    assert "error" not in ten and ten["value"] > 1e9 and ten["bound"] == "hbm" and ten["kernel"].startswith("gpsx::k_acq_")
    assert line["single_search"]["ms_per_search"] > 0
    keys = np.load(dump)
    assert keys.shape == (24, 32, 21)
    blocks = synth.cold_start_block(24, seed=11, amp_scale=0.25)
    prns = np.arange(1, 33, dtype=np.uint8)
    for s_ in (3, 20):                                                       # one search of each rank's run of units
        want = oracle.acq_grid(blocks[s_:s_ + 1], 1, prns, -5000, 500, 21, 8, n_threads=max(4, min(32, len(os.sched_getaffinity(0)))), live=True)
        fine = 8 * want["phase"].astype(np.int64) + np.arange(8)[None, None, :]
        assert np.array_equal(keys[s_], ((want["max_val"].astype(np.int64) << 14) | (16383 - fine)).max(axis=2)), s_


def test_eight_ranks_default_workload(tmp_path):
    """The driver's 8-GPU launch shape with no --n-ms: eight ranks on device 0 over gloo, two captures per rank of the N = 1
    workload, the ten-block leg sharded over the eight ranks beside it; merged table = unsharded sweep, oracle sample inside the run."""
    from stm32f4_sdr_gps_amd import benchline
    env = dict(os.environ, GPSX_BENCH_SHARE_DEVICE="1", GPSX_BENCH_BACKEND="gloo", GPSX_BENCH_VERIFY="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--searches", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "VERIFY sharded == unsharded" in res.stdout
    line = benchline.check([l for l in res.stdout.splitlines() if l.strip()][-1])
    assert line["n_gpus"] == 8 and line["communicator"]["rccl_ranks"] == 8 and line["config"]["blocks_per_search"] == 1
    assert line["config"]["hypotheses_per_step"] == 16 * 32 * 21 * 16368 and line["parity"]["parity_checked"] is True
    assert "error" not in line["configs3_sharded"] and line["configs3_sharded"]["value"] > 1e9 and line["single_search"]["ms_per_search"] > 0
