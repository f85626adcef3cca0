#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X GPS L1 C/A correlator engine.

Metric (BASELINE.json): acquisition hypotheses / second (PRN x Doppler x code phase).
Workload (BASELINE.json configs[2], SURVEY.md 8(d) "Config 3"): cold-start grid, all 32 PRN x 21 Doppler bins
(+-5 kHz @ 500 Hz) x 16368 code phases (2046 byte offsets x 8 replica bit shifts), 1 ms coherent, synthetic
16.368 Msps IF with six satellites in view (each below the noise floor; --amp-scale), by default as 2-bit sign/magnitude
pairs (--if-format; the magnitude bit travels and is ignored, as in the reference).  One STEP = one gpsx_acq_grid_dev() call over a batch of
`--searches` independent 1 ms captures per GPU, inputs already resident in HBM, results (per-hypothesis-unit peak
triplets + packed peak keys) left in HBM.

N GPUs (torchrun, one rank per GPU): the job holds N x searches captures; every capture's (PRN group, Doppler) grid
units are dealt round-robin to the ranks (per-GPU work is constant: weak scaling) and ONE all-reduce(MAX) of the packed
(energy, phase) key table over RCCL merges the peaks -- the only collective on the path.

Prints one JSON line on rank 0.  `roofline` is priced the way SURVEY.md 8(d) prescribes for a bytes-based roofline
(6138 operand bytes per hypothesis as the reference streams them); the kernel itself keeps its operands in LDS and is
bound by integer VALU issue, which `roofline_valu` prices (2048 lane-ops per hypothesis in the reference's XOR/popcount
formulation; the polyphase kernel issues ~9x fewer).  `cpu_baseline` times the reference's own C (oracle/_ref, built in
place from the reference tree) -- or the CPU oracle port when that build is absent -- on a bounded sample.  `tracking`
is BASELINE.json's second metric, bounded to a few hundred steps per channel count (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PRN, N_DOPP, DOPP_MIN, DOPP_STEP, N_PHASE = 32, 21, -5000, 500, 16368
HYP_PER_SEARCH = N_PRN * N_DOPP * N_PHASE          # 10 999 296
BYTES_PER_HYP = 6138                                 # I + Q + replica, 3 x 2046 B per hypothesis
LANE_OPS_PER_HYP = 2048                              # 1024 xor + 1024 bcnt
HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_INT_PEAK_TOPS = 256 * 64 * 2.4e9 / 1e12         # 39.3 T lane-ops/s: 64 int lanes/clk/CU measured (tools/microbench)


def _ref_prn_slice(ref, blk, prn_list, deadline):
    """All 21 Doppler bins x 8 bit shifts x 2046 offsets for each PRN of the slice, with the reference's own calls."""
    done = 0
    for p in prn_list:
        chips = ref.ca_code(p)
        reps = [ref.replica(chips, b) for b in range(8)]
        for d in range(N_DOPP):
            di, dq = ref.wipeoff(blk, float(4092000 + DOPP_MIN + d * DOPP_STEP))
            for b in range(8):
                ref.correlation_search(reps[b], di, dq, 0, 2046)
            done += 8 * 2046
        if time.perf_counter() > deadline:
            break
    return done


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return f"{line.split(':', 1)[1].strip()} ({os.cpu_count()} logical CPUs)"
    except OSError:
        pass
    return f"unknown ({os.cpu_count()} logical CPUs)"


def tracking_channels(eng_cls, dev_index, steps=400):
    """BASELINE.json's second metric, bounded: the largest channel count of a fixed ladder whose per-millisecond E/P/L step
    (gpsx_track_epl_batch: block + states in, one launch, states + accumulators out) keeps its p99 under 1 ms."""
    from stm32f4_sdr_gps_amd import capi, synth
    eng = eng_cls(dev_index)
    stream = synth.default_four_sv(8, seed=7)
    rows, best = [], None
    for n in (256, 4096, 65536, 131072, 196608, 229376, 262144):
        st = np.zeros(n, capi.TRK_DTYPE)
        st["prn"] = (np.arange(n) % 32) + 1
        st["code_phase_fine"] = (61 * np.arange(n) % 16368).astype(np.float32)
        st["if_freq_offset_hz"] = (-5000 + 39 * (np.arange(n) % 256)).astype(np.float32)
        for k in range(20):
            eng.track_epl(stream[k % 8], st)
        lat = np.zeros(steps)
        for k in range(steps):
            t0 = time.perf_counter()
            eng.track_epl(stream[k % 8], st)
            lat[k] = time.perf_counter() - t0
        p50, p99 = float(np.percentile(lat, 50) * 1e6), float(np.percentile(lat, 99) * 1e6)
        rows.append({"channels": n, "p50_us": p50, "p99_us": p99})
        if p99 < 1000.0:
            best = n
        else:
            break
    eng.close()
    return {"metric": "real-time tracking channels (p99 of the E/P/L step per ms < 1 ms, host round trip included)",
            "value": best, "steps_per_count": steps, "ladder": rows,
            "note": "10000-step measurements and the closed-loop figure are in profiles/r01_tracking_*.json"}


def cpu_baseline(blocks, budget_s=20.0):
    """Time the CPU path on a bounded sample of the same workload (capture 0 of the batch, the bench's own grid).
    Preferred: the reference's own C (oracle/_ref/libref_pm.so, built in place from the reference tree with gcc -O2) --
    on one core (`cpu_baseline`) and on many cores (`cpu_baseline_allcores`: threads calling the same library, which
    only shares its read-only popcount table).  Without that build: the CPU oracle port (OpenMP)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle
    pyoracle.build()
    out = {}
    blk = np.ascontiguousarray(blocks[0])
    what = ("reference C (gps_misc.c, gcc -O2): gps_shift_to_zero_freq + gps_generate_prn_data2 + correlation_search "
            "per (PRN, Doppler, bit shift), PRN-major order")
    if pyoracle.RefPM.available():
        ref = pyoracle.RefPM()
        t0 = time.perf_counter()
        done = _ref_prn_slice(ref, blk, range(1, N_PRN + 1), t0 + 0.6 * budget_s)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": done / dt, "unit": "hypotheses/s", "cores": 1, "kind": "reference",
                               "cpu": _cpu_model(),
                               "sample": f"{done} of the {HYP_PER_SEARCH} hypotheses of capture 0 in {dt:.1f} s; {what}"}
        threads = max(1, min(32, (os.cpu_count() or 2) // 2))
        reps = 4
        slices = [[(i % N_PRN) + 1 for i in range(t, N_PRN * reps, threads)] for t in range(threads)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            done = sum(ex.map(lambda sl: _ref_prn_slice(ref, blk, sl, t0 + 0.4 * budget_s), slices))
        dt = time.perf_counter() - t0
        out["cpu_baseline_allcores"] = {"value": done / dt, "unit": "hypotheses/s", "cores": threads, "kind": "reference",
                                        "cpu": _cpu_model(),
                                        "sample": f"{done} hypotheses ({reps} passes over capture 0's grid) in {dt:.1f} s "
                                                  f"on {threads} threads; {what}"}
        return out
    orc = pyoracle.Oracle()
    threads = max(1, min(64, (os.cpu_count() or 2) // 2))
    prns = np.arange(1, N_PRN + 1, dtype=np.uint8)
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.acq_grid(blk[None, :], 1, prns, DOPP_MIN, DOPP_STEP, N_DOPP, 8, n_threads=threads)
        reps += 1
        if time.perf_counter() - t0 > budget_s / 2:
            break
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": reps * HYP_PER_SEARCH / dt, "unit": "hypotheses/s", "cores": threads, "kind": "port",
                           "cpu": _cpu_model(),
                           "sample": f"{reps} full grids of capture 0 in {dt:.1f} s: oracle/gpsx_oracle.c, OpenMP over "
                                     "(PRN, Doppler) pairs"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--searches", type=int, default=64, help="1 ms captures per GPU per step")
    ap.add_argument("--amp-scale", type=float, default=0.25,
                    help="scale of the six synthetic satellites' amplitudes: 0.25 (default) puts each satellite below the "
                         "noise like a live antenna; 1.0 is the strong test signal, whose long runs of saturated block sums "
                         "take the kernel's exact-correction pass far more often (reported in profiles/ as the slow case)")
    ap.add_argument("--n-ms", type=int, default=1,
                    help="blocks integrated non-coherently per search (BASELINE.json configs[3] uses 10); hypotheses are "
                         "then counted per block, as SURVEY.md 8(d) config 4 does")
    ap.add_argument("--if-format", choices=["2bit", "1bit"], default="2bit",
                    help="sample format of the captures in HBM: 2bit = MAX2769-style sign/magnitude pairs, 4092 bytes per ms, "
                         "unpacked to the sign plane in LDS inside the kernels (the reference's correlator never looks at "
                         "the magnitude bit either); 1bit = the 2046-byte sign stream the firmware's SPI delivers")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tracking", action="store_true",
                    help="skip the secondary metric (real-time tracking channels: E/P/L steps of growing channel counts)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # Test hooks (tests/test_gpu_bench_dist.py, one-GPU boxes): GPSX_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and
    # GPSX_BENCH_BACKEND=gloo swaps RCCL for gloo (RCCL refuses two ranks on one device); the sharded data path is unchanged.
    dev_index = 0 if os.environ.get("GPSX_BENCH_SHARE_DEVICE") == "1" else local_rank
    backend = os.environ.get("GPSX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("GPSX_BENCH_FORCE_DIST") == "1"   # the latter: exercise RCCL at world 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from stm32f4_sdr_gps_amd import capi, synth  # after torch: one HIP runtime per process

    stream = torch.cuda.Stream(device=dev)
    eng = capi.Engine(dev_index, stream=stream.cuda_stream)
    dev_name, cus, clk_khz = eng.device_info()

    n_search = args.searches * world
    # synthetic captures: consecutive milliseconds of one stream; identical on every rank (each rank reads all of it)
    n_ms = args.n_ms
    two_bit = args.if_format == "2bit"
    # the sign plane as 2046-byte blocks: the device input in 1-bit mode, and what the CPU leg (rank 0, N = 1) is timed on
    need_sign_plane = not two_bit or (world == 1 and not args.no_cpu_baseline)
    blocks = synth.cold_start_block(n_search * n_ms if not two_bit else 1, seed=11, amp_scale=args.amp_scale) \
        if need_sign_plane else None
    if two_bit:
        dev_blocks = synth.cold_start_block(n_search * n_ms, seed=11, amp_scale=args.amp_scale, two_bit=True)
        eng.set_if_format(capi.IF_2BIT_SM)
    else:
        dev_blocks = blocks
    prns = np.arange(1, N_PRN + 1, dtype=np.uint8)
    g = eng.grid_desc(prns, n_search=n_search, n_ms=n_ms, search_stride_blocks=n_ms, dopp_min_hz=DOPP_MIN,
                      dopp_step_hz=DOPP_STEP, n_dopp=N_DOPP, phase_mode=capi.PHASES_FINE, win=(0, 2046),
                      shard=(rank, world))
    import ctypes as C
    with torch.cuda.stream(stream):
        d_if = torch.from_numpy(np.concatenate([dev_blocks.reshape(-1), np.zeros(2, np.uint8)])).to(dev)
        d_peaks = torch.zeros((n_search, N_PRN, N_DOPP, 8, 4), dtype=torch.int32, device=dev)
        # two key tables: the all-reduce of step k (RCCL's own stream) overlaps the grid kernel of step k + 1
        key_bufs = [torch.zeros((n_search, N_PRN, N_DOPP), dtype=torch.int64, device=dev) for _ in range(2)]
        pending = [None, None]
        step_no = [0]

        def step():
            slot = step_no[0] & 1
            step_no[0] += 1
            if pending[slot] is not None:      # the table is about to be overwritten: its exchange must have finished
                pending[slot].wait()
                pending[slot] = None
            keys_t = key_bufs[slot]
            rc = eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(g), d_if.data_ptr(), n_search * n_ms, d_peaks.data_ptr(),
                                           keys_t.data_ptr(), None, None, None)
            if rc != 0:
                raise RuntimeError(f"gpsx_acq_grid_dev -> {rc}: {eng.lib.gpsx_last_error(eng.h).decode()}")
            if use_dist:   # the ONE collective of the path: max-merge of the packed (energy, phase) keys
                pending[slot] = dist.all_reduce(keys_t, op=dist.ReduceOp.MAX, async_op=True)

        def drain():
            for i in range(2):
                if pending[i] is not None:
                    pending[i].wait()
                    pending[i] = None

        for _ in range(args.warmup):
            step()
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = eng.event(), eng.event()
        t0 = time.perf_counter()
        eng.record(ev0)
        for _ in range(args.steps):
            step()
        eng.record(ev1)
        drain()                                  # every step's exchange is complete inside the timed region
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gpu_ms = eng.elapsed_ms(ev0, ev1)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed_s = float(elapsed.item())

    # PCIe-inclusive rate of the host-buffer entry point (H2D of the captures + launch + D2H of peaks and keys); reported
    # as an extra, never as `value`
    pcie = None
    if world == 1 and n_ms == 1:
        g1 = eng.grid_desc(prns, n_search=n_search, n_ms=1, search_stride_blocks=1, dopp_min_hz=DOPP_MIN,
                           dopp_step_hz=DOPP_STEP, n_dopp=N_DOPP, phase_mode=capi.PHASES_FINE)
        # host buffers in pinned memory, as SURVEY.md 8(d) defines the metric (torch only provides the pinned pages)
        pin_pk = torch.zeros(n_search * N_PRN * N_DOPP * 8 * capi.PEAK_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
        pin_keys = torch.zeros(n_search * N_PRN * N_DOPP, dtype=torch.int64).pin_memory()
        pin_if = torch.from_numpy(dev_blocks.reshape(-1).copy()).pin_memory()
        h_peaks, h_keys, h_if = pin_pk.numpy(), pin_keys.numpy(), pin_if.numpy()
        reps = 10
        for i in range(reps + 2):
            if i == 2:
                tp = time.perf_counter()
            rc = eng.lib.gpsx_acq_grid(eng.h, C.byref(g1), h_if.ctypes.data, n_search, h_peaks.ctypes.data,
                                       h_keys.ctypes.data)
            assert rc == 0
        pcie = reps * n_search * HYP_PER_SEARCH / (time.perf_counter() - tp)

    # sanity outside the timed region: the merged key table must hold the six synthetic satellites' peaks
    d_keys = key_bufs[(step_no[0] - 1) & 1]
    keys = d_keys.cpu().numpy()
    energy = keys >> 14
    assert (energy > 0).all() and energy.max() > 1500 * min(1.0, args.amp_scale), "acquisition grid produced no peaks"

    if use_dist and os.environ.get("GPSX_BENCH_VERIFY") == "1":
        # the merged table must equal what one GPU computes for the whole grid (checked outside the timed region)
        g_all = eng.grid_desc(prns, n_search=n_search, n_ms=n_ms, search_stride_blocks=n_ms, dopp_min_hz=DOPP_MIN,
                              dopp_step_hz=DOPP_STEP, n_dopp=N_DOPP, phase_mode=capi.PHASES_FINE)
        with torch.cuda.stream(stream):
            d_keys_all = torch.zeros_like(d_keys)
            rc = eng.lib.gpsx_acq_grid_dev(eng.h, C.byref(g_all), d_if.data_ptr(), n_search * n_ms, d_peaks.data_ptr(),
                                           d_keys_all.data_ptr(), None, None, None)
            assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(d_keys_all.cpu(), torch.from_numpy(keys)), "sharded sweep + all-reduce != unsharded sweep"
        if rank == 0:
            print("VERIFY sharded == unsharded", flush=True)

    if rank == 0:
        total_hyp = float(args.steps) * n_search * n_ms * HYP_PER_SEARCH
        value = total_hyp / elapsed_s
        launch_ms = gpu_ms / args.steps                   # HIP events on the engine's stream around the K launches
        hyp_per_launch = args.searches * n_ms * HYP_PER_SEARCH   # per GPU
        ach_gbs = hyp_per_launch * BYTES_PER_HYP / (launch_ms * 1e-3) / 1e9
        ach_tops = hyp_per_launch * LANE_OPS_PER_HYP / (launch_ms * 1e-3) / 1e12
        traffic = None
        tr_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tr_file):
            with open(tr_file) as f:
                tr = json.load(f)
            if (tr.get("searches_per_launch") == args.searches and n_ms == 1
                    and os.environ.get("GPSX_ACQ_ALGO", "poly") == "poly"):   # measured for the default kernel only
                traffic = tr.get("hbm_bytes_per_launch")
        line = {
            "metric": "acquisition hypotheses/sec (PRN x Doppler x phase)",
            "value": value,
            "unit": "hypotheses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_s / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u1 (bit planes; u32 popcount accumulators)",
            "data": f"synthetic (6 SVs, amplitude scale {args.amp_scale}, U(-1,1) noise, seed 11; "
                    f"{'4092-byte 2-bit' if two_bit else '2046-byte 1-bit'} blocks)",
            "config": {
                "workload": ("cold-start acquisition grid: 32 PRN x 21 Doppler (+-5 kHz @ 500 Hz) x 16368 code phases, "
                             f"1 ms coherent, 16.368 Msps {'2-bit sign/magnitude' if two_bit else '1-bit'} IF "
                             "(BASELINE.json configs[2])") if n_ms == 1 else
                            (f"32-PRN acquisition grid (21 Doppler x 16368 phases) with {n_ms} ms non-coherent integration "
                             "(BASELINE.json configs[3]); hypotheses counted per 1 ms block"),
                "searches_per_gpu_per_step": args.searches,
                "hypotheses_per_step": n_search * n_ms * HYP_PER_SEARCH,
                "blocks_per_search": n_ms,
                "parallelism": f"grid units dealt round-robin to {world} rank(s); one all-reduce(MAX) of packed peak keys"
                               if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": ach_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                # what the chip's memory system actually moved (rocprofv3 FETCH_SIZE + WRITE_SIZE of this kernel and
                # launch size, profiles/hbm_traffic.json), as a rate over this run's launch time and against the peak
                "traffic_gbs": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None,
                "traffic_frac_of_peak": (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "kernel": {"poly": "gpsx::k_acq_poly<8,16,0>", "dot8": "gpsx::k_acq<8,false,dot8>", "sad": "gpsx::k_acq<8,false,sad>", "ds": "gpsx::k_acq_ds<21>"}.get(os.environ.get("GPSX_ACQ_ALGO", "poly"), "gpsx::k_acq_poly<8,16,0>"),
                "kernel_ms": launch_ms,
                "note": "algorithmic bytes = 6138 B/hypothesis as the reference streams its operands (SURVEY.md 8(d)); "
                        "the kernel stages the 2 KB capture in LDS, so real HBM traffic is ~0 and frac may exceed 1; "
                        "the binding resource is integer VALU issue, see roofline_valu",
            },
            "roofline_valu": {
                "bound": "valu-int",
                "achieved": ach_tops,
                "peak": VALU_INT_PEAK_TOPS,
                "unit": "Tlane-op/s",
                "frac": ach_tops / VALU_INT_PEAK_TOPS,
                "note": "algorithmic lane-ops = 2048/hypothesis (reference XOR+popcount formulation); the polyphase "
                        "kernel issues ~210/hypothesis, so frac > 1; issued-op efficiency (VALU issue saturated) is in "
                        "profiles/",
            },
            "device": {"name": dev_name, "compute_units": cus, "clock_khz": clk_khz},
        }
        if pcie is not None:
            line["pcie_inclusive"] = {"value": pcie, "unit": "hypotheses/s",
                                      "note": "gpsx_acq_grid() with pinned host buffers: H2D captures + launch + D2H peaks/keys"}
        if not args.no_tracking and world == 1:
            line["tracking"] = tracking_channels(capi.Engine, dev_index)
        if not args.no_cpu_baseline and world == 1:
            line.update(cpu_baseline(blocks, args.cpu_budget_s))
            line["cpu_host"] = {"logical_cpus": os.cpu_count()}
        print(json.dumps(line), flush=True)

    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
