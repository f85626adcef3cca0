set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "group or steps or host_demo or chunk" 2>&1 | tail -5 > gpurun_out/r03b_tests.log
cat gpurun_out/r03b_tests.log
python tools/bench_tracking_closed_loop.py --channels 256 16384 65536 131072 262144 --ms 1200 --signals 32 > gpurun_out/r03b_closed_loop.jsonl 2> gpurun_out/r03b_closed_loop.err
cat gpurun_out/r03b_closed_loop.jsonl; tail -3 gpurun_out/r03b_closed_loop.err
( time python bench.py > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/r03b_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03b_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("pcie_inclusive", {}).get("value"))
t = d["tracking"]
print(t["value"], [(r["channels"], round(r["p50_us"]), round(r["p99_us"]), round(r["max_us"])) for r in t["ladder"]])
print(t["closed_loop"]["value"], [(r["channels"], round(r["p50_us"]), round(r["p99_us"]), round(r["max_us"]), r["steps_over_1ms"], r["code_and_carrier_lock"]) for r in t["closed_loop"]["ladder"]])
PY
bash tools/profile_track.sh r03_track > gpurun_out/r03b_proft.log 2>&1
python tools/bench_track_kernel.py 212992 > gpurun_out/r03_track_kernel_us.json 2>&1
cat gpurun_out/r03_track_kernel_us.json
for ch in 4 8 12 16; do GPSX_TRACK_CHUNKS=$ch python - <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, ".")
from stm32f4_sdr_gps_amd import capi, synth
eng = capi.Engine(0); eng.bind_thread_to_device()
stream = synth.default_four_sv(8, seed=7)
blocks = eng.host_array(stream.shape, np.uint8); blocks[:] = stream
out = []
for n in (524288, 786432, 1048576, 1310720):
    st = eng.host_array(n, capi.TRK_DTYPE); iq = eng.host_array((n, 6), np.int16)
    st["prn"] = (np.arange(n) % 32) + 1
    st["code_phase_fine"] = (61 * np.arange(n) % 16368).astype(np.float32)
    st["if_freq_offset_hz"] = (-5000 + 39 * (np.arange(n) % 256)).astype(np.float32)
    for k in range(10): eng.track_epl(blocks[k % 8], st, iq)
    lat = np.zeros(300)
    for k in range(300):
        t0 = time.perf_counter(); eng.track_epl(blocks[k % 8], st, iq); lat[k] = time.perf_counter() - t0
    out.append((n, round(float(np.percentile(lat, 50) * 1e6)), round(float(np.percentile(lat, 99) * 1e6))))
print("chunks", os.environ["GPSX_TRACK_CHUNKS"], out)
PY
done 2>&1 | grep chunks | tee gpurun_out/r03b_chunks.txt
