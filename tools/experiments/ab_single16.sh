python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from stm32f4_sdr_gps_amd import capi, synth
blocks = synth.cold_start_block(30, seed=77, amp_scale=0.3)
prns = np.concatenate([np.arange(1, 33), [33, 40, 61, 100, 120, 150, 200, 210]]).astype(np.uint8)
os.environ["GPSX_ACQ_SINGLE16"] = "1"
e = capi.Engine(0)
del os.environ["GPSX_ACQ_SINGLE16"]
ref = capi.Engine(0)
for kw in (dict(n_search=30, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21), dict(n_search=7, dopp_min_hz=-1000, dopp_step_hz=250, n_dopp=5, win=(5, 2001)),
           dict(n_search=30, dopp_min_hz=-5000, dopp_step_hz=500, n_dopp=21, shard=(1, 3))):
    for pr in (prns[:32], prns):
        a, ak = e.acq_grid(blocks, pr, **kw)
        k1 = e.lib.gpsx_last_kernel(e.h)
        b, bk = ref.acq_grid(blocks, pr, **kw)
        print(k1, ref.lib.gpsx_last_kernel(ref.h), "equal:", np.array_equal(a, b) and np.array_equal(ak, bk), int((a["max_val"] != b["max_val"]).sum()), int((a["sum"] != b["sum"]).sum()))
PY
for i in 1 2 3; do
  echo -n "16 waves: "; GPSX_ACQ_SINGLE16=1 python tools/bench_grid_kernel.py 256 1 20 2>/dev/null | tail -1 | cut -c60-200
  echo -n "8 waves:  "; python tools/bench_grid_kernel.py 256 1 20 2>/dev/null | tail -1 | cut -c60-200
done
