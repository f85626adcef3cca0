"""Diagnostic (GPU box): who moves libc's rand() state during the 64-channel reference scenario?  The state buffer is installed
with initstate() so that every change is visible; run under LD_PRELOAD=tools/experiments/randshim.so (gcc -shared -fPIC randshim.c -ldl)
to see the callers -- libamd_comgr draws hundreds of values whenever a code object is loaded (EXPERIMENTS.md, round 4)."""
import ctypes as C, os, sys, zlib
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import steps_driver as sd
from golden_util import load
from stm32f4_sdr_gps_amd import capi, synth
g = load("f7_steps_config5_64ch.npz")
sats, chans, seed = sd.config5_64ch_scenario()
n_ms = 460
stream = synth.make_if(n_ms, sats, noise_amp=1.0, seed=seed)
lib = capi.load_library()
steps = sd.StepsLib(lib, False)
lib.gps_tracking_process_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint8]
lib.gps_tracking_process_batch.restype = None
libc = C.CDLL("libc.so.6")
buf = (C.c_char * 128)()
libc.initstate.restype = C.c_void_p
libc.initstate.argtypes = [C.c_uint, C.c_void_p, C.c_size_t]
libc.initstate(1, buf, 128)
def h(): return zlib.crc32(bytes(buf))
table = np.stack([sd.preset_channel(steps, *c) for c in chans])
h0 = h()
for t in range(n_ms):
    steps.set_time(t)
    lib.gps_tracking_process_batch(table.ctypes.data, len(chans), stream[t].ctypes.data, t & 3)
    h1 = h()
    if t in (453, 454): print(t, "ch36 freq", table[36, 64:68].view("<f4")[0])
    if h1 != h0:
        print("rand state changed during ms", t); h0 = h1
    crc = sd.snapshot_crcs(table)
    bad = np.flatnonzero(crc != g["crcs"][t])
    if len(bad):
        print("MISMATCH ms", t, bad[:8], "freq", table[bad[0], 64:68].view("<f4")[0], "found", table[bad[0], 2:4].view("<i2")[0])
        print("next draws", [libc.rand() % 500 for _ in range(4)]); break
print("done")
