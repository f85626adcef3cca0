#!/usr/bin/env python3
"""What the chip's memory system gives a read-modify-write stream -- every byte read once and written once, the access mix of the
walk form's running sums (k_acq_mx<3>: 2 B read + 2 B written per hypothesis and block) -- at footprints below and above the
256 MB memory-side cache: x += 1 in place over int32 tensors (torch's elementwise kernel: 16-byte accesses, fully coalesced),
bytes moved = 2 x footprint per pass.  The bound the walk form's 3.7 TB/s is to be read against.
  tools/microbench/rmw_stream.py"""
import json
import time

import torch

dev = torch.device("cuda", 0)
for mb in (64, 128, 192, 256, 268, 320, 512, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.zeros(n, dtype=torch.int32, device=dev)
    for _ in range(3):
        x.add_(1)
    torch.cuda.synchronize()
    reps = max(5, 20000 // mb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        x.add_(1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"footprint_MB": mb, "ms_per_pass": round(ms, 4), "read_plus_write_TBps": round(2 * mb * (1 << 20) / (ms * 1e-3) / 1e12, 3)}), flush=True)
    del x
