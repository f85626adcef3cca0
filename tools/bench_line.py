#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> `searches ms_per_step hyp/s` (for shell loops over configurations)."""
import json
import sys

d = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", d["config"]["searches_per_gpu_per_step"], round(d["ms_per_step"], 4),
      "%.4g" % d["value"])
