"""Build recipe for libgpsx.so (hipcc, gfx950, in-tree).  `python -m stm32f4_sdr_gps_amd.build` or build.build()."""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(PKG, "lib", "libgpsx.so")


def build(verbose: bool = False, jobs: int = 4) -> str:
    """Compile every HIP source for gfx950 and link lib/libgpsx.so.  Cross-compiles without a GPU."""
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = ["make", "-C", os.path.join(PKG, "csrc"), f"-j{jobs}"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0 or not os.path.exists(LIB):
        raise RuntimeError("building libgpsx.so failed")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
