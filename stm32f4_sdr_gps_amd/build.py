"""Build recipe for libgpsx.so (hipcc, gfx950, in-tree).  `python -m stm32f4_sdr_gps_amd.build` or build.build()."""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(PKG, "lib", "libgpsx.so")
LAB_LIB = os.path.join(PKG, "lib", "libgpsx_lab.so")


def build(verbose: bool = False, jobs: int = 4) -> str:
    """Compile every HIP source for gfx950 and link lib/libgpsx.so.  Cross-compiles without a GPU."""
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = ["make", "-C", os.path.join(PKG, "csrc"), f"-j{jobs}"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0 or not os.path.exists(LIB) or not os.path.exists(LAB_LIB):
        raise RuntimeError("building libgpsx.so failed")
    check_no_scratch()
    return LIB


def kernel_resources(obj: str | None = None) -> dict:
    """{kernel symbol: {"vgprs", "sgprs", "scratch_bytes", "lds_bytes"}} of the device code in build/k_acq_mx.o (or `obj`), read
    from the code object's metadata notes (llvm-readelf --notes on the unbundled gfx950 image)."""
    import re
    import tempfile
    obj = obj or os.path.join(PKG, "build", "k_acq_mx.o")
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as tmp:
        raw = open(obj, "rb").read()
        at = raw.find(b"__CLANG_OFFLOAD_BUNDLE__")     # the .hip_fatbin section of the host object
        if at < 0:
            raise RuntimeError(f"{obj}: no offload bundle inside")
        fat, img = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        with open(fat, "wb") as f:
            f.write(raw[at:])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={img}"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", img], text=True)
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))   # noqa: E731
        out[name.group(1)] = {"vgprs": get("vgpr_count"), "sgprs": get("sgpr_count"), "scratch_bytes": get("private_segment_fixed_size"),
                              "lds_bytes": get("group_segment_fixed_size")}
    return out


def check_no_scratch() -> dict:
    """The matrix-core grid kernels live one register from the spill cliff (k_acq_mx<3>: 255 VGPRs): a build whose k_acq_mx
    instance spills to scratch memory is refused here, not discovered as a slow kernel on the GPU box."""
    res = kernel_resources()
    mx = {k: v for k, v in res.items() if "k_acq_mx" in k}
    if not mx:
        raise RuntimeError("no k_acq_mx kernels found in build/k_acq_mx.o's code object metadata")
    # ... and the device tracking loops, whose occupancy (three waves per SIMD: 168 VGPRs) is asked for by __launch_bounds__
    loops = {k: v for k, v in kernel_resources(os.path.join(PKG, "build", "k_track_loop.o")).items() if "k_track_loop" in k}
    if len(loops) != 4:
        raise RuntimeError(f"expected four k_track_loop instances, found {sorted(loops)}")
    bad = {k: v for k, v in {**mx, **loops}.items() if v["scratch_bytes"] != 0}
    if bad:
        raise RuntimeError(f"kernels with scratch memory (register spills): {bad}")
    return {**mx, **loops}


if __name__ == "__main__":
    print(build(verbose=True))
