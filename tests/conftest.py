"""pytest configuration: marker registration and shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, oracle vs the in-place reference build (when present), host logic,
                C-ABI symbol checks, gloo multi-process tests.
`-m gpu`      : parity tests proper -- the HIP engine through the C-ABI against the oracle / golden vectors.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def oracle_threads():
    """threads for the oracle's grid sweeps: the CPUs this process may use, 8 at least (GPU boxes cap the container at ~16 CPUs
    of a 256-thread host), 32 at most"""
    return max(8, min(32, len(os.sched_getaffinity(0))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle.  Its LARGE grid sweeps (ten-block searches, multi-search batches: tens of seconds of CPU each, 300 s of
    the GPU suite in round 5) are answered from tests/golden/f11_grids/ -- this oracle's own outputs for exactly those input
    bytes, generated in the build container (oracle/gen_golden_grids.py) -- unless the call says live=True or
    $GPSX_ORACLE_LIVE=1; everything smaller is computed live, so every kernel form keeps live-oracle cases."""
    from oracle import pyoracle
    orc = pyoracle.Oracle()
    if os.environ.get("GPSX_ORACLE_LIVE") != "1":
        orc.grid_fixtures = os.path.join(ROOT, "tests", "golden", "f11_grids")
    return orc


@pytest.fixture(scope="session")
def ref_pm():
    from oracle import pyoracle
    if not pyoracle.RefPM.available():
        pyoracle.build_ref()
    if not pyoracle.RefPM.available():
        pytest.skip("oracle/_ref/libref_pm.so not built (needs /root/reference)")
    return pyoracle.RefPM()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def lib_path():
    """Path of the built product library (hipcc cross-compiles without a GPU) -- or, under tools/run_sanitizers.sh, of the
    sanitizer build of the same sources that $GPSX_LIB_PATH names (capi.load_library honours the same variable)."""
    from stm32f4_sdr_gps_amd import build
    built = build.build()
    return os.environ.get("GPSX_LIB_PATH") or built
