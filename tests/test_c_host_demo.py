"""The plain-C host (examples/gpsx_demo.c) built with gcc against libgpsx.so: "host code stays in C".
CPU: it compiles and links with nothing but include/gpsx_compat.h, and refuses to run without a GPU (abort + message).
GPU: it drives acquisition -> pre-track -> tracking over 3000 ms of IF read from a raw capture file and must end in
exactly the channel state the reference's own C reaches on that stream (tests/golden/f7_steps_hints.npz)."""
import os
import re
import subprocess

import numpy as np
import pytest

from golden_util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def demo_exe():
    from stm32f4_sdr_gps_amd import build
    build.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "gpsx_demo"], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "examples", "gpsx_demo")


def _have_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_have_gpu(), reason="a GPU is present")
def test_c_host_builds_and_fails_loudly_without_gpu(demo_exe, tmp_path):
    cap = tmp_path / "cap.bin"
    np.zeros(2046 * 4, np.uint8).tofile(cap)
    res = subprocess.run([demo_exe, str(cap)], capture_output=True, text=True)
    assert res.returncode != 0                     # abort(), no silent CPU path
    assert "no CPU path" in res.stderr and "gpsx_create" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("device_loops", [False, True])
def test_c_host_reaches_the_reference_end_state(demo_exe, tmp_path, device_loops):
    """... with the tracking steps through the reference-named calls, and (--device-loops) with the loops on the GPU in the
    firmware's 17 ms multiplex from the first cycle start on which all four channels track: the same end state, bit for bit."""
    from stm32f4_sdr_gps_amd import synth
    g = load("f7_steps_hints.npz")
    n_ms = int(g["n_ms"])
    cap = tmp_path / "rec_file.bin"
    synth.four_sv_with_nav(n_ms, seed=7).tofile(cap)
    out = subprocess.run([demo_exe, str(cap)] + (["--device-loops"] if device_loops else []), capture_output=True, text=True, check=True).stdout
    lines = [l for l in out.splitlines() if l.startswith("PRN=")]
    assert len(lines) == 4 and f"processed_ms={n_ms}" in out
    handed = int(re.search(r"handed_to_the_device_at_ms=(-?\d+)", out).group(1))
    assert (400 < handed < 600 and handed % 17 == 0) if device_loops else handed == -1
    last = g["snaps"][-1]
    for i, line in enumerate(lines):
        m = re.search(r"acq_state=(\d+) code_phase=(\d+) doppler_hz=(-?\d+) trk_state=(\d+) code_phase_fine=\S+\(0x(\w+)\) "
                      r"if_freq_offset_hz=\S+\(0x(\w+)\) nco=0x(\w+)", line)
        acq_state, code_phase, dopp, trk_state = (int(m.group(k)) for k in range(1, 5))
        fine_bits, freq_bits, nco = (int(m.group(k), 16) for k in (5, 6, 7))
        a = last[i]
        assert acq_state == int(a[16:20].view("<i4")[0]) == 9
        assert code_phase == int(a[6:8].view("<u2")[0])
        assert dopp == int(a[2:4].view("<i2")[0])
        assert trk_state == int(a[60 + 148:60 + 152].view("<i4")[0]) == 4
        assert fine_bits == int(a[60 + 80:60 + 84].view("<u4")[0])        # float loop state, bit for bit
        assert freq_bits == int(a[60 + 4:60 + 8].view("<u4")[0])
        assert nco == int(a[60 + 8:60 + 12].view("<u4")[0])


@pytest.fixture(scope="module")
def coldstart_exe():
    from stm32f4_sdr_gps_amd import build
    build.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "gpsx_coldstart"], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "examples", "gpsx_coldstart")


@pytest.mark.skipif(_have_gpu(), reason="a GPU is present")
def test_c_coldstart_builds_and_reports_the_missing_gpu(coldstart_exe, tmp_path):
    cap = tmp_path / "cap.bin"
    np.zeros(2046 * 4, np.uint8).tofile(cap)
    res = subprocess.run([coldstart_exe, str(cap)], capture_output=True, text=True)
    assert res.returncode == 1 and "gpsx_create" in res.stderr            # GPSX_ENODEV, no CPU path


@pytest.mark.gpu
def test_c_coldstart_finds_the_satellites_of_a_recording(coldstart_exe, tmp_path):
    """examples/gpsx_coldstart.c (plain C on include/gpsx.h): one gpsx_acq_grid call over 4 ms of a raw IF file must list
    exactly the four satellites of the recording with their Doppler bin and code phase."""
    from stm32f4_sdr_gps_amd import synth
    cap = tmp_path / "rec_file.bin"
    synth.default_four_sv(6, seed=7).tofile(cap)
    out = subprocess.run([coldstart_exe, str(cap), "4", "1"], capture_output=True, text=True, check=True).stdout
    found = {}
    for line in out.splitlines():
        m = re.match(r"PRN=(\d+) doppler_hz=(-?\d+) code_phase_samples=(\d+) energy=(\d+) ratio=([\d.]+)", line)
        if m:
            found[int(m.group(1))] = (int(m.group(2)), int(m.group(3)), float(m.group(5)))
    truth = {5: (912.5, 1600), 14: (4037.0, 4000), 20: (-1025.0, 9000), 30: (2018.0, 13000)}
    assert set(found) == set(truth), out
    for prn, (dopp, delay) in truth.items():
        # (the strongest Doppler bin is not always the nearest: the reference's one-sided clip, quirk Q4, favours the bin
        #  whose residual carrier keeps I and Q positive over the window)
        assert abs(found[prn][0] - dopp) <= 750 and abs(found[prn][1] - delay) <= 2 and found[prn][2] > 2.5, (prn, found[prn])
    assert f"hypotheses={32 * 21 * 16368 * 4}" in out


@pytest.fixture(scope="module")
def track_loop_exe():
    from stm32f4_sdr_gps_amd import build
    build.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "gpsx_track_loop"], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "examples", "gpsx_track_loop")


def test_c_track_loop_host_builds(track_loop_exe):
    assert os.access(track_loop_exe, os.X_OK)


@pytest.mark.gpu
def test_c_track_loop_host_ends_in_the_reference_state(track_loop_exe, tmp_path):
    """examples/gpsx_track_loop.c: host mode to tick 600, then the tracking loops on the GPU (K = 20 ms per launch) to the end of
    the 2500 ms stream -- code phase, carrier and NCO accumulator of every channel must be, bit for bit, what the reference's
    own gps_tracking_process reaches on that stream (tests/golden/f7_steps_continuous.npz)."""
    from stm32f4_sdr_gps_amd import synth
    g = load("f7_steps_continuous.npz")
    n_ms = int(g["n_ms"])
    cap = tmp_path / "rec_file.bin"
    synth.four_sv_with_nav(n_ms, seed=7).tofile(cap)
    presets = [f"{int(p)}:{int(f)}:{int(c)}" for p, f, c in zip(g["prns"], g["found_freq"], g["found_phase"])]
    out = subprocess.run([track_loop_exe, str(cap), "600"] + presets, capture_output=True, text=True, check=True).stdout
    lines = [l for l in out.splitlines() if l.startswith("PRN=")]
    assert len(lines) == 4 and f"processed_ms={n_ms} handed_over_at_ms=600" in out
    last = g["snaps"][-1]
    for i, line in enumerate(lines):
        m = re.search(r"trk_state=(\d+) code_phase_fine=\S+\(0x(\w+)\) if_freq_offset_hz=\S+\(0x(\w+)\) nco=0x(\w+) snr_db=\S+ "
                      r"bit_sync=(\d+) false_lock_jumps=(\d+)", line)
        a = last[i]
        assert int(m.group(1)) == 4 and int(m.group(6)) == 0
        assert int(m.group(2), 16) == int(a[60 + 80:60 + 84].view("<u4")[0])
        assert int(m.group(3), 16) == int(a[60 + 4:60 + 8].view("<u4")[0])
        assert int(m.group(4), 16) == int(a[60 + 8:60 + 12].view("<u4")[0])
        assert int(m.group(5)) == int(a[212])
