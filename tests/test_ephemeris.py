"""The default gps_nav_data_decode_subframe (stm32f4_sdr_gps_amd/csrc/gpsx_ephemeris.cpp, host code: runs without a GPU)
against the reference's decoder (PM/GPS/nav_data_decode.c) on tests/golden/f8_ephemeris.npz: 60 subframe images of
every ID, random, all-ones and all-zeros payloads, decoded one after the other into one channel record.  Every byte of
eph_data after every call -- integer fields, the doubles (scale factors applied in the reference's order with the
reference's constants, three of which are not exact powers of two), times, counters and masks."""
import ctypes as C
import os

import numpy as np

import steps_driver as sd
from golden_util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ephemeris_decode_matches_reference_bytes(lib_path):
    g = load("f8_ephemeris.npz")
    lib = C.CDLL(lib_path)
    ids, snaps = sd.run_ephemeris(lib, g["imgs"])
    assert np.array_equal(ids, g["ids"])
    bad = np.argwhere(snaps != g["snaps"])
    assert len(bad) == 0, bad[:5]
    # spot values, so that the fixture itself is pinned to something readable: case 1 is a subframe 2
    eph = snaps[1]
    assert int(g["ids"][1]) == 2 and eph[4:8].view("<i4")[0] == int(np.packbits(
        np.unpackbits(g["imgs"][1], bitorder="little")[60:68])[0])                       # IODE: bits 60..67, MSB first
    assert 0.0 <= eph[88:96].view("<f8")[0] < 0.5                                        # eccentricity = 32 bits * ~2^-33


def test_ephemeris_decoder_is_an_overridable_hook(lib_path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", lib_path], text=True)
    assert any(l.split()[-1] == "gps_nav_data_decode_subframe" and " W " in l for l in out.splitlines())
