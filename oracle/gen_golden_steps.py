#!/usr/bin/env python3
"""oracle/gen_golden_steps.py -- TEST INFRASTRUCTURE.  Records golden state traces of the reference's STEP logic
(acquisition.c + tracking.c + nav_data.c, compiled in place into oracle/_ref/libref_steps.so together with
oracle/ref_time_source.c, the one symbol those files expect from the MCU capture driver) driven in the firmware's own
call order by tests/steps_driver.py on this repo's seeded synthetic IF stream.  Output: tests/golden/f7_steps_*.npz.
Build container only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import steps_driver as sd  # noqa: E402
from golden_util import fnv1a32  # noqa: E402
from oracle import pyoracle  # noqa: E402
from stm32f4_sdr_gps_amd import synth  # noqa: E402

SCENARIOS = {
    # name: (n_ms, prns, Doppler hints (0 = frequency search), stream function)
    "hints": (3000, [5, 14, 20, 30], [900, 4000, -1000, 2000]),   # PM/main.c:59-73, the firmware's default table
    "cold": (1800, [5, 14, 20, 30], [0, 4000, -1000, 2000]),      # channel 0 has no hint: full frequency search
}


def main():
    pyoracle.build_ref()
    for name, (n_ms, prns, hints) in SCENARIOS.items():
        lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_steps.so"))
        C.CDLL("libc.so.6").srand(1)
        stream = synth.four_sv_with_nav(n_ms, seed=7)
        snaps = sd.run_scenario(sd.StepsLib(lib, True), stream, prns, hints, n_ms)
        path = os.path.join(ROOT, "tests", "golden", f"f7_steps_{name}.npz")
        np.savez_compressed(path, snaps=snaps, prns=np.array(prns), hints=np.array(hints), n_ms=np.int32(n_ms),
                            stream_fnv=np.uint32(fnv1a32(stream[::97])))
        print(name, os.path.getsize(path), "bytes")
        for r in sd.summarize(snaps):
            print("  ", r)
        del lib


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def gen_continuous():
    """Golden for the batched step: the reference's gps_tracking_process run on ONE channel per library instance (so the
    file-static slot buffers of tracking.c / nav_data.c are private to the channel), every millisecond served, index =
    t & 3; acquisition results are preset from the known delays / hints."""
    import shutil
    import tempfile
    pyoracle.build_ref()
    n_ms = 2500
    prns, found_freq, found_phase = [5, 14, 20, 30], [900, 4000, -1000, 2000], [200, 500, 1124, 1624]
    stream = synth.four_sv_with_nav(n_ms, seed=7)
    snaps = np.zeros((n_ms, 4, sd.SNAP), np.uint8)
    src = os.path.join(ROOT, "oracle", "_ref", "libref_steps.so")
    with tempfile.TemporaryDirectory() as td:
        for c in range(4):
            path = os.path.join(td, f"ref_copy_{c}.so")
            shutil.copy(src, path)
            steps = sd.StepsLib(C.CDLL(path), True)
            C.CDLL("libc.so.6").srand(1)
            table = sd.preset_channel(steps, prns[c], found_freq[c], found_phase[c])
            for t in range(n_ms):
                steps.set_time(t)
                blk = np.ascontiguousarray(stream[t])
                steps.lib.gps_tracking_process(table.ctypes.data, blk.ctypes.data, t & 3)
                snaps[t, c] = sd.snapshot(table[None, :])[0]
    path = os.path.join(ROOT, "tests", "golden", "f7_steps_continuous.npz")
    np.savez_compressed(path, snaps=snaps, prns=np.array(prns), found_freq=np.array(found_freq),
                        found_phase=np.array(found_phase), n_ms=np.int32(n_ms),
                        stream_fnv=np.uint32(fnv1a32(stream[::97])))
    print("continuous", os.path.getsize(path), "bytes")
    for r in sd.summarize(snaps):
        print("  ", r)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "continuous":
    gen_continuous()


def gen_lnav():
    """Golden for the word layer (nav_data.c:257-451): 15 s of the 4-SV table carrying parity-correct LNAV subframes, two
    satellites with inverted data polarity.  Too long to keep every snapshot: a CRC of the full channel state (acq_data,
    tracking_data, all of nav_data) per millisecond, the full state every 500 ms, and the final state."""
    pyoracle.build_ref()
    n_ms, prns, hints = 15000, [5, 14, 20, 30], [900, 4000, -1000, 2000]
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_steps.so"))
    C.CDLL("libc.so.6").srand(1)
    stream = synth.four_sv_with_lnav(n_ms, seed=7)
    crcs, checkpoints, final = sd.run_scenario(sd.StepsLib(lib, True), stream, prns, hints, n_ms, digest=True)
    path = os.path.join(ROOT, "tests", "golden", "f7_steps_lnav.npz")
    np.savez_compressed(path, crcs=crcs, checkpoints=checkpoints, final=final, prns=np.array(prns), hints=np.array(hints),
                        n_ms=np.int32(n_ms), stream_fnv=np.uint32(fnv1a32(stream[::97])))
    print("lnav", os.path.getsize(path), "bytes")
    nav = final[:, 212:324]
    for i in range(4):
        print("  PRN", prns[i], "inv_polarity", nav[i, 13], "polarity_found", nav[i, 14], "word_cnt", nav[i, 46],
              "words ok", int(nav[i, 56:60].view("<u4")[0]), "subframes", int(nav[i, 68:70].view("<u2")[0]),
              "last_subframe_time", int(nav[i, 60:64].view("<u4")[0]))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "lnav":
    gen_lnav()


def ephemeris_cases():
    """Subframe images (38 bytes, bit n of the subframe = bit n & 7 of byte n >> 3) fed to gps_nav_data_decode_subframe one
    after the other on ONE channel record (the decoder accumulates: subframe 2 uses subframe 1's week): random payloads
    of every subframe ID incl. the IDs it ignores, all-ones and all-zeros fields."""
    rng = np.random.Generator(np.random.PCG64(77))
    imgs = []
    for rep in range(6):
        for sub_id in (1, 2, 3, 4, 5, 0, 7, 3, 2, 1):
            bits = rng.integers(0, 2, 304).astype(np.uint8)
            if rep == 4:
                bits[:] = 1
            if rep == 5:
                bits[:] = 0
            bits[49:52] = [(sub_id >> 2) & 1, (sub_id >> 1) & 1, sub_id & 1]
            imgs.append(np.packbits(bits, bitorder="little")[:38])
    return np.array(imgs, np.uint8)


def gen_ephemeris():
    pyoracle.build_ref()
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_steps.so"))
    imgs = ephemeris_cases()
    ids, snaps = sd.run_ephemeris(lib, imgs)
    path = os.path.join(ROOT, "tests", "golden", "f8_ephemeris.npz")
    np.savez_compressed(path, imgs=imgs, ids=ids, snaps=snaps)
    print("ephemeris", os.path.getsize(path), "bytes; ids", ids[:10], "eccentricity of case 1:",
          snaps[1, 88:96].view("<f8")[0])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ephemeris":
    gen_ephemeris()



def gen_config5():
    """Golden for the batched step beyond four channels: 64 channels (tests/steps_driver.py config5_64ch_scenario), the
    reference's gps_tracking_process on one private library instance per channel (file-static slot buffers), all of them
    called millisecond by millisecond in channel order, so that the one thing they share -- libc's rand(), drawn by the
    false-lock reseed -- is drawn in the order a single-threaded loop over the channels draws it.  1500 ms; kept: a CRC of
    every channel's 226 state bytes per millisecond, full snapshots every 100 ms and at the end, and which channels
    reseeded when (a jump of if_freq_offset_hz by more than 150 Hz between two milliseconds)."""
    import shutil
    import tempfile
    pyoracle.build_ref()
    n_ms = sd.CONFIG5_MS
    sats, chans, seed = sd.config5_64ch_scenario()
    stream = synth.make_if(n_ms, sats, noise_amp=1.0, seed=seed)
    n = len(chans)
    crcs = np.zeros((n_ms, n), np.uint32)
    checkpoints = np.zeros((n_ms // 100, n, sd.SNAP), np.uint8)
    reseed_ms = []
    src = os.path.join(ROOT, "oracle", "_ref", "libref_steps.so")
    with tempfile.TemporaryDirectory() as td:
        insts = []
        table = np.zeros((n, sd.CH_SIZE), np.uint8)
        for c in range(n):
            path = os.path.join(td, f"ref_copy_{c}.so")
            shutil.copy(src, path)
            insts.append(sd.StepsLib(C.CDLL(path), True))
            table[c] = sd.preset_channel(insts[c], *chans[c])
        C.CDLL("libc.so.6").srand(1)
        prev = np.zeros(n, np.float32)
        for t in range(n_ms):
            blk = np.ascontiguousarray(stream[t])
            for c in range(n):
                insts[c].set_time(t)
                insts[c].lib.gps_tracking_process(table[c].ctypes.data, blk.ctypes.data, t & 3)
            freq = table[:, 64:68].copy().view("<f4")[:, 0]
            for c in np.flatnonzero((np.abs(freq - prev) > 150) & (t > 0)):
                reseed_ms.append((t, int(c)))
            prev = freq
            crcs[t] = sd.snapshot_crcs(table)
            if (t + 1) % 100 == 0:
                checkpoints[(t + 1) // 100 - 1] = sd.snapshot(table)
        final = sd.snapshot(table)
    path = os.path.join(ROOT, "tests", "golden", "f7_steps_config5_64ch.npz")
    np.savez_compressed(path, crcs=crcs, checkpoints=checkpoints, final=final, chans=np.array(chans, np.int32),
                        reseeds=np.array(reseed_ms, np.int32), n_ms=np.int32(n_ms),
                        stream_fnv=np.uint32(fnv1a32(stream[::97])))
    state = final[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0]
    print("config5", os.path.getsize(path), "bytes;", len(reseed_ms), "reseeds on", len({c for _, c in reseed_ms}),
          "channels; tracking:", int((state == sd.TRK_RUN).sum()), "of", n, "; not tracking:", np.flatnonzero(state != sd.TRK_RUN))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "config5":
    gen_config5()



def gen_config5_literal():
    """SURVEY.md 8(d) config 5 to the letter through the reference: 256 channels on 256 signals, 10 000 ms, one private
    instance of the reference's step sources per channel, channel order within each millisecond (rand() as in
    gen_config5).  Kept: a CRC of every channel's state every 100 ms, the final snapshots, the reference's lock mask."""
    import shutil
    import tempfile
    pyoracle.build_ref()
    n_ms = sd.CONFIG5_LITERAL_MS
    stream, chans, dopp, delay = sd.config5_literal_scenario(n_ms)
    n = len(chans)
    crcs = np.zeros((n_ms // 100, n), np.uint32)
    src = os.path.join(ROOT, "oracle", "_ref", "libref_steps.so")
    with tempfile.TemporaryDirectory() as td:
        insts = []
        table = np.zeros((n, sd.CH_SIZE), np.uint8)
        for c in range(n):
            path = os.path.join(td, f"ref_copy_{c}.so")
            shutil.copy(src, path)
            insts.append(sd.StepsLib(C.CDLL(path), True))
            table[c] = sd.preset_channel(insts[c], *chans[c])
        C.CDLL("libc.so.6").srand(1)
        for t in range(n_ms):
            blk = stream[t]
            for c in range(n):
                insts[c].set_time(t)
                insts[c].lib.gps_tracking_process(table[c].ctypes.data, blk.ctypes.data, t & 3)
            if (t + 1) % 100 == 0:
                crcs[(t + 1) // 100 - 1] = sd.snapshot_crcs(table)
        final = sd.snapshot(table)
        locked = sd.lock_mask(table, dopp, delay)
    path = os.path.join(ROOT, "tests", "golden", "f7_steps_config5_256ch.npz")
    np.savez_compressed(path, crcs=crcs, final=final, locked=locked, chans=np.array(chans, np.int32), n_ms=np.int32(n_ms),
                        stream_fnv=np.uint32(fnv1a32(stream[::97])))
    state = final[:, 60 + 148:60 + 152].copy().view("<i4")[:, 0]
    print("config5 literal", os.path.getsize(path), "bytes; tracking", int((state == sd.TRK_RUN).sum()), "locked", int(locked.sum()),
          "of", n, "; not locked:", np.flatnonzero(~locked).tolist())


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "config5_literal":
    gen_config5_literal()
