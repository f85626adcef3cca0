#!/usr/bin/env python3
"""oracle/gen_golden.py -- TEST INFRASTRUCTURE.  Regenerates tests/golden/*.npz|json.

Every EXPECTED value written here is produced by the reference's own C, compiled in place from /root/reference
into oracle/_ref/ (oracle/Makefile, target `ref`): gps_misc.c + common_ram.c (libref_pm.so) and the single-satellite
simulator (libref_ss_sim.so).  Inputs are either that simulator's block or this repo's seeded synthetic IF generator.
Run in the build container only (the GPU box has no /root/reference):

    python oracle/gen_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from stm32f4_sdr_gps_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
IF_HZ = pyoracle.IF_HZ


def fnv1a32(buf) -> int:
    h = 0x811C9DC5
    for b in np.asarray(buf).view(np.uint8).reshape(-1).tolist():
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h


def main():
    if not pyoracle.build_ref():
        raise SystemExit("needs /root/reference")
    ref = pyoracle.RefPM()
    sim = pyoracle.RefSSSim()
    os.makedirs(OUT, exist_ok=True)
    meta = {}

    # ---- F1: C/A codes -------------------------------------------------------------------------------------
    chips = np.stack([ref.ca_code(p) for p in range(1, 33)])
    fnv = np.array([fnv1a32(ref.ca_code(p)) for p in range(1, 211)], np.uint32)
    np.savez_compressed(os.path.join(OUT, "f1_ca_codes.npz"), chips_1_32=chips, fnv_1_210=fnv)
    meta["f1_first10_octal"] = {str(p): oct(int("".join(map(str, chips[p - 1][:10])), 2))[2:] for p in range(1, 33)}

    # ---- F2: wipe-off ----------------------------------------------------------------------------------------
    rng = np.random.default_rng(20240901)
    blocks = rng.integers(0, 256, (3, 2046), dtype=np.uint8)
    dopp = np.arange(-7000, 7001, 500, dtype=np.int32)
    fi = np.zeros((3, len(dopp)), np.uint32)
    fq = np.zeros_like(fi)
    for k in range(3):
        for j, d in enumerate(dopp):
            di, dq = ref.wipeoff(blocks[k], float(IF_HZ + int(d)))
            fi[k, j] = fnv1a32(di[:1022])
            fq[k, j] = fnv1a32(dq[:1022])
    ex_i, ex_q = ref.wipeoff(blocks[0], float(IF_HZ + 900))
    # stateful sequence: 20 ms with a changing float offset, accumulators recorded
    offs = rng.uniform(-5000, 5000, 20).astype(np.float32)
    acc = 0x12345678
    accs = []
    fti = []
    for ms in range(20):
        di, dq, acc = ref.wipeoff_track(blocks[ms % 3], float(offs[ms]), acc)
        accs.append(acc)
        fti.append([fnv1a32(di[:1022]), fnv1a32(dq[:1022])])
    rew = [ref.rewind(float(offs[i]), int(accs[i]), s) for i, s in enumerate([0, 1, 13, 255, 7] * 4)]
    np.savez_compressed(os.path.join(OUT, "f2_wipeoff.npz"), blocks=blocks, doppler_hz=dopp, fnv_i=fi, fnv_q=fq,
                        example_i=ex_i, example_q=ex_q, track_offsets=offs, track_acc0=np.uint32(0x12345678),
                        track_accs=np.array(accs, np.uint32), track_fnv=np.array(fti, np.uint32),
                        rewind_steps=np.array([0, 1, 13, 255, 7] * 4, np.uint8), rewind_out=np.array(rew, np.uint32))

    # ---- F3: replica ------------------------------------------------------------------------------------------
    rep = np.zeros((3, 16, 1024), np.uint16)
    for a, prn in enumerate((1, 5, 32)):
        for b in range(16):
            rep[a, b] = ref.replica(ref.ca_code(prn), b)
    np.savez_compressed(os.path.join(OUT, "f3_replica.npz"), prns=np.array([1, 5, 32]), replica=rep)

    # ---- F4/F5: correlation planes + search triplets on the seeded 4-SV stream -------------------------------
    stream = synth.default_four_sv(8, seed=7)
    blk = stream[5]
    full = [(5, 900), (14, 4000), (1, -5000)]
    hashed = [(20, -1000), (30, 2000), (1, 0), (32, 7000), (7, -7000)]
    cnt_i = np.zeros((len(full), 8, 2047), np.uint16)
    cnt_q = np.zeros_like(cnt_i)
    corr8 = np.zeros((len(full), 8, 2047), np.int16)
    search = []
    for a, (prn, d) in enumerate(full):
        di, dq = ref.wipeoff(blk, float(IF_HZ + d))
        for b in range(8):
            r = ref.replica(ref.ca_code(prn), b)
            for o in range(2047):
                cnt_i[a, b, o], cnt_q[a, b, o] = ref.mult_and_summ(di, dq, r, o)
                corr8[a, b, o] = ref.correlation8(r, di, dq, o)
            for (s0, s1) in [(0, 2046), (0, 500), (250, 750), (1990, 2046), (170, 230), (7, 8)]:
                mx, av, ph = ref.correlation_search(r, di, dq, s0, s1)
                search.append([prn, d, b, s0, s1, mx, av, ph])
    hashes = []
    for (prn, d) in hashed:
        di, dq = ref.wipeoff(blk, float(IF_HZ + d))
        for b in range(8):
            r = ref.replica(ref.ca_code(prn), b)
            c8 = np.array([ref.correlation8(r, di, dq, o) for o in range(2046)], np.int16)
            mx, av, ph = ref.correlation_search(r, di, dq, 0, 2046)
            hashes.append([prn, d, b, fnv1a32(c8), mx, av, ph])
    np.savez_compressed(os.path.join(OUT, "f4_corr.npz"), stream=stream, block_index=np.int32(5),
                        full_cases=np.array(full, np.int32), cnt_i=cnt_i, cnt_q=cnt_q, corr8=corr8,
                        search=np.array(search, np.int32), hashed=np.array(hashes, np.int64))

    # ---- F6: config 1 = the reference's commented self-test (SS/main.c:59-69) ---------------------------------
    f6 = {}
    chips1 = ref.ca_code(1)
    rep1 = ref.replica(chips1, 0)
    sim_blocks = []
    for noise in (0, 15, 30, 45):
        b = sim.block(noise, srand_seed=1)
        sim_blocks.append(b)
        di, dq = ref.wipeoff(b, float(IF_HZ + 2000))
        mx, av, ph = ref.correlation_search(rep1, di, dq, 0, 2046)
        iq = ref.correlation_iq(rep1, di, dq, 100)
        f6[str(noise)] = dict(max=mx, avr=av, phase=ph, i_at_100=iq[0], q_at_100=iq[1])
    np.savez_compressed(os.path.join(OUT, "f6_config1.npz"), noise_levels=np.array([0, 15, 30, 45]),
                        blocks=np.stack(sim_blocks))
    meta["f6_config1"] = f6
    assert f6["0"] == dict(max=7904, avr=65, phase=100, i_at_100=32, q_at_100=7904), f6["0"]

    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("golden vectors written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn:24s} {os.path.getsize(os.path.join(OUT, fn)):8d} B")


if __name__ == "__main__":
    main()
