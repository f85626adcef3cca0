"""Diagnostic (GPU box): the IF-samples-to-position chain of tests/test_gpu_pvt_chain.py with a status line per channel every 3 s
(code phase / Doppler error against the orbit, SNR, bit sync, polarity, words, subframes, stamps, ephemeris mask) and every
position fix with its error.  usage: diag_pvt_chain.py [n_ms [satellite seed]]"""
import ctypes as C, os, sys, math
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pvt_chain as pc
from pvt_types import Sol, Obsd, geodetic_to_ecef
from stm32f4_sdr_gps_amd import capi
RX = geodetic_to_ecef(48.1374, 11.5755, 520.0)
TOW0 = 388800 + 30 * 37
N_MS = int(sys.argv[1]) if len(sys.argv) > 1 else 26500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 29
sats = pc.pick_satellites(RX, TOW0, 4, seed=seed)
stream, first = pc.make_if_from_orbits(N_MS, sats, RX, TOW0, cycle=3)
print("first", first)
lib = capi.load_library()
lib.gps_master_handling.argtypes = [C.c_void_p, C.c_uint8]
lib.acquisition_process.argtypes = [C.c_void_p, C.c_void_p]
lib.gps_tracking_process.argtypes = [C.c_void_p, C.c_void_p, C.c_uint8]
lib.gpsx_compat_set_packet_cnt.argtypes = [C.c_uint32]
lib.gps_fill_summ_table()
table = (pc.GpsCh * 4)()
for i, (raw, row) in enumerate(sats):
    table[i].prn = row["sat"]
    table[i].acq_data.given_freq_offset_hz = int(round(first[i][0])) or 1
    lib.gps_channell_prepare(C.byref(table[i]))
lib.gps_pos_solve_init(table)
sol = Sol.in_dll(lib, "gps_sol")
lib.gpsx_compat_set_packet_cnt(0)
lib.gps_master_handling(table, 0)
for t in range(N_MS):
    lib.gpsx_compat_set_packet_cnt(t)
    data = stream[t].ctypes.data
    if lib.gps_master_need_acq():
        lib.acquisition_process(table, data)
        lib.gps_master_handling(table, 0)
        if not lib.gps_master_need_acq(): print("acquired at", t, [(c.acq_data.found_freq_offset_hz, c.acq_data.found_code_phase) for c in table])
    else:
        big = t % 17
        sat = big // 4 if big < 16 else 0
        index = 0xFF if big == 16 else big % 4
        lib.gps_tracking_process(C.byref(table[sat]), data, index)
        was = lib.solving_is_busy()
        lib.gps_master_handling(table, index)
        if index == 0xFF and not was and lib.solving_is_busy():
            print("FIX at", t, "err m", np.linalg.norm(np.array(list(sol.rr)[:3]) - RX), "dtr", sol.dtr[0], "P", [round(c.obs_data.pseudorange_m) for c in table], "tow", [round(c.obs_data.tow_s, 4) for c in table])
    if t % 3000 == 2999:
        for i, ch in enumerate(table):
            tau, dts, _ = pc.travel_time(sats[i][1], RX, np.array([TOW0 + t * 1e-3, TOW0 + (t + 1) * 1e-3]))
            lag = tau - dts
            want = (lag[0] * 1e3 % 1.0) * 16368.0
            err = (ch.tracking_data.code_phase_fine - want + 8184.0) % 16368.0 - 8184.0
            ferr = ch.tracking_data.if_freq_offset_hz + pc.F_L1 * (lag[1] - lag[0]) / 1e-3
            n = ch.nav_data
            print(t, "ch", i, "st", ch.tracking_data.state, "perr %.2f ferr %.1f snr %.1f" % (err, ferr, ch.tracking_data.snr_value), "sync", n.period_sync_ok_flag, "inv", n.inv_polarity_flag, "polfound", n.polarity_found, "words", n.word_cnt_test, "wc", n.word_cnt, "subfr", n.subframe_cnt, "last", n.last_subframe_time, "first", n.first_subframe_time, "mask", ch.eph_data.received_mask_proc, "swapflag", ch.tracking_data.code_phase_swap_flag, "filtcnt", ch.tracking_data.code_filt_cnt)
