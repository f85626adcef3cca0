"""Physics for the end-to-end test of the pseudorange step (csrc/gpsx_nav_master.cpp, PARITY UNPINNED -- the reference's
gps_master.c cannot be compiled in place): satellites on broadcast orbits, the signal a receiver at a chosen position would
see from them (code, carrier and LNAV data delayed by the true travel time, millisecond by millisecond), and the LNAV
encoding of their ephemerides (IS-GPS-200 20.3.3.3-20.3.3.4: the inverse of csrc/gpsx_ephemeris.cpp's field tables).
Test infrastructure; nothing here is product code."""
import math

import numpy as np

from pvt_types import CLIGHT, MU, OMGE, geodetic_to_ecef

F_L1 = 1575.42e6
SC = 3.1415926535898           # IS-GPS-200's pi
WEEK = 2300                    # 10-bit week 252; the decoder resolves the roll-over around its build week 2290
SCALE = {4: 16.0, -5: 0.03125, -19: 1.907348632812500E-06, -29: 1.862645149230957E-09, -31: 4.656612873077393E-10,
         -33: 1.164153218269348E-10, -43: 1.136868377216160E-13, -55: 2.775557561562891E-17}   # the decoder's literals

# field -> (scale exponent, semicircles, signed, bits); raw integers are what is transmitted
FIELDS = {"crs": (-5, False, True, 16), "deln": (-43, True, True, 16), "M0": (-31, True, True, 32), "cuc": (-29, False, True, 16),
          "e": (-33, False, False, 32), "cus": (-29, False, True, 16), "sqrtA": (-19, False, False, 32),
          "toes": (4, False, False, 16), "cic": (-29, False, True, 16), "OMG0": (-31, True, True, 32),
          "cis": (-29, False, True, 16), "i0": (-31, True, True, 32), "crc": (-5, False, True, 16),
          "omg": (-31, True, True, 32), "OMGd": (-43, True, True, 24), "idot": (-43, True, True, 14),
          "tgd0": (-31, False, True, 8), "toc": (4, False, False, 16), "f2": (-55, False, True, 8),
          "f1": (-43, False, True, 16), "f0": (-31, False, True, 22)}


def quantize(row):
    """Broadcast elements (pvt_types rows, radians) -> (raw integers as transmitted, the row a decoder gets back)."""
    src = dict(row, sqrtA=math.sqrt(row["A"]), toc=row["toes"])
    raw, out = {}, dict(row)
    for k, (exp, semi, signed, bits) in FIELDS.items():
        v = src[k] / (SC if semi else 1.0) / SCALE[exp]
        q = int(round(v))
        lo, hi = (-(1 << (bits - 1)), (1 << (bits - 1)) - 1) if signed else (0, (1 << bits) - 1)
        assert lo <= q <= hi, (k, q)
        raw[k] = q
        val = float(q) * SCALE[exp]
        if semi:
            val = val * SC
        out[k] = val
    out["A"] = out.pop("sqrtA") ** 2
    del out["toc"]
    for k in ("iode", "iodc", "sva", "svh"):
        raw[k] = int(row[k])
    return raw, out


def _put(bits, pos, length, value):
    for i in range(length):
        bits[pos + i] = (value >> (length - 1 - i)) & 1


def _put2(bits, p1, l1, p2, l2, value):
    _put(bits, p1, l1, (value >> l2) & ((1 << l1) - 1))
    _put(bits, p2, l2, value & ((1 << l2) - 1))


def subframe_payloads(raw):
    """{1, 2, 3} -> eight words x 24 source bits (words 3..10) carrying the raw fields at the positions the decoder reads
    (0-based transmitted-bit positions, 30 per word)."""
    out = {}
    m = lambda v, n: v & ((1 << n) - 1)     # two's complement into n bits
    b = [0] * 300
    _put(b, 60, 10, WEEK % 1024), _put(b, 70, 2, 1), _put(b, 72, 4, raw["sva"]), _put(b, 76, 6, raw["svh"])
    _put(b, 82, 2, raw["iodc"] >> 8), _put(b, 196, 8, m(raw["tgd0"], 8)), _put(b, 210, 8, raw["iodc"] & 255)
    _put(b, 218, 16, raw["toc"]), _put(b, 240, 8, m(raw["f2"], 8)), _put(b, 248, 16, m(raw["f1"], 16))
    _put(b, 270, 22, m(raw["f0"], 22))
    out[1] = b
    b = [0] * 300
    _put(b, 60, 8, raw["iode"]), _put(b, 68, 16, m(raw["crs"], 16)), _put(b, 90, 16, m(raw["deln"], 16))
    _put2(b, 106, 8, 120, 24, m(raw["M0"], 32)), _put(b, 150, 16, m(raw["cuc"], 16)), _put2(b, 166, 8, 180, 24, raw["e"])
    _put(b, 210, 16, m(raw["cus"], 16)), _put2(b, 226, 8, 240, 24, raw["sqrtA"]), _put(b, 270, 16, raw["toes"])
    out[2] = b
    b = [0] * 300
    _put(b, 60, 16, m(raw["cic"], 16)), _put2(b, 76, 8, 90, 24, m(raw["OMG0"], 32)), _put(b, 120, 16, m(raw["cis"], 16))
    _put2(b, 136, 8, 150, 24, m(raw["i0"], 32)), _put(b, 180, 16, m(raw["crc"], 16))
    _put2(b, 196, 8, 210, 24, m(raw["omg"], 32)), _put(b, 240, 24, m(raw["OMGd"], 24)), _put(b, 270, 8, raw["iode"])
    _put(b, 278, 14, m(raw["idot"], 14))
    out[3] = b
    return {k: [v[30 * w:30 * w + 24] for w in range(2, 10)] for k, v in out.items()}


def lnav_stream(raw, first_subframe_index, n_subframes, seed, cycle=5):
    """0/1 navigation bits of n_subframes whole subframes, the first one being subframe number first_subframe_index of the
    week (it starts at GPS time 6 * index; its ID is index mod 5 + 1, its hand-over word announces index + 1).  cycle = 3: a
    test signal whose frames consist of subframes 1, 2, 3 only, so that a receiver has a whole ephemeris after any three."""
    from stm32f4_sdr_gps_amd import synth
    rng = np.random.Generator(np.random.PCG64(seed))
    pay = subframe_payloads(raw)
    bits = []
    for k in range(first_subframe_index, first_subframe_index + n_subframes):
        sub_id = k % cycle + 1
        bits += synth.lnav_subframe(sub_id, k + 1, rng, pay.get(sub_id))
    return np.array(bits, np.uint8)


# ---- the signal in space ------------------------------------------------------------------------------------------------
def sat_state(row, t):
    """ECEF position (at the instant t, in the frame of t) and clock offset (polynomial + relativity) of a satellite at GPS
    time-of-week t (numpy array): IS-GPS-200 table 20-IV."""
    t = np.asarray(t, np.float64)
    tk = t - row["toes"]
    n = math.sqrt(MU / row["A"] ** 3) + row["deln"]
    M = row["M0"] + n * tk
    E = M.copy()
    for _ in range(12):
        E = E - (E - row["e"] * np.sin(E) - M) / (1 - row["e"] * np.cos(E))
    u = np.arctan2(math.sqrt(1 - row["e"] ** 2) * np.sin(E), np.cos(E) - row["e"]) + row["omg"]
    r = row["A"] * (1 - row["e"] * np.cos(E))
    inc = row["i0"] + row["idot"] * tk
    s2, c2 = np.sin(2 * u), np.cos(2 * u)
    u = u + row["cus"] * s2 + row["cuc"] * c2
    r = r + row["crs"] * s2 + row["crc"] * c2
    inc = inc + row["cis"] * s2 + row["cic"] * c2
    x, y = r * np.cos(u), r * np.sin(u)
    O = row["OMG0"] + (row["OMGd"] - OMGE) * tk - OMGE * row["toes"]
    pos = np.stack([x * np.cos(O) - y * np.cos(inc) * np.sin(O), x * np.sin(O) + y * np.cos(inc) * np.cos(O), y * np.sin(inc)], -1)
    dts = row["f0"] + row["f1"] * tk + row["f2"] * tk * tk - 2.0 * math.sqrt(MU * row["A"]) * row["e"] * np.sin(E) / CLIGHT ** 2
    return pos, dts


def _geodetic(r):
    a, f = 6378137.0, 1 / 298.257223563
    e2 = f * (2 - f)
    p = math.hypot(r[0], r[1])
    lat = math.atan2(r[2], p * (1 - e2))
    for _ in range(8):
        v = a / math.sqrt(1 - e2 * math.sin(lat) ** 2)
        lat = math.atan2(r[2] + v * e2 * math.sin(lat), p)
    v = a / math.sqrt(1 - e2 * math.sin(lat) ** 2)
    return lat, math.atan2(r[1], r[0]), p / math.cos(lat) - v


def _azel(rx, geo, sat):
    los = sat - rx
    los = los / np.linalg.norm(los, axis=-1, keepdims=True)
    sp, cp, sl, cl = math.sin(geo[0]), math.cos(geo[0]), math.sin(geo[1]), math.cos(geo[1])
    e = -sl * los[..., 0] + cl * los[..., 1]
    n = -sp * cl * los[..., 0] - sp * sl * los[..., 1] + cp * los[..., 2]
    u = cp * cl * los[..., 0] + cp * sl * los[..., 1] + sp * los[..., 2]
    return np.mod(np.arctan2(e, n), 2 * math.pi), np.arcsin(u)


def _klobuchar(tow, geo, az, el):
    """broadcast ionosphere model with the 2004 default coefficients (what the solver assumes when none were broadcast)"""
    ion = (0.1118E-07, -0.7451E-08, -0.5961E-07, 0.1192E-06, 0.1167E+06, -0.2294E+06, -0.1311E+06, 0.1049E+07)
    psi = 0.0137 / (el / math.pi + 0.11) - 0.022
    phi = np.clip(geo[0] / math.pi + psi * np.cos(az), -0.416, 0.416)
    lam = geo[1] / math.pi + psi * np.sin(az) / np.cos(phi * math.pi)
    phi = phi + 0.064 * np.cos((lam - 1.617) * math.pi)
    tt = np.mod(43200.0 * lam + tow, 86400.0)
    slant = 1.0 + 16.0 * (0.53 - el / math.pi) ** 3
    amp = np.maximum(ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3])), 0.0)
    per = np.maximum(ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7])), 72000.0)
    x = 2.0 * math.pi * (tt - 50400.0) / per
    return CLIGHT * slant * np.where(np.abs(x) < 1.57, 5E-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)), 5E-9)


def _saastamoinen(geo, el, humidity=0.7):
    hgt = max(geo[2], 0.0)
    pres = 1013.25 * (1.0 - 2.2557E-5 * hgt) ** 5.2568
    temp = 15.0 - 6.5E-3 * hgt + 273.16
    e = 6.108 * humidity * math.exp((17.15 * temp - 4684.0) / (temp - 38.45))
    z = math.pi / 2.0 - el
    dry = 0.0022768 * pres / (1.0 - 0.00266 * math.cos(2.0 * geo[0]) - 0.00028 * hgt / 1E3) / np.cos(z)
    return dry + 0.002277 * (1255.0 / temp + 0.05) * e / np.cos(z)


def travel_time(row, rx, t_rx, atmosphere=True):
    """Seconds a signal received at GPS time-of-week t_rx (array) at ECEF rx has travelled, the satellite's clock offset at
    transmission, and its elevation: geometric range in the frame of reception (earth rotation during the flight) plus --
    when asked -- the ionosphere and troposphere the solver will model away."""
    t_rx = np.asarray(t_rx, np.float64)
    geo = _geodetic(rx)
    tau = np.full_like(t_rx, 0.075)
    for _ in range(5):
        pos, dts = sat_state(row, t_rx - tau)
        th = OMGE * tau
        rot = np.stack([np.cos(th) * pos[..., 0] + np.sin(th) * pos[..., 1], -np.sin(th) * pos[..., 0] + np.cos(th) * pos[..., 1],
                        pos[..., 2]], -1)
        az, el = _azel(rx, geo, rot)
        rng = np.linalg.norm(rot - rx, axis=-1)
        if atmosphere:
            rng = rng + _klobuchar(t_rx, geo, az, el) + _saastamoinen(geo, el)
        tau = rng / CLIGHT
    return tau, dts, el


def pick_satellites(rx, tow, n, seed, min_el_deg=25.0):
    """n synthetic satellites on GPS-like orbits, all above min_el_deg at rx around time-of-week tow and spread in azimuth;
    rows already quantized to what LNAV can carry.  Returns [(raw integers, row)]."""
    rng = np.random.default_rng(seed)
    toes = float(int(tow) // 7200 * 7200 + 3600)
    geo = _geodetic(rx)
    out, prn = [], 0
    while len(out) < n:
        row = dict(sat=0, iode=int(rng.integers(1, 255)), iodc=0, sva=0, svh=0, week=WEEK,
                   A=26559710.0 + rng.normal(0, 3e3), e=float(rng.uniform(0.002, 0.012)), i0=0.96 + rng.normal(0, 0.02),
                   OMG0=float(rng.uniform(-math.pi, math.pi)), omg=float(rng.uniform(-math.pi, math.pi)),
                   M0=float(rng.uniform(-math.pi, math.pi)), deln=float(rng.normal(4.5e-9, 5e-10)),
                   OMGd=float(rng.normal(-8.0e-9, 3e-10)), idot=float(rng.normal(0, 2e-10)),
                   crc=float(rng.normal(220, 60)), crs=float(rng.normal(0, 60)), cuc=float(rng.normal(0, 3e-6)),
                   cus=float(rng.normal(6e-6, 3e-6)), cic=float(rng.normal(0, 1e-7)), cis=float(rng.normal(0, 1e-7)),
                   toes=toes, f0=float(rng.normal(0, 2e-4)), f1=float(rng.normal(0, 5e-12)), f2=0.0, tgd0=0.0)
        row["iodc"] = row["iode"]
        pos, _ = sat_state(row, np.array([float(tow)]))
        az, el = _azel(rx, geo, pos)
        if math.degrees(el[0]) < min_el_deg:
            continue
        if any(abs((az[0] - a + math.pi) % (2 * math.pi) - math.pi) < 2 * math.pi / (n + 2) for a, _ in [(o[2], 0) for o in out]):
            continue
        prn += 1 + int(rng.integers(0, 4))
        row["sat"] = prn
        raw, q = quantize(row)
        out.append((raw, q, float(az[0])))
    return [(raw, q) for raw, q, _ in out]


def make_if_from_orbits(n_ms, sats, rx, tow0, amp=0.6, noise_amp=1.0, seed=7, nav_seed=100, cycle=5):
    """1-bit IF blocks [n_ms, 2046] a receiver at ECEF rx records from GPS time-of-week tow0 on (its clock IS GPS time):
    per satellite (raw, row) code, LNAV data and carrier, all delayed by the travel time of that millisecond (linear inside
    it) and shifted by the satellite's own clock offset.  tow0 must be a multiple of 6 s (a subframe boundary at the
    satellites).  Also returns per satellite the Doppler (Hz) and code delay (samples into a block) at the first block."""
    from stm32f4_sdr_gps_amd import synth
    assert tow0 % 6 == 0
    t_edges = tow0 + np.arange(n_ms + 1) * 1e-3
    per_sat = []
    k0 = int(tow0) // 6 - 1                                   # one subframe earlier: signals in flight at tow0
    n_sub = n_ms // 6000 + 3
    for j, (raw, row) in enumerate(sats):
        tau, dts, _ = travel_time(row, rx, t_edges)
        bits = 1.0 - 2.0 * lnav_stream(raw, k0, n_sub, nav_seed + j, cycle).astype(np.float64)
        code = 1.0 - 2.0 * synth.ca_code(row["sat"]).astype(np.float64)
        per_sat.append((tau, dts, bits, code))
    frac = np.arange(synth.SAMPLES_PER_MS, dtype=np.float64) / synth.SAMPLES_PER_MS
    t_in_ms = frac * 1e-3
    quarter = (np.arange(synth.SAMPLES_PER_MS) % 4) * 0.25     # IF / fs = 1/4 cycle per sample, exactly
    out = np.zeros((n_ms, synth.BYTES_PER_MS), np.uint8)

    def chunk(m0m1):      # a run of milliseconds as [ms, sample] arrays: the same float64 arithmetic per sample whatever the
        m0, m1 = m0m1     # chunking (noise by position), large enough operations for the threads to run side by side
        x = synth.noise_generator(seed, m0 * synth.SAMPLES_PER_MS).uniform(-noise_amp, noise_amp, (m1 - m0, synth.SAMPLES_PER_MS))
        ms = np.array([m * 1e-3 for m in range(m0, m1)])[:, None]
        for tau, dts, bits, code in per_sat:
            lag = tau[m0:m1, None] + (tau[m0 + 1:m1 + 1] - tau[m0:m1])[:, None] * frac[None, :] - dts[m0:m1, None]   # reception time minus the satellite clock's reading
            t_sv = (ms + t_in_ms[None, :]) - lag                       # ... relative to tow0, seconds
            chip = np.floor(np.mod(t_sv, 1e-3) * 1.023e6).astype(np.int64) % synth.CHIPS
            bit = np.floor(t_sv / 0.02).astype(np.int64) + 300        # the stream began one subframe before tow0
            cyc = quarter[None, :] - F_L1 * lag
            x = x + amp * bits[bit] * code[chip] * np.cos(2.0 * math.pi * (cyc - np.floor(cyc)))
        out[m0:m1] = np.packbits(x >= 0, axis=1, bitorder="little")

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(synth._n_threads()) as ex:
        list(ex.map(chunk, synth._ms_chunks(n_ms, 8)))
    first = []
    for tau, dts, _, _ in per_sat:
        doppler = -F_L1 * ((tau[1] - dts[1]) - (tau[0] - dts[0])) / 1e-3
        delay = ((tau[0] - dts[0]) % 1e-3) * synth.FS_HZ
        first.append((float(doppler), float(delay)))
    return out, first


# ---- ctypes mirror of gps_ch_t (include/gpsx_compat.h; sizes checked against the library in the tests) -------------------
import ctypes as C  # noqa: E402

from pvt_types import Eph  # noqa: E402


class AcqData(C.Structure):
    _fields_ = [("freq_index", C.c_uint8), ("found_freq_offset_hz", C.c_int16), ("given_freq_offset_hz", C.c_int16),
                ("found_code_phase", C.c_uint16), ("code_search_start", C.c_uint16), ("code_search_stop", C.c_uint16),
                ("code_hist_step", C.c_uint16), ("state", C.c_int), ("code_phase_histogram", C.c_uint8 * 32),
                ("start_timestamp", C.c_uint32), ("hist_ratio", C.c_float)]


class TrackingData(C.Structure):
    _fields_ = [("code_search_start", C.c_uint16), ("code_search_stop", C.c_uint16), ("if_freq_offset_hz", C.c_float),
                ("if_freq_accum", C.c_uint32), ("pre_track_phases", C.c_uint16 * 30), ("pre_track_count", C.c_uint8),
                ("prev_track_timestamp", C.c_uint32), ("code_phase_fine", C.c_float), ("old_code_phase_fine", C.c_float),
                ("code_phase_swap_flag", C.c_uint8), ("dll_code_err", C.c_float), ("pll_code_err", C.c_float),
                ("fll_old_i", C.c_int16), ("fll_old_q", C.c_int16), ("fll_err", C.c_float), ("pll_check_buf", C.c_int16 * 4),
                ("pll_bad_state_cnt", C.c_uint8), ("pll_bad_state_master_cnt", C.c_uint16), ("i_part_summ", C.c_uint32),
                ("q_part_summ", C.c_uint32), ("snr_summ_cnt", C.c_uint16), ("snr_value", C.c_float),
                ("filt_start_time_ms", C.c_uint32), ("code_filt_cnt", C.c_uint16), ("code_phase_fine_filt", C.c_float),
                ("state", C.c_int)]


class NavData(C.Structure):
    _fields_ = [("period_sync_ok_flag", C.c_uint8), ("right_period_cnt", C.c_uint8), ("old_swap_time", C.c_uint32),
                ("old_reminder", C.c_uint8), ("accurate_swap_time", C.c_uint8), ("accurate_swap_ok", C.c_uint8),
                ("last_bit_pos_cnt", C.c_uint8), ("last_bit_neg_cnt", C.c_uint8), ("inv_polarity_flag", C.c_uint8),
                ("polarity_found", C.c_uint8), ("inv_preabmle_cnt", C.c_uint8), ("word_buf", C.c_uint8 * 30),
                ("word_cnt", C.c_uint8), ("word_bit_cnt", C.c_uint8), ("old_D29", C.c_uint8), ("old_D30", C.c_uint8),
                ("word_detection_timestamp", C.c_uint32), ("word_cnt_test", C.c_uint32), ("last_subframe_time", C.c_uint32),
                ("first_subframe_time", C.c_uint32), ("subframe_cnt", C.c_uint16), ("new_subframe_flag", C.c_uint8),
                ("subframe_data", C.c_uint8 * 38)]


class ObsData(C.Structure):
    _fields_ = [("pseudorange_m", C.c_double), ("tow_s", C.c_double)]


class SdrEph(C.Structure):
    _fields_ = [("eph", Eph), ("ctype", C.c_int), ("tow_gpst", C.c_double), ("week_gpst", C.c_int), ("cnt", C.c_int),
                ("cntth", C.c_int), ("update", C.c_int), ("prn", C.c_int), ("week_gst", C.c_int), ("sub_cnt", C.c_uint16),
                ("received_mask", C.c_uint8), ("received_mask_proc", C.c_uint8)]


class GpsCh(C.Structure):
    _fields_ = [("acq_data", AcqData), ("tracking_data", TrackingData), ("nav_data", NavData), ("obs_data", ObsData),
                ("eph_data", SdrEph), ("prn", C.c_uint8), ("prn_code", C.c_uint8 * 1023)]


assert C.sizeof(GpsCh) == 1688 and GpsCh.tracking_data.offset == 60 and GpsCh.nav_data.offset == 212
assert GpsCh.obs_data.offset == 328 and GpsCh.eph_data.offset == 344 and GpsCh.prn.offset == 664
assert NavData.last_subframe_time.offset == 60 and TrackingData.code_phase_fine_filt.offset == 144


def subframe_image(raw, sub_id, tow_count):
    """the 38-byte image the word layer hands the decoder: source bits at their transmitted positions (bit n of the subframe =
    bit n & 7 of byte n >> 3), hand-over word filled in"""
    words = subframe_payloads(raw)[sub_id]
    bits = np.zeros(304, np.uint8)
    for w in range(8):
        bits[30 * (w + 2):30 * (w + 2) + 24] = words[w]
    bits[30:47] = [(tow_count >> (16 - i)) & 1 for i in range(17)]
    bits[49:52] = [(sub_id >> 2) & 1, (sub_id >> 1) & 1, sub_id & 1]
    return np.packbits(bits, bitorder="little")[:38]
