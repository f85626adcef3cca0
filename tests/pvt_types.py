"""ctypes mirrors of the position-solver records (include/gpsx_compat.h = PM/GPS/RTK/rtk_common.h:49-58,104-108,
PM/GPS/RTK/solving.h:17-30, gps_misc.h:141-182) and a small scenario builder, shared by oracle/gen_golden_pvt.py (which
runs the reference's own solver, compiled in place, to record the expected answers) and tests/test_pvt.py."""
import ctypes as C
import math

import numpy as np

UNIX2GPS = 315964800
CLIGHT = 299792458.0
MU = 3.9860050E14
OMGE = 7.2921151467E-5


def load_lazy(path):
    """dlopen(path, RTLD_LAZY) through libc (ctypes.CDLL itself always adds RTLD_NOW): the reference's solver library has ONE
    unresolved function, the MCU cycle counter get_dwt_value() used by gps_pos_solve()'s stopwatch (solving.c:119-138);
    with lazy binding it is only looked up if called, and nothing here calls it."""
    libc = C.CDLL(None)
    libc.dlopen.restype = C.c_void_p
    libc.dlopen.argtypes = [C.c_char_p, C.c_int]
    handle = libc.dlopen(path.encode(), 1)   # RTLD_LAZY
    if not handle:
        raise OSError(f"dlopen({path}) failed")
    return C.CDLL(path, handle=handle)


class GTime(C.Structure):
    _fields_ = [("time", C.c_int64), ("sec", C.c_double)]


class Eph(C.Structure):
    _fields_ = [("sat", C.c_int), ("iode", C.c_int), ("iodc", C.c_int), ("sva", C.c_int), ("svh", C.c_int),
                ("week", C.c_int), ("code", C.c_int), ("flag", C.c_int),
                ("toe", GTime), ("toc", GTime), ("ttr", GTime),
                ("A", C.c_double), ("e", C.c_double), ("i0", C.c_double), ("OMG0", C.c_double), ("omg", C.c_double),
                ("M0", C.c_double), ("deln", C.c_double), ("OMGd", C.c_double), ("idot", C.c_double),
                ("crc", C.c_double), ("crs", C.c_double), ("cuc", C.c_double), ("cus", C.c_double), ("cic", C.c_double),
                ("cis", C.c_double), ("toes", C.c_double), ("fit", C.c_double),
                ("f0", C.c_double), ("f1", C.c_double), ("f2", C.c_double), ("tgd", C.c_double * 4)]


class Obsd(C.Structure):
    _fields_ = [("time", GTime), ("sat", C.c_ubyte), ("rcv", C.c_ubyte), ("SNR", C.c_ubyte * 1), ("LLI", C.c_ubyte * 1),
                ("code", C.c_ubyte * 1), ("L", C.c_double * 1), ("P", C.c_double * 1), ("D", C.c_float * 1)]


class Nav(C.Structure):
    _fields_ = [("n", C.c_int), ("eph", C.POINTER(Eph) * 4), ("ion_gps", C.c_double * 8)]


class Sol(C.Structure):
    _fields_ = [("time", GTime), ("rr", C.c_double * 6), ("qr", C.c_float * 6), ("dtr", C.c_double * 6),
                ("type", C.c_ubyte), ("stat", C.c_ubyte), ("ns", C.c_ubyte), ("age", C.c_float), ("ratio", C.c_float)]


EPH_FIELDS = ["sat", "iode", "iodc", "sva", "svh", "week", "A", "e", "i0", "OMG0", "omg", "M0", "deln", "OMGd", "idot",
              "crc", "crs", "cuc", "cus", "cic", "cis", "toes", "f0", "f1", "f2", "tgd0"]


def eph_from_row(row):
    """row: dict with EPH_FIELDS -> Eph (toe = toc = start of `week` + toes)."""
    e = Eph()
    for k in EPH_FIELDS:
        if k == "tgd0":
            e.tgd[0] = float(row[k])
        elif k in ("sat", "iode", "iodc", "sva", "svh", "week"):
            setattr(e, k, int(row[k]))
        else:
            setattr(e, k, float(row[k]))
    t = UNIX2GPS + 604800 * int(row["week"]) + int(row["toes"])
    e.toe = GTime(t, float(row["toes"]) - int(row["toes"]))
    e.toc = GTime(t, float(row["toes"]) - int(row["toes"]))
    e.ttr = e.toe
    e.fit = 4.0
    return e


def sat_ecef(row, tk):
    """Satellite position from broadcast elements, tk seconds after toe (generator-side geometry only: it decides which
    synthetic satellites are in view and what their pseudoranges are; it is not what the tests check against)."""
    n = math.sqrt(MU / row["A"] ** 3) + row["deln"]
    M = row["M0"] + n * tk
    E = M
    for _ in range(30):
        E = E - (E - row["e"] * math.sin(E) - M) / (1 - row["e"] * math.cos(E))
    u = math.atan2(math.sqrt(1 - row["e"] ** 2) * math.sin(E), math.cos(E) - row["e"]) + row["omg"]
    r = row["A"] * (1 - row["e"] * math.cos(E))
    i = row["i0"] + row["idot"] * tk
    s2, c2 = math.sin(2 * u), math.cos(2 * u)
    u += row["cus"] * s2 + row["cuc"] * c2
    r += row["crs"] * s2 + row["crc"] * c2
    i += row["cis"] * s2 + row["cic"] * c2
    x, y = r * math.cos(u), r * math.sin(u)
    O = row["OMG0"] + (row["OMGd"] - OMGE) * tk - OMGE * row["toes"]
    return np.array([x * math.cos(O) - y * math.cos(i) * math.sin(O), x * math.sin(O) + y * math.cos(i) * math.cos(O),
                     y * math.sin(i)])


def geodetic_to_ecef(lat_deg, lon_deg, h):
    a, f = 6378137.0, 1 / 298.257223563
    e2 = f * (2 - f)
    lat, lon = math.radians(lat_deg), math.radians(lon_deg)
    v = a / math.sqrt(1 - e2 * math.sin(lat) ** 2)
    return np.array([(v + h) * math.cos(lat) * math.cos(lon), (v + h) * math.cos(lat) * math.sin(lon),
                     (v * (1 - e2) + h) * math.sin(lat)])


def make_scenario(seed, lat, lon, h, week=2200, toes=345600.0, t_after=1800.25, clk_bias_s=3.1e-4, noise_m=0.0,
                  ecc_max=0.012):
    """Four synthetic satellites above 12 degrees at (lat, lon, h), their broadcast elements, and the pseudoranges a receiver
    with clock bias clk_bias_s would measure t_after seconds after toe.  Returns (eph_rows, tow_rx, pseudoranges)."""
    rng = np.random.default_rng(seed)
    rx = geodetic_to_ecef(lat, lon, h)
    up = rx / np.linalg.norm(rx)
    rows = []
    used = set()
    while len(rows) < 4:
        prn = int(rng.integers(1, 33))
        if prn in used:
            continue
        row = dict(sat=prn, iode=int(rng.integers(0, 256)), iodc=0, sva=int(rng.integers(0, 4)), svh=0, week=week,
                   A=26559710.0 + rng.normal(0, 3e3), e=float(rng.uniform(0.001, ecc_max)), i0=0.96 + rng.normal(0, 0.02),
                   OMG0=float(rng.uniform(-math.pi, math.pi)), omg=float(rng.uniform(-math.pi, math.pi)),
                   M0=float(rng.uniform(-math.pi, math.pi)), deln=float(rng.normal(4.5e-9, 5e-10)),
                   OMGd=float(rng.normal(-8.0e-9, 3e-10)), idot=float(rng.normal(0, 2e-10)),
                   crc=float(rng.normal(220, 60)), crs=float(rng.normal(0, 60)), cuc=float(rng.normal(0, 3e-6)),
                   cus=float(rng.normal(6e-6, 3e-6)), cic=float(rng.normal(0, 1e-7)), cis=float(rng.normal(0, 1e-7)),
                   toes=toes, f0=float(rng.normal(0, 2e-4)), f1=float(rng.normal(0, 5e-12)), f2=0.0,
                   tgd0=float(rng.normal(-8e-9, 4e-9)))
        row["iodc"] = row["iode"]
        los = sat_ecef(row, t_after) - rx
        el = math.degrees(math.asin(float(np.dot(los, up)) / float(np.linalg.norm(los))))
        if el < 12.0:
            continue
        if any(abs(math.atan2(*(np.cross(los, r2["_los"]) @ up, np.dot(los, r2["_los"])))) < 0.35 for r2 in rows):
            continue   # keep the satellites apart in azimuth-ish terms: a usable geometry
        row["_los"], row["_el"] = los, el
        used.add(prn)
        rows.append(row)
    tow_rx = toes + t_after
    prs = []
    for row in rows:
        tau = 0.075
        for _ in range(6):
            ps = sat_ecef(row, t_after - tau)
            th = OMGE * tau
            ps = np.array([math.cos(th) * ps[0] + math.sin(th) * ps[1], -math.sin(th) * ps[0] + math.cos(th) * ps[1], ps[2]])
            tau = float(np.linalg.norm(ps - rx)) / CLIGHT
        dts = row["f0"] + row["f1"] * (t_after - tau)
        atm = 2.4 / math.sin(math.radians(row["_el"])) + 4.0
        prs.append(tau * CLIGHT + CLIGHT * (clk_bias_s - dts) + atm + CLIGHT * row["tgd0"] + float(rng.normal(0, 1.0)) * noise_m)
    for row in rows:
        del row["_los"], row["_el"]
    return rows, tow_rx, prs


def build_inputs(rows, week, tow_rx, prs, ion=None, sats=None):
    """-> (ephs keepalive list, Nav, Obsd * 4)"""
    ephs = [eph_from_row(r) for r in rows]
    nav = Nav()
    nav.n = 4
    for i in range(4):
        nav.eph[i] = C.pointer(ephs[i])
    if ion is not None:
        for i in range(8):
            nav.ion_gps[i] = float(ion[i])
    obs = (Obsd * 4)()
    t = UNIX2GPS + 604800 * week + int(tow_rx)
    for i in range(4):
        obs[i].time = GTime(t, tow_rx - int(tow_rx))
        obs[i].sat = int(sats[i] if sats is not None else rows[i]["sat"])
        obs[i].rcv = 1
        obs[i].code[0] = 1
        obs[i].P[0] = float(prs[i])
    return ephs, nav, obs
