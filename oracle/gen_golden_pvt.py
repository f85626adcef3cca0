#!/usr/bin/env python3
"""Golden vectors for the position solution: (observations, ephemerides) -> what the REFERENCE's own
solver returns.  Runs only where /root/reference exists: oracle/_ref/libref_pvt.so is PM/GPS/RTK/solving.c +
rtklib_common.c compiled in place (oracle/Makefile), no stand-ins; pntpos() and ecef2pos() (solving.h:35-36) are called
directly.  Output: tests/golden/f9_pvt.npz (inputs as plain arrays + expected outputs)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pvt_types as T  # noqa: E402

CASES = [
    # name, scenario kwargs, ion parameters, start position ("zero" | "near" | "prev"), mutation
    ("moscow_cold", dict(seed=1, lat=55.75, lon=37.62, h=180.0), None, "zero", None),
    ("moscow_warm_noisy", dict(seed=1, lat=55.75, lon=37.62, h=180.0, noise_m=3.0), None, "near", None),
    ("quito_broadcast_iono", dict(seed=2, lat=-0.2, lon=-78.5, h=2850.0, clk_bias_s=-7.7e-4),
     [1.1e-8, 2.2e-8, -5.9e-8, -1.2e-7, 9.0e4, 1.3e5, -6.5e4, -5.2e5], "zero", None),
    ("sydney_eccentric", dict(seed=3, lat=-33.87, lon=151.21, h=40.0, ecc_max=0.02, t_after=5400.5, noise_m=1.5), None, "zero", None),
    ("tromso_late_in_fit", dict(seed=4, lat=69.65, lon=18.96, h=10.0, t_after=7100.0, clk_bias_s=1.0e-6), None, "zero", None),
    ("week_edge", dict(seed=5, lat=35.0, lon=139.0, h=900.0, toes=604784.0, t_after=10.5), None, "zero", None),
    ("unhealthy_satellite", dict(seed=6, lat=48.1, lon=11.6, h=520.0), None, "zero", "svh"),
    ("duplicated_observation", dict(seed=7, lat=48.1, lon=11.6, h=520.0), None, "zero", "dup"),
]


def main():
    from oracle import pyoracle
    pyoracle.build_ref()
    lib = T.load_lazy(os.path.join(HERE, "_ref", "libref_pvt.so"))   # get_dwt_value() stays unbound: it is never called
    lib.pntpos.argtypes = [C.POINTER(T.Obsd), C.c_int, C.POINTER(T.Nav), C.POINTER(T.Sol)]
    lib.pntpos.restype = C.c_int
    lib.ecef2pos.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    azel = (C.c_double * 8).in_dll(lib, "azel")
    out = {"names": np.array([c[0] for c in CASES])}
    for name, kw, ion, start, mut in CASES:
        week = 2200
        rows, tow, prs = T.make_scenario(**kw)
        sats = [r["sat"] for r in rows]
        if mut == "svh":
            rows[2]["svh"] = 1
        if mut == "dup":
            sats[2] = sats[1]
        ephs, nav, obs = T.build_inputs(rows, week, tow, prs, ion, sats)
        sol = T.Sol()
        truth = T.geodetic_to_ecef(kw["lat"], kw["lon"], kw["h"])
        if start == "near":
            for i in range(3):
                sol.rr[i] = truth[i] + (37.0, -52.0, 18.0)[i]
        rr0 = np.array(sol.rr[:3])
        rc = lib.pntpos(obs, 4, C.byref(nav), C.byref(sol))
        geo = (C.c_double * 3)()
        lib.ecef2pos(sol.rr, geo)
        p = name + "/"
        out[p + "eph"] = np.array([[float(r[k]) for k in T.EPH_FIELDS] for r in rows])
        out[p + "sats"] = np.array(sats, np.int32)
        out[p + "week_tow"] = np.array([week, tow])
        out[p + "pr"] = np.array(prs)
        out[p + "ion"] = np.array(ion if ion is not None else [0.0] * 8)
        out[p + "rr0"] = rr0
        out[p + "truth"] = truth
        out[p + "rc"] = np.array([rc, sol.stat, sol.ns], np.int32)
        out[p + "rr"] = np.array(sol.rr[:])
        out[p + "dtr0"] = np.array([sol.dtr[0]])
        out[p + "qr"] = np.array(sol.qr[:], np.float32)
        out[p + "time"] = np.array([float(sol.time.time), sol.time.sec])
        out[p + "azel_deg"] = np.array(azel[:])
        out[p + "geo"] = np.array(geo[:])
        err = np.linalg.norm(np.array(sol.rr[:3]) - truth)
        print(f"{name:26s} rc={rc} stat={sol.stat} ns={sol.ns} |rr - truth| = {err:10.3f} m  dtr = {sol.dtr[0]:.3e}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "f9_pvt.npz"), **out)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main()
