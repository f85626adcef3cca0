"""oracle/pyoracle.py -- TEST INFRASTRUCTURE: ctypes/numpy bindings of oracle/liboracle.so (the CPU restatement)
and, where it has been built, of oracle/_ref/*.so (the reference's own C compiled in place by oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BYTES = 2046
WORDS16 = 1023
CHIPS = 1023
IF_HZ = 4092000

_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile liboracle.so (gcc, seconds).  Building the checker is not using it."""
    so = os.path.join(HERE, "liboracle.so")
    src = [os.path.join(HERE, f) for f in ("gpsx_oracle.c", "gpsx_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_ref() -> bool:
    """Compile oracle/_ref/*.so from /root/reference when that tree is present (this container only)."""
    if not os.path.isdir("/root/reference/Firmware/project_main"):
        return False
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    return True


class PeakT(C.Structure):
    _fields_ = [("max_val", C.c_uint32), ("phase", C.c_uint32), ("sum", C.c_uint32), ("avr", C.c_uint32)]


PEAK_DTYPE = np.dtype([("max_val", "<u4"), ("phase", "<u4"), ("sum", "<u4"), ("avr", "<u4")])


# ---- committed fixtures of the oracle's large grid sweeps (tests/golden/f11_grids/) ---------------------------------------
FIXTURE_MIN_UNITS = 5000      # (block, PRN, Doppler, bit shift) units: from one full 32 x 21 x 8 one-block grid (5376 units, ~1.5 s) up


def grid_key(blocks, prns, args) -> str:
    """sha256 over the input bytes of an acq_grid call: the blocks, the PRN list, (n_ms, dopp_min, dopp_step, n_dopp, n_bits)"""
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(blocks, np.uint8).tobytes())
    h.update(b"|" + np.ascontiguousarray(prns, np.uint8).tobytes() + b"|")
    h.update(np.asarray(args, "<i8").tobytes())
    return h.hexdigest()[:32]


def save_grid_fixture(directory, key, blocks, prns, args, peaks, **meta):
    os.makedirs(directory, exist_ok=True)
    np.savez_compressed(os.path.join(directory, key + ".npz"), blocks=np.ascontiguousarray(blocks, np.uint8),
                        prns=np.ascontiguousarray(prns, np.uint8), args=np.asarray(args, np.int64), peaks=peaks,
                        meta=np.array(json.dumps(meta)))


def load_grid_fixture(directory, key, blocks, prns, args):
    """The stored peaks when <directory>/<key>.npz exists AND holds exactly these inputs (the key is a hash: the bytes are
    compared as well); else None."""
    path = os.path.join(directory, key + ".npz")
    if not os.path.exists(path):
        return None
    with np.load(path) as z:
        if (not np.array_equal(z["blocks"], blocks) or not np.array_equal(z["prns"], prns)
                or tuple(int(a) for a in z["args"]) != tuple(args)):
            return None
        return z["peaks"].astype(PEAK_DTYPE, copy=True)


class Oracle:
    """The from-scratch restatement (liboracle.so)."""

    grid_fixtures = None          # a directory of committed grid fixtures (set by tests/conftest.py), or None: always live
    fixture_hits = 0

    def __init__(self):
        self.lib = L = C.CDLL(build())
        L.orc_ca_code.argtypes = [C.c_int, _u8p]
        L.orc_ca_code.restype = C.c_int
        L.orc_replica.argtypes = [_u8p, C.c_uint, _u16p]
        L.orc_nco_step.argtypes = [C.c_float]
        L.orc_nco_step.restype = C.c_uint32
        L.orc_wipeoff.argtypes = [_u8p, C.c_float, C.POINTER(C.c_uint32), _u8p, _u8p]
        L.orc_rewind.argtypes = [C.c_float, C.c_uint32, C.c_uint]
        L.orc_rewind.restype = C.c_uint32
        L.orc_mult_and_summ.argtypes = [_u8p, _u8p, _u8p, C.c_uint, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
        L.orc_correlation8.argtypes = [_u16p, _u16p, _u16p, C.c_uint]
        L.orc_correlation8.restype = C.c_int16
        L.orc_correlation_iq.argtypes = [_u16p, _u16p, _u16p, C.c_uint, C.POINTER(C.c_int16), C.POINTER(C.c_int16)]
        L.orc_correlation_search.argtypes = [_u16p, _u16p, _u16p, C.c_uint, C.c_uint,
                                             C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
        L.orc_correlation_search.restype = C.c_uint16
        L.orc_mag8.argtypes = [C.c_int, C.c_int]
        L.orc_mag8.restype = C.c_int16
        L.orc_search_job.argtypes = [_u8p, C.c_int, _u8p, C.c_float, C.c_uint, C.c_uint, C.c_uint,
                                     C.POINTER(PeakT), C.c_void_p, C.c_void_p]
        L.orc_acq_grid.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int]
        L.orc_track_epl.argtypes = [_u8p, _u8p, C.c_float, C.c_float, C.POINTER(C.c_uint32), _i16p]
        L.orc_weighted_iq.argtypes = [_u8p, C.c_int, C.c_float, C.c_uint, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_acq_grid_weighted.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int]

    def ca_code(self, prn: int) -> np.ndarray:
        chips = np.zeros(CHIPS, np.uint8)
        rc = self.lib.orc_ca_code(prn, chips)
        if rc != 0:
            raise ValueError(f"unsupported prn {prn}")
        return chips

    def replica(self, chips: np.ndarray, offset_bits: int, pad_in: int = 0) -> np.ndarray:
        out = np.zeros(WORDS16 + 1, np.uint16)
        out[WORDS16] = pad_in
        self.lib.orc_replica(np.ascontiguousarray(chips, np.uint8), offset_bits, out)
        return out

    def nco_step(self, freq_hz: float) -> int:
        return int(self.lib.orc_nco_step(np.float32(freq_hz)))

    def wipeoff(self, signal: np.ndarray, freq_hz: float, accum: int = 0, prefill=None):
        """returns (data_i[1024 u16], data_q[1024 u16], accum_out); bytes 2044.. keep `prefill` (default 0)."""
        di = np.zeros(WORDS16 + 1, np.uint16) if prefill is None else prefill[0].copy()
        dq = np.zeros(WORDS16 + 1, np.uint16) if prefill is None else prefill[1].copy()
        acc = C.c_uint32(accum)
        self.lib.orc_wipeoff(np.ascontiguousarray(signal, np.uint8), np.float32(freq_hz), C.byref(acc),
                             di.view(np.uint8), dq.view(np.uint8))
        return di, dq, acc.value

    def rewind(self, if_freq_offset_hz: float, accum: int, steps: int) -> int:
        return int(self.lib.orc_rewind(np.float32(if_freq_offset_hz), accum, steps))

    def mult_and_summ(self, di, dq, rep, offset):
        ci, cq = C.c_uint16(), C.c_uint16()
        self.lib.orc_mult_and_summ(di.view(np.uint8), dq.view(np.uint8), rep.view(np.uint8), offset,
                                   C.byref(ci), C.byref(cq))
        return ci.value, cq.value

    def correlation8(self, rep, di, dq, offset) -> int:
        return int(self.lib.orc_correlation8(rep, di, dq, offset))

    def correlation_iq(self, rep, di, dq, offset):
        ri, rq = C.c_int16(), C.c_int16()
        self.lib.orc_correlation_iq(rep, di, dq, offset, C.byref(ri), C.byref(rq))
        return ri.value, rq.value

    def correlation_search(self, rep, di, dq, start, stop):
        av, ph = C.c_uint16(), C.c_uint16()
        mx = self.lib.orc_correlation_search(rep, di, dq, start, stop, C.byref(av), C.byref(ph))
        return int(mx), av.value, ph.value

    def mag8(self, cnt_i: int, cnt_q: int) -> int:
        return int(self.lib.orc_mag8(cnt_i, cnt_q))

    def search_job(self, if_blocks, n_ms, chips, freq_hz, offset_bits=0, start=0, stop=BYTES,
                   want_energy=False, want_per_ms=False):
        pk = PeakT()
        energy = np.zeros(BYTES, np.uint32) if want_energy else None
        per_ms = np.zeros(n_ms, PEAK_DTYPE) if want_per_ms else None
        self.lib.orc_search_job(np.ascontiguousarray(if_blocks, np.uint8).reshape(-1), n_ms,
                                np.ascontiguousarray(chips, np.uint8), np.float32(freq_hz), offset_bits, start, stop,
                                C.byref(pk), energy.ctypes.data if want_energy else None,
                                per_ms.ctypes.data if want_per_ms else None)
        peak = dict(max_val=pk.max_val, phase=pk.phase, sum=pk.sum, avr=pk.avr)
        return peak, energy, per_ms

    def acq_grid(self, if_blocks, n_ms, prns, dopp_min_hz, dopp_step_hz, n_dopp, n_bits, n_threads=1, live=False):
        """Peak triplets of a whole (PRN, Doppler, bit shift) grid over n_ms blocks.  With `grid_fixtures` set (tests/conftest.py
        does, for the GPU suite) a LARGE sweep -- FIXTURE_MIN_UNITS (block, PRN, Doppler, bit shift) units or more, i.e. several
        seconds of CPU -- is first looked up among the committed fixtures tests/golden/f11_grids/<sha256 of the inputs>.npz: this
        oracle's own outputs for exactly these input bytes, generated in the build container by oracle/gen_golden_grids.py (which
        recomputes every one of them from the inputs stored in the same file).  `live=True` never looks.  A miss computes live --
        and, under $GPSX_GOLDEN_RECORD=<dir>, leaves the case (inputs + outputs) in <dir> for gen_golden_grids.py to pick up."""
        prns = np.ascontiguousarray(prns, np.uint8)
        blocks = np.ascontiguousarray(if_blocks, np.uint8).reshape(-1)
        big = n_ms * len(prns) * n_dopp * n_bits >= FIXTURE_MIN_UNITS
        args = (int(n_ms), int(dopp_min_hz), int(dopp_step_hz), int(n_dopp), int(n_bits))
        key = None
        if big and not live and (self.grid_fixtures or os.environ.get("GPSX_GOLDEN_RECORD")):
            key = grid_key(blocks[:n_ms * BYTES], prns, args)
            hit = load_grid_fixture(self.grid_fixtures, key, blocks[:n_ms * BYTES], prns, args) if self.grid_fixtures else None
            if hit is not None:
                self.fixture_hits += 1
                return hit
        peaks = np.zeros((len(prns), n_dopp, n_bits), PEAK_DTYPE)
        t0 = time.perf_counter()
        self.lib.orc_acq_grid(blocks, n_ms, prns, len(prns), dopp_min_hz, dopp_step_hz, n_dopp, n_bits, peaks.ctypes.data, n_threads)
        if key is not None and os.environ.get("GPSX_GOLDEN_RECORD"):
            save_grid_fixture(os.environ["GPSX_GOLDEN_RECORD"], key, blocks[:n_ms * BYTES], prns, args, peaks,
                              seconds=time.perf_counter() - t0, threads=n_threads)
        return peaks

    # -- extension: weighted two-bit correlation (not in the reference; gpsx_oracle.h) -------------------------------
    def weighted_iq(self, block_2bit, prn, freq_hz, tau, use_magnitude=True):
        i, q = C.c_int32(), C.c_int32()
        self.lib.orc_weighted_iq(np.ascontiguousarray(block_2bit, np.uint8).reshape(-1), prn, np.float32(freq_hz), tau,
                                 1 if use_magnitude else 0, C.byref(i), C.byref(q))
        return i.value, q.value

    def acq_grid_weighted(self, blocks_2bit, n_search, prns, dopp_min_hz, dopp_step_hz, n_dopp, use_magnitude=True, stride_blocks=1,
                          n_threads=1):
        prns = np.ascontiguousarray(prns, np.uint8)
        peaks = np.zeros((n_search, len(prns), n_dopp), PEAK_DTYPE)
        self.lib.orc_acq_grid_weighted(np.ascontiguousarray(blocks_2bit, np.uint8).reshape(-1), n_search, stride_blocks, prns, len(prns),
                                       dopp_min_hz, dopp_step_hz, n_dopp, 1 if use_magnitude else 0, peaks.ctypes.data, n_threads)
        return peaks

    def track_epl(self, signal, chips, code_phase_fine, if_freq_offset_hz, accum):
        acc = C.c_uint32(accum)
        iq = np.zeros(6, np.int16)
        self.lib.orc_track_epl(np.ascontiguousarray(signal, np.uint8), np.ascontiguousarray(chips, np.uint8),
                               np.float32(code_phase_fine), np.float32(if_freq_offset_hz), C.byref(acc), iq)
        return iq, acc.value


class RefPM:
    """The reference's own primitives (oracle/_ref/libref_pm.so = PM/GPS/gps_misc.c + common_ram.c, unmodified)."""

    PATH = os.path.join(HERE, "_ref", "libref_pm.so")

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = L = C.CDLL(self.PATH)
        L.gps_fill_summ_table()
        L.gps_generate_prn.argtypes = [_u8p, C.c_int]
        L.gps_mult_and_summ.argtypes = [_u8p, _u8p, _u8p, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16),
                                        C.c_uint16, C.c_uint16]
        L.gps_correlation8.argtypes = [_u16p, _u16p, _u16p, C.c_uint16]
        L.gps_correlation8.restype = C.c_int16
        L.gps_correlation_iq.argtypes = [_u16p, _u16p, _u16p, C.c_uint16, C.POINTER(C.c_int16), C.POINTER(C.c_int16)]
        L.correlation_search.argtypes = [_u16p, _u16p, _u16p, C.c_uint16, C.c_uint16,
                                         C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
        L.correlation_search.restype = C.c_uint16
        L.gps_shift_to_zero_freq.argtypes = [_u8p, _u8p, _u8p, C.c_float]
        L.gps_shift_to_zero_freq_track.argtypes = [C.c_void_p, _u8p, _u8p, _u8p]
        L.gps_generate_prn_data2.argtypes = [C.c_void_p, _u16p, C.c_uint16]
        L.gps_rewind_if_phase.argtypes = [C.c_void_p, C.c_uint8]
        # layout probes of the reference structs on this ABI (x86-64 gcc): see SURVEY.md 4.2
        self.CH_SIZE, self.CH_TRACKING, self.CH_PRN, self.CH_PRN_CODE = 1688, 60, 664, 665
        self.TRK_FREQ, self.TRK_ACCUM = 4, 8  # offsets inside gps_tracking_t

    def ca_code(self, prn):
        chips = np.zeros(CHIPS, np.uint8)
        self.lib.gps_generate_prn(chips, prn)
        return chips

    def _channel(self, chips, prn=1):
        ch = np.zeros(self.CH_SIZE, np.uint8)
        ch[self.CH_PRN] = prn
        ch[self.CH_PRN_CODE:self.CH_PRN_CODE + CHIPS] = chips
        return ch

    def replica(self, chips, offset_bits, pad_in=0):
        ch = self._channel(chips)
        out = np.zeros(WORDS16 + 1, np.uint16)
        out[WORDS16] = pad_in
        self.lib.gps_generate_prn_data2(ch.ctypes.data, out, offset_bits)
        return out

    def wipeoff(self, signal, freq_hz, prefill=None):
        di = np.zeros(WORDS16 + 1, np.uint16) if prefill is None else prefill[0].copy()
        dq = np.zeros(WORDS16 + 1, np.uint16) if prefill is None else prefill[1].copy()
        self.lib.gps_shift_to_zero_freq(np.ascontiguousarray(signal, np.uint8), di.view(np.uint8), dq.view(np.uint8),
                                        np.float32(freq_hz))
        return di, dq

    def wipeoff_track(self, signal, if_freq_offset_hz, accum):
        trk = np.zeros(152, np.uint8)
        trk[self.TRK_FREQ:self.TRK_FREQ + 4] = np.frombuffer(np.float32(if_freq_offset_hz).tobytes(), np.uint8)
        trk[self.TRK_ACCUM:self.TRK_ACCUM + 4] = np.frombuffer(np.uint32(accum).tobytes(), np.uint8)
        di = np.zeros(WORDS16 + 1, np.uint16)
        dq = np.zeros(WORDS16 + 1, np.uint16)
        self.lib.gps_shift_to_zero_freq_track(trk.ctypes.data, np.ascontiguousarray(signal, np.uint8),
                                              di.view(np.uint8), dq.view(np.uint8))
        return di, dq, int(trk[self.TRK_ACCUM:self.TRK_ACCUM + 4].view(np.uint32)[0])

    def rewind(self, if_freq_offset_hz, accum, steps):
        trk = np.zeros(152, np.uint8)
        trk[self.TRK_FREQ:self.TRK_FREQ + 4] = np.frombuffer(np.float32(if_freq_offset_hz).tobytes(), np.uint8)
        trk[self.TRK_ACCUM:self.TRK_ACCUM + 4] = np.frombuffer(np.uint32(accum).tobytes(), np.uint8)
        self.lib.gps_rewind_if_phase(trk.ctypes.data, steps)
        return int(trk[self.TRK_ACCUM:self.TRK_ACCUM + 4].view(np.uint32)[0])

    def mult_and_summ(self, di, dq, rep, offset):
        ci, cq = C.c_uint16(), C.c_uint16()
        self.lib.gps_mult_and_summ(di.view(np.uint8), dq.view(np.uint8), rep.view(np.uint8),
                                   C.byref(ci), C.byref(cq), BYTES, offset)
        return ci.value, cq.value

    def correlation8(self, rep, di, dq, offset):
        return int(self.lib.gps_correlation8(rep, di, dq, offset))

    def correlation_iq(self, rep, di, dq, offset):
        ri, rq = C.c_int16(), C.c_int16()
        self.lib.gps_correlation_iq(rep, di, dq, offset, C.byref(ri), C.byref(rq))
        return ri.value, rq.value

    def correlation_search(self, rep, di, dq, start, stop):
        av, ph = C.c_uint16(), C.c_uint16()
        mx = self.lib.correlation_search(rep, di, dq, start, stop, C.byref(av), C.byref(ph))
        return int(mx), av.value, ph.value


class RefSSSim:
    """The reference's single-satellite signal simulator (SS/GPS/simulator.c, unmodified)."""

    PATH = os.path.join(HERE, "_ref", "libref_ss_sim.so")

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = C.CDLL(self.PATH)
        self.lib.sim_generate_data.restype = C.POINTER(C.c_uint16)
        self.lib.sim_add_noise.argtypes = [C.POINTER(C.c_uint16), C.c_uint8]
        self.libc = C.CDLL("libc.so.6")

    def block(self, noise_level: int = 0, srand_seed: int = 1) -> np.ndarray:
        p = self.lib.sim_generate_data()
        if noise_level:
            self.libc.srand(srand_seed)
            self.lib.sim_add_noise(p, noise_level)
        return np.ctypeslib.as_array(p, shape=(WORDS16,)).copy().view(np.uint8)
