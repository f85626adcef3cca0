"""ctypes binding of the C ABI in include/gpsx.h / include/gpsx_compat.h (libgpsx.so).

Used by tests/ and bench.py.  It adds nothing to the computation: arrays in, arrays out, every call forwarded to the
shared library, which runs on the GPU or fails.  If the library has not been built, or there is no gfx950 device,
construction raises -- there is no Python/CPU fallback.

Import torch BEFORE this module in processes that use both (one HIP runtime per process: torch's bundled
libamdhip64 and /opt/rocm's share a SONAME, whichever loads first serves both).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libgpsx.so")
LAB_LIB_PATH = os.path.join(PKG, "lib", "libgpsx_lab.so")

BYTES_PER_MS = 2046
BYTES_PER_MS_2BIT = 4092
IF_1BIT, IF_2BIT_SM = 0, 1
PHASES_BYTE = 2046
PHASES_FINE = 16368
IF_HZ = 4092000
SCHED_EVERY_MS, SCHED_MUX17 = 0, 1
WORDSYNC_DEVICE, WORDSYNC_HOST = 0, 1
DRAWS_XORSHIFT, DRAWS_LIBC = 0, 1
ACQ_PATH_MATRIX, ACQ_PATH_VECTOR = 0, 1

LOOP_DTYPE = np.dtype([("prn", "<i4"), ("code_phase_fine", "<f4"), ("if_freq_offset_hz", "<f4"), ("if_freq_accum", "<u4"),
                       ("dll_code_err", "<f4"), ("pll_code_err", "<f4"), ("fll_err", "<f4"), ("fll_old_i", "<i2"),
                       ("fll_old_q", "<i2"), ("pll_check_buf", "<i2", 4), ("pll_bad_state_master_cnt", "<u2"),
                       ("pll_bad_state_cnt", "u1"), ("period_sync_ok_flag", "u1"), ("found_freq_offset_hz", "<i2"),
                       ("reseed_count", "<u2"), ("rng", "<u4"), ("i_part_summ", "<u4"), ("q_part_summ", "<u4"),
                       ("snr_value", "<f4"), ("snr_summ_cnt", "<u2"), ("code_filt_cnt", "<u2"), ("code_phase_fine_filt", "<f4"),
                       ("old_swap_time", "<u4"), ("slot_start_ticks", "<u4"), ("slot_ip", "<i2", 4), ("slot_bits", "u1"),
                       ("right_period_cnt", "u1"), ("old_reminder", "u1"), ("accurate_swap_time", "u1"),
                       ("accurate_swap_ok", "u1"), ("last_bit_pos_cnt", "u1"), ("last_bit_neg_cnt", "u1"),
                       ("inv_polarity_flag", "u1"), ("prev_track_timestamp", "<u4"), ("snr_i_latch", "<u4"),
                       ("snr_q_latch", "<u4"), ("word_buf", "<u4"), ("word_detection_timestamp", "<u4"), ("word_cnt", "u1"),
                       ("word_bit_cnt", "u1"), ("inv_preabmle_cnt", "u1"), ("word_flags", "u1")])   # gpsx_loop_state_t, 120 bytes
LOOP_TRACE_DTYPE = np.dtype([("iq", "<i2", 6), ("code_phase_fine", "<f4"), ("if_freq_offset_hz", "<f4"), ("if_freq_accum", "<u4")])
assert LOOP_DTYPE.itemsize == 120 and LOOP_TRACE_DTYPE.itemsize == 24
PEAK_DTYPE = np.dtype([("max_val", "<u4"), ("phase", "<u4"), ("sum", "<u4"), ("avr", "<u4")])
TRACK_CHUNK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)   # gpsx_track_chunk_fn
TRK_DTYPE = np.dtype([("prn", "<i4"), ("code_phase_fine", "<f4"), ("if_freq_offset_hz", "<f4"),
                      ("if_freq_accum", "<u4")])
JOB_DTYPE = np.dtype([("block", "<i4"), ("n_ms", "<i4"), ("prn", "<i4"), ("freq_hz", "<f4"), ("offset_bits", "<i4"),
                      ("win_start", "<i4"), ("win_stop", "<i4")])


class AcqGrid(C.Structure):
    _fields_ = [("n_search", C.c_int32), ("n_ms", C.c_int32), ("search_stride_blocks", C.c_int32),
                ("n_prn", C.c_int32), ("prns", C.c_void_p), ("dopp_min_hz", C.c_int32), ("dopp_step_hz", C.c_int32),
                ("n_dopp", C.c_int32), ("phase_mode", C.c_int32), ("win_start", C.c_int32), ("win_stop", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32)]


class Config(C.Structure):   # gpsx_config_t
    _fields_ = [("sample_rate_hz", C.c_uint32), ("if_hz", C.c_int32)]


class AcqWeightedT(C.Structure):          # gpsx_acq_weighted_t
    _fields_ = [("n_search", C.c_int32), ("search_stride_blocks", C.c_int32), ("n_prn", C.c_int32), ("prns", C.POINTER(C.c_uint8)),
                ("dopp_min_hz", C.c_int32), ("dopp_step_hz", C.c_int32), ("n_dopp", C.c_int32), ("weights", C.c_int32)]


class GpsxError(RuntimeError):
    pass


CAPTURE_BLOCK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long)   # gpsx_capture_block_fn


def load_library(lab: bool | None = None) -> C.CDLL:
    """lib/libgpsx.so -- or, lab=True (or $GPSX_USE_LAB_LIBRARY=1 when lab is None): lib/libgpsx_lab.so, the same sources
    built with -DGPSX_LAB, the only build that reads the $GPSX_ACQ_* / $GPSX_TRACK_WAVE_FROM knobs forcing a kernel form."""
    if lab is None:
        lab = os.environ.get("GPSX_USE_LAB_LIBRARY") == "1"
    path = LAB_LIB_PATH if lab else LIB_PATH
    if os.environ.get("GPSX_LIB_PATH") and not lab:
        path = os.environ["GPSX_LIB_PATH"]      # another build of the product sources, e.g. lib/libgpsx_asan.so (tools/run_sanitizers.sh)
    if not os.path.exists(path):
        raise GpsxError(f"{path} is missing: run `python -m stm32f4_sdr_gps_amd.build` (needs hipcc)")
    lib = C.CDLL(path)
    assert lib.gpsx_is_lab_build() == (1 if lab else 0)
    lib.gpsx_last_error.restype = C.c_char_p
    lib.gpsx_strerror.restype = C.c_char_p
    lib.gpsx_last_kernel.restype = C.c_char_p
    lib.gpsx_last_kernel.argtypes = [C.c_void_p]
    lib.gpsx_acq_peaks_count.restype = C.c_size_t
    lib.gpsx_acq_keys_count.restype = C.c_size_t
    lib.gpsx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.gpsx_destroy.argtypes = [C.c_void_p]
    for name in ("gpsx_synchronize",):
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.gpsx_last_error.argtypes = [C.c_void_p]
    lib.gpsx_malloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]
    lib.gpsx_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.gpsx_host_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]
    lib.gpsx_bind_thread_to_device.argtypes = [C.c_void_p]
    lib.gpsx_host_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.gpsx_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.gpsx_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.gpsx_event_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.gpsx_event_record.argtypes = [C.c_void_p, C.c_void_p]
    lib.gpsx_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    lib.gpsx_event_destroy.argtypes = [C.c_void_p, C.c_void_p]
    lib.gpsx_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.gpsx_ca_codes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.gpsx_acq_grid_dev.argtypes = [C.c_void_p, C.POINTER(AcqGrid), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gpsx_acq_grid.argtypes = [C.c_void_p, C.POINTER(AcqGrid), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.gpsx_acq_grid_async.argtypes = lib.gpsx_acq_grid.argtypes
    lib.gpsx_acq_jobs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.gpsx_track_epl_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.gpsx_track_epl_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.gpsx_track_epl_batch_chunked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                 TRACK_CHUNK_FN, C.c_void_p]
    lib.gpsx_rewind.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.gpsx_track_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.gpsx_track_loop_dev.argtypes = lib.gpsx_track_loop.argtypes
    lib.gpsx_loop_set_polarity.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.gpsx_loop_reset_code_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.gpsx_loop_set_schedule.argtypes = [C.c_void_p, C.c_int]
    lib.gpsx_loop_set_word_sync.argtypes = [C.c_void_p, C.c_int]
    lib.gpsx_loop_set_draws.argtypes = [C.c_void_p, C.c_int]
    lib.gpsx_set_acq_path.argtypes = [C.c_void_p, C.c_int]
    lib.gpsx_acq_grid_weighted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.gpsx_acq_grid_weighted_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.gps_tracking_words_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_int]
    lib.gpsx_loop_state_from_channel.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.gpsx_loop_state_from_channel.restype = None
    lib.gpsx_loop_state_to_channel.argtypes = [C.c_void_p, C.c_void_p]
    lib.gpsx_loop_state_to_channel.restype = None
    lib.gpsx_wipeoff.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
    lib.gpsx_replica.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
    lib.gpsx_corr_offsets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gpsx_set_if_format.argtypes = [C.c_void_p, C.c_int]
    lib.gpsx_config_default.argtypes = [C.POINTER(Config)]
    lib.gpsx_config_default.restype = None
    lib.gpsx_set_config.argtypes = [C.c_void_p, C.POINTER(Config)]
    lib.gpsx_get_config.argtypes = [C.c_void_p, C.POINTER(Config)]
    lib.gpsx_if_unpack2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.gpsx_mag8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.gpsx_corr_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    # in-process multi-GPU group
    lib.gpsx_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
    lib.gpsx_group_destroy.argtypes = [C.c_void_p]
    lib.gpsx_group_destroy.restype = None
    lib.gpsx_acq_grid_sharded.argtypes = [C.c_void_p, C.POINTER(AcqGrid), C.POINTER(C.c_void_p), C.c_int,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    # capture ring (IF ingest)
    lib.gpsx_capture_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.gpsx_capture_destroy.argtypes = [C.c_void_p]
    lib.gpsx_capture_destroy.restype = None
    lib.gpsx_capture_write_slot.argtypes = [C.c_void_p]
    lib.gpsx_capture_write_slot.restype = C.c_void_p
    lib.gpsx_capture_commit.argtypes = [C.c_void_p]
    lib.gpsx_capture_push.argtypes = [C.c_void_p, C.c_void_p]
    lib.gpsx_capture_ready_buf.argtypes = [C.c_void_p]
    lib.gpsx_capture_ready_buf.restype = C.c_void_p
    lib.gpsx_capture_window_dev.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.gpsx_capture_packet_cnt.argtypes = [C.c_void_p]
    lib.gpsx_capture_packet_cnt.restype = C.c_uint32
    lib.gpsx_capture_block_bytes.argtypes = [C.c_void_p]
    lib.gpsx_capture_block_bytes.restype = C.c_size_t
    lib.gpsx_capture_replay_file.argtypes = [C.c_void_p, C.c_char_p, C.c_long, C.c_long, CAPTURE_BLOCK_FN, C.c_void_p]
    lib.gpsx_capture_replay_file.restype = C.c_long
    for name in ("signal_capture_get_ready_buf", "signal_capture_get_copy_buf"):
        getattr(lib, name).restype = C.c_void_p
    for name in ("signal_capture_have_irq", "signal_capture_check_copied"):
        getattr(lib, name).restype = C.c_uint8
    lib.signal_capture_get_packet_cnt.restype = C.c_uint32
    lib.gpsx_compat_capture_push.argtypes = [C.c_void_p]
    lib.gpsx_compat_capture_push.restype = None
    # compat (reference names)
    lib.gps_generate_prn.argtypes = [C.c_void_p, C.c_int]
    lib.gps_channell_prepare.argtypes = [C.c_void_p]
    lib.gps_correlation8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint16]
    lib.gps_correlation8.restype = C.c_int16
    lib.gps_correlation_iq.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint16, C.POINTER(C.c_int16),
                                       C.POINTER(C.c_int16)]
    lib.correlation_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint16, C.c_uint16,
                                       C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
    lib.correlation_search.restype = C.c_uint16
    lib.gps_shift_to_zero_freq.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    lib.gps_shift_to_zero_freq_track.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gps_generate_prn_data2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint16]
    lib.gps_rewind_if_phase.argtypes = [C.c_void_p, C.c_uint8]
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data


class Engine:
    """One gpsx context.  `stream` may be a raw hipStream_t handle (int), e.g. torch.cuda.current_stream().cuda_stream."""

    def __init__(self, device: int = 0, stream: int | None = None, lab: bool | None = None):
        self.lib = load_library(lab)
        h = C.c_void_p()
        rc = self.lib.gpsx_create(C.byref(h), device, C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise GpsxError(f"gpsx_create(device={device}) -> {rc}: {self.lib.gpsx_strerror(rc).decode()} "
                            "(the correlator engine needs an MI355X; it has no CPU path)")
        self.h = h
        self.block_bytes = BYTES_PER_MS

    def close(self):
        if getattr(self, "h", None):
            for p in getattr(self, "_pinned", []):
                self.lib.gpsx_host_free(self.h, C.c_void_p(p))
            self._pinned = []
            self.lib.gpsx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise GpsxError(f"{what} -> {rc}: {self.lib.gpsx_strerror(rc).decode()}: "
                            f"{self.lib.gpsx_last_error(self.h).decode()}")

    # -- plumbing ----------------------------------------------------------------------------------------------
    def device_info(self):
        name = C.create_string_buffer(128)
        cus, khz = C.c_int(), C.c_int()
        self.lib.gpsx_device_info(self.h, name, 128, C.byref(cus), C.byref(khz))
        return name.value.decode(), cus.value, khz.value

    def synchronize(self):
        self._chk(self.lib.gpsx_synchronize(self.h), "gpsx_synchronize")

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self.lib.gpsx_malloc(self.h, C.byref(p), nbytes), "gpsx_malloc")
        return p.value

    def bind_thread_to_device(self) -> bool:
        """Pin the calling thread to the CPUs local to this context's GPU; False when the topology is not exposed."""
        return self.lib.gpsx_bind_thread_to_device(self.h) == 0

    def host_array(self, shape, dtype) -> np.ndarray:
        """A numpy array in page-locked host memory (gpsx_host_alloc); lives until the engine is closed."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._chk(self.lib.gpsx_host_alloc(self.h, C.byref(p), max(n, 1)), "gpsx_host_alloc")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)
        buf = (C.c_ubyte * max(n, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self, dptr: int):
        self._chk(self.lib.gpsx_free(self.h, dptr), "gpsx_free")

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._chk(self.lib.gpsx_memcpy_h2d(self.h, dptr, arr.ctypes.data, arr.nbytes), "gpsx_memcpy_h2d")

    def d2h(self, arr: np.ndarray, dptr: int):
        assert arr.flags.c_contiguous
        self._chk(self.lib.gpsx_memcpy_d2h(self.h, arr.ctypes.data, dptr, arr.nbytes), "gpsx_memcpy_d2h")

    def event(self) -> int:
        e = C.c_void_p()
        self._chk(self.lib.gpsx_event_create(self.h, C.byref(e)), "gpsx_event_create")
        return e.value

    def record(self, ev: int):
        self._chk(self.lib.gpsx_event_record(self.h, ev), "gpsx_event_record")

    def elapsed_ms(self, ev0: int, ev1: int) -> float:
        ms = C.c_float()
        self._chk(self.lib.gpsx_event_elapsed_ms(self.h, ev0, ev1, C.byref(ms)), "gpsx_event_elapsed_ms")
        return ms.value

    # -- receiver constants (gpsx_config_t) ---------------------------------------------------------------------
    def get_config(self) -> Config:
        cfg = Config()
        self._chk(self.lib.gpsx_get_config(self.h, C.byref(cfg)), "gpsx_get_config")
        return cfg

    def set_config(self, if_hz: int = IF_HZ, sample_rate_hz: int = 16368000):
        cfg = Config(sample_rate_hz, if_hz)
        self._chk(self.lib.gpsx_set_config(self.h, C.byref(cfg)), "gpsx_set_config")

    # -- IF sample format (N3 ingest) ---------------------------------------------------------------------------
    def set_if_format(self, fmt: int):
        self._chk(self.lib.gpsx_set_if_format(self.h, fmt), "gpsx_set_if_format")
        self.block_bytes = BYTES_PER_MS_2BIT if fmt == IF_2BIT_SM else BYTES_PER_MS

    def if_unpack2(self, if_2bit: np.ndarray):
        blocks = np.ascontiguousarray(if_2bit, np.uint8).reshape(-1, BYTES_PER_MS_2BIT)
        sign = np.zeros((len(blocks), BYTES_PER_MS), np.uint8)
        mag = np.zeros((len(blocks), BYTES_PER_MS), np.uint8)
        self._chk(self.lib.gpsx_if_unpack2(self.h, blocks.ctypes.data, len(blocks), sign.ctypes.data, mag.ctypes.data),
                  "gpsx_if_unpack2")
        return sign, mag

    # -- K1 ---------------------------------------------------------------------------------------------------
    def ca_codes(self, prns) -> np.ndarray:
        prns = np.ascontiguousarray(prns, np.uint8)
        out = np.zeros((len(prns), 1023), np.uint8)
        self._chk(self.lib.gpsx_ca_codes(self.h, prns.ctypes.data, len(prns), out.ctypes.data), "gpsx_ca_codes")
        return out

    # -- acquisition ------------------------------------------------------------------------------------------
    @staticmethod
    def grid_desc(prns: np.ndarray, n_search=1, n_ms=1, search_stride_blocks=None, dopp_min_hz=-5000,
                  dopp_step_hz=500, n_dopp=21, phase_mode=PHASES_FINE, win=(0, 2046), shard=(0, 1)) -> AcqGrid:
        g = AcqGrid()
        g.n_search, g.n_ms = n_search, n_ms
        g.search_stride_blocks = n_ms if search_stride_blocks is None else search_stride_blocks
        g.n_prn, g.prns = len(prns), prns.ctypes.data
        g.dopp_min_hz, g.dopp_step_hz, g.n_dopp = dopp_min_hz, dopp_step_hz, n_dopp
        g.phase_mode = phase_mode
        g.win_start, g.win_stop = win
        g.shard_index, g.shard_count = shard
        return g

    def acq_grid(self, if_blocks: np.ndarray, prns, want_keys=True, **kw):
        """Host-buffer grid search.  Returns (peaks[n_search, n_prn, n_dopp, n_bits], keys[n_search, n_prn, n_dopp])."""
        prns = np.ascontiguousarray(prns, np.uint8)
        blocks = np.ascontiguousarray(if_blocks, np.uint8).reshape(-1, self.block_bytes)
        g = self.grid_desc(prns, **kw)
        n_bits = 8 if g.phase_mode == PHASES_FINE else 1
        peaks = np.zeros((g.n_search, g.n_prn, g.n_dopp, n_bits), PEAK_DTYPE)
        keys = np.zeros((g.n_search, g.n_prn, g.n_dopp), np.int64) if want_keys else None
        self._chk(self.lib.gpsx_acq_grid(self.h, C.byref(g), blocks.ctypes.data, len(blocks), peaks.ctypes.data,
                                         _ptr(keys)), "gpsx_acq_grid")
        return peaks, keys

    def acq_grid_debug(self, if_blocks: np.ndarray, prns, want_per_ms=False, want_energy=False, want_cnt=False, **kw):
        """Device-pointer path with the optional inspection outputs, staged through gpsx_malloc'ed buffers."""
        prns = np.ascontiguousarray(prns, np.uint8)
        blocks = np.ascontiguousarray(if_blocks, np.uint8).reshape(-1, BYTES_PER_MS)
        g = self.grid_desc(prns, **kw)
        n_bits = 8 if g.phase_mode == PHASES_FINE else 1
        shape = (g.n_search, g.n_prn, g.n_dopp, n_bits)
        peaks = np.zeros(shape, PEAK_DTYPE)
        keys = np.zeros(shape[:3], np.int64)
        per_ms = np.zeros(shape + (g.n_ms,), PEAK_DTYPE) if want_per_ms else None
        energy = np.zeros(shape + (2046,), np.uint32) if want_energy else None
        cnt = np.zeros(shape + (2046, 2), np.uint16) if want_cnt else None
        bufs = []

        def dev(arr):
            if arr is None:
                return None
            p = self.malloc(arr.nbytes)
            self.h2d(p, arr)
            bufs.append(p)
            return p

        try:
            d_if = dev(np.concatenate([blocks.reshape(-1), np.zeros(2, np.uint8)]))
            d_peaks, d_keys, d_per, d_en, d_cnt = dev(peaks), dev(keys), dev(per_ms), dev(energy), dev(cnt)
            self._chk(self.lib.gpsx_acq_grid_dev(self.h, C.byref(g), d_if, len(blocks), d_peaks, d_keys, d_per, d_en,
                                                 d_cnt), "gpsx_acq_grid_dev")
            self.synchronize()
            for arr, p in ((peaks, d_peaks), (keys, d_keys), (per_ms, d_per), (energy, d_en), (cnt, d_cnt)):
                if arr is not None:
                    self.d2h(arr, p)
        finally:
            for p in bufs:
                self.free(p)
        return dict(peaks=peaks, keys=keys, per_ms=per_ms, energy=energy, cnt=cnt)

    def acq_jobs(self, if_blocks: np.ndarray, jobs: np.ndarray, want_energy=False):
        blocks = np.ascontiguousarray(if_blocks, np.uint8).reshape(-1, self.block_bytes)
        jobs = np.ascontiguousarray(jobs, JOB_DTYPE)
        peaks = np.zeros(len(jobs), PEAK_DTYPE)
        energy = np.zeros((len(jobs), 2046), np.uint32) if want_energy else None
        self._chk(self.lib.gpsx_acq_jobs(self.h, jobs.ctypes.data, len(jobs), blocks.ctypes.data, len(blocks),
                                         peaks.ctypes.data, _ptr(energy)), "gpsx_acq_jobs")
        return peaks, energy

    # -- tracking ---------------------------------------------------------------------------------------------
    def track_epl(self, if_block: np.ndarray, states: np.ndarray, iq_out: np.ndarray | None = None) -> np.ndarray:
        """states: TRK_DTYPE array, updated in place (if_freq_accum).  Returns int16 [n_ch, 6] = IE,QE,IP,QP,IL,QL
        (written into iq_out when given, e.g. a page-locked array from host_array())."""
        assert states.dtype == TRK_DTYPE and states.flags.c_contiguous
        blk = np.ascontiguousarray(if_block, np.uint8)
        iq = np.zeros((len(states), 6), np.int16) if iq_out is None else iq_out
        self._chk(self.lib.gpsx_track_epl_batch(self.h, blk.ctypes.data, states.ctypes.data, len(states),
                                                iq.ctypes.data), "gpsx_track_epl_batch")
        return iq

    def track_epl_chunked(self, if_block: np.ndarray, states: np.ndarray, n_chunks: int, on_chunk,
                          iq_out: np.ndarray | None = None) -> np.ndarray:
        """track_epl in n_chunks pieces; on_chunk(first, n) is called on this thread as each piece's results land in
        `states` / the returned array while the GPU works on the next pieces (gpsx_track_epl_batch_chunked)."""
        assert states.dtype == TRK_DTYPE and states.flags.c_contiguous
        blk = np.ascontiguousarray(if_block, np.uint8)
        iq = np.zeros((len(states), 6), np.int16) if iq_out is None else iq_out
        assert iq.dtype == np.int16 and iq.flags.c_contiguous and iq.shape == (len(states), 6)
        raised = []

        def trampoline(user, first, n):   # (ctypes would print and swallow an exception raised inside the callback)
            if not raised:
                try:
                    on_chunk(first, n)
                except BaseException as exc:   # noqa: BLE001 -- re-raised below, on the caller's stack
                    raised.append(exc)

        cb = TRACK_CHUNK_FN(trampoline)
        rc = self.lib.gpsx_track_epl_batch_chunked(self.h, blk.ctypes.data, states.ctypes.data, len(states),
                                                   iq.ctypes.data, n_chunks, cb, None)
        if raised:
            raise raised[0]
        self._chk(rc, "gpsx_track_epl_batch_chunked")
        return iq

    def acq_grid_weighted(self, blocks_2bit: np.ndarray, prns, n_search: int, dopp_min_hz: int, dopp_step_hz: int, n_dopp: int,
                          use_magnitude: bool = True, stride_blocks: int = 1) -> np.ndarray:
        """EXTENSION (not in the reference): the acquisition grid on weighted two-bit samples -> PEAK_DTYPE [n_search, n_prn, n_dopp]"""
        blocks = np.ascontiguousarray(blocks_2bit, np.uint8).reshape(-1, BYTES_PER_MS_2BIT)
        prns = np.ascontiguousarray(prns, np.uint8)
        g = AcqWeightedT(n_search, stride_blocks, len(prns), prns.ctypes.data_as(C.POINTER(C.c_uint8)), dopp_min_hz, dopp_step_hz, n_dopp,
                         1 if use_magnitude else 0)
        peaks = np.zeros((n_search, len(prns), n_dopp), PEAK_DTYPE)
        self._chk(self.lib.gpsx_acq_grid_weighted(self.h, C.byref(g), blocks.ctypes.data, len(blocks), peaks.ctypes.data),
                  "gpsx_acq_grid_weighted")
        return peaks

    def set_loop_schedule(self, schedule: int) -> None:
        """SCHED_EVERY_MS or SCHED_MUX17 (the reference's four-channel 17 ms multiplex) for this context's track_loop launches"""
        self._chk(self.lib.gpsx_loop_set_schedule(self.h, schedule), "gpsx_loop_set_schedule")

    def set_acq_path(self, path: int) -> None:
        """ACQ_PATH_MATRIX (default) or ACQ_PATH_VECTOR (no MFMA: the polyphase popcount kernel)"""
        self._chk(self.lib.gpsx_set_acq_path(self.h, path), "gpsx_set_acq_path")

    def set_loop_draws(self, draws: int) -> None:
        """DRAWS_XORSHIFT (default) or DRAWS_LIBC (the reference's rand(), drawn on the host in the reference's order)"""
        self._chk(self.lib.gpsx_loop_set_draws(self.h, draws), "gpsx_loop_set_draws")

    def set_loop_word_sync(self, owner: int) -> None:
        """WORDSYNC_DEVICE (default: the kernel decides the data polarity itself) or WORDSYNC_HOST (gpsx_loop_set_polarity only)"""
        self._chk(self.lib.gpsx_loop_set_word_sync(self.h, owner), "gpsx_loop_set_word_sync")

    def track_loop(self, if_blocks: np.ndarray, d_state: int, n_ch: int, first_tick: int, want_trace=False):
        """K = len(if_blocks) milliseconds of the device tracking loops on the n_ch states at device address d_state.
        Returns (flags uint8 [K, n_ch], trace LOOP_TRACE_DTYPE [K, n_ch] or None)."""
        blocks = np.ascontiguousarray(if_blocks, np.uint8).reshape(-1, self.block_bytes)
        flags = np.zeros((len(blocks), n_ch), np.uint8)
        trace = np.zeros((len(blocks), n_ch), LOOP_TRACE_DTYPE) if want_trace else None
        self._chk(self.lib.gpsx_track_loop(self.h, blocks.ctypes.data, len(blocks), d_state, n_ch, first_tick,
                                           flags.ctypes.data, _ptr(trace)), "gpsx_track_loop")
        return flags, trace

    def rewind(self, states: np.ndarray, steps) -> None:
        steps = np.ascontiguousarray(steps, np.uint8)
        self._chk(self.lib.gpsx_rewind(self.h, states.ctypes.data, len(states), steps.ctypes.data), "gpsx_rewind")

    # -- per-call primitives ------------------------------------------------------------------------------------
    def wipeoff(self, signal, freq_hz, accum=0, prefill=None):
        di = np.zeros(1024, np.uint16) if prefill is None else prefill[0].copy()
        dq = np.zeros(1024, np.uint16) if prefill is None else prefill[1].copy()
        acc = C.c_uint32(accum)
        sig = np.ascontiguousarray(signal, np.uint8)
        self._chk(self.lib.gpsx_wipeoff(self.h, sig.ctypes.data, np.float32(freq_hz), C.byref(acc), di.ctypes.data,
                                        dq.ctypes.data), "gpsx_wipeoff")
        return di, dq, acc.value

    def replica(self, chips, offset_bits, pad_in=0):
        out = np.zeros(1024, np.uint16)
        out[1023] = pad_in
        chips = np.ascontiguousarray(chips, np.uint8)
        self._chk(self.lib.gpsx_replica(self.h, chips.ctypes.data, offset_bits, out.ctypes.data), "gpsx_replica")
        return out

    def corr_offsets(self, rep, di, dq, offsets):
        offsets = np.ascontiguousarray(offsets, np.uint16)
        n = len(offsets)
        ci, cq, c8 = np.zeros(n, np.uint16), np.zeros(n, np.uint16), np.zeros(n, np.int16)
        self._chk(self.lib.gpsx_corr_offsets(self.h, rep.ctypes.data, di.ctypes.data, dq.ctypes.data,
                                             offsets.ctypes.data, n, ci.ctypes.data, cq.ctypes.data, c8.ctypes.data),
                  "gpsx_corr_offsets")
        return ci, cq, c8

    def mag8(self, cnt_i, cnt_q) -> np.ndarray:
        ci = np.ascontiguousarray(cnt_i, np.uint16)
        cq = np.ascontiguousarray(cnt_q, np.uint16)
        out = np.zeros(len(ci), np.int16)
        self._chk(self.lib.gpsx_mag8(self.h, ci.ctypes.data, cq.ctypes.data, len(ci), out.ctypes.data), "gpsx_mag8")
        return out

    def corr_search(self, rep, di, dq, start, stop):
        pk = np.zeros(1, PEAK_DTYPE)
        self._chk(self.lib.gpsx_corr_search(self.h, rep.ctypes.data, di.ctypes.data, dq.ctypes.data, start, stop,
                                            pk.ctypes.data), "gpsx_corr_search")
        return int(pk["max_val"][0]), int(pk["avr"][0]), int(pk["phase"][0])


class Capture:
    """A capture ring of `eng` (include/gpsx.h "IF ingest"): pinned host slots mirrored in HBM by asynchronous copies."""

    def __init__(self, eng: Engine, n_slots: int):
        self.eng, self.lib = eng, eng.lib
        h = C.c_void_p()
        eng._chk(self.lib.gpsx_capture_create(eng.h, n_slots, C.byref(h)), "gpsx_capture_create")
        self.h = h
        self.block_bytes = int(self.lib.gpsx_capture_block_bytes(h))

    def close(self):
        if getattr(self, "h", None) and getattr(self.eng, "h", None):
            self.lib.gpsx_capture_destroy(self.h)
        self.h = None

    def push(self, block: np.ndarray):
        block = np.ascontiguousarray(block, np.uint8).reshape(-1)
        assert block.size == self.block_bytes
        self.eng._chk(self.lib.gpsx_capture_push(self.h, _ptr(block)), "gpsx_capture_push")

    def ready_ptr(self) -> int:
        """Host address of the newest committed block (what signal_capture_get_ready_buf returns), 0 before the first."""
        return int(self.lib.gpsx_capture_ready_buf(self.h) or 0)

    def ready_view(self, n_blocks: int = 1) -> np.ndarray:
        """numpy view (no copy) of the pinned slot(s) ending at the newest block; only windows that do not wrap."""
        p = self.ready_ptr() - (n_blocks - 1) * self.block_bytes
        buf = (C.c_uint8 * (n_blocks * self.block_bytes)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(n_blocks, self.block_bytes)

    def window_dev(self, n_blocks: int) -> int:
        d = C.c_void_p()
        self.eng._chk(self.lib.gpsx_capture_window_dev(self.h, n_blocks, C.byref(d)), "gpsx_capture_window_dev")
        return int(d.value)

    def packet_cnt(self) -> int:
        return int(self.lib.gpsx_capture_packet_cnt(self.h))

    def replay_file(self, path: str, first_block=0, max_blocks=-1, on_block=None) -> int:
        cb = CAPTURE_BLOCK_FN((lambda user, cap, idx: int(on_block(idx) or 0)) if on_block else 0)
        n = int(self.lib.gpsx_capture_replay_file(self.h, path.encode(), first_block, max_blocks, cb, None))
        if n < 0:
            self.eng._chk(n, "gpsx_capture_replay_file")
        return n
