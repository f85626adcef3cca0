"""stm32f4_sdr_gps_amd -- MI355X-native GPS L1 C/A correlator engine (drop-in for the correlation hot path of
iliasam/STM32F4_SDR_GPS).  The product is the C-ABI shared library built from csrc/ (see include/gpsx.h);
this package holds its build recipe, a ctypes binding used by tests and bench.py, and host-side tooling."""

__all__ = ["capi", "synth", "build"]
