"""csrc/gpsx_libm.hpp -- the float arctangents the device tracking loops run (restated from glibc's / fdlibm's published
algorithm so that the device produces the bits the reference's x86 build gets from its C library) -- compiled for the HOST and
compared with the C library of this machine, bit for bit: atanf on every 7th float of both signs, atan2f on a grid of the
integer pairs the loops feed it (|I|, |Q| <= 8184), atanf of their quotients.  log10f_near (fdlibm's split around a correctly
rounded logarithm) is allowed its stated 0.3 % of one-ulp differences."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r"""
#include "gpsx_libm.hpp"
#include <math.h>
#include <stdio.h>
#include <string.h>
using namespace gpsx_libm;
static int differ(float a, float b) { return memcmp(&a, &b, 4) != 0; }
int main()
{
  long bad_atan = 0, n_atan = 0, bad_atan2 = 0, n_atan2 = 0, bad_q = 0, bad_log = 0, n_log = 0;
  for (uint32_t u = 0; u < 0x7f800000u; u += 7) {
    const float x = i2f((int32_t)u);
    bad_atan += differ(atanf(x), atanf_fdlibm(x)) + differ(atanf(-x), atanf_fdlibm(-x));
    n_atan += 2;
  }
  for (int q = -8184; q <= 8184; q += 3)
    for (int i = -8184; i <= 8184; i += 5) {
      bad_atan2 += differ(atan2f((float)q, (float)i), atan2f_fdlibm((float)q, (float)i));
      n_atan2++;
      if (i)
        bad_q += differ(atanf((float)q / (float)i), atanf_fdlibm((float)q / (float)i));
    }
  for (int q = 0; q <= 8184; q += 1)      /* the edges: an axis, the diagonal, x = 1 */
    for (int i = -1; i <= 1; i++) {
      bad_atan2 += differ(atan2f((float)q, (float)i), atan2f_fdlibm((float)q, (float)i));
      bad_atan2 += differ(atan2f((float)-q, (float)i), atan2f_fdlibm((float)-q, (float)i));
      bad_atan2 += differ(atan2f((float)i, (float)q), atan2f_fdlibm((float)i, (float)q));
      bad_atan2 += differ(atan2f((float)q, (float)q), atan2f_fdlibm((float)q, (float)q));
      n_atan2 += 4;
    }
  for (uint32_t u = 0x3c000000u; u < 0x47000000u; u += 13) {
    const float x = i2f((int32_t)u);
    bad_log += differ(log10f(x), log10f_near(x));
    n_log++;
  }
  printf("%ld %ld %ld %ld %ld %ld %ld\n", bad_atan, n_atan, bad_atan2, n_atan2, bad_q, bad_log, n_log);
  return 0;
}
"""


def _glibc_version():
    import ctypes
    f = ctypes.CDLL(None).gnu_get_libc_version
    f.restype = ctypes.c_char_p
    return tuple(int(x) for x in f().decode().split(".")[:2])


@pytest.mark.skipif(_glibc_version() > (2, 40), reason="glibc 2.41+ ships correctly rounded CORE-MATH atanf / atan2f: the installed "
                    "libm is no longer the reference build's fdlibm (the library calls its own restatement on host and device)")
def test_float_arctangents_are_the_c_librarys_bit_for_bit(tmp_path):
    src = tmp_path / "libm_check.cpp"
    src.write_text(PROGRAM)
    exe = tmp_path / "libm_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "stm32f4_sdr_gps_amd", "csrc"),
                           "-o", str(exe), str(src), "-lm"])
    out = subprocess.check_output([str(exe)], text=True).split()
    bad_atan, n_atan, bad_atan2, n_atan2, bad_q, bad_log, n_log = map(int, out)
    print("atanf", bad_atan, "of", n_atan, "| atan2f", bad_atan2, "of", n_atan2, "| atanf(q / i)", bad_q, "| log10f_near", bad_log, "of", n_log)
    assert n_atan > 6e8 and n_atan2 > 1.7e7
    assert bad_atan == 0 and bad_atan2 == 0 and bad_q == 0
    assert bad_log < 0.003 * n_log
