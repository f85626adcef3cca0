// gpsx_libm.hpp -- the three libm functions the reference's tracking loops call (PM/GPS/tracking.c:175-256 atan2f / atanf,
// :141-169 log10f), as the x86 build of the reference gets them from glibc 2.35 -- the libm the oracle (the reference's own
// C, compiled in place) and the golden traces ran on.  glibc's float arctangent is the classic fdlibm one (Sun's
// s_atanf.c / e_atan2f.c: argument reduction to four intervals, an 11-term odd polynomial in float arithmetic); it is
// faithfully, not correctly, rounded -- 4.8 % of the quotients I/Q the loops feed it come out one ulp away from the
// correctly rounded value -- so a device loop that wants the reference's bits has to do the reference's float operations.
// The HOST mode's loops (csrc/gpsx_steps.cpp) call these as well, not the installed libm: glibc 2.41 and later ship correctly
// rounded CORE-MATH arctangents, and a host on such a glibc would otherwise leave the reference build's (and the device loops')
// trajectory at the ulp level.  The bits in question are therefore those of the reference as built on glibc <= 2.40.
// Restated from the published algorithm; tests/test_libm_restatement.py compiles this header for the host and compares it
// with the C library bit for bit (every 7th float for atanf, a grid of integer pairs for atan2f).  Float arithmetic only,
// no contraction (the whole library is built with -ffp-contract=off), correctly rounded division.
//
// log10f is fdlibm's too (k + mantissa split, ivln10 * logf(m)), but its inner logf is glibc's table-driven one, which is not
// restated here: log10f_near uses the correctly rounded logarithm instead and agrees with glibc on 99.85 % of the floats.  The
// one consumer (snr_value, a display value nothing feeds back) is made exact another way: the device latches the two sums
// the estimate is made of and gpsx_loop_state_to_channel takes the logarithm on the host (csrc/gpsx_compat.cpp).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GPSX_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define GPSX_HD inline
#endif

namespace gpsx_libm {

GPSX_HD int32_t f2i(float x) { return __builtin_bit_cast(int32_t, x); }
GPSX_HD float i2f(int32_t i) { return __builtin_bit_cast(float, i); }

GPSX_HD float atanf_fdlibm(float x)
{
  // atan(0.5), atan(1), atan(1.5), atan(inf): high and low parts
  const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
  const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
  const float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f, a3 = -1.1111110449e-01f,
              a4 = 9.0908870101e-02f, a5 = -7.6918758452e-02f, a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f,
              a8 = 4.9768779427e-02f, a9 = -3.6531571299e-02f, a10 = 1.6285819933e-02f;
  const int32_t hx = f2i(x), ix = hx & 0x7fffffff;
  float hi = 0.0f, lo = 0.0f;
  bool reduced = true;
  if (ix >= 0x4c000000) {                 // |x| >= 2^25 (or NaN)
    if (ix > 0x7f800000)
      return x + x;
    return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
  }
  if (ix < 0x3ee00000) {                  // |x| < 7/16: the polynomial directly
    if (ix < 0x31000000)                  // |x| < 2^-29
      return x;
    reduced = false;
  } else {
    x = i2f(ix);
    if (ix < 0x3f980000) {                // |x| < 19/16
      if (ix < 0x3f300000) { hi = hi0; lo = lo0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else                 { hi = hi1; lo = lo1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { hi = hi2; lo = lo2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else                 { hi = hi3; lo = lo3; x = -1.0f / x; }
    }
  }
  const float z = x * x, w = z * z;
  const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  if (!reduced)
    return x - x * (s1 + s2);
  const float r = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -r : r;
}

// finite arguments
GPSX_HD float atan2f_fdlibm(float y, float x)
{
  const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int32_t hx = f2i(x), hy = f2i(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (hx == 0x3f800000)
    return atanf_fdlibm(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);   // 2 sign(x) + sign(y)
  if (iy == 0) {
    if (m < 2)
      return y;
    return m == 2 ? pi + tiny : -pi - tiny;
  }
  if (ix == 0)
    return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60)
    z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60)
    z = 0.0f;
  else
    z = atanf_fdlibm(i2f(f2i(y / x) & 0x7fffffff));
  if (m == 0)
    return z;
  if (m == 1)
    return i2f(f2i(z) ^ (int32_t)0x80000000);
  if (m == 2)
    return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}

// positive, finite, normal x
GPSX_HD float log10f_near(float x)
{
  const float ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
  int32_t hx = f2i(x);
  const int32_t k = (hx >> 23) - 127;
  const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
  hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
  const float y = (float)(k + i);
  const float z = y * log10_2lo + ivln10 * (float)log((double)i2f(hx));
  return z + y * log10_2hi;
}

}  // namespace gpsx_libm
