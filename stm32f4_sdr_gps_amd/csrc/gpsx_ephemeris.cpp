// gpsx_ephemeris.cpp -- default gps_nav_data_decode_subframe: LNAV subframes 1-3 -> broadcast ephemeris record.
//
// Host code, no GPU.  Same results as the reference's decoder (PM/GPS/nav_data_decode.c:34-141, itself after GNSS-SDRLIB):
// the same fields at the same bit positions of the 300-bit subframe image (IS-GPS-200 20.3.3.3-20.3.3.4), the same scale
// factors applied in the same order in double arithmetic, the same counters and masks -- written as field tables.
// The subframe image is the one the word layer builds (gpsx_steps.cpp): bit n of the subframe is bit (n & 7) of byte
// n >> 3, parity bits included, data bits already un-inverted.
#include <cstddef>
#include <cstdint>

#include "../../include/gpsx_compat.h"

namespace {

constexpr double kSemiCircle = 3.1415926535898;   // IS-GPS-200's pi, as the reference's SC2RAD
constexpr int kBuildWeek = 2290;                  // PM/config.h:73: resolves the 10-bit week number's roll-over
constexpr long kUnixToGps = 315964800;            // s between the Unix and the GPS epoch

uint32_t take(const uint8_t *sf, int pos, int len)   // len bits from subframe bit pos, first bit most significant
{
  uint32_t v = 0;
  for (int i = pos; i < pos + len; i++)
    v = (v << 1) | ((sf[i >> 3] >> (i & 7)) & 1u);
  return v;
}

// A field: one or two bit runs (the split ones straddle a word's parity bits), two's complement or not, a power-of-two
// scale, optionally semicircles -> radians, and where it goes.
struct Field {
  int p1, l1, p2, l2;
  bool is_signed;
  int scale_exp;        // value * 2^scale_exp
  bool semicircles;
  size_t offset;        // in eph_t
  bool to_int;
};

// 2^e as the reference multiplies by it.  RTKLIB writes its scale factors as 16-digit decimal literals, and three of
// them (2^-33, 2^-43, 2^-55) are one or two ulps BELOW the power of two once parsed; the eccentricity, the rates and the
// clock drift rate inherit that, so the same literals are used here.
double scale_factor(int e)
{
  switch (e) {
  case 4: return 16.0;
  case -5: return 0.03125;
  case -19: return 1.907348632812500E-06;
  case -29: return 1.862645149230957E-09;
  case -31: return 4.656612873077393E-10;
  case -33: return 1.164153218269348E-10;
  case -43: return 1.136868377216160E-13;
  case -55: return 2.775557561562891E-17;
  default: return 1.0;
  }
}

double field_value(const uint8_t *sf, const Field &f, uint32_t *raw_out = nullptr)
{
  const int len = f.l1 + f.l2;
  uint32_t raw = take(sf, f.p1, f.l1);
  if (f.l2)
    raw = (raw << f.l2) | take(sf, f.p2, f.l2);
  if (raw_out)
    *raw_out = raw;
  double v;
  if (f.is_signed && len < 32 && (raw >> (len - 1)))
    v = (double)(int32_t)(raw | (~0u << len));
  else if (f.is_signed)
    v = (double)(int32_t)raw;
  else
    v = (double)raw;
  if (f.scale_exp)
    v = v * scale_factor(f.scale_exp);
  if (f.semicircles)
    v = v * kSemiCircle;
  return v;
}

void store(eph_t &e, const uint8_t *sf, const Field *fields, size_t n)
{
  for (size_t i = 0; i < n; i++) {
    const double v = field_value(sf, fields[i]);
    char *dst = reinterpret_cast<char *>(&e) + fields[i].offset;
    if (fields[i].to_int)
      *reinterpret_cast<int *>(dst) = (int)v;
    else
      *reinterpret_cast<double *>(dst) = v;
  }
}

gtime_t gps_time(int week, double sec)   // RTKLIB's gpst2time as the reference carries it (rtklib_common.c)
{
  gtime_t t;
  if (sec < -1e9 || 1e9 < sec)
    sec = 0.0;
  t.time = (time_t)kUnixToGps + (time_t)(86400 * 7 * week + (int)sec);
  t.sec = sec - (int)sec;
  return t;
}

#define F_INT(p, l, member) {p, l, 0, 0, false, 0, false, offsetof(eph_t, member), true}
#define F_U(p, l, exp, member) {p, l, 0, 0, false, exp, false, offsetof(eph_t, member), false}
#define F_S(p, l, exp, member) {p, l, 0, 0, true, exp, false, offsetof(eph_t, member), false}
#define F_S_SC(p, l, exp, member) {p, l, 0, 0, true, exp, true, offsetof(eph_t, member), false}
#define F_S2_SC(p1, l1, p2, l2, exp, member) {p1, l1, p2, l2, true, exp, true, offsetof(eph_t, member), false}
#define F_U2(p1, l1, p2, l2, exp, member) {p1, l1, p2, l2, false, exp, false, offsetof(eph_t, member), false}

const Field kSubframe1[] = {
    F_INT(70, 2, code), F_INT(72, 4, sva), F_INT(76, 6, svh), F_INT(90, 1, flag),
    F_S(196, 8, -31, tgd[0]), F_S(240, 8, -55, f2), F_S(248, 16, -43, f1), F_S(270, 22, -31, f0),
};
const Field kSubframe2[] = {
    F_INT(60, 8, iode), F_S(68, 16, -5, crs), F_S_SC(90, 16, -43, deln), F_S2_SC(106, 8, 120, 24, -31, M0),
    F_S(150, 16, -29, cuc), F_U2(166, 8, 180, 24, -33, e), F_S(210, 16, -29, cus), F_U(270, 16, 4, toes), F_U(286, 1, 0, fit),
};
const Field kSubframe3[] = {
    F_S(60, 16, -29, cic), F_S2_SC(76, 8, 90, 24, -31, OMG0), F_S(120, 16, -29, cis), F_S2_SC(136, 8, 150, 24, -31, i0),
    F_S(180, 16, -5, crc), F_S2_SC(196, 8, 210, 24, -31, omg), F_S_SC(240, 24, -43, OMGd), F_INT(270, 8, iode),
    F_S_SC(278, 14, -43, idot),
};

}  // namespace

extern "C" __attribute__((weak)) uint8_t gps_nav_data_decode_subframe(gps_ch_t *channel)
{
  const uint8_t *sf = channel->nav_data.subframe_data;
  sdreph_t &s = channel->eph_data;
  eph_t &e = s.eph;
  const uint32_t id = take(sf, 49, 3);   // hand-over word, bits 20-22
  e.sat = channel->prn;
  const double tow = (double)take(sf, 30, 17) * 6.0;   // hand-over word: time of week of the NEXT subframe, 6 s units
  switch (id) {
  case 1: {
    s.tow_gpst = tow;
    const int week10 = (int)take(sf, 60, 10) + 1024;
    store(e, sf, kSubframe1, sizeof kSubframe1 / sizeof kSubframe1[0]);
    e.iodc = (int)((take(sf, 82, 2) << 8) + take(sf, 210, 8));
    const double toc = (double)take(sf, 218, 16) * 16.0;
    e.week = week10 + (kBuildWeek - week10 + 512) / 1024 * 1024;
    s.week_gpst = e.week;
    e.ttr = gps_time(e.week, s.tow_gpst);
    e.toc = gps_time(e.week, toc);
    s.cnt++;
    break;
  }
  case 2: {
    s.tow_gpst = tow;
    store(e, sf, kSubframe2, sizeof kSubframe2 / sizeof kSubframe2[0]);
    const Field root_a = F_U2(226, 8, 240, 24, -19, A);
    const double sqrt_a = field_value(sf, root_a);
    e.A = sqrt_a * sqrt_a;
    e.toe = gps_time(e.week, e.toes);
    s.cnt++;
    break;
  }
  case 3:
    s.tow_gpst = tow;
    store(e, sf, kSubframe3, sizeof kSubframe3 / sizeof kSubframe3[0]);
    s.cnt++;
    break;
  case 4:
    s.tow_gpst = tow;
    s.cnt++;
    break;
  case 5:
    s.tow_gpst = tow;
    break;
  default:
    break;
  }
  if (id >= 1 && id <= 5) {
    s.received_mask |= (uint8_t)(1u << (id - 1));
    s.received_mask_proc |= (uint8_t)(1u << (id - 1));
  }
  s.sub_cnt++;
  return (uint8_t)id;
}
