// gpsx_pvt.cpp -- single-point position solution, the last consumer of the correlator path.
//
// What the reference computes in PM/GPS/RTK/solving.c (an RTKLIB pntpos subset, sliced there into < 1 ms pieces):
//   satposs  :910-964   transmission time t = t_rx - P / c - dt_sat, satellite position and clock at t
//   eph2pos  :1165-1216 Kepler orbit from the broadcast elements, harmonic corrections, earth rotation, relativity
//   rescode  :711-796   residuals P - (rho + sagnac + iono + tropo + c dt_rx - c dt_sat), design rows, variances
//   estpos   :376-449   iterated weighted least squares, |dx|^2 < 1e-8, at most 10 iterations
//   ionmodel :622-660 (Klobuchar), tropmodel :679-700 (Saastamoinen), varerr :591-597, lsq :1452
// restated here from the models, in its own structure (per-satellite records, a dense 7 x 7 normal-equation solve).  Host
// double precision only; parity with the reference is to a stated tolerance (tests/test_pvt.py), not bit for bit -- the
// normal equations are solved by a different elimination order.  Kept: the reference's seven unknowns (x, y, z, c dt and
// three unused inter-system biases tied to zero by pseudo-observations of variance 0.01), its variance model, its time
// arithmetic (timeadd never carries into the integer second: rtklib_common.c:47-54), its convergence test.
#include <cmath>
#include <cstring>
#include <ctime>

#include "../../include/gpsx_compat.h"
#include "gpsx_compat_internal.hpp"

namespace {

constexpr double kC = 299792458.0;
constexpr double kPi = 3.1415926535897932;
constexpr double kMu = 3.9860050E14;        // WGS-84 GM as IS-GPS-200 uses it
constexpr double kOmegaE = 7.2921151467E-5;
constexpr double kReWgs84 = 6378137.0;
constexpr double kFeWgs84 = 1.0 / 298.257223563;
constexpr int kNx = 7;                      // x, y, z, c dt, three tied-down biases
constexpr int kMaxIter = 10;
constexpr double kMaxDtoe = 7200.0;

double time_diff(gtime_t a, gtime_t b) { return difftime(a.time, b.time) + a.sec - b.sec; }

gtime_t time_add(gtime_t t, double sec)   // the reference's: no carry into t.time
{
  t.sec += sec;
  const double whole = std::floor(t.sec);
  t.sec += whole;
  t.sec -= whole;
  return t;
}

double gps_seconds_of_week(gtime_t t)
{
  const time_t s = t.time - 315964800;
  const int week = (int)(s / (86400 * 7));
  return (double)(s - (time_t)week * 86400 * 7) + t.sec;
}

const eph_t *select_eph(const nav_t *nav, int sat, gtime_t when)
{
  const eph_t *best = nullptr;
  double t_best = kMaxDtoe + 2.0;
  for (int i = 0; i < nav->n; i++) {
    const eph_t *e = nav->eph[i];
    if (e->sat != sat)
      continue;
    const double t = std::fabs(time_diff(e->toe, when));
    if (t > kMaxDtoe + 1.0)
      continue;
    if (t <= t_best) {
      best = e;
      t_best = t;
    }
  }
  return best;
}

double clock_poly(const eph_t &e, double t) { return e.f0 + e.f1 * t + e.f2 * t * t; }

double sat_clock_bias(const eph_t &e, gtime_t when)   // eph2clk: two fixed-point passes on the transmission time
{
  double t = time_diff(when, e.toc);
  for (int i = 0; i < 2; i++)
    t -= clock_poly(e, t);
  return clock_poly(e, t);
}

struct SatState {
  double pos[3] = {0, 0, 0};
  double clk = 0.0;
  double var = 0.0;
  bool ok = false;
  int health = 0;
};

double ura_variance(int idx)
{
  static const double m[] = {2.4, 3.4, 4.85, 6.85, 9.65, 13.65, 24.0, 48.0, 96.0, 192.0, 384.0, 768.0, 1536.0, 3072.0, 6144.0};
  const double v = (idx < 0 || idx > 15) ? 6144.0 : m[idx];   // (index 15 reads one past the reference's table too: keep to 14)
  return v * v;
}

// position, clock (with the relativistic term) and variance at `when`; false when the Kepler iteration does not settle
bool orbit(const eph_t &e, gtime_t when, double *pos, double *clk, double *var)
{
  if (e.A <= 0.0) {
    pos[0] = pos[1] = pos[2] = *clk = *var = 0.0;
    return true;
  }
  double tk = time_diff(when, e.toe);
  const double mean = e.M0 + (std::sqrt(kMu / (e.A * e.A * e.A)) + e.deln) * tk;
  double ecc = mean, prev = 0.0;
  int n = 0;
  for (; std::fabs(ecc - prev) > 1E-14 && n < 30; n++) {
    prev = ecc;
    ecc -= (ecc - e.e * std::sin(ecc) - mean) / (1.0 - e.e * std::cos(ecc));
  }
  if (n >= 30)
    return false;
  const double se = std::sin(ecc), ce = std::cos(ecc);
  double u = std::atan2(std::sqrt(1.0 - e.e * e.e) * se, ce - e.e) + e.omg;
  double r = e.A * (1.0 - e.e * ce);
  double inc = e.i0 + e.idot * tk;
  const double s2 = std::sin(2.0 * u), c2 = std::cos(2.0 * u);
  u += e.cus * s2 + e.cuc * c2;
  r += e.crs * s2 + e.crc * c2;
  inc += e.cis * s2 + e.cic * c2;
  const double x = r * std::cos(u), y = r * std::sin(u), ci = std::cos(inc);
  const double node = e.OMG0 + (e.OMGd - kOmegaE) * tk - kOmegaE * e.toes;
  const double sn = std::sin(node), cn = std::cos(node);
  pos[0] = x * cn - y * ci * sn;
  pos[1] = x * sn + y * ci * cn;
  pos[2] = y * std::sin(inc);
  tk = time_diff(when, e.toc);
  *clk = clock_poly(e, tk) - 2.0 * std::sqrt(kMu * e.A) * e.e * se / (kC * kC);
  *var = ura_variance(e.sva > 14 ? -1 : e.sva);
  return true;
}

void geodetic(const double *r, double *pos)   // ecef2pos: fixed-point iteration on z to 1e-4 m
{
  const double e2 = kFeWgs84 * (2.0 - kFeWgs84), r2 = r[0] * r[0] + r[1] * r[1];
  double v = kReWgs84, z = r[2], zk = 0.0;
  while (std::fabs(z - zk) >= 1E-4) {
    zk = z;
    const double sp = z / std::sqrt(r2 + z * z);
    v = kReWgs84 / std::sqrt(1.0 - e2 * sp * sp);
    z = r[2] + v * e2 * sp;
  }
  pos[0] = r2 > 1E-12 ? std::atan(z / std::sqrt(r2)) : (r[2] > 0.0 ? kPi / 2.0 : -kPi / 2.0);
  pos[1] = r2 > 1E-12 ? std::atan2(r[1], r[0]) : 0.0;
  pos[2] = std::sqrt(r2 + z * z) - v;
}

double klobuchar(gtime_t t, const double *ion_in, const double *pos, double az, double el)
{
  static const double dflt[8] = {0.1118E-07, -0.7451E-08, -0.5961E-07, 0.1192E-06, 0.1167E+06, -0.2294E+06, -0.1311E+06, 0.1049E+07};
  if (pos[2] < -1E3 || el <= 0)
    return 0.0;
  double nrm = 0.0;
  for (int i = 0; i < 8; i++)
    nrm += ion_in[i] * ion_in[i];
  const double *ion = nrm <= 0.0 ? dflt : ion_in;
  const double psi = 0.0137 / (el / kPi + 0.11) - 0.022;
  double phi = pos[0] / kPi + psi * std::cos(az);
  phi = phi > 0.416 ? 0.416 : (phi < -0.416 ? -0.416 : phi);
  const double lam = pos[1] / kPi + psi * std::sin(az) / std::cos(phi * kPi);
  phi += 0.064 * std::cos((lam - 1.617) * kPi);
  double tt = 43200.0 * lam + gps_seconds_of_week(t);
  tt -= std::floor(tt / 86400.0) * 86400.0;
  const double slant = 1.0 + 16.0 * std::pow(0.53 - el / kPi, 3.0);
  double amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3]));
  double per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]));
  amp = amp < 0.0 ? 0.0 : amp;
  per = per < 72000.0 ? 72000.0 : per;
  const double x = 2.0 * kPi * (tt - 50400.0) / per;
  return kC * slant * (std::fabs(x) < 1.57 ? 5E-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) : 5E-9);
}

double saastamoinen(const double *pos, double el, double humidity)
{
  if (pos[2] < -100.0 || 1E4 < pos[2] || el <= 0)
    return 0.0;
  const double hgt = pos[2] < 0.0 ? 0.0 : pos[2];
  const double pres = 1013.25 * std::pow(1.0 - 2.2557E-5 * hgt, 5.2568);
  const double temp = 15.0 - 6.5E-3 * hgt + 273.16;
  const double e = 6.108 * humidity * std::exp((17.15 * temp - 4684.0) / (temp - 38.45));
  const double z = kPi / 2.0 - el;
  const double dry = 0.0022768 * pres / (1.0 - 0.00266 * std::cos(2.0 * pos[0]) - 0.00028 * hgt / 1E3) / std::cos(z);
  const double wet = 0.002277 * (1255.0 / temp + 0.05) * e / std::cos(z);
  return dry + wet;
}

// solve (sum_k a_k a_k^T) x = sum_k a_k y_k for the kNx unknowns, q = the inverse of the normal matrix; false: singular
bool normal_solve(const double (*a)[kNx], const double *y, int rows, double *x, double *q)
{
  double m[kNx][2 * kNx], rhs[kNx];
  for (int i = 0; i < kNx; i++) {
    rhs[i] = 0.0;
    for (int j = 0; j < 2 * kNx; j++)
      m[i][j] = j >= kNx ? (j - kNx == i ? 1.0 : 0.0) : 0.0;
    for (int k = 0; k < rows; k++) {
      rhs[i] += a[k][i] * y[k];
      for (int j = 0; j < kNx; j++)
        m[i][j] += a[k][i] * a[k][j];
    }
  }
  for (int c = 0; c < kNx; c++) {   // Gauss-Jordan with partial pivoting on [N | I]
    int piv = c;
    for (int r = c + 1; r < kNx; r++)
      if (std::fabs(m[r][c]) > std::fabs(m[piv][c]))
        piv = r;
    if (std::fabs(m[piv][c]) < 1E-300)
      return false;
    if (piv != c)
      for (int j = 0; j < 2 * kNx; j++) {
        const double t = m[c][j];
        m[c][j] = m[piv][j];
        m[piv][j] = t;
      }
    const double d = m[c][c];
    for (int j = 0; j < 2 * kNx; j++)
      m[c][j] /= d;
    for (int r = 0; r < kNx; r++) {
      if (r == c)
        continue;
      const double f = m[r][c];
      if (f != 0.0)
        for (int j = 0; j < 2 * kNx; j++)
          m[r][j] -= f * m[c][j];
    }
  }
  for (int i = 0; i < kNx; i++) {
    x[i] = 0.0;
    for (int j = 0; j < kNx; j++) {
      q[i * kNx + j] = m[i][kNx + j];
      x[i] += m[i][kNx + j] * rhs[j];
    }
  }
  return true;
}

double g_azel[2 * GPSX_PVT_MAXSAT];
nav_t g_nav;
uint8_t g_phase = 0;   // gps_pos_solve: 0 idle, 1 a solution is waiting for its geodetic conversion

int solve(const obsd_t *obs, int n, const nav_t *nav, sol_t *sol)
{
  sol->stat = SOLQ_NONE;
  if (n <= 0)
    return 0;
  if (n > GPSX_PVT_MAXSAT)
    n = GPSX_PVT_MAXSAT;
  sol->time = obs[0].time;

  // ---- satellites at their transmission times (satposs) ----
  SatState sv[GPSX_PVT_MAXSAT];
  for (int i = 0; i < n; i++) {
    const eph_t *e = select_eph(nav, obs[i].sat, sol->time);
    if (!e)
      continue;
    gtime_t t = time_add(obs[i].time, -obs[i].P[0] / kC);
    t = time_add(t, -sat_clock_bias(*e, t));
    sv[i].health = -1;
    double var = 0.0;
    if (!orbit(*e, t, sv[i].pos, &sv[i].clk, &var))
      continue;   // (the reference leaves stale values here; a Kepler iteration that has not settled in 30 steps is no orbit)
    sv[i].var = var;
    sv[i].health = e->svh;
    sv[i].ok = true;
    if (sv[i].clk == 0.0)   // "no precise clock": the broadcast polynomial without the relativistic term
      sv[i].clk = sat_clock_bias(*e, t);
  }

  // ---- iterated weighted least squares (estpos / rescode) ----
  double x[kNx] = {sol->rr[0], sol->rr[1], sol->rr[2], 0, 0, 0, 0};
  for (int iter = 0; iter < kMaxIter; iter++) {
    double rows[GPSX_PVT_MAXSAT + 4][kNx], v[GPSX_PVT_MAXSAT + 4], var[GPSX_PVT_MAXSAT + 4];
    double geo[3];
    geodetic(x, geo);
    int nv = 0, ns = 0;
    for (int i = 0; i < n; i++) {
      g_azel[2 * i] = g_azel[2 * i + 1] = 0.0;
      if (i < n - 1 && obs[i].sat == obs[i + 1].sat) {   // duplicated observation: both dropped, as in the reference
        i++;
        continue;
      }
      const SatState &s = sv[i];
      const double rn = std::sqrt(s.pos[0] * s.pos[0] + s.pos[1] * s.pos[1] + s.pos[2] * s.pos[2]);
      if (rn < kReWgs84)
        continue;
      double los[3] = {s.pos[0] - x[0], s.pos[1] - x[1], s.pos[2] - x[2]};
      const double range = std::sqrt(los[0] * los[0] + los[1] * los[1] + los[2] * los[2]);
      for (double &c : los)
        c /= range;
      const double rho = range + kOmegaE * (s.pos[0] * x[1] - s.pos[1] * x[0]) / kC;   // Sagnac
      if (rho <= 0.0)
        continue;
      // azimuth / elevation in the local east-north-up frame
      double az = 0.0, el = kPi / 2.0;
      if (geo[2] > -kReWgs84) {
        const double sp = std::sin(geo[0]), cp = std::cos(geo[0]), sl = std::sin(geo[1]), cl = std::cos(geo[1]);
        const double east = -sl * los[0] + cl * los[1];
        const double north = -sp * cl * los[0] - sp * sl * los[1] + cp * los[2];
        const double up = cp * cl * los[0] + cp * sl * los[1] + sp * los[2];
        az = east * east + north * north < 1E-12 ? 0.0 : std::atan2(east, north);
        if (az < 0.0)
          az += 2 * kPi;
        el = std::asin(up);
      }
      g_azel[2 * i] = az;
      g_azel[2 * i + 1] = el;
      if (el < 0.0)
        continue;
      double tgd = 0.0;
      for (int k = 0; k < nav->n; k++)
        if (nav->eph[k]->sat == obs[i].sat) {
          tgd = kC * nav->eph[k]->tgd[0];
          break;
        }
      if (s.health)
        continue;
      const double ion = klobuchar(obs[i].time, nav->ion_gps, geo, az, el);
      const double trp = saastamoinen(geo, el, 0.7);
      const double v_ion = (ion * 0.5) * (ion * 0.5);
      const double sin_el = std::sin(el);
      const double v_trp = (0.3 / (sin_el + 0.1)) * (0.3 / (sin_el + 0.1));
      const double ev = 0.003 * 0.003;
      const double v_meas = ev * (ev + ev / sin_el);
      v[nv] = (obs[i].P[0] - tgd) - (rho + ion + trp + x[3] - kC * s.clk);
      for (int j = 0; j < kNx; j++)
        rows[nv][j] = j < 3 ? -los[j] : (j == 3 ? 1.0 : 0.0);
      var[nv] = v_meas + s.var + v_ion + v_trp;
      nv++;
      ns++;
    }
    for (int b = 4; b < kNx; b++) {   // the three unused bias unknowns, tied to zero
      v[nv] = 0.0;
      for (int j = 0; j < kNx; j++)
        rows[nv][j] = j == b ? 1.0 : 0.0;
      var[nv] = 0.01;
      nv++;
    }
    if (nv < kNx)
      break;
    for (int k = 0; k < nv; k++) {
      const double sig = std::sqrt(var[k]);
      v[k] /= sig;
      for (int j = 0; j < kNx; j++)
        rows[k][j] /= sig;
    }
    double dx[kNx], q[kNx * kNx];
    if (!normal_solve(rows, v, nv, dx, q))
      break;
    for (int j = 0; j < kNx; j++)
      x[j] += dx[j];
    if (dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2] + dx[3] * dx[3] < 1E-8) {
      sol->type = 0;
      sol->time = time_add(obs[0].time, -x[3] / kC);
      sol->dtr[0] = x[3] / kC;
      for (int j = 0; j < 6; j++)
        sol->rr[j] = j < 3 ? x[j] : 0.0;
      for (int j = 0; j < 3; j++)
        sol->qr[j] = (float)q[j + j * kNx];
      sol->qr[3] = (float)q[1];
      sol->qr[4] = (float)q[2 + kNx];
      sol->qr[5] = (float)q[2];
      sol->ns = (unsigned char)ns;
      sol->age = sol->ratio = 0.0f;
      sol->stat = SOLQ_SINGLE;
      for (int i = 0; i < 2 * GPSX_PVT_MAXSAT; i++)
        g_azel[i] *= 180.0 / kPi;
      return 1;
    }
  }
  sol->stat = SOLQ_NONE;
  for (int i = 0; i < 2 * GPSX_PVT_MAXSAT; i++)
    g_azel[i] *= 180.0 / kPi;
  return 0;
}

}  // namespace

extern "C" {

sol_t gps_sol;
double final_pos[3];

}

void gpsx_pvt_reset()
{
  g_phase = 0;
  std::memset(&gps_sol, 0, sizeof gps_sol);
  std::memset(final_pos, 0, sizeof final_pos);
  std::memset(g_azel, 0, sizeof g_azel);
  std::memset(&g_nav, 0, sizeof g_nav);
}

extern "C" {

const double *gpsx_pvt_azel(void) { return g_azel; }

void ecef2pos(const double *r, double *pos) { geodetic(r, pos); }

int pntpos(const obsd_t *obs, int n, const nav_t *nav, sol_t *sol) { return solve(obs, n, nav, sol); }

int pntpos_iterative(const obsd_t *obs, int n, const nav_t *nav, sol_t *sol)
{
  if (n <= 0) {
    sol->stat = SOLQ_NONE;
    return -2;
  }
  return solve(obs, n, nav, sol) > 0 ? 1 : -1;
}

void gps_pos_solve_init(gps_ch_t *channels)
{
  for (int i = 0; i < GPS_SAT_CNT; i++)
    g_nav.eph[i] = &channels[i].eph_data.eph;
  g_nav.n = GPS_SAT_CNT;
  std::memset(g_nav.ion_gps, 0, sizeof g_nav.ion_gps);
}

void gps_pos_solve(obsd_t *obs)
{
  if (g_phase) {   // second call: the geodetic form of the solution found by the first
    geodetic(gps_sol.rr, final_pos);
    final_pos[0] *= 180.0 / kPi;
    final_pos[1] *= 180.0 / kPi;
    g_phase = 0;
  } else if (pntpos_iterative(obs, GPS_SAT_CNT, &g_nav, &gps_sol) > 0) {
    g_phase = 1;
  }
}

uint8_t solving_is_busy(void) { return g_phase; }

// Channel observations -> the solver's observation records (rtklib_common.c:75-92): the step between the pseudorange
// calculation and gps_pos_solve.  The time of reception is week + tow_s (gpst2time: out-of-range seconds read as 0, whole
// seconds into .time, the fraction into .sec); the SNR byte is the reference's `(unsigned char)(snr + 20.0f) * 4` -- the cast
// binds to the sum, the product is then truncated to a byte.
void sdrobs2obsd(gps_ch_t *channels, int ns, obsd_t *out)
{
  for (int i = 0; i < ns; i++) {
    const gps_ch_t &ch = channels[i];
    double sec = ch.obs_data.tow_s;
    if (sec < -1E9 || 1E9 < sec)
      sec = 0.0;
    gtime_t t;
    t.time = (time_t)315964800 + (time_t)(86400 * 7 * ch.eph_data.week_gpst + (int)sec);   // (int arithmetic, as the reference's)
    t.sec = sec - (int)sec;
    out[i].time = t;
    out[i].rcv = 1;
    out[i].sat = ch.prn;
    out[i].P[0] = ch.obs_data.pseudorange_m;
    out[i].L[0] = 0;
    out[i].D[0] = (float)ch.tracking_data.if_freq_offset_hz;
    // the reference converts the float straight to unsigned char (rtklib_common.c:86), which is undefined below 0 -- a channel
    // whose SNR estimate is under -20 dB (UBSan: -3.69568 on the config-2 trace).  What x86 does with it -- truncate to a 32-bit
    // integer, keep the low byte -- is written out here, so the byte is the reference's and the conversion is defined
    out[i].SNR[0] = (unsigned char)((unsigned char)(int)(ch.tracking_data.snr_value + 20.0f) * 4);
    out[i].LLI[0] = 0;
    out[i].code[0] = 1;   // CODE_L1C
  }
}

}  // extern "C"
