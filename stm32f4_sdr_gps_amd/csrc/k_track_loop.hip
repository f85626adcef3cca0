// k_track_loop.hip -- the tracking loops on the device (include/gpsx.h "the tracking LOOPS on the device").
//
// One launch advances every channel by K milliseconds: per millisecond the wave-per-channel E/P/L correlators of
// gpsx_track_wave.hpp (bit-exact: the same device function k_track_epl_wave launches), then, for the up to sixteen channels of
// a wave at once, what gps_tracking_data_process does with the six accumulators (PM/GPS/tracking.c:132-170):
//   gps_tracking_dll :338-393   gps_tracking_pll :175-205   gps_tracking_fll :208-256 with gps_tracking_pll_check :261-327
//   gps_nav_data_analyse_new_code, PM/GPS/nav_data.c:46-253 (20 ms bit period, bit edge, bit votes)   SNR :141-169
//   the polarity-deciding part of gps_nav_data_words_detection, nav_data.c:257-352 (nav_word_sync)
// restated from csrc/gpsx_steps.cpp's host versions expression by expression: the same float32 operations in the same order
// (this file is built with -ffp-contract=off and correctly rounded division like the rest), the float arctangents as the C
// library computes them (gpsx_libm.hpp): bit-identical to the host mode but for the double-precision atan2 of the PLL's
// IP <= 0 branch (one argument pair in ~2^29) and the SNR's logarithm (a display value; the host record gets the host's).
// Serving schedules (every millisecond / the reference's 17 ms four-channel multiplex), the false-lock draws' two sources and
// what stays in registers: in front of the kernel below.
//
// Lanes: lane 4 c + k of a wave holds channel c of the wave (k = 0 / 1 / 2 = Early / Prompt / Late in the correlators).
// All four lanes of a quad carry the channel's whole loop state and run the loops redundantly -- they all need the new
// code phase and carrier for the next millisecond's correlators, and a broadcast would cost what the arithmetic does.
// HBM traffic per launch: K x 2 KB of samples per workgroup (L2 hits after the first), 100 B of live state in and out (per slot
// under the multiplex), the 20 cold bytes where a bit completes, and K flag bytes per channel: the kernel is bound by the
// correlators' vector instructions exactly as k_track_epl_wave is.
#include <hip/hip_runtime.h>

#include <initializer_list>

#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"
#include "gpsx_libm.hpp"
#include "gpsx_track_wave.hpp"

namespace gpsx {

namespace {

constexpr double kPiD = 3.14159265358979323846;
constexpr float kDll1C1 = 1.0f, kDll1C2 = 300.0f;          // TRACKING_DLL1_C1 / _C2 (PM/config.h)
constexpr float kPll1C1 = 4.0f, kPll1C2 = 3000.0f, kPll2C1 = 8.0f, kPll2C2 = 5000.0f;
constexpr float kFll1C1 = 200.0f, kFll1C2 = 2000.0f;
constexpr int kPllBadThreshold = 80;                        // tracking.c:14
constexpr int kSnrLength = 200;                             // tracking.c:26
constexpr int kSearchStepHz = 500;                          // ACQ_SEARCH_STEP_HZ
constexpr int kSlotMs = 4, kSlots = 4;                      // TRACKING_CH_LENGTH, GPS_SAT_CNT (PM/config.h): the 17 ms multiplex

template <int K>
__device__ __forceinline__ u32 quad_get(u32 v)   // lane k of this lane's quad
{
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true);
}

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// The two four-entry int16 arrays of the state (pll_check_buf, slot_ip) live in one 64-bit register pair each while the loops
// run: indexing a register array with the millisecond's index would send the whole state to scratch memory.
struct Quad16 {
  unsigned long long v;
  __device__ __forceinline__ int get(int i) const { return (int16_t)(uint16_t)(v >> (16 * i)); }
  __device__ __forceinline__ void set(int i, int x) { v = (v & ~(0xFFFFull << (16 * i))) | ((unsigned long long)(uint16_t)x << (16 * i)); }
};
__device__ __forceinline__ Quad16 quad16_load(const int16_t (&a)[4])
{
  return Quad16{(unsigned long long)(uint16_t)a[0] | ((unsigned long long)(uint16_t)a[1] << 16) |
                ((unsigned long long)(uint16_t)a[2] << 32) | ((unsigned long long)(uint16_t)a[3] << 48)};
}
__device__ __forceinline__ void quad16_store(const Quad16 &q, int16_t (&a)[4])
{
#pragma unroll
  for (int i = 0; i < 4; i++)
    a[i] = (int16_t)q.get(i);
}

// gps_tracking_dll
__device__ __forceinline__ void loop_dll(gpsx_loop_state_t &s, int IE, int QE, int IL, int QL)
{
  const int32_t e2 = IE * IE + QE * QE, l2 = IL * IL + QL * QL;
  float err = (float)(e2 - l2) / (float)(e2 + l2);
  err = -err;
  const float dt = 0.001f;
  s.code_phase_fine += (kDll1C1 * (err - s.dll_code_err) + kDll1C2 * dt * err);
  const float span = (float)(2 * kChips * 8);
  bool wrapped = false;
  if (s.code_phase_fine < 0.0f) {
    s.code_phase_fine = span - s.code_phase_fine;   // sic (tracking.c:356-361)
    wrapped = true;
  } else if (s.code_phase_fine > span) {
    s.code_phase_fine = s.code_phase_fine - span;
    wrapped = true;
  }
  if (wrapped) {
    s.code_phase_fine_filt = -1.0f;
  } else if (s.code_phase_fine_filt >= 0.0f) {
    s.code_phase_fine_filt += s.code_phase_fine;
    s.code_filt_cnt++;
  }
  s.dll_code_err = err;
}

// gps_tracking_pll (index 0 only has an effect).  The reference's atan2f is glibc's float one, restated operation by
// operation (gpsx_libm.hpp); its double-precision atan2 on the other branch -> the device's, whose result rounded to float
// after the division is the reference's in all but ~1 case in 2^29.
__device__ __forceinline__ void loop_pll(gpsx_loop_state_t &s, int IP, int QP)
{
  float phase_err;   // in units of pi
  if (IP > 0)
    phase_err = (float)((double)gpsx_libm::atan2f_fdlibm((float)QP, (float)IP) / kPiD);
  else
    phase_err = (float)(atan2((double)(float)-QP, (double)(float)-IP) / kPiD);
  float step = phase_err - s.pll_code_err;
  if ((double)step > kPiD / 2)
    step = (float)(kPiD - (double)step);
  if ((double)step < -kPiD / 2)
    step = (float)(-kPiD - (double)step);
  const float dt = 0.001f;
  if (s.period_sync_ok_flag)
    s.if_freq_offset_hz -= kPll2C1 * step + (kPll2C2 * dt * phase_err);
  else
    s.if_freq_offset_hz -= kPll1C1 * step + (kPll1C2 * dt * phase_err);
  s.pll_code_err = phase_err;
}

// gps_tracking_pll_check's bookkeeping; true = the carrier must jump (counters already cleared, as the reference clears them
// before it draws)
__device__ __forceinline__ bool loop_false_lock_detect(gpsx_loop_state_t &s, Quad16 &chk, int index, int IP)
{
  chk.set(index, IP);
  if (index < 3)
    return false;
  int flips = 0;
  int prev = chk.get(0) > 0;
#pragma unroll
  for (int i = 1; i < 4; i++) {
    const int cur = chk.get(i) > 0;
    flips += cur != prev;
    prev = cur;
  }
  if (flips > 1) {
    if (s.pll_bad_state_cnt < 10)
      s.pll_bad_state_cnt++;
  } else if (s.pll_bad_state_cnt > 0) {
    s.pll_bad_state_cnt--;
  }
  if (s.pll_bad_state_cnt > 9)
    s.pll_bad_state_master_cnt++;
  else if (s.pll_bad_state_cnt == 0)
    s.pll_bad_state_master_cnt = 0;
  if (s.pll_bad_state_master_cnt <= kPllBadThreshold)
    return false;
  s.pll_bad_state_master_cnt = 0;
  s.pll_bad_state_cnt = 0;
  return true;
}

// the jump: a random carrier offset around the acquired one, at least 200 Hz from where the loop stands (tracking.c:309-326);
// the draws come from the channel's own xorshift32 (GPSX_DRAWS_XORSHIFT) -- libc's rand() is one sequence per PROCESS
__device__ __forceinline__ void loop_false_lock_jump(gpsx_loop_state_t &s)
{
  int16_t candidate;
  int delta;
  u32 x = s.rng ? s.rng : 0x9E3779B9u;
  do {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    const int r = (int)(x % (u32)kSearchStepHz);
    candidate = (int16_t)(s.found_freq_offset_hz - r + kSearchStepHz / 2);
    delta = (int16_t)((int16_t)s.if_freq_offset_hz - candidate);
  } while (iabs(delta) < 200);
  s.rng = x;
  s.reseed_count++;
  s.if_freq_offset_hz = (float)candidate;
}

__device__ __forceinline__ float atan_ratio(int q, int i)   // the reference's (i == 0) ? pi / 2 : atanf((float)q / (float)i)
{
  if (i == 0)
    return (float)(kPiD / 2);
  return gpsx_libm::atanf_fdlibm((float)q / (float)i);
}

// gps_tracking_fll behind its call of the check.  `before_cache`: atan_ratio(fll_old_q, fll_old_i) as far as this launch has
// already computed it -- this millisecond's "before" is the last millisecond's "now" (same inputs, same function; NaN = not
// there: the first millisecond of a launch, or after index 0, computes it afresh)
__device__ __forceinline__ void loop_fll(gpsx_loop_state_t &s, int index, int IP, int QP, float &before_cache)
{
  if (index == 0) {
    s.fll_old_i = (int16_t)IP;
    s.fll_old_q = (int16_t)QP;
    before_cache = __builtin_nanf("");
    return;
  }
  const float now = atan_ratio(QP, IP);
  const float before = before_cache == before_cache ? before_cache : atan_ratio(s.fll_old_q, s.fll_old_i);
  before_cache = now;
  float rot = now - before;
  if ((double)rot > kPiD / 2)
    rot = (float)(kPiD - (double)rot);
  if ((double)rot < -kPiD / 2)
    rot = (float)(-kPiD - (double)rot);
  float change = rot - s.fll_err;
  if ((double)change > kPiD / 2)
    change = (float)(kPiD - (double)change);
  if ((double)change < -kPiD / 2)
    change = (float)(-kPiD - (double)change);
  const float dt = 0.001f;
  const float hz = kFll1C1 * dt * change + (kFll1C2 * dt * rot);
  s.if_freq_offset_hz -= hz;
  s.fll_old_i = (int16_t)IP;
  s.fll_old_q = (int16_t)QP;
  s.fll_err = rot;
}

// nav_data.c:145-218: which millisecond of the 4 ms group holds the bit edge
// returns 0, or the edge (1 / 2) it located
__device__ __forceinline__ int nav_refine_edge(gpsx_loop_state_t &s, const Quad16 &sip)
{
  const int ip[4] = {sip.get(0), sip.get(1), sip.get(2), sip.get(3)};
  if (iabs(ip[1]) > iabs(ip[0]))
    return 0;
  if (ip[3] == 0)
    return 0;
  const float whole = (float)iabs(ip[0]) / (float)iabs(ip[3]);
  if (whole > 1.5f || whole < 0.7f)
    return 0;
  const int chip = (int16_t)((int16_t)s.code_phase_fine / 16);
  if (chip < 0 || chip > kChips)
    return 0;
  int edge = 0;
  if (chip < kChips / 4 || chip > kChips * 3 / 4) {
    if (ip[1] == 0)
      return 0;
    const float jump = (float)iabs(ip[0]) / (float)iabs(ip[1]);
    if (jump > 1.5f || jump < 0.7f)
      return 0;
    edge = chip < kChips / 4 ? 2 : 1;
  } else {
    const int d1 = (uint16_t)iabs(ip[0] - ip[1]), d2 = (uint16_t)iabs(ip[2] - ip[3]);
    if (d1 > d2) {
      if (d2 == 0)
        return 0;
      if ((float)d1 / (float)d2 < 2.5f)
        return 0;
      edge = 1;
    } else {
      if (d1 == 0)
        return 0;
      if ((float)d2 / (float)d1 < 2.5f)
        return 0;
      edge = 2;
    }
  }
  s.accurate_swap_time = (uint8_t)((s.slot_start_ticks + (u32)edge) % 20u);
  s.accurate_swap_ok = 1;
  return edge;
}

// IS-GPS-200 table 20-XIV as csrc/gpsx_steps.cpp holds it: source bits d1..d24 (bit i - 1) entering parity bits D25..D30
constexpr u32 parity_bits(std::initializer_list<int> bits)
{
  u32 m = 0;
  for (int b : bits)
    m |= 1u << (b - 1);
  return m;
}
constexpr u32 kParityMask[6] = {
    parity_bits({1, 2, 3, 5, 6, 10, 11, 12, 13, 14, 17, 18, 20, 23}),  parity_bits({2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24}),
    parity_bits({1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22}),   parity_bits({2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23}),
    parity_bits({1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24}), parity_bits({3, 5, 6, 8, 9, 10, 11, 13, 15, 19, 22, 23, 24}),
};
constexpr u32 kParityFromD30 = 0x1Au;      // parity bits 1, 3, 4 start from D30*, the others from D29*
constexpr u32 kPreambleBits = 0xD1u;       // 1 0 0 0 1 0 1 1, first bit in bit 0
constexpr u32 kBadPolarityTimeoutMs = 12000;

// The part of gps_nav_data_words_detection (PM/GPS/nav_data.c:257-352) that decides the data POLARITY, on a 30-bit word
// buffer: preamble hunt (upright / inverted), word collection, parity, the two-subframe timeout.  The polarity flag is used
// by the very next millisecond's vote and sign-change detection (nav_data.c:60-66), so it is decided here, where it is used;
// subframe images, time stamps and the ephemeris stay with the host's word layer, which sees the same bits at the same
// ticks and therefore takes the same decisions (gps_tracking_words_batch).
// ... on the state's COLD part: the fields a channel touches once per completed bit (20 ms) or once per SNR estimate (201 ms)
// do not ride in registers through the correlators -- they stay in HBM and are read, changed and written back where they are
// needed (GPSX_DRAWS_LIBC keeps them in registers: a channel that stops for its draw must leave its state in HBM untouched).
struct WordSync {
  u32 buf, ts, cnt, bit_cnt, inv_cnt, flags;   // word_buf, word_detection_timestamp, word_cnt, word_bit_cnt, inv_preabmle_cnt, word_flags
};
__device__ __forceinline__ WordSync word_sync_load(const gpsx_loop_state_t &g)
{
  return WordSync{g.word_buf, g.word_detection_timestamp, g.word_cnt, g.word_bit_cnt, g.inv_preabmle_cnt, g.word_flags};
}
__device__ __forceinline__ void word_sync_store(const WordSync &w, gpsx_loop_state_t &g)
{
  g.word_buf = w.buf;
  g.word_detection_timestamp = w.ts;
  g.word_cnt = (uint8_t)w.cnt;
  g.word_bit_cnt = (uint8_t)w.bit_cnt;
  g.inv_preabmle_cnt = (uint8_t)w.inv_cnt;
  g.word_flags = (uint8_t)w.flags;
}
// returns the data polarity flag after the bit (`inv` = before it)
__device__ __forceinline__ u32 nav_word_sync(WordSync &w, u32 inv, u32 new_bit, u32 now)
{
  u32 buf = w.buf;
  if (w.cnt == 0) {
    buf = (buf >> 1) | (new_bit << 29);
    if ((buf & 0xFFu) == kPreambleBits) {
      w.flags = (w.flags & ~3u) | ((buf >> 28) & 3u);   // old_D29, old_D30
      w.cnt = 1;
      w.bit_cnt = 0;
      w.inv_cnt = 0;
    }
    if (!(w.flags & 4u) && w.cnt == 0) {
      if ((buf & 0xFFu) == (kPreambleBits ^ 0xFFu))
        w.inv_cnt = (w.inv_cnt + 1u) & 0xFFu;           // (a uint8_t in the reference)
      if (w.inv_cnt >= 2)
        inv = 1;
    }
    if (w.flags & 4u) {
      if (now - w.ts > kBadPolarityTimeoutMs) {
        w.ts = now;
        w.flags &= ~4u;
        inv = 0;
      }
    }
    w.buf = buf;
    return inv;
  }
  buf = (buf & ~(1u << w.bit_cnt)) | (new_bit << w.bit_cnt);
  w.bit_cnt++;
  w.buf = buf;
  if (w.bit_cnt < 30)
    return inv;
  const u32 d29 = w.flags & 1u, d30 = (w.flags >> 1) & 1u;
  if (d30)
    buf ^= 0xFFFFFFu;          // the D30* inversion comes off the 24 data bits in place (nav_data.c:439-440)
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const u32 p = ((u32)__popc(buf & kParityMask[k]) & 1u) ^ (((kParityFromD30 >> k) & 1u) ? d30 : d29);
    ok = ok && ((buf >> (24 + k)) & 1u) == p;
  }
  if (!ok) {
    w.cnt = 0;
    w.buf = 0;
    return inv;
  }
  w.flags = ((buf >> 28) & 3u) | 4u;   // old_D29, old_D30 of the next word; polarity_found
  w.cnt++;
  w.bit_cnt = 0;
  w.ts = now;
  w.buf = buf;
  if (w.cnt == 10) {
    w.cnt = 0;
    w.buf = 0;
  }
  return inv;
}

// gps_nav_data_analyse_new_code on the channel's own slot state; returns flag bits 1 / 2 (a bit was completed / its value)
// and 5 / 6 (the bit edge inside the 20 ms grid was located this millisecond / it was edge 2, not 1)
__device__ __forceinline__ u32 nav_bit_sync(gpsx_loop_state_t &s, Quad16 &sip, int index, int IP, u32 now)
{
  u32 out = 0;
  u32 bit = IP > 0 ? 1u : 0u;
  if (s.inv_polarity_flag)
    bit ^= 1u;
  s.slot_bits = (uint8_t)((s.slot_bits & ~(1u << index)) | (bit << index));
  sip.set(index, IP);
  if (index == 0)
    s.slot_start_ticks = now;
  if (s.period_sync_ok_flag == 1) {   // nav_data.c:223-253
    const u32 rem = (now - s.old_swap_time) % 20u;
    if (rem < s.old_reminder) {
      const u32 nav_bit = s.last_bit_pos_cnt > s.last_bit_neg_cnt ? 1u : 0u;
      out = 2u | (nav_bit << 2);
      // (the caller runs the word sync on this bit: nothing below depends on the polarity it may change -- this millisecond's own
      //  vote was formed with the polarity of before, as in the reference)
      s.last_bit_pos_cnt = 0;
      s.last_bit_neg_cnt = 0;
    }
    if (bit)
      s.last_bit_pos_cnt++;
    else
      s.last_bit_neg_cnt++;
    s.old_reminder = (uint8_t)rem;
  }
  if (index < 3)
    return out;
  int flips = 0, flip_at = 0;
  u32 prev = s.slot_bits & 1u;
#pragma unroll
  for (int i = 1; i < 4; i++) {
    const u32 cur = (s.slot_bits >> i) & 1u;
    if (cur != prev) {
      flips++;
      flip_at = i;
    }
    prev = cur;
  }
  if (flips != 1)
    return out;
  const u32 edge_time = s.slot_start_ticks + (u32)flip_at;
  const u32 rem = (edge_time - s.old_swap_time) % 20u;
  if (rem < 2 || rem == 19) {
    if (s.right_period_cnt < 10)
      s.right_period_cnt++;
    if (s.right_period_cnt > 8)
      s.period_sync_ok_flag = 1;
  } else {
    if (s.right_period_cnt > 0)
      s.right_period_cnt--;
    if (s.right_period_cnt < 3)
      s.period_sync_ok_flag = 0;
  }
  s.old_swap_time = edge_time;
  if (s.period_sync_ok_flag && flip_at == 2) {
    const int edge = nav_refine_edge(s, sip);
    if (edge)
      out |= 32u | (edge == 2 ? 64u : 0u);
  }
  return out;
}

// returns 0, or 1 = a new estimate was made of (*latch_i, *latch_q), or 2 = the q == 0 case (snr_value = 1, no latch)
__device__ __forceinline__ int loop_snr(gpsx_loop_state_t &s, int IP, int QP, u32 &latch_i, u32 &latch_q)
{
  s.i_part_summ += (u32)iabs(IP);
  s.q_part_summ += (u32)iabs(QP);
  s.snr_summ_cnt++;
  if (s.snr_summ_cnt > kSnrLength) {
    if (s.q_part_summ == 0) {
      s.snr_value = 1.0f;
      latch_i = 0;
      latch_q = 0;
      return 2;   // sic: the sums are not cleared on this path (tracking.c:152-156)
    }
    const float ratio = (float)s.i_part_summ / (float)s.q_part_summ;
    s.snr_value = s.i_part_summ ? 10.0f * gpsx_libm::log10f_near(ratio) : 10.0f * log10f(ratio);   // (log10f(0) = -inf)
    latch_i = s.i_part_summ;   // what the estimate was made of: gpsx_loop_state_to_channel takes the host's logarithm
    latch_q = s.q_part_summ;
    s.snr_summ_cnt = 0;
    s.i_part_summ = 0;
    s.q_part_summ = 0;
    return 1;
  }
  return 0;
}

// the part of the state that rides in registers: everything in front of snr_i_latch
constexpr int kLiveDwords = 25;
static_assert(offsetof(gpsx_loop_state_t, snr_i_latch) == 4 * kLiveDwords && sizeof(gpsx_loop_state_t) == 120, "gpsx_loop_state_t layout");
struct LiveWords { u32 w[kLiveDwords]; };

}  // namespace

// Which channels a wave holds.  GPSX_SCHED_EVERY_MS: cpw consecutive channels, in registers for the whole launch.
// GPSX_SCHED_MUX17: the reference's receiver is four channels sharing one correlator in a 17 ms cycle (PM/main.c:139-152) --
// channel c is slot (c & 3) of receiver c >> 2 and is served on the ticks t with (t % 17) / 4 == slot, t % 17 != 16.  A wave
// then holds cpw RECEIVERS: on a slot's first millisecond it loads the states of that slot's channels, on its last it stores
// them (240 B of traffic per channel and slot: nothing next to the correlators' work), so that every wave works on every
// millisecond but the cycle's idle one -- the multiplex costs no occupancy.  (The second pass of GPSX_DRAWS_LIBC, one listed
// channel per wave, keeps its channel for the launch and idles through the other slots.)
// LIBC: GPSX_DRAWS_LIBC -- the whole state, cold part included, stays in registers and a stopped channel is not stored.
template <bool MUX, bool LIBC>
__global__ __launch_bounds__(256, LIBC ? 2 : 3) void k_track_loop(const uint8_t *__restrict__ if_blocks, u32 block_stride, int n_blocks,
                                                       int if_format, int if_hz, gpsx_loop_state_t *__restrict__ st, int n_ch, int cpw,
                                                       u32 first_tick, const u32 *__restrict__ chipbits_all,
                                                       const u32 *__restrict__ rep_all, uint8_t *__restrict__ flags,
                                                       gpsx_loop_trace_t *__restrict__ trace, u32 *__restrict__ bad_prn, int word_sync,
                                                       const int *__restrict__ ch_map, int n_map,
                                                       const gpsx_loop_reseed_t *__restrict__ reseeds,
                                                       gpsx_loop_event_t *__restrict__ events, u32 *__restrict__ n_events)
{
  __shared__ u32 s_x[2][512];        // this and the next millisecond's sign plane: one barrier per millisecond
  __shared__ uint2 s_carrier[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int c_l = lane >> 2, k_l = lane & 3;
  const bool swap = MUX && !(LIBC && ch_map);   // the wave holds receivers and swaps channel states at the slot boundaries
  int n_here;                        // channels (swap: receivers) of this wave; 0: an idle wave of the last workgroup still stages blocks
  int unit0;                         // the wave's first channel (swap: receiver)
  if (LIBC && ch_map) {              // the second pass of GPSX_DRAWS_LIBC: one listed channel per wave
    const int v = (int)blockIdx.x * 4 + wave;
    n_here = v < n_map ? 1 : 0;
    unit0 = n_here ? ch_map[v] : 0;
  } else {
    const int n_units = swap ? (n_ch + 3) >> 2 : n_ch;
    unit0 = ((int)blockIdx.x * 4 + wave) * cpw;
    n_here = unit0 < n_units ? min(cpw, n_units - unit0) : 0;
  }
  const bool in_wave = c_l < n_here;
  const int unit_l = unit0 + (in_wave ? c_l : 0);
  gpsx_loop_state_t s = {};
  int prn = 0, ch_l = swap ? 0 : unit_l, cur_slot = -1;
  bool have = in_wave;               // this lane's quad holds a channel right now
  Quad16 chk = {0}, sip = {0};
  bool stalled = false;
  u32 stalled_slots = 0;             // swap: the slots whose channel of this receiver stopped for its draw in this launch
  float fll_before = __builtin_nanf("");

  auto load_state = [&](int ch) {
    ch_l = ch;
    if (LIBC) {
      s = st[ch];
    } else {     // the live part only: the cold fields are never read from `s` in this instantiation
      const LiveWords lw = *reinterpret_cast<const LiveWords *>(&st[ch]);
      __builtin_memcpy(&s, &lw, sizeof lw);
    }
    prn = track_prn(s.prn, bad_prn, have && k_l == 0);
    chk = quad16_load(s.pll_check_buf);
    sip = quad16_load(s.slot_ip);
    fll_before = __builtin_nanf("");
  };
  auto store_state = [&]() {
    if (!(have && k_l == 0) || stalled)
      return;
    quad16_store(chk, s.pll_check_buf);
    quad16_store(sip, s.slot_ip);
    if (LIBC) {
      st[ch_l] = s;
    } else {
      LiveWords lw;
      __builtin_memcpy(&lw, &s, sizeof lw);
      *reinterpret_cast<LiveWords *>(&st[ch_l]) = lw;
    }
  };
  if (n_here && !swap)
    load_state(unit_l < n_ch ? unit_l : 0);
  const int slot_fixed = ch_l & 3;   // (MUX without swapping: the channel's slot)
  int state_from = 0;                // the millisecond of this launch at which the state now in registers was in HBM
  const int replay_from = (LIBC && ch_map && n_here) ? reseeds[ch_l].ms_from : 0;

#pragma unroll 1
  for (int ms = 0; ms < n_blocks; ms++) {
    u32 *sx = s_x[ms & 1];
    trkwave::stage_block(if_blocks + (size_t)ms * block_stride, if_format, sx, s_carrier);
    __syncthreads();
    if (!n_here)
      continue;
    if (LIBC && ms < replay_from)    // a replayed channel: its state in HBM is the one it had HERE (the launch's earlier
      continue;                      // milliseconds are the first pass's, and valid)
    const u32 now = first_tick + (u32)ms;
    int index = (int)(now & 3u);
    int n_act = n_here;              // quads with a channel this millisecond
    int slot_now = 0;
    if (MUX) {
      const u32 big = now % (u32)(kSlotMs * kSlots + 1);   // PM/main.c:139-152
      slot_now = (int)(big / (u32)kSlotMs);
      const bool idle = big == (u32)(kSlotMs * kSlots);
      if (swap) {
        n_act = idle ? 0 : max(0, min(n_here, (n_ch - slot_now - 4 * unit0 + 3) >> 2));
        if (n_act == 0) {            // the cycle's idle millisecond (or a ragged last receiver without this slot)
          const int c = 4 * unit_l + k_l;
          if (in_wave && c < n_ch) {
            flags[(size_t)ms * n_ch + c] = 0;
            if (trace)
              trace[(size_t)ms * n_ch + c] = gpsx_loop_trace_t{};
          }
          continue;
        }
        if (slot_now != cur_slot) {  // a slot begins (or the launch begins inside one): its channels' states come in
          cur_slot = slot_now;
          have = c_l < n_act;
          stalled = (stalled_slots >> slot_now) & 1u;
          load_state(have ? 4 * unit_l + slot_now : 0);
          state_from = ms;
        }
      } else if (idle || slot_now != slot_fixed) {
        // not this channel's slot (or the cycle's idle millisecond): nothing of the channel moves
        if (in_wave && k_l == 0) {
          flags[(size_t)ms * n_ch + ch_l] = 0;
          if (trace)
            trace[(size_t)ms * n_ch + ch_l] = gpsx_loop_trace_t{};
        }
        continue;
      }
      index = (int)(big % (u32)kSlotMs);
      // tracking.c:102-113: the carrier NCO kept running while the other channels were served
      u32 elapsed = now - s.prev_track_timestamp;
      if (elapsed > 50u)   // "startup check"
        elapsed = 1u;
      if (elapsed != 1u) {
        const u64 adv = (u64)nco_step_per_sample((float)if_hz + s.if_freq_offset_hz) * (u64)kSamples * (u64)((elapsed - 1u) & 0xFFu);
        s.if_freq_accum += (u32)adv;   // gps_rewind_if_phase, gps_misc.c:196-204
      }
    }
    s.prev_track_timestamp = now;
    const bool mine = have && k_l < 3;
    const int fine = (int)(int16_t)(int)s.code_phase_fine;
    const u32 step = nco_step_per_word((float)if_hz + s.if_freq_offset_hz);
    const u32 iq = trkwave::wave_epl(sx, s_carrier, lane, n_act, mine, prn, fine, step, s.if_freq_accum, chipbits_all, rep_all);
    const u32 e = quad_get<0>(iq), p = quad_get<1>(iq), l = quad_get<2>(iq);
    const int IE = (int16_t)(e & 0xFFFFu), QE = (int16_t)(e >> 16), IP = (int16_t)(p & 0xFFFFu), QP = (int16_t)(p >> 16);
    const int IL = (int16_t)(l & 0xFFFFu), QL = (int16_t)(l >> 16);
    s.if_freq_accum += step * (u32)kWords32;
    // tracking.c:132-170, in the reference's order
    loop_dll(s, IE, QE, IL, QL);
    if (index == 0)
      loop_pll(s, IP, QP);
    bool moved = false;
    if (loop_false_lock_detect(s, chk, index, IP)) {
      if (!LIBC) {
        loop_false_lock_jump(s);
        moved = true;
      } else {
        // GPSX_DRAWS_LIBC: the draw is the host's (libc's rand(), in the reference's order).  First pass: report and stop
        // advancing this channel -- its state in HBM stays what it was when it came into the registers (the launch's input;
        // under the multiplex the state at the start of the slot it stops in); second pass: replayed from there, the host's
        // candidate for this millisecond is in the table.
        const gpsx_loop_reseed_t r = reseeds[ch_l];
        if (r.ms == ms) {
          s.reseed_count++;
          s.if_freq_offset_hz = (float)(int16_t)r.candidate;
          moved = true;
        } else if (!stalled) {
          stalled = true;
          stalled_slots |= 1u << slot_now;
          if (have && k_l == 0) {
            const u32 ev = atomicAdd(n_events, 1u);
            events[ev] = gpsx_loop_event_t{ch_l, ms, (int)(int16_t)s.if_freq_offset_hz, (int)s.found_freq_offset_hz, state_from};
          }
        }
      }
    }
    loop_fll(s, index, IP, QP, fll_before);
    u32 flag = nav_bit_sync(s, sip, index, IP, now);
    if (word_sync && (flag & 2u)) {   // a navigation bit was completed: the polarity-deciding part of the word layer
      if (LIBC) {
        WordSync w = word_sync_load(s);
        s.inv_polarity_flag = (uint8_t)nav_word_sync(w, s.inv_polarity_flag, (flag >> 2) & 1u, now);
        word_sync_store(w, s);
      } else {
        u32 inv = s.inv_polarity_flag;
        if (have && k_l == 0) {       // the cold part, in HBM: one lane of the quad reads, changes and writes it
          WordSync w = word_sync_load(st[ch_l]);
          inv = nav_word_sync(w, inv, (flag >> 2) & 1u, now);
          word_sync_store(w, st[ch_l]);
        }
        s.inv_polarity_flag = (uint8_t)quad_get<0>(inv);   // (every lane of the quad runs the loops: they must agree)
      }
    }
    {
      u32 latch_i = 0, latch_q = 0;
      const int made = loop_snr(s, IP, QP, latch_i, latch_q);
      if (made) {
        if (LIBC) {
          s.snr_i_latch = latch_i;
          s.snr_q_latch = latch_q;
        } else if (have && k_l == 0) {
          st[ch_l].snr_i_latch = latch_i;
          st[ch_l].snr_q_latch = latch_q;
        }
      }
    }
    flag |= (IP > 0 ? 1u : 0u) | (s.period_sync_ok_flag ? 8u : 0u) | (moved ? 16u : 0u) | 128u;
    gpsx_loop_trace_t t = {};
    if (trace) {
      t.iq[0] = (int16_t)IE; t.iq[1] = (int16_t)QE; t.iq[2] = (int16_t)IP; t.iq[3] = (int16_t)QP; t.iq[4] = (int16_t)IL; t.iq[5] = (int16_t)QL;
      t.code_phase_fine = s.code_phase_fine;
      t.if_freq_offset_hz = s.if_freq_offset_hz;
      t.if_freq_accum = s.if_freq_accum;
    }
    if (swap) {
      // the quad's four lanes write the receiver's four channels: the served one's flag byte, zero for the others
      const int c = 4 * unit_l + k_l;
      if (in_wave && c < n_ch) {
        const bool served = k_l == slot_now && have;
        flags[(size_t)ms * n_ch + c] = served ? (uint8_t)flag : (uint8_t)0;
        if (trace)
          trace[(size_t)ms * n_ch + c] = served ? t : gpsx_loop_trace_t{};
      }
      if (index == kSlotMs - 1 || ms == n_blocks - 1) {   // the slot (or the launch) ends: the states go back
        store_state();
        cur_slot = -1;
      }
    } else if (have && k_l == 0) {
      flags[(size_t)ms * n_ch + ch_l] = (uint8_t)flag;
      if (trace)
        trace[(size_t)ms * n_ch + ch_l] = t;
    }
  }
  if (n_here && !swap)
    store_state();
}

// the host's word layer changed its mind about the data polarity of n channels
__global__ void k_loop_set_polarity(gpsx_loop_state_t *__restrict__ st, const int *__restrict__ channels,
                                    const uint8_t *__restrict__ values, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    st[channels[i]].inv_polarity_flag = values[i] ? 1 : 0;
}

void launch_loop_set_polarity(hipStream_t s, gpsx_loop_state_t *d_st, const int *d_channels, const uint8_t *d_values, int n)
{
  if (n > 0)
    hipLaunchKernelGGL(k_loop_set_polarity, dim3((n + 255) / 256), dim3(256), 0, s, d_st, d_channels, d_values, n);
}

// GPSX_DRAWS_LIBC: the host's candidates for the channels that reported a false lock, one upload, scattered to their slots
__global__ void k_loop_scatter_reseeds(gpsx_loop_reseed_t *__restrict__ table, const int *__restrict__ channels,
                                       const gpsx_loop_reseed_t *__restrict__ cand, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    table[channels[i]] = cand[i];
}

void launch_loop_scatter_reseeds(hipStream_t s, gpsx_loop_reseed_t *d_table, const int *d_channels, const gpsx_loop_reseed_t *d_cand, int n)
{
  if (n > 0)
    hipLaunchKernelGGL(k_loop_scatter_reseeds, dim3((n + 255) / 256), dim3(256), 0, s, d_table, d_channels, d_cand, n);
}

void launch_track_loop(hipStream_t s, const uint8_t *d_if_blocks, uint32_t block_stride, int n_blocks, int if_format, int if_hz,
                       gpsx_loop_state_t *d_st, int n_ch, uint32_t first_tick, int schedule, int word_sync,
                       const uint32_t *d_chipbits, const uint32_t *d_trk_rep, uint8_t *d_flags, gpsx_loop_trace_t *d_trace,
                       uint32_t *d_bad_prn, const int *d_ch_map, int n_map, const gpsx_loop_reseed_t *d_reseeds,
                       gpsx_loop_event_t *d_events, uint32_t *d_n_events)
{
  if (n_ch <= 0 || n_blocks <= 0 || (d_ch_map && n_map <= 0))
    return;
  const bool mux = schedule == GPSX_SCHED_MUX17, libc = d_events != nullptr;
  const int n_units = mux ? (n_ch + 3) / 4 : n_ch;   // MUX: a wave holds receivers (k_track_loop)
  int cpw = n_units / (4 * 256 * 4);   // as launch_track_epl: ~4 workgroups per CU, 16 channels per wave at most
  cpw = cpw < 1 ? 1 : (cpw > 16 ? 16 : cpw);
  const dim3 grid(d_ch_map ? (n_map + 3) / 4 : (n_units + 4 * cpw - 1) / (4 * cpw));
#define GPSX_LAUNCH_LOOP(M, L)                                                                                                  \
  hipLaunchKernelGGL((k_track_loop<M, L>), grid, dim3(256), 0, s, d_if_blocks, block_stride, n_blocks, if_format, if_hz, d_st, n_ch, \
                     cpw, first_tick, d_chipbits, d_trk_rep, d_flags, d_trace, d_bad_prn, word_sync, d_ch_map, n_map, d_reseeds,  \
                     d_events, d_n_events)
  if (mux && libc)
    GPSX_LAUNCH_LOOP(true, true);
  else if (mux)
    GPSX_LAUNCH_LOOP(true, false);
  else if (libc)
    GPSX_LAUNCH_LOOP(false, true);
  else
    GPSX_LAUNCH_LOOP(false, false);
#undef GPSX_LAUNCH_LOOP
}

}  // namespace gpsx
