// k_track_loop.hip -- the tracking loops on the device (include/gpsx.h "the tracking LOOPS on the device").
//
// One launch advances every channel by K milliseconds: per millisecond the wave-per-channel E/P/L correlators of
// gpsx_track_wave.hpp (bit-exact: the same device function k_track_epl_wave launches), then, for the up to sixteen channels of
// a wave at once, what gps_tracking_data_process does with the six accumulators (PM/GPS/tracking.c:132-170):
//   gps_tracking_dll :338-393   gps_tracking_pll :175-205   gps_tracking_fll :208-256 with gps_tracking_pll_check :261-327
//   gps_nav_data_analyse_new_code, PM/GPS/nav_data.c:46-253 (20 ms bit period, bit edge, bit votes)   SNR :141-169
// restated from csrc/gpsx_steps.cpp's host versions expression by expression: the same float32 operations in the same order
// (this file is built with -ffp-contract=off and correctly rounded division like the rest), so that everything but the
// three arctangents and the SNR's logarithm is bit-identical to the host mode.
//
// Lanes: lane 4 c + k of a wave holds channel c of the wave (k = 0 / 1 / 2 = Early / Prompt / Late in the correlators).
// All four lanes of a quad carry the channel's whole loop state and run the loops redundantly -- they all need the new
// code phase and carrier for the next millisecond's correlators, and a broadcast would cost what the arithmetic does.
// HBM traffic per launch: K x 2 KB of samples per workgroup (L2 hits after the first), 96 B of state in and out and K flag
// bytes per channel: the kernel is bound by the correlators' vector instructions exactly as k_track_epl_wave is.
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

// gps_tracking_fll behind its call of the check
__device__ __forceinline__ void loop_fll(gpsx_loop_state_t &s, int index, int IP, int QP)
{
  if (index == 0) {
    s.fll_old_i = (int16_t)IP;
    s.fll_old_q = (int16_t)QP;
    return;
  }
  const float now = atan_ratio(QP, IP), before = atan_ratio(s.fll_old_q, s.fll_old_i);
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
__device__ __forceinline__ void nav_word_sync(gpsx_loop_state_t &s, u32 new_bit, u32 now)
{
  u32 buf = s.word_buf;
  if (s.word_cnt == 0) {
    buf = (buf >> 1) | (new_bit << 29);
    if ((buf & 0xFFu) == kPreambleBits) {
      s.word_flags = (uint8_t)((s.word_flags & ~3u) | ((buf >> 28) & 3u));   // old_D29, old_D30
      s.word_cnt = 1;
      s.word_bit_cnt = 0;
      s.inv_preabmle_cnt = 0;
    }
    if (!(s.word_flags & 4u) && s.word_cnt == 0) {
      if ((buf & 0xFFu) == (kPreambleBits ^ 0xFFu))
        s.inv_preabmle_cnt++;
      if (s.inv_preabmle_cnt >= 2)
        s.inv_polarity_flag = 1;
    }
    if (s.word_flags & 4u) {
      if (now - s.word_detection_timestamp > kBadPolarityTimeoutMs) {
        s.word_detection_timestamp = now;
        s.word_flags &= (uint8_t)~4u;
        s.inv_polarity_flag = 0;
      }
    }
    s.word_buf = buf;
    return;
  }
  buf = (buf & ~(1u << s.word_bit_cnt)) | (new_bit << s.word_bit_cnt);
  s.word_bit_cnt++;
  s.word_buf = buf;
  if (s.word_bit_cnt < 30)
    return;
  const u32 d29 = s.word_flags & 1u, d30 = (s.word_flags >> 1) & 1u;
  if (d30)
    buf ^= 0xFFFFFFu;          // the D30* inversion comes off the 24 data bits in place (nav_data.c:439-440)
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const u32 p = ((u32)__popc(buf & kParityMask[k]) & 1u) ^ (((kParityFromD30 >> k) & 1u) ? d30 : d29);
    ok = ok && ((buf >> (24 + k)) & 1u) == p;
  }
  if (!ok) {
    s.word_cnt = 0;
    s.word_buf = 0;
    return;
  }
  s.word_flags = (uint8_t)(((buf >> 28) & 3u) | 4u);   // old_D29, old_D30 of the next word; polarity_found
  s.word_cnt++;
  s.word_bit_cnt = 0;
  s.word_detection_timestamp = now;
  s.word_buf = buf;
  if (s.word_cnt == 10) {
    s.word_cnt = 0;
    s.word_buf = 0;
  }
}

// gps_nav_data_analyse_new_code on the channel's own slot state; returns flag bits 1 / 2 (a bit was completed / its value)
// and 5 / 6 (the bit edge inside the 20 ms grid was located this millisecond / it was edge 2, not 1)
__device__ __forceinline__ u32 nav_bit_sync(gpsx_loop_state_t &s, Quad16 &sip, int index, int IP, u32 now, bool word_sync)
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
      if (word_sync)
        nav_word_sync(s, nav_bit, now);   // (this millisecond's own vote was formed with the polarity of before, as in the reference)
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

__device__ __forceinline__ void loop_snr(gpsx_loop_state_t &s, int IP, int QP)
{
  s.i_part_summ += (u32)iabs(IP);
  s.q_part_summ += (u32)iabs(QP);
  s.snr_summ_cnt++;
  if (s.snr_summ_cnt > kSnrLength) {
    if (s.q_part_summ == 0) {
      s.snr_value = 1.0f;
      s.snr_i_latch = 0;
      s.snr_q_latch = 0;
      return;   // sic: the sums are not cleared on this path (tracking.c:152-156)
    }
    const float ratio = (float)s.i_part_summ / (float)s.q_part_summ;
    s.snr_value = s.i_part_summ ? 10.0f * gpsx_libm::log10f_near(ratio) : 10.0f * log10f(ratio);   // (log10f(0) = -inf)
    s.snr_i_latch = s.i_part_summ;   // what the estimate was made of: gpsx_loop_state_to_channel takes the host's logarithm
    s.snr_q_latch = s.q_part_summ;
    s.snr_summ_cnt = 0;
    s.i_part_summ = 0;
    s.q_part_summ = 0;
  }
}

}  // namespace

// Which channels a wave holds.  GPSX_SCHED_EVERY_MS: cpw consecutive channels.  GPSX_SCHED_MUX17: the reference's receiver is
// four channels sharing one correlator in a 17 ms cycle (PM/main.c:139-152) -- channel c is slot (c & 3) of receiver c >> 2
// and is served on the ticks t with (t % 17) / 4 == slot, t % 17 != 16.  A wave then holds channels of ONE slot (wave w of the
// workgroup: slot w of the workgroup's cpw receivers), so that "is this channel served this millisecond" is wave-uniform: three
// of a workgroup's four waves skip a millisecond's correlators whole, all four skip the idle slot.
template <bool MUX>
__global__ __launch_bounds__(256) void k_track_loop(const uint8_t *__restrict__ if_blocks, u32 block_stride, int n_blocks,
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
  int n_here, ch_l;                  // channels of this wave (0: an idle wave of the last workgroup still stages blocks)
  int slot = wave;                   // MUX: the slot of the receiver's 17 ms cycle this wave's channels are served in
  if (ch_map) {                      // the second pass of GPSX_DRAWS_LIBC: one listed channel per wave
    const int v = (int)blockIdx.x * 4 + wave;
    n_here = v < n_map ? 1 : 0;
    ch_l = n_here ? ch_map[v] : 0;
    slot = ch_l & 3;
  } else if (MUX) {
    const int first = (int)blockIdx.x * 4 * cpw + wave;   // the wave's channels: first, first + 4, ...
    n_here = first < n_ch ? min(cpw, (n_ch - first + 3) >> 2) : 0;
    ch_l = first + 4 * (c_l < n_here ? c_l : 0);
  } else {
    const int ch0 = ((int)blockIdx.x * 4 + wave) * cpw;
    n_here = ch0 < n_ch ? min(cpw, n_ch - ch0) : 0;
    ch_l = ch0 + (c_l < n_here ? c_l : 0);
  }
  const bool in_wave = c_l < n_here;
  const bool mine = in_wave && k_l < 3;
  gpsx_loop_state_t s = {};
  int prn = 0;
  if (n_here) {
    s = st[ch_l < n_ch ? ch_l : 0];
    prn = track_prn(s.prn, bad_prn, mine && k_l == 0);
  }
  Quad16 chk = quad16_load(s.pll_check_buf), sip = quad16_load(s.slot_ip);
  bool stalled = false;

#pragma unroll 1
  for (int ms = 0; ms < n_blocks; ms++) {
    u32 *sx = s_x[ms & 1];
    trkwave::stage_block(if_blocks + (size_t)ms * block_stride, if_format, sx, s_carrier);
    __syncthreads();
    if (!n_here)
      continue;
    const u32 now = first_tick + (u32)ms;
    int index = (int)(now & 3u);
    if (MUX) {
      const u32 big = now % (u32)(kSlotMs * kSlots + 1);   // PM/main.c:139-152
      if (big == (u32)(kSlotMs * kSlots) || (int)(big / (u32)kSlotMs) != slot) {
        // not this wave's slot (or the cycle's idle millisecond): nothing of the channel moves
        if (in_wave && k_l == 0) {
          flags[(size_t)ms * n_ch + ch_l] = 0;
          if (trace) {
            gpsx_loop_trace_t t = {};
            t.code_phase_fine = s.code_phase_fine;
            t.if_freq_offset_hz = s.if_freq_offset_hz;
            t.if_freq_accum = s.if_freq_accum;
            trace[(size_t)ms * n_ch + ch_l] = t;
          }
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
    const int fine = (int)(int16_t)(int)s.code_phase_fine;
    const u32 step = nco_step_per_word((float)if_hz + s.if_freq_offset_hz);
    const u32 iq = trkwave::wave_epl(sx, s_carrier, lane, n_here, mine, prn, fine, step, s.if_freq_accum, chipbits_all, rep_all);
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
      if (!events) {
        loop_false_lock_jump(s);
        moved = true;
      } else {
        // GPSX_DRAWS_LIBC: the draw is the host's (libc's rand(), in the reference's order).  First pass: report and stop
        // advancing this channel -- its state in HBM stays the launch's input; second pass: the host's candidate for this
        // millisecond is in the table.
        const gpsx_loop_reseed_t r = reseeds[ch_l];
        if (r.ms == ms) {
          s.reseed_count++;
          s.if_freq_offset_hz = (float)(int16_t)r.candidate;
          moved = true;
        } else if (!stalled) {
          stalled = true;
          if (in_wave && k_l == 0) {
            const u32 e = atomicAdd(n_events, 1u);
            events[e] = gpsx_loop_event_t{ch_l, ms, (int)(int16_t)s.if_freq_offset_hz, (int)s.found_freq_offset_hz};
          }
        }
      }
    }
    loop_fll(s, index, IP, QP);
    u32 flag = nav_bit_sync(s, sip, index, IP, now, word_sync != 0);
    loop_snr(s, IP, QP);
    flag |= (IP > 0 ? 1u : 0u) | (s.period_sync_ok_flag ? 8u : 0u) | (moved ? 16u : 0u) | 128u;
    if (in_wave && k_l == 0) {
      flags[(size_t)ms * n_ch + ch_l] = (uint8_t)flag;
      if (trace) {
        gpsx_loop_trace_t t;
        t.iq[0] = (int16_t)IE; t.iq[1] = (int16_t)QE; t.iq[2] = (int16_t)IP; t.iq[3] = (int16_t)QP; t.iq[4] = (int16_t)IL; t.iq[5] = (int16_t)QL;
        t.code_phase_fine = s.code_phase_fine;
        t.if_freq_offset_hz = s.if_freq_offset_hz;
        t.if_freq_accum = s.if_freq_accum;
        trace[(size_t)ms * n_ch + ch_l] = t;
      }
    }
  }
  if (in_wave && k_l == 0 && !stalled) {
    quad16_store(chk, s.pll_check_buf);
    quad16_store(sip, s.slot_ip);
    st[ch_l] = s;
  }
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

void launch_track_loop(hipStream_t s, const uint8_t *d_if_blocks, uint32_t block_stride, int n_blocks, int if_format, int if_hz,
                       gpsx_loop_state_t *d_st, int n_ch, uint32_t first_tick, int schedule, int word_sync,
                       const uint32_t *d_chipbits, const uint32_t *d_trk_rep, uint8_t *d_flags, gpsx_loop_trace_t *d_trace,
                       uint32_t *d_bad_prn, const int *d_ch_map, int n_map, const gpsx_loop_reseed_t *d_reseeds,
                       gpsx_loop_event_t *d_events, uint32_t *d_n_events)
{
  if (n_ch <= 0 || n_blocks <= 0 || (d_ch_map && n_map <= 0))
    return;
  int cpw = n_ch / (4 * 256 * 4);   // as launch_track_epl: ~4 workgroups per CU, 16 channels per wave at most
  cpw = cpw < 1 ? 1 : (cpw > 16 ? 16 : cpw);
  const dim3 grid(d_ch_map ? (n_map + 3) / 4 : (n_ch + 4 * cpw - 1) / (4 * cpw));
  if (schedule == GPSX_SCHED_MUX17)
    hipLaunchKernelGGL(k_track_loop<true>, grid, dim3(256), 0, s, d_if_blocks, block_stride, n_blocks, if_format, if_hz, d_st, n_ch,
                       cpw, first_tick, d_chipbits, d_trk_rep, d_flags, d_trace, d_bad_prn, word_sync, d_ch_map, n_map, d_reseeds,
                       d_events, d_n_events);
  else
    hipLaunchKernelGGL(k_track_loop<false>, grid, dim3(256), 0, s, d_if_blocks, block_stride, n_blocks, if_format, if_hz, d_st, n_ch,
                       cpw, first_tick, d_chipbits, d_trk_rep, d_flags, d_trace, d_bad_prn, word_sync, d_ch_map, n_map, d_reseeds,
                       d_events, d_n_events);
}

}  // namespace gpsx
