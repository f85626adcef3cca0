// gpsx_compat.cpp -- the reference's per-call correlator interface (include/gpsx_compat.h) on top of the engine.
// Host marshalling only: every function forwards to a gpsx_* entry point that runs HIP kernels.
#include "../../include/gpsx_compat.h"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/gpsx.h"

// layout parity with the reference header on LP64 (probed with the reference's gps_misc.h)
#if defined(__x86_64__) && defined(__LP64__)
static_assert(sizeof(gps_acq_t) == 60, "gps_acq_t layout");
static_assert(offsetof(gps_acq_t, state) == 16 && offsetof(gps_acq_t, code_phase_histogram) == 20, "gps_acq_t layout");
static_assert(sizeof(gps_tracking_t) == 152, "gps_tracking_t layout");
static_assert(offsetof(gps_tracking_t, if_freq_offset_hz) == 4 && offsetof(gps_tracking_t, if_freq_accum) == 8,
              "gps_tracking_t layout");
static_assert(offsetof(gps_tracking_t, code_phase_fine) == 80 && offsetof(gps_tracking_t, state) == 148,
              "gps_tracking_t layout");
static_assert(offsetof(gps_ch_t, tracking_data) == 60 && offsetof(gps_ch_t, nav_data) == 212, "gps_ch_t layout");
static_assert(sizeof(gps_nav_data_t) == 112 && offsetof(gps_nav_data_t, inv_polarity_flag) == 13 &&
                  offsetof(gps_nav_data_t, word_buf) == 16 && offsetof(gps_nav_data_t, word_cnt) == 46 &&
                  offsetof(gps_nav_data_t, word_detection_timestamp) == 52 && offsetof(gps_nav_data_t, subframe_cnt) == 68 &&
                  offsetof(gps_nav_data_t, subframe_data) == 71,
              "gps_nav_data_t layout");
static_assert(sizeof(gtime_t) == 16 && sizeof(eph_t) == 272 && offsetof(eph_t, toe) == 32 && offsetof(eph_t, A) == 80 &&
                  offsetof(eph_t, crc) == 152 && offsetof(eph_t, toes) == 200 && offsetof(eph_t, f0) == 216 &&
                  offsetof(eph_t, tgd) == 240,
              "eph_t layout");
static_assert(sizeof(sdreph_t) == 320 && offsetof(sdreph_t, tow_gpst) == 280 && offsetof(sdreph_t, cnt) == 292 &&
                  offsetof(sdreph_t, sub_cnt) == 312 && offsetof(sdreph_t, received_mask) == 314 &&
                  sizeof(gps_obs_data_t) == 16,
              "sdreph_t layout");
static_assert(offsetof(gps_ch_t, obs_data) == 328 && offsetof(gps_ch_t, eph_data) == 344, "gps_ch_t layout");
static_assert(offsetof(gps_ch_t, prn) == 664 && offsetof(gps_ch_t, prn_code) == 665 && sizeof(gps_ch_t) == 1688,
              "gps_ch_t layout");
#endif

extern "C" {
uint16_t tmp_prn_data[GPS_DATA_WORDS_CNT];
uint16_t tmp_data_i[GPS_DATA_WORDS_CNT];
uint16_t tmp_data_q[GPS_DATA_WORDS_CNT];
}

#include "gpsx_compat_internal.hpp"

namespace {

gpsx_ctx *g_ctx = nullptr;

}  // namespace

[[noreturn]] void gpsx_compat_die(const char *what, int rc)
{
  std::fprintf(stderr, "libgpsx (compat): %s failed: %s (%s). The correlator runs on the GPU only; there is no CPU path.\n",
               what, gpsx_strerror(rc), g_ctx ? gpsx_last_error(g_ctx) : "no context");
  std::abort();
}

gpsx_ctx *gpsx_compat_ctx()
{
  if (!g_ctx) {
    const char *dev = std::getenv("GPSX_DEVICE");
    const int rc = gpsx_create(&g_ctx, dev ? std::atoi(dev) : 0, nullptr);
    if (rc != GPSX_OK)
      gpsx_compat_die("gpsx_create", rc);
  }
  return g_ctx;
}

namespace {
inline gpsx_ctx *ctx() { return gpsx_compat_ctx(); }
[[noreturn]] inline void die(const char *what, int rc) { gpsx_compat_die(what, rc); }
}  // namespace

extern "C" {

void gpsx_compat_shutdown(void)
{
  gpsx_compat_capture_forget();   // the rings die with the context
  gpsx_destroy(g_ctx);
  g_ctx = nullptr;
}

void gps_fill_summ_table(void) { (void)ctx(); }

void gps_generate_prn(uint8_t *dest, int prn)
{
  if (prn < 1)
    return;
  const uint8_t p = (uint8_t)prn;
  const int rc = gpsx_ca_codes(ctx(), &p, 1, dest);
  if (rc != GPSX_OK)
    die("gps_generate_prn", rc);
}

void gps_channell_prepare(gps_ch_t *channel)
{
  if (channel->prn < 1)
    return;
  gps_generate_prn(channel->prn_code, channel->prn);
}

int16_t gps_correlation8(uint16_t *prn_p, uint16_t *data_i, uint16_t *data_q, uint16_t offset)
{
  int16_t corr = 0;
  const int rc = gpsx_corr_offsets(ctx(), prn_p, data_i, data_q, &offset, 1, nullptr, nullptr, &corr);
  if (rc != GPSX_OK)
    die("gps_correlation8", rc);
  return corr;
}

void gps_correlation_iq(uint16_t *prn_p, uint16_t *data_i, uint16_t *data_q, uint16_t offset, int16_t *res_i,
                        int16_t *res_q)
{
  uint16_t ci = 0, cq = 0;
  const int rc = gpsx_corr_offsets(ctx(), prn_p, data_i, data_q, &offset, 1, &ci, &cq, nullptr);
  if (rc != GPSX_OK)
    die("gps_correlation_iq", rc);
  *res_i = (int16_t)((int)ci - BITS_IN_PRN / 2);
  *res_q = (int16_t)((int)cq - BITS_IN_PRN / 2);
}

uint16_t correlation_search(uint16_t *prn_p, uint16_t *data_i, uint16_t *data_q, uint16_t start_shift,
                            uint16_t stop_shift, uint16_t *aver_val, uint16_t *phase)
{
  gpsx_peak_t pk;
  const int rc = gpsx_corr_search(ctx(), prn_p, data_i, data_q, start_shift, stop_shift, &pk);
  if (rc != GPSX_OK)
    die("correlation_search", rc);
  *aver_val = (uint16_t)pk.avr;
  *phase = (uint16_t)pk.phase;
  return (uint16_t)pk.max_val;
}

void gps_shift_to_zero_freq(uint8_t *signal_data, uint8_t *data_i, uint8_t *data_q, float freq_hz)
{
  uint32_t accum = 0;
  const int rc = gpsx_wipeoff(ctx(), signal_data, freq_hz, &accum, data_i, data_q);
  if (rc != GPSX_OK)
    die("gps_shift_to_zero_freq", rc);
}

void gps_shift_to_zero_freq_track(gps_tracking_t *trk_channel, uint8_t *signal_data, uint8_t *data_i, uint8_t *data_q)
{
  const float freq_hz = (float)IF_FREQ_HZ + trk_channel->if_freq_offset_hz;
  const int rc = gpsx_wipeoff(ctx(), signal_data, freq_hz, &trk_channel->if_freq_accum, data_i, data_q);
  if (rc != GPSX_OK)
    die("gps_shift_to_zero_freq_track", rc);
}

void gps_generate_prn_data2(gps_ch_t *channel, uint16_t *data, uint16_t offset_bits)
{
  const int rc = gpsx_replica(ctx(), channel->prn_code, offset_bits, data);
  if (rc != GPSX_OK)
    die("gps_generate_prn_data2", rc);
}

void gps_rewind_if_phase(gps_tracking_t *trk_channel, uint8_t steps)
{
  gpsx_trk_state_t st;
  st.prn = 1;
  st.code_phase_fine = 0.0f;
  st.if_freq_offset_hz = trk_channel->if_freq_offset_hz;
  st.if_freq_accum = trk_channel->if_freq_accum;
  const int rc = gpsx_rewind(ctx(), &st, 1, &steps);
  if (rc != GPSX_OK)
    die("gps_rewind_if_phase", rc);
  trk_channel->if_freq_accum = st.if_freq_accum;
}

}  // extern "C"


// ---- channel record <-> device loop state (include/gpsx.h gpsx_loop_state_t) ------------------------------------------------
static_assert(sizeof(gpsx_loop_state_t) == 120 && offsetof(gpsx_loop_state_t, pll_check_buf) == 32 &&
                  offsetof(gpsx_loop_state_t, slot_ip) == 80 && sizeof(gpsx_loop_trace_t) == 24,
              "gpsx_loop_state_t layout");

extern "C" {

// A tracking channel (state GPS_TRACKING_RUN) handed to the device loop.  The 4 ms group state of the bit synchroniser is
// not part of gps_ch_t (the reference keeps it in file statics): hand channels over on a millisecond whose tick & 3 == 0,
// where a group starts.  rng_seed: any non-zero value, e.g. the channel number + 1.
void gpsx_loop_state_from_channel(const gps_ch_t *ch, uint32_t rng_seed, gpsx_loop_state_t *out)
{
  const gps_tracking_t &t = ch->tracking_data;
  const gps_nav_data_t &n = ch->nav_data;
  gpsx_loop_state_t s;
  std::memset(&s, 0, sizeof s);
  s.prn = ch->prn;
  s.code_phase_fine = t.code_phase_fine;
  s.if_freq_offset_hz = t.if_freq_offset_hz;
  s.if_freq_accum = t.if_freq_accum;
  s.dll_code_err = t.dll_code_err;
  s.pll_code_err = t.pll_code_err;
  s.fll_err = t.fll_err;
  s.fll_old_i = t.fll_old_i;
  s.fll_old_q = t.fll_old_q;
  std::memcpy(s.pll_check_buf, t.pll_check_buf, sizeof s.pll_check_buf);
  s.pll_bad_state_master_cnt = t.pll_bad_state_master_cnt;
  s.pll_bad_state_cnt = t.pll_bad_state_cnt;
  s.period_sync_ok_flag = n.period_sync_ok_flag;
  s.found_freq_offset_hz = ch->acq_data.found_freq_offset_hz;
  s.rng = rng_seed ? rng_seed : 1u;
  s.i_part_summ = t.i_part_summ;
  s.q_part_summ = t.q_part_summ;
  s.snr_value = t.snr_value;
  s.snr_summ_cnt = t.snr_summ_cnt;
  s.code_filt_cnt = t.code_filt_cnt;
  s.code_phase_fine_filt = t.code_phase_fine_filt;
  s.old_swap_time = n.old_swap_time;
  s.right_period_cnt = n.right_period_cnt;
  s.old_reminder = n.old_reminder;
  s.accurate_swap_time = n.accurate_swap_time;
  s.accurate_swap_ok = n.accurate_swap_ok;
  s.last_bit_pos_cnt = n.last_bit_pos_cnt;
  s.last_bit_neg_cnt = n.last_bit_neg_cnt;
  s.inv_polarity_flag = n.inv_polarity_flag;
  s.prev_track_timestamp = t.prev_track_timestamp;
  // the word layer's polarity-deciding part (the device's word sync continues where the host's word layer stands)
  for (int i = 0; i < GPS_NAV_WORD_LENGTH; i++)
    s.word_buf |= (uint32_t)(n.word_buf[i] & 1u) << i;
  s.word_detection_timestamp = n.word_detection_timestamp;
  s.word_cnt = n.word_cnt;
  s.word_bit_cnt = n.word_bit_cnt;
  s.inv_preabmle_cnt = n.inv_preabmle_cnt;
  s.word_flags = (uint8_t)((n.old_D29 & 1u) | ((n.old_D30 & 1u) << 1) | (n.polarity_found ? 4u : 0u));
  *out = s;
}

// ... and back: everything the loops own is written into the record, the rest of it (acquisition result, word layer,
// observations, ephemeris) is left alone -- the word layer's fields are the HOST's word layer's (gps_tracking_words_batch),
// which runs on the same bits as the device's word sync; inv_polarity_flag, which both decide identically, is written.
void gpsx_loop_state_to_channel(const gpsx_loop_state_t *in, gps_ch_t *ch)
{
  gps_tracking_t &t = ch->tracking_data;
  gps_nav_data_t &n = ch->nav_data;
  const gpsx_loop_state_t &s = *in;
  t.code_phase_fine = s.code_phase_fine;
  t.if_freq_offset_hz = s.if_freq_offset_hz;
  t.if_freq_accum = s.if_freq_accum;
  t.dll_code_err = s.dll_code_err;
  t.pll_code_err = s.pll_code_err;
  t.fll_err = s.fll_err;
  t.fll_old_i = s.fll_old_i;
  t.fll_old_q = s.fll_old_q;
  std::memcpy(t.pll_check_buf, s.pll_check_buf, sizeof s.pll_check_buf);
  t.pll_bad_state_master_cnt = s.pll_bad_state_master_cnt;
  t.pll_bad_state_cnt = s.pll_bad_state_cnt;
  t.i_part_summ = s.i_part_summ;
  t.q_part_summ = s.q_part_summ;
  // snr_value: the device's logarithm agrees with the C library's on 99.85 % of the ratios (csrc/gpsx_libm.hpp); the record gets
  // the C library's, from the sums the device latched when it made the estimate (tracking.c:158-162)
  t.snr_value = s.snr_q_latch ? 10.0f * log10f((float)s.snr_i_latch / (float)s.snr_q_latch) : s.snr_value;
  t.snr_summ_cnt = s.snr_summ_cnt;
  t.code_filt_cnt = s.code_filt_cnt;
  t.code_phase_fine_filt = s.code_phase_fine_filt;
  n.period_sync_ok_flag = s.period_sync_ok_flag;
  n.old_swap_time = s.old_swap_time;
  n.right_period_cnt = s.right_period_cnt;
  n.old_reminder = s.old_reminder;
  n.accurate_swap_time = s.accurate_swap_time;
  n.accurate_swap_ok = s.accurate_swap_ok;
  n.last_bit_pos_cnt = s.last_bit_pos_cnt;
  n.last_bit_neg_cnt = s.last_bit_neg_cnt;
  n.inv_polarity_flag = s.inv_polarity_flag;
  // the tick the device last served the channel on: a host that takes the channel back into gps_tracking_process sees the
  // elapsed time the reference would (tracking.c:102-113), after any number of device milliseconds
  t.prev_track_timestamp = s.prev_track_timestamp;
}

}  // extern "C"
