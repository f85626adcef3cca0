// gpsx_nav_master.cpp -- the pseudorange step: subframe epochs + filtered code phases -> pseudoranges and reception times,
// then the position solution's scheduling (PM/GPS/gps_master.c:159-430: gps_master_nav_handling,
// gps_master_final_pseudorange_calc, gps_master_filter_code_phase, gps_master_code_phase_filter_reset,
// gps_master_calculate_pos).  Host code, no GPU: a few dozen operations per channel every 17 ms.
//
// PARITY UNPINNED.  Every other restated layer of this library is compared with the reference's own object code; this one
// cannot be: gps_master.c includes the MCU's peripheral headers, whose chain ends at Firmware/Libraries/CMSIS/stm32f4xx.h:817
// `#include "core_cm4.h"` -- a header that is not in the reference tree -- so the file cannot be compiled in place without
// a stand-in.  What is written here follows the source text; what it is tested on is physics: tests/test_gpu_pvt_chain.py
// synthesises an IF stream from satellites on broadcast orbits around a chosen receiver position, runs it through
// tracking -> nav words -> ephemeris -> THIS step -> the (pinned) solver, and asks for the position back.
//
// The arithmetic of one pseudorange, as the reference forms it:
//   * every channel stamps the millisecond tick at which its last subframe ended (gpsx_steps.cpp stamp_subframe);
//   * the channel whose stamp is the earliest is the reference satellite: its signal is *declared* to have travelled
//     68.802 ms, the others that plus the difference of the stamps (whole milliseconds) plus the difference of the code
//     phases (fractions of one, averaged over the filter window) -- the common error lands in the receiver clock term;
//   * a code phase that wrapped through 0 / 16368 since the last subframe moves the whole-millisecond count by one,
//     in the direction the Doppler's sign says;
//   * the reception time is the reference satellite's hand-over word plus the ticks since its subframe ended, with 4 ms per
//     channel index added because the 17 ms multiplex serves the channels one after the other.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/gpsx_compat.h"
#include "gpsx_compat_internal.hpp"

namespace {

constexpr double kOffsetTimeMs = 68.802;                       // GPS_OFFSET_TIME_MS
constexpr double kMetresPerMs = 299792458.0 / PRN_SPEED_HZ;    // CLIGHT_NORM
constexpr uint32_t kSubframeMs = 6000;                         // SUBFRAME_DURATION_MS
constexpr uint32_t kCalcPosPeriodMs = 500;                     // GPS_CALC_POS_PERIOD_MS
constexpr uint16_t kCodeFilterLength = 100;                    // CODE_FILTER_LENGTH (PM/config.h:38)
constexpr uint32_t kEpochSpreadMs = 100;                       // subframes of one epoch arrive within this

uint32_t g_prev_calc_ms = 0;   // gps_master_calculate_pos's static

// The step works on a VIEW of the channel table: the channels at index[0..n) (the reference: its four channels in order),
// position k of the view being served k * slot_ms milliseconds into the cycle (the reference's multiplex: 4).
struct View {
  gps_ch_t *ch;
  const int *index;   // nullptr: the identity
  int n;
  uint32_t slot_ms;
  gps_ch_t &operator[](int k) const { return ch[index ? index[k] : k]; }
};

void open_filter_window(const View &v, uint32_t now)
{
  for (int i = 0; i < v.n; i++) {
    gps_tracking_t &t = v[i].tracking_data;
    t.code_phase_fine_filt = 0.0f;
    t.code_filt_cnt = 0;
    t.filt_start_time_ms = now;
  }
}

// The window average of every channel's code phase, in place; 0 when the window is not usable (too few points somewhere,
// a wrap inside it, or a window older than a second -- the last two restart it), else the window's length in ms.
uint16_t close_filter_window(const View &v, uint32_t now)
{
  for (int i = 0; i < v.n; i++)
    if (!(v[i].tracking_data.code_filt_cnt > kCodeFilterLength))
      return 0;
  bool wrapped = false;
  for (int i = 0; i < v.n; i++)
    wrapped |= v[i].tracking_data.code_phase_fine_filt < -0.5f;   // the DLL marks a wrap with -1
  const uint32_t length = now - v[0].tracking_data.filt_start_time_ms;   // (all windows are opened together)
  if (wrapped || length > 1000) {
    open_filter_window(v, now);
    return 0;
  }
  for (int i = 0; i < v.n; i++) {
    gps_tracking_t &t = v[i].tracking_data;
    t.code_phase_fine_filt = t.code_phase_fine_filt / t.code_filt_cnt;
  }
  return (uint16_t)length;
}

void pseudoranges(const View &v, uint32_t since_ref_subframe_ms, uint32_t ref_epoch_ms, int ref)
{
  for (int i = 0; i < v.n; i++) {
    const gps_tracking_t &t = v[i].tracking_data;
    const int32_t whole_ms = (int32_t)(v[i].nav_data.last_subframe_time - ref_epoch_ms);
    double travel_ms = (double)whole_ms + t.code_phase_fine_filt / ((double)PRN_LENGTH * 16.0f);
    if (t.code_phase_swap_flag == 1)   // wrapped since the subframe stamp: the stamp is one code period off
      travel_ms = travel_ms - (t.if_freq_offset_hz < 0.0f ? -1.0 : 1.0);
    v[i].obs_data.pseudorange_m = (kOffsetTimeMs + travel_ms) * kMetresPerMs;
    const float since_s = (float)(since_ref_subframe_ms + (uint32_t)i * v.slot_ms) / (float)PRN_SPEED_HZ;
    v[i].obs_data.tow_s = v[ref].eph_data.tow_gpst + since_s;
  }
}

// The epoch bookkeeping in front of the pseudoranges.  Returns false when there is nothing to compute yet.
bool nav_epoch(const View &v, int &ref, uint32_t &ref_epoch_ms)
{
  int stamped = 0, unlocked = 0;
  uint32_t earliest = 0xFFFFFFFFu, latest = 0;
  uint16_t most_subframes = 0;
  ref = 0;
  for (int i = 0; i < v.n; i++) {
    const gps_nav_data_t &nd = v[i].nav_data;
    stamped += nd.last_subframe_time != 0;
    unlocked += nd.first_subframe_time == 0;
    if (nd.last_subframe_time < earliest) {   // strict: the first of equals is the reference satellite
      earliest = nd.last_subframe_time;
      ref = i;                                // (the reference's uint8_t index: its table has four entries)
    }
    if (nd.last_subframe_time > latest)
      latest = nd.last_subframe_time;
    if (nd.subframe_cnt > most_subframes)
      most_subframes = nd.subframe_cnt;
  }
  if (earliest == 0)
    return false;
  if (latest - earliest > kEpochSpreadMs)   // some channels already have the next subframe: wait for the others
    return false;
  if (stamped == v.n && unlocked == v.n)    // once: the zero moment, from which subframes are counted
    for (int i = 0; i < v.n; i++) {
      v[i].nav_data.first_subframe_time = v[i].nav_data.last_subframe_time;
      v[i].nav_data.subframe_cnt = 0;
    }
  if (v[0].nav_data.first_subframe_time == 0)
    return false;
  // (the count taken BEFORE the zeroing above, as the reference's locals hold it)
  ref_epoch_ms = v[ref].nav_data.first_subframe_time + (uint32_t)most_subframes * kSubframeMs;
  return true;
}

void watch_code_wraps(const View &v)
{
  for (int i = 0; i < v.n; i++) {
    gps_tracking_t &t = v[i].tracking_data;
    if (t.code_phase_swap_flag && v[i].nav_data.new_subframe_flag) {   // a fresh stamp absorbs the wrap
      v[i].nav_data.new_subframe_flag = 0;
      t.code_phase_swap_flag = 0;
    }
    const float jump = (float)std::fabs(t.old_code_phase_fine - t.code_phase_fine);
    if (jump > (float)PRN_LENGTH * 16.0f / 2.0f)
      t.code_phase_swap_flag = 1;
    t.old_code_phase_fine = t.code_phase_fine;
  }
}

// 1 = pseudoranges and reception times were renewed
int nav_step(const View &v, uint32_t now)
{
  int ref;
  uint32_t ref_epoch_ms;
  if (!nav_epoch(v, ref, ref_epoch_ms))
    return -1;
  watch_code_wraps(v);
  int32_t since = (int32_t)now - (int32_t)v[ref].nav_data.last_subframe_time;
  if (since < 0)
    since = since % (int32_t)kSubframeMs;
  const uint16_t window = close_filter_window(v, now);
  since = since - window / 2;               // the averaged code phase belongs to the middle of its window
  if (window < 1)
    return 0;
  pseudoranges(v, (uint32_t)since, ref_epoch_ms, ref);
  open_filter_window(v, now);
  return 1;
}

View reference_table(gps_ch_t *channels) { return View{channels, nullptr, GPS_SAT_CNT, TRACKING_CH_LENGTH}; }

}  // namespace

extern "C" {

obsd_t obsd[GPS_SAT_CNT];

}

void gpsx_nav_master_reset()
{
  g_prev_calc_ms = 0;
  std::memset(obsd, 0, sizeof obsd);
}

extern "C" {

void gps_master_final_pseudorange_calc(gps_ch_t *channels, uint32_t curr_tick_time, uint32_t ref_time_diff_ms,
                                       uint32_t ref_time_ms, uint8_t ref_idx)
{
  (void)curr_tick_time;
  pseudoranges(reference_table(channels), ref_time_diff_ms, ref_time_ms, ref_idx);
}

uint16_t gps_master_filter_code_phase(gps_ch_t *channels, uint32_t curr_tick_time)
{
  return close_filter_window(reference_table(channels), curr_tick_time);
}

void gps_master_code_phase_filter_reset(gps_ch_t *channels, uint32_t curr_tick_time)
{
  open_filter_window(reference_table(channels), curr_tick_time);
}

// "Call until not busy" around the solver, twice a second, once every channel holds subframes 1-3.
void gps_master_calculate_pos(gps_ch_t *channels)
{
  if (solving_is_busy()) {
    gps_pos_solve(obsd);
    return;
  }
  const uint32_t now = signal_capture_get_packet_cnt();
  if (now - g_prev_calc_ms > kCalcPosPeriodMs) {
    g_prev_calc_ms = now;
    int complete = 0;
    for (int i = 0; i < GPS_SAT_CNT; i++)
      complete += (channels[i].eph_data.received_mask_proc & 0x7) == 0x7;
    sdrobs2obsd(channels, GPS_SAT_CNT, obsd);
    if (complete == GPS_SAT_CNT)
      gps_pos_solve(obsd);
  }
}

// The idle slot of the 17 ms cycle (gps_master_handling, index 0xFF).  Weak, as before: a host may still bring its own.
__attribute__((weak)) void gps_master_nav_handling(gps_ch_t *channels)
{
  if (nav_step(reference_table(channels), signal_capture_get_packet_cnt()) < 0)
    return;
  gps_master_calculate_pos(channels);
}

// Not in the reference: the same step for ONE receiver of any number of channels (every loop above runs over GPS_SAT_CNT = 4
// in the reference; nothing in the arithmetic depends on the 4), served in the reference's multiplex: channel i of the table
// 4 i ms into the cycle.  Returns 1 when pseudoranges and reception times were renewed, 0 when the step ran but the filter
// window was not ready, -1 when the subframe epochs are not there yet.
int gpsx_nav_pseudoranges(gps_ch_t *channels, int n_ch, uint32_t now_ms)
{
  if (!channels || n_ch <= 0)
    return -1;
  return nav_step(View{channels, nullptr, n_ch, TRACKING_CH_LENGTH}, now_ms);
}

// The step for a receiver that is a SUBSET of a large channel table (device-tracked tables hold thousands of channels; one
// receiver's epoch, averaging window and wrap bookkeeping must not wait for, or be restarted by, channels of another):
// the channels index[0 .. n), position k of the list served k * slot_ms milliseconds into the cycle (4: the reference's
// multiplex, GPSX_SCHED_MUX17 with the list in slot order; 0: all served on the same millisecond, GPSX_SCHED_EVERY_MS).
// Same return values.
int gpsx_nav_pseudoranges_subset(gps_ch_t *channels, const int *index, int n, uint32_t now_ms, uint32_t slot_ms)
{
  if (!channels || !index || n <= 0)
    return -1;
  for (int k = 0; k < n; k++)
    if (index[k] < 0)
      return -1;
  return nav_step(View{channels, index, n, slot_ms}, now_ms);
}

}  // extern "C"
