/* examples/gpsx_demo.c -- a plain C host driving the MI355X correlator engine through the reference's own interface.
 *
 * This is the shape of the firmware's main loop (Firmware/project_main/main.c:45-168 of iliasam/STM32F4_SDR_GPS) on
 * a hosted system: the 4-satellite channel table with PRN + Doppler-hint inputs, acquisition steps on captured
 * milliseconds until every channel is acquired, then 17-slot multiplexed tracking steps -- with libgpsx.so in place of
 * gps_misc.c / acquisition.c / tracking.c.  The SPI/DMA capture driver is fed from a raw IF file (1 bit per sample,
 * LSB first, 2046 bytes per millisecond: the format PC_SpiLight replays) through the firmware's own signal_capture_*
 * interface; gps_master_handling() / gps_master_need_acq() (gps_master.h) sequence the channels as in the firmware.
 *
 *   gcc -O2 -I include examples/gpsx_demo.c -L stm32f4_sdr_gps_amd/lib -lgpsx \
 *       -Wl,-rpath,$PWD/stm32f4_sdr_gps_amd/lib -lm -o gpsx_demo
 *   ./gpsx_demo capture.bin [max_ms] [--device-loops]
 *
 * --device-loops: from the first cycle start on which all four channels track, the tracking steps run ON THE GPU in the firmware's
 * own 17 ms multiplex (include/gpsx.h gpsx_track_loop under GPSX_SCHED_MUX17: one launch per cycle, channel state resident in
 * GPU memory); the host keeps what the firmware does in its idle slot and per navigation bit: the word layer on the launch's
 * flag bytes (gps_tracking_words_batch), the loop state back into the channel records, gps_master_handling(.., 0xFF).  The
 * records follow the reference's byte for byte either way (tests/test_c_host_demo.py compares the end state).
 *
 * Prints one line per channel: acquisition result and the tracking loop state after the last millisecond.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpsx_compat.h"

#define BLOCK_BYTES (PRN_SPI_WORDS_CNT * 2)

static gps_ch_t gps_channels[GPS_SAT_CNT];

int main(int argc, char **argv)
{
  if (argc < 2) {
    fprintf(stderr, "usage: %s capture.bin [max_ms]\n", argv[0]);
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  long max_ms = 0x7fffffffL;
  int device_loops = 0;
  for (int a = 2; a < argc; a++) {
    if (strcmp(argv[a], "--device-loops") == 0)
      device_loops = 1;
    else
      max_ms = atol(argv[a]);
  }

  /* PM/main.c:54-73: the firmware's default table */
  static const uint8_t prn[GPS_SAT_CNT] = {5, 14, 20, 30};
  static const int16_t hint_hz[GPS_SAT_CNT] = {900, 4000, -1000, 2000};
  gps_fill_summ_table();                    /* opens the GPU context; aborts loudly if there is none */
  memset(gps_channels, 0, sizeof gps_channels);
  for (int i = 0; i < GPS_SAT_CNT; i++) {
    gps_channels[i].prn = prn[i];
    gps_channels[i].acq_data.given_freq_offset_hz = hint_hz[i];
    gps_channell_prepare(&gps_channels[i]);
  }

  /* The capture driver: each block read from the file is what one DMA half/full-transfer interrupt delivers
   * (PM/signal_capture.c:57-82) -- gpsx_compat_capture_push() moves the ready pointer, counts the 1 ms tick, raises
   * the flag and starts the block's copy to the GPU.  Unlike the MCU this host never drops a millisecond while it
   * is acquiring, so the "not realtime" branch takes every block too. */
  uint8_t block[BLOCK_BYTES];
  long t = 0, acquired_at = -1;
  signal_capture_init();
  gpsx_compat_set_packet_cnt(0);
  gps_master_handling(gps_channels, 0);                 /* boot: starts the first channel's search at tick 0 */
  gpsx_compat_set_packet_cnt(0xFFFFFFFFu);              /* so that the first block received gets tick 0 too */
  /* --device-loops */
  enum { CYCLE = TRACKING_CH_LENGTH * GPS_SAT_CNT + 1 };
  gpsx_ctx *gx = NULL;
  gpsx_loop_state_t st[GPS_SAT_CNT], *d_state = NULL;
  static uint8_t cycle_blocks[CYCLE * BLOCK_BYTES], flags[CYCLE * GPS_SAT_CNT];
  long handed_over_at = -1;
  for (; t < max_ms; t++) {
    if (device_loops && d_state == NULL && !gps_master_need_acq() && t % CYCLE == 0) {
      int tracking = 0;
      for (int i = 0; i < GPS_SAT_CNT; i++)
        tracking += gps_channels[i].tracking_data.state == GPS_TRACKING_RUN;
      if (tracking == GPS_SAT_CNT) {          /* a cycle starts and every channel tracks: the loops move to the device */
        /* this host strides d_state by ITS sizeof(gpsx_loop_state_t): refuse a library built from another layout */
        if (gpsx_abi_check(GPSX_VERSION, sizeof(gpsx_loop_state_t), sizeof(gpsx_acq_grid_t), sizeof(gpsx_peak_t)) != GPSX_OK) {
          fprintf(stderr, "libgpsx.so (version %d) does not match the header this host was built with (%d)\n", gpsx_version(), GPSX_VERSION);
          return 1;
        }
        if (gpsx_create(&gx, 0, NULL) != GPSX_OK || gpsx_malloc(gx, (void **)&d_state, sizeof st) != GPSX_OK ||
            gpsx_loop_set_schedule(gx, GPSX_SCHED_MUX17) != GPSX_OK)
          return 1;
        for (int i = 0; i < GPS_SAT_CNT; i++)
          gpsx_loop_state_from_channel(&gps_channels[i], (uint32_t)i + 1, &st[i]);
        gpsx_memcpy_h2d(gx, d_state, st, sizeof st);
        handed_over_at = t;
      }
    }
    if (d_state != NULL) {                    /* one 17 ms cycle per launch */
      long want = max_ms - t < CYCLE ? max_ms - t : CYCLE;
      const long got = (long)fread(cycle_blocks, BLOCK_BYTES, (size_t)want, f);
      if (got <= 0)
        break;
      if (gpsx_track_loop(gx, cycle_blocks, (int)got, d_state, GPS_SAT_CNT, (uint32_t)t, flags, NULL) != GPSX_OK) {
        fprintf(stderr, "gpsx_track_loop: %s\n", gpsx_last_error(gx));
        return 1;
      }
      gps_tracking_words_batch(gps_channels, GPS_SAT_CNT, flags, (int)got, (uint32_t)t, NULL, 0);   /* the word layer, per completed bit */
      gpsx_memcpy_d2h(gx, st, d_state, sizeof st);
      for (int i = 0; i < GPS_SAT_CNT; i++)
        gpsx_loop_state_to_channel(&st[i], &gps_channels[i]);
      if (got == CYCLE) {                     /* the cycle's idle millisecond: the firmware's navigation slot */
        const uint16_t window_before = gps_channels[0].tracking_data.code_filt_cnt;
        gpsx_compat_set_packet_cnt((uint32_t)(t + CYCLE - 1));
        gps_master_handling(gps_channels, 0xFF);
        if (gps_channels[0].tracking_data.code_filt_cnt != window_before)   /* the pseudorange step consumed the averaging window */
          gpsx_loop_reset_code_filter(gx, d_state, GPS_SAT_CNT);
      }
      t += got - 1;
      continue;
    }
    if (fread(block, 1, BLOCK_BYTES, f) != BLOCK_BYTES)
      break;
    gpsx_compat_capture_push(block);
    if (gps_master_need_acq()) {                          /* main_slow_data_proc, PM/main.c:106-125 */
      signal_capture_need_data_copy();
      signal_capture_handling();
      if (signal_capture_check_copied()) {
        acquisition_process(gps_channels, signal_capture_get_copy_buf());   /* main_process_acq_data, :163-168 */
        gps_master_handling(gps_channels, 0);
      }
      if (!gps_master_need_acq())
        acquired_at = t;
    } else if (signal_capture_have_irq()) {               /* main_fast_data_proc, PM/main.c:134-158 */
      uint8_t *signal_p = signal_capture_get_ready_buf();
      const uint32_t time_cnt = signal_capture_get_packet_cnt();
      const uint32_t index_big = time_cnt % (TRACKING_CH_LENGTH * GPS_SAT_CNT + 1);
      uint32_t sat = index_big / TRACKING_CH_LENGTH;
      if (sat >= GPS_SAT_CNT) sat = 0;
      const uint8_t index = index_big == TRACKING_CH_LENGTH * GPS_SAT_CNT ? 0xFF : (uint8_t)(index_big % TRACKING_CH_LENGTH);
      gps_tracking_process(&gps_channels[sat], signal_p, index);
      gps_master_handling(gps_channels, index);
    }
  }
  fclose(f);

  printf("processed_ms=%ld acquired_at_ms=%ld handed_to_the_device_at_ms=%ld\n", t, acquired_at, handed_over_at);
  for (int i = 0; i < GPS_SAT_CNT; i++) {
    const gps_ch_t *c = &gps_channels[i];
    uint32_t fine_bits, freq_bits;
    memcpy(&fine_bits, &c->tracking_data.code_phase_fine, 4);
    memcpy(&freq_bits, &c->tracking_data.if_freq_offset_hz, 4);
    printf("PRN=%u acq_state=%d code_phase=%u doppler_hz=%d trk_state=%d code_phase_fine=%.3f(0x%08x) "
           "if_freq_offset_hz=%.3f(0x%08x) nco=0x%08x snr_db=%.2f bit_sync=%u\n",
           c->prn, (int)c->acq_data.state, c->acq_data.found_code_phase, c->acq_data.found_freq_offset_hz,
           (int)c->tracking_data.state, c->tracking_data.code_phase_fine, fine_bits, c->tracking_data.if_freq_offset_hz,
           freq_bits, c->tracking_data.if_freq_accum, c->tracking_data.snr_value, c->nav_data.period_sync_ok_flag);
  }
  if (d_state) {
    gpsx_free(gx, d_state);
    gpsx_destroy(gx);
  }
  gpsx_compat_shutdown();
  return 0;
}
