/* examples/gpsx_track_loop.c -- a plain C host that tracks with the loops ON THE GPU (include/gpsx.h gpsx_track_loop).
 *
 * Channels start from an acquisition result (PRN : Doppler bin : code phase in bytes, as gps_master_handling hands them over),
 * run pre-tracking and the first milliseconds of tracking in the host mode (gps_tracking_process_batch: the reference's loops on
 * the CPU behind every correlator launch), are handed to the device at a 4 ms group boundary (gpsx_loop_state_from_channel)
 * and from there advance K = 20 ms per launch: correlators, DLL / PLL / FLL, false-lock check, SNR and bit synchroniser in one
 * kernel, channel state resident in GPU memory, ONE byte per channel and millisecond back, which the host's word layer consumes
 * (gps_tracking_words_batch).  At the end the states come back into the channel records (gpsx_loop_state_to_channel).
 *
 *   gcc -O2 -I include examples/gpsx_track_loop.c -L stm32f4_sdr_gps_amd/lib -lgpsx -Wl,-rpath,$PWD/stm32f4_sdr_gps_amd/lib -o gpsx_track_loop
 *   ./gpsx_track_loop capture.bin hand_over_ms prn:doppler_hz:code_phase [prn:doppler_hz:code_phase ...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpsx_compat.h"

#define BLOCK 2046
#define K 20
#define MAX_CH 64

int main(int argc, char **argv)
{
  if (argc < 4) {
    fprintf(stderr, "usage: %s capture.bin hand_over_ms prn:doppler_hz:code_phase ...\n", argv[0]);
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  const long hand = atol(argv[2]) / 4 * 4;   /* a 4 ms group boundary */
  const int n = argc - 3 > MAX_CH ? MAX_CH : argc - 3;
  static gps_ch_t ch[MAX_CH];
  memset(ch, 0, sizeof ch);
  gps_fill_summ_table();                      /* opens the GPU context; aborts loudly if there is none */
  for (int i = 0; i < n; i++) {
    int prn, hz, phase;
    if (sscanf(argv[3 + i], "%d:%d:%d", &prn, &hz, &phase) != 3)
      return 2;
    ch[i].prn = (uint8_t)prn;
    gps_channell_prepare(&ch[i]);
    ch[i].acq_data.found_freq_offset_hz = (int16_t)hz;
    ch[i].acq_data.found_code_phase = (uint16_t)phase;
    ch[i].acq_data.state = GPS_ACQ_DONE;
    ch[i].tracking_data.state = GPS_NEED_PRE_TRACK;
  }
  /* host mode up to the hand-over */
  uint8_t block[K * BLOCK];
  long t = 0;
  for (; t < hand && fread(block, 1, BLOCK, f) == BLOCK; t++) {
    gpsx_compat_set_packet_cnt((uint32_t)t);
    gps_tracking_process_batch(ch, n, block, (uint8_t)(t & 3));
  }
  /* the loops move to the device */
  gpsx_ctx *gx = NULL;
  gpsx_loop_state_t st[MAX_CH], *d_state = NULL;
  static uint8_t flags[K * MAX_CH];
  int changed[MAX_CH];
  if (gpsx_create(&gx, 0, NULL) != GPSX_OK || gpsx_malloc(gx, (void **)&d_state, sizeof st) != GPSX_OK)
    return 1;
  for (int i = 0; i < n; i++)
    gpsx_loop_state_from_channel(&ch[i], (uint32_t)i + 1, &st[i]);
  gpsx_memcpy_h2d(gx, d_state, st, (size_t)n * sizeof st[0]);
  long words = 0;
  for (;;) {
    const size_t got = fread(block, BLOCK, K, f);
    if (got == 0)
      break;
    if (gpsx_track_loop(gx, block, (int)got, d_state, n, (uint32_t)t, flags, NULL) != GPSX_OK) {
      fprintf(stderr, "gpsx_track_loop: %s\n", gpsx_last_error(gx));
      return 1;
    }
    const int m = gps_tracking_words_batch(ch, n, flags, (int)got, (uint32_t)t, changed, MAX_CH);
    if (m > 0) {   /* the word layer found (or gave up) inverted polarity on m channels: tell the device */
      uint8_t vals[MAX_CH];
      for (int i = 0; i < m; i++)
        vals[i] = ch[changed[i]].nav_data.inv_polarity_flag;
      gpsx_loop_set_polarity(gx, d_state, changed, vals, m);
    }
    t += (long)got;
  }
  fclose(f);
  gpsx_memcpy_d2h(gx, st, d_state, (size_t)n * sizeof st[0]);
  printf("processed_ms=%ld handed_over_at_ms=%ld\n", t, hand);
  for (int i = 0; i < n; i++) {
    gpsx_loop_state_to_channel(&st[i], &ch[i]);
    const gps_tracking_t *tr = &ch[i].tracking_data;
    uint32_t fine_bits, freq_bits;
    memcpy(&fine_bits, &tr->code_phase_fine, 4);
    memcpy(&freq_bits, &tr->if_freq_offset_hz, 4);
    words += ch[i].nav_data.word_cnt_test;
    printf("PRN=%u trk_state=%d code_phase_fine=%.3f(0x%08x) if_freq_offset_hz=%.3f(0x%08x) nco=0x%08x snr_db=%.2f bit_sync=%u "
           "false_lock_jumps=%u\n", ch[i].prn, (int)tr->state, tr->code_phase_fine, fine_bits, tr->if_freq_offset_hz, freq_bits,
           tr->if_freq_accum, tr->snr_value, ch[i].nav_data.period_sync_ok_flag, st[i].reseed_count);
  }
  printf("good_words=%ld\n", words);
  gpsx_free(gx, d_state);
  gpsx_destroy(gx);
  gpsx_compat_shutdown();
  return 0;
}
