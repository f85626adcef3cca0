/* oracle/gpsx_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the GPS L1 C/A correlator hot path of iliasam/STM32F4_SDR_GPS.
 * It exists only so that tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg have a checker
 * that travels with the repo (the reference tree does not exist on the GPU box).  Nothing under
 * stm32f4_sdr_gps_amd/ includes, links or calls it.
 *
 * Pinning: every function here is checked bit-for-bit against the reference's own C compiled in place
 * (oracle/_ref/libref_pm.so, see oracle/Makefile) by tests/test_oracle_vs_ref.py (runs wherever
 * /root/reference exists) and against the committed golden vectors in tests/golden/ (runs everywhere).
 *
 * Reference citations: PM = /root/reference/Firmware/project_main.
 */
#ifndef GPSX_ORACLE_H
#define GPSX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  ORC_CHIPS    = 1023,   /* PM/config.h:28  PRN_LENGTH            */
  ORC_SAMPLES  = 16368,  /* PM/config.h:26  BITS_IN_PRN           */
  ORC_BYTES    = 2046,   /* 1-bit samples, LSB first              */
  ORC_WORDS16  = 1023,   /* PM/config.h:27  PRN_SPI_WORDS_CNT     */
  ORC_IF_HZ    = 4092000 /* PM/config.h:23  IF_FREQ_HZ            */
};

/* a1: C/A Gold code, chips as 0/1 bytes (PM/GPS/gps_misc.c:317-372).  Returns 0, or -1 for an unsupported prn
 * (prn < 1 leaves `chips` untouched exactly like the reference's silent return). */
int orc_ca_code(int prn, uint8_t chips[ORC_CHIPS]);

/* a3: replica expansion, 16 samples per chip, shifted left by offset_bits&15 (PM/GPS/gps_misc.c:282-300).
 * `out` has 1024 words; words 0..1022 are fully defined; word 1023 (pad) is OR-ed with the spill of chip 1022. */
void orc_replica(const uint8_t chips[ORC_CHIPS], unsigned offset_bits, uint16_t out[ORC_WORDS16 + 1]);

/* NCO accumulator step for ONE sample: (uint32)(freq_hz / 0.003810972f), float32 arithmetic
 * (PM/GPS/gps_misc.c:219,250; PM/config.h:50). */
uint32_t orc_nco_step(float freq_hz);

/* a4/a5: carrier wipe-off of one 1 ms block.  *accum is the NCO accumulator on entry and exit (pass a zeroed
 * variable for the stateless gps_shift_to_zero_freq, PM/GPS/gps_misc.c:211-240; the tracking variant
 * :244-274 passes trk->if_freq_accum and freq_hz = (float)IF + if_freq_offset_hz).  Only bytes 0..2043 of
 * data_i/data_q are written. */
void orc_wipeoff(const uint8_t signal[ORC_BYTES], float freq_hz, uint32_t *accum,
                 uint8_t *data_i, uint8_t *data_q);

/* a6: NCO accumulator after `steps` skipped milliseconds (PM/GPS/gps_misc.c:196-204). */
uint32_t orc_rewind(float if_freq_offset_hz, uint32_t accum, unsigned steps);

/* a7: XOR + popcount of data vs replica at a byte offset 0..2046 (PM/GPS/gps_misc.c:48-93). */
void orc_mult_and_summ(const uint8_t *data_i, const uint8_t *data_q, const uint8_t *replica,
                       unsigned offset, uint16_t *cnt_i, uint16_t *cnt_q);

/* a8 (PM/GPS/gps_misc.c:98-122), a9 (:128-145), a10 (:155-191) */
int16_t  orc_correlation8(const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q, unsigned offset);
void     orc_correlation_iq(const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q,
                            unsigned offset, int16_t *res_i, int16_t *res_q);
uint16_t orc_correlation_search(const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q,
                                unsigned start_shift, unsigned stop_shift, uint16_t *aver_val, uint16_t *phase);

/* magnitude part of a8 on already centred counts (shared with the batched oracles) */
int16_t orc_mag8(int cnt_i, int cnt_q);

/* ---- batched semantics of the engine's Tier-3 entry points (include/gpsx.h), as loops over the primitives ---- */

typedef struct {
  uint32_t max_val;   /* max over the window of  sum_ms corr8                                    */
  uint32_t phase;     /* first offset reaching it (byte offset 0..2045); 0 if max_val == 0       */
  uint32_t sum;       /* sum over the window                                                      */
  uint32_t avr;       /* sum / 2046   (PM/GPS/gps_misc.c:178: divisor is constant)                */
} orc_peak_t;

/* One (PRN, Doppler, replica bit shift) search over n_ms consecutive blocks, firmware buffer semantics
 * (data words 1022 of I/Q stay zero: PM/GPS/common_ram.c:3-5 + gps_misc.c:229).
 * energy_opt (may be NULL): 2046 per-offset accumulated corr8 values (offsets outside the window = 0).
 * per_ms_opt (may be NULL): n_ms peaks, one per block, each exactly the triplet correlation_search returns. */
void orc_search_job(const uint8_t *if_blocks, int n_ms, const uint8_t chips[ORC_CHIPS], float freq_hz,
                    unsigned offset_bits, unsigned start_shift, unsigned stop_shift,
                    orc_peak_t *peak, uint32_t *energy_opt, orc_peak_t *per_ms_opt);

/* Full grid: peaks[prn_idx][dopp_idx][b] (b = 0..n_bits-1), freq = (float)(IF + dopp_min + idx*dopp_step)
 * evaluated in int then converted, as PM/GPS/acquisition.c:285-289 does.  n_threads <= 1: serial. */
void orc_acq_grid(const uint8_t *if_blocks, int n_ms, const uint8_t *prns, int n_prn,
                  int dopp_min_hz, int dopp_step_hz, int n_dopp, int n_bits,
                  orc_peak_t *peaks, int n_threads);

/* One tracking correlator step (PM/GPS/tracking.c:115-138): replica with (fine & 7) shift, stateful wipe-off,
 * E/P/L I/Q.  iq_out = {IE,QE,IP,QP,IL,QL}.  *accum is advanced. */
void orc_track_epl(const uint8_t signal[ORC_BYTES], const uint8_t chips[ORC_CHIPS], float code_phase_fine,
                   float if_freq_offset_hz, uint32_t *accum, int16_t iq_out[6]);

/* ---- EXTENSION (not in the reference: it wires the MAX2769's sign bit only, PM/config.h:16): weighted correlation of
 * two-bit sign/magnitude captures (include/gpsx.h gpsx_acq_grid_weighted).  Own entry points; nothing above calls them.
 *   sample n of a 4092-byte block: sign = bit 2 (n & 3) of byte n >> 2, magnitude = the bit above it;
 *   value v[n] = (sign ? +1 : -1) * (magnitude && use_magnitude ? 3 : 1);
 *   carrier wipe-off = the reference's NCO on the sign (orc_wipeoff, phase 0 at the block's start): the sixteen samples
 *   it never mixes (n >= 16352) carry weight 0;
 *   replica at fine phase tau: chip ((n - tau) mod 16368) / 16, +1 for a 0 chip, -1 for a 1 chip;
 *   I(tau) = sum_n vI[n] c(n, tau), Q likewise; magnitude = floor(sqrt(I^2 + Q^2)), exact integers. */
/* one hypothesis straight from the definition, sample by sample */
void orc_weighted_iq(const uint8_t if_2bit[2 * ORC_BYTES], int prn, float freq_hz, unsigned tau, int use_magnitude,
                     int32_t *i_out, int32_t *q_out);
/* the grid: peaks[search][prn][dopp] = {max over tau, first tau reaching it, sum over tau, sum / 16368}; search s reads
 * block s * stride_blocks; freq = (float)(IF + dopp_min + idx * dopp_step) as orc_acq_grid */
void orc_acq_grid_weighted(const uint8_t *if_2bit_blocks, int n_search, int stride_blocks, const uint8_t *prns, int n_prn,
                           int dopp_min_hz, int dopp_step_hz, int n_dopp, int use_magnitude, orc_peak_t *peaks, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
