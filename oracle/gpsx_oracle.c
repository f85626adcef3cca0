/* oracle/gpsx_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see gpsx_oracle.h).
 *
 * From-scratch CPU restatement of the correlator hot path of iliasam/STM32F4_SDR_GPS.  Each function names the
 * reference lines whose observable behaviour it reproduces (PM = /root/reference/Firmware/project_main).
 * The formulations are deliberately NOT the reference's (bit-stream / circular-alignment views instead of
 * pointer walks and 32-bit read-modify-writes) so that agreement with oracle/_ref is evidence, not tautology.
 */
#include "gpsx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------------------
 * a1  C/A Gold code                                                         PM/GPS/gps_misc.c:317-372
 * PRN 1..32 are generated with the IS-GPS-200 G2 phase-selector taps (an independent parametrisation of the
 * same codes the reference builds from G2 delays); PRN 33..210 use the G2 delay in chips.
 * ---------------------------------------------------------------------------------------------------------- */
static const uint8_t k_g2_taps[32][2] = {
  {2, 6}, {3, 7}, {4, 8}, {5, 9}, {1, 9}, {2, 10}, {1, 8}, {2, 9}, {3, 10}, {2, 3}, {3, 4}, {5, 6}, {6, 7}, {7, 8},
  {8, 9}, {9, 10}, {1, 4}, {2, 5}, {3, 6}, {4, 7}, {5, 8}, {6, 9}, {1, 3}, {4, 6}, {5, 7}, {6, 8}, {7, 9}, {8, 10},
  {1, 6}, {2, 7}, {3, 8}, {4, 9}
};

/* G2 delay in chips for PRN 33..210 (IS-GPS-200 table 3-Ia/3-Ib as used by the reference's table,
 * gps_misc.c:319-341, including its PRN 34 == PRN 37 duplicate). */
static const uint16_t k_g2_delay_33_210[178] = {
  863, 950, 947, 948, 950, 67, 103, 91,
  19, 679, 225, 625, 946, 638, 161, 1001, 554, 280,
  710, 709, 775, 864, 558, 220, 397, 55, 898, 759,
  367, 299, 1018, 729, 695, 780, 801, 788, 732, 34,
  320, 327, 389, 407, 525, 405, 221, 761, 260, 326,
  955, 653, 699, 422, 188, 438, 959, 539, 879, 677,
  586, 153, 792, 814, 446, 264, 1015, 278, 536, 819,
  156, 957, 159, 712, 885, 461, 248, 713, 126, 807,
  279, 122, 197, 693, 632, 771, 467, 647, 203, 145,
  175, 52, 21, 237, 235, 886, 657, 634, 762, 355,
  1012, 176, 603, 130, 359, 595, 68, 386, 797, 456,
  499, 883, 307, 127, 211, 121, 118, 163, 628, 853,
  484, 289, 811, 202, 1021, 463, 568, 904, 670, 230,
  911, 684, 309, 644, 932, 12, 314, 891, 212, 185,
  675, 503, 150, 395, 345, 846, 798, 992, 357, 995,
  877, 112, 144, 476, 193, 109, 445, 291, 87, 399,
  292, 901, 339, 208, 711, 189, 263, 537, 663, 942,
  173, 900, 30, 500, 935, 556, 373, 85, 652, 310
};

int orc_ca_code(int prn, uint8_t chips[ORC_CHIPS])
{
  if (prn < 1)
    return 0; /* silent, buffer untouched (gps_misc.c:345) */
  if (prn > 210)
    return -1;

  /* 10-stage Fibonacci LFSRs as bit masks: bit (k-1) holds stage k, all ones at start. */
  unsigned g1 = 0x3FF, g2 = 0x3FF;
  uint8_t g1_seq[ORC_CHIPS], g2_seq[ORC_CHIPS], g2_sel[ORC_CHIPS];
  const int t1 = (prn <= 32) ? k_g2_taps[prn - 1][0] : 10;
  const int t2 = (prn <= 32) ? k_g2_taps[prn - 1][1] : 10;
  for (int i = 0; i < ORC_CHIPS; i++) {
    g1_seq[i] = (g1 >> 9) & 1;
    g2_seq[i] = (g2 >> 9) & 1;
    g2_sel[i] = ((g2 >> (t1 - 1)) ^ (g2 >> (t2 - 1))) & 1;
    unsigned f1 = ((g1 >> 2) ^ (g1 >> 9)) & 1;                                              /* 1 + x^3 + x^10 */
    unsigned f2 = ((g2 >> 1) ^ (g2 >> 2) ^ (g2 >> 5) ^ (g2 >> 7) ^ (g2 >> 8) ^ (g2 >> 9)) & 1; /* x^2,3,6,8,9,10 */
    g1 = ((g1 << 1) | f1) & 0x3FF;
    g2 = ((g2 << 1) | f2) & 0x3FF;
  }
  if (prn <= 32) {
    for (int i = 0; i < ORC_CHIPS; i++)
      chips[i] = g1_seq[i] ^ g2_sel[i];
  } else {
    const int delay = k_g2_delay_33_210[prn - 33];
    for (int i = 0; i < ORC_CHIPS; i++)
      chips[i] = g1_seq[i] ^ g2_seq[(i + ORC_CHIPS - delay) % ORC_CHIPS];
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * a3  replica                                                               PM/GPS/gps_misc.c:282-300
 * Bit-stream view: sample n of the replica is chip (n - b) / 16 for n >= b and 0 below (quirk Q5: the shift is
 * not circular).  Samples 16368..16383 land in the pad word, which the reference never clears (its memset
 * covers 2046 bytes) and only ORs into.
 * ---------------------------------------------------------------------------------------------------------- */
void orc_replica(const uint8_t chips[ORC_CHIPS], unsigned offset_bits, uint16_t out[ORC_WORDS16 + 1])
{
  const unsigned b = offset_bits & 15u;
  uint16_t pad = out[ORC_WORDS16];
  memset(out, 0, ORC_BYTES);
  for (unsigned n = b; n < (unsigned)ORC_SAMPLES + b; n++) {
    if (chips[(n - b) >> 4]) {
      if (n < (unsigned)ORC_SAMPLES)
        out[n >> 4] |= (uint16_t)(1u << (n & 15));
      else
        pad |= (uint16_t)(1u << (n & 15));
    }
  }
  out[ORC_WORDS16] = pad;
}

/* ------------------------------------------------------------------------------------------------------------
 * a4/a5/a6  carrier NCO                                                     PM/GPS/gps_misc.c:196-274
 * ---------------------------------------------------------------------------------------------------------- */
uint32_t orc_nco_step(float freq_hz)
{
  const float hz_per_lsb = 0.003810972f;  /* PM/config.h:50 (a rounded 16.368e6 / 2^32) */
  volatile float q = freq_hz / hz_per_lsb; /* binary32 quotient, no excess precision */
  return (uint32_t)q;
}

/* Fs/4 square-wave patterns, 32 samples each, indexed by NCO quadrant.  Quadrant 0 of the in-phase table and
 * quadrant 1 of the quadrature table are the reference's 7-nibble literal 0x9999999 (quirk Q1). */
static const uint32_t k_carrier_i[4] = { 0x09999999u, 0xCCCCCCCCu, 0x66666666u, 0x33333333u };
static const uint32_t k_carrier_q[4] = { 0x33333333u, 0x09999999u, 0xCCCCCCCCu, 0x66666666u };

void orc_wipeoff(const uint8_t signal[ORC_BYTES], float freq_hz, uint32_t *accum, uint8_t *data_i, uint8_t *data_q)
{
  const uint32_t step_word = (uint32_t)((uint64_t)orc_nco_step(freq_hz) * 32u);
  uint32_t acc = *accum;
  for (int w = 0; w < 511; w++) {      /* 511 whole 32-sample words; samples 16352..16367 are not mixed (Q2) */
    const unsigned quad = acc >> 30;
    for (int k = 0; k < 4; k++) {
      const uint8_t x = signal[4 * w + k];
      data_i[4 * w + k] = x ^ (uint8_t)(k_carrier_i[quad] >> (8 * k));
      data_q[4 * w + k] = x ^ (uint8_t)(k_carrier_q[quad] >> (8 * k));
    }
    acc += step_word;
  }
  *accum = acc;
}

uint32_t orc_rewind(float if_freq_offset_hz, uint32_t accum, unsigned steps)
{
  const uint32_t step = orc_nco_step((float)ORC_IF_HZ + if_freq_offset_hz);
  const uint64_t adv = (uint64_t)step * (uint64_t)ORC_SAMPLES * (uint64_t)(steps & 0xFFu);
  return accum + (uint32_t)adv;
}

/* ------------------------------------------------------------------------------------------------------------
 * a7  XOR / popcount core                                                   PM/GPS/gps_misc.c:48-93
 * Circular-alignment view: replica word i meets data bytes (o + 2i) mod 2046 and the next one.
 * Even offsets (and 2046 == 0) use all 1023 words.  Odd offsets drop the word that would straddle the buffer
 * wrap, p1 = (2045 - o) / 2, and the last word, 1022 (quirk Q3).
 * ---------------------------------------------------------------------------------------------------------- */
static inline unsigned pop16(unsigned v) { return (unsigned)__builtin_popcount(v & 0xFFFFu); }

static inline unsigned rd16(const uint8_t *p) { return (unsigned)p[0] | ((unsigned)p[1] << 8); }

void orc_mult_and_summ(const uint8_t *data_i, const uint8_t *data_q, const uint8_t *replica, unsigned offset,
                       uint16_t *cnt_i, uint16_t *cnt_q)
{
  const unsigned o = offset % ORC_BYTES;      /* 2046 behaves as 0 */
  const unsigned odd = o & 1u;
  const unsigned wrap_word = (ORC_BYTES - o) / 2; /* first replica word whose data would start at/after the wrap */
  unsigned ci = 0, cq = 0;

  /* run 1: replica words [0, wrap_word) against data starting at byte o */
  for (unsigned i = 0; i < wrap_word; i++) {
    const unsigned r = rd16(replica + 2 * i);
    ci += pop16(rd16(data_i + o + 2 * i) ^ r);
    cq += pop16(rd16(data_q + o + 2 * i) ^ r);
  }
  /* run 2: after the wrap.  Odd offsets skip the straddling word and resume at data byte 1; they also stop one
   * replica word early. */
  const unsigned first = wrap_word + odd;
  const unsigned last = ORC_WORDS16 - odd; /* exclusive */
  for (unsigned i = first; i < last; i++) {
    const unsigned r = rd16(replica + 2 * i);
    const unsigned d = odd + 2 * (i - first);
    ci += pop16(rd16(data_i + d) ^ r);
    cq += pop16(rd16(data_q + d) ^ r);
  }
  *cnt_i = (uint16_t)ci;
  *cnt_q = (uint16_t)cq;
}

/* ------------------------------------------------------------------------------------------------------------
 * a8/a9/a10                                                                 PM/GPS/gps_misc.c:98-191
 * ---------------------------------------------------------------------------------------------------------- */
int16_t orc_mag8(int cnt_i, int cnt_q)
{
  int i = cnt_i - ORC_SAMPLES / 2;
  int q = cnt_q - ORC_SAMPLES / 2;
  if (i < 0) i = 0;   /* one-sided clip (quirk Q4) */
  if (q < 0) q = 0;
  const float e = (float)(i * i) + (float)(q * q);
  return (int16_t)sqrtf(e);
}

int16_t orc_correlation8(const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q, unsigned offset)
{
  uint16_t ci, cq;
  orc_mult_and_summ((const uint8_t *)data_i, (const uint8_t *)data_q, (const uint8_t *)replica, offset, &ci, &cq);
  return orc_mag8(ci, cq);
}

void orc_correlation_iq(const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q, unsigned offset,
                        int16_t *res_i, int16_t *res_q)
{
  uint16_t ci, cq;
  orc_mult_and_summ((const uint8_t *)data_i, (const uint8_t *)data_q, (const uint8_t *)replica, offset, &ci, &cq);
  *res_i = (int16_t)((int)ci - ORC_SAMPLES / 2);
  *res_q = (int16_t)((int)cq - ORC_SAMPLES / 2);
}

uint16_t orc_correlation_search(const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q,
                                unsigned start_shift, unsigned stop_shift, uint16_t *aver_val, uint16_t *phase)
{
  int best = 0;
  unsigned best_at = 0;
  int32_t total = 0;
  for (unsigned o = start_shift; o < stop_shift; o++) {
    const int c = orc_correlation8(replica, data_i, data_q, o);
    if (c > best) { best = c; best_at = o; }   /* strict: the first maximum wins, all-zero leaves phase 0 */
    total += c;
  }
  total /= 2 * ORC_CHIPS;                       /* constant divisor whatever the window (quirk Q9) */
  if (total < 0) total = 0;
  *aver_val = (uint16_t)total;
  *phase = (uint16_t)best_at;
  return (uint16_t)best;
}

/* ------------------------------------------------------------------------------------------------------------
 * Batched semantics (engine Tier 3), defined as loops over the primitives above with the firmware's buffer
 * discipline: I/Q scratch words 1022 are zero because nothing ever writes them (common_ram.c:3-5).
 * ---------------------------------------------------------------------------------------------------------- */
void orc_search_job(const uint8_t *if_blocks, int n_ms, const uint8_t chips[ORC_CHIPS], float freq_hz,
                    unsigned offset_bits, unsigned start_shift, unsigned stop_shift, orc_peak_t *peak,
                    uint32_t *energy_opt, orc_peak_t *per_ms_opt)
{
  uint16_t rep[ORC_WORDS16 + 1], di[ORC_WORDS16 + 1], dq[ORC_WORDS16 + 1];
  uint32_t energy[ORC_BYTES];
  memset(rep, 0, sizeof rep);
  memset(di, 0, sizeof di);
  memset(dq, 0, sizeof dq);
  memset(energy, 0, sizeof energy);
  orc_replica(chips, offset_bits, rep);
  if (stop_shift > (unsigned)ORC_BYTES) stop_shift = ORC_BYTES;

  for (int ms = 0; ms < n_ms; ms++) {
    uint32_t acc = 0;
    orc_wipeoff(if_blocks + (size_t)ms * ORC_BYTES, freq_hz, &acc, (uint8_t *)di, (uint8_t *)dq);
    uint32_t mx = 0, at = 0, sum = 0;
    for (unsigned o = start_shift; o < stop_shift; o++) {
      const uint32_t c = (uint32_t)orc_correlation8(rep, di, dq, o);
      energy[o] += c;
      if (c > mx) { mx = c; at = o; }
      sum += c;
    }
    if (per_ms_opt) {
      per_ms_opt[ms].max_val = mx;
      per_ms_opt[ms].phase = at;
      per_ms_opt[ms].sum = sum;
      per_ms_opt[ms].avr = sum / (2 * ORC_CHIPS);
    }
  }
  uint32_t mx = 0, at = 0, sum = 0;
  for (unsigned o = start_shift; o < stop_shift; o++) {
    if (energy[o] > mx) { mx = energy[o]; at = o; }
    sum += energy[o];
  }
  peak->max_val = mx;
  peak->phase = at;
  peak->sum = sum;
  peak->avr = sum / (2 * ORC_CHIPS);
  if (energy_opt)
    memcpy(energy_opt, energy, sizeof energy);
}

void orc_acq_grid(const uint8_t *if_blocks, int n_ms, const uint8_t *prns, int n_prn, int dopp_min_hz,
                  int dopp_step_hz, int n_dopp, int n_bits, orc_peak_t *peaks, int n_threads)
{
  const int n_jobs = n_prn * n_dopp;
#ifdef _OPENMP
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#else
  (void)n_threads;
#endif
  for (int job = 0; job < n_jobs; job++) {
    const int p = job / n_dopp, d = job % n_dopp;
    uint8_t chips[ORC_CHIPS];
    memset(chips, 0, sizeof chips);
    orc_ca_code(prns[p], chips);
    const float freq_hz = (float)(ORC_IF_HZ + dopp_min_hz + d * dopp_step_hz);
    for (int b = 0; b < n_bits; b++)
      orc_search_job(if_blocks, n_ms, chips, freq_hz, (unsigned)b, 0, ORC_BYTES,
                     &peaks[((size_t)p * n_dopp + d) * n_bits + b], NULL, NULL);
  }
}

/* ------------------------------------------------------------------------------------------------------------
 * Tracking correlator step                                                  PM/GPS/tracking.c:115-138
 * ---------------------------------------------------------------------------------------------------------- */
void orc_track_epl(const uint8_t signal[ORC_BYTES], const uint8_t chips[ORC_CHIPS], float code_phase_fine,
                   float if_freq_offset_hz, uint32_t *accum, int16_t iq_out[6])
{
  uint16_t rep[ORC_WORDS16 + 1], di[ORC_WORDS16 + 1], dq[ORC_WORDS16 + 1];
  memset(rep, 0, sizeof rep);
  memset(di, 0, sizeof di);
  memset(dq, 0, sizeof dq);

  const int16_t fine = (int16_t)code_phase_fine;
  orc_replica(chips, (unsigned)(fine & 7), rep);
  orc_wipeoff(signal, (float)ORC_IF_HZ + if_freq_offset_hz, accum, (uint8_t *)di, (uint8_t *)dq);

  const uint16_t prompt = (uint16_t)(fine / 8);
  uint16_t early = (uint16_t)(prompt - 1), late = (uint16_t)(prompt + 1);
  if (early >= 2 * ORC_CHIPS) early = 2 * ORC_CHIPS - 1;
  if (late >= 2 * ORC_CHIPS) late = 0;

  orc_correlation_iq(rep, di, dq, early, &iq_out[0], &iq_out[1]);
  orc_correlation_iq(rep, di, dq, prompt, &iq_out[2], &iq_out[3]);
  orc_correlation_iq(rep, di, dq, late, &iq_out[4], &iq_out[5]);
}


/* ------------------------------------------------------------------------------------------------------------
 * EXTENSION: weighted two-bit correlation (see gpsx_oracle.h).  Not a restatement of reference code: the reference
 * has no such mode.  The grid works on sums of sixteen samples (one chip's worth) per code-phase offset; the single-hypothesis
 * function below it is the definition itself and the tests pin one against the other.
 * ---------------------------------------------------------------------------------------------------------- */
static void weighted_streams(const uint8_t *blk2, float freq_hz, int use_magnitude, int8_t *vi, int8_t *vq)
{
  uint8_t sign[ORC_BYTES], mag[ORC_BYTES], di[ORC_BYTES], dq[ORC_BYTES];
  memset(sign, 0, sizeof sign);
  memset(mag, 0, sizeof mag);
  memset(di, 0, sizeof di);
  memset(dq, 0, sizeof dq);
  for (int n = 0; n < ORC_SAMPLES; n++) {
    const unsigned pair = (blk2[n >> 2] >> (2 * (n & 3))) & 3u;
    sign[n >> 3] |= (uint8_t)((pair & 1u) << (n & 7));
    mag[n >> 3] |= (uint8_t)((pair >> 1) << (n & 7));
  }
  uint32_t acc = 0;
  orc_wipeoff(sign, freq_hz, &acc, di, dq);
  for (int n = 0; n < ORC_SAMPLES; n++) {
    const int w = (use_magnitude && ((mag[n >> 3] >> (n & 7)) & 1)) ? 3 : 1;
    const int mixed = n < 511 * 32;            /* the NCO loop's 511 words; the last sixteen samples stay out */
    vi[n] = (int8_t)(mixed ? (((di[n >> 3] >> (n & 7)) & 1) ? w : -w) : 0);
    vq[n] = (int8_t)(mixed ? (((dq[n >> 3] >> (n & 7)) & 1) ? w : -w) : 0);
  }
}

void orc_weighted_iq(const uint8_t if_2bit[2 * ORC_BYTES], int prn, float freq_hz, unsigned tau, int use_magnitude,
                     int32_t *i_out, int32_t *q_out)
{
  static int8_t vi[ORC_SAMPLES], vq[ORC_SAMPLES];
  uint8_t chips[ORC_CHIPS];
  orc_ca_code(prn, chips);
  weighted_streams(if_2bit, freq_hz, use_magnitude, vi, vq);
  int32_t si = 0, sq = 0;
  for (int n = 0; n < ORC_SAMPLES; n++) {
    const int c = chips[((n + ORC_SAMPLES - (int)(tau % ORC_SAMPLES)) % ORC_SAMPLES) / 16] ? -1 : 1;
    si += c * vi[n];
    sq += c * vq[n];
  }
  *i_out = si;
  *q_out = sq;
}

static uint32_t isqrt_u64(uint64_t e)
{
  uint64_t r = (uint64_t)sqrt((double)e);
  while (r * r > e) r--;
  while ((r + 1) * (r + 1) <= e) r++;
  return (uint32_t)r;
}

void orc_acq_grid_weighted(const uint8_t *if_2bit_blocks, int n_search, int stride_blocks, const uint8_t *prns, int n_prn,
                           int dopp_min_hz, int dopp_step_hz, int n_dopp, int use_magnitude, orc_peak_t *peaks, int n_threads)
{
  const int n_jobs = n_search * n_dopp;
#ifdef _OPENMP
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#else
  (void)n_threads;
#endif
  for (int job = 0; job < n_jobs; job++) {
    const int s = job / n_dopp, d = job % n_dopp;
    const float freq_hz = (float)(ORC_IF_HZ + dopp_min_hz + d * dopp_step_hz);
    int8_t *vi = malloc(ORC_SAMPLES), *vq = malloc(ORC_SAMPLES);
    /* chip sums: sum[t0][k] = sum_{j < 16} v[(16 k + t0 + j) mod 16368], doubled so that k + c needs no wrap */
    int16_t (*sumi)[2 * ORC_CHIPS] = malloc(sizeof(int16_t[16][2 * ORC_CHIPS]));
    int16_t (*sumq)[2 * ORC_CHIPS] = malloc(sizeof(int16_t[16][2 * ORC_CHIPS]));
    weighted_streams(if_2bit_blocks + (size_t)s * stride_blocks * 2 * ORC_BYTES, freq_hz, use_magnitude, vi, vq);
    for (int t0 = 0; t0 < 16; t0++)
      for (int k = 0; k < ORC_CHIPS; k++) {
        int a = 0, b = 0;
        for (int j = 0; j < 16; j++) {
          const int n = (16 * k + t0 + j) % ORC_SAMPLES;
          a += vi[n];
          b += vq[n];
        }
        sumi[t0][k] = sumi[t0][k + ORC_CHIPS] = (int16_t)a;
        sumq[t0][k] = sumq[t0][k + ORC_CHIPS] = (int16_t)b;
      }
    for (int p = 0; p < n_prn; p++) {
      uint8_t chips[ORC_CHIPS];
      int16_t csign[ORC_CHIPS];
      memset(chips, 0, sizeof chips);
      orc_ca_code(prns[p], chips);
      for (int c = 0; c < ORC_CHIPS; c++)
        csign[c] = chips[c] ? -1 : 1;
      uint32_t mx = 0, at = 0, sum = 0;
      for (int q = 0; q < ORC_CHIPS; q++)
        for (int t0 = 0; t0 < 16; t0++) {
          /* tau = 16 q + t0: chip c of the replica lies on samples 16 (q + c) + t0 .. + 15 */
          int32_t ai = 0, aq = 0;
          const int16_t *ri = &sumi[t0][q], *rq = &sumq[t0][q];
          for (int c = 0; c < ORC_CHIPS; c++) {
            ai += csign[c] * ri[c];
            aq += csign[c] * rq[c];
          }
          const uint32_t m = isqrt_u64((uint64_t)((int64_t)ai * ai) + (uint64_t)((int64_t)aq * aq));
          const uint32_t tau = (uint32_t)(16 * q + t0);
          if (m > mx || (m == mx && m > 0 && tau < at)) { mx = m; at = tau; }
          sum += m;
        }
      orc_peak_t *pk = &peaks[((size_t)s * n_prn + p) * n_dopp + d];
      pk->max_val = mx;
      pk->phase = at;
      pk->sum = sum;
      pk->avr = sum / ORC_SAMPLES;
    }
    free(vi); free(vq); free(sumi); free(sumq);
  }
}
