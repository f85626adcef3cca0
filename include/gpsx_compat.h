/* include/gpsx_compat.h -- per-call, symbol-compatible mirror of the reference's correlator interface
 * (Firmware/project_main/GPS/gps_misc.h:195-216 of iliasam/STM32F4_SDR_GPS), backed by the MI355X engine.
 *
 * A C program written against the reference header links against libgpsx.so unchanged: same function names,
 * argument meaning and (silent) error behaviour; the channel structures below keep the reference's field names
 * and, on LP64 x86-64, its exact layout (gps_misc.h:43-99,184-193; checked by static asserts in gpsx_compat.cpp
 * and against the reference header by tests/test_abi_and_host.py::test_compat_struct_layout_matches_reference_header_when_present).  Every call runs on the GPU through the
 * process-wide default context (device $GPSX_DEVICE, default 0); if no gfx950 device can be opened the first call
 * prints a diagnostic and aborts -- there is no CPU path.
 *
 * One call = a few small copies + one or two kernel launches (tens of microseconds).  That is the price of keeping
 * the reference's call granularity; throughput work belongs on the batched interface in gpsx.h.
 * Like the reference (global scratch buffers, PM/GPS/common_ram.c), this interface is not re-entrant.
 */
#ifndef GPSX_COMPAT_H
#define GPSX_COMPAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* signal geometry, PM/config.h:23-28,41-59 */
#define IF_FREQ_HZ                 4092000
#define SPI_BAUDRATE_HZ            16368000
#define BITS_IN_PRN                16368
#define PRN_SPI_WORDS_CNT          1023
#define PRN_LENGTH                 1023
#define ACQ_SEARCH_FREQ_HZ         7000
#define ACQ_SEARCH_STEP_HZ         500
#define ACQ_COUNT                  (ACQ_SEARCH_FREQ_HZ * 2 / ACQ_SEARCH_STEP_HZ + 1)
#define ACQ_PHASE1_HIST_SIZE       32
#define PRE_TRACK_POINTS_MAX_CNT   30
#define TRACKING_CH_LENGTH         4
#define PRN_SPEED_HZ               1000   /* code periods per second (PM/config.h:25) */
#define GPS_SAT_CNT                4
#define GPS_DATA_WORDS_CNT         (PRN_SPI_WORDS_CNT + 1)   /* PM/GPS/common_ram.h:10 */
#define IF_NCO_STEP_HZ             (0.003810972f)            /* PM/config.h:50 */
/* tracking loop gains, PM/config.h:61-71 */
#define TRACKING_DLL1_C1           (1.0f)
#define TRACKING_DLL1_C2           (300.0f)
#define TRACKING_PLL1_C1           (4.0f)
#define TRACKING_PLL1_C2           (3000.0f)
#define TRACKING_PLL2_C1           (8.0f)
#define TRACKING_PLL2_C2           (5000.0f)
#define TRACKING_FLL1_C1           (200.0f)
#define TRACKING_FLL1_C2           (2000.0f)

/* Channel state machines.  The member names and their order are the reference's (a host reads and writes them, and the
 * step logic is checked byte for byte against the reference's records), so the enumerators and the structure members
 * below cannot differ from gps_misc.h:20-99; same-typed neighbours are declared together. */
typedef enum {
  GPS_ACQ_NEED_FREQ_SEARCH = 0, GPS_ACQ_FREQ_SEARCH_RUN, GPS_ACQ_FREQ_SEARCH_DONE,      /* Doppler sweep                  */
  GPS_ACQ_CODE_PHASE_SEARCH1, GPS_ACQ_CODE_PHASE_SEARCH1_DONE,                           /* coarse phase histogram         */
  GPS_ACQ_CODE_PHASE_SEARCH2, GPS_ACQ_CODE_PHASE_SEARCH2_DONE,                           /* narrowed window                */
  GPS_ACQ_CODE_PHASE_SEARCH3, GPS_ACQ_CODE_PHASE_SEARCH3_DONE,                           /* all channels together          */
  GPS_ACQ_DONE
} gps_acq_state_t;

typedef enum { GPS_TRACKNG_IDLE = 0, GPS_NEED_PRE_TRACK, GPS_PRE_TRACK_RUN, GPS_PRE_TRACK_DONE, GPS_TRACKING_RUN } gps_tracking_state_t;

/* acquisition state of one channel (60 bytes) */
typedef struct {
  uint8_t  freq_index;                                     /* Doppler bin under test: -7000 Hz + index * 500 Hz          */
  int16_t  found_freq_offset_hz, given_freq_offset_hz;     /* result; user hint (non-zero skips the frequency search)    */
  uint16_t found_code_phase, code_search_start, code_search_stop, code_hist_step;   /* byte offsets, 0 .. 2046         */
  gps_acq_state_t state;
  uint8_t  code_phase_histogram[ACQ_PHASE1_HIST_SIZE];
  uint32_t start_timestamp;                                /* 1 ms ticks                                                 */
  float    hist_ratio;
} gps_acq_t;

/* tracking state of one channel (152 bytes; the reference built with ENABLE_CODE_FILTER == 1) */
typedef struct {
  uint16_t code_search_start, code_search_stop;            /* pre-tracking window, byte offsets                          */
  float    if_freq_offset_hz;                              /* Doppler estimate the carrier NCO runs at                   */
  uint32_t if_freq_accum;                                  /* carrier NCO accumulator carried across milliseconds        */
  uint16_t pre_track_phases[PRE_TRACK_POINTS_MAX_CNT];
  uint8_t  pre_track_count;
  uint32_t prev_track_timestamp;
  float    code_phase_fine, old_code_phase_fine;           /* samples, 0 .. 16368                                        */
  uint8_t  code_phase_swap_flag;
  float    dll_code_err, pll_code_err;                     /* loop filter memories                                       */
  int16_t  fll_old_i, fll_old_q;
  float    fll_err;
  int16_t  pll_check_buf[TRACKING_CH_LENGTH];              /* false-lock detector                                        */
  uint8_t  pll_bad_state_cnt;
  uint16_t pll_bad_state_master_cnt;
  uint32_t i_part_summ, q_part_summ;                       /* SNR estimator                                              */
  uint16_t snr_summ_cnt;
  float    snr_value;
  uint32_t filt_start_time_ms;
  uint16_t code_filt_cnt;
  float    code_phase_fine_filt;
  gps_tracking_state_t state;
} gps_tracking_t;

/* Navigation-data state of a channel (gps_misc.h:101-133): 20 ms bit synchronisation, which the tracking step reads
 * (PLL gain selection) and its nav-bit hook writes; then the word layer -- preamble search, parity, polarity, subframe
 * assembly and the subframe's time stamp.  Observation and ephemeris state (pseudorange, decoded orbit) are outside this
 * library and are carried as opaque storage of the reference's size so that gps_ch_t keeps its layout
 * (gps_misc.h:134-182). */
#define GPS_NAV_WORD_LENGTH           30   /* bits */
#define GPS_NAV_SUBFRAME_LENGTH_BYTES 38   /* 300 bits */
typedef struct {
  uint8_t  period_sync_ok_flag, right_period_cnt;          /* 20 ms bit period found / confidence counter                */
  uint32_t old_swap_time;                                  /* ms tick of the last sign change                            */
  uint8_t  old_reminder;
  uint8_t  accurate_swap_time, accurate_swap_ok;           /* bit edge inside the 20 ms grid, 0..19 / valid              */
  uint8_t  last_bit_pos_cnt, last_bit_neg_cnt;             /* votes for the bit being integrated                         */
  uint8_t  inv_polarity_flag, polarity_found;              /* Costas loop 180 degrees off / confirmed by a good word     */
  uint8_t  inv_preabmle_cnt;                               /* inverted preambles seen while hunting                      */
  uint8_t  word_buf[GPS_NAV_WORD_LENGTH];                  /* one bit per byte: the word being collected                 */
  uint8_t  word_cnt, word_bit_cnt;                         /* words of the subframe received (0 = hunting) / bits        */
  uint8_t  old_D29, old_D30;                               /* last two bits of the previous word (parity equations)      */
  uint32_t word_detection_timestamp, word_cnt_test;        /* ms tick of the last good word / good words so far          */
  uint32_t last_subframe_time, first_subframe_time;        /* ms tick of the bit edge that began the subframe            */
  uint16_t subframe_cnt;
  uint8_t  new_subframe_flag;
  uint8_t  subframe_data[GPS_NAV_SUBFRAME_LENGTH_BYTES];   /* bit n of the subframe = bit (n & 7) of byte n >> 3         */
} gps_nav_data_t;
/* Observation of a channel (gps_misc.h:135-139); written by the pseudorange step, which this library does not have. */
typedef struct {
  double pseudorange_m;
  double tow_s;
} gps_obs_data_t;

/* Broadcast ephemeris as the reference keeps it (gps_misc.h:141-182: RTKLIB's time and GPS ephemeris records inside
 * GNSS-SDRLIB's per-channel wrapper), filled by gps_nav_data_decode_subframe from subframes 1-3. */
#include <time.h>
typedef struct {
  time_t time;          /* s since the Unix epoch                                       */
  double sec;           /* fraction of a second                                         */
} gtime_t;
typedef struct {
  int     sat;                                             /* satellite (PRN)                                            */
  int     iode, iodc, sva, svh;                            /* issues of data, URA index, health (0 = ok)                 */
  int     week, code, flag;                                /* GPS week (roll-over resolved), codes on L2, L2 P data flag */
  gtime_t toe, toc, ttr;                                   /* ephemeris / clock reference epochs, transmission time      */
  double  A, e, i0, OMG0, omg, M0, deln, OMGd, idot;       /* Keplerian set: m, -, rad, rad/s                            */
  double  crc, crs, cuc, cus, cic, cis;                    /* harmonic corrections                                       */
  double  toes, fit;                                       /* toe as s of week, fit interval flag                        */
  double  f0, f1, f2, tgd[4];                              /* clock polynomial, group delay (tgd[0] = T_GD)              */
} eph_t;
typedef struct {
  eph_t    eph;
  int      ctype;
  double   tow_gpst;    /* time of week of the last decoded subframe's hand-over word   */
  int      week_gpst;
  int      cnt;         /* subframes 1-4 decoded                                        */
  int      cntth;
  int      update;
  int      prn;
  int      week_gst;
  uint16_t sub_cnt;     /* subframes handed to the decoder                              */
  uint8_t  received_mask;        /* bit n - 1: subframe n seen (cleared by the consumer) */
  uint8_t  received_mask_proc;   /* the same, never cleared                             */
} sdreph_t;

typedef struct {
  gps_acq_t      acq_data;
  gps_tracking_t tracking_data;
  gps_nav_data_t nav_data;
  gps_obs_data_t obs_data;
  sdreph_t       eph_data;
  uint8_t        prn;                    /* satellite PRN number, 1 .. 210                          */
  uint8_t        prn_code[PRN_LENGTH];   /* C/A chips as 0/1 bytes                                  */
} gps_ch_t;

/* --- the nine functions of gps_misc.h:195-216 ------------------------------------------------------------- */

/* The reference fills a 64 KiB popcount table here (gps_misc.c:19-25); the GPU has v_bcnt_u32_b32, so this only
 * opens the default context (and aborts loudly if there is no GPU). */
void gps_fill_summ_table(void);
/* channel->prn < 1: silent return (gps_misc.c:306-311) */
void gps_channell_prepare(gps_ch_t *channel);
/* dest: 1023 bytes; prn < 1: silent return (gps_misc.c:317-372; global in the reference though not in its header) */
void gps_generate_prn(uint8_t *dest, int prn);

int16_t gps_correlation8(uint16_t *prn_p, uint16_t *data_i, uint16_t *data_q, uint16_t offset);
void gps_correlation_iq(uint16_t *prn_p, uint16_t *data_i, uint16_t *data_q, uint16_t offset, int16_t *res_i,
                        int16_t *res_q);
uint16_t correlation_search(uint16_t *prn_p, uint16_t *data_i, uint16_t *data_q, uint16_t start_shift,
                            uint16_t stop_shift, uint16_t *aver_val, uint16_t *phase);
void gps_shift_to_zero_freq(uint8_t *signal_data, uint8_t *data_i, uint8_t *data_q, float freq_hz);
void gps_shift_to_zero_freq_track(gps_tracking_t *trk_channel, uint8_t *signal_data, uint8_t *data_i,
                                  uint8_t *data_q);
void gps_generate_prn_data2(gps_ch_t *channel, uint16_t *data, uint16_t offset_bits);
void gps_rewind_if_phase(gps_tracking_t *trk_channel, uint8_t steps);

/* The three scratch buffers of PM/GPS/common_ram.c:3-5 (1023 words + 1 pad, zero-initialised). */
extern uint16_t tmp_prn_data[GPS_DATA_WORDS_CNT];
extern uint16_t tmp_data_i[GPS_DATA_WORDS_CNT];
extern uint16_t tmp_data_q[GPS_DATA_WORDS_CNT];

/* --- the step-level entry points (PM/GPS/acquisition.h:7-12, PM/GPS/tracking.h:6) ------------------------------ */

/* One acquisition step on one captured millisecond for the GPS_SAT_CNT-channel table: ONE GPU launch computes the
 * (max, average, best phase) triplet every channel needs in its current state (frequency-search bin or code-phase
 * window, PRN + Doppler hint honoured), then the reference's voting / histogram logic runs on the host. */
void      acquisition_process(gps_ch_t *channel /* [GPS_SAT_CNT] */, uint8_t *data);
uint32_t *acquisition_get_hist(void);
void      acquisition_start_channel(gps_ch_t *channel);
void      acquisition_start_code_search_channel(gps_ch_t *channel);
void      acquisition_start_code_search3_channel(gps_ch_t *channel);
/* One tracking step (pre-tracking or E/P/L + DLL/PLL/FLL) of one channel; index = 0..3, or 0xFF for the idle slot. */
void      gps_tracking_process(gps_ch_t *channel, uint8_t *data, uint8_t index);

/* --- channel sequencing (PM/GPS/gps_master.h:7-14; gps_master.c:68-129,458-510) ------------------------------------- */
/* Starts acquisition channel by channel, opens the code-phase searches together, hands finished channels to tracking
 * (GPS_NEED_PRE_TRACK).  Call it after every acquisition_process / gps_tracking_process, as PM/main.c:157,167 does.
 * In the idle slot (index 0xFF) it calls gps_master_nav_handling, the pseudorange step below (weak: a host may bring its
 * own); the reference's UI duties are not reproduced, key_up_presed is a weak variable a host may set. */
void    gps_master_handling(gps_ch_t *channels /* [GPS_SAT_CNT] */, uint8_t index);
uint8_t gps_master_need_acq(void);
uint8_t gps_master_need_freq_search(gps_ch_t *channels);
uint8_t gps_master_is_code_search3(gps_ch_t *channels);
void    gps_master_reset_to_aqc_start(gps_ch_t *channels);
extern uint8_t key_up_presed;

/* --- the pseudorange step (PM/GPS/gps_master.c:159-430; csrc/gpsx_nav_master.cpp) -- PARITY UNPINNED -------------------
 * gps_master.c cannot be compiled in place (its include chain ends at CMSIS' core_cm4.h, which the reference tree does
 * not ship), so these follow the source text and are tested end to end on physics (tests/test_gpu_pvt_chain.py), not
 * against the reference's object code.
 * gps_master_nav_handling: subframe epochs of the four channels -> reference satellite (earliest stamp, declared
 * 68.802 ms away) -> obs_data.pseudorange_m / obs_data.tow_s of every channel from the stamps' differences and the
 * code phases averaged over the filter window (code_phase_fine_filt, accumulated by the DLL), code-phase wraps
 * compensated; then gps_master_calculate_pos: twice a second, once every channel holds subframes 1-3,
 * sdrobs2obsd + gps_pos_solve (call gps_pos_solve_init(channels) once before). */
void     gps_master_nav_handling(gps_ch_t *channels /* [GPS_SAT_CNT] */);
void     gps_master_final_pseudorange_calc(gps_ch_t *channels, uint32_t curr_tick_time, uint32_t ref_time_diff_ms,
                                           uint32_t ref_time_ms, uint8_t ref_idx);
uint16_t gps_master_filter_code_phase(gps_ch_t *channels, uint32_t curr_tick_time);
void     gps_master_code_phase_filter_reset(gps_ch_t *channels, uint32_t curr_tick_time);
void     gps_master_calculate_pos(gps_ch_t *channels);
/* not in the reference: the same step (without the position call) for ONE receiver of n_ch channels served in the reference's
 * multiplex (channel i of the table 4 i ms into the cycle).  1 = pseudoranges and reception times renewed, 0 = filter window
 * not ready, -1 = subframe epochs not there yet. */
int      gpsx_nav_pseudoranges(gps_ch_t *channels, int n_ch, uint32_t now_ms);
/* ... and for a receiver that is a subset of a large table: channels index[0 .. n), position k of the list served
 * k * slot_ms ms into the cycle (4 = the reference's multiplex; 0 = every channel on the same millisecond).  The epoch, the
 * averaging window and the wrap bookkeeping are the subset's own: other channels of the table neither hold it up nor restart it. */
int      gpsx_nav_pseudoranges_subset(gps_ch_t *channels, const int *index, int n, uint32_t now_ms, uint32_t slot_ms);

/* NOT in the reference: one tracking step of n_ch channels on the same millisecond, each channel served every
 * millisecond as in the single-satellite firmware's schedule (project_single_sat/main.c:96-109; index cycles 0..3).
 * All channels' pre-tracking searches and E/P/L correlators go out as one launch each, which is what keeps hundreds of
 * channels inside the 1 ms budget; per channel the result is what gps_tracking_process would give a receiver that had
 * only that channel.  Nav-bit synchronisation uses the built-in default (not the overridable hook). */
void      gps_tracking_process_batch(gps_ch_t *channel, int n_ch, uint8_t *data, uint8_t index);
/* threads (the caller included) the batched step spreads its per-channel host loops over from 2048 channels on: sized at the
 * first such call from the calling thread's CPUs and the container's CPU quota ($GPSX_STEP_THREADS overrides) */
int       gps_tracking_batch_workers(void);
int       gps_tracking_batch_last_workers(void);   /* ... and how many the last call used (1 when the host overrides a hook) */

/* Link-time dependencies of the step logic, as in the reference.  libgpsx provides WEAK defaults that a host program
 * overrides simply by defining the symbol:
 *   signal_capture_get_packet_cnt  1 ms tick (PM/signal_capture.c:35); default: a counter set by gpsx_compat_set_packet_cnt
 *   gps_nav_data_analyse_new_code  prompt-I hook (PM/GPS/nav_data.c:46-138); default: 20 ms bit-period synchronisation
 *                                  and bit integration only, ending in gps_nav_data_words_detection
 *   gps_nav_data_words_detection   word layer (PM/GPS/nav_data.c:257-351): preamble search, parity, polarity detection,
 *                                  subframe assembly, ending in gps_nav_data_decode_subframe once per subframe
 *   gps_nav_data_decode_subframe   ephemeris decode (PM/GPS/nav_data_decode.c:34-141): subframes 1-3 into
 *                                  channel->eph_data (clock, orbit, times), 4-5 time of week only; returns the subframe ID */
uint32_t signal_capture_get_packet_cnt(void);
void     gps_nav_data_analyse_new_code(gps_ch_t *channel, uint8_t index, int16_t new_i);
void     gps_nav_data_words_detection(gps_ch_t *channel, uint8_t new_bit);
uint8_t  gps_nav_data_decode_subframe(gps_ch_t *channel);
void     gpsx_compat_set_packet_cnt(uint32_t ticks_ms);

/* The capture interface of PM/signal_capture.h (weak, like the hooks above) on top of the engine's capture rings
 * (include/gpsx.h): a two-slot circular buffer standing for the DMA target, plus the copy buffer for long processing.
 * gpsx_compat_capture_push (not in the reference) is the DMA half / full transfer interrupt (PM/signal_capture.c:57-82):
 * it takes one 1 ms block, moves the ready pointer, advances the 1 ms tick, raises the "new data" flag -- and starts the
 * block's copy to the device, so that the step calls above, handed signal_capture_get_ready_buf() /
 * signal_capture_get_copy_buf(), read it from HBM without copying it again.  The interrupt deadline test of
 * signal_capture_handling (900 us) has no equivalent: blocks arrive when the host pushes them. */
void     signal_capture_init(void);
void     signal_capture_need_data_copy(void);
void     signal_capture_handling(void);
uint8_t  signal_capture_have_irq(void);
uint8_t  signal_capture_check_copied(void);
uint8_t *signal_capture_get_copy_buf(void);
uint8_t *signal_capture_get_ready_buf(void);
void     gpsx_compat_capture_push(const uint8_t *block);

/* --- position solution (PM/GPS/RTK/solving.h:32-37, rtk_common.h:49-58,104-108; host arithmetic in double precision) --
 * Single-point positioning as the reference runs it (an RTKLIB pntpos subset): satellite positions and clocks from the
 * broadcast ephemerides at the transmission times, then iterated weighted least squares over (x, y, z, c dt) with the
 * Klobuchar ionosphere (default coefficients when none were broadcast), the Saastamoinen troposphere at 70 % humidity and
 * the reference's variance model.  No part of it touches the GPU: it is the last consumer of the correlator hot path
 * (SURVEY.md 8(f) N4), four satellites twice a second. */
#define GPSX_PVT_MAXSAT 4                /* MAXSAT, rtk_common.h:42 */
typedef struct {                         /* obsd_t, rtk_common.h:49-58 */
  gtime_t       time;                    /* receiver sampling time (GPST) */
  unsigned char sat, rcv;                /* satellite (PRN) / receiver number */
  unsigned char SNR[1], LLI[1], code[1];
  double        L[1];                    /* carrier phase (cycles; unused) */
  double        P[1];                    /* pseudorange (m) */
  float         D[1];                    /* Doppler (Hz; unused) */
} obsd_t;
typedef struct {                         /* nav_t, rtk_common.h:104-108 */
  int    n;                              /* ephemerides in eph[] */
  eph_t *eph[GPSX_PVT_MAXSAT];
  double ion_gps[8];                     /* Klobuchar a0..a3, b0..b3; all zero: the 2004 default set */
} nav_t;
#define SOLQ_NONE   0
#define SOLQ_SINGLE 5
typedef struct {                         /* sol_t, solving.h:17-30 */
  gtime_t       time;                    /* solution time (GPST): reception time minus the receiver clock bias */
  double        rr[6];                   /* ECEF position (m); velocity entries are zeroed */
  float         qr[6];                   /* position covariance: xx, yy, zz, xy, yz, zx (m^2) */
  double        dtr[6];                  /* dtr[0] = receiver clock bias (s) */
  unsigned char type, stat, ns;          /* 0 = ECEF / SOLQ_* / satellites used */
  float         age, ratio;
} sol_t;
/* One complete solution: 1 = found (sol->stat = SOLQ_SINGLE), 0 = not found.  The iteration starts from sol->rr.
 * pntpos_iterative is the same computation (the reference slices it into < 1 ms pieces; here it finishes in its first
 * call and returns 1, or -1 / -2 on failure as the reference does). */
int  pntpos(const obsd_t *obs, int n, const nav_t *nav, sol_t *sol);
int  pntpos_iterative(const obsd_t *obs, int n, const nav_t *nav, sol_t *sol);
void ecef2pos(const double *r /* ECEF m */, double *pos /* lat, lon (rad), ellipsoidal height (m) */);
/* The receiver-level calls: bind the four channels' ephemerides, then call gps_pos_solve(obs) until solving_is_busy()
 * returns 0 (PM/GPS/gps_master.c:334-389).  Results: gps_sol, final_pos = {lat deg, lon deg, height m}. */
void    gps_pos_solve_init(gps_ch_t *channels /* [GPS_SAT_CNT] */);
void    gps_pos_solve(obsd_t *obs /* [GPS_SAT_CNT] */);
uint8_t solving_is_busy(void);
/* obs_data / eph_data / tracking_data of ns channels -> observation records (PM/GPS/RTK/rtklib_common.c:75-92) */
void    sdrobs2obsd(gps_ch_t *channels, int ns, obsd_t *out);
extern sol_t  gps_sol;
extern double final_pos[3];
extern obsd_t obsd[GPS_SAT_CNT];         /* the observation records gps_master_calculate_pos hands to the solver (gps_master.c:42) */
/* azimuth / elevation (deg) of the four satellites of the last solution (the reference's global `azel`) */
const double *gpsx_pvt_azel(void);

/* not in the reference: a tracking channel's loop state to / from the device-resident form of include/gpsx.h
 * (gpsx_track_loop: DLL / PLL / FLL and the bit synchroniser on the GPU).  Hand a channel over on a tick with
 * tick & 3 == 0 (the 4 ms group state of the bit synchroniser is not part of gps_ch_t). */
#include "gpsx.h"
void gpsx_loop_state_from_channel(const gps_ch_t *ch, uint32_t rng_seed, gpsx_loop_state_t *out);
void gpsx_loop_state_to_channel(const gpsx_loop_state_t *in, gps_ch_t *ch);

/* The word layer behind the device tracking loops: feeds the completed navigation bits of one gpsx_track_loop launch
 * (flags [n_blocks][n_ch], first block at tick first_tick) to gps_nav_data_words_detection with each bit's own tick and
 * keeps the records' bit-edge time current (and period_sync_ok_flag as of the channel's last completed bit or located edge:
 * the bit synchroniser itself lives on the device); returns how many channels changed their inv_polarity_flag (listed in
 * changed_opt up to max_changed -- hand them to gpsx_loop_set_polarity).  Host code.
 * ONLY flag bytes with bit 7 ("served": the schedule gave this channel this millisecond, set by gpsx_track_loop from library
 * version 110 on) are looked at; a flag buffer from an older producer, without bit 7, is skipped byte for byte -- check
 * gpsx_abi_check / gpsx_version when flags and word layer may come from different builds. */
int gps_tracking_words_batch(gps_ch_t *channel, int n_ch, const uint8_t *flags, int n_blocks, uint32_t first_tick,
                             int *changed_opt, int max_changed);

/* not in the reference: what a reboot does to the firmware -- the step logic's file-scope state (start flag, need-acquisition
 * flag, search buffers, slot statics, the pseudorange step's and the solver's memories, gps_sol, final_pos, obsd) back at its
 * initial values, for a host that starts a second receiver run in the same process.  The channel table is the caller's. */
void gpsx_compat_receiver_reset(void);

/* not in the reference: release the default context (optional, for leak checkers) */
void gpsx_compat_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* GPSX_COMPAT_H */
