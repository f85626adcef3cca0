/* include/gpsx.h -- C ABI of libgpsx.so, the MI355X (gfx950) GPS L1 C/A correlator engine.
 *
 * This is the "Tier 3" batched interface: plain pointers and sizes, no C++/torch types.  It
 * replaces, for many channels / hypotheses per call, what the reference firmware does one call at a time through
 * Firmware/project_main/GPS/gps_misc.h:195-216 (the per-call, symbol-compatible "Tier 1" mirror of that header is
 * include/gpsx_compat.h).  Everything below executes on the GPU; there is no CPU fallback -- without a usable HIP
 * device gpsx_create() fails with GPSX_ENODEV and nothing else can be called.
 *
 * Sample format (reference: PM/config.h:23-28, PM/signal_capture.c:9-11): 1 bit per sample (MAX2769 sign bit),
 * LSB first, fs = 16.368 MHz, IF = 4.092 MHz, one "block" = 1 ms = 16368 samples = 2046 bytes.
 *
 * Conventions: functions return 0 (GPSX_OK) or a negative errno-style code; *_dev variants take DEVICE pointers and
 * only enqueue work on the context's HIP stream (call gpsx_synchronize, or synchronize the stream you passed to
 * gpsx_create); the variants without _dev take HOST pointers and return when the results are in host memory.
 * A context is thread-compatible, not thread-safe (one stream, one set of scratch buffers).
 */
#ifndef GPSX_H
#define GPSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPSX_VERSION            110        /* 0.1.1: gpsx_loop_state_t is 120 bytes (96 up to 0.1.0), flag bit 7 = "served" is new and
                                            * gps_tracking_words_batch skips flag bytes without it, the product library reads no
                                            * $GPSX_ACQ_* / $GPSX_TRACK_WAVE_FROM knobs (lib/libgpsx_lab.so does).  A host built against an
                                            * older header must not run on this library: call gpsx_abi_check once at start-up. */
#define GPSX_BYTES_PER_MS       2046       /* PM/config.h:26-27: 16368 one-bit samples                    */
#define GPSX_PHASES_BYTE        2046       /* code-phase hypotheses at byte (0.5 chip) granularity         */
#define GPSX_PHASES_FINE        16368      /* byte offset x 8 replica bit shifts (PM/GPS/tracking.c:23)    */
#define GPSX_IF_HZ              4092000    /* PM/config.h:23                                               */
#define GPSX_MAX_PRN            210        /* PM/GPS/gps_misc.c:319-341                                    */

/* IF sample formats accepted wherever an entry point takes IF blocks (select with gpsx_set_if_format):
 *   GPSX_IF_1BIT    the reference's format: MAX2769 I1 (sign) only, 8 samples per byte LSB first, 2046 bytes per ms
 *   GPSX_IF_2BIT_SM MAX2769 I1/I0 sign + magnitude: 4 samples per byte LSB first, sample n in bits 2(n&3) (sign) and
 *                   2(n&3)+1 (magnitude), 4092 bytes per ms.  The kernels unpack the pairs in LDS and correlate on the
 *                   SIGN plane, i.e. exactly what the reference computes from the same front end (it wires I1 only,
 *                   PM/config.h:16); the magnitude plane is available through gpsx_if_unpack2. */
#define GPSX_IF_1BIT            0
#define GPSX_IF_2BIT_SM         1
#define GPSX_BYTES_PER_MS_2BIT  4092

#define GPSX_OK       0
#define GPSX_EIO     (-5)    /* a HIP runtime call failed (see gpsx_last_error)   */
#define GPSX_ENOMEM  (-12)
#define GPSX_ENODEV  (-19)   /* no usable gfx950 device                            */
#define GPSX_EINVAL  (-22)

typedef struct gpsx_ctx gpsx_ctx;

/* What correlation_search() (PM/GPS/gps_misc.c:155-191) returns, plus the un-divided sum. */
typedef struct {
  uint32_t max_val;  /* return value: largest correlation magnitude in the window                            */
  uint32_t phase;    /* *phase: first byte offset reaching it; 0 if nothing exceeded 0                       */
  uint32_t sum;      /* sum of magnitudes over the window                                                     */
  uint32_t avr;      /* *aver_val: sum / 2046 (the divisor is constant whatever the window)                   */
} gpsx_peak_t;

/* ---- context ------------------------------------------------------------------------------------------------ */

/* device: HIP device ordinal.  stream: a hipStream_t to enqueue on (e.g. torch's current stream), or NULL to let
 * the context create its own. */
int         gpsx_create(gpsx_ctx **ctx, int device, void *stream);
/* Receiver constants (the reference fixes them at compile time, PM/config.h:23-28).  sample_rate_hz is structural: the
 * 2046-byte millisecond and the 16368-phase grid are compiled into the kernels, any value but 16368000 is refused with
 * GPSX_EINVAL.  if_hz -- the centre of the acquisition grid's Doppler axis and the frequency a tracking channel's
 * if_freq_offset_hz is relative to -- is a run-time value of the context (front ends with another IF plan); entry points
 * that take absolute frequencies (gpsx_acq_jobs, gpsx_wipeoff) and the reference-named calls of gpsx_compat.h, whose
 * IF_FREQ_HZ is the reference's #define, do not look at it. */
typedef struct {
  uint32_t sample_rate_hz;   /* 16368000                                                                   */
  int32_t  if_hz;            /* GPSX_IF_HZ by default; 0 < if_hz < sample_rate_hz / 2                      */
} gpsx_config_t;
void        gpsx_config_default(gpsx_config_t *cfg);
int         gpsx_set_config(gpsx_ctx *ctx, const gpsx_config_t *cfg);
int         gpsx_get_config(const gpsx_ctx *ctx, gpsx_config_t *cfg);
void        gpsx_destroy(gpsx_ctx *ctx);
int         gpsx_synchronize(gpsx_ctx *ctx);
const char *gpsx_last_error(const gpsx_ctx *ctx);   /* text of the last failure on this context */
const char *gpsx_strerror(int code);
/* name of the dominant kernel the last gpsx_acq_grid* call launched (which form of the grid kernel the size picked) */
const char *gpsx_last_kernel(const gpsx_ctx *ctx);
int         gpsx_version(void);
/* The ABI handshake: pass the header's GPSX_VERSION and the sizes the host was COMPILED with,
 *   gpsx_abi_check(GPSX_VERSION, sizeof(gpsx_loop_state_t), sizeof(gpsx_acq_grid_t), sizeof(gpsx_peak_t))
 * GPSX_OK when the library was built from the same layout; GPSX_EINVAL when not (a host that strides d_state by another
 * sizeof(gpsx_loop_state_t) would corrupt device memory without any error).  Needs no context. */
int         gpsx_abi_check(int header_version, size_t sizeof_loop_state, size_t sizeof_acq_grid, size_t sizeof_peak);
/* name / CU count / clock of the device behind the context (for bench reports) */
int         gpsx_device_info(const gpsx_ctx *ctx, char *name, size_t name_len, int *compute_units, int *clock_khz);

/* Which hardware the fine acquisition grid (gpsx_acq_grid*, GPSX_PHASES_FINE) runs on.  Both give the same triplets and keys,
 * bit for bit (tests/test_gpu_parity.py; bench.py's letter_compliant leg compares the key tables of a whole 256-capture launch).
 *   GPSX_ACQ_PATH_MATRIX (default)  the exact MX-FP4 Toeplitz GEMM on the matrix cores (k_acq_mx)
 *   GPSX_ACQ_PATH_VECTOR            bit planes, v_and + v_bcnt polyphase recurrence, wave reductions on the vector ALU
 *                                   (k_acq_poly): no MFMA, about a sixth of the rate
 * The weighted two-bit extension (gpsx_acq_grid_weighted) follows the same switch: k_acq_mxw / k_acq_weighted. */
#define GPSX_ACQ_PATH_MATRIX 0
#define GPSX_ACQ_PATH_VECTOR 1
int gpsx_set_acq_path(gpsx_ctx *ctx, int path);
/* 0 for lib/libgpsx.so.  1 for lib/libgpsx_lab.so, the same sources built with -DGPSX_LAB: it additionally reads the
 * $GPSX_ACQ_* / $GPSX_TRACK_WAVE_FROM knobs that force kernel forms (tests of the alternative kernels, A/B timing). */
int gpsx_is_lab_build(void);

/* Sample format of the IF blocks passed to gpsx_acq_* and gpsx_track_* from now on (default GPSX_IF_1BIT). */
int gpsx_set_if_format(gpsx_ctx *ctx, int if_format);
/* Split n_blocks x 4092 bytes of GPSX_IF_2BIT_SM samples into the sign and magnitude bit planes, each n_blocks x 2046
 * bytes in the 1-bit layout (either output may be NULL).  Host buffers. */
int gpsx_if_unpack2(gpsx_ctx *ctx, const uint8_t *if_2bit, int n_blocks, uint8_t *sign_plane, uint8_t *magnitude_plane);

/* device memory + HIP-event timing on the context's stream (so a C caller needs no HIP headers) */
int gpsx_malloc(gpsx_ctx *ctx, void **dptr, size_t bytes);
/* page-locked host memory for the buffers a real-time host hands to the host-pointer entry points every millisecond
 * (capture blocks, channel states, accumulators): copies to and from it are plain DMA, without the runtime's staging of
 * pageable pages and its jitter */
int gpsx_host_alloc(gpsx_ctx *ctx, void **hptr, size_t bytes);
int gpsx_host_free(gpsx_ctx *ctx, void *hptr);
/* Pins the CALLING thread to the CPUs of the NUMA node the context's GPU hangs off (sysfs local_cpulist of its PCI
 * function), so that the per-millisecond buffers -- allocate them with gpsx_host_alloc afterwards: first touch -- and the
 * thread that fills them sit on the socket the DMA goes to.  On a two-socket host the far socket costs the E/P/L step of
 * 65536 channels 15-20 % and most of its jitter.  Returns GPSX_ENODEV (and changes nothing) when the topology is not
 * exposed; undo with sched_setaffinity. */
int gpsx_bind_thread_to_device(gpsx_ctx *ctx);
int gpsx_free(gpsx_ctx *ctx, void *dptr);
int gpsx_memcpy_h2d(gpsx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int gpsx_memcpy_d2h(gpsx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int gpsx_event_create(gpsx_ctx *ctx, void **event);
int gpsx_event_record(gpsx_ctx *ctx, void *event);
int gpsx_event_elapsed_ms(gpsx_ctx *ctx, void *start, void *stop, float *ms);  /* synchronizes on `stop` */
int gpsx_event_destroy(gpsx_ctx *ctx, void *event);

/* ---- IF ingest: the capture ring  (replaces the circular DMA buffer and its half/full-transfer interrupt,
 *      PM/signal_capture.c:14-24,57-82, and the file replay of PC_SpiLight: raw stream, 2046 bytes per ms in the 1-bit
 *      format, 4092 in GPSX_IF_2BIT_SM) -------------------------------------------------------------------------------
 * A ring of n_slots 1 ms blocks in pinned host memory with a mirror in HBM.  The producer fills the write slot and
 * commits it -- what the DMA interrupt does: the ready pointer moves on, the packet counter counts, and the block goes
 * to the device by an asynchronous copy from pinned memory, enqueued on the context's stream (the call does not wait).
 * Host entry points of this library (gpsx_acq_grid, gpsx_acq_jobs, gpsx_track_epl_batch above 32768 channels, and the
 * reference-named step calls on top of them) recognise a pointer into a committed part of the ring and read the HBM
 * mirror instead of copying the block again.  (The per-millisecond tracking step of up to 32768 channels is one captured
 * graph that stages its own 2 KB block next to the channel states: one launch beats a saved 2 KB copy.)  The format is the context's if_format at creation time. */
typedef struct gpsx_capture gpsx_capture;
int            gpsx_capture_create(gpsx_ctx *ctx, int n_slots, gpsx_capture **cap);      /* 1 <= n_slots <= 4096 */
void           gpsx_capture_destroy(gpsx_capture *cap);
/* the slot the next block goes into (block_bytes of pinned memory); valid until the next commit */
uint8_t       *gpsx_capture_write_slot(gpsx_capture *cap);
int            gpsx_capture_commit(gpsx_capture *cap);
/* convenience: copy one block into the write slot and commit it */
int            gpsx_capture_push(gpsx_capture *cap, const uint8_t *block);
/* newest committed block, host view (NULL before the first commit) */
const uint8_t *gpsx_capture_ready_buf(const gpsx_capture *cap);
/* the last n_blocks committed blocks, oldest first, contiguous in HBM (n_blocks <= min(n_slots, blocks committed));
 * a window that wraps around the ring is gathered into a side buffer on the stream, valid until the next call.  Pass
 * the pointer to the *_dev entry points. */
int            gpsx_capture_window_dev(gpsx_capture *cap, int n_blocks, const void **d_blocks);
uint32_t       gpsx_capture_packet_cnt(const gpsx_capture *cap);                          /* blocks committed so far */
size_t         gpsx_capture_block_bytes(const gpsx_capture *cap);
/* Replay a recorded raw IF file through the ring: blocks first_block .. (at most max_blocks, < 0 = to the end) are read
 * into the write slot and committed one by one; after each commit on_block(user, cap, index) runs (may be NULL) and a
 * non-zero return stops the replay.  Returns the number of blocks committed, or a negative GPSX_E* code. */
typedef int (*gpsx_capture_block_fn)(void *user, gpsx_capture *cap, long block_index);
long           gpsx_capture_replay_file(gpsx_capture *cap, const char *path, long first_block, long max_blocks,
                                        gpsx_capture_block_fn on_block, void *user);

/* ---- K1: C/A Gold codes  (replaces gps_generate_prn / gps_channell_prepare, PM/GPS/gps_misc.c:306-372) -------- */

/* chips_out: n_prn x 1023 bytes of 0/1.  prn must be 1..210 (the reference silently ignores prn < 1; this
 * interface validates instead and returns GPSX_EINVAL). */
int gpsx_ca_codes(gpsx_ctx *ctx, const uint8_t *prns, int n_prn, uint8_t *chips_out);

/* ---- K2+K3+K4: acquisition grid  (replaces the data-parallel prefix of acquisition_freq_search /
 *      acquisition_code_phase_search, PM/GPS/acquisition.c:196-312: gps_generate_prn_data2 + gps_shift_to_zero_freq
 *      + correlation_search per (channel, Doppler bin, ms), here for every PRN x Doppler x phase in one launch) ---- */

typedef struct {
  int32_t        n_search;             /* independent searches (e.g. consecutive capture instants)              */
  int32_t        n_ms;                 /* blocks summed non-coherently per search; 1 = the reference            */
  int32_t        search_stride_blocks; /* search s starts at block s * stride                                   */
  int32_t        n_prn;
  const uint8_t *prns;                 /* HOST pointer, n_prn PRN numbers                                       */
  int32_t        dopp_min_hz;          /* Doppler bin d is dopp_min_hz + d * dopp_step_hz  (acquisition.c:285)  */
  int32_t        dopp_step_hz;
  int32_t        n_dopp;
  int32_t        phase_mode;           /* GPSX_PHASES_BYTE (replica shift 0 only) or GPSX_PHASES_FINE (0..7)   */
  int32_t        win_start, win_stop;  /* byte-offset window [start, stop); 0, 2046 for a full search           */
  int32_t        shard_index;          /* multi-GPU: of the U = n_search * n_dopp * ceil(n_prn / 8) work units      */
  int32_t        shard_count;          /*   u = (search * n_dopp + dopp) * ceil(n_prn / 8) + prn_idx / 8 this process  */
                                       /*   computes the run [index * U / count, (index + 1) * U / count); 0/1: all    */
} gpsx_acq_grid_t;

/* number of replica bit shifts a phase_mode implies (1 or 8) */
int gpsx_acq_bits(int phase_mode);

/* Sizes (in elements) of the result arrays for a descriptor: peaks[n_search][n_prn][n_dopp][n_bits],
 * keys[n_search][n_prn][n_dopp]. */
size_t gpsx_acq_peaks_count(const gpsx_acq_grid_t *g);
size_t gpsx_acq_keys_count(const gpsx_acq_grid_t *g);

/* Enqueue one grid.  d_if_blocks: n_blocks x 2046 bytes, blocks contiguous.  d_peaks: gpsx_acq_peaks_count()
 * entries; entries of (search, PRN, Doppler) units owned by other shards are written as zero.  d_keys (may be NULL):
 * one packed int64 per (search, PRN, Doppler): (max_val << 14) | (16383 - fine_phase), fine_phase = 8 * phase + bit
 * shift, maximised over the bit shifts -- zero for units of other shards, so that ONE all-reduce(MAX) over the ranks
 * yields every unit's peak, ties resolved to the lowest fine phase like correlation_search's strict '>'.
 * Optional debug/inspection outputs (NULL to skip), indexed like d_peaks with one more trailing axis:
 *   d_per_ms [..][n_ms]   the triplet the reference would have produced for each single block
 *   d_energy [..][2046]   accumulated magnitude per byte offset (0 outside the window)
 *   d_cnt    [..][2046][2] raw popcounts cnt_i, cnt_q of the LAST block (what gps_mult_and_summ returns) */
int gpsx_acq_grid_dev(gpsx_ctx *ctx, const gpsx_acq_grid_t *g, const void *d_if_blocks, int n_blocks,
                      gpsx_peak_t *d_peaks, int64_t *d_keys, gpsx_peak_t *d_per_ms, uint32_t *d_energy,
                      uint16_t *d_cnt);

/* Host-buffer convenience: copies the blocks in, runs, copies peaks (and keys if non-NULL) out. */
int gpsx_acq_grid(gpsx_ctx *ctx, const gpsx_acq_grid_t *g, const uint8_t *if_blocks, int n_blocks,
                  gpsx_peak_t *peaks, int64_t *keys);
/* The same, enqueued only: copy in, sweep and copies out are put on the context's stream and the call returns; the
 * host buffers (pinned memory, or the copies are not asynchronous) belong to the engine until gpsx_synchronize(ctx).
 * Two contexts used alternately overlap one call's PCIe transfers with the other's sweep -- how a host that streams
 * captures through the engine reaches the HBM-resident rate (bench.py `pcie_inclusive`). */
int gpsx_acq_grid_async(gpsx_ctx *ctx, const gpsx_acq_grid_t *g, const uint8_t *if_blocks, int n_blocks,
                        gpsx_peak_t *peaks, int64_t *keys);

/* Explicit job list: one search per job, each with its own PRN, carrier frequency, replica shift and window --
 * what acquisition_process() needs for the reference's 4-channel table with per-channel Doppler hints
 * (PM/GPS/acquisition.c:51-57,72-79) and what pre-tracking needs (PM/GPS/tracking.c:398-450). */
typedef struct {
  int32_t  block;        /* first block of the search                              */
  int32_t  n_ms;
  int32_t  prn;
  float    freq_hz;      /* IF + Doppler, as passed to gps_shift_to_zero_freq      */
  int32_t  offset_bits;  /* replica shift 0..15 (gps_generate_prn_data2)           */
  int32_t  win_start, win_stop;
} gpsx_acq_job_t;

int gpsx_acq_jobs(gpsx_ctx *ctx, const gpsx_acq_job_t *jobs, int n_jobs, const uint8_t *if_blocks, int n_blocks,
                  gpsx_peak_t *peaks /* n_jobs */, uint32_t *energy_opt /* n_jobs x 2046 or NULL */);

/* unpack a key produced by gpsx_acq_grid* */
static inline uint32_t gpsx_key_energy(int64_t key) { return (uint32_t)(key >> 14); }
static inline uint32_t gpsx_key_fine_phase(int64_t key) { return 16383u - (uint32_t)(key & 16383); }

/* ---- the sharded sweep inside ONE process (a C host has no torch.distributed): a group of contexts, one per GPU,
 *      joined by RCCL communicators (ncclCommInitAll; librccl is loaded when the first group is created).
 *      gpsx_acq_grid_sharded = what each rank of `bench.py --gpus N` does, for all the group's devices at once:
 *      context i sweeps the run of grid units [i * U / n, (i + 1) * U / n) (g's own shard fields are ignored) on its own copy of the
 *      captures, then ONE all-reduce(MAX) of the packed keys leaves the merged table in every d_keys[i].  Everything is
 *      enqueued on the contexts' streams; synchronize the contexts (or read through gpsx_memcpy_d2h) before using it. */
typedef struct gpsx_group gpsx_group;
int  gpsx_group_create(gpsx_ctx *const *ctxs, int n, gpsx_group **group);   /* n >= 1 contexts on n DIFFERENT devices */
void gpsx_group_destroy(gpsx_group *group);
int  gpsx_acq_grid_sharded(gpsx_group *group, const gpsx_acq_grid_t *g, const void *const *d_if_blocks, int n_blocks,
                           gpsx_peak_t *const *d_peaks, int64_t *const *d_keys);

/* ---- EXTENSION, not in the reference: the acquisition grid on WEIGHTED two-bit samples ---------------------------------
 * The reference wires the MAX2769's sign bit only (PM/config.h:16), and everything else in this header computes what the
 * reference computes -- from GPSX_IF_2BIT_SM captures too, whose magnitude bit it ignores.  This entry point is the one place
 * that uses both bits; it has its own CPU restatement to be tested against (tests/test_gpu_weighted.py) and cannot change a one-bit result.
 *   sample value   v[n] = (sign ? +1 : -1) * (magnitude ? 3 : 1)       GPSX_WEIGHTS_SIGN_MAGNITUDE
 *                  v[n] = (sign ? +1 : -1)                              GPSX_WEIGHTS_SIGN_ONLY (the same correlator on the sign
 *                                                                       plane: the point a processing gain is measured from)
 *   carrier        the reference's NCO (gps_shift_to_zero_freq, PM/GPS/gps_misc.c:211-240: the accumulator's quadrant picks the
 *                  Fs/4 pattern per 32-sample word, phase 0 at the block's start); the sixteen samples it never mixes: weight 0
 *   replica        the C/A code circularly at fine phase tau: chip ((n - tau) mod 16368) / 16
 *   result         per (search, PRN, Doppler bin): max over the 16368 phases of floor(sqrt(I^2 + Q^2)) (exact integers), the first
 *                  phase reaching it (0 .. 16367), the sum over the phases and sum / 16368 -- peaks[n_search][n_prn][n_dopp].
 * One 1 ms block per search (search s reads block s * search_stride_blocks), 4092-byte blocks whatever the context's format.
 * Runs on the matrix cores (k_acq_mxw: the Toeplitz GEMM of the sign-only grid on sums of sixteen weighted samples, MX-FP4 operands,
 * exact; about 9 x 10^11 hypotheses/s) or, under GPSX_ACQ_PATH_VECTOR, on the vector ALU (k_acq_weighted: v_dot4_i32_i8, about
 * 4 x 10^10): the same records, bit for bit. */
#define GPSX_WEIGHTS_SIGN_ONLY      0
#define GPSX_WEIGHTS_SIGN_MAGNITUDE 1
typedef struct {
  int32_t        n_search, search_stride_blocks;
  int32_t        n_prn;
  const uint8_t *prns;                 /* HOST pointer, n_prn PRN numbers 1 .. 210 */
  int32_t        dopp_min_hz, dopp_step_hz, n_dopp;
  int32_t        weights;              /* GPSX_WEIGHTS_* */
} gpsx_acq_weighted_t;
int gpsx_acq_grid_weighted_dev(gpsx_ctx *ctx, const gpsx_acq_weighted_t *g, const void *d_if_blocks_2bit, int n_blocks,
                               gpsx_peak_t *d_peaks);
int gpsx_acq_grid_weighted(gpsx_ctx *ctx, const gpsx_acq_weighted_t *g, const uint8_t *if_blocks_2bit, int n_blocks,
                           gpsx_peak_t *peaks);

/* ---- K2+K3+K5: Early/Prompt/Late tracking correlators  (replaces the correlator part of
 *      gps_tracking_data_process, PM/GPS/tracking.c:115-138, for n_ch channels at once) ------------------------- */

typedef struct {
  int32_t  prn;
  float    code_phase_fine;    /* gps_tracking_t.code_phase_fine, samples 0..16368       */
  float    if_freq_offset_hz;  /* gps_tracking_t.if_freq_offset_hz                       */
  uint32_t if_freq_accum;      /* gps_tracking_t.if_freq_accum: read, advanced, written  */
} gpsx_trk_state_t;

/* if_block: the current 1 ms block (2046 bytes, shared by all channels).  iq_out: n_ch x {IE,QE,IP,QP,IL,QL}.
 * PRNs are validated BY THE KERNELS, not before the launch (a host loop over the states costs a sixth of the millisecond at
 * 400 000 channels): a channel whose prn is outside 1..210 is correlated against the empty code, every channel's
 * if_freq_accum and accumulators ARE written, and the call then returns GPSX_EINVAL.  gpsx_track_epl_batch reports it
 * itself (it waits for its kernels); gpsx_track_epl_batch_dev only enqueues: its report comes from the next
 * gpsx_synchronize() on the context (a separate flag: step calls in between neither consume nor clear it). */
int gpsx_track_epl_batch(gpsx_ctx *ctx, const uint8_t *if_block, gpsx_trk_state_t *st, int n_ch, int16_t *iq_out);
int gpsx_track_epl_batch_dev(gpsx_ctx *ctx, const void *d_if_block, gpsx_trk_state_t *d_st, int n_ch,
                             int16_t *d_iq_out);
/* The same step for a host that has work of its own per channel (the reference's DLL / PLL / FLL after the correlators):
 * the channels go through in n_chunks (1..16) pieces on the copy / correlate / copy pipeline, and on_chunk(user, first, n) is
 * called ON THE CALLING THREAD as soon as st[first .. first + n) and iq_out of those channels are in the caller's arrays --
 * while the GPU works on the next pieces.  Page-locked arrays (gpsx_host_alloc) make the copies asynchronous.  Returns after
 * the last callback; the PRN verdict is the whole step's.  The callback must not call into the SAME context (its arena and
 * side streams are in use by the pieces still in flight): every gpsx_* entry point on it returns GPSX_EINVAL while a callback
 * runs; other contexts are free. */
typedef void (*gpsx_track_chunk_fn)(void *user, int first_channel, int n_channels);
int gpsx_track_epl_batch_chunked(gpsx_ctx *ctx, const uint8_t *if_block, gpsx_trk_state_t *st, int n_ch, int16_t *iq_out,
                                 int n_chunks, gpsx_track_chunk_fn on_chunk, void *user);

/* ---- the tracking LOOPS on the device: correlators + DLL / PLL / FLL + false-lock check + SNR + 20 ms bit synchroniser,
 *      K milliseconds per launch, channel state resident in HBM  (gps_tracking_data_process, PM/GPS/tracking.c:92-170,
 *      with gps_tracking_dll / _pll / _fll / _pll_check :175-393 and gps_nav_data_analyse_new_code, PM/GPS/nav_data.c:46-253)
 *
 * The bit-exact mode of this library runs those float loops on the host behind every correlator launch
 * (gps_tracking_process / gps_tracking_process_batch, include/gpsx_compat.h).  This is the other mode: a receiver that tracks
 * tens of thousands of channels keeps each channel's loop state in a gpsx_loop_state_t in device memory and hands the
 * engine K consecutive 1 ms blocks; one kernel runs, per channel and millisecond, the E/P/L correlators (bit-exact, as
 * above) and then the reference's loop arithmetic in its own order of float operations.  Nothing crosses the link per
 * millisecond but the IF block in and ONE byte per channel out:
 *     bit 0  prompt in-phase accumulator > 0          bit 1  a navigation bit was completed this millisecond ...
 *     bit 2  ... and this is its value                bit 3  20 ms bit period synchronised (after this millisecond)
 *     bit 4  the false-lock detector moved the carrier (tracking.c:309-326)
 *     bit 5  the bit edge inside the 20 ms grid was located (nav_data.c:145-218): accurate_swap_time =
 *            (tick - 3 + (bit 6 ? 2 : 1)) % 20 -- what the subframe time stamp is made of
 *     bit 7  the channel was served this millisecond (always set under GPSX_SCHED_EVERY_MS; under GPSX_SCHED_MUX17 the
 *            bytes of a channel's unserved milliseconds are 0)
 * -- what the word layer (gps_nav_data_words_detection, one call per completed bit; gps_tracking_words_batch in
 * include/gpsx_compat.h does it for a whole launch) needs.
 * Serving schedule (gpsx_loop_set_schedule), both the reference's own:
 *   GPSX_SCHED_EVERY_MS (default)  every channel every millisecond, index = tick & 3 (project_single_sat/main.c:96-109, as
 *       gps_tracking_process_batch).  The 4 ms groups then sit still on the 20 ms bit grid, and the reference's bit-edge
 *       locator -- it looks at groups with the sign change between index 1 and 2 only, nav_data.c:131-137 -- resolves the
 *       edge for the channels whose bit edges happen to fall there: one in four.
 *   GPSX_SCHED_MUX17  project_main's receiver: four channels share one correlator in a 17 ms cycle (PM/main.c:139-152).
 *       Channel c is slot (c & 3) of receiver (c >> 2); it is served on the ticks t with (t % 17) / 4 == slot, with
 *       index = (t % 17) % 4; t % 17 == 16 is the idle millisecond (the reference's navigation slot).  The milliseconds a
 *       channel was not served are made up in its carrier NCO when it is served again (gps_rewind_if_phase with the elapsed
 *       ticks - 1, PM/GPS/tracking.c:102-113, from prev_track_timestamp); code phase and loop filters stand still, as in the
 *       reference.  Every 17 ms the group start moves by 17 mod 20 over the bit grid, so every channel gets its bit edge
 *       located, hence accurate_swap_time, hence subframe time stamps and pseudoranges -- the complete receiver on the
 *       device loop (tests/test_gpu_track_mux.py: the reference's multiplexed traces, tests/test_gpu_pvt_chain.py: IF
 *       samples to position).  Hand channels over (gpsx_loop_state_from_channel) on a tick with t % 17 == 0.
 * Arithmetic against the host mode (= the reference's C on the host CPU): the float arctangents are the C library's own
 * algorithm restated operation by operation (csrc/gpsx_libm.hpp, compared with glibc bit for bit on the CPU), so the loops'
 * floats come out identical -- observed on every committed reference trace and on 26 000 channels from random states: every
 * byte of every record.  What remains different by construction: the double-precision atan2 of the PLL's IP <= 0 branch is the
 * device's (its result rounded to float differs from glibc's in about one argument pair in 2^29), and snr_value's logarithm
 * (a display value; the record gets the host's log10f of the sums the device latched, gpsx_loop_state_to_channel).  The stated
 * tolerance of SURVEY.md 8(c) -- |d code_phase_fine| <= 0.01 sample, |d if_freq_offset_hz| <= 0.5 Hz -- stays the tests'
 * fallback bar.  The false-lock jump draws by default from a per-channel xorshift32 (`rng`, never 0) instead of libc's
 * process-global rand(); gpsx_loop_set_draws(GPSX_DRAWS_LIBC) gives the reference's draws in the reference's order.
 * Data polarity (gpsx_loop_set_word_sync): the reference's word layer flips inv_polarity_flag when it has seen two inverted
 * preambles, and the very next millisecond's vote and sign-change detection use the new value (nav_data.c:60-66, 284-291).
 *   GPSX_WORDSYNC_DEVICE (default)  the kernel runs the polarity-deciding part of the word layer itself (preamble hunt, word
 *       collection, parity, the two-subframe timeout: 12 bytes of state) on every completed bit, so the flag changes on the
 *       millisecond the reference changes it on, whatever the launch length.  The host's word layer
 *       (gps_tracking_words_batch) sees the same bits at the same ticks and takes the same decisions; it still lists the
 *       channels whose flag changed, and handing them to gpsx_loop_set_polarity is harmless (the device already has the value).
 *   GPSX_WORDSYNC_HOST  the device never touches the flag: a host with its own word layer (one that overrides
 *       gps_nav_data_words_detection) writes it through gpsx_loop_set_polarity; it then takes effect at the next launch. */
typedef struct {
  int32_t  prn;                                /* 1 .. 210 */
  float    code_phase_fine;                    /* gps_tracking_t, same names, same meaning (include/gpsx_compat.h) */
  float    if_freq_offset_hz;
  uint32_t if_freq_accum;
  float    dll_code_err, pll_code_err, fll_err;
  int16_t  fll_old_i, fll_old_q;
  int16_t  pll_check_buf[4];
  uint16_t pll_bad_state_master_cnt;
  uint8_t  pll_bad_state_cnt;
  uint8_t  period_sync_ok_flag;                /* gps_nav_data_t: 20 ms bit period found (selects the PLL's gain set) */
  int16_t  found_freq_offset_hz;               /* gps_acq_t: centre of the false-lock jump */
  uint16_t reseed_count;                       /* false-lock jumps so far */
  uint32_t rng;                                /* xorshift32 state of those jumps; must not be 0 */
  uint32_t i_part_summ, q_part_summ;           /* SNR estimator */
  float    snr_value;
  uint16_t snr_summ_cnt;
  uint16_t code_filt_cnt;                      /* code-phase averaging window of the pseudorange step */
  float    code_phase_fine_filt;
  uint32_t old_swap_time;                      /* gps_nav_data_t: bit synchroniser */
  uint32_t slot_start_ticks;                   /* tick of index 0 of the current 4 ms group */
  int16_t  slot_ip[4];                         /* prompt I of the group so far */
  uint8_t  slot_bits;                          /* bit i = sign bit of index i of the group */
  uint8_t  right_period_cnt, old_reminder, accurate_swap_time, accurate_swap_ok;
  uint8_t  last_bit_pos_cnt, last_bit_neg_cnt;
  uint8_t  inv_polarity_flag;                  /* data polarity inverted: decided by the device's word sync (below), or written
                                                  through gpsx_loop_set_polarity under GPSX_WORDSYNC_HOST */
  uint32_t prev_track_timestamp;               /* gps_tracking_t: tick the channel was last served on */
  uint32_t snr_i_latch, snr_q_latch;           /* the sums snr_value was last made of (q = 0: snr_value stands as it is) */
  uint32_t word_buf;                           /* gps_nav_data_t.word_buf: bit i = word_buf[i]  (the device's word sync) */
  uint32_t word_detection_timestamp;
  uint8_t  word_cnt, word_bit_cnt, inv_preabmle_cnt;
  uint8_t  word_flags;                         /* bit 0 old_D29, bit 1 old_D30, bit 2 polarity_found */
} gpsx_loop_state_t;                           /* 120 bytes */

typedef struct {                               /* optional per-millisecond record, for tests and inspection */
  int16_t  iq[6];                              /* IE, QE, IP, QP, IL, QL of this millisecond */
  float    code_phase_fine, if_freq_offset_hz; /* AFTER this millisecond's loop updates */
  uint32_t if_freq_accum;
} gpsx_loop_trace_t;                           /* 24 bytes */

/* d_if_blocks: n_blocks consecutive 1 ms blocks in device memory (the context's IF format); d_state: n_ch states in device
 * memory, read, advanced by n_blocks milliseconds, written; first_tick_ms: the millisecond tick of the first block;
 * d_flags: [n_blocks][n_ch] bytes (above); d_trace_opt: NULL or [n_blocks][n_ch] records.  Enqueues and returns. */
int gpsx_track_loop_dev(gpsx_ctx *ctx, const void *d_if_blocks, int n_blocks, gpsx_loop_state_t *d_state, int n_ch,
                        uint32_t first_tick_ms, uint8_t *d_flags, gpsx_loop_trace_t *d_trace_opt);
/* The same with the blocks and the flag bytes in host memory (page-locked: gpsx_host_alloc): copies the blocks in,
 * runs, copies the flags (and trace records, if asked for) out, waits.  The states stay on the device. */
int gpsx_track_loop(gpsx_ctx *ctx, const uint8_t *if_blocks, int n_blocks, gpsx_loop_state_t *d_state, int n_ch,
                    uint32_t first_tick_ms, uint8_t *flags, gpsx_loop_trace_t *trace_opt);

/* Serving schedule of this context's gpsx_track_loop* launches from now on (above). */
#define GPSX_SCHED_EVERY_MS 0
#define GPSX_SCHED_MUX17    1
int gpsx_loop_set_schedule(gpsx_ctx *ctx, int schedule);

/* Where the false-lock detector's random carrier jump (PM/GPS/tracking.c:309-326) draws from.
 *   GPSX_DRAWS_XORSHIFT (default)  the channel's own xorshift32 (`rng`): any number of channels, nothing leaves the device.
 *   GPSX_DRAWS_LIBC  the reference's: libc's rand(), drawn on the host in the order a single-threaded loop over the
 *       milliseconds and channels makes the draws -- so a receiver that seeds as the reference does jumps where the reference
 *       jumps (tests/test_gpu_track_loop.py: the 64-channel trace with its 21 jumps).  A launch whose channels want to jump is
 *       run twice for those channels (they report, the host draws, they are replayed from the launch's input state); launches
 *       are limited to 320 ms and gpsx_track_loop_dev WAITS for its kernels in this mode.  rand() is process-global: the
 *       caller seeds it (and see INTEGRATION.md on the ROCm runtime's own draws when code objects load).
 *       A channel has ONE candidate slot per launch (its detector needs 324 ms to fill again, a launch is at most 320): should
 *       a replay not settle in four passes, or a channel report twice inside one launch, the call returns GPSX_EIO and
 *       d_state, flags and trace of that call are UNDEFINED -- restore the states from the caller's copy or drop the channels. */
#define GPSX_DRAWS_XORSHIFT 0
#define GPSX_DRAWS_LIBC     1
int gpsx_loop_set_draws(gpsx_ctx *ctx, int draws);

#define GPSX_WORDSYNC_DEVICE 0
#define GPSX_WORDSYNC_HOST   1
int gpsx_loop_set_word_sync(gpsx_ctx *ctx, int owner);

/* The host's word layer found (or gave up) inverted data polarity on n channels: d_state[channels[i]].inv_polarity_flag =
 * values[i] (host arrays; enqueued on the context's stream in front of the next launch). */
int gpsx_loop_set_polarity(gpsx_ctx *ctx, gpsx_loop_state_t *d_state, const int *channels, const uint8_t *values, int n);

/* The pseudorange step consumed the code-phase averaging window (gps_master_code_phase_filter_reset, gps_master.c:383-389):
 * code_phase_fine_filt = 0, code_filt_cnt = 0 for all n_ch device states (enqueued on the context's stream). */
int gpsx_loop_reset_code_filter(gpsx_ctx *ctx, gpsx_loop_state_t *d_state, int n_ch);

/* ---- per-call primitives on caller buffers (the device work behind include/gpsx_compat.h) --------------------- */

/* gps_shift_to_zero_freq(_track): *accum is the NCO accumulator in/out (0 for the stateless call).  Writes bytes
 * 0..2043 of data_i / data_q only (PM/GPS/gps_misc.c:229: the last 16 samples are never mixed). */
int gpsx_wipeoff(gpsx_ctx *ctx, const uint8_t *signal, float freq_hz, uint32_t *accum, uint8_t *data_i,
                 uint8_t *data_q);
/* gps_generate_prn_data2: chips = 1023 bytes 0/1; out = 1024 words, word 1023 is OR-ed with the spill. */
int gpsx_replica(gpsx_ctx *ctx, const uint8_t *chips, unsigned offset_bits, uint16_t *out);
/* gps_mult_and_summ + gps_correlation8 + gps_correlation_iq for a list of byte offsets (each 0..2046) on arbitrary
 * 2046-byte buffers.  Any of cnt_i/cnt_q (raw popcounts), corr8 may be NULL. */
int gpsx_corr_offsets(gpsx_ctx *ctx, const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q,
                      const uint16_t *offsets, int n, uint16_t *cnt_i, uint16_t *cnt_q, int16_t *corr8);
/* the magnitude stage of gps_correlation8 alone (PM/GPS/gps_misc.c:106-118) for n raw popcount pairs: centre by 8184,
 * clip negatives to zero, (int16) sqrtf((float)(I*I) + (float)(Q*Q)) */
int gpsx_mag8(gpsx_ctx *ctx, const uint16_t *cnt_i, const uint16_t *cnt_q, int n, int16_t *out);
/* correlation_search on arbitrary buffers */
int gpsx_corr_search(gpsx_ctx *ctx, const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q,
                     unsigned start_shift, unsigned stop_shift, gpsx_peak_t *peak);
/* gps_rewind_if_phase (PM/GPS/gps_misc.c:196-204) for n channel states (device arithmetic, same rounding) */
int gpsx_rewind(gpsx_ctx *ctx, gpsx_trk_state_t *st, int n_ch, const uint8_t *steps);

#ifdef __cplusplus
}
#endif
#endif /* GPSX_H */
