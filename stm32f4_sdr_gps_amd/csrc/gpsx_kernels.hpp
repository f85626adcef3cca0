// gpsx_kernels.hpp -- host-visible launch interface of the HIP kernels (implemented in k_*.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/gpsx.h"

namespace gpsx {

constexpr int kAcqGroup = 8;      // PRNs per accumulator set (one main-loop pass) in the grid kernel
constexpr int kSuperGroups = 1;   // groups per sharding unit: a unit is (search, Doppler bin, 8-PRN group) -- 84 units per
                                  // 32 PRN x 21 Doppler search, SURVEY.md 8(e).  Unit index u = (search * n_dopp + dopp) *
                                  // n_groups + group (group fastest); shard r of W owns the contiguous run
                                  // [r * U / W, (r + 1) * U / W) -- balanced to one unit, and the four groups of a
                                  // (search, Doppler) pair stay on one GPU (the matrix-core kernel sweeps 32 PRNs at once)
constexpr int kCodeWords = 256;   // 4-chip code words per PRN (1023 chips + 1 masked pad)
constexpr int kMaxMs = 128;       // keeps (energy << 11 | phase) and the window sum inside 32 bits
constexpr int kAlgoSad = 0;       // main loop: v_msad_u8 on 8-bit block sums, 4 chips per instruction
constexpr int kAlgoDot8 = 1;      // main loop: v_dot8_u32_u4 on 4-bit block sums, 8 chips per instruction
constexpr int kAlgoPoly = 2;      // fine grid only: polyphase recurrence across the 16 sample offsets, AND + popcount
constexpr int kAlgoMx = 4;        // the same recurrence as a Toeplitz GEMM on the matrix cores, MX-FP4: the default at every
                                  // launch size, for fine grids and single-block byte-phase grids without inspection outputs

// One search = one workgroup pass: `count` (<= group size) consecutive code-table slots, one carrier frequency,
// one replica bit shift, n_ms consecutive blocks.
struct AcqJobRec {
  int32_t block;        // first IF block
  int32_t slot;         // first code-table slot
  float freq_hz;        // IF + Doppler
  int32_t offset_bits;  // replica shift b
  int32_t win_start, win_stop;
  int32_t out_index;    // index of slot 0's result
};

struct AcqParams {
  // arithmetic job decode (grid mode, jobs == nullptr)
  int32_t n_ms;
  int32_t search_stride_blocks;
  int32_t n_prn, n_groups, n_dopp, dopp_min_hz, dopp_step_hz;
  int32_t n_bits;
  int32_t unit_lo, unit_hi;   // this shard's run of sharding units
  int32_t win_start, win_stop;
  int32_t if_format;        // GPSX_IF_1BIT / GPSX_IF_2BIT_SM
  int32_t if_hz;            // gpsx_config_t.if_hz: centre of the Doppler axis
  // set by launch_acq_mx for its own forms:
  int32_t split_segs;       //   k_acq_mx<5>: workgroups per cluster (2, 4 or 8)
  int32_t n_clusters;       //   k_acq_mx<4>: clusters of the launch (one persistent workgroup per CU walks them)
  uint64_t n_planes;        //   k_acq_mx<5>: entries per result plane (packed keys [0, n), sums [n, 2 n) behind `energy`)
  int32_t experiment;       // ablations of k_acq_mx for timing (results are then wrong); always 0 unless the library was built
                            // with -DGPSX_MX_ABLATIONS, which alone makes gpsx_api.hip read $GPSX_MX_EXPERIMENT
  // explicit job list (job mode)
  const AcqJobRec *jobs;
  // outputs (optional ones may be null)
  gpsx_peak_t *peaks;
  int64_t *keys;            // k_acq_mx: packed keys written next to the triplets (unsharded launches; null: k_acq_keys follows)
  gpsx_peak_t *per_ms;
  uint32_t *energy;
  uint16_t *cnt;
};

// K1: Gold codes + derived tables for `n_slots` code-table slots (padded slots have prn 0 -> all-zero tables).
//   chips    [n_slots][1024]  0/1 bytes
//   chipbits [n_slots][32]    packed, bit (i & 31) of word (i >> 5) = chip i
//   cw       [n_slots/group][256][group]  4 chips per word as SAD reference bytes: 17 (chip 1), 1 (chip 0), 0 (pad)
//   cw8      [n_slots/group][128][group]  8 chips per word as 0/1 nibbles (pad chip 1023 = 0)
void launch_build_codes(hipStream_t s, const uint8_t *d_prns, int n_slots, int group, uint8_t *d_chips,
                        uint32_t *d_chipbits, uint32_t *d_cw, uint32_t *d_cw8);

// K2+K3+K4 fused acquisition search.  group = kAcqGroup (grid; local_units = sharding units of this rank) or 1 (job
// list; local_units = jobs).  d_cw is the table matching `algo` (cw for kAlgoSad, cw8 for kAlgoDot8).
void launch_acq(hipStream_t s, int group, int algo, long local_units, const AcqParams &prm, const uint8_t *d_if,
                const uint32_t *d_cw, const uint32_t *d_chipbits);
// Polyphase variant for phase_mode FINE, no inspection outputs (k_acq_poly.hip).  n_ms > 1 needs d_energy:
// acq_poly_energy_bytes(local_units) of scratch for the running per-hypothesis sums between blocks.  d_keyacc / d_sumacc: two
// u32 scratch planes of n_peaks entries, used (zeroed, merged with atomics, converted into d_peaks) only when the launch
// is split into two 8-offset workgroups per chip; the one-workgroup-per-chip form writes d_peaks directly.
// Returns the name of the dominant kernel it launched (for gpsx_last_kernel / bench reports).
const char *launch_acq_poly(hipStream_t s, long local_units, const AcqParams &prm, const uint8_t *d_if, const uint32_t *d_cw8,
                     const uint32_t *d_chipbits, uint32_t *d_keyacc, uint32_t *d_sumacc, size_t n_peaks,
                     gpsx_peak_t *d_peaks, bool peaks_are_zero, uint32_t *d_energy, bool block_parallel,
                     int seg_force);
// scratch of the block-parallel multi-block form: every block's magnitudes, u16 per hypothesis
inline size_t acq_poly_vals_bytes(int n_search, int n_ms, int n_prn, int n_dopp)
{
  return (size_t)n_search * n_ms * n_prn * n_dopp * 16 * 1024 * sizeof(uint16_t);
}
inline size_t acq_poly_energy_bytes(long local_units)
{
  return (size_t)local_units * kSuperGroups * kAcqGroup * 16 * 1024 * sizeof(uint32_t);
}
// Matrix-core variant (k_acq_mx.hip): phase_mode FINE, no inspection outputs; one 512-thread workgroup per (search, Doppler,
// 32 PRN slots).  Tables: mx_a [sets][4096] A fragments, mx_t [sets][1032] transposed chip words (launch_build_mx_tables).
// n_ms > 1 needs d_energy = acq_mx_energy_bytes(clusters) of scratch.
void launch_build_mx_tables(hipStream_t s, const uint32_t *d_chipbits, int n_slots, uint32_t *d_mx_a, uint32_t *d_mx_t);
long acq_mx_clusters(const AcqParams &prm);
constexpr size_t kMxZeroRecBytes = 16384;          // a tile-row's worth of all-zero records in front of the flags: what the walk
                                                   // forms "read back" in the first block of a search (16 offsets x ... of the
                                                   // same addresses: 4 tiles x 64 lanes x 4 groups x 12 B = 12 KB, rounded up)
inline size_t acq_mx_energy_bytes(long clusters)   // per workgroup: 8 waves x 16 offsets x 4 tiles x 4 groups x 64 lanes x 12 B
{                                                  // (the 24-bit records of the fallback form), + the zero records + one overflow flag
  return (size_t)clusters * 8 * (16 * 4 * 4 * 64) * 12 + kMxZeroRecBytes + (size_t)clusters * 4;
}
// block_parallel (n_ms > 1): a workgroup per (cluster, block) writes magnitudes into d_energy (acq_poly_vals_bytes of it, u16),
// k_acq_vals_search sums and searches -- the form for a handful of multi-block searches
// d_planes (may be null): 2 * n_peaks u32 of scratch; with it, single-block fine grids of at most n_cus / 2 clusters run as two
// workgroups per cluster (sample offsets 0..7 / 8..15) that merge through the planes + k_acq_finalize ("k_acq_mx<5>")
void launch_acq_finalize_from(hipStream_t s, uint32_t *d_keyacc, uint32_t *d_sumacc, size_t first, size_t n_peaks,
                              gpsx_peak_t *d_peaks, int n_prn, int n_dopp, int n_bits, int n_sets, int cluster_from,
                              int64_t *d_keys_opt = nullptr);   // d_keys_opt (n_bits = 8): the packed keys as well
const char *launch_acq_mx(hipStream_t s, const AcqParams &prm, const uint8_t *d_if, const uint32_t *d_mx_a,
                          const uint32_t *d_mx_t, gpsx_peak_t *d_peaks, uint32_t *d_energy, bool block_parallel, size_t n_peaks,
                          uint32_t *d_planes, int n_cus, bool *keys_done);
void launch_acq_vals_search(hipStream_t s, const AcqParams &prm, const uint16_t *d_vals, gpsx_peak_t *d_peaks, size_t n_peaks);
void launch_acq_finalize(hipStream_t s, uint32_t *d_keyacc, uint32_t *d_sumacc, size_t n_peaks,
                         gpsx_peak_t *d_peaks, int64_t *d_keys_opt = nullptr);

// keys[unit pair] = max over bit shifts of (max_val << 14 | 16383 - (8 * phase + b)); 0 for pairs of other shards
void launch_acq_keys(hipStream_t s, const gpsx_peak_t *d_peaks, int64_t *d_keys, int n_search, int n_prn, int n_groups,
                     int n_dopp, int n_bits, int unit_lo, int unit_hi);

// generic per-call primitives on caller-shaped buffers
void launch_wipeoff(hipStream_t s, const uint8_t *d_signal, float freq_hz, uint32_t accum_in, uint8_t *d_i,
                    uint8_t *d_q, uint32_t *d_accum_out);
void launch_replica(hipStream_t s, const uint8_t *d_chips, unsigned offset_bits, uint16_t *d_out);
void launch_corr_offsets(hipStream_t s, const uint8_t *d_rep, const uint8_t *d_i, const uint8_t *d_q,
                         const uint16_t *d_offsets, int first_offset, int n, uint16_t *d_cnt_i, uint16_t *d_cnt_q,
                         int16_t *d_corr8);
void launch_mag8(hipStream_t s, const uint16_t *d_cnt_i, const uint16_t *d_cnt_q, int n, int16_t *d_out,
                 uint32_t *d_disagree);
void launch_search_reduce(hipStream_t s, const int16_t *d_corr8, int n, int first_offset, gpsx_peak_t *d_peak);

// K2+K3+K5 tracking correlators, one workgroup per channel
// d_bad_prn (may be null): set to 1 by a channel whose PRN is outside 1..210 (it correlates against the empty code)
// d_trk_rep: the replica bit streams of every PRN slot (launch_build_track_rep), read by the wave-per-channel form
void launch_track_epl(hipStream_t s, const uint8_t *d_if_block, int if_format, int if_hz, gpsx_trk_state_t *d_st, int n_ch,
                      const uint8_t *d_chips, const uint32_t *d_chipbits, const uint32_t *d_trk_rep, int16_t *d_iq,
                      uint32_t *d_bad_prn, int wave_from);
constexpr int kTrackRepStride = 1032;   // words per PRN row of d_trk_rep
void launch_build_track_rep(hipStream_t s, const uint32_t *d_chipbits_all, int n_slots, uint32_t *d_rep);
// extension: the acquisition grid on weighted two-bit samples (k_acq_weighted.hip); -1 if the kernel's LDS size is refused
int launch_acq_weighted(hipStream_t s, const uint8_t *d_if_blocks, int n_search, int stride_blocks, int n_prn,
                        const uint8_t *d_chips_all, const uint8_t *d_prns, int if_hz, int dopp_min_hz, int dopp_step_hz, int n_dopp,
                        int use_magnitude, gpsx_peak_t *d_peaks);
// the same grid on the matrix cores (k_acq_mx.hip: k_acq_mxw): d_mx_a = the chip tables of the sign-only grid (launch_build_mx_tables)
void launch_acq_mxw(hipStream_t s, const uint8_t *d_if_blocks, int n_search, int stride_blocks, int n_prn, const uint32_t *d_mx_a,
                    int if_hz, int dopp_min_hz, int dopp_step_hz, int n_dopp, int use_magnitude, gpsx_peak_t *d_peaks);
// GPSX_DRAWS_LIBC (include/gpsx.h): a channel's false-lock jump reported by the first pass / its carrier candidate for the second
// (ms_from: the millisecond of the launch at which the channel's state in HBM is valid -- 0, or, under the multiplex, the first
//  millisecond of the slot it stopped in: its earlier slots of the launch were stored when they ended -- the replay starts there)
struct gpsx_loop_event_t { int32_t channel, ms, if_freq_i16, found_freq_hz, ms_from; };
struct gpsx_loop_reseed_t { int32_t ms, candidate, ms_from; };   // ms < 0: none
void launch_loop_scatter_reseeds(hipStream_t s, gpsx_loop_reseed_t *d_table, const int *d_channels, const gpsx_loop_reseed_t *d_cand, int n);
void launch_track_loop(hipStream_t s, const uint8_t *d_if_blocks, uint32_t block_stride, int n_blocks, int if_format, int if_hz,
                       gpsx_loop_state_t *d_st, int n_ch, uint32_t first_tick, int schedule, int word_sync,
                       const uint32_t *d_chipbits, const uint32_t *d_trk_rep, uint8_t *d_flags, gpsx_loop_trace_t *d_trace,
                       uint32_t *d_bad_prn, const int *d_ch_map, int n_map, const gpsx_loop_reseed_t *d_reseeds,
                       gpsx_loop_event_t *d_events, uint32_t *d_n_events);
void launch_loop_set_polarity(hipStream_t s, gpsx_loop_state_t *d_st, const int *d_channels, const uint8_t *d_values, int n);
constexpr int kTrackPadPrn = -2147483647 - 1;   // gpsx_trk_state_t.prn of a padding channel: the empty code, not an error
// N3: 2-bit sign/magnitude samples -> two 1-bit planes
void launch_unpack2(hipStream_t s, const uint8_t *d_in, int n_blocks, uint8_t *d_sign, uint8_t *d_mag);
void launch_rewind(hipStream_t s, int if_hz, gpsx_trk_state_t *d_st, int n_ch, const uint8_t *d_steps);

}  // namespace gpsx
