// k_acq_grid.hip -- the acquisition search kernel (K2 + K3 + K4 of SURVEY.md 2.1, fused), gfx950 / wave64.
//
// What it replaces (PM = Firmware/project_main of the reference):
//   gps_generate_prn_data2   PM/GPS/gps_misc.c:282-300   replica, 16 samples per chip, shifted by b bits
//   gps_shift_to_zero_freq   PM/GPS/gps_misc.c:211-240   1-bit carrier wipe-off, NCO restarted at phase 0
//   gps_mult_and_summ        PM/GPS/gps_misc.c:48-93     XOR + popcount of I and Q against the replica
//   gps_correlation8         PM/GPS/gps_misc.c:98-122    centre, clip, sqrtf(I^2 + Q^2)
//   correlation_search       PM/GPS/gps_misc.c:155-191   max / first argmax / sum over byte offsets
// called once per (channel, Doppler bin, ms) by PM/GPS/acquisition.c:196-312; here one workgroup does one
// (search, Doppler, bit shift) for G PRNs at once and returns exactly the triplets those calls would return.
//
// Formulation (derivation + CPU model: tests/test_formulation.py):
//   With D the wiped 16368-sample stream (circular; samples 16352.. are 0 because the NCO loop never mixes them)
//   and s = 8 o + b = 16 q + t0:
//       cnt(o, b) = C0(s) + [chip1022] (2 pop(D[8o, 8o+b)) - b) - [o odd] (w(p1) + [p1 != 1022] w(1022))
//       C0(16 q + t0) = sum_c | S'_t0[q + c] - B[c] |
//   S'_t0[k] = 1 + popcount(D[16k + t0, +16)) are block sums of the data, B[c] = 17 / 1 for chip 1 / 0.
//   So the 2 x 16368 one-bit MACs of a hypothesis become 2 x 256 byte-SADs (v_msad_u8: four chips per instruction,
//   the masked form drops the 1024th pad chip).  No matrix cores: this is integer compare-accumulate.
//
// Work split: 256 lanes; lane l owns chip offsets q = 4l .. 4l+3 (byte offsets o = 2q + t0/8) for G PRNs and both
// I and Q: 4 x G x 2 accumulators.  Per 4-chip step j it reads one new dword of S' per stream from LDS, forms the
// three unaligned windows with v_alignbyte_b32 and issues 8 G SADs against G wave-uniform code words (SGPRs).
//
// This kernel serves byte-granular grids, multi-block (non-coherent) searches, job lists and the inspection outputs; the
// single-block fine grid of the cold-start sweep runs k_acq_poly (k_acq_poly.hip), which reuses B0/C below.
//
// ALGO_DOT8 (default here; ALGO_SAD is kept for A/B) halves the main loop again: with 0/1 code nibbles,
//       C0(16 q + t0) = pop(D) + 8192 - 2 M,     M = sum_c chip[c] * S_t0[q + c]
//   and v_dot8_u32_u4 multiplies EIGHT 4-bit block sums by eight chips per instruction.  A block sum is 0..16; the one
//   value that does not fit a nibble (16 = a window of sixteen ones) is stored as 15 and its deficit restored exactly
//   by E = sum_c chip[c] * F[q + c] over the bit plane F of saturated windows -- an AND + popcount pass that only runs
//   for the (t0, I/Q) arrays that contain such a window at all (2^-16 per window on noise-like data; dense on clean
//   synthetic carriers, where the pass costs about half of the dot loop).
#include <cstdlib>

#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"

namespace gpsx {

namespace {

constexpr int kThreads = 256;
constexpr int kSDwords = 520;  // S' arrays: 2046 + pad bytes, doubled circular copy, as dwords
constexpr int kNibDwords = 264;  // 4-bit block sums: 2046 + pad nibbles, doubled circular copy, as dwords
constexpr int kFullWords = 68;   // bit plane of saturated windows, doubled circular copy

template <int G, int GPW>
struct AcqShared {
  uint16_t x[1024];              // raw IF block
  u32 d[2][514];                 // wiped I / Q streams: 511 words, word 511 = wrap-around copy, zero pad
  u32 s[2][2][kSDwords];         // [t0 index][I/Q] S' bytes (ALGO_SAD) or 4-bit block sums (ALGO_DOT8, first 264 dwords)
  u32 full[2][2][kFullWords];    // ALGO_DOT8: windows whose sum is 16
  u32 any_full[2][2];            // ALGO_DOT8: does the array hold any such window
  u32 ones[2];                   // ALGO_DOT8: pop(D) per stream
  u32 chipbits[GPW * G][34];     // 32 words of chips + zero pad for the 64-bit window reads
  u32 red[4][G][2];              // cross-wave reduction scratch
  u32 carry[GPW][2 * G][kThreads];  // each lane's running (best key, sum) per PRN, parked here while the dot loops run
};

__device__ __forceinline__ u32 lds_byte(const u32 *words, int byte_index)
{
  return (words[byte_index >> 2] >> ((byte_index & 3) * 8)) & 0xFFu;
}

// Which chip offset q (0..1023) hypothesis i (0..3) of lane `tid` is.
//   byte-SAD loop : four consecutive offsets, q = 4 tid + i (windows formed with v_alignbyte from one dword stream)
//   dot8 loop     : q = 32 (tid / 8) + (tid % 8) + 8 i -- the lane's offsets are one 8-nibble dword apart, so the data
//                   window of (i, step j) is the window of (i - 1, step j + 1): ONE new v_alignbit per step serves all four
template <int ALGO>
__device__ __forceinline__ int chip_offset_of(int tid, int i)
{
  return ALGO == kAlgoDot8 ? 32 * (tid >> 3) + (tid & 7) + 8 * i : 4 * tid + i;
}

// E[i][p] = sum_c chip_p[c] * F[q_i + c]: AND + popcount of the saturated-window bit plane against the packed chips
template <int G>
__device__ __forceinline__ void add_saturation_deficit(u32 (&acc)[4][G], const u32 *full_bits, const u32 *__restrict__ chipbits,
                                                    int tid)
{
#pragma unroll 1
  for (int w = 0; w < 32; w++) {
    u32 fwin[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int bit = chip_offset_of<kAlgoDot8>(tid, i) + 32 * w;
      fwin[i] = __builtin_amdgcn_alignbit(full_bits[(bit >> 5) + 1], full_bits[bit >> 5], (u32)(bit & 31));
    }
#pragma unroll
    for (int p = 0; p < G; p++) {
      const u32 chips32 = chipbits[p * 32 + w];  // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; i++)
        acc[i][p] += __popc(fwin[i] & chips32);
    }
  }
}

}  // namespace

// waves per SIMD the register allocator must leave room for: the byte-SAD single-block kernel fits 128 VGPRs (4 waves);
// the dot-product variant carries more live state and gets 168 (3 waves), the multi-block variants (64 extra energy
// registers) 256 (2 waves), instead of spilling to scratch
// DBG: the optional inspection outputs (raw counts, energy plane, per-block triplets) exist only in this instantiation
// GPW: PRN groups of G handled per workgroup, one after the other, on the SAME wiped data and block sums -- the
//      per-millisecond preamble (capture load, wipe-off, block sums, barriers) is paid once for GPW * G PRNs
template <int G, bool MULTI, int ALGO, bool DBG, int GPW>
__global__ __launch_bounds__(kThreads, MULTI ? 2 : ((ALGO == kAlgoDot8 || GPW > 1) ? 3 : 4)) void k_acq(const AcqParams prm, const uint8_t *__restrict__ if_blocks,
                                                  const u32 *__restrict__ cw, const u32 *__restrict__ chipbits)
{
  constexpr bool DOT8 = ALGO == kAlgoDot8;
  constexpr int kSteps = DOT8 ? kCodeWords / 2 : kCodeWords;   // 8 or 4 chips per main-loop step
  __shared__ AcqShared<G, GPW> sh;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- decode this workgroup's search ------------------------------------------------------------------------
  // Grid mode: the sharding unit is (search, Doppler, G-PRN group), one workgroup per unit and replica bit shift.
  int first_block, group0, n_groups_here, b, win_start, win_stop, dopp = 0, search = 0;
  float freq_hz;
  AcqJobRec jr{};
  if (prm.jobs) {
    jr = prm.jobs[blockIdx.x];
    first_block = jr.block;
    group0 = jr.slot / G;   // G == 1 in job mode
    n_groups_here = 1;
    freq_hz = jr.freq_hz;
    b = jr.offset_bits & 15;
    win_start = jr.win_start;
    win_stop = jr.win_stop;
  } else {
    constexpr int kParts = kSuperGroups / GPW;   // workgroups per (super unit, bit shift)
    int id = blockIdx.x;
    const int part = id % kParts;
    id /= kParts;
    b = id % prm.n_bits;
    const int unit_local = id / prm.n_bits;
    const int unit = prm.unit_lo + unit_local;
    const int t = unit / prm.n_groups;
    dopp = t % prm.n_dopp;
    search = t / prm.n_dopp;
    first_block = search * prm.search_stride_blocks;
    group0 = unit % prm.n_groups + part * GPW;
    n_groups_here = prm.n_groups - group0 < GPW ? prm.n_groups - group0 : GPW;
    if (n_groups_here <= 0)
      return;   // odd number of groups: the last super group has one
    // int arithmetic, then one conversion: PM/GPS/acquisition.c:285-289
    freq_hz = (float)(prm.if_hz + prm.dopp_min_hz + dopp * prm.dopp_step_hz);
    win_start = prm.win_start;
    win_stop = prm.win_stop;
  }
  const u32 step_word = nco_step_per_word(freq_hz);
  const u32 low_mask = (1u << b) - 1u;                 // replica bits of word i that still belong to chip i-1
  const u32 high_mask = (0xFFFFu << b) & 0xFFFFu;      // ... and to chip i

  for (int i = tid; i < GPW * G * 34; i += kThreads) {
    const int p = i / 34, w = i - p * 34;
    sh.chipbits[p][w] = (w < 32 && p < n_groups_here * G) ? chipbits[(size_t)(group0 * G + p) * 32 + w] : 0u;
  }

  u32 energy[MULTI ? 2 : 1][MULTI ? 4 : 1][MULTI ? G : 1];   // MULTI implies GPW == 1
  if (MULTI) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int p = 0; p < G; p++)
          energy[MULTI ? a : 0][MULTI ? i : 0][MULTI ? p : 0] = 0;
  }

  for (int ms = 0; ms < prm.n_ms; ms++) {
    const bool last_ms = ms == prm.n_ms - 1;
    // ---- A1: IF block -> LDS (coalesced 16-bit loads: a block starts on an even byte) -------------------------
    const size_t block_bytes = prm.if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : kBytes;
    const uint8_t *blk = if_blocks + (size_t)(first_block + ms) * block_bytes;
    __syncthreads();  // previous iteration's readers of sh.* are done
    for (int i = tid; i < 1024; i += kThreads)   // 2-bit captures are unpacked to their sign plane on the way into LDS
      sh.x[i] = i < kWords16 ? load_sign16(blk, i, prm.if_format) : (uint16_t)0;
    if (DOT8) {
      for (int i = tid; i < 4 * kFullWords; i += kThreads)
        (&sh.full[0][0][0])[i] = 0;
      if (tid < 4)
        (&sh.any_full[0][0])[tid] = 0;
      if (tid < 2)
        sh.ones[tid] = 0;
    }
    __syncthreads();
    // ---- A2: carrier wipe-off (K3).  Word w sees the NCO after w steps from zero. -----------------------------
    const u32 *x32 = reinterpret_cast<const u32 *>(sh.x);
    u32 ones_i = 0, ones_q = 0;
    for (int w = tid; w < 514; w += kThreads) {
      u32 vi = 0, vq = 0;
      if (w < kWords32) {
        const u32 quad = (step_word * (u32)w) >> 30;
        vi = carrier_i(quad) ^ x32[w];
        vq = carrier_q(quad) ^ x32[w];
      }
      sh.d[0][w] = vi;
      sh.d[1][w] = vq;
      ones_i += __popc(vi);
      ones_q += __popc(vq);
    }
    if (DOT8) {
      ones_i = wave_sum_u32(ones_i);
      ones_q = wave_sum_u32(ones_q);
      if (lane == 0) {
        atomicAdd(&sh.ones[0], ones_i);
        atomicAdd(&sh.ones[1], ones_q);
      }
    }
    __syncthreads();
    if (tid < 2)  // word 511: samples 16352..16367 are zero, then the stream wraps to sample 0
      sh.d[tid][511] = sh.d[tid][0] << 16;
    __syncthreads();
    // ---- A3: block sums S'_t0[k] = 1 + pop(D[16k + t0, +16)), k circular over 1023, for t0 = b and b + 8 --------
    if (!DOT8) {
      for (int m = tid; m < 4 * kSDwords; m += kThreads) {
        const int arr = m / kSDwords;       // 0..3 = (t0 index, I/Q)
        const int dw = m - arr * kSDwords;
        const int t0 = b + 8 * (arr >> 1);
        const u32 *dd = sh.d[arr & 1];
        u32 packed = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          int k = dw * 4 + e;
          k = k >= 2 * kChips ? k - 2 * kChips : (k >= kChips ? k - kChips : k);
          const int pos = 16 * k + t0;
          const u32 win = __builtin_amdgcn_alignbit(dd[(pos >> 5) + 1], dd[pos >> 5], (u32)(pos & 31));
          packed |= (1u + pop16(win)) << (8 * e);
        }
        sh.s[arr >> 1][arr & 1][dw] = packed;
      }
    } else {
      // 4-bit sums min(S, 15), eight per dword; windows with S == 16 are flagged in the bit plane `full`.
      // One task = one dword of BOTH t0 arrays of one stream: the sixteen windows D[16 k + b, +16) and D[16 k + b + 8, +16),
      // k = k0 .. k0 + 7, are cut from five funnel-shifted words (consecutive windows are consecutive 16-bit fields).
      for (int m = tid; m < 2 * kNibDwords; m += kThreads) {
        const int iq = m / kNibDwords;
        const int dw = m - iq * kNibDwords;
        const u32 *dd = sh.d[iq];
        const int kd0 = dw * 8;   // index in the doubled array
        const int k0 = kd0 >= 2 * kChips ? kd0 - 2 * kChips : (kd0 >= kChips ? kd0 - kChips : kd0);
        u32 sum0[8], sum1[8];
        if (k0 + 7 < kChips) {
          const int w = k0 >> 1;
          const u32 sft = 16u * (u32)(k0 & 1) + (u32)b;
          u32 x[5];
#pragma unroll
          for (int j = 0; j < 5; j++)
            x[j] = __builtin_amdgcn_alignbit(dd[w + j + 1], dd[w + j], sft);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            sum0[2 * j] = pop16(x[j]);
            sum0[2 * j + 1] = (u32)__popc(x[j] >> 16);
            sum1[2 * j] = pop16(x[j] >> 8);
            sum1[2 * j + 1] = pop16(__builtin_amdgcn_alignbit(x[j + 1], x[j], 24u));
          }
        } else {  // the two dwords per array where k wraps around 1023
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int kd = kd0 + e;
            const int k = kd >= 2 * kChips ? kd - 2 * kChips : (kd >= kChips ? kd - kChips : kd);
            const int pos = 16 * k + b;
            const u32 lo = dd[pos >> 5], mid = dd[(pos >> 5) + 1];
            sum0[e] = pop16(__builtin_amdgcn_alignbit(mid, lo, (u32)(pos & 31)));
            const int pos1 = pos + 8;
            sum1[e] = pop16(__builtin_amdgcn_alignbit(dd[(pos1 >> 5) + 1], dd[pos1 >> 5], (u32)(pos1 & 31)));
          }
        }
        u32 packed0 = 0, packed1 = 0, full0 = 0, full1 = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          packed0 |= (sum0[e] - (sum0[e] >> 4)) << (4 * e);   // 16 -> 15, everything else unchanged
          packed1 |= (sum1[e] - (sum1[e] >> 4)) << (4 * e);
          full0 |= (sum0[e] >> 4) << e;
          full1 |= (sum1[e] >> 4) << e;
        }
        sh.s[0][iq][dw] = packed0;
        sh.s[1][iq][dw] = packed1;
        if (full0) {   // rare on noise-like data
          atomicOr(&sh.full[0][iq][kd0 >> 5], full0 << (kd0 & 31));
          sh.any_full[0][iq] = 1u;
        }
        if (full1) {
          atomicOr(&sh.full[1][iq][kd0 >> 5], full1 << (kd0 & 31));
          sh.any_full[1][iq] = 1u;
        }
      }
    }
    __syncthreads();

    const u32 wrap_i = (sh.d[0][0] & 0xFFu) << 8;  // data bytes (2045, 0): the word odd offsets skip at the wrap
    const u32 wrap_q = (sh.d[1][0] & 0xFFu) << 8;

    // ---- B: the SAD loops, t0 = b (even byte offsets) then t0 = b + 8 (odd byte offsets) ------------------------
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
#pragma unroll 1
     for (int gi = 0; gi < n_groups_here; gi++) {
      // per-group addressing (wave-uniform scalars)
      const int slot0 = (group0 + gi) * G;
      const u32 *cw_group = cw + (size_t)(group0 + gi) * kSteps * G;
      const int out_pstride = prm.jobs ? 1 : prm.n_dopp * prm.n_bits;
      const int out0 = prm.jobs ? jr.out_index : ((search * prm.n_prn + slot0) * prm.n_dopp + dopp) * prm.n_bits + b;
      const int n_valid = prm.jobs ? 1 : (prm.n_prn - slot0 < G ? prm.n_prn - slot0 : G);
      u32 best[G], total[G];   // running result per PRN: packed (value << 11 | 2047 - offset) maximum, window sum
#pragma unroll
      for (int p = 0; p < G; p++) {
        best[p] = 0;
        total[p] = 0;
      }
      u32 acc_i[4][G], acc_q[4][G];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int p = 0; p < G; p++) {
          acc_i[i][p] = 0;
          acc_q[i][p] = 0;
        }
      if (!DOT8) {
        const u32 *si = sh.s[half][0] + tid;
        const u32 *sq = sh.s[half][1] + tid;
        u32 cur_i = si[0], cur_q = sq[0];
#pragma unroll 2
        for (int j = 0; j < kCodeWords; j++) {
          const u32 nxt_i = si[j + 1];
          const u32 nxt_q = sq[j + 1];
          u32 wi[4], wq[4];
          wi[0] = cur_i;
          wq[0] = cur_q;
#pragma unroll
          for (int i = 1; i < 4; i++) {
            wi[i] = __builtin_amdgcn_alignbyte(nxt_i, cur_i, (u32)i);
            wq[i] = __builtin_amdgcn_alignbyte(nxt_q, cur_q, (u32)i);
          }
#pragma unroll
          for (int p = 0; p < G; p++) {
            const u32 code = cw_group[j * G + p];  // wave-uniform -> scalar load
#pragma unroll
            for (int i = 0; i < 4; i++) {
              acc_i[i][p] = __builtin_amdgcn_msad_u8(wi[i], code, acc_i[i][p]);
              acc_q[i][p] = __builtin_amdgcn_msad_u8(wq[i], code, acc_q[i][p]);
            }
          }
          cur_i = nxt_i;
          cur_q = nxt_q;
        }
      } else {
        // lane's offsets q = 32 a + c + 8 i (a = tid / 8, c = tid % 8): nibble q + 8 j sits in dword 4 a + i + j at bit 4 c,
        // so step j uses windows W[j .. j + 3] of the lane's stream W[m] = alignbit(dw[m + 1], dw[m], 4 c)
        int tid_m = tid;
        asm volatile("" : "+v"(tid_m));   // as in the epilogue: recompute the lane's addresses rather than keep them live
        const u32 *ni = sh.s[half][0] + 4 * (tid_m >> 3);
        const u32 *nq = sh.s[half][1] + 4 * (tid_m >> 3);
        const u32 shn = 4u * (u32)(tid_m & 7);
        u32 wi[4], wq[4];
        u32 prev_i = ni[3], prev_q = nq[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
          wi[m] = __builtin_amdgcn_alignbit(ni[m + 1], ni[m], shn);
          wq[m] = __builtin_amdgcn_alignbit(nq[m + 1], nq[m], shn);
        }
#pragma unroll 4
        for (int j = 0; j < kSteps; j++) {
          const u32 new_i = ni[j + 4];
          const u32 new_q = nq[j + 4];
          wi[3] = __builtin_amdgcn_alignbit(new_i, prev_i, shn);
          wq[3] = __builtin_amdgcn_alignbit(new_q, prev_q, shn);
          prev_i = new_i;
          prev_q = new_q;
#pragma unroll
          for (int p = 0; p < G; p++) {
            const u32 code = cw_group[j * G + p];  // eight 0/1 chip nibbles, wave-uniform -> scalar load
#pragma unroll
            for (int i = 0; i < 4; i++) {
              acc_i[i][p] = __builtin_amdgcn_udot8(wi[i], code, acc_i[i][p], false);
              acc_q[i][p] = __builtin_amdgcn_udot8(wq[i], code, acc_q[i][p], false);
            }
          }
#pragma unroll
          for (int m = 0; m < 3; m++) {
            wi[m] = wi[m + 1];
            wq[m] = wq[m + 1];
          }
        }
        // deficit of the saturated windows: E = sum_c chip[c] * F[q + c], only where such windows exist
        if (sh.any_full[half][0])
          add_saturation_deficit<G>(acc_i, sh.full[half][0], chipbits + (size_t)slot0 * 32, tid_m);
        if (sh.any_full[half][1])
          add_saturation_deficit<G>(acc_q, sh.full[half][1], chipbits + (size_t)slot0 * 32, tid_m);
      }

      // ---- C: per-hypothesis corrections, magnitude, running search result -------------------------------------
      // The running (best, total) of the even offsets were parked in LDS so that they do not occupy registers during
      // the dot loop of the odd offsets (lane-private columns: conflict-free).
      if (half == 1) {
#pragma unroll
        for (int p = 0; p < G; p++) {
          best[p] = sh.carry[gi][2 * p][tid];
          total[p] = sh.carry[gi][2 * p + 1][tid];
        }
      }
      // (tid laundered through an empty asm: everything derived from it below -- offsets, window tests, LDS addresses --
      //  is then recomputed here, a handful of ALU ops, instead of being hoisted out of the millisecond loop and held
      //  in ~25 registers across the dot loop, which is what pushed the kernel into scratch spills)
      int tid_e = tid;
      asm volatile("" : "+v"(tid_e));
      // Wave-uniform pieces first.  The counts are affine in the accumulators: cnt = base + sign * acc.
      const int base_i = DOT8 ? __builtin_amdgcn_readfirstlane((int)sh.ones[0]) + kHalf + 8 : 0;   // C0 = pop(D) + 8192 - 2 M
      const int base_q = DOT8 ? __builtin_amdgcn_readfirstlane((int)sh.ones[1]) + kHalf + 8 : 0;
      constexpr int kSign = DOT8 ? -2 : 1;
      // Odd offsets skip replica word p1 = 1022 - q, which meets data bytes (2045, 0) = `wrap`.  That word's replica
      // bits are (chip[p1 - 1] ? low : 0) | (chip[p1] ? high : 0): four possible popcounts per stream, precomputed.
      // (packed one byte each into a scalar; the lane's two chip bits then select a byte with one v_bfe_u32)
      u32 wrap_tab_i = 0, wrap_tab_q = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 r = ((k & 1) ? low_mask : 0u) | ((k & 2) ? high_mask : 0u);
        wrap_tab_i |= pop16(wrap_i ^ r) << (8 * k);
        wrap_tab_q |= pop16(wrap_q ^ r) << (8 * k);
      }
      wrap_tab_i = (u32)__builtin_amdgcn_readfirstlane((int)wrap_tab_i);
      wrap_tab_q = (u32)__builtin_amdgcn_readfirstlane((int)wrap_tab_q);
      // One 32-chip window per PRN covers chips p1 - 1, p1 of the lane's four offsets (they lie within 26 chips).
      const int q_hi = chip_offset_of<ALGO>(tid_e, 3);
      const int chip_base = kChips - 2 - q_hi;   // lowest chip needed; bit k of chipwin = chip (chip_base + k)
      const int chip_lo = chip_base < 0 ? 0 : chip_base;
      u32 chipwin[G];
      if (half) {
#pragma unroll
        for (int p = 0; p < G; p++) {
          const u32 *cb = sh.chipbits[gi * G + p];
          const u64 two = (u64)cb[chip_lo >> 5] | ((u64)cb[(chip_lo >> 5) + 1] << 32);
          const u32 w = (u32)(two >> (chip_lo & 31));
          chipwin[p] = chip_base < 0 ? w << (chip_lo - chip_base) : w;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = chip_offset_of<ALGO>(tid_e, i);
        const int o = 2 * q + half;
        const bool exists = q < kChips;
        const bool in_win = exists && o >= win_start && o < win_stop;
        const int oc = exists ? o : 0;
        // quirk Q5 (replica shift not circular), applies when chip 1022 is set: + 2 pop(D[8o, 8o + b)) - b
        const int adj_i = 2 * (int)__popc(lds_byte(sh.d[0], oc) & low_mask) - b;
        const int adj_q = 2 * (int)__popc(lds_byte(sh.d[1], oc) & low_mask) - b;
        // quirk Q3: odd offsets also skip the last replica word (1022) -- data bytes (o - 2, o - 1) -- unless it IS p1
        const bool odd_tail = half && q > 0 && exists;
        u32 prev_i = 0, prev_q = 0;
        if (odd_tail) {
          prev_i = lds_byte(sh.d[0], oc - 2) | (lds_byte(sh.d[0], oc - 1) << 8);
          prev_q = lds_byte(sh.d[1], oc - 2) | (lds_byte(sh.d[1], oc - 1) << 8);
        }
        const int k0 = exists ? (kChips - 2 - q) - chip_base : 0;   // bit of chip p1 - 1 in chipwin (p1 = 1022 - q)
        const u32 key_lo = (u32)(2047 - o);
#pragma unroll
        for (int p = 0; p < G; p++) {
          const u32 tail_bits = chipbits[(size_t)(slot0 + p) * 32 + 31];  // wave-uniform -> scalar load
          const bool c1022 = (tail_bits >> 30) & 1u, c1021 = (tail_bits >> 29) & 1u;
          int ci = base_i + kSign * (int)acc_i[i][p] + (c1022 ? adj_i : 0);
          int cq = base_q + kSign * (int)acc_q[i][p] + (c1022 ? adj_q : 0);
          if (half) {
            const u32 sel8 = ((chipwin[p] >> k0) & 3u) * 8u;   // bit 0 = chip[p1 - 1], bit 1 = chip[p1]
            ci -= (int)__builtin_amdgcn_ubfe(wrap_tab_i, sel8, 8u);
            cq -= (int)__builtin_amdgcn_ubfe(wrap_tab_q, sel8, 8u);
            const u32 r_last = (c1021 ? low_mask : 0u) | (c1022 ? high_mask : 0u);   // wave-uniform
            ci -= odd_tail ? (int)__popc(prev_i ^ r_last) : 0;
            cq -= odd_tail ? (int)__popc(prev_q ^ r_last) : 0;
          }
          u32 val = in_win ? (u32)mag8_fast(ci, cq) : 0u;
          const int out_idx = out0 + p * out_pstride;
          if (DBG) {
            if (prm.cnt && last_ms && exists && p < n_valid) {
              prm.cnt[((size_t)out_idx * kBytes + o) * 2 + 0] = (uint16_t)ci;
              prm.cnt[((size_t)out_idx * kBytes + o) * 2 + 1] = (uint16_t)cq;
            }
          }
          if (MULTI) {
            if (DBG && prm.per_ms) {  // single-block triplet of this ms (what correlation_search would have returned)
              const u32 key = in_win ? (val << 11) | key_lo : 0u;
              best[p] = key > best[p] ? key : best[p];
              total[p] += val;
            }
            if (half == 0) {
              energy[0][MULTI ? i : 0][MULTI ? p : 0] += val;
              val = energy[0][MULTI ? i : 0][MULTI ? p : 0];
            } else {
              energy[MULTI ? 1 : 0][MULTI ? i : 0][MULTI ? p : 0] += val;
              val = energy[MULTI ? 1 : 0][MULTI ? i : 0][MULTI ? p : 0];
            }
          } else {
            const u32 key = in_win ? (val << 11) | key_lo : 0u;
            best[p] = key > best[p] ? key : best[p];
            total[p] += val;
          }
          if (DBG) {
            if (prm.energy && last_ms && exists && p < n_valid)
              prm.energy[(size_t)out_idx * kBytes + o] = val;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the four q epilogues sequential: registers, not ILP, are scarce
      }
      if (half == 0) {
#pragma unroll
        for (int p = 0; p < G; p++) {
          sh.carry[gi][2 * p][tid] = best[p];
          sh.carry[gi][2 * p + 1][tid] = total[p];
        }
      }

      // ---- D: workgroup reduction -> one triplet per PRN (per ms in MULTI mode only if asked) --------------------
      const bool reduce_now = half == 1 && (MULTI ? (DBG && prm.per_ms != nullptr) : true);
      if (reduce_now) {
#pragma unroll
        for (int p = 0; p < G; p++) {
          const u32 k = wave_max_to_lane63(best[p]);   // DPP network, result in lane 63
          const u32 t = wave_sum_to_lane63(total[p]);
          if (lane == 63) {
            sh.red[wave][p][0] = k;
            sh.red[wave][p][1] = t;
          }
        }
        __syncthreads();
        if (tid < n_valid) {
          u32 k = 0, t = 0;
          for (int w = 0; w < 4; w++) {
            k = sh.red[w][tid][0] > k ? sh.red[w][tid][0] : k;
            t += sh.red[w][tid][1];
          }
          gpsx_peak_t pk;
          pk.max_val = k >> 11;
          pk.phase = pk.max_val ? 2047u - (k & 2047u) : 0u;
          pk.sum = t;
          pk.avr = t / (2u * kChips);
          const int out_idx = out0 + tid * out_pstride;
          if (MULTI)
            prm.per_ms[(size_t)out_idx * prm.n_ms + ms] = pk;
          else
            prm.peaks[out_idx] = pk;
        }
        __syncthreads();   // sh.red is reused by the next group
      }
     }  // gi
    }  // half
  }  // ms

  if (MULTI) {
    // search over the accumulated energies (GPW == 1: the one group of this workgroup)
    const int slot0 = group0 * G;
    const int out_pstride = prm.jobs ? 1 : prm.n_dopp * prm.n_bits;
    const int out0 = prm.jobs ? jr.out_index : ((search * prm.n_prn + slot0) * prm.n_dopp + dopp) * prm.n_bits + b;
    const int n_valid = prm.jobs ? 1 : (prm.n_prn - slot0 < G ? prm.n_prn - slot0 : G);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < G; p++) {
      u32 k = 0, t = 0;
#pragma unroll
      for (int half = 0; half < 2; half++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int q = chip_offset_of<ALGO>(tid, i);
          const int o = 2 * q + half;
          const bool in_win = q < kChips && o >= win_start && o < win_stop;
          const u32 e = energy[MULTI ? half : 0][MULTI ? i : 0][MULTI ? p : 0];
          const u32 key = in_win ? (e << 11) | (u32)(2047 - o) : 0u;
          k = key > k ? key : k;
          t += in_win ? e : 0u;
        }
      k = wave_max_to_lane63(k);
      t = wave_sum_to_lane63(t);
      if (lane == 63) {
        sh.red[wave][p][0] = k;
        sh.red[wave][p][1] = t;
      }
    }
    __syncthreads();
    if (tid < n_valid) {
      u32 k = 0, t = 0;
      for (int w = 0; w < 4; w++) {
        k = sh.red[w][tid][0] > k ? sh.red[w][tid][0] : k;
        t += sh.red[w][tid][1];
      }
      gpsx_peak_t pk;
      pk.max_val = k >> 11;
      pk.phase = pk.max_val ? 2047u - (k & 2047u) : 0u;
      pk.sum = t;
      pk.avr = t / (2u * kChips);
      prm.peaks[out0 + tid * out_pstride] = pk;
    }
  }
}

template <int G, int ALGO, int GPW>
static void launch_acq_t(hipStream_t s, int n_workgroups, const AcqParams &prm, const uint8_t *d_if, const uint32_t *d_cw,
                         const uint32_t *d_chipbits)
{
  const dim3 grid(n_workgroups), block(kThreads);
  const bool dbg = prm.per_ms || prm.energy || prm.cnt;
  if constexpr (GPW == 1) {
    if (prm.n_ms > 1) {
      if (dbg)
        hipLaunchKernelGGL((k_acq<G, true, ALGO, true, 1>), grid, block, 0, s, prm, d_if, d_cw, d_chipbits);
      else
        hipLaunchKernelGGL((k_acq<G, true, ALGO, false, 1>), grid, block, 0, s, prm, d_if, d_cw, d_chipbits);
    } else {
      if (dbg)
        hipLaunchKernelGGL((k_acq<G, false, ALGO, true, 1>), grid, block, 0, s, prm, d_if, d_cw, d_chipbits);
      else
        hipLaunchKernelGGL((k_acq<G, false, ALGO, false, 1>), grid, block, 0, s, prm, d_if, d_cw, d_chipbits);
    }
  } else {
    hipLaunchKernelGGL((k_acq<G, false, ALGO, false, GPW>), grid, block, 0, s, prm, d_if, d_cw, d_chipbits);
  }
}

void launch_acq(hipStream_t s, int group, int algo, long local_units, const AcqParams &prm, const uint8_t *d_if,
                const uint32_t *d_cw, const uint32_t *d_chipbits)
{
  if (local_units <= 0)
    return;
  if (group == kAcqGroup) {
    // (a two-groups-per-workgroup form, sharing the preamble between 16 PRNs, measured neutral in round 1 and went
    //  when the sharding unit became one 8-PRN group)
    const int n_wg = (int)(local_units * prm.n_bits * kSuperGroups);
    if (algo == kAlgoDot8)
      launch_acq_t<kAcqGroup, kAlgoDot8, 1>(s, n_wg, prm, d_if, d_cw, d_chipbits);
    else
      launch_acq_t<kAcqGroup, kAlgoSad, 1>(s, n_wg, prm, d_if, d_cw, d_chipbits);
  } else {   // job list: one PRN, one workgroup per job
    if (algo == kAlgoDot8)
      launch_acq_t<1, kAlgoDot8, 1>(s, (int)local_units, prm, d_if, d_cw, d_chipbits);
    else
      launch_acq_t<1, kAlgoSad, 1>(s, (int)local_units, prm, d_if, d_cw, d_chipbits);
  }
}

// One packed key per (search, PRN, Doppler): max over the bit shifts, zero for units this shard did not compute.
__global__ void k_acq_keys(const gpsx_peak_t *__restrict__ peaks, int64_t *__restrict__ keys, int n_search, int n_prn,
                           int n_groups, int n_dopp, int n_bits, int unit_lo, int unit_hi)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = n_search * n_prn * n_dopp;
  if (idx >= n)
    return;
  const int dopp = idx % n_dopp;
  const int prn = (idx / n_dopp) % n_prn;
  const int search = idx / (n_dopp * n_prn);
  const int unit = (search * n_dopp + dopp) * n_groups + prn / kAcqGroup;
  int64_t best = 0;
  if (unit >= unit_lo && unit < unit_hi) {
    for (int b = 0; b < n_bits; b++) {
      const gpsx_peak_t pk = peaks[(size_t)idx * n_bits + b];
      const int64_t key = ((int64_t)pk.max_val << 14) | (int64_t)(16383 - (int)(8 * pk.phase + b));
      best = key > best ? key : best;
    }
  }
  keys[idx] = best;
}

void launch_acq_keys(hipStream_t s, const gpsx_peak_t *d_peaks, int64_t *d_keys, int n_search, int n_prn, int n_groups,
                     int n_dopp, int n_bits, int unit_lo, int unit_hi)
{
  const int n = n_search * n_prn * n_dopp;
  if (n <= 0)
    return;
  hipLaunchKernelGGL(k_acq_keys, dim3((n + 255) / 256), dim3(256), 0, s, d_peaks, d_keys, n_search, n_prn, n_groups,
                     n_dopp, n_bits, unit_lo, unit_hi);
}

}  // namespace gpsx
