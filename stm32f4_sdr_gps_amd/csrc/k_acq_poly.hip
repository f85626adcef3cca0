// k_acq_poly.hip -- the acquisition grid kernel for the fine (16368-phase) cold-start sweep, polyphase formulation.
//
// Same contract as k_acq (k_acq_grid.hip): per (search, PRN, Doppler, replica bit shift) the triplet correlation_search
// (PM/GPS/gps_misc.c:155-191) returns, bit for bit; same preamble (capture -> LDS, carrier wipe-off K3), same
// per-hypothesis corrections and magnitude.  What changes is how the 2 x 16368 sample-granular correlations
//      M_t0(q) = sum_c chip[c] * S_t0[q + c],   S_t0[k] = pop(D[16 k + t0, +16)),   s = 16 q + t0
// are obtained for the sixteen sample offsets t0 inside a chip.  Moving the windows by ONE sample changes each block sum
// by one sample leaving and one entering:  S_{t0+1}[k] = S_t0[k] - d_t0[k] + d_t0[k + 1],  d_t0[k] = D(16 k + t0), so
//      M_{t0+1}(q) = M_t0(q) - X_t0(q) + X_t0(q + 1),          X_t0(q) = sum_c chip[c] * d_t0[q + c]
// and X is a correlation of two BIT vectors of length 1023: 32 words of v_and_b32 + v_bcnt_u32_b32 (accumulating) per
// hypothesis and stream -- 64 instructions against the 128 v_dot8_u32_u4 of the direct form (and 2046 XOR/popcount
// pairs in the reference).  d_t0 is the t0-th polyphase component of the wiped stream (every 16th sample).
// A workgroup owns 1024 chip offsets q x G PRNs x I,Q for a SEGMENT of 4, 8 or 16 consecutive t0: the first is computed
// directly (4-bit block sums, v_dot8_u32_u4, exact saturation pass -- as in k_acq), the others by the recurrence.
// Per-(PRN, bit shift) search results are merged across even / odd byte offsets (t0 = b and b + 8) and the four waves:
// in LDS when the workgroup walks all 16 offsets (the triplet is then written once), through atomicMax / atomicAdd on
// two global u32 planes + k_acq_finalize when the chip is split between two 8-offset workgroups.
//
// Non-coherent integration over n_ms > 1 blocks (MULTI): the workgroup walks the blocks itself; between blocks the
// running energy of every hypothesis it owns sits in a private 64 KB-per-PRN slice of an HBM scratch buffer (read-add-
// write per block, dword per lane, coalesced; the last block searches on the sums instead of writing them back).
//
// (MODE kPolyMulti).  When the launch has few searches the blocks are spread over workgroups instead (kPolyStore): a
// workgroup per (unit, block) writes its magnitudes as u16, k_acq_vals_search sums the blocks and searches.
//
// Lane l owns q = 4 l .. 4 l + 3; X_t0(4 l + 4) is lane l + 1's first value, exchanged through LDS.
#include <cstdlib>
#include <type_traits>

#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"

namespace gpsx {

namespace {

constexpr int kThreads = 256;
constexpr int kNibDwords = 264;   // 4-bit block sums, 2046 + pad nibbles (circular copy appended)
constexpr int kFullWords = 68;    // bit plane of saturated windows
constexpr int kPlaneWords = 66;   // one polyphase bit plane: 1023 bits + circular copy
constexpr int kMaxSegment = 16;   // sample offsets per workgroup: 16 (one workgroup per chip), 8 (two) or 4 (four)
constexpr int kPolySingle = 0, kPolyMulti = 1, kPolyStore = 2;   // k_acq_poly's MODE
constexpr int kPH = 8;            // PRNs per X pass (all of the group: X and M registers together still fit 168 VGPRs)

template <int G, int SEG>
struct PolyShared {
  uint16_t x[1024];                        // raw IF block
  u32 d[2][514];                           // wiped I / Q streams (word 511 = wrap-around copy, then zero pad)
  u32 s[2][kNibDwords];                    // 4-bit block sums of the segment's first offset, I / Q
  u32 full[2][kFullWords];                 // their saturated windows
  u32 any_full[2];
  u32 ones[2];                             // pop(D) per stream
  u32 plane[2][SEG - 1][kPlaneWords];      // d_t0 for the SEG - 1 recurrence steps, I / Q
  u32 chipbits[G][34];
  u32 xch[2][2 * kPH][kThreads];           // first X value of every lane, for its left neighbour (two buffers, by step
                                           // parity: the readers of one step never meet the writers of the next)
  u32 part[8][G][2];                       // SEG == 16: (packed best key, sum) per bit shift and PRN, merged over the
                                           // four waves and the two offsets (t0 = b: even byte offsets, b + 8: odd)
};

__device__ __forceinline__ u32 lds_byte(const u32 *words, int byte_index)
{
  return (words[byte_index >> 2] >> ((byte_index & 3) * 8)) & 0xFFu;
}

// Corrections of the reference's quirks, magnitude, windowed max / sum, merge into the global planes.
// m_i / m_q: M_t0(q) for the lane's four q and G PRNs.
// MULTI (non-coherent integration over several blocks): the per-hypothesis energies live in HBM between blocks --
// energy[workgroup][PRN of the group][t0][i][tid] (q = 4 tid + i), 64 KB per PRN; every access is one dword per lane,
// 256 B contiguous per wave.  ms_first: nothing to read yet; ms_last: search on the sums instead of writing them back.
template <int G, int SEG, int MODE>
__device__ __forceinline__ void poly_finish_offset(PolyShared<G, SEG> &sh, int tid, int lane, int t0, const u32 (&m_i)[4][G],
                                                   const u32 (&m_q)[4][G], int win_start, int win_stop,
                                                   const u32 *__restrict__ chipbits_g, int n_valid, size_t out0,
                                                   size_t out_pstride, u32 *__restrict__ keyacc, u32 *__restrict__ sumacc,
                                                   gpsx_peak_t *__restrict__ peaks, u32 *__restrict__ energy,
                                                   size_t energy_pair0, bool ms_first, bool ms_last)
{
  constexpr bool MULTI = MODE == kPolyMulti, STORE = MODE == kPolyStore;
  const int b = t0 & 7, half = t0 >> 3;
  const u32 low_mask = (1u << b) - 1u;
  const u32 high_mask = (0xFFFFu << b) & 0xFFFFu;
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));   // recompute lane-derived values here instead of keeping them live across the loops
  const int base_i = __builtin_amdgcn_readfirstlane((int)sh.ones[0]) + kHalf + 8;   // C0 = pop(D) + 8192 - 2 M
  const int base_q = __builtin_amdgcn_readfirstlane((int)sh.ones[1]) + kHalf + 8;
  const u32 wrap_i = (sh.d[0][0] & 0xFFu) << 8;   // data bytes (2045, 0): the word odd offsets skip at the wrap
  const u32 wrap_q = (sh.d[1][0] & 0xFFu) << 8;
  u32 wrap_tab_i = 0, wrap_tab_q = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const u32 r = ((k & 1) ? low_mask : 0u) | ((k & 2) ? high_mask : 0u);
    wrap_tab_i |= pop16(wrap_i ^ r) << (8 * k);
    wrap_tab_q |= pop16(wrap_q ^ r) << (8 * k);
  }
  wrap_tab_i = (u32)__builtin_amdgcn_readfirstlane((int)wrap_tab_i);
  wrap_tab_q = (u32)__builtin_amdgcn_readfirstlane((int)wrap_tab_q);
  const int q_hi = 4 * tid_e + 3;
  const int chip_base = kChips - 2 - q_hi;   // lowest chip needed; bit k of chipwin = chip (chip_base + k)
  const int chip_lo = chip_base < 0 ? 0 : chip_base;
  u32 chipwin[G];
  if (half) {
#pragma unroll
    for (int p = 0; p < G; p++) {
      const u32 *cb = sh.chipbits[p];
      const u64 two = (u64)cb[chip_lo >> 5] | ((u64)cb[(chip_lo >> 5) + 1] << 32);
      const u32 w = (u32)(two >> (chip_lo & 31));
      chipwin[p] = chip_base < 0 ? w << (chip_lo - chip_base) : w;
    }
  }
  u32 best[G], total[G];
#pragma unroll
  for (int p = 0; p < G; p++) {
    best[p] = 0;
    total[p] = 0;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int q = 4 * tid_e + i;
    const int o = 2 * q + half;
    const bool exists = q < kChips;
    const bool in_win = exists && o >= win_start && o < win_stop;
    const int oc = exists ? o : 0;
    // quirk Q5 applies to PRNs whose chip 1022 is set: two candidate bases per offset, picked per PRN by a scalar mask
    const int base0_i = base_i, base0_q = base_q;
    const int base1_i = base_i + 2 * (int)__popc(lds_byte(sh.d[0], oc) & low_mask) - b;
    const int base1_q = base_q + 2 * (int)__popc(lds_byte(sh.d[1], oc) & low_mask) - b;
    const bool odd_tail = half && q > 0 && exists;                             // quirk Q3, replica word 1022
    u32 prev_i = 0, prev_q = 0;
    if (odd_tail) {
      prev_i = lds_byte(sh.d[0], oc - 2) | (lds_byte(sh.d[0], oc - 1) << 8);
      prev_q = lds_byte(sh.d[1], oc - 2) | (lds_byte(sh.d[1], oc - 1) << 8);
    }
    const int k0 = exists ? (kChips - 2 - q) - chip_base : 0;
    const u32 key_lo = (u32)(2047 - o);
    // MULTI: the running sums of this lane's eight hypotheses, fetched up front so that the eight magnitudes below
    // cover the latency (slices exist for all G PRNs of the group and whatever they hold before the first block is
    // never used: no guards, no branches around the loads)
    u32 *e_grp = energy + energy_pair0 * (16 * 1024);                 // wave-uniform base, 32-bit lane offset
    const u32 e_off = (u32)(t0 * 1024 + i * 256 + tid_e);
    u32 prev[MULTI ? G : 1];
    if (MULTI) {
#pragma unroll
      for (int p = 0; p < G; p++)
        prev[MULTI ? p : 0] = (e_grp + p * (16 * 1024))[e_off];
    }
#pragma unroll
    for (int p = 0; p < G; p++) {
      const u32 tail_bits = chipbits_g[p * 32 + 31];   // wave-uniform -> scalar load
      const bool c1022 = (tail_bits >> 30) & 1u, c1021 = (tail_bits >> 29) & 1u;
      int ci = (c1022 ? base1_i : base0_i) + __mul24((int)m_i[i][p], -2);   // -> v_cndmask + v_mad_i32_i24
      int cq = (c1022 ? base1_q : base0_q) + __mul24((int)m_q[i][p], -2);
      if (half) {
        const u32 sel8 = ((chipwin[p] >> k0) & 3u) * 8u;   // bit 0 = chip[p1 - 1], bit 1 = chip[p1], p1 = 1022 - q
        ci -= (int)__builtin_amdgcn_ubfe(wrap_tab_i, sel8, 8u);
        cq -= (int)__builtin_amdgcn_ubfe(wrap_tab_q, sel8, 8u);
        const u32 r_last = (c1021 ? low_mask : 0u) | (c1022 ? high_mask : 0u);
        ci -= odd_tail ? (int)__popc(prev_i ^ r_last) : 0;
        cq -= odd_tail ? (int)__popc(prev_q ^ r_last) : 0;
      }
      u32 val = in_win ? (u32)mag8_fast<!MULTI>(ci, cq) : 0u;   // (MULTI: the shortcut's extra branch costs registers there)
      if (STORE) {
        // one block of a multi-block search handled as a workgroup of its own: the magnitudes go out as they are (u16,
        // [t0][i][tid] like the energy slices: 128 B contiguous per wave), k_acq_vals_search sums and searches them
        uint16_t *v = reinterpret_cast<uint16_t *>(energy) + (energy_pair0 + (size_t)p * out_pstride * (16 * 1024 / 8)) +
                      (size_t)(t0 * 1024 + i * 256 + tid_e);
        if (p < n_valid)
          *v = (uint16_t)val;
        continue;
      }
      if (MULTI) {
        val += ms_first ? 0u : prev[MULTI ? p : 0];
        if (!ms_last)
          (e_grp + p * (16 * 1024))[e_off] = val;
      }
      // val is already 0 outside the window: such a key is below every key with val > 0, and if nothing exceeds 0 the
      // reported phase is 0 whatever the low bits say -- no second select needed
      const u32 key = (val << 11) | key_lo;
      best[p] = key > best[p] ? key : best[p];
      total[p] += val;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (STORE || (MULTI && !ms_last))
    return;
  // Wave reduction, then merge.  SEG == 16: the workgroup sees both offsets of every bit shift, so the merge stays in
  // LDS and the finished triplet is written once; SEG == 8: the other offset belongs to another workgroup, merge through
  // global atomics (k_acq_finalize converts the planes afterwards).
#pragma unroll
  for (int p = 0; p < G; p++) {
    const u32 k = wave_max_to_lane63(best[p]);   // DPP network: 6 ops each, against 6 LDS permutes + 6 ops
    const u32 t = wave_sum_to_lane63(total[p]);
    if (lane == 63 && p < n_valid) {
      if (SEG == 16) {
        atomicMax(&sh.part[b][p][0], k);
        atomicAdd(&sh.part[b][p][1], t);
      } else {
        const size_t idx = out0 + (size_t)p * out_pstride + (size_t)b;
        atomicMax(&keyacc[idx], k);
        atomicAdd(&sumacc[idx], t);
      }
    }
  }
  if (SEG == 16 && half) {
    __syncthreads();
    if (tid < n_valid) {
      const u32 k = sh.part[b][tid][0], t = sh.part[b][tid][1];
      gpsx_peak_t pk;
      pk.max_val = k >> 11;
      pk.phase = pk.max_val ? 2047u - (k & 2047u) : 0u;
      pk.sum = t;
      pk.avr = t / (2u * kChips);
      peaks[out0 + (size_t)tid * out_pstride + (size_t)b] = pk;
    }
  }
}

}  // namespace

// (kPolyMulti: two workgroups per CU -- 256 registers per lane: at three its 64 + 64 accumulators spilled 50 VGPRs to
//  scratch; it is the fallback now, the matrix-core kernel takes the launches large enough to walk their blocks)
template <int G, int SEG, int MODE>
__global__ __launch_bounds__(kThreads, MODE == kPolyMulti ? 2 : 3) void k_acq_poly(const AcqParams prm, const uint8_t *__restrict__ if_blocks,
                                                          const u32 *__restrict__ cw8, const u32 *__restrict__ chipbits,
                                                          u32 *__restrict__ keyacc, u32 *__restrict__ sumacc,
                                                          gpsx_peak_t *__restrict__ peaks, u32 *__restrict__ energy)
{
  constexpr bool MULTI = MODE == kPolyMulti, STORE = MODE == kPolyStore;
  __shared__ PolyShared<G, SEG> sh;
  const int tid = threadIdx.x;
  const int lane = tid & 63;

  // ---- decode: (sharding unit = search x Doppler x 8-PRN group) x segment [x block, STORE] --------------------------------
  constexpr int kSegs = kMaxSegment / SEG;   // workgroups per chip
  int id = blockIdx.x;
  const int seg = id % kSegs;
  id /= kSegs;
  const int gsel = id % kSuperGroups;
  id /= kSuperGroups;
  const int ms_store = STORE ? id % prm.n_ms : 0;
  const int unit_local = STORE ? id / prm.n_ms : id;
  const int unit = prm.unit_lo + unit_local;
  const int t = unit / prm.n_groups;
  const int dopp = t % prm.n_dopp;
  const int search = t / prm.n_dopp;
  const int group = unit % prm.n_groups + gsel;
  if (group >= prm.n_groups)
    return;
  const int slot0 = group * G;
  const int n_valid = prm.n_prn - slot0 < G ? prm.n_prn - slot0 : G;
  const int t0_first = seg * SEG;
  const float freq_hz = (float)(prm.if_hz + prm.dopp_min_hz + dopp * prm.dopp_step_hz);   // PM/GPS/acquisition.c:285-289
  const u32 step_word = nco_step_per_word(freq_hz);
  const u32 *cw_group = cw8 + (size_t)group * (kCodeWords / 2) * G;
  const u32 *chipbits_g = chipbits + (size_t)slot0 * 32;
  const size_t out_pstride = (size_t)prm.n_dopp * 8;
  const size_t out0 = ((size_t)(search * prm.n_prn + slot0) * prm.n_dopp + dopp) * 8;

  for (int i = tid; i < G * 34; i += kThreads) {
    const int p = i / 34, w = i - p * 34;
    sh.chipbits[p][w] = w < 32 ? chipbits_g[p * 32 + w] : 0u;
  }

  // MULTI: this workgroup's private slices, one per PRN of its group (u32 units).  STORE: the (block, PRN, Doppler)
  // plane of this workgroup's first PRN in the magnitude buffer (u16 units), PRN stride n_dopp planes
  const size_t energy_pair0 = STORE ? ((size_t)((search * prm.n_ms + ms_store) * prm.n_prn + slot0) * prm.n_dopp + dopp) * (16 * 1024)
                                    : (size_t)blockIdx.x * G;
  for (int i = tid; i < 8 * G * 2; i += kThreads)
    (&sh.part[0][0][0])[i] = 0;

  const int n_ms = MULTI ? prm.n_ms : 1;
#pragma unroll 1
  for (int ms = 0; ms < n_ms; ms++) {
  const bool ms_first = ms == 0, ms_last = ms == n_ms - 1;
  __syncthreads();   // the previous block's readers of the LDS arrays are done
  // ---- A1 / A2: capture -> LDS, carrier wipe-off (as k_acq) ---------------------------------------------------------
  const size_t block_bytes = prm.if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : kBytes;
  const uint8_t *blk = if_blocks + (size_t)(search * prm.search_stride_blocks + ms + ms_store) * block_bytes;
  for (int i = tid; i < 1024; i += kThreads)
    sh.x[i] = i < kWords16 ? load_sign16(blk, i, prm.if_format) : (uint16_t)0;
  for (int i = tid; i < 2 * kFullWords; i += kThreads)
    (&sh.full[0][0])[i] = 0;
  if (tid < 2) {
    sh.any_full[tid] = 0;
    sh.ones[tid] = 0;
  }
  __syncthreads();
  {
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
    ones_i = wave_sum_u32(ones_i);
    ones_q = wave_sum_u32(ones_q);
    if (lane == 0) {
      atomicAdd(&sh.ones[0], ones_i);
      atomicAdd(&sh.ones[1], ones_q);
    }
  }
  __syncthreads();
  if (tid < 2)
    sh.d[tid][511] = sh.d[tid][0] << 16;   // samples 16352..16367 are zero, then the stream wraps to sample 0
  __syncthreads();

  // ---- A3: block sums of the segment's first offset (nibbles + saturated-window plane) --------------------------------
  for (int m = tid; m < 2 * kNibDwords; m += kThreads) {
    const int iq = m / kNibDwords;
    const int dw = m - iq * kNibDwords;
    const u32 *dd = sh.d[iq];
    const int kd0 = dw * 8;
    u32 packed = 0, fullbits = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int kd = kd0 + e;
      const int k = kd >= 2 * kChips ? kd - 2 * kChips : (kd >= kChips ? kd - kChips : kd);
      const int pos = 16 * k + t0_first;
      const u32 sum = pop16(__builtin_amdgcn_alignbit(dd[(pos >> 5) + 1], dd[pos >> 5], (u32)(pos & 31)));
      packed |= (sum - (sum >> 4)) << (4 * e);
      fullbits |= (sum >> 4) << e;
    }
    sh.s[iq][dw] = packed;
    if (fullbits) {
      atomicOr(&sh.full[iq][kd0 >> 5], fullbits << (kd0 & 31));
      sh.any_full[iq] = 1u;
    }
  }
  // ---- A4: polyphase bit planes d_t0[k] = D(16 k + t0) for the seven recurrence steps, circular copy appended ------------
  for (int m = tid; m < 2 * (SEG - 1) * kPlaneWords; m += kThreads) {
    const int iq = m / ((SEG - 1) * kPlaneWords);
    const int r = m - iq * (SEG - 1) * kPlaneWords;
    const int st = r / kPlaneWords;
    const int w = r - st * kPlaneWords;
    const int t0 = t0_first + st;
    const u32 *dd = sh.d[iq];
    u32 bits = 0;
#pragma unroll 8
    for (int e = 0; e < 32; e++) {
      const int kd = 32 * w + e;
      const int k = kd >= 2 * kChips ? kd - 2 * kChips : (kd >= kChips ? kd - kChips : kd);
      const int pos = 16 * k + t0;   // < 16368: the stream's own samples (the last 16 are zero)
      bits |= ((dd[pos >> 5] >> (pos & 31)) & 1u) << e;
    }
    sh.plane[iq][st][w] = bits;
  }
  __syncthreads();

  // ---- B0: M for the first offset, directly: v_dot8_u32_u4 on the 4-bit sums, then the exact saturation deficit ------
  u32 m_i[4][G], m_q[4][G];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int p = 0; p < G; p++) {
      m_i[i][p] = 0;
      m_q[i][p] = 0;
    }
  {
    // nibble 4 tid + i + 8 j lives in dword tid / 2 + j at bit 16 (tid & 1) + 4 i
    const u32 *ni = sh.s[0] + (tid >> 1);
    const u32 *nq = sh.s[1] + (tid >> 1);
    const u32 sh0 = 16u * (u32)(tid & 1);
    u32 cur_i = ni[0], cur_q = nq[0];
#pragma unroll 2
    for (int j = 0; j < kCodeWords / 2; j++) {
      const u32 nxt_i = ni[j + 1];
      const u32 nxt_q = nq[j + 1];
      u32 wi[4], wq[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        wi[i] = __builtin_amdgcn_alignbit(nxt_i, cur_i, sh0 + 4u * (u32)i);
        wq[i] = __builtin_amdgcn_alignbit(nxt_q, cur_q, sh0 + 4u * (u32)i);
      }
#pragma unroll
      for (int p = 0; p < G; p++) {
        const u32 code = cw_group[j * G + p];   // eight 0/1 chip nibbles, wave-uniform -> scalar load
#pragma unroll
        for (int i = 0; i < 4; i++) {
          m_i[i][p] = __builtin_amdgcn_udot8(wi[i], code, m_i[i][p], false);
          m_q[i][p] = __builtin_amdgcn_udot8(wq[i], code, m_q[i][p], false);
        }
      }
      cur_i = nxt_i;
      cur_q = nxt_q;
    }
#pragma unroll 1
    for (int iq = 0; iq < 2; iq++) {
      if (!sh.any_full[iq])
        continue;
      const u32 *fw = sh.full[iq];
#pragma unroll 1
      for (int w = 0; w < 32; w++) {
        u32 fwin[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int bit = 4 * tid + i + 32 * w;
          fwin[i] = __builtin_amdgcn_alignbit(fw[(bit >> 5) + 1], fw[bit >> 5], (u32)(bit & 31));
        }
#pragma unroll
        for (int p = 0; p < G; p++) {
          const u32 chips32 = chipbits_g[p * 32 + w];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const u32 add = (u32)__popc(fwin[i] & chips32);
            if (iq == 0)
              m_i[i][p] += add;
            else
              m_q[i][p] += add;
          }
        }
      }
    }
  }
  poly_finish_offset<G, SEG, MODE>(sh, tid, lane, t0_first, m_i, m_q, prm.win_start, prm.win_stop, chipbits_g, n_valid,
                                    out0, out_pstride, keyacc, sumacc, peaks, energy, energy_pair0, ms_first, ms_last);

  // ---- B1..B7: one sample further each: M += X(q + 1) - X(q), X = AND + popcount against the polyphase plane ----------
#pragma unroll 1
  for (int st = 0; st < SEG - 1; st++) {
    int tid_m = tid;
    asm volatile("" : "+v"(tid_m));
    const int wbase = tid_m >> 3;                    // bit 4 tid + i + 32 w  ->  word tid / 8 + w, bit 4 (tid % 8) + i
    const u32 shb = 4u * (u32)(tid_m & 7);
#pragma unroll
    for (int ph = 0; ph < G / kPH; ph++) {   // unrolled: ph indexes the M registers
      u32 x_i[4][kPH], x_q[4][kPH];   // defined by the first word's popcounts (no zeroing pass: 64 fewer ops per step)
      const u32 *pi = sh.plane[0][st] + wbase;
      const u32 *pq = sh.plane[1][st] + wbase;
      u32 cur_i = pi[0], cur_q = pq[0];
      // Four words per trip: the 16 chip words of a trip arrive by four s_load_dwordx4 (the loop is NOT unrolled further:
      // hoisting all 128 chip words into SGPRs makes the compiler spill them through v_readlane, which costs more VALU
      // issue slots than the correlation itself).  The popcount accumulates in the instruction (v_bcnt_u32_b32 d, s, d);
      // written as asm because the optimiser otherwise reassociates the sums into bcnt + v_add3 chains.
      auto trip = [&](int w4, auto first_trip) {
        u32 c32[kPH][4];
#pragma unroll
        for (int p = 0; p < kPH; p++) {
          const uint4 c = *reinterpret_cast<const uint4 *>(chipbits_g + (ph * kPH + p) * 32 + 4 * w4);   // wave-uniform
          c32[p][0] = c.x;
          c32[p][1] = c.y;
          c32[p][2] = c.z;
          c32[p][3] = c.w;
        }
#pragma unroll
        for (int ww = 0; ww < 4; ww++) {
          const int w = 4 * w4 + ww;
          const u32 nxt_i = pi[w + 1];
          const u32 nxt_q = pq[w + 1];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const u32 wi = __builtin_amdgcn_alignbit(nxt_i, cur_i, shb + (u32)i);
            const u32 wq = __builtin_amdgcn_alignbit(nxt_q, cur_q, shb + (u32)i);
#pragma unroll
            for (int p = 0; p < kPH; p++) {
              const u32 ai = wi & c32[p][ww];
              const u32 aq = wq & c32[p][ww];
              if (decltype(first_trip)::value && ww == 0) {
                asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(x_i[i][p]) : "v"(ai));
                asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(x_q[i][p]) : "v"(aq));
              } else {
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x_i[i][p]) : "v"(ai));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x_q[i][p]) : "v"(aq));
              }
            }
          }
          cur_i = nxt_i;
          cur_q = nxt_q;
        }
      };
      trip(0, std::true_type{});
#pragma unroll 1
      for (int w4 = 1; w4 < 8; w4++)
        trip(w4, std::false_type{});
      // X(4 tid + 4) is the right neighbour's first value; lane 255's is X(1024) = X(1) = lane 0's second... not needed:
      // lane 255 owns q = 1020..1023 and only q <= 1022 exist, so its i = 3 result is never used.
#pragma unroll
      for (int p = 0; p < kPH; p++) {
        sh.xch[st & 1][2 * p][tid] = x_i[0][p];
        sh.xch[st & 1][2 * p + 1][tid] = x_q[0][p];
      }
      __syncthreads();
      const int nb = tid < kThreads - 1 ? tid + 1 : tid;
#pragma unroll
      for (int p = 0; p < kPH; p++) {
        const u32 right_i = sh.xch[st & 1][2 * p][nb];
        const u32 right_q = sh.xch[st & 1][2 * p + 1][nb];
        const int pp = ph * kPH + p;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          m_i[i][pp] += x_i[i + 1][p] - x_i[i][p];
          m_q[i][pp] += x_q[i + 1][p] - x_q[i][p];
        }
        m_i[3][pp] += right_i - x_i[3][p];
        m_q[3][pp] += right_q - x_q[3][p];
      }
    }
    poly_finish_offset<G, SEG, MODE>(sh, tid, lane, t0_first + st + 1, m_i, m_q, prm.win_start, prm.win_stop, chipbits_g,
                                      n_valid, out0, out_pstride, keyacc, sumacc, peaks, energy, energy_pair0,
                                      ms_first, ms_last);
  }
  }  // ms
}

// (packed key, sum) planes -> gpsx_peak_t
// The planes are all-zero between launches (allocated so, and every entry read here is put back to zero): no memset in front of
// the kernels that accumulate into them.
// keys_opt (fine grids: eight bit shifts per (search, PRN, Doppler) = eight adjacent threads, entry idx / 8): the packed key
// k_acq_keys would make of the triplets, while they are in registers
__device__ __forceinline__ void finalize_key8(const gpsx_peak_t &pk, size_t idx, int64_t *__restrict__ keys)
{
  const u32 b = (u32)(idx & 7);
  unsigned long long key = ((unsigned long long)pk.max_val << 14) | (unsigned long long)(16383u - (8u * pk.phase + b));
  u32 lo = (u32)key, hi = (u32)(key >> 32);
#define GPSX_MAX8(ctrl)                                                                         \
  {                                                                                             \
    const u32 lo2 = (u32)__builtin_amdgcn_mov_dpp((int)lo, ctrl, 0xF, 0xF, true);               \
    const u32 hi2 = (u32)__builtin_amdgcn_mov_dpp((int)hi, ctrl, 0xF, 0xF, true);               \
    const bool g = hi2 > hi || (hi2 == hi && lo2 > lo);                                         \
    lo = g ? lo2 : lo;                                                                          \
    hi = g ? hi2 : hi;                                                                          \
  }
  GPSX_MAX8(0xB1)    // quad_perm [1, 0, 3, 2]
  GPSX_MAX8(0x4E)    // quad_perm [2, 3, 0, 1]
  GPSX_MAX8(0x141)   // row_half_mirror
#undef GPSX_MAX8
  if (b == 0)
    keys[idx >> 3] = (int64_t)(((unsigned long long)hi << 32) | lo);
}

__global__ void k_acq_finalize(u32 *__restrict__ keyacc, u32 *__restrict__ sumacc, size_t n, gpsx_peak_t *__restrict__ peaks,
                               int64_t *__restrict__ keys_opt)
{
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n)   // (n is a multiple of eight wherever keys_opt is given: whole groups leave)
    return;
  const u32 k = keyacc[idx], t = sumacc[idx];
  keyacc[idx] = 0u;
  sumacc[idx] = 0u;
  gpsx_peak_t pk;
  pk.max_val = k >> 11;
  pk.phase = pk.max_val ? 2047u - (k & 2047u) : 0u;
  pk.sum = t;
  pk.avr = t / (2u * kChips);
  peaks[idx] = pk;
  if (keys_opt)
    finalize_key8(pk, idx, keys_opt);
}

// The same for the peaks of clusters >= cluster_from only (a launch whose last, partly filled round went to the split form:
// the full rounds' workgroups wrote their triplets themselves).  Peak idx = ((search n_prn + prn) n_dopp + dopp) n_bits + b;
// cluster = (search n_dopp + dopp) n_sets + prn / 32.
__global__ void k_acq_finalize_from(u32 *__restrict__ keyacc, u32 *__restrict__ sumacc, size_t first, size_t n,
                                    gpsx_peak_t *__restrict__ peaks, int n_prn, int n_dopp, int n_bits, int n_sets, int cluster_from,
                                    int64_t *__restrict__ keys_opt)
{
  const size_t idx = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n)
    return;
  const size_t pd = idx / (size_t)n_bits;
  const int dopp = (int)(pd % (size_t)n_dopp);
  const size_t sp = pd / (size_t)n_dopp;
  const int prn = (int)(sp % (size_t)n_prn), search = (int)(sp / (size_t)n_prn);
  if ((search * n_dopp + dopp) * n_sets + prn / 32 < cluster_from)
    return;
  const u32 k = keyacc[idx], t = sumacc[idx];
  keyacc[idx] = 0u;
  sumacc[idx] = 0u;
  gpsx_peak_t pk;
  pk.max_val = k >> 11;
  pk.phase = pk.max_val ? 2047u - (k & 2047u) : 0u;
  pk.sum = t;
  pk.avr = t / (2u * kChips);
  peaks[idx] = pk;
  if (keys_opt)   // (n_bits = 8 there: a cluster holds whole groups of eight)
    finalize_key8(pk, idx, keys_opt);
}

// Multi-block searches handled block-parallel (k_acq_poly<.., kPolyStore> wrote every block's magnitudes): sum over the
// blocks, then correlation_search's max / first argmax / sum over the window, per replica bit shift.  One workgroup per
// (search, PRN, Doppler) this shard owns; 32 KB per block read once, coalesced.
__global__ __launch_bounds__(kThreads) void k_acq_vals_search(const AcqParams prm, const uint16_t *__restrict__ vals,
                                                              gpsx_peak_t *__restrict__ peaks)
{
  __shared__ u32 red[4][8][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int id = blockIdx.x;
  const int dopp = id % prm.n_dopp;
  id /= prm.n_dopp;
  const int prn = id % prm.n_prn;
  const int search = id / prm.n_prn;
  const int unit = (search * prm.n_dopp + dopp) * prm.n_groups + prn / kAcqGroup;
  if (unit < prm.unit_lo || unit >= prm.unit_hi)
    return;
  const size_t plane = (size_t)prm.n_prn * prm.n_dopp * (16 * 1024);                      // one block's magnitudes
  const uint16_t *v0 = vals + ((size_t)(search * prm.n_ms) * prm.n_prn * prm.n_dopp + (size_t)prn * prm.n_dopp + dopp) * (16 * 1024);
#pragma unroll 1
  for (int b = 0; b < 8; b++) {
    u32 best = 0, total = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int t0 = b + 8 * half;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = 4 * tid + i, o = 2 * q + half;
        const bool in_win = q < kChips && o >= prm.win_start && o < prm.win_stop;
        u32 e = 0;
        for (int ms = 0; ms < prm.n_ms; ms++)
          e += v0[(size_t)ms * plane + (size_t)(t0 * 1024 + i * 256 + tid)];
        const u32 key = in_win ? (e << 11) | (u32)(2047 - o) : 0u;
        best = key > best ? key : best;
        total += in_win ? e : 0u;
      }
    }
    best = wave_max_to_lane63(best);
    total = wave_sum_to_lane63(total);
    if (lane == 63) {
      red[wave][b][0] = best;
      red[wave][b][1] = total;
    }
  }
  __syncthreads();
  if (tid < 8) {
    u32 k = 0, t = 0;
    for (int w = 0; w < 4; w++) {
      k = red[w][tid][0] > k ? red[w][tid][0] : k;
      t += red[w][tid][1];
    }
    gpsx_peak_t pk;
    pk.max_val = k >> 11;
    pk.phase = pk.max_val ? 2047u - (k & 2047u) : 0u;
    pk.sum = t;
    pk.avr = t / (2u * kChips);
    peaks[((size_t)(search * prm.n_prn + prn) * prm.n_dopp + dopp) * 8 + tid] = pk;
  }
}

void launch_acq_vals_search(hipStream_t s, const AcqParams &prm, const uint16_t *d_vals, gpsx_peak_t *d_peaks, size_t n_peaks)
{
  hipLaunchKernelGGL(k_acq_vals_search, dim3((unsigned)(n_peaks / 8)), dim3(kThreads), 0, s, prm, d_vals, d_peaks);
}

void launch_acq_finalize(hipStream_t s, uint32_t *d_keyacc, uint32_t *d_sumacc, size_t n_peaks,
                         gpsx_peak_t *d_peaks, int64_t *d_keys_opt)
{
  hipLaunchKernelGGL(k_acq_finalize, dim3((unsigned)((n_peaks + 255) / 256)), dim3(256), 0, s, d_keyacc, d_sumacc, n_peaks,
                     d_peaks, d_keys_opt);
}

void launch_acq_finalize_from(hipStream_t s, uint32_t *d_keyacc, uint32_t *d_sumacc, size_t first, size_t n_peaks,
                              gpsx_peak_t *d_peaks, int n_prn, int n_dopp, int n_bits, int n_sets, int cluster_from,
                              int64_t *d_keys_opt)
{
  if (first >= n_peaks)
    return;
  hipLaunchKernelGGL(k_acq_finalize_from, dim3((unsigned)((n_peaks - first + 255) / 256)), dim3(256), 0, s, d_keyacc, d_sumacc,
                     first, n_peaks, d_peaks, n_prn, n_dopp, n_bits, n_sets, cluster_from, d_keys_opt);
}

const char *launch_acq_poly(hipStream_t s, long local_units, const AcqParams &prm, const uint8_t *d_if, const uint32_t *d_cw8,
                     const uint32_t *d_chipbits, uint32_t *d_keyacc, uint32_t *d_sumacc, size_t n_peaks,
                     gpsx_peak_t *d_peaks, bool peaks_are_zero, uint32_t *d_energy, bool block_parallel,
                     int seg_force)
{
  if (local_units <= 0 || n_peaks == 0)
    return "";
  if (prm.n_ms > 1 && block_parallel) {
    // Few multi-block searches: a workgroup per (unit, block) instead of per unit walking its blocks, the blocks'
    // magnitudes through HBM (d_energy holds them as u16), one more small kernel to sum and search them.  Eight-offset
    // workgroups when that is still a small launch: no merge is needed here, every hypothesis is stored on its own.
    const long wg16 = local_units * kSuperGroups * prm.n_ms;
    if (wg16 >= 6 * 768)
      hipLaunchKernelGGL((k_acq_poly<kAcqGroup, 16, kPolyStore>), dim3((unsigned)wg16), dim3(kThreads), 0, s, prm, d_if, d_cw8,
                         d_chipbits, d_keyacc, d_sumacc, d_peaks, d_energy);
    else
      hipLaunchKernelGGL((k_acq_poly<kAcqGroup, 8, kPolyStore>), dim3((unsigned)(wg16 * 2)), dim3(kThreads), 0, s, prm, d_if,
                         d_cw8, d_chipbits, d_keyacc, d_sumacc, d_peaks, d_energy);
    hipLaunchKernelGGL(k_acq_vals_search, dim3((unsigned)(n_peaks / 8)), dim3(kThreads), 0, s, prm,
                       reinterpret_cast<const uint16_t *>(d_energy), d_peaks);
    return wg16 >= 6 * 768 ? "k_acq_poly<8,16,2>" : "k_acq_poly<8,8,2>";
  }
  if (prm.n_ms > 1) {
    // Non-coherent integration: always one workgroup per chip -- the energies of a (PRN, Doppler) pair then have one
    // owner, which walks the blocks itself and keeps the running sums in its own 64 KB-per-PRN slice of d_energy.
    hipLaunchKernelGGL((k_acq_poly<kAcqGroup, 16, kPolyMulti>), dim3((unsigned)(local_units * kSuperGroups)), dim3(kThreads), 0,
                       s, prm, d_if, d_cw8, d_chipbits, d_keyacc, d_sumacc, d_peaks, d_energy);
    return "k_acq_poly<8,16,1>";
  }
  // One workgroup per chip (16 offsets: one direct step + 15 recurrence steps; results merged in LDS and written once)
  // when that still leaves several waves of workgroups per CU slot; otherwise two (8 offsets each) or, for launches of
  // a capture or two, four (4 offsets each), merged through global atomics on two scratch planes and converted by
  // k_acq_finalize -- balance and latency against the extra direct steps.
  const long wg16 = local_units * kSuperGroups;
  const int seg = seg_force ? seg_force : (wg16 >= 6 * 768 ? 16 : (wg16 >= 768 ? 8 : 4));   // ($GPSX_ACQ_SEG forces one)
  if (seg == 16) {
    (void)peaks_are_zero;   // units of other shards keep whatever the caller zeroed
    hipLaunchKernelGGL((k_acq_poly<kAcqGroup, 16, kPolySingle>), dim3((unsigned)wg16), dim3(kThreads), 0, s, prm, d_if, d_cw8,
                       d_chipbits, d_keyacc, d_sumacc, d_peaks, (u32 *)nullptr);
    return "k_acq_poly<8,16,0>";
  }
  // (d_sumacc = d_keyacc + n_peaks; the planes are all-zero between launches: k_acq_finalize puts back what it reads)
  if (seg == 4)
    hipLaunchKernelGGL((k_acq_poly<kAcqGroup, 4, kPolySingle>), dim3((unsigned)(wg16 * 4)), dim3(kThreads), 0, s, prm, d_if, d_cw8,
                       d_chipbits, d_keyacc, d_sumacc, d_peaks, (u32 *)nullptr);
  else
    hipLaunchKernelGGL((k_acq_poly<kAcqGroup, 8, kPolySingle>), dim3((unsigned)(wg16 * 2)), dim3(kThreads), 0, s, prm, d_if, d_cw8,
                       d_chipbits, d_keyacc, d_sumacc, d_peaks, (u32 *)nullptr);
  hipLaunchKernelGGL(k_acq_finalize, dim3((unsigned)((n_peaks + 255) / 256)), dim3(256), 0, s, d_keyacc, d_sumacc, n_peaks,
                     d_peaks, (int64_t *)nullptr);
  return seg == 4 ? "k_acq_poly<8,4,0>" : "k_acq_poly<8,8,0>";
}

}  // namespace gpsx
