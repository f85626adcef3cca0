// k_generic.hip -- the correlator primitives on caller-shaped buffers, and the E/P/L tracking kernel (K5).
//
// These kernels keep the reference's own data layout (2046-byte I/Q buffers, 1023-word replica) so that they are
// exact for ARBITRARY buffer contents, which the per-call interface (include/gpsx_compat.h) must be:
//   k_wipeoff       gps_shift_to_zero_freq / _track   PM/GPS/gps_misc.c:211-274
//   k_replica       gps_generate_prn_data2            PM/GPS/gps_misc.c:282-300
//   k_corr_offsets  gps_mult_and_summ + gps_correlation8 + gps_correlation_iq   PM/GPS/gps_misc.c:48-145
//   k_search_reduce correlation_search's max / argmax / average                 PM/GPS/gps_misc.c:155-191
//   k_track_epl     the correlator half of gps_tracking_data_process            PM/GPS/tracking.c:115-138
//   k_rewind        gps_rewind_if_phase                                         PM/GPS/gps_misc.c:196-204
// XOR + v_bcnt_u32_b32 on 16-bit words, wave64 shuffle reductions for the I/Q sums.
#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"

namespace gpsx {

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_wipeoff(const uint8_t *__restrict__ signal, float freq_hz, u32 accum_in,
                                                 uint8_t *__restrict__ out_i, uint8_t *__restrict__ out_q,
                                                 u32 *__restrict__ accum_out)
{
  const int w = threadIdx.x;
  const u32 step = nco_step_per_word(freq_hz);
  if (w < kWords32) {
    const u32 quad = (accum_in + step * (u32)w) >> 30;
    // byte-wise so that the (2-byte aligned at best) caller buffers need no alignment
    const u32 ci = carrier_i(quad), cq = carrier_q(quad);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint8_t x = signal[4 * w + k];
      out_i[4 * w + k] = x ^ (uint8_t)(ci >> (8 * k));
      out_q[4 * w + k] = x ^ (uint8_t)(cq >> (8 * k));
    }
  }
  if (w == 0)
    *accum_out = accum_in + step * (u32)kWords32;
}

void launch_wipeoff(hipStream_t s, const uint8_t *d_signal, float freq_hz, uint32_t accum_in, uint8_t *d_i,
                    uint8_t *d_q, uint32_t *d_accum_out)
{
  hipLaunchKernelGGL(k_wipeoff, dim3(1), dim3(512), 0, s, d_signal, freq_hz, accum_in, d_i, d_q, d_accum_out);
}

// ---------------------------------------------------------------------------------------------------------------
// out[i] (i < 1023) = bits [16 i, 16 i + 16) of the stream r(n) = n < b ? 0 : chip[(n - b) >> 4];
// out[1023] |= chip 1022's spill (the reference never clears the pad word).
__global__ __launch_bounds__(256) void k_replica(const uint8_t *__restrict__ chips, unsigned offset_bits,
                                                 uint16_t *__restrict__ out)
{
  const u32 b = offset_bits & 15u;
  const u32 low = (1u << b) - 1u, high = (0xFFFFu << b) & 0xFFFFu;
  for (int i = threadIdx.x; i <= kWords16; i += 256) {
    const u32 prev = i > 0 ? chips[i - 1] : 0u;
    if (i < kWords16)
      out[i] = (uint16_t)((prev ? low : 0u) | (chips[i] ? high : 0u));
    else
      out[i] = (uint16_t)(out[i] | (prev ? low : 0u));
  }
}

void launch_replica(hipStream_t s, const uint8_t *d_chips, unsigned offset_bits, uint16_t *d_out)
{
  hipLaunchKernelGGL(k_replica, dim3(1), dim3(256), 0, s, d_chips, offset_bits, d_out);
}

// ---------------------------------------------------------------------------------------------------------------
// gps_mult_and_summ for one byte offset, computed by one wave.  Replica word i meets data bytes
// (o + 2 i) mod 2046 and the following byte; odd offsets skip the word that straddles the buffer wrap,
// p1 = (2045 - o) / 2, and the last word, 1022 (PM/GPS/gps_misc.c:60-90).  Offset 2046 behaves as 0.
template <typename Bytes>
__device__ __forceinline__ void corr_one_offset(const Bytes &di, const Bytes &dq, const Bytes &rep, int offset, int lane,
                                                u32 &cnt_i, u32 &cnt_q)
{
  const int o = offset >= kBytes ? offset - kBytes : offset;
  const int odd = o & 1;
  const int p1 = (kBytes - o) >> 1;
  u32 ci = 0, cq = 0;
  for (int i = lane; i < kWords16; i += 64) {
    if (odd && (i == p1 || i == kWords16 - 1))
      continue;
    int d0 = o + 2 * i;
    d0 = d0 >= kBytes ? d0 - kBytes : d0;
    const int d1 = d0 + 1 >= kBytes ? 0 : d0 + 1;
    const u32 r = (u32)rep[2 * i] | ((u32)rep[2 * i + 1] << 8);
    ci += pop16(((u32)di[d0] | ((u32)di[d1] << 8)) ^ r);
    cq += pop16(((u32)dq[d0] | ((u32)dq[d1] << 8)) ^ r);
  }
  cnt_i = wave_sum_u32(ci);
  cnt_q = wave_sum_u32(cq);
}

__global__ __launch_bounds__(256) void k_corr_offsets(const uint8_t *__restrict__ rep, const uint8_t *__restrict__ di,
                                                      const uint8_t *__restrict__ dq,
                                                      const uint16_t *__restrict__ offsets, int first_offset, int n,
                                                      uint16_t *__restrict__ cnt_i, uint16_t *__restrict__ cnt_q,
                                                      int16_t *__restrict__ corr8)
{
  __shared__ uint8_t s_rep[2048], s_i[2048], s_q[2048];
  for (int i = threadIdx.x; i < kBytes; i += 256) {
    s_rep[i] = rep[i];
    s_i[i] = di[i];
    s_q[i] = dq[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= n)
    return;
  const int offset = offsets ? offsets[idx] : first_offset + idx;
  u32 ci, cq;
  corr_one_offset(s_i, s_q, s_rep, offset, lane, ci, cq);
  if (lane == 0) {
    if (cnt_i) cnt_i[idx] = (uint16_t)ci;
    if (cnt_q) cnt_q[idx] = (uint16_t)cq;
    if (corr8) corr8[idx] = (int16_t)mag8((int)ci, (int)cq);
  }
}

void launch_corr_offsets(hipStream_t s, const uint8_t *d_rep, const uint8_t *d_i, const uint8_t *d_q,
                         const uint16_t *d_offsets, int first_offset, int n, uint16_t *d_cnt_i, uint16_t *d_cnt_q,
                         int16_t *d_corr8)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(k_corr_offsets, dim3((n + 3) / 4), dim3(256), 0, s, d_rep, d_i, d_q, d_offsets, first_offset, n,
                     d_cnt_i, d_cnt_q, d_corr8);
}

// out = the generic magnitude; *disagree counts inputs where the grid kernel's trimmed variant (mag8_fast) differs
__global__ void k_mag8(const uint16_t *__restrict__ cnt_i, const uint16_t *__restrict__ cnt_q, int n,
                       int16_t *__restrict__ out, u32 *__restrict__ disagree)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) {
    const int a = mag8((int)cnt_i[idx], (int)cnt_q[idx]);
    const int b = mag8_fast((int)cnt_i[idx], (int)cnt_q[idx]);
    out[idx] = (int16_t)a;
    if (a != b)
      atomicAdd(disagree, 1u);
  }
}

void launch_mag8(hipStream_t s, const uint16_t *d_cnt_i, const uint16_t *d_cnt_q, int n, int16_t *d_out,
                 uint32_t *d_disagree)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(k_mag8, dim3((n + 255) / 256), dim3(256), 0, s, d_cnt_i, d_cnt_q, n, d_out, d_disagree);
}

// max (strict: first maximum wins, nothing above 0 leaves phase 0), sum, sum / 2046 over n consecutive offsets
__global__ __launch_bounds__(256) void k_search_reduce(const int16_t *__restrict__ corr8, int n, int first_offset,
                                                       gpsx_peak_t *__restrict__ peak)
{
  __shared__ u32 red[4][2];
  u32 best = 0, total = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const u32 v = (u32)(int)corr8[i];
    const u32 key = (v << 11) | (u32)(2047 - (first_offset + i));
    best = key > best ? key : best;
    total += v;
  }
  best = wave_max_u32(best);
  total = wave_sum_u32(total);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6][0] = best;
    red[threadIdx.x >> 6][1] = total;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 k = 0, t = 0;
    for (int w = 0; w < 4; w++) {
      k = red[w][0] > k ? red[w][0] : k;
      t += red[w][1];
    }
    gpsx_peak_t pk;
    pk.max_val = k >> 11;
    pk.phase = pk.max_val ? 2047u - (k & 2047u) : 0u;
    pk.sum = t;
    pk.avr = t / (2u * kChips);
    *peak = pk;
  }
}

void launch_search_reduce(hipStream_t s, const int16_t *d_corr8, int n, int first_offset, gpsx_peak_t *d_peak)
{
  hipLaunchKernelGGL(k_search_reduce, dim3(1), dim3(256), 0, s, d_corr8, n, first_offset, d_peak);
}

// ---------------------------------------------------------------------------------------------------------------
// K5: one workgroup (4 waves) per tracking channel and millisecond.
//   fine = (int16) code_phase_fine;  replica shift = fine & 7;  prompt offset = fine / 8, early = prompt - 1
//   (wraps to 2045), late = prompt + 1 (wraps to 0)                                   PM/GPS/tracking.c:115-130
//   carrier NCO continues from if_freq_accum at (float)IF + if_freq_offset_hz and is stored back  gps_misc.c:244-274
// The PRN of a channel state as the tracking kernels use it.  Validation lives here, not in a host loop over the states (at
// 400 000 channels that loop is a sixth of the millisecond): a PRN outside 1..210 correlates against the empty code and
// raises *bad_prn -- page-locked host memory the step call looks at after its wait.  kTrackPadPrn marks the padding
// channels of a captured step (gpsx_api.hip: graphs are cached per capacity) and raises nothing.
__device__ __forceinline__ int track_prn(int prn, u32 *bad_prn, bool reporter)
{
  if (prn >= 1 && prn <= GPSX_MAX_PRN)
    return prn;
  if (prn != kTrackPadPrn && reporter && bad_prn)
    *bad_prn = 1u;
  return 0;
}

// Wave w < 3 computes offset E/P/L; the IF block is read once per channel with coalesced 16-bit loads.
__global__ __launch_bounds__(256) void k_track_epl(const uint8_t *__restrict__ if_block, int if_format, int if_hz,
                                                   gpsx_trk_state_t *__restrict__ st, int n_ch,
                                                   const uint8_t *__restrict__ chips_all, int16_t *__restrict__ iq_out,
                                                   u32 *__restrict__ bad_prn)
{
  __shared__ __attribute__((aligned(4))) uint8_t s_i[2048];
  __shared__ __attribute__((aligned(4))) uint8_t s_q[2048];
  __shared__ __attribute__((aligned(4))) uint8_t s_rep[2048];
  __shared__ uint16_t s_x[1024];
  const int ch = blockIdx.x;
  if (ch >= n_ch)
    return;
  const int tid = threadIdx.x;
  const gpsx_trk_state_t state = st[ch];
  const int prn = track_prn(state.prn, bad_prn, tid == 0);
  const uint8_t *chips = chips_all + (size_t)prn * 1024;

  const int fine = (int)(int16_t)(int)state.code_phase_fine;
  const u32 b = (u32)fine & 7u;
  const u32 low = (1u << b) - 1u, high = (0xFFFFu << b) & 0xFFFFu;
  const float freq_hz = (float)if_hz + state.if_freq_offset_hz;
  const u32 step = nco_step_per_word(freq_hz);

  for (int i = tid; i < 1024; i += 256) {
    s_x[i] = i < kWords16 ? load_sign16(if_block, i, if_format) : (uint16_t)0;
    // replica word i (K2)
    const u32 prev = (i > 0 && i <= kWords16) ? chips[i - 1] : 0u;
    const u32 cur = i < kWords16 ? chips[i] : 0u;
    reinterpret_cast<uint16_t *>(s_rep)[i] = (uint16_t)((prev ? low : 0u) | (cur ? high : 0u));
  }
  __syncthreads();
  const u32 *x32 = reinterpret_cast<const u32 *>(s_x);
  for (int w = tid; w < 512; w += 256) {
    u32 vi = 0, vq = 0;  // word 511 holds the 16 samples the NCO loop never mixes: zero in the firmware's buffers
    if (w < kWords32) {
      const u32 quad = (state.if_freq_accum + step * (u32)w) >> 30;
      vi = carrier_i(quad) ^ x32[w];
      vq = carrier_q(quad) ^ x32[w];
    }
    reinterpret_cast<u32 *>(s_i)[w] = vi;
    reinterpret_cast<u32 *>(s_q)[w] = vq;
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  if (wave < 3) {
    const unsigned prompt = (unsigned)(uint16_t)(fine / 8);
    unsigned off = wave == 0 ? (unsigned)(uint16_t)(prompt - 1u) : (wave == 1 ? prompt : (unsigned)(uint16_t)(prompt + 1u));
    if (wave == 0 && off >= 2u * kChips) off = 2u * kChips - 1u;
    if (wave == 2 && off >= 2u * kChips) off = 0u;
    if (off > 2u * kChips) off = 0u;  // the reference would read out of bounds here; keep the access in range
    u32 ci, cq;
    corr_one_offset(s_i, s_q, s_rep, (int)off, lane, ci, cq);
    if (lane == 0) {
      iq_out[ch * 6 + wave * 2 + 0] = (int16_t)((int)ci - kHalf);
      iq_out[ch * 6 + wave * 2 + 1] = (int16_t)((int)cq - kHalf);
    }
  }
  if (tid == 0)
    st[ch].if_freq_accum = state.if_freq_accum + step * (u32)kWords32;
}

// ---------------------------------------------------------------------------------------------------------------
// K5, many channels: one WAVE per tracking channel and millisecond, four channels per workgroup (from 2048 channels on).
//   fine = (int16) code_phase_fine;  replica shift b = fine & 7;  prompt offset = fine / 8, early = prompt - 1
//   (wraps to 2045), late = prompt + 1 (wraps to 0)                                   PM/GPS/tracking.c:115-130
//   carrier NCO continues from if_freq_accum at (float)IF + if_freq_offset_hz and is stored back  gps_misc.c:244-274
// Sample-stream formulation: gps_mult_and_summ at byte offset o pairs replica word i with data bytes (o + 2 i) mod 2046
// (PM/GPS/gps_misc.c:60-90), i.e. replica BIT s with wiped sample (8 o + s) mod 16368 -- a rotation of the stream.  The
// count over all 1023 words is therefore  pop(rot(D, 8 o) ^ R)  on 32-bit words: 512 XOR + popcount pairs per stream and
// offset where the 16-bit form needs 1023 (plus their masks).  R, the replica as a bit stream -- chips delayed by b
// samples, its first b bits zero (quirk Q5) -- is never stored: word j is cut from chips 2 j - 1 .. 2 j + 1 of the
// packed code.  The wiped stream sits in LDS as 32-bit words with three words of continuation past the wrap (16368 =
// 511.5 words: after the wrap the stream is 16 bits out of step), so any window starting before the wrap is two
// neighbouring words.  Odd offsets then take back the two words the reference skips: p1 = (2045 - o) / 2, which pairs with
// data bytes (2045, 0), and word 1022 (unless it is p1).  The four channels of a workgroup share one staging of the
// block's sign plane (the 2-bit unpack happens once per workgroup, not once per channel).
// r1 -> r2: 1231 -> 546 vector instructions per channel, 408 -> 223 us for 212 992 channels (profiles/r02_track_*).
namespace {

struct TrackLds {
  // the wiped streams as (I, Q) word pairs, two periods back to back (16368 samples = 511.5 words: the second period sits 16
  // bits out of step), one zero pair in front: pair 1 + w = word w; words 0..510 mixed, the low half of 511 = the sixteen
  // never-mixed (zero) samples, its high half = samples 0..15 again, and so on to word 1022.  Any 32-bit window of the
  // circular stream that starts in the first period is two neighbouring pairs: one 16-byte read serves I and Q.
  uint2 dd[1026];
  u32 cb[36];       // packed chips with one zero word in front (chip -1 = 0) and zeros behind
};

__device__ __forceinline__ uint4 lds_read_pairs(const uint2 *p)   // p[0], p[1]: 8-byte aligned, one LDS instruction
{
  uint4 v;
  __builtin_memcpy(&v, __builtin_assume_aligned(p, 8), 16);
  return v;
}

}  // namespace

__global__ __launch_bounds__(256) void k_track_epl_wave(const uint8_t *__restrict__ if_block, int if_format, int if_hz,
                                                        gpsx_trk_state_t *__restrict__ st, int n_ch,
                                                        const u32 *__restrict__ chipbits_all,
                                                        int16_t *__restrict__ iq_out, u32 *__restrict__ bad_prn)
{
  __shared__ u32 s_x[512];
  __shared__ uint2 s_carrier[4];   // (in-phase, quadrature) carrier word per NCO quadrant: one LDS read instead of two selects
  __shared__ TrackLds lds[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ch_raw = blockIdx.x * 4 + wave;
  const bool live = ch_raw < n_ch;
  const int ch = live ? ch_raw : n_ch - 1;   // idle waves of the last workgroup shadow a real channel (no early exit
                                             // before the barrier) and write nothing
  TrackLds &L = lds[wave];
  if (threadIdx.x < 4)
    s_carrier[threadIdx.x] = uint2{carrier_i(threadIdx.x), carrier_q(threadIdx.x)};
  const gpsx_trk_state_t state = st[ch];
  const int prn = track_prn(state.prn, bad_prn, live && lane == 0);

  // the block's sign plane as 32-bit words, once per workgroup (word 511 = 16-bit word 1022 alone)
  for (int w = threadIdx.x; w < 512; w += 256) {
    const u32 lo = load_sign16(if_block, 2 * w, if_format);
    const u32 hi = 2 * w + 1 < kWords16 ? (u32)load_sign16(if_block, 2 * w + 1, if_format) : 0u;
    s_x[w] = lo | (hi << 16);
  }
  if (lane < 36)
    L.cb[lane] = (lane >= 1 && lane <= 32) ? chipbits_all[(size_t)prn * 32 + (lane - 1)] : 0u;
  if (lane == 0)
    L.dd[0] = uint2{0u, 0u};
  __syncthreads();

  const int fine = (int)(int16_t)(int)state.code_phase_fine;
  const u32 b = (u32)fine & 7u;
  const u32 low = (1u << b) - 1u, high = (0xFFFFu << b) & 0xFFFFu;
  const float freq_hz = (float)if_hz + state.if_freq_offset_hz;
  const u32 step = nco_step_per_word(freq_hz);

  // K3: wipe-off, word w sees NCO phase accum + w * step
  {
    u32 acc = state.if_freq_accum + step * (u32)lane;
    const u32 step64 = step * 64u;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int w = lane + 64 * it;
      u32 vi = 0, vq = 0;   // word 511: the 16 samples the NCO loop never mixes read as zero (PM/GPS/gps_misc.c:229,261)
      if (w < kWords32) {
        const u32 x = s_x[w];
        const uint2 c = s_carrier[acc >> 30];
        vi = c.x ^ x;
        vq = c.y ^ x;
      }
      L.dd[1 + w] = uint2{vi, vq};
      acc += step64;
    }
    // the second period: word 511 + m = word (m - 1) >> 16 | word m << 16, m = 0..511 (word -1 = the zero pair in front, word
    // 511 of the first period = zero: its low half is all this needs, and its slot is being overwritten meanwhile)
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int m = lane + 64 * it;
      uint4 v = lds_read_pairs(&L.dd[m]);
      if (it == 7 && lane == 63)
        v.z = v.w = 0u;
      L.dd[1 + kWords32 + m] = uint2{__builtin_amdgcn_alignbit(v.z, v.x, 16u), __builtin_amdgcn_alignbit(v.w, v.y, 16u)};
    }
  }

  // offsets exactly as tracking.c forms them, then reduced to the circle
  const unsigned prompt = (unsigned)(uint16_t)(fine / 8);
  unsigned off[3] = {(unsigned)(uint16_t)(prompt - 1u), prompt, (unsigned)(uint16_t)(prompt + 1u)};
  if (off[0] >= 2u * kChips) off[0] = 2u * kChips - 1u;
  if (off[2] >= 2u * kChips) off[2] = 0u;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (off[k] > 2u * kChips) off[k] = 0u;   // the reference would read out of bounds here; keep the access in range
    if (off[k] == 2u * kChips) off[k] = 0u;  // offset 2046 behaves as 0
  }

  // stream position of replica bit 32 lane at each offset: the window of iteration `it` starts 2048 it bits further on
  const uint2 *win[3];
  u32 sh[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int t = 8 * (int)off[k] + 32 * lane;
    win[k] = &L.dd[1 + (t >> 5)];
    sh[k] = (u32)(t & 31);
  }
  u32 ci[3] = {0, 0, 0}, cq[3] = {0, 0, 0};
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int j = lane + 64 * it;
    // replica word j = bits [32 j, 32 j + 32) of the delayed chip stream: chips 2 j - 1 (low b bits), 2 j, 2 j + 1
    const int cbit = 2 * j - 1 + 32;
    const u32 cw = __builtin_amdgcn_alignbit(L.cb[(cbit >> 5) + 1], L.cb[cbit >> 5], (u32)(cbit & 31));
    // (an 8-entry LDS table of the replica words there are was measured 6 % slower: the lookup sits on the critical path)
    u32 r = ((cw & 1u) ? low : 0u) | ((cw & 2u) ? (high | (low << 16)) : 0u) | ((cw & 4u) ? (high << 16) : 0u);
    const u32 m = (it == 7 && lane == 63) ? 0xFFFFu : 0xFFFFFFFFu;   // word 511 is half a word (16-bit word 1022)
    r &= m;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const uint4 v = lds_read_pairs(win[k] + 64 * it);
      const u32 wi = __builtin_amdgcn_alignbit(v.z, v.x, sh[k]);
      const u32 wq = __builtin_amdgcn_alignbit(v.w, v.y, sh[k]);
      ci[k] += (u32)__popc((wi & m) ^ r);
      cq[k] += (u32)__popc((wq & m) ^ r);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    // (both counts in one reduction: a lane's share is at most 8 x 32, the totals stay below 2^15)
    const u32 both = wave_sum_to_lane63(ci[k] | (cq[k] << 16));
    u32 si = both & 0xFFFFu, sq = both >> 16;
    if (off[k] & 1u) {   // wave-uniform
      const int o = (int)off[k];
      const int p1 = (kBytes - o) >> 1;
      auto rep16 = [&](int i) {   // replica word i of the 16-bit form: chip i - 1 below bit b, chip i from bit b on
        const int cb0 = i - 1 + 32;
        const u32 c2 = __builtin_amdgcn_alignbit(L.cb[(cb0 >> 5) + 1], L.cb[cb0 >> 5], (u32)(cb0 & 31));
        return ((c2 & 1u) ? low : 0u) | ((c2 & 2u) ? high : 0u);
      };
      auto win16 = [&](int t, u32 &wi, u32 &wq) {
        const uint4 v = lds_read_pairs(&L.dd[1 + (t >> 5)]);
        wi = __builtin_amdgcn_alignbit(v.z, v.x, (u32)(t & 31)) & 0xFFFFu;
        wq = __builtin_amdgcn_alignbit(v.w, v.y, (u32)(t & 31)) & 0xFFFFu;
      };
      const u32 r1 = rep16(p1);
      u32 wi, wq;
      win16(kSamples - 8, wi, wq);   // data bytes (2045, 0)
      u32 sub_i = pop16(wi ^ r1), sub_q = pop16(wq ^ r1);
      if (p1 != kWords16 - 1) {
        const u32 r2 = rep16(kWords16 - 1);
        win16(8 * (o - 2), wi, wq);   // data bytes (o - 2, o - 1); o >= 3 here
        sub_i += pop16(wi ^ r2);
        sub_q += pop16(wq ^ r2);
      }
      si -= sub_i;
      sq -= sub_q;
    }
    if (lane == 63 && live) {
      iq_out[ch * 6 + k * 2 + 0] = (int16_t)((int)si - kHalf);
      iq_out[ch * 6 + k * 2 + 1] = (int16_t)((int)sq - kHalf);
    }
  }
  if (lane == 0 && live)
    st[ch].if_freq_accum = state.if_freq_accum + step * (u32)kWords32;
}

constexpr int kTrackWaveFormFrom = 2048;

void launch_track_epl(hipStream_t s, const uint8_t *d_if_block, int if_format, int if_hz, gpsx_trk_state_t *d_st, int n_ch,
                      const uint8_t *d_chips, const uint32_t *d_chipbits, int16_t *d_iq, uint32_t *d_bad_prn)
{
  if (n_ch <= 0)
    return;
  if (n_ch < kTrackWaveFormFrom)
    hipLaunchKernelGGL(k_track_epl, dim3(n_ch), dim3(256), 0, s, d_if_block, if_format, if_hz, d_st, n_ch, d_chips, d_iq,
                       d_bad_prn);
  else
    hipLaunchKernelGGL(k_track_epl_wave, dim3((n_ch + 3) / 4), dim3(256), 0, s, d_if_block, if_format, if_hz, d_st, n_ch,
                       d_chipbits, d_iq, d_bad_prn);
}

// N3 ingest: MAX2769 sign/magnitude pairs -> sign plane and magnitude plane in the reference's 1-bit layout.
// One thread per output 16-bit word (16 samples = 4 input bytes); loads and stores are coalesced.
__global__ void k_unpack2(const uint8_t *__restrict__ in, int n_blocks, uint8_t *__restrict__ sign,
                          uint8_t *__restrict__ mag)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_blocks * kWords16)
    return;
  const int blk = idx / kWords16, w = idx - blk * kWords16;
  const uint16_t *p = reinterpret_cast<const uint16_t *>(in + (size_t)blk * GPSX_BYTES_PER_MS_2BIT) + 2 * w;
  const u32 pairs = (u32)p[0] | ((u32)p[1] << 16);
  if (sign)
    reinterpret_cast<uint16_t *>(sign + (size_t)blk * kBytes)[w] = (uint16_t)even_bits16(pairs);
  if (mag)
    reinterpret_cast<uint16_t *>(mag + (size_t)blk * kBytes)[w] = (uint16_t)even_bits16(pairs >> 1);
}

void launch_unpack2(hipStream_t s, const uint8_t *d_in, int n_blocks, uint8_t *d_sign, uint8_t *d_mag)
{
  const int n = n_blocks * kWords16;
  if (n <= 0)
    return;
  hipLaunchKernelGGL(k_unpack2, dim3((n + 255) / 256), dim3(256), 0, s, d_in, n_blocks, d_sign, d_mag);
}

// gps_rewind_if_phase: accum += (uint32)((uint64)step_per_sample * 16368 * steps)
__global__ void k_rewind(int if_hz, gpsx_trk_state_t *__restrict__ st, int n_ch, const uint8_t *__restrict__ steps)
{
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= n_ch)
    return;
  const u32 step = nco_step_per_sample((float)if_hz + st[ch].if_freq_offset_hz);
  const u64 adv = (u64)step * (u64)kSamples * (u64)steps[ch];
  st[ch].if_freq_accum += (u32)adv;
}

void launch_rewind(hipStream_t s, int if_hz, gpsx_trk_state_t *d_st, int n_ch, const uint8_t *d_steps)
{
  if (n_ch <= 0)
    return;
  hipLaunchKernelGGL(k_rewind, dim3((n_ch + 63) / 64), dim3(64), 0, s, if_hz, d_st, n_ch, d_steps);
}

}  // namespace gpsx
