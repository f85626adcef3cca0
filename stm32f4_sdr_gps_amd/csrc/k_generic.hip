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
#include <cstdlib>

#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"
#include "gpsx_track_wave.hpp"

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
// K5, many channels (from 2048 on): one WAVE per tracking channel and millisecond, `cpw` channels per wave one after the
// other, four waves per workgroup.
//   fine = (int16) code_phase_fine;  replica shift b = fine & 7;  prompt offset = fine / 8, early = prompt - 1
//   (wraps to 2045), late = prompt + 1 (wraps to 0)                                   PM/GPS/tracking.c:115-130
//   carrier NCO continues from if_freq_accum at (float)IF + if_freq_offset_hz and is stored back  gps_misc.c:244-274
// Round 3 formulation (r2's kernel: 523 vector instructions per channel, vector-issue bound):
//  * gps_mult_and_summ at byte offset o pairs replica bit s with wiped sample (8 o + s) mod 16368
//    (PM/GPS/gps_misc.c:60-90): the count is  pop(D ^ rot(R, -8 o))  -- the DATA stay where they are, word-aligned, in the
//    registers of the lane that mixed them (word j = x[j] ^ carrier(quadrant of acc + j step): no LDS round trip, no second
//    period, no funnel shift per stream and offset), and the REPLICA is what gets rotated: one stream for I and Q.
//  * The replica as a circular bit stream comes from a table in global memory (k_build_track_rep: per PRN the chips
//    expanded to 16 samples, two and a bit periods back to back so that any window of one period is contiguous; 4 KB per
//    PRN, L2-resident): a lane reads the six words that cover its four for all three offsets (Early / Prompt / Late sit 8
//    samples apart) and cuts them with 5 + 4 + 4 funnel shifts -- no replica arithmetic at all.
//  * What the reference's quirks change against that circular count touches at most three 16-bit words per offset: the
//    first b samples of the non-circular replica shift (Q5) and the two words odd byte offsets skip (Q3).  They are not
//    computed per channel by a whole wave: lane 3 c + k of the wave computes them for (channel c, offset k) after the
//    loop -- as it computed the offsets, NCO step and table positions before it -- so every wave-uniform quantity of a
//    channel costs 1/cpw of an instruction, and the results leave in one coalesced store.
//  Per channel: 8 words x (5 wipe-off + 12 xor/popcount) + 26 funnel shifts + 3 reductions; see DESIGN.md 4.3.
// The table: bit p of a PRN's row = chip[(p mod 16368) >> 4], kTrackRepWords words (+ padding to the row stride).
constexpr int kTrackRepWords = 1028;    // 2 periods (1023 words) + what the last windows read past them
static_assert(kTrackRepWords <= kTrackRepStride, "row stride");

__global__ void k_build_track_rep(const u32 *__restrict__ chipbits_all, int n_slots, u32 *__restrict__ rep)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_slots * kTrackRepStride)
    return;
  const int slot = idx / kTrackRepStride, w = idx - slot * kTrackRepStride;
  u32 v = 0;
  if (w < kTrackRepWords) {
    const u32 *cb = chipbits_all + (size_t)slot * 32;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int c = ((32 * w + 16 * half) % kSamples) >> 4;   // word boundaries are chip boundaries in every period
      if ((cb[c >> 5] >> (c & 31)) & 1u)
        v |= 0xFFFFu << (16 * half);
    }
  }
  rep[idx] = v;
}

void launch_build_track_rep(hipStream_t s, const uint32_t *d_chipbits_all, int n_slots, uint32_t *d_rep)
{
  const int n = n_slots * kTrackRepStride;
  hipLaunchKernelGGL(k_build_track_rep, dim3((n + 255) / 256), dim3(256), 0, s, d_chipbits_all, n_slots, d_rep);
}

__global__ __launch_bounds__(256) void k_track_epl_wave(const uint8_t *__restrict__ if_block, int if_format, int if_hz,
                                                        gpsx_trk_state_t *__restrict__ st, int n_ch, int cpw,
                                                        const u32 *__restrict__ chipbits_all, const u32 *__restrict__ rep_all,
                                                        int16_t *__restrict__ iq_out, u32 *__restrict__ bad_prn)
{
  __shared__ u32 s_x[512];
  __shared__ uint2 s_carrier[4];   // (in-phase, quadrature) carrier word per NCO quadrant: one LDS read instead of two selects
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int ch0 = ((int)blockIdx.x * 4 + wave) * cpw;   // this wave's channels: ch0 .. ch0 + cpw - 1
  trkwave::stage_block(if_block, if_format, s_x, s_carrier);   // the block's sign plane, once per workgroup
  __syncthreads();
  if (ch0 >= n_ch)   // (wave-uniform: an idle wave of the last workgroup)
    return;
  const int n_here = min(cpw, n_ch - ch0);

  // ---- per (channel, offset): lane 4 c + k, k = 0 / 1 / 2 = Early / Prompt / Late (k = 3 idles) --------------------------
  const int c_l = lane >> 2, k_l = lane & 3;
  const bool mine = c_l < n_here && k_l < 3;
  const int ch_l = ch0 + (c_l < n_here ? c_l : 0);
  const gpsx_trk_state_t state = st[ch_l];
  const int prn = track_prn(state.prn, bad_prn, mine && k_l == 0);
  const int fine = (int)(int16_t)(int)state.code_phase_fine;
  const u32 step = nco_step_per_word((float)if_hz + state.if_freq_offset_hz);
  const u32 iq = trkwave::wave_epl(s_x, s_carrier, lane, n_here, mine, prn, fine, step, state.if_freq_accum, chipbits_all, rep_all);
  if (!mine)
    return;
  reinterpret_cast<u32 *>(iq_out)[(size_t)ch_l * 3 + k_l] = iq;   // (IE,QE) (IP,QP) (IL,QL)
  if (k_l == 0)
    st[ch_l].if_freq_accum = state.if_freq_accum + step * (u32)kWords32;
}

// wave_from: channel count from which the wave-per-channel kernel serves the step.  Round 3's k_track_epl_wave is the faster
// one at every count (4 channels: 3.3 us against 6.2 us; 2047: 4.5 against 9.4), so contexts use it from 1 channel on;
// k_track_epl -- one workgroup per channel, the reference's 16-bit-word indexing -- stays as the second implementation the
// tests compare it with ($GPSX_TRACK_WAVE_FROM).
void launch_track_epl(hipStream_t s, const uint8_t *d_if_block, int if_format, int if_hz, gpsx_trk_state_t *d_st, int n_ch,
                      const uint8_t *d_chips, const uint32_t *d_chipbits, const uint32_t *d_trk_rep, int16_t *d_iq,
                      uint32_t *d_bad_prn, int wave_from)
{
  if (n_ch <= 0)
    return;
  if (n_ch < wave_from) {
    hipLaunchKernelGGL(k_track_epl, dim3(n_ch), dim3(256), 0, s, d_if_block, if_format, if_hz, d_st, n_ch, d_chips, d_iq,
                       d_bad_prn);
    return;
  }
  // channels per wave: as many as leave ~4 workgroups per CU (16 at most: lanes 3 c + k carry the per-channel values)
  int cpw = n_ch / (4 * 256 * 4);
  cpw = cpw < 1 ? 1 : (cpw > 16 ? 16 : cpw);
  hipLaunchKernelGGL(k_track_epl_wave, dim3((n_ch + 4 * cpw - 1) / (4 * cpw)), dim3(256), 0, s, d_if_block, if_format, if_hz,
                     d_st, n_ch, cpw, d_chipbits, d_trk_rep, d_iq, d_bad_prn);
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
