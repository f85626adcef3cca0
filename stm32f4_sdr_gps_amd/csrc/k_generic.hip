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
// Wave w < 3 computes offset E/P/L; the IF block is read once per channel with coalesced 16-bit loads.
__global__ __launch_bounds__(256) void k_track_epl(const uint8_t *__restrict__ if_block, int if_format,
                                                   gpsx_trk_state_t *__restrict__ st, int n_ch,
                                                   const uint8_t *__restrict__ chips_all, int16_t *__restrict__ iq_out)
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
  const int prn = state.prn >= 0 && state.prn <= GPSX_MAX_PRN ? state.prn : 0;
  const uint8_t *chips = chips_all + (size_t)prn * 1024;

  const int fine = (int)(int16_t)(int)state.code_phase_fine;
  const u32 b = (u32)fine & 7u;
  const u32 low = (1u << b) - 1u, high = (0xFFFFu << b) & 0xFFFFu;
  const float freq_hz = (float)kIfHz + state.if_freq_offset_hz;
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
// K5, many channels: one WAVE per tracking channel and millisecond, four channels per workgroup (from 2048 channels on:
// 30 % less time per channel than the workgroup-per-channel form above, which has the shorter latency for a few).
//   fine = (int16) code_phase_fine;  replica shift = fine & 7;  prompt offset = fine / 8, early = prompt - 1
//   (wraps to 2045), late = prompt + 1 (wraps to 0)                                   PM/GPS/tracking.c:115-130
//   carrier NCO continues from if_freq_accum at (float)IF + if_freq_offset_hz and is stored back  gps_misc.c:244-274
// The wave stages the channel's replica words and its wiped I / Q streams in LDS -- the streams with four bytes of
// circular padding in front and six behind, so that the 16 data bits replica word i meets at byte offset
// (o + 2 i) mod 2046 and their neighbours one byte either side are one contiguous 32-bit window -- and then walks the
// 1023 replica words once: one replica read and one two-dword read per stream serve Early, Prompt and Late together
// whenever they are the neighbours they normally are (any other triple of offsets, e.g. out of a negative code phase,
// takes one window per offset).  gps_mult_and_summ's rules (PM/GPS/gps_misc.c:60-90) per offset: odd offsets skip word
// p1 = (2046 - o) / 2 and word 1022.
namespace {

struct TrackLds {
  u32 w[2][516];       // wiped stream, I / Q: byte k of w = data byte (k - 4) mod 2046
  uint16_t rep[1024];  // replica words
};

__device__ __forceinline__ u32 window32(const u32 *w, int byte_index)   // 32 bits from padded byte index
{
  const int j = byte_index >> 2;
  return __builtin_amdgcn_alignbit(w[j + 1], w[j], 8u * (u32)(byte_index & 3));
}

}  // namespace

__global__ __launch_bounds__(256) void k_track_epl_wave(const uint8_t *__restrict__ if_block, int if_format,
                                                        gpsx_trk_state_t *__restrict__ st, int n_ch,
                                                        const uint8_t *__restrict__ chips_all,
                                                        int16_t *__restrict__ iq_out)
{
  __shared__ TrackLds lds[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ch_raw = blockIdx.x * 4 + wave;
  const bool live = ch_raw < n_ch;
  const int ch = live ? ch_raw : n_ch - 1;   // idle waves of the last workgroup shadow a real channel (no early exit
                                             // before the barriers) and write nothing
  TrackLds &L = lds[wave];
  const gpsx_trk_state_t state = st[ch];
  const int prn = state.prn >= 0 && state.prn <= GPSX_MAX_PRN ? state.prn : 0;
  const uint8_t *chips = chips_all + (size_t)prn * 1024;

  const int fine = (int)(int16_t)(int)state.code_phase_fine;
  const u32 b = (u32)fine & 7u;
  const u32 low = (1u << b) - 1u, high = (0xFFFFu << b) & 0xFFFFu;
  const float freq_hz = (float)kIfHz + state.if_freq_offset_hz;
  const u32 step = nco_step_per_word(freq_hz);

  // replica words (K2) and the wiped streams (K3): dword w of the block -> padded dword w + 1
  for (int i = lane; i < 1024; i += 64) {
    const u32 prev = (i > 0 && i <= kWords16) ? chips[i - 1] : 0u;
    const u32 cur = i < kWords16 ? chips[i] : 0u;
    L.rep[i] = (uint16_t)((prev ? low : 0u) | (cur ? high : 0u));
  }
  for (int w = lane; w < 512; w += 64) {
    u32 vi = 0, vq = 0;   // dword 511 = word 1022 + nothing: the 16 samples the NCO loop never mixes read as zero
    if (w < kWords32) {
      const u32 x = (u32)load_sign16(if_block, 2 * w, if_format) | ((u32)load_sign16(if_block, 2 * w + 1, if_format) << 16);
      const u32 quad = (state.if_freq_accum + step * (u32)w) >> 30;
      vi = carrier_i(quad) ^ x;
      vq = carrier_q(quad) ^ x;
    }
    L.w[0][w + 1] = vi;
    L.w[1][w + 1] = vq;
  }
  __syncthreads();
  if (lane < 2) {
    // circular padding: bytes -4..-1 = data bytes 2042..2045 (dword 511's low half holds 2044, 2045; 2046, 2047 are
    // outside the circle), bytes 2046..2051 = data bytes 0..5
    u32 *w = L.w[lane];
    const u32 tail = (w[511] >> 16) | (w[512] << 16);                 // data bytes 2042..2045
    const u32 head0 = w[1], head1 = w[2];                              // data bytes 0..3, 4..7
    w[0] = tail;
    w[512] = (w[512] & 0xFFFFu) | (head0 << 16);                       // 2044, 2045, then 0, 1
    w[513] = (head0 >> 16) | (head1 << 16);                            // 2, 3, 4, 5
    w[514] = head1 >> 16;
  }
  __syncthreads();

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
  int p1[3];
  bool odd[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    odd[k] = off[k] & 1u;
    p1[k] = (kBytes - (int)off[k]) >> 1;
  }
  const bool neighbours = off[0] == (off[1] + kBytes - 1) % kBytes && off[2] == (off[1] + 1) % kBytes;
  u32 ci[3] = {0, 0, 0}, cq[3] = {0, 0, 0};
  for (int i = lane; i < kWords16; i += 64) {
    const u32 r = L.rep[i];
    u32 di[3], dq[3];
    if (neighbours) {
      int a = (int)off[1] + 2 * i;
      a = a >= kBytes ? a - kBytes : a;
      const u32 wi = window32(L.w[0], a + 3), wq = window32(L.w[1], a + 3);   // data bytes a - 1 .. a + 2
      di[0] = wi & 0xFFFFu;
      di[1] = (wi >> 8) & 0xFFFFu;
      di[2] = wi >> 16;
      dq[0] = wq & 0xFFFFu;
      dq[1] = (wq >> 8) & 0xFFFFu;
      dq[2] = wq >> 16;
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        int a = (int)off[k] + 2 * i;
        a = a >= kBytes ? a - kBytes : a;
        di[k] = window32(L.w[0], a + 4) & 0xFFFFu;
        dq[k] = window32(L.w[1], a + 4) & 0xFFFFu;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const bool skip = odd[k] && (i == p1[k] || i == kWords16 - 1);
      ci[k] += skip ? 0u : pop16(di[k] ^ r);
      cq[k] += skip ? 0u : pop16(dq[k] ^ r);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const u32 si = wave_sum_to_lane63(ci[k]), sq = wave_sum_to_lane63(cq[k]);
    if (lane == 63 && live) {
      iq_out[ch * 6 + k * 2 + 0] = (int16_t)((int)si - kHalf);
      iq_out[ch * 6 + k * 2 + 1] = (int16_t)((int)sq - kHalf);
    }
  }
  if (lane == 0 && live)
    st[ch].if_freq_accum = state.if_freq_accum + step * (u32)kWords32;
}

constexpr int kTrackWaveFormFrom = 2048;

void launch_track_epl(hipStream_t s, const uint8_t *d_if_block, int if_format, gpsx_trk_state_t *d_st, int n_ch,
                      const uint8_t *d_chips, int16_t *d_iq)
{
  if (n_ch <= 0)
    return;
  if (n_ch < kTrackWaveFormFrom)
    hipLaunchKernelGGL(k_track_epl, dim3(n_ch), dim3(256), 0, s, d_if_block, if_format, d_st, n_ch, d_chips, d_iq);
  else
    hipLaunchKernelGGL(k_track_epl_wave, dim3((n_ch + 3) / 4), dim3(256), 0, s, d_if_block, if_format, d_st, n_ch, d_chips,
                       d_iq);
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
__global__ void k_rewind(gpsx_trk_state_t *__restrict__ st, int n_ch, const uint8_t *__restrict__ steps)
{
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= n_ch)
    return;
  const u32 step = nco_step_per_sample((float)kIfHz + st[ch].if_freq_offset_hz);
  const u64 adv = (u64)step * (u64)kSamples * (u64)steps[ch];
  st[ch].if_freq_accum += (u32)adv;
}

void launch_rewind(hipStream_t s, gpsx_trk_state_t *d_st, int n_ch, const uint8_t *d_steps)
{
  if (n_ch <= 0)
    return;
  hipLaunchKernelGGL(k_rewind, dim3((n_ch + 63) / 64), dim3(64), 0, s, d_st, n_ch, d_steps);
}

}  // namespace gpsx
