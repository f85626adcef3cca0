// gpsx_track_wave.hpp -- the wave-per-channel E/P/L correlators as a device function, shared by k_track_epl_wave (one
// millisecond per launch, k_generic.hip) and k_track_loop (K milliseconds per launch with the tracking loops behind the
// correlators, k_track_loop.hip).  DESIGN.md 4.3; the comments in front of k_track_epl_wave describe the formulation.
#pragma once
#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"

namespace gpsx {

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

namespace trkwave {

struct alignas(4) TrkW4 { u32 w[4]; };   // four table words at a dword-aligned address: one global_load_dwordx4
struct alignas(4) TrkW2 { u32 w[2]; };

// (I byte, Q byte) `pos` (0..2045) of the wiped streams of a channel, recomputed from the staged block: bytes 2044 / 2045 are
// the sixteen samples the NCO loop never mixes and read as zero (PM/GPS/gps_misc.c:229,261)
__device__ __forceinline__ uint2 trk_dbyte(const u32 *s_x, const uint2 *s_carrier, u32 acc0, u32 step, int pos)
{
  const int w = pos >> 2;
  const u32 x = s_x[w];
  const uint2 c = s_carrier[(acc0 + step * (u32)w) >> 30];
  const u32 sh = 8u * ((u32)pos & 3u);
  uint2 r = uint2{((x ^ c.x) >> sh) & 0xFFu, ((x ^ c.y) >> sh) & 0xFFu};
  if (pos >= 2 * kWords32 * 2)
    r = uint2{0u, 0u};
  return r;
}

// acc + pop(x): v_bcnt_u32_b32's own addend (left to itself the compiler counts into a zero and adds three at a time)
__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc)
{
  u32 r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}

// pop(x ^ carrier ^ replica window) for NK offsets 8 samples apart (NK = 3: Late, Prompt, Early from ONE six-word read per
// four data words) or one offset (NK = 1): `row` = the PRN's table row advanced to the word of the window's first bit,
// `sh` = that bit's position in it.  cnt[k0 + k] += this lane's share, I in [0], Q in [1].  The wipe-off (x ^ carrier word
// of the NCO quadrant) never exists by itself: the three-input XOR is one v_bitop3_b32.
template <int NK>
__device__ __forceinline__ void trk_correlate(const u32 *__restrict__ row, u32 sh, u32 lane4, const u32 (&x)[8], const uint2 (&cw)[8],
                                              u32 (&cnt)[3][2], int k0)
{
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const u32 *p = row + lane4 + 256 * it;
    const TrkW4 t4 = *reinterpret_cast<const TrkW4 *>(p);
    const TrkW2 t2 = *reinterpret_cast<const TrkW2 *>(p + 4);
    const u32 t[6] = {t4.w[0], t4.w[1], t4.w[2], t4.w[3], t2.w[0], t2.w[1]};
    u32 a[5];
#pragma unroll
    for (int u = 0; u < (NK == 3 ? 5 : 4); u++)
      a[u] = __builtin_amdgcn_alignbit(t[u + 1], t[u], sh);
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int k = 0; k < NK; k++) {
        const u32 r = k == 0 ? a[u] : __builtin_amdgcn_alignbit(a[u + 1], a[u], 8u * (u32)k);
        cnt[k0 + k][0] = bcnt_acc(__builtin_amdgcn_bitop3_b32(x[4 * it + u], cw[4 * it + u].x, r, 0x96), cnt[k0 + k][0]);
        cnt[k0 + k][1] = bcnt_acc(__builtin_amdgcn_bitop3_b32(x[4 * it + u], cw[4 * it + u].y, r, 0x96), cnt[k0 + k][1]);
      }
    }
  }
}

template <int CTRL>
__device__ __forceinline__ u32 dpp_get(u32 v)
{
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}


// The block's sign plane as 32-bit words (word 511 = 16-bit word 1022 alone) and the carrier words per NCO quadrant, staged
// by the whole workgroup; the caller synchronises.
__device__ __forceinline__ void stage_block(const uint8_t *__restrict__ if_block, int if_format, u32 *s_x, uint2 *s_carrier)
{
  if (threadIdx.x < 4)
    s_carrier[threadIdx.x] = uint2{carrier_i(threadIdx.x), carrier_q(threadIdx.x)};
  for (int w = threadIdx.x; w < 512; w += blockDim.x) {
    const u32 lo = load_sign16(if_block, 2 * w, if_format);
    const u32 hi = 2 * w + 1 < kWords16 ? (u32)load_sign16(if_block, 2 * w + 1, if_format) : 0u;
    s_x[w] = lo | (hi << 16);
  }
}

// One millisecond of Early / Prompt / Late for the n_here channels of a wave.  Lane 4 c + k (k = 0 / 1 / 2) carries channel
// c's validated PRN, code phase (samples, as tracking.c truncates it), NCO step per word and accumulator at the start of the
// block, and gets back (I & 0xFFFF) | (Q << 16) of offset k -- the reference's signed 16-bit accumulators.  `mine` = this
// lane holds a (channel, offset) of the wave.
__device__ __forceinline__ u32 wave_epl(const u32 *s_x, const uint2 *s_carrier, int lane, int n_here, bool mine, int prn, int fine,
                                        u32 step, u32 acc_in, const u32 *__restrict__ chipbits_all,
                                        const u32 *__restrict__ rep_all)
{
  const int c_l = lane >> 2, k_l = lane & 3;
  const u32 b = (u32)fine & 7u;
  const u32 low = (1u << b) - 1u, high = (0xFFFFu << b) & 0xFFFFu;
  // the offset exactly as tracking.c forms it, then reduced to the circle
  const unsigned prompt = (unsigned)(uint16_t)(fine / 8);
  unsigned off = (unsigned)(uint16_t)(prompt + (unsigned)k_l - 1u);
  if (k_l == 0 && off >= 2u * kChips) off = 2u * kChips - 1u;
  if (k_l == 2 && off >= 2u * kChips) off = 0u;
  if (off >= 2u * kChips) off = 0u;   // beyond 2046 the reference would read out of bounds (kept in range); 2046 behaves as 0
  // where the window of replica bit stream word 0 starts in the PRN's table row: bit (-8 off - b) mod 16368
  const u32 t0 = (u32)(3 * kSamples - 8 * (int)off - (int)b) % (u32)kSamples;
  const u32 info = ((u32)prn << 14) | t0;

  const u32 lane4 = 4u * (u32)lane;
  const int xor16 = (lane ^ 16) << 2, xor32 = (lane ^ 32) << 2;
  u32 sums = 0;   // lane 4 c + k: (count_I | count_Q << 16) against the circular replica, before the quirk terms

#pragma unroll 1
  for (int c = 0; c < n_here; c++) {
    const u32 acc0 = (u32)__builtin_amdgcn_readlane((int)acc_in, 4 * c);
    const u32 stp = (u32)__builtin_amdgcn_readlane((int)step, 4 * c);
    const u32 inf_e = (u32)__builtin_amdgcn_readlane((int)info, 4 * c);
    const u32 inf_p = (u32)__builtin_amdgcn_readlane((int)info, 4 * c + 1);
    const u32 inf_l = (u32)__builtin_amdgcn_readlane((int)info, 4 * c + 2);
    // K3: the carrier words of this lane's eight stream words, word w sees NCO phase accum + w step
    // (PM/GPS/gps_misc.c:253-262); word 511 (lane 63's last) is taken back after the loop
    u32 x[8];
    uint2 cw[8];
    {
      u32 acc = acc0 + stp * lane4;
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const uint4 x4 = *reinterpret_cast<const uint4 *>(&s_x[lane4 + 256 * it]);
        x[4 * it] = x4.x; x[4 * it + 1] = x4.y; x[4 * it + 2] = x4.z; x[4 * it + 3] = x4.w;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          cw[4 * it + u] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(s_carrier) + ((acc >> 27) & 0x18u));
          acc += stp;
        }
        acc += stp * 252u;
      }
    }
    u32 cnt[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    const u32 t_e = inf_e & 0x3FFFu, t_p = inf_p & 0x3FFFu, t_l = inf_l & 0x3FFFu;
    const u32 *row = rep_all + (size_t)(inf_l >> 14) * kTrackRepStride;
    const u32 d_pl = t_p >= t_l ? t_p - t_l : t_p + (u32)kSamples - t_l, d_el = t_e >= t_l ? t_e - t_l : t_e + (u32)kSamples - t_l;
    u32 p_e, p_p, p_l;   // I | Q << 16 (a lane's share is at most 8 x 32: the totals stay below 2^15)
    if (d_pl == 8u && d_el == 16u) {
      // Late at the window's first bit, Prompt 8 and Early 16 samples further on (cnt[0] = Late here)
      trk_correlate<3>(row + (t_l >> 5), t_l & 31u, lane4, x, cw, cnt, 0);
      p_l = cnt[0][0] | (cnt[0][1] << 16);
      p_p = cnt[1][0] | (cnt[1][1] << 16);
      p_e = cnt[2][0] | (cnt[2][1] << 16);
    } else {   // (code phases outside [0, 16368): the three offsets are not neighbours)
      trk_correlate<1>(row + (t_e >> 5), t_e & 31u, lane4, x, cw, cnt, 0);
      trk_correlate<1>(row + (t_p >> 5), t_p & 31u, lane4, x, cw, cnt, 1);
      trk_correlate<1>(row + (t_l >> 5), t_l & 31u, lane4, x, cw, cnt, 2);
      p_e = cnt[0][0] | (cnt[0][1] << 16);
      p_p = cnt[1][0] | (cnt[1][1] << 16);
      p_l = cnt[2][0] | (cnt[2][1] << 16);
    }
    // Three wave sums in one transposing reduction: after the two quad steps lane class (lane & 3) = 0 / 1 / 2 / 3 carries
    // Early / Prompt / Late / Late, the rest of the butterfly keeps the class -- 13 vector instructions instead of 3 x 8.
    const bool odd = lane & 1, upper = lane & 2;
    u32 ab = (odd ? p_p : p_e) + dpp_get<0xB1>(odd ? p_e : p_p);   // quad_perm [1,0,3,2]
    u32 cc = p_l + dpp_get<0xB1>(p_l);
    u32 v = (upper ? cc : ab) + dpp_get<0x4E>(upper ? ab : cc);    // quad_perm [2,3,0,1]
    v += dpp_get<0x124>(v);                                        // row_ror:4
    v += dpp_get<0x128>(v);                                        // row_ror:8
    v += (u32)__builtin_amdgcn_ds_bpermute(xor16, (int)v);
    v += (u32)__builtin_amdgcn_ds_bpermute(xor32, (int)v);
    sums = c_l == c ? v : sums;
  }

  // ---- the reference's quirks against the circular count, per (channel, offset) -----------------------------------------
  if (!mine)
    return 0u;
  const u32 acc0 = acc_in;
  const u32 *cb = chipbits_all + (size_t)prn * 32;
  const u32 cb31 = cb[31];
  const bool c1022 = (cb31 >> 30) & 1u, c1021 = (cb31 >> 29) & 1u;
  u32 total = sums;
  {
    // stream word 511: only its low half exists, the sixteen samples the NCO loop never mixes, which read as zero
    // (PM/GPS/gps_misc.c:229,261) -- the loop counted 32 mixed bits there: take them back, count the replica's sixteen
    const u32 *rw = rep_all + (size_t)prn * kTrackRepStride + (t0 >> 5) + kWords32;
    const u32 r = __builtin_amdgcn_alignbit(rw[1], rw[0], t0 & 31u);
    const u32 x511 = s_x[kWords32];
    const uint2 c511 = s_carrier[(acc0 + step * (u32)kWords32) >> 30];
    const u32 right = pop16(r);
    total += (right - (u32)__popc(x511 ^ c511.x ^ r)) + ((right - (u32)__popc(x511 ^ c511.y ^ r)) << 16);
  }
  if (c1022 && b) {   // Q5: the replica's first b samples are zero, not the tail of chip 1022 (PM/GPS/gps_misc.c:290-297)
    const uint2 by = trk_dbyte(s_x, s_carrier, acc0, step, (int)off);
    total += (u32)(2 * (int)__popc(by.x & low) - (int)b) + ((u32)(2 * (int)__popc(by.y & low) - (int)b) << 16);
  }
  if (off & 1u) {     // Q3: odd byte offsets skip the replica word at the wrap and the last one (PM/GPS/gps_misc.c:66-90)
    const int o = (int)off;
    const int p1 = (kBytes - o) >> 1;
    // 16-bit replica word p1: chip p1 - 1 below bit b (chip -1 = 0), chip p1 from bit b on
    const int i0 = p1 - 1;
    const int wlo = i0 > 0 ? i0 >> 5 : 0;
    u32 c2 = __builtin_amdgcn_alignbit(cb[wlo < 31 ? wlo + 1 : 31], cb[wlo], (u32)(i0 & 31));
    if (i0 < 0)
      c2 = cb[0] << 1;
    const u32 r1 = ((c2 & 1u) ? low : 0u) | ((c2 & 2u) ? high : 0u);
    const uint2 b0 = trk_dbyte(s_x, s_carrier, acc0, step, 0);   // data bytes (2045, 0): byte 2045 is never mixed
    u32 sub_i = pop16((b0.x << 8) ^ r1), sub_q = pop16((b0.y << 8) ^ r1);
    if (p1 != kWords16 - 1) {   // o >= 3
      const u32 r2 = (c1021 ? low : 0u) | (c1022 ? high : 0u);
      const uint2 ba = trk_dbyte(s_x, s_carrier, acc0, step, o - 2), bb = trk_dbyte(s_x, s_carrier, acc0, step, o - 1);
      sub_i += pop16((ba.x | (bb.x << 8)) ^ r2);
      sub_q += pop16((ba.y | (bb.y << 8)) ^ r2);
    }
    total -= sub_i + (sub_q << 16);
  }
  const u32 res_i = (total & 0xFFFFu) - (u32)kHalf, res_q = (total >> 16) - (u32)kHalf;
  return (res_i & 0xFFFFu) | (res_q << 16);
}

}  // namespace trkwave
}  // namespace gpsx
