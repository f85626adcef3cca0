// k_acq_ds.hip -- acquisition grid kernel, fine (16368-phase) sweep, Doppler-SHARED polyphase formulation.
//
// Same contract and same outputs as k_acq_poly (k_acq_poly.hip): per (search, PRN, Doppler, replica bit shift) the
// triplet correlation_search (PM/GPS/gps_misc.c:155-191) returns, bit for bit.  Same recurrence over the sixteen sample
// offsets t0 inside a chip,   M_{t0+1}(q) = M_t0(q) - X_t0(q) + X_t0(q + 1),   X_j(q) = sum_k cc[(k - q) mod 1023] d_j[k],
// d_j[k] = D(16 k + j) the j-th polyphase bit plane of the carrier-wiped stream.  What changes is how X is obtained.
//
// The wiped sample is the raw sample XOR a carrier bit, d_j[k] = x_j[k] ^ sig[k], and the carrier bit only depends on
// the NCO quadrant of the 32-sample carrier word k >> 1 (PM/GPS/gps_misc.c:211-240): it is piecewise constant in k,
// with 4 |f_doppler| sign changes per millisecond -- 10.5 on average over a +-5 kHz grid.  Hence, with
//      T[k] = cc[(k - q) mod 1023] ^ x_j[k]        (no Doppler in it)
// the Hamming distance H(q) = sum_k T[k] ^ sig[k] = 512 + pop(d_j) - 2 X_j(q) is a signed sum of PREFIX popcounts of T
// taken at the sign changes K_b,   H(q) = const + sum_b 2 z_b P(K_b) + P(1023),  P(K) = sum_{k<K} T[k],  z_b = +-1
// (tests/test_formulation_dshare.py is this algebra in numpy, quirks Q1 / Q2 included: where the reference's 7-nibble
// carrier literal makes even and odd k differ, a second prefix over even k enters with its own coefficients).
// One pass of 32 x (v_xor + v_bcnt) per (q, plane) therefore serves EVERY Doppler bin and both streams -- against
// 32 x 2 x (v_and + v_bcnt) per bin in k_acq_poly -- and each bin then costs one masked popcount per sign change.
//
// Work layout: one wave per (search, PRN, chunk of NF Doppler bins, tile of 63 chip offsets); lane l owns q = 63 tile + l
// (lane 63 is the right-hand halo: X(q + 1) comes from the neighbour lane).  The lane's 32 chip-window words live in
// LDS ([w][lane]), the raw plane words and the per-(plane class, word) boundary rows are wave-uniform scalar loads, the
// 2 x NF running correlations M and sign-weighted prefix sums Z are registers.  M_0 comes from a first sweep over the
// sixteen planes (M_0 = sum_j X_j), then fifteen recurrence steps, each followed by the per-hypothesis epilogue of
// k_acq_poly (corrections for quirks Q3 / Q5, magnitude, windowed search).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"

namespace gpsx {

namespace {

constexpr int kNF = kDsDopplersPerWave;
constexpr int kTileQ = 63;                 // chip offsets finished per wave
constexpr int kTiles = 17;                 // ceil(1023 / 63)
constexpr int kRows = 33;                  // boundary rows per (chunk, class): the 32 words + the K = 1022 end terms
constexpr int kClasses = 6;                // plane classes: j & 3, and the two planes quirk Q1 touches (j = 12, 15)
constexpr int kRowDwords = 16;             // boundary row of one (chunk, class, word): has, has_e, 24 op bytes, 24 E-op bytes
constexpr int kRowOps = 2, kRowOpsE = 8;   // dword offsets of the two byte arrays
constexpr int kRecDwords = kDsRecDwords;   // interleaved wiped bytes of one (search, Doppler, alignment copy)

__host__ __device__ inline int plane_class(int j) { return j == 12 ? 4 : (j == 15 ? 5 : (j & 3)); }

// ---------------------------------------------------------------------------------------------------------------------
// prepare: per (search, Doppler) the wiped streams as the epilogue wants them, pop(D); per search the raw bit planes
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ds_prepare(const uint8_t *__restrict__ if_blocks, int if_format,
                                                     int search_stride_blocks, int n_dopp, int dopp_min_hz,
                                                     int dopp_step_hz, u32 *__restrict__ hdr, u32 *__restrict__ rec,
                                                     u32 *__restrict__ xpl, uint4 *__restrict__ etab)
{
  __shared__ u32 x32[512];
  __shared__ u32 d[2][512];
  __shared__ u32 ones[2];
  const int tid = threadIdx.x;
  const int f = blockIdx.x % n_dopp;
  const int search = blockIdx.x / n_dopp;
  const size_t block_bytes = if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : kBytes;
  const uint8_t *blk = if_blocks + (size_t)(search * search_stride_blocks) * block_bytes;
  uint16_t *x16 = reinterpret_cast<uint16_t *>(x32);
  for (int i = tid; i < 1024; i += 256)
    x16[i] = i < kWords16 ? load_sign16(blk, i, if_format) : (uint16_t)0;
  if (tid < 2)
    ones[tid] = 0;
  __syncthreads();
  const float freq_hz = (float)(kIfHz + dopp_min_hz + f * dopp_step_hz);   // PM/GPS/acquisition.c:285-289
  const u32 step_word = nco_step_per_word(freq_hz);
  u32 oi = 0, oq = 0;
  for (int w = tid; w < 512; w += 256) {
    u32 vi = 0, vq = 0;
    if (w < kWords32) {   // the last 16 samples are never mixed and read as zero (quirk Q2)
      const u32 quad = (step_word * (u32)w) >> 30;
      vi = carrier_i(quad) ^ x32[w];
      vq = carrier_q(quad) ^ x32[w];
    }
    d[0][w] = vi;
    d[1][w] = vq;
    oi += __popc(vi);
    oq += __popc(vq);
  }
  oi = wave_sum_u32(oi);
  oq = wave_sum_u32(oq);
  if ((tid & 63) == 0) {
    atomicAdd(&ones[0], oi);
    atomicAdd(&ones[1], oq);
  }
  __syncthreads();
  u32 *h = hdr + (size_t)(search * n_dopp + f) * 4;
  if (tid == 0) {
    h[0] = ones[0];
    h[1] = ones[1];
    h[2] = d[0][0] & 0xFFu;
    h[3] = d[1][0] & 0xFFu;
  }
  if (tid < 16) {
    // what the epilogue of sample offset t0 = tid needs that is the same for every chip offset: the two bases
    // (C0 = pop(D) + 8192 - 2 M) and, for odd byte offsets, the popcounts of the wrap word (data bytes 2045, 0) against
    // the four possible replica patterns (quirk Q3), one byte each
    const int b = tid & 7, half = tid >> 3;
    const u32 low_mask = (1u << b) - 1u, high_mask = (0xFFFFu << b) & 0xFFFFu;
    const u32 wrap_i = (d[0][0] & 0xFFu) << 8, wrap_q = (d[1][0] & 0xFFu) << 8;
    u32 ti = 0, tq = 0;
    for (int k = 0; k < 4; k++) {
      const u32 r = ((k & 1) ? low_mask : 0u) | ((k & 2) ? high_mask : 0u);
      ti |= pop16(wrap_i ^ r) << (8 * k);
      tq |= pop16(wrap_q ^ r) << (8 * k);
    }
    etab[(size_t)(search * n_dopp + f) * 16 + tid] = uint4{ones[0] + kHalf + 8, ones[1] + kHalf + 8, half ? ti : 0u, half ? tq : 0u};
  }
  // entry e of alignment copy a holds bytes (I, Q) of data byte e - 2 - a: the epilogue's 8-byte window over bytes
  // o - 2 .. o + 1 then starts at a dword boundary for even o (copy 0) and for odd o (copy 1)
  const uint8_t *bi = reinterpret_cast<const uint8_t *>(d[0]);
  const uint8_t *bq = reinterpret_cast<const uint8_t *>(d[1]);
  for (int a = 0; a < 2; a++) {
    u32 *r = rec + ((size_t)(search * n_dopp + f) * 2 + a) * kRecDwords;
    for (int i = tid; i < kRecDwords; i += 256) {
      u32 v = 0;
      for (int e2 = 0; e2 < 2; e2++) {
        const int byte = 2 * i + e2 - 2 - a;
        if (byte >= 0 && byte < kBytes)
          v |= ((u32)bi[byte] | ((u32)bq[byte] << 8)) << (16 * e2);
      }
      r[i] = v;
    }
  }
  if (f == 0) {
    // raw polyphase planes: bit i of word w of plane j = sample 16 (32 w + i) + j; k = 1022 (unmixed, reads as 0) and
    // the pad k = 1023 are zero
    for (int m = tid; m < 16 * 32; m += 256) {
      const int j = m >> 5, w = m & 31;
      u32 v = 0;
      for (int i = 0; i < 32; i++) {
        const int k = 32 * w + i;
        if (k < 1022) {
          const int s = 16 * k + j;
          v |= ((x32[s >> 5] >> (s & 31)) & 1u) << i;
        }
      }
      xpl[((size_t)search * 16 + j) * 32 + w] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------------------------------------------
struct DsShared {
  u32 cw[32][64];          // chip windows: bit i of cw[w][lane] = cc[(32 w + i - q) mod 1023]
  u32 cb[32];              // packed code
  u32 part[kNF][8][2];     // (packed best key, sum) of the even byte offsets, waiting for the odd ones
};

// 32 chips starting at chip s (0 <= s < 1023) of the periodic code
__device__ __forceinline__ u32 cc_bits32(const u32 *cb, int s)
{
  const int wi = s >> 5;
  const u32 lo = cb[wi];
  const u32 hi = wi < 31 ? cb[wi + 1] : 0u;
  u32 v = __builtin_amdgcn_alignbit(hi, lo, (u32)(s & 31));
  const int n1 = kChips - s;   // chips left before the code wraps
  if (n1 < 32)
    v = (v & ((1u << n1) - 1u)) | (cb[0] << n1);
  return v;
}

// value of lane + 1 (lane 63 keeps its own)
__device__ __forceinline__ int from_next_lane(int v) { return (int)dpp<0x130>((u32)v, (u32)v); }   // wave_shl:1

__device__ __forceinline__ int s2(u32 code, int bit) { return ((int)(code << (30 - bit))) >> 30; }   // 2-bit signed field

// The sign changes of the chunk's Doppler bins that fall into one 32-chip word: each costs a masked popcount of T on top
// of the running prefix, added with its sign to the bin's Z.  The row is wave-uniform (SGPRs): everything but the three
// or four vector instructions per sign change is scalar work.
template <int NF>
__device__ __forceinline__ void ds_apply_row(u32 t, u32 pre, u32 pre_e, const uint4 &r0, const uint4 &r1, const uint4 &r2,
                                             const uint4 &r3, bool quirk, int (&zi)[NF], int (&zq)[NF])
{
  const u32 r[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
  const u32 has = r[0];
  if (!has)
    return;
  const u32 has_e = quirk ? r[1] : 0u;
#pragma unroll
  for (int f = 0; f < NF; f++) {
    if ((has >> f) & 1u) {
      const u32 op = (r[kRowOps + (f >> 2)] >> (8 * (f & 3))) & 0xFFu;
      const u32 tm = t & ((1u << (2u * (op & 15u))) - 1u);
      const int p = (int)((u32)__popc(tm) + pre);
      if (op & 0x30u)   // a sign change flips the carrier bit of ONE stream (both only at the K = 1022 end term)
        zi[f] += __mul24(s2(op, 4), p);
      if (op & 0xC0u)
        zq[f] += __mul24(s2(op, 6), p);
      if ((has_e >> f) & 1u) {
        const u32 ope = (r[kRowOpsE + (f >> 2)] >> (8 * (f & 3))) & 0xFFu;
        const int pe = (int)((u32)__popc(tm & 0x55555555u) + pre_e);
        zi[f] += __mul24(s2(ope, 4), pe);
        zq[f] += __mul24(s2(ope, 6), pe);
      }
    }
  }
}

// One plane: Doppler-independent prefix popcounts of T = window ^ raw plane, and for each Doppler bin of the chunk the
// sign-weighted sum Z of those prefixes at its carrier sign changes (accumulated into zi / zq).  Rows 0..31 belong to the
// 32 words; row 32 holds the K = 1022 end terms (k = 1022 is never mixed: whatever sign the last carrier run had ends
// there), evaluated on word 31 again.  Next word's row, plane word and chip window are fetched before this word's work
// and first touched after it, so that their latency hides behind it.
template <int NF>
__device__ __forceinline__ void ds_plane(const DsShared &sh, int lane, const u32 *__restrict__ rows, const u32 *__restrict__ xw,
                                         bool quirk, int (&zi)[NF], int (&zq)[NF], u32 &total)
{
  const uint4 *rp = reinterpret_cast<const uint4 *>(rows);
  uint4 c0 = rp[0], c1 = rp[1], c2 = rp[2], c3 = rp[3];
  u32 x_cur = xw[0], cw_cur = sh.cw[0][lane];
  u32 pre = 0, pre_e = 0, pre_31 = 0, pre_e_31 = 0, t = 0;
#pragma unroll 1
  for (int w = 0; w < 32; w++) {
    const int wn = w < 31 ? w + 1 : 31;
    const uint4 n0 = rp[(w + 1) * 4 + 0], n1 = rp[(w + 1) * 4 + 1], n2 = rp[(w + 1) * 4 + 2], n3 = rp[(w + 1) * 4 + 3];
    const u32 x_next = xw[wn], cw_next = sh.cw[wn][lane];
    t = cw_cur ^ x_cur;
    if (w == 31) {
      pre_31 = pre;
      pre_e_31 = pre_e;
    }
    ds_apply_row<NF>(t, pre, pre_e, c0, c1, c2, c3, quirk, zi, zq);
    pre += (u32)__popc(t);
    if (quirk)
      pre_e += (u32)__popc(t & 0x55555555u);
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    x_cur = x_next;
    cw_cur = cw_next;
  }
  ds_apply_row<NF>(t, pre_31, pre_e_31, c0, c1, c2, c3, quirk, zi, zq);
  total = pre;
}

template <int NF>
__global__ __launch_bounds__(64, 3) void k_acq_ds(const DsParams P)
{
  __shared__ DsShared sh;
  const int lane = threadIdx.x;
  int id = blockIdx.x;
  const int tile = id % kTiles;
  id /= kTiles;
  const int chunk = id % P.n_chunks;
  id /= P.n_chunks;
  const int slot = id % P.n_prn;
  const int search = id / P.n_prn;
  const int f0 = chunk * NF;
  const int nf = P.n_dopp - f0 < NF ? P.n_dopp - f0 : NF;
  const int q = tile * kTileQ + lane;
  const int qm = q >= kChips ? q - kChips : q;
  const bool emit = lane < kTileQ && q < kChips;

  const u32 *chipbits_p = P.chipbits + (size_t)slot * 32;
  if (lane < 32)
    sh.cb[lane] = chipbits_p[lane];
  for (int i = lane; i < kNF * 8 * 2; i += 64)
    (&sh.part[0][0][0])[i] = 0;
  __syncthreads();
#pragma unroll 1
  for (int w = 0; w < 32; w++) {
    int s = 32 * w - qm;
    s = s < 0 ? s + kChips : s;
    u32 v = cc_bits32(sh.cb, s);
    if (w == 31)
      v &= 0x7FFFFFFFu;   // k = 1023 does not exist
    sh.cw[w][lane] = v;
  }
  // replica bits of the word odd offsets skip at the wrap (quirk Q3): chips p1 - 1, p1 with p1 = 1022 - q
  const int p1 = kChips - 1 - qm;
  const u32 c_p1 = (sh.cb[p1 >> 5] >> (p1 & 31)) & 1u;
  const u32 c_pm1 = p1 > 0 ? (sh.cb[(p1 - 1) >> 5] >> ((p1 - 1) & 31)) & 1u : 0u;
  const u32 sel8 = (c_pm1 | (c_p1 << 1)) * 8u;
  const u32 tail_bits = chipbits_p[31];   // wave-uniform -> scalar
  const bool c1022 = (tail_bits >> 30) & 1u, c1021 = (tail_bits >> 29) & 1u;
  __syncthreads();

  const u32 *__restrict__ rows_chunk = P.rows + (size_t)chunk * kClasses * kRows * kRowDwords;
  const u32 *__restrict__ xpl_s = P.xpl + (size_t)search * 16 * 32;
  const u32 *__restrict__ hdr_s = P.hdr + (size_t)(search * P.n_dopp + f0) * 4;
  const u32 *__restrict__ rec_s = P.rec + (size_t)(search * P.n_dopp + f0) * 2 * kRecDwords;
  const uint4 *__restrict__ etab_s = P.etab + (size_t)(search * P.n_dopp + f0) * 16;
  const int lane_f = lane < nf ? lane : 0;   // lane f fetches bin f's per-offset uniforms; v_readlane hands them out

  // ---- M_0(q) = sum_j X_j(q) = (16 * 512 + pop(D) - sum_j H_j(q)) / 2 ------------------------------------------------
  int m_i[NF], m_q[NF];
  {
    int zi[NF], zq[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) {
      zi[f] = 0;
      zq[f] = 0;
    }
    u32 f_sum = 0;
#pragma unroll 1
    for (int j = 0; j < 16; j++) {
      const int c = plane_class(j);
      u32 total;
      ds_plane<NF>(sh, lane, rows_chunk + (size_t)c * kRows * kRowDwords, xpl_s + j * 32, c >= 4, zi, zq, total);
      f_sum += total;
    }
    const int pop_i_l = (int)hdr_s[lane_f * 4 + 0], pop_q_l = (int)hdr_s[lane_f * 4 + 1];
    const int cst_i_l = P.cst0[(f0 + lane_f) * 2 + 0], cst_q_l = P.cst0[(f0 + lane_f) * 2 + 1];
    const int k_i_l = 8192 + pop_i_l - cst_i_l, k_q_l = 8192 + pop_q_l - cst_q_l;
#pragma unroll
    for (int f = 0; f < NF; f++) {
      m_i[f] = (__builtin_amdgcn_readlane(k_i_l, f) - 2 * zi[f] - (int)f_sum) >> 1;
      m_q[f] = (__builtin_amdgcn_readlane(k_q_l, f) - 2 * zq[f] - (int)f_sum) >> 1;
    }
  }

  // ---- sixteen offsets: epilogue, then one recurrence step ---------------------------------------------------------------
  // The data bytes a lane's corrections look at only depend on the byte offset, i.e. on (bin, half): fetched when the
  // half begins, kept packed: prev = bytes o - 2, o - 1 of I | of Q << 16; cur = byte o of I | of Q << 8.
  u32 prev_iq[NF], cur_iq[NF];
  uint4 et = etab_s[(size_t)lane_f * 16 + 0];
#pragma unroll 1
  for (int t0 = 0; t0 < 16; t0++) {
    const int b = t0 & 7, half = t0 >> 3;
    const u32 low_mask = (1u << b) - 1u;
    const u32 high_mask = (0xFFFFu << b) & 0xFFFFu;
    const u32 r_last = (c1021 ? low_mask : 0u) | (c1022 ? high_mask : 0u);
    const int o = 2 * q + half;
    const bool in_win = emit && o >= P.win_start && o < P.win_stop;
    const bool odd_tail = half && q > 0;
    const u32 key_lo = (u32)(2047 - o);
    if (b == 0) {
      const u32 *rec = rec_s + (size_t)half * kRecDwords + (emit ? (o + half) >> 1 : 0);
#pragma unroll
      for (int f = 0; f < NF; f++) {
        const int ff = f < nf ? f : 0;
        const u32 x0 = rec[(size_t)ff * 2 * kRecDwords], x1 = rec[(size_t)ff * 2 * kRecDwords + 1];
        prev_iq[f] = (x0 & 0xFFu) | ((x0 >> 8) & 0xFF00u) | ((x0 << 8) & 0xFF0000u) | (x0 & 0xFF000000u);
        cur_iq[f] = x1 & 0xFFFFu;
      }
    }
    const uint4 et_cur = et;
    if (t0 < 15)
      et = etab_s[(size_t)lane_f * 16 + t0 + 1];   // next offset's uniforms: in flight during the plane sweep below
#pragma unroll
    for (int f = 0; f < NF; f++) {
      if (f < nf) {
        const int base_i = __builtin_amdgcn_readlane((int)et_cur.x, f), base_q = __builtin_amdgcn_readlane((int)et_cur.y, f);
        int ci = base_i + __mul24(m_i[f], -2);
        int cq = base_q + __mul24(m_q[f], -2);
        if (c1022) {   // quirk Q5
          ci += 2 * (int)__popc(cur_iq[f] & low_mask) - b;
          cq += 2 * (int)__popc((cur_iq[f] >> 8) & low_mask) - b;
        }
        if (half) {    // quirk Q3
          const u32 wrap_tab_i = (u32)__builtin_amdgcn_readlane((int)et_cur.z, f);
          const u32 wrap_tab_q = (u32)__builtin_amdgcn_readlane((int)et_cur.w, f);
          ci -= (int)__builtin_amdgcn_ubfe(wrap_tab_i, sel8, 8u);
          cq -= (int)__builtin_amdgcn_ubfe(wrap_tab_q, sel8, 8u);
          ci -= odd_tail ? (int)__popc((prev_iq[f] & 0xFFFFu) ^ r_last) : 0;
          cq -= odd_tail ? (int)__popc((prev_iq[f] >> 16) ^ r_last) : 0;
        }
        const u32 val = in_win ? (u32)mag8_fast(ci, cq) : 0u;
        const u32 key = in_win ? (val << 11) | key_lo : 0u;
        u32 k = wave_max_to_lane63(key);
        u32 t = wave_sum_to_lane63(val);
        if (lane == 63) {
          if (half == 0) {
            sh.part[f][b][0] = k;
            sh.part[f][b][1] = t;
          } else {
            const u32 k0 = sh.part[f][b][0];
            k = k0 > k ? k0 : k;
            t += sh.part[f][b][1];
            const size_t idx = ((size_t)(search * P.n_prn + slot) * P.n_dopp + f0 + f) * 8 + b;
            atomicMax(&P.keyacc[idx], k);
            atomicAdd(&P.sumacc[idx], t);
          }
        }
      }
    }
    if (t0 == 15)
      break;
    // recurrence step: plane t0
    int zi[NF], zq[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) {
      zi[f] = 0;
      zq[f] = 0;
    }
    const int c = plane_class(t0);
    u32 total;
    ds_plane<NF>(sh, lane, rows_chunk + (size_t)c * kRows * kRowDwords, xpl_s + t0 * 32, c >= 4, zi, zq, total);
    const int d_total = (from_next_lane((int)total) - (int)total) >> 1;
#pragma unroll
    for (int f = 0; f < NF; f++) {
      // X(q + 1) - X(q) = -(H(q + 1) - H(q)) / 2,  H = const + 2 Z + P(1023)
      m_i[f] -= (from_next_lane(zi[f]) - zi[f]) + d_total;
      m_q[f] -= (from_next_lane(zq[f]) - zq[f]) + d_total;
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host: boundary tables of a Doppler grid
// ---------------------------------------------------------------------------------------------------------------------
bool build_ds_tables(int dopp_min_hz, int dopp_step_hz, int n_dopp, std::vector<uint32_t> &rows, std::vector<int32_t> &cst0)
{
  static const uint32_t kCos[4] = {0x09999999u, 0xCCCCCCCCu, 0x66666666u, 0x33333333u};   // PM/GPS/gps_misc.c:216-217
  static const uint32_t kSin[4] = {0x33333333u, 0x09999999u, 0xCCCCCCCCu, 0x66666666u};
  static const int kRep[kClasses] = {0, 1, 2, 3, 12, 15};   // one plane of each class
  const int n_chunks = (n_dopp + kNF - 1) / kNF;
  rows.assign((size_t)n_chunks * kClasses * kRows * kRowDwords, 0u);
  cst0.assign((size_t)n_dopp * 2, 0);
  for (int f = 0; f < n_dopp; f++) {
    const float freq_hz = (float)(kIfHz + dopp_min_hz + f * dopp_step_hz);
    const volatile float quot = freq_hz / 0.003810972f;                       // IEEE binary32, as the device computes it
    const uint32_t step_word = (uint32_t)((uint64_t)(uint32_t)quot * 32ull);
    const int chunk = f / kNF, fl = f % kNF;
    for (int s = 0; s < 2; s++) {
      const uint32_t *pat = s == 0 ? kCos : kSin;
      for (int j = 0; j < 16; j++)
        for (int m = 0; m < kWords32; m++) {
          const uint32_t p = pat[(uint32_t)(step_word * (uint32_t)m) >> 30];
          cst0[f * 2 + s] += (int)((p >> j) & 1u) + (int)((p >> (16 + j)) & 1u);
        }
      for (int c = 0; c < kClasses; c++) {
        const int j = kRep[c];
        int cp_prev = 0, ce_prev = 0;
        for (int m = 0; m < kWords32; m++) {
          const uint32_t p = pat[(uint32_t)(step_word * (uint32_t)m) >> 30];
          const int se = (int)((p >> j) & 1u), so = (int)((p >> (16 + j)) & 1u);
          const int cp = 1 - 2 * so, ce = 2 * (so - se);
          if (m > 0 && (cp != cp_prev || ce != ce_prev)) {
            const int zp = (cp_prev - cp) / 2, ze = (ce_prev - ce) / 2;
            const int K = 2 * m, w = K >> 5, sh = (K & 31) >> 1;
            uint32_t *r = &rows[(((size_t)chunk * kClasses + c) * kRows + w) * kRowDwords];
            const int byte_shift = 8 * (fl & 3);
            const uint32_t old = (r[kRowOps + (fl >> 2)] >> byte_shift) & 0xFFu;
            const uint32_t old_e = (r[kRowOpsE + (fl >> 2)] >> byte_shift) & 0xFFu;
            if (((r[0] >> fl) & 1u) && (old & 15u) != (uint32_t)sh)
              return false;   // two sign changes of one bin inside one 32-chip word: |Doppler| too high for this form
            uint32_t op = (old & ~15u) | (uint32_t)sh;
            uint32_t ope = old_e;
            op |= (uint32_t)(zp & 3) << (s == 0 ? 4 : 6);
            ope |= (uint32_t)(ze & 3) << (s == 0 ? 4 : 6);
            r[kRowOps + (fl >> 2)] = (r[kRowOps + (fl >> 2)] & ~(0xFFu << byte_shift)) | (op << byte_shift);
            r[kRowOpsE + (fl >> 2)] = (r[kRowOpsE + (fl >> 2)] & ~(0xFFu << byte_shift)) | (ope << byte_shift);
            r[0] |= 1u << fl;
            if (ze)
              r[1] |= 1u << fl;
          }
          cp_prev = cp;
          ce_prev = ce;
        }
        const int zp_end = (cp_prev - 1) / 2, ze_end = ce_prev / 2;   // carrier bit 0 from k = 1022 on
        if (zp_end || ze_end) {
          uint32_t *r = &rows[(((size_t)chunk * kClasses + c) * kRows + 32) * kRowDwords];   // K = 1022: word 31, bit 30
          const int byte_shift = 8 * (fl & 3);
          uint32_t op = ((r[kRowOps + (fl >> 2)] >> byte_shift) & 0xFFu) | 15u;
          uint32_t ope = (r[kRowOpsE + (fl >> 2)] >> byte_shift) & 0xFFu;
          op |= (uint32_t)(zp_end & 3) << (s == 0 ? 4 : 6);
          ope |= (uint32_t)(ze_end & 3) << (s == 0 ? 4 : 6);
          r[kRowOps + (fl >> 2)] = (r[kRowOps + (fl >> 2)] & ~(0xFFu << byte_shift)) | (op << byte_shift);
          r[kRowOpsE + (fl >> 2)] = (r[kRowOpsE + (fl >> 2)] & ~(0xFFu << byte_shift)) | (ope << byte_shift);
          r[0] |= 1u << fl;
          if (ze_end)
            r[1] |= 1u << fl;
        }
      }
    }
  }
  return true;
}

void launch_acq_ds(hipStream_t s, const DsParams &prm, const uint8_t *d_if, int if_format, int search_stride_blocks,
                   int dopp_min_hz, int dopp_step_hz, size_t n_peaks, gpsx_peak_t *d_peaks)
{
  hipLaunchKernelGGL(k_ds_prepare, dim3((unsigned)(prm.n_search * prm.n_dopp)), dim3(256), 0, s, d_if, if_format,
                     search_stride_blocks, prm.n_dopp, dopp_min_hz, dopp_step_hz, prm.hdr, prm.rec, prm.xpl, prm.etab);
  (void)hipMemsetAsync(prm.keyacc, 0, 2 * n_peaks * sizeof(uint32_t), s);   // sumacc = keyacc + n_peaks
  const long waves = (long)prm.n_search * prm.n_prn * prm.n_chunks * kTiles;
  hipLaunchKernelGGL((k_acq_ds<kNF>), dim3((unsigned)waves), dim3(64), 0, s, prm);
  launch_acq_finalize(s, prm.keyacc, prm.sumacc, n_peaks, d_peaks);
}

}  // namespace gpsx
