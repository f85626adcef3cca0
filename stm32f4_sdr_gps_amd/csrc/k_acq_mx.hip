// k_acq_mx.hip -- the acquisition grid kernel for the fine (16368-phase) sweep with the correlations on the matrix cores.
//
// Same contract as k_acq / k_acq_poly: per (search, PRN, Doppler, replica bit shift) the triplet correlation_search
// (PM/GPS/gps_misc.c:155-191) returns, bit for bit; same preamble (capture -> LDS, carrier wipe-off K3), same
// per-hypothesis corrections of the reference's quirks and the same magnitude.  What changes is where the sums
//      M_t0(q) = sum_c chip_p[c] * S_t0[q + c],   S_t0[k] = pop(D[16 k + t0, +16)),   sample offset s = 16 q + t0
// come from.  In the polyphase form (k_acq_poly.hip)  M_{t0+1}(q) - M_t0(q) = sum_c chip_p[c] * e_t0[q + c]  with
// e_t0[k] = d_t0[k + 1] - d_t0[k] in {-1, 0, +1},  d_t0[k] = D(16 k + t0): for the 32 PRNs of a workgroup and the 1023
// chip offsets q that is a GEMM   C[p][q] += A[p][c] * B[c][q],   A = chips (32 x 1024),  B = the TOEPLITZ matrix
// B[c][q] = e[(q + c) mod 1023]  of ONE 1023-element vector -- and C, kept in the accumulator registers from one sample
// offset to the next, IS M_t0.  Operands are MX-FP4 (E2M1: 0, +-1, 2, 3, 4 exact; block scale E8M0 2^0 or 2^2),
// v_mfma_scale_f32_32x32x64_f8f6f4 accumulates in f32: every partial sum is an integer below 2^24, so the result is
// exact whatever the order (tools/microbench/mfma_fp4_corr.hip checks layouts and exactness on the device).
// M for the first offset takes two passes: S = (S & 3) + 4 (S >> 2), both parts FP4-exact, the second at a block scale.
//
// Data movement: B never exists.  The nibble vector (2048 entries: one period + its wrap-around) sits in LDS in eight
// copies, copy c starting at nibble c, so that lane (n, h) of tile (Q, kappa) -- column q = 32 Q + n, chips
// 64 kappa + 32 h .. + 31 -- finds its 32 nibbles dword-aligned at dword 4 (Q + 2 kappa + h) + n / 8 of copy n % 8, bank
// conflict free.  Tile (Q, kappa) reads what (Q - 2, kappa + 1) reads: a wave owns q-tiles Q0, Q0 + 2, Q0 + 4, Q0 + 6
// and walks the anti-diagonals f = Q + 2 kappa, 19 fragment loads for 64 MFMAs per stream.
//
// Nothing linear is left to the vector ALU.  What the reference's quirks add to a popcount is linear in chip bits
// (DESIGN.md 4.1), so it rides in the same accumulators: the accumulator of (q, PRN p) holds, after the pass of sample
// offset t0, exactly  cnt(q, t0, p) - 8184  -- the number gps_correlation8 clips and squares:
//   * the vector carries -2 e (values 0, +-2), the accumulators start at pop(D) + 8192 - 8184 (or at -2^20 for byte offsets
//     outside the search window: they clip to zero by themselves);
//   * odd byte offsets skip the replica word at the wrap (quirk Q3), a popcount against chips (1021 - q, 1022 - q): two
//     impulses of -1 / +1 in the vector at entries 1021 and 1022 (not in their wrap-around copies) per step;
//   * the terms "PRN flag x per-offset value" (chip 1022: quirk Q5 and the tail word of Q3; chip 1021: the tail word) are
//     one more K step of the GEMM: A column 0 of lane half 0 = chip 1022 of the PRN, of half 1 = chip 1021, B = the
//     per-offset deltas (0, +-1, +-2) read as one byte per lane;
//   * at the switch from even to odd byte offsets (t0 = 8) every such term jumps; that one step is patched into the
//     accumulators by the vector ALU (mx_half_switch).
// The epilogue of a sample offset is then: clip, square, add, correctly rounded root, truncate, (add the running sum of
// earlier blocks,) pack the key, max, sum.
//
// A workgroup = 8 waves = one (search, Doppler) pair x 32 PRNs (four 8-PRN sharding units) x all 16 sample offsets.
// Lane (n, h) of a wave holds, per tile, column q for the 16 PRNs p = (r & 3) + 8 (r >> 2) + 4 h, r = 0..15 -- the
// per-offset work (corrections, window test) is shared by 16 hypotheses.  The matrix pipe and the vector ALU of a SIMD
// are separate: waves 0..3 and 4..7 (one of each per SIMD) run half a step apart, one group's MFMA pass under the
// other's epilogue, with one barrier per step.
//
// Forms (k_acq_mx<MODE>): 0 single block, one workgroup per cluster (the headline sweep); 3 / 1 a workgroup walks the blocks
// of its search, running sums as 16- / 24-bit records through HBM scratch; 2 a workgroup per (cluster, block), magnitudes out
// for k_acq_vals_search; 4 the byte-phase grid (sample offsets 0 and 8, each started directly from its own block sums; one
// persistent workgroup per CU runs its clusters as ONE software pipeline: mx_byte_pipe); 5 small launches: 2 / 4 / 8 workgroups per cluster, each started directly
// at its own sample offset (mx_direct_terms: every quirk term as a start value), results merged through global planes.
#if !defined(GPSX_LAB) && (defined(WALK_ABL_NO_LOAD) || defined(WALK_ABL_NO_STORE) || defined(GPSX_MX_ABLATIONS) || \
                           defined(GPSX_MX_NO_PIECES) || defined(GPSX_MX_TIMELINE) || defined(MX_BUILD_BEHIND) || defined(GPSX_MX_NT) || \
                           defined(MX_VARIANT_B) || defined(WALK_ABL_ALIAS) || defined(MXW_ABL) || defined(GPSX_MX_CYCLES))
#error "timing ablations / instrumented variants of k_acq_mx (some give wrong results) build with -DGPSX_LAB only: tools/build_variant.sh"
#endif
#include <cstdlib>

#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"

namespace gpsx {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kMxThreads = 512;
constexpr int kMxTiles = 4;            // q-tiles (32 chip offsets each) per wave
constexpr int kCopyDwords = 260;       // one shifted copy of the nibble vector: 256 dwords + slack; 260 = 4 (mod 32): the 32
                                       // lanes of a fragment read (copy n % 8, dword n / 8 + ..) hit 32 different banks
constexpr int kVecDwords = 258;        // dwords of copy 0 that the shifted copies are cut from
constexpr int kPlaneWordsMx = 66;      // polyphase bit plane: 1023 bits + circular extension to 2112
constexpr u32 kScaleOne = 0x7F7F7F7Fu;   // E8M0 127 = 2^0
// The accumulators hold (cnt - 8184) / 8192: the A operand's block scale is 2^-13 (E8M0 114).  A power of two changes no
// rounding anywhere, and it puts every in-window value inside (-1, 1), where one v_mul_f32 c, |c| with the clamp modifier
// IS the reference's clip-and-square (mx_clip_square).
constexpr u32 kScaleA = 0x72727272u;
constexpr float kAccScale = 1.0f / 8192.0f;
constexpr float kUnscaleSq = 67108864.0f;   // 2^26: scaled squares -> integers
constexpr u32 kScaleEight = 0x82828282u; // E8M0 130 = 2^3
constexpr int kPasses = 17;            // 2 for the first offset + 15 recurrence steps

struct MxShared {
  uint16_t x[1024];                      // raw IF block (sign plane)
  u32 d[2][514];                         // wiped I / Q streams (word 511 = wrap-around copy, then zero pad)
  u32 plane[2][16][kPlaneWordsMx];       // d_t0 for the 16 sample offsets, circularly extended
  u32 base[2][kCopyDwords];              // nibble vector of the pass in preparation (copy 0), I / Q
  u32 e8[2][2][8][kCopyDwords];          // [buffer][stream][copy][dword]
  u32 corr[2][2][2][128];                // [buffer][stream][chip 1022 / chip 1021 term][q / 8]: FP4 codes of the step's deltas
  v4i chips_a[16][2][32];                // A fragments: [kappa][h][PRN] = 32 FP4 chips 64 kappa + 32 h ..
  u32 chip_t[1032];                      // chip_t[c + 1]: bit p = chip c of PRN p of this cluster; [0] = chip -1 = 0
  u32 ones[2];                           // pop(D) per stream
  u32 t_lut[768];                        // [0, 512): 9 adjacent bits -> FP4 codes of -2 (bit k+1 - bit k), k = 0..7;
                                         // [512, 768): 8 bits -> FP4 codes of 2 bit - 1 (mx_fill_tables)
  alignas(16) u32 part[8][32][2][32];              // (packed best key, sum) per bit shift, PRN and lane of the wave half that holds the
                                         // PRN: every lane folds its own results in with LDS atomics (no return value, no
                                         // conflicts), the 32 lanes meet once, when the workgroup writes its triplets
};

__device__ __forceinline__ v8i widen(v4i x) { return v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0}; }

__device__ __forceinline__ u32 lds_byte(const u32 *words, int byte_index)
{
  return (words[byte_index >> 2] >> ((byte_index & 3) * 8)) & 0xFFu;
}

// 8 bits -> 8 nibbles (bit k -> bit 4 k)
__device__ __forceinline__ u32 spread8(u32 x)
{
  u32 t = (x | (x << 12)) & 0x000F000Fu;
  t = (t | (t << 6)) & 0x03030303u;
  t = (t | (t << 3)) & 0x11111111u;
  return t;
}

__device__ __forceinline__ int wrap1023(int i)   // i < 3 * 1023
{
  i = i >= 2 * kChips ? i - 2 * kChips : i;
  return i >= kChips ? i - kChips : i;
}

// bits [pos, pos + 9) of a plane
__device__ __forceinline__ u32 plane_bits9(const u32 *pl, int pos)
{
  return __builtin_amdgcn_alignbit(pl[(pos >> 5) + 1], pl[pos >> 5], (u32)(pos & 31)) & 0x1FFu;
}

// ---- per block: capture -> LDS, wipe-off, polyphase planes -------------------------------------------------------------
// (in two parts: the block's load goes out together with the cluster's tables -- one global-memory latency, not two)
__device__ __forceinline__ void mx_load_block(MxShared &sh, const uint8_t *blk, int if_format, int tid)
{
  for (int i = tid; i < 1024; i += kMxThreads)
    sh.x[i] = i < kWords16 ? load_sign16(blk, i, if_format) : (uint16_t)0;
  if (tid < 2)
    sh.ones[tid] = 0;
}
__device__ void mx_wipe_block(MxShared &sh, u32 step_word, int tid, int lane)
{
  {
    const u32 *x32 = reinterpret_cast<const u32 *>(sh.x);
    u32 ones_i = 0, ones_q = 0;
    for (int w = tid; w < 514; w += kMxThreads) {
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
    ones_i = wave_sum_to_lane63(ones_i);   // (DPP: no lane-address constants to keep in -- or spill from -- registers)
    ones_q = wave_sum_to_lane63(ones_q);
    if (lane == 63) {
      atomicAdd(&sh.ones[0], ones_i);
      atomicAdd(&sh.ones[1], ones_q);
    }
  }
  __syncthreads();
  if (tid < 2)
    sh.d[tid][511] = sh.d[tid][0] << 16;   // samples 16352..16367 are zero, then the stream wraps to sample 0
  __syncthreads();
  // plane[iq][t0] bit i = D(16 (i mod 1023) + t0), i < 2112.  First period: word w of offset t0 takes bit t0 and bit 16 + t0
  // of the stream words 16 w .. 16 w + 15 (bit 1023 = D(16368 + t0) is the wrap-around copy in word 511: D(t0), as it has
  // to be); the 16 threads of a word read the same 16 addresses (LDS broadcast).
  for (int m = tid; m < 2 * 32 * 16; m += kMxThreads) {
    const int t0 = m & 15, w = (m >> 4) & 31, iq = m >> 9;
    const u32 *src = &sh.d[iq][16 * w];
    u32 bits = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const u32 sk = src[k];
      bits |= ((sk >> t0) & 1u) << (2 * k);
      bits |= ((sk >> (16 + t0)) & 1u) << (2 * k + 1);
    }
    sh.plane[iq][t0][w] = bits;
  }
  __syncthreads();
  // circular extension: word w >= 32 = the 32 bits from position 32 w mod 1023 of the 1023-bit period
  for (int m = tid; m < 2 * 16 * (kPlaneWordsMx - 32); m += kMxThreads) {
    const int w = 32 + m % (kPlaneWordsMx - 32);
    const int r = m / (kPlaneWordsMx - 32);
    const u32 *pl = sh.plane[r >> 4][r & 15];
    const int pos = 32 * w - (w >= 64 ? 2 * kChips : kChips);
    const int lo = pos >> 5;
    u32 v = __builtin_amdgcn_alignbit(lo < 31 ? pl[lo + 1] : 0u, pl[lo], (u32)(pos & 31));
    if (pos + 32 > kChips) {   // the period ends inside the word: its first bits follow
      const int k = kChips - pos;
      v = (v & ((1u << k) - 1u)) | (pl[0] << k);
    }
    sh.plane[r >> 4][r & 15][w] = v;
  }
  // (the caller's next barrier publishes the planes)
}

// FP4 (E2M1) code of a small integer: 0, +-1, +-2, +-3, +-4 (and 6)
__device__ __forceinline__ u32 fp4_code(int v)
{
  const u32 m = (u32)(v < 0 ? -v : v);
  return ((0x0765420u >> (4u * (m > 5u ? 5u : m))) & 0xFu) | (v < 0 ? 8u : 0u);   // |v|: 0 1 2 3 4 6 -> 0 2 4 5 6 7
}

// The vector builders' lookup tables (once per workgroup): what they replace is the bit -> nibble spreading, a dozen
// vector instructions per dword of the vectors -- and the vectors are built once per sample offset next to the MFMA passes,
// by waves that have better things to do.
__device__ void mx_fill_tables(MxShared &sh, int tid)
{
  for (int w = tid; w < 512; w += kMxThreads) {
    const u32 cur = (u32)w & 0xFFu, nxt = ((u32)w >> 1) & 0xFFu;
    const u32 plus = spread8(nxt & ~cur), minus = spread8(cur & ~nxt);   // e = +1 -> -2 (code C), e = -1 -> +2 (code 4)
    sh.t_lut[w] = (plus << 2) | (plus << 3) | (minus << 2);
  }
  for (int x = tid; x < 256; x += kMxThreads)
    sh.t_lut[512 + x] = (spread8((u32)x) << 1) | (spread8(~(u32)x & 0xFFu) * 0xAu);   // +1 -> code 2, -1 -> code A
}

// ---- per pass: the nibble vector, copy 0 (phase 1), then its eight shifted copies (phase 2) --------------------------------
// pass 0: -2 (S_0 & 3), pass 1: -(S_0 >> 2) at scale 2^3, pass p >= 2 (producing sample offset t0 = p - 1 from plane
// p - 2): -2 e_{p-2}, plus the wrap-word impulses when t0 is 9..15; and the byte vectors of the extra K step.
__device__ void mx_vector_phase1(MxShared &sh, int pass, int buf, int tid, int nthreads)
{
  const int t0 = pass - 1;
  // nibbles 0 .. 2055 are ever read (dword 4 * 62 + 3 + 3 of copy 7): 258 dwords of copy 0; then the 2 x 128 dword pairs of
  // the extra K step
  for (int m = tid; m < 2 * kVecDwords + 2 * 128; m += nthreads) {
    if (m < 2 * kVecDwords) {
      const int iq = m >= kVecDwords, dw = m - iq * kVecDwords;
      u32 packed = 0;
      if (pass < 2) {
        const u32 *dd = sh.d[iq];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int k = wrap1023(8 * dw + e);
          const int pos = 16 * k;
          const u32 sum = pop16(__builtin_amdgcn_alignbit(dd[(pos >> 5) + 1], dd[pos >> 5], (u32)(pos & 31)));
          // pass 0: -2 (S & 3) = 0, -2, -4, -6 -> codes 0, C, E, F;  pass 1: -(S >> 2) = 0 .. -4 -> codes 0, A, C, D, E
          const u32 code = pass == 0 ? (0xFEC0u >> (4u * (sum & 3u))) & 0xFu : (0xEDCA0u >> (4u * (sum >> 2))) & 0xFu;
          packed |= code << (4 * e);
        }
      } else {
        const u32 w = plane_bits9(sh.plane[iq][pass - 2], 8 * dw);
        packed = sh.t_lut[w];
        if (dw == 127 && t0 >= 9) {
          // entries 1021 / 1022 (nibbles 5 / 6 of this dword, first period only): the skipped wrap word's coefficients
          // alpha_b = b on chip 1021 - q and beta_b = const - b on chip 1022 - q move by +1 / -1 per step; they are
          // subtracted from the count: -1 / +1 here
          const u32 cur = w & 0xFFu, nxt = (w >> 1) & 0xFFu;
          const int e5 = (int)((nxt >> 5) & 1u) - (int)((cur >> 5) & 1u), e6 = (int)((nxt >> 6) & 1u) - (int)((cur >> 6) & 1u);
          packed = (packed & ~0x0FF00000u) | (fp4_code(-2 * e5 - 1) << 20) | (fp4_code(-2 * e6 + 1) << 24);
        }
      }
      sh.base[iq][dw] = packed;
    } else {
      // extra K step: deltas of  c1022 * A'(q)  and  c1021 * B'(q)  (see mx_half_switch for the terms themselves), eight
      // chip offsets per dword:
      //   t0 = 1..7, 9..15:  A_b = 2 pop(byte_o & low_b) - b grows by 2 D(8 o + b) - 1 = 2 d[q] - 1
      //   t0 = 9..15, q > 0: the tail word (o - 2, o - 1) of odd offsets, whose bit b is d[q - 1]: A' += 1 - 2 d[q - 1]
      //                      (together 2 (d[q] - d[q - 1])), B' grows by 2 d[q - 1] - 1
      const int mm = m - 2 * kVecDwords;
      const int iq = mm >> 7, dw = mm & 127;
      u32 ca = 0, cb = 0;
      if (pass >= 2 && t0 != 8) {
        const u32 *pl = sh.plane[iq][pass - 2];
        // bits 8 dw - 1 .. 8 dw + 7 of the plane (bit -1 = 0): d[q] = bit k + 1, d[q - 1] = bit k for the eight q of this dword
        const u32 w = dw ? plane_bits9(pl, 8 * dw - 1) : (pl[0] << 1) & 0x1FFu;
        const u32 exist = dw == 127 ? 0x0FFFFFFFu : 0xFFFFFFFFu;   // q = 1023 does not exist
        if (t0 < 8) {
          ca = sh.t_lut[512 + (w >> 1)] & exist;
        } else {
          const u32 diff = sh.t_lut[w];                             // -2 (d - dm)
          ca = (diff ^ ((diff & 0x44444444u) << 1)) & exist;        // 2 (d - dm): the sign bit of the non-zero codes flips
          cb = sh.t_lut[512 + (w & 0xFFu)] & exist;                     // 2 dm - 1
          if (dw == 0) {                                            // q = 0 has no tail word: A' grows by 2 d - 1, B' stays
            ca = (ca & ~0xFu) | ((w & 2u) ? 0x2u : 0xAu);
            cb &= ~0xFu;
          }
        }
      }
      sh.corr[buf][iq][0][dw] = ca;
      sh.corr[buf][iq][1][dw] = cb;
    }
  }
}

__device__ void mx_vector_phase2(MxShared &sh, int buf, int tid, int nthreads)
{
  for (int m = tid; m < 2 * 256; m += nthreads) {
    const int iq = m >> 8, j = m & 255;
    const u32 lo = sh.base[iq][j], hi = sh.base[iq][j + 1];
#pragma unroll
    for (int c = 0; c < 8; c++)
      sh.e8[buf][iq][c][j] = c ? __builtin_amdgcn_alignbit(hi, lo, 4u * (u32)c) : lo;
  }
}

// ---- one MFMA pass: acc[stream][tile] += chips x Toeplitz(vector) -------------------------------------------------------
typedef __attribute__((address_space(3))) const u32 lds_cu32;

// an LDS address the compiler cannot see through: what is added to it afterwards are small constants that fit the DS
// instructions' offset fields (left alone it rebuilds "variable part + offset of the array in the LDS block + 32 s" with a
// v_add per load: the array's offset does not fit the 8-bit dword offsets of ds_read2_b32)
__device__ __forceinline__ lds_cu32 *lds_opaque(const u32 *p)
{
  u32 a = (u32)(size_t)(lds_cu32 *)p;
  asm volatile("" : "+v"(a));
  return (lds_cu32 *)(size_t)a;
}
__device__ __forceinline__ v4i lds_frag(lds_cu32 *w, int dw)   // four dwords, dword aligned only
{
  return v4i{(int)w[dw], (int)w[dw + 1], (int)w[dw + 2], (int)w[dw + 3]};
}

// ---- the vector of pass p_vec >= 2 in one phase ------------------------------------------------------------------------
// Same values as mx_vector_phase1 + phase2 (which build the first two vectors, before the loop), without the copy-0 round
// trip through LDS: thread (stream, j) looks up dwords j and j + 1 of copy 0 itself and writes dword j of the eight shifted
// copies; thread (stream, term, dw) one dword of the extra K step's vectors.  512 threads, three dependent LDS accesses.
// dword 127 of a vector, sample offsets 9..15: the wrap word's impulses at entries 1021 / 1022 (see mx_vector_phase1)
__device__ __forceinline__ u32 mx_patch_wrap(u32 packed, u32 w9, bool patch)
{
  const u32 i5 = 1u + ((w9 >> 6) & 1u) - ((w9 >> 5) & 1u), i6 = 1u + ((w9 >> 7) & 1u) - ((w9 >> 6) & 1u);   // e + 1
  const u32 c5 = (0xDA2u >> (4u * i5)) & 0xFu;   // -2 e - 1 = 1, -1, -3 -> codes 2, A, D
  const u32 c6 = (0xA25u >> (4u * i6)) & 0xFu;   // -2 e + 1 = 3, 1, -1 -> codes 5, 2, A
  const u32 patched = (packed & ~0x0FF00000u) | (c5 << 20) | (c6 << 24);
  return patch ? patched : packed;
}

__device__ __forceinline__ void mx_vector_build(MxShared &sh, int p_vec, int tid)
{
  const int t0 = p_vec - 1, buf = p_vec & 1;
  const bool late = t0 >= 9;
  const int iq = tid >> 8, j = tid & 255, which = (tid >> 7) & 1, dwc = tid & 127;
  const u32 *pl = sh.plane[iq][p_vec - 2];
  // 17 plane bits from 8 j: the 9-bit windows of dwords j and j + 1
  const u32 x = __builtin_amdgcn_alignbit(pl[(j >> 2) + 1], pl[j >> 2], 8u * (u32)(j & 3));
  // bits 8 dwc - 1 .. 8 dwc + 7 (bit -1 = 0) for the extra K step: d[q] = bit k + 1, d[q - 1] = bit k of the dword's eight q
  const int pos = dwc ? 8 * dwc - 1 : 0;
  const u32 y = __builtin_amdgcn_alignbit(pl[(pos >> 5) + 1], pl[pos >> 5], (u32)(pos & 31));
  const u32 cw = (dwc ? y : y << 1) & 0x1FFu;
  u32 lo = sh.t_lut[x & 0x1FFu], hi = sh.t_lut[(x >> 8) & 0x1FFu];
  // term 0: A' deltas: 2 d - 1 before the half switch, 2 (d - dm) after it; term 1: B' deltas 2 dm - 1 (after it only)
  u32 v = sh.t_lut[which ? 512u + (cw & 0xFFu) : (late ? cw : 512u + (cw >> 1))];
  if (late) {
    lo = mx_patch_wrap(lo, x & 0x1FFu, j == 127);
    hi = mx_patch_wrap(hi, (x >> 8) & 0x1FFu, j == 126);
    if (which == 0)
      v ^= (v & 0x44444444u) << 1;                         // -2 (d - dm) -> 2 (d - dm): the sign of the non-zero codes
    if (dwc == 0)                                          // q = 0 has no tail word: A' grows by 2 d - 1, B' stays
      v = (v & ~0xFu) | (which ? 0u : ((cw & 2u) ? 0x2u : 0xAu));
  }
  v &= dwc == 127 ? 0x0FFFFFFFu : 0xFFFFFFFFu;             // q = 1023 does not exist
  if ((which && !late) || t0 == 8)
    v = 0;
  sh.corr[buf][iq][which][dwc] = v;
  u32 *dst = &sh.e8[buf][iq][0][j];
#pragma unroll
  for (int c = 0; c < 8; c++)
    dst[c * kCopyDwords] = c ? __builtin_amdgcn_alignbit(hi, lo, 4u * (u32)c) : lo;
}

// ---- the two vectors of a DIRECT start at sample offset t0s, in one phase ---------------------------------------------------
// M_t0s(q) = sum_c chip[c] S_t0s[q + c] from the block sums S_t0s[k] = pop(D[16 k + t0s, +16)) themselves, as passes 0 and 1
// do it for t0s = 0 (mx_vector_phase1): which = 0: -2 (S & 3), which = 1: -(S >> 2) at block scale 2^3.  Thread (stream, j)
// builds dwords j and j + 1 of copy 0 (sixteen block sums) and writes dword j of the eight shifted copies -- no round trip
// through sh.base, no second barrier.  Used where a workgroup does not walk to an offset but starts there: offset 8 of the
// byte-phase grid (the reference's own search, PM/GPS/acquisition.c:280-312) and the second half of a split fine grid.
__device__ __forceinline__ void mx_vector_build_direct(MxShared &sh, int which, int t0s, u32 *e8_dst, int tid)
{
  const int iq = tid >> 8, j = tid & 255;
  const u32 *dd = sh.d[iq];
  u32 w2[2] = {0, 0};
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k = wrap1023(8 * j + e);
    const int pos = 16 * k + t0s;
    const u32 sum = pop16(__builtin_amdgcn_alignbit(dd[(pos >> 5) + 1], dd[pos >> 5], (u32)(pos & 31)));
    const u32 code = which == 0 ? (0xFEC0u >> (4u * (sum & 3u))) & 0xFu : (0xEDCA0u >> (4u * (sum >> 2))) & 0xFu;
    w2[e >> 3] |= code << (4 * (e & 7));
  }
  u32 *dst = e8_dst + (iq * 8) * kCopyDwords + j;   // [stream][copy][dword]
#pragma unroll
  for (int c = 0; c < 8; c++)
    dst[c * kCopyDwords] = c ? __builtin_amdgcn_alignbit(w2[1], w2[0], 4u * (u32)c) : w2[0];
}

// One anti-diagonal of a pass (fragment Q0 + 2 S): request the fragments of the next one, then the MFMAs of this one.
// The sched_group_barriers pin that order -- the DS reads first, (8 MFMAs = 260 cycles ahead of their use) -- which the
// scheduler, short of registers, would otherwise turn into "requested one MFMA before the wait": the LDS is kept busy by
// the four waves of the other role, a wave that waits for it at every step loses a third of the matrix pipe's time.
template <int S, int NT, u32 SCALE_A = kScaleA>
__device__ __forceinline__ void mx_pass_step(lds_cu32 *wi, lds_cu32 *wq, const v4i *ca, v4i (&a)[16], v4i &fi, v4i &fq,
                                             v16f (&acc)[2][NT], u32 scale_b)
{
  constexpr int kSteps = 16 + NT - 1;
  constexpr bool more = S + 1 < kSteps;
  v4i fi_next = fi, fq_next = fq;
  if constexpr (more) {
    fi_next = lds_frag(wi, 8 * (S + 1));
    fq_next = lds_frag(wq, 8 * (S + 1));
    if constexpr (S + 1 < 16)
      a[S + 1] = ca[(S + 1) * 64];                         // chips_a[S + 1][h][n]
  }
  constexpr int j_lo = S - 15 > 0 ? S - 15 : 0, j_hi = S < NT - 1 ? S : NT - 1;
#pragma unroll
  for (int j = j_lo; j <= j_hi; j++) {
    acc[0][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a[S - j]), widen(fi), acc[0][j], 4, 4, 0, SCALE_A, 0, scale_b);
    acc[1][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a[S - j]), widen(fq), acc[1][j], 4, 4, 0, SCALE_A, 0, scale_b);
  }
  if constexpr (more) {
    __builtin_amdgcn_sched_group_barrier(0x100, S + 1 < 16 ? 5 : 4, 0);   // DS reads
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * (j_hi - j_lo + 1), 0);   // MFMAs
  }
  __builtin_amdgcn_sched_barrier(0);
  fi = fi_next;
  fq = fq_next;
  if constexpr (more)
    mx_pass_step<S + 1, NT, SCALE_A>(wi, wq, ca, a, fi, fq, acc, scale_b);
}

// The same without a second set of fragment registers (the walk forms, whose prefetched sums leave none): the I fragment of the
// next anti-diagonal is requested INTO the registers of this one's as soon as its MFMAs have been issued, under the Q stream's
// MFMAs, the Q fragment under the next step's I MFMAs -- four MFMAs (130 cycles) of cover each instead of eight; the A fragment
// (registers of its own) a whole step ahead.
template <int S, int NT>
__device__ __forceinline__ void mx_pass_step_inplace(lds_cu32 *wi, lds_cu32 *wq, const v4i *ca, v4i (&a)[16], v4i &fi, v4i &fq,
                                                     v16f (&acc)[2][NT], u32 scale_b)
{
  constexpr int kSteps = 16 + NT - 1;
  constexpr bool more = S + 1 < kSteps;
  if constexpr (more && S + 1 < 16)
    a[S + 1] = ca[(S + 1) * 64];                           // chips_a[S + 1][h][n]
  constexpr int j_lo = S - 15 > 0 ? S - 15 : 0, j_hi = S < NT - 1 ? S : NT - 1;
#pragma unroll
  for (int j = j_lo; j <= j_hi; j++)
    acc[0][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a[S - j]), widen(fi), acc[0][j], 4, 4, 0, kScaleA, 0, scale_b);
  if constexpr (more)
    fi = lds_frag(wi, 8 * (S + 1));
#pragma unroll
  for (int j = j_lo; j <= j_hi; j++)
    acc[1][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a[S - j]), widen(fq), acc[1][j], 4, 4, 0, kScaleA, 0, scale_b);
  if constexpr (more)
    fq = lds_frag(wq, 8 * (S + 1));
  if constexpr (more) {
    if constexpr (S + 1 < 16)
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                 // DS read: the A fragment
    __builtin_amdgcn_sched_group_barrier(0x008, j_hi - j_lo + 1, 0);     // MFMAs, I
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                   // DS reads: the next I fragment
    __builtin_amdgcn_sched_group_barrier(0x008, j_hi - j_lo + 1, 0);     // MFMAs, Q
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                   // DS reads: the next Q fragment
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (more)
    mx_pass_step_inplace<S + 1, NT>(wi, wq, ca, a, fi, fq, acc, scale_b);
}

// The byte-phase form's two passes of a stage (low vector at 2^0, high vector at 2^3) as ONE walk over the anti-diagonals: the
// A fragments are fetched once instead of twice, there is no gap between the passes, and every fragment is requested into the
// registers of its predecessor as soon as that one's MFMAs have been issued -- under the twelve MFMAs of the other three streams.
template <int S, int NT>
__device__ __forceinline__ void mx_pass2_step(lds_cu32 *const (&w)[4], const v4i *ca, v4i (&a)[16], v4i (&f)[4], v16f (&acc)[2][NT])
{
  constexpr int kSteps = 16 + NT - 1;
  constexpr bool more = S + 1 < kSteps;
  if constexpr (more && S + 1 < 16)
    a[S + 1] = ca[(S + 1) * 64];                           // chips_a[S + 1][h][n]
  constexpr int j_lo = S - 15 > 0 ? S - 15 : 0, j_hi = S < NT - 1 ? S : NT - 1;
#pragma unroll
  for (int v = 0; v < 4; v++) {   // I low, Q low, I high, Q high
#pragma unroll
    for (int j = j_lo; j <= j_hi; j++)
      acc[v & 1][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a[S - j]), widen(f[v]), acc[v & 1][j], 4, 4, 0, kScaleA, 0,
                                                                      v < 2 ? kScaleOne : kScaleEight);
    if constexpr (more)
      f[v] = lds_frag(w[v], 8 * (S + 1));
  }
  if constexpr (more) {
    if constexpr (S + 1 < 16)
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                 // DS read: the A fragment
#pragma unroll
    for (int v = 0; v < 4; v++) {
      __builtin_amdgcn_sched_group_barrier(0x008, j_hi - j_lo + 1, 0);   // MFMAs of one stream
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                 // DS reads: its next fragment
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (more)
    mx_pass2_step<S + 1, NT>(w, ca, a, f, acc);
}
template <int NT>
__device__ __forceinline__ void mx_pass2(const MxShared &sh, int lane, int q0_tile, v16f (&acc)[2][NT], const u32 *e8_low,
                                         const u32 *e8_high)
{
  const int n = lane & 31, h = lane >> 5;
  const int off = (n & 7) * kCopyDwords + 4 * (q0_tile + h) + (n >> 3);
  lds_cu32 *const w[4] = {lds_opaque(e8_low + off), lds_opaque(e8_low + 8 * kCopyDwords + off), lds_opaque(e8_high + off),
                          lds_opaque(e8_high + 8 * kCopyDwords + off)};
  const v4i *ca = &sh.chips_a[0][h][n];
  v4i a[16];
  v4i f[4] = {lds_frag(w[0], 0), lds_frag(w[1], 0), lds_frag(w[2], 0), lds_frag(w[3], 0)};
  a[0] = ca[0];
  mx_pass2_step<0, NT>(w, ca, a, f, acc);
}

// AHEAD = false (the walk forms): mx_pass_step_inplace
// NT = q-tiles of this call (q0_tile + 2 j, j < NT): four everywhere but in the byte-phase form, which works in tile pairs
// SCALE_A: the A operand's block scale (the weighted extension keeps plain integers in its accumulators: 2^0; AHEAD only)
template <bool AHEAD, int NT, u32 SCALE_A = kScaleA>
__device__ __forceinline__ void mx_pass(const MxShared &sh, int buf, int lane, int q0_tile, v16f (&acc)[2][NT],
                                        u32 scale_b, v4i a_corr, bool with_corr, const u32 *e8_buf = nullptr)
{
  const int n = lane & 31, h = lane >> 5;
  // (e8_buf: a vector's eight shifted copies somewhere else than sh.e8[buf] -- the byte-phase form keeps four vectors)
  const u32 *e8 = e8_buf ? e8_buf : &sh.e8[buf][0][0][0];
  lds_cu32 *wi = lds_opaque(e8 + (n & 7) * kCopyDwords + 4 * (q0_tile + h) + (n >> 3));
  lds_cu32 *wq = lds_opaque(e8 + (8 + (n & 7)) * kCopyDwords + 4 * (q0_tile + h) + (n >> 3));
  const v4i *ca = &sh.chips_a[0][h][n];
  v4i a[16];
  if constexpr (AHEAD) {
    v4i fi = lds_frag(wi, 0), fq = lds_frag(wq, 0);
    a[0] = ca[0];
    mx_pass_step<0, NT, SCALE_A>(wi, wq, ca, a, fi, fq, acc, scale_b);
  } else {
    static_assert(SCALE_A == kScaleA, "the in-place walk is the sign-only grid's");
    v4i fi = lds_frag(wi, 0), fq = lds_frag(wq, 0);
    a[0] = ca[0];
    mx_pass_step_inplace<0, NT>(wi, wq, ca, a, fi, fq, acc, scale_b);
  }
  if (with_corr) {   // wave-uniform
    // the extra K step: only column 0 of each lane half of A is set (chip 1022 / chip 1021 of the PRN), so only the first
    // nibble of a lane's B window counts: the step's delta for (stream, term h, q)
    // (dword q >> 3 = 4 (q0_tile + 2 j) + n / 8 of the term's vector: one address, constant offsets per tile and stream)
    lds_cu32 *cw = lds_opaque(&sh.corr[buf][0][h][4 * q0_tile + (n >> 3)]);
#pragma unroll
    for (int j = 0; j < NT; j++) {
      // (nibble q & 7 of dword q >> 3 moved to nibble 0; what is left above it meets zero columns of A)
      const v4i gi = v4i{(int)(cw[8 * j] >> (4 * (n & 7))), 0, 0, 0};
      const v4i gq = v4i{(int)(cw[8 * j + 2 * 128] >> (4 * (n & 7))), 0, 0, 0};
      acc[0][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a_corr), widen(gi), acc[0][j], 4, 4, 0, kScaleA, 0,
                                                                   kScaleOne);
      acc[1][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a_corr), widen(gq), acc[1][j], 4, 4, 0, kScaleA, 0,
                                                                   kScaleOne);
    }
  }
}

// gps_correlation8's magnitude (PM/GPS/gps_misc.c:106-118) on the centred counts as the accumulators hold them (exact
// integers / 8192 in f32): one-sided clip and square in one instruction, the f32 sum of the two, the correctly rounded root
// (v_sqrt_f32 + the neighbour test, as mag8_fast), truncation.
__device__ __forceinline__ u32 root_trunc(float e);
// max(c, 0)^2 for |c| < 1 (and 0 for any c <= -1): c |c| clamped to [0, 1].  The product of the exact count with itself,
// rounded once: what (float)(I * I) is -- at 2^-26.
__device__ __forceinline__ float mx_clip_square(float c)
{
  // (v_mul_f32 c, |c| clamp: the median with 0 and 1 folds into the multiplication's clamp bit)
  return __builtin_amdgcn_fmed3f(c * __builtin_fabsf(c), 0.0f, 1.0f);
}
__device__ __forceinline__ float clip_square_sum(float ci, float cq)   // (I^2 + Q^2) / 2^26
{
  return mx_clip_square(ci) + mx_clip_square(cq);
}
__device__ __forceinline__ u32 mag8_f32(float ci, float cq)
{
  return root_trunc(clip_square_sum(ci, cq) * kUnscaleSq);
}

// (int) of the correctly rounded f32 root
__device__ __forceinline__ u32 root_trunc(float e)
{
  float r = __builtin_amdgcn_sqrtf(e);
  const float r_dn = __uint_as_float(__float_as_uint(r) - 1u);
  const float r_up = __uint_as_float(__float_as_uint(r) + 1u);
  const float res_dn = __builtin_fmaf(-r_dn, r, e);
  const float res_up = __builtin_fmaf(-r_up, r, e);
  r = res_dn <= 0.0f ? r_dn : r;
  r = res_up > 0.0f ? r_up : r;
  return (u32)(int)r;
}

// ---- the rounding mode of an epilogue ---------------------------------------------------------------------------------------
// The small-radius path wants floor(root) as an integer in the low mantissa bits of an f32: with the f32 rounding mode at
// "toward zero" that is ONE v_fma_f32 behind the root -- root * 2^13 (1 + 2^-22) + 2^23 -- where round-to-nearest took an add of
// 1/2 in front of the root (so that the approximate root of a square does not land below it) and an add of 2^23 - 1/2 behind
// it.  The factor is the guard: v_sqrt_f32 is good to one ulp (2^-23 relative), so root (1 + 2^-23) <= x <= root (1 + 1.5 *
// 2^-22), never below the true root, and below the next integer as long as 1.5 * 2^-22 < 1 / (2 (m + 1)^2): m + 1 < 1182, the
// small path ends at 1024.  Everything else on that path is exact in any mode (integers < 2^24 at a power-of-two scale).
// The exact path -- (float)(I * I), the f32 sum, the correctly rounded root -- is the reference's arithmetic and runs in
// round-to-nearest: it switches the mode back for its own instructions (its inputs and results go through the switching
// asm statements, so none of them can be scheduled outside the pair).  MODE.FP_ROUND[1:0]: 0 = nearest even, 3 = toward zero.
constexpr float kRootGuard = 8192.001953125f;   // 2^13 (1 + 2^-22): scaled root -> root, nudged up past v_sqrt_f32's ulp
__device__ __forceinline__ void mx_round_toward_zero()
{
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void mx_round_to_nearest()
{
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1" ::: "memory");
}
// 0x4B000000 + floor(root of e * 2^26), e * 2^26 < 2^20 an integer; needs mx_round_toward_zero()
__device__ __forceinline__ u32 root_bits_small(float e)
{
  return __float_as_uint(__builtin_fmaf(__builtin_amdgcn_sqrtf(e), kRootGuard, 8388608.0f));
}
// the exact path of N hypotheses, in round-to-nearest whatever the mode around it
template <int N>
__device__ __forceinline__ void mx_roots_exact(float (&ci)[N], float (&cq)[N], u32 (&mag)[N])
{
  static_assert(N == 4 || N == 8, "group size");
  if constexpr (N == 8)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1"
                 : "+v"(ci[0]), "+v"(ci[1]), "+v"(ci[2]), "+v"(ci[3]), "+v"(ci[4]), "+v"(ci[5]), "+v"(ci[6]), "+v"(ci[7]),
                   "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+v"(cq[3]), "+v"(cq[4]), "+v"(cq[5]), "+v"(cq[6]), "+v"(cq[7]));
  else
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1"
                 : "+v"(ci[0]), "+v"(ci[1]), "+v"(ci[2]), "+v"(ci[3]), "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+v"(cq[3]));
#pragma unroll
  for (int i = 0; i < N; i++)
    mag[i] = mag8_f32(ci[i], cq[i]);
  if constexpr (N == 8)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1"
                 : "+v"(mag[0]), "+v"(mag[1]), "+v"(mag[2]), "+v"(mag[3]), "+v"(mag[4]), "+v"(mag[5]), "+v"(mag[6]), "+v"(mag[7]));
  else
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1"
                 : "+v"(mag[0]), "+v"(mag[1]), "+v"(mag[2]), "+v"(mag[3]));
}

constexpr float kOutside = -1048576.0f * kAccScale;   // start value (scaled) of hypotheses outside the search window: stays
                                                      // below -1, clips to 0

// Start of a block: every accumulator = the part of  cnt - 8184  that does not depend on the code (even byte offsets)
// (ones: pop(D) of the two streams -- sh.ones, or the other block's pair in the pipelined byte-phase form)
template <int NT>
__device__ __forceinline__ void mx_init_acc(const u32 *ones, int lane, int q0_tile, v16f (&acc)[2][NT], int win_start,
                                            int win_stop)
{
  const int n = lane & 31;
  const float base_i = (float)((int)ones[0] + 8192 - kHalf) * kAccScale, base_q = (float)((int)ones[1] + 8192 - kHalf) * kAccScale;
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int q = 32 * (q0_tile + 2 * j) + n;
    const int o = 2 * q;
    const bool in_win = q < kChips && o >= win_start && o < win_stop;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      acc[0][j][r] = in_win ? base_i : base_i + kOutside;
      acc[1][j][r] = in_win ? base_q : base_q + kOutside;
    }
  }
}

// After the pass of sample offset 8 (the first odd byte offset, b = 0), before its epilogue: every quirk term jumps.
//   cnt = C0 + c1022 A_b(q)                                                                    even offsets o = 2 q
//   cnt = C0 + c1022 A_b(q) - [pop(W) + chip[1021 - q] alpha_b + chip[1022 - q] beta_b]        odd offsets o = 2 q + 1
//            - T(q) [pop(P) + c1021 (b - 2 pop(P & low_b)) + c1022 (16 - b - 2 pop(P & high_b))]
// A_b = 2 pop(byte_o & low_b) - b (quirk Q5); W = data bytes (2045, 0), the word at the wrap; P = data bytes (o - 2, o - 1),
// T = [q > 0]: the two replica words odd offsets skip (quirk Q3); alpha_b = b, beta_b = 16 - 2 pop(W) - b because the low
// byte of W (data byte 2045, never mixed) is zero.  At b = 0: A = 0, alpha = 0.
template <int NT>
__device__ __forceinline__ void mx_half_switch(const MxShared &sh, int lane, int q0_tile, v16f (&acc)[2][NT], int win_start,
                                               int win_stop)
{
  const int n = lane & 31, h = lane >> 5;
  const u32 *d_i = sh.d[0], *d_q = sh.d[1];
  const u32 wrap_i = (d_i[0] & 0xFFu) << 8, wrap_q = (d_q[0] & 0xFFu) << 8;
  const float beta0_i = (float)(16 - 2 * (int)__popc(wrap_i)) * kAccScale, beta0_q = (float)(16 - 2 * (int)__popc(wrap_q)) * kAccScale;
  const int popw_i = (int)__popc(wrap_i), popw_q = (int)__popc(wrap_q);
  const u32 f22 = sh.chip_t[1022 + 1] >> (4 * h);
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int q = 32 * (q0_tile + 2 * j) + n;
    const bool exists = q < kChips;
    const int qc = exists ? q : 0;
    const bool in0 = exists && 2 * q >= win_start && 2 * q < win_stop;
    const bool in1 = exists && 2 * q + 1 >= win_start && 2 * q + 1 < win_stop;
    // A_7 of the even offset goes, A_0 = 0 of the odd one comes
    int fa_i = -(2 * (int)__popc(lds_byte(d_i, 2 * qc) & 0x7Fu) - 7);
    int fa_q = -(2 * (int)__popc(lds_byte(d_q, 2 * qc) & 0x7Fu) - 7);
    int fk_i = -popw_i, fk_q = -popw_q;
    if (q > 0 && exists) {
      const u32 prev_i = lds_byte(d_i, 2 * qc - 1) | (lds_byte(d_i, 2 * qc) << 8);
      const u32 prev_q = lds_byte(d_q, 2 * qc - 1) | (lds_byte(d_q, 2 * qc) << 8);
      fk_i -= (int)__popc(prev_i);
      fk_q -= (int)__popc(prev_q);
      fa_i -= 16 - 2 * (int)__popc(prev_i);
      fa_q -= 16 - 2 * (int)__popc(prev_q);
    }
    float fkf_i = (float)fk_i * kAccScale, fkf_q = (float)fk_q * kAccScale;
    if (in0 != in1) {   // the window edge falls between the two byte offsets of this chip offset
      fkf_i += in1 ? -kOutside : kOutside;
      fkf_q += in1 ? -kOutside : kOutside;
    }
    const float faf_i = (float)fa_i * kAccScale, faf_q = (float)fa_q * kAccScale;
    const u32 w1 = sh.chip_t[(exists ? kChips - 1 - q : 0) + 1] >> (4 * h);   // chip 1022 - q of the lane's PRNs
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int pb = (r & 3) + 8 * (r >> 2);
      const float c22 = (float)((f22 >> pb) & 1u), c1 = (float)((w1 >> pb) & 1u);
      acc[0][j][r] += fkf_i + c22 * faf_i - c1 * beta0_i;
      acc[1][j][r] += fkf_q + c22 * faf_q - c1 * beta0_q;
    }
  }
}

// ---- sample offset 8 started directly, with the odd byte offset's terms in the start values and in ONE extra K step per pass -----
// (the byte-phase form, mx_byte_pipe; the formula is the one in front of mx_half_switch at b = 0, without the A_7 that a walk
//  from the even offsets would have left in the accumulators.)
//   extra(q, p) = - pop(W) - chip_p[1022 - q] beta_0 + T(q) [ (2 c1022_p - 1) S_8[q - 1] - 16 c1022_p ]
// because P = data bytes (2 q - 1, 2 q) IS the block D[16 (q - 1) + 8, +16) whose popcount the offset-8 vectors already carry
// as entry q - 1: the tail word acts as one more chip, "chip -1" = chip 1022 in +-1 form.  So
//   * start values:  base - pop(W)   (mx_init_acc_odd);
//   * - chip_p[1022 - q] beta_0 is entry 1022 of the offset-8 vectors' first period lowered by beta_0 = 16 - 2 pop(W) -- and
//     pop(W) is that entry's own block sum: the entry is the constant -16 (mx_byte_wipe_codes): nothing to compute at all;
//   * per pass one MFMA per tile and stream (mx_odd_tail_steps): A column 0 of lane half 0 = -(2 c1022 - 1) / 2 against nibble
//     q - 1 of the pass's own vector (-2 (S & 3), then -(S >> 2) at 2^3), and in the high pass column 0 of lane half 1 = c1022
//     against -2 at 2^3; both B entries zero for q = 0.
template <int NT>
__device__ __forceinline__ void mx_init_acc_odd(const u32 *ones, const u32 *d_i, const u32 *d_q, int lane, int q0_tile,
                                                v16f (&acc)[2][NT], int win_start, int win_stop)
{
  const int n = lane & 31;
  const float base_i = (float)((int)ones[0] + 8192 - kHalf - (int)__popc(d_i[0] & 0xFFu)) * kAccScale;
  const float base_q = (float)((int)ones[1] + 8192 - kHalf - (int)__popc(d_q[0] & 0xFFu)) * kAccScale;
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int q = 32 * (q0_tile + 2 * j) + n;
    const bool in1 = q < kChips && 2 * q + 1 >= win_start && 2 * q + 1 < win_stop;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      acc[0][j][r] = in1 ? base_i : base_i + kOutside;
      acc[1][j][r] = in1 ? base_q : base_q + kOutside;
    }
  }
}
// (both passes' extra steps in one go, BEFORE the passes: their operands come from LDS under the start values' moves)
template <int NT>
__device__ __forceinline__ void mx_odd_tail_operands(const u32 *v_low, const u32 *v_high, int lane, int q0_tile, u32 (&b_low)[2][NT],
                                                     u32 (&b_high)[2][NT])
{
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int q = 32 * (q0_tile + 2 * j) + n;
    const int e = q > 0 ? q - 1 : 0;
    // entry q - 1 of a vector (copy 0, dword e / 8) moved to nibble 0; what is left above it meets zero columns of A
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const u32 lo = v_low[s * 8 * kCopyDwords + (e >> 3)] >> (4 * (e & 7)), hi = v_high[s * 8 * kCopyDwords + (e >> 3)] >> (4 * (e & 7));
      b_low[s][j] = h || q == 0 ? 0u : lo;
      b_high[s][j] = q == 0 ? 0u : h ? 0xCu /* FP4 -2 */ : hi;
    }
  }
}
template <int NT>
__device__ __forceinline__ void mx_odd_tail_steps(const MxShared &sh, int lane, const u32 (&b_low)[2][NT], const u32 (&b_high)[2][NT],
                                                  v16f (&acc)[2][NT])
{
  const int n = lane & 31, h = lane >> 5;
  const u32 c22 = (sh.chip_t[1022 + 1] >> n) & 1u;   // A row n = PRN n of the cluster
  const v4i a_low = v4i{(int)(h ? 0u : c22 ? 0x9u : 0x1u), 0, 0, 0};             // FP4 -0.5 / +0.5
  const v4i a_high = v4i{(int)(h ? c22 << 1 : c22 ? 0x9u : 0x1u), 0, 0, 0};      // lane half 1: 1.0 where chip 1022 is set
#pragma unroll
  for (int j = 0; j < NT; j++)
#pragma unroll
    for (int s = 0; s < 2; s++) {
      acc[s][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a_low), widen(v4i{(int)b_low[s][j], 0, 0, 0}), acc[s][j], 4, 4, 0,
                                                                  kScaleA, 0, kScaleOne);
      acc[s][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a_high), widen(v4i{(int)b_high[s][j], 0, 0, 0}), acc[s][j], 4, 4,
                                                                  0, kScaleA, 0, kScaleEight);
    }
}

// The general form of the above for a workgroup that STARTS at sample offset t0s = 8 half + b (mx_vector_build_direct gave
// it C0): everything the walk would have accumulated or patched in by then, as start values --
//   + c1022 A_b(q),  A_b = 2 pop(byte_o & low_b) - b                                                     (quirk Q5)
//   - half [ pop(W) + chip[1021 - q] b + chip[1022 - q] (16 - b - 2 pop(W))                               (Q3, wrap word)
//            + T(q) ( pop(P) + c1021 (b - 2 pop(P & low_b)) + c1022 (16 - b - 2 pop(P & high_b)) ) ]      (Q3, tail word)
// with o = 2 q + half, W = data bytes (2045, 0), P = data bytes (o - 2, o - 1), T = [q > 0] (the formula in front of
// mx_half_switch; tests/test_formulation.py).  Five multiply-adds per hypothesis, once per workgroup.
__device__ __forceinline__ void mx_direct_terms(const MxShared &sh, int lane, int q0_tile, v16f (&acc)[2][kMxTiles], int t0s,
                                                int win_start, int win_stop)
{
  const int n = lane & 31, h = lane >> 5;
  const int b = t0s & 7, half = t0s >> 3;
  const u32 low = (1u << b) - 1u, high = (0xFFFFu << b) & 0xFFFFu;
  const float fh = (float)half;
  const u32 f22 = sh.chip_t[1022 + 1] >> (4 * h), f21 = sh.chip_t[1021 + 1] >> (4 * h);
  float popw[2], beta[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int pw = (int)__popc(sh.d[s][0] & 0xFFu);   // W = byte 0 << 8: its low byte (data byte 2045) is never mixed
    popw[s] = (float)pw;
    beta[s] = (float)(16 - b - 2 * pw);
  }
#pragma unroll
  for (int j = 0; j < kMxTiles; j++) {
    const int q = 32 * (q0_tile + 2 * j) + n;
    const bool exists = q < kChips;
    const int qc = exists ? q : 0;
    const int o = 2 * qc + half;
    const float tq = (qc > 0) ? fh : 0.0f;            // half * T(q)
    float k0[2], k21[2], k22[2], kb[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int fa = 2 * (int)__popc(lds_byte(sh.d[s], o) & low) - b;
      int fp = 0, gl = 0, gh = 0;
      if (qc > 0 || half == 0) {   // (o >= 2 wherever the tail term counts; for half == 0 the values are multiplied by 0)
        const int o2 = o >= 2 ? o : 2;
        const u32 prev = lds_byte(sh.d[s], o2 - 2) | (lds_byte(sh.d[s], o2 - 1) << 8);
        fp = (int)__popc(prev);
        gl = b - 2 * (int)__popc(prev & low);
        gh = 16 - b - 2 * (int)__popc(prev & high);
      }
      k0[s] = -(fh * popw[s] + tq * (float)fp) * kAccScale;
      k21[s] = -tq * (float)gl * kAccScale;
      k22[s] = ((float)fa - tq * (float)gh) * kAccScale;
      kb[s] = -fh * beta[s] * kAccScale;
    }
    const float ka = -fh * (float)b * kAccScale;
    if (half) {   // the accumulators were started for the even byte offset's window position
      const bool in0 = exists && 2 * q >= win_start && 2 * q < win_stop;
      const bool in1 = exists && 2 * q + 1 >= win_start && 2 * q + 1 < win_stop;
      if (in0 != in1) {
        k0[0] += in1 ? -kOutside : kOutside;
        k0[1] += in1 ? -kOutside : kOutside;
      }
    }
    const u32 w1 = sh.chip_t[(exists ? kChips - 1 - q : 0) + 1] >> (4 * h);   // chip 1022 - q of the lane's PRNs
    const u32 w0 = sh.chip_t[(exists ? kChips - 2 - q : -1) + 1] >> (4 * h);  // chip 1021 - q (chip -1 = 0)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int pb = (r & 3) + 8 * (r >> 2);
      const float c22 = (float)((f22 >> pb) & 1u), c21 = (float)((f21 >> pb) & 1u);
      const float ca = (float)((w0 >> pb) & 1u), cb = (float)((w1 >> pb) & 1u);
      acc[0][j][r] += k0[0] + ca * ka + cb * kb[0] + c21 * k21[0] + c22 * k22[0];
      acc[1][j][r] += k0[1] + ca * ka + cb * kb[1] + c21 * k21[1] + c22 * k22[1];
    }
  }
}

// MULTI: the running sums between blocks: the scratch stream is what binds this form (bench.py `roofline` of the
// multi-block run), so its records are as small as exactness allows.
//   S16 = true  (kMxWalk16, the form that runs first): four 16-bit sums in two dwords, moved on by SATURATING packed adds
//               (v_pk_add_u16 clamp: two sums per instruction, no unpacking).  n_ms x 11573 fits 16 bits up to 5 blocks;
//               beyond that only hypotheses with magnitudes above 6553 in every block -- clean carriers, not signals in
//               noise -- can reach 0xFFFF, where a sum then stays: the last block's epilogue, which unpacks the sums for
//               its keys, raises the workgroup's flag when it meets one, and the launcher's second kernel (kMxWalk,
//               below), whose workgroups leave at once where no flag is up, does such a cluster again with
//   S16 = false (kMxWalk): four sums of 24 bits (128 x 11573 < 2^21) in three dwords.
template <bool S16>
struct SumRecT {
  u32 w[S16 ? 2 : 3];
};
// The records are a stream: written once per block, read once a block later, a megabyte per workgroup in between -- nothing
// a cache keeps.  GPSX_MX_NT: non-temporal loads / stores for them (A/B: tools/build_variant.sh).
template <bool S16>
__device__ __forceinline__ SumRecT<S16> rec_load(const SumRecT<S16> *p)
{
#ifdef WALK_ABL_NO_LOAD   // (timing ablation: results are wrong)
  SumRecT<S16> z{};
  asm volatile("" : "+v"(z.w[0]), "+v"(z.w[1]));
  return z;
#endif
#ifdef GPSX_MX_NT
  SumRecT<S16> r;
#pragma unroll
  for (int i = 0; i < (S16 ? 2 : 3); i++)
    r.w[i] = __builtin_nontemporal_load(&p->w[i]);
  return r;
#else
  return *p;
#endif
}
template <bool S16>
__device__ __forceinline__ void rec_store(SumRecT<S16> *p, const SumRecT<S16> &r)
{
#ifdef WALK_ABL_NO_STORE   // (timing ablation: results are wrong)
  u32 a = r.w[0], b = r.w[1];
  asm volatile("" : : "v"(a), "v"(b));
  return;
#endif
#ifdef GPSX_MX_NT
#pragma unroll
  for (int i = 0; i < (S16 ? 2 : 3); i++)
    __builtin_nontemporal_store(r.w[i], &p->w[i]);
#else
  *p = r;
#endif
}
template <bool S16>
__device__ __forceinline__ void sums_unpack(const SumRecT<S16> &r, u32 (&s)[4])
{
  if constexpr (S16) {
    s[0] = r.w[0] & 0xFFFFu;
    s[1] = r.w[0] >> 16;
    s[2] = r.w[1] & 0xFFFFu;
    s[3] = r.w[1] >> 16;
  } else {
    s[0] = r.w[0] & 0xFFFFFFu;
    s[1] = __builtin_amdgcn_alignbit(r.w[1], r.w[0], 24u) & 0xFFFFFFu;
    s[2] = __builtin_amdgcn_alignbit(r.w[2], r.w[1], 16u) & 0xFFFFFFu;
    s[3] = r.w[2] >> 8;
  }
}
template <bool S16>
__device__ __forceinline__ SumRecT<S16> sums_pack(const u32 (&s)[4])
{
  SumRecT<S16> r;
  if constexpr (S16) {
    // (the low halves of two sums; a sum that does not fit has raised the workgroup's flag, what is stored is not used)
    r.w[0] = __builtin_amdgcn_perm(s[1], s[0], 0x05040100u);
    r.w[1] = __builtin_amdgcn_perm(s[3], s[2], 0x05040100u);
  } else {
    r.w[0] = s[0] | (s[1] << 24);
    r.w[1] = (s[1] >> 8) | (s[2] << 16);
    r.w[2] = (s[2] >> 16) | (s[3] << 8);
  }
  return r;
}

// two 16-bit sums + two magnitudes in one instruction, saturating: a half that reaches 0xFFFF stays there, and the last
// block's epilogue, which unpacks the sums anyway, raises the flag when it meets one
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pk_add_sat_u16(u32 a, u32 b)
{
  return __builtin_bit_cast(u32, __builtin_elementwise_add_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}

// MULTI: request the running sums of sample offset t0, records [first, first + count) of this lane's 16; zero for the first
// block.  The first two tiles' records are requested before the wave's MFMA pass, the other two's at the start of the
// epilogue: a pass or two tiles of epilogue ahead of their use, never more than 12 records in registers.
// Record layout of a wave's slice: [sample offset][tile][lane][8-PRN group] -- a lane's four records of a tile are contiguous
// (32 bytes with the 16-bit records), so two of them move per instruction: half the loads and stores of the
// [offset][tile][group][lane] layout of round 2, still 1 KB contiguous per wave instruction.
template <bool S16>
struct alignas(8) RecPairT {
  SumRecT<S16> a, b;
};
template <int FIRST, int COUNT, bool S16>
__device__ __forceinline__ void mx_prefetch_sums(const u32 *__restrict__ energy, const u32 *__restrict__ zero_recs, int lane,
                                                 int t0, bool ms_first, SumRecT<S16> (&pre)[16])
{
  static_assert(FIRST % 2 == 0 && COUNT % 2 == 0, "records move in pairs");
  // (first block of a search: the running sums are zero -- read from a small all-zero region, `zero_recs`, instead of being
  //  set by the vector ALU: register writes into the prefetch array made the compiler drain every outstanding store first,
  //  an s_waitcnt vmcnt(0) per step; a load is ordered behind them by the memory system and costs nothing)
#ifdef WALK_ABL_ALIAS   // (timing ablation, results are wrong: sample offsets t0 and t0 & WALK_ABL_ALIAS share their records -- the same
  t0 &= WALK_ABL_ALIAS;   //  traffic on 1/2 (7), 1/4 (3) ... of the footprint: does the stream fit the memory-side cache then?)
#endif
  if constexpr (S16) {
    // 16-bit records: [sample offset][tile][group pair][lane][2] -- a wave's 16-byte loads and stores of a pair cover ONE
    // contiguous kilobyte (whole 128-byte lines per instruction, not half lines twice)
    const SumRecT<S16> *e2 = ms_first ? reinterpret_cast<const SumRecT<S16> *>(zero_recs) + (size_t)lane * 2
                                      : reinterpret_cast<const SumRecT<S16> *>(energy) + (size_t)(t0 * kMxTiles) * 256 + (size_t)lane * 2;
#pragma unroll
    for (int i = FIRST; i < FIRST + COUNT; i += 2) {
      const RecPairT<S16> rp = *reinterpret_cast<const RecPairT<S16> *>(&e2[(size_t)(i >> 2) * 256 + (size_t)((i & 3) >> 1) * 128]);
      pre[i] = rp.a;
      pre[i + 1] = rp.b;
    }
    return;
  }
  const SumRecT<S16> *e4 = ms_first ? reinterpret_cast<const SumRecT<S16> *>(zero_recs) + (size_t)lane * 4
                                    : reinterpret_cast<const SumRecT<S16> *>(energy) + ((size_t)(t0 * kMxTiles) * 64 + lane) * 4;
#pragma unroll
  for (int i = FIRST; i < FIRST + COUNT; i += 2) {
    const RecPairT<S16> rp = *reinterpret_cast<const RecPairT<S16> *>(&e4[(size_t)(i >> 2) * 256 + (i & 3)]);
    pre[i] = rp.a;
    pre[i + 1] = rp.b;
  }
}

// ---- epilogue of one sample offset: magnitude, windowed max / sum -------------------------------------------------------
// SEARCH = false (MULTI, not the last block): only the running sums move on -- no key, no maximum, no window sum
template <bool MULTI, bool SEARCH, bool S16>
__device__ __forceinline__ void mx_epilogue(MxShared &sh, int lane, int q0_tile, int t0, const v16f (&acc)[2][kMxTiles],
                                            u32 group_mask, u32 *__restrict__ energy, const u32 *__restrict__ zero_recs,
                                            SumRecT<S16> (&pre)[MULTI ? 16 : 1], bool ms_first, u32 &witness)
{
  typedef SumRecT<S16> SumRec;
  constexpr bool ms_last = SEARCH;
  const int n = lane & 31, h = lane >> 5;
  const int b = t0 & 7, half = t0 >> 3;
  // Search results: every lane owns a (bit shift, PRN) slot pair of LDS -- atomics without a return value at per-PRN
  // constant offsets from one address, no conflicts.  Single-block searches keep a running maximum / sum per PRN in
  // registers and fold them in once per sample offset (32 atomics); the multi-block form, short of registers next to its
  // prefetched sums, folds every hypothesis in directly (measured: 3 % slower for the single-block form, 4 % faster here).
  constexpr bool DIRECT = MULTI;
  u32 *slot = SEARCH ? &sh.part[b][4 * h][0][n] : nullptr;
  u32 best[SEARCH && !DIRECT ? 16 : 1], total[SEARCH && !DIRECT ? 16 : 1];
#pragma unroll
  for (int r = 0; r < (SEARCH && !DIRECT ? 16 : 1); r++) {
    best[r] = 0;
    total[r] = 0;
  }
  // MULTI: the running sums of a lane's four hypotheses of a group are one 12-byte record ([offset][tile][group][lane]: a
  // wave reads / writes 768 contiguous bytes per instruction); the first records of this offset were requested before
  // the wave's MFMA pass (mx_prefetch_sums) -- the scratch is HBM, the pass hides its latency
  SumRec *e4 = reinterpret_cast<SumRec *>(energy) + ((size_t)(t0 * kMxTiles) * 64 + lane) * 4;   // [tile][lane][group]
  mx_round_toward_zero();
#pragma unroll
  for (int j = 0; j < kMxTiles; j++) {
    const int q = 32 * (q0_tile + 2 * j) + n;
    const u32 key_lo = (u32)(2047 - (2 * q + half));   // 0 .. 2047 (q = 1023 does not exist: its magnitude is 0)
    if constexpr (MULTI) {
      // (deeper -- two tiles ahead with the 16-bit records -- measured 3 % slower: more registers, nothing gained)
      // (two tiles ahead: with the spilled registers gone -- round 3 -- the deeper request is 2 % faster; round 2 measured it
      //  1-3 % slower, next to 46 spilled registers)
      if (j == 0)
        mx_prefetch_sums<8, 8, S16>(energy, zero_recs, lane, t0, ms_first, pre);
    }
    // (all four 8-PRN groups, whether this shard owns them or not: a workgroup that owns only some -- at the ends of a
    //  shard's run, or a ragged PRN list -- does a little unused work here instead of branching around register arrays;
    //  group_mask decides below what is published)
    // GS hypotheses at a time: enough independent chains for a wave that has its SIMD's vector ALU to itself (its partner
    // is in the MFMA pass) to cover the ALU, transcendental and branch latencies, few enough to stay in registers next to
    // the 128 accumulators.
    constexpr int GS = MULTI ? 4 : 8;
    SumRec held{};   // (16-bit records, not the last block: the even group's new record, until the odd group's is there)
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += GS) {
      u32 prev[GS];
#pragma unroll
      for (int i = 0; i < GS; i++)
        prev[i] = 0;
      if (MULTI && !(S16 && !SEARCH)) {
#pragma unroll
        for (int g = 0; g < GS / 4; g++) {
          u32 p4[4];
          sums_unpack<S16>(pre[MULTI ? j * 4 + r0 / 4 + g : 0], p4);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            prev[4 * g + i] = p4[i];
            if constexpr (S16)
              witness = max(witness, p4[i]);   // (last block: a saturated sum, 0xFFFF, shows here)
          }
        }
      }
      SumRec *e_rec = e4 + (size_t)j * 256 + r0 / 4;
      u32 out[GS];
      // Magnitudes of the group's hypotheses.  When all of them (GS x 64 lanes) lie below radius 1024 -- noise hypotheses
      // sit at a few hundred -- e < 2^20 is an exact integer and trunc(v_sqrt_f32(e + 1/2)) is its integer root with no
      // fix-up (the true root of n^2 + r + 1/2 keeps 1 / (4 (n + 1)) away from the integers around it, one ulp below 1024
      // is half of that; gps_mag8 checks every pair of that domain on the device): one wave-uniform test per group
      // instead of GS neighbour tests.
      float ev[GS];
      u32 e_max = 0;
#pragma unroll
      for (int i = 0; i < GS; i++) {
        ev[i] = clip_square_sum(acc[0][j][r0 + i], acc[1][j][r0 + i]);
        e_max = max(e_max, __float_as_uint(ev[i]));   // (on the bit patterns: non-negative floats order like integers)
      }
      const bool small = __builtin_amdgcn_ballot_w64(e_max >= 0x3C800000u /* 2^20 / 2^26 as f32 */) == 0;
      u32 mag[GS];
      constexpr bool PACKED = MULTI && S16 && !SEARCH;   // 16-bit records, not the last block: sums move on in packed form
      if (small) {
#pragma unroll
        for (int i = 0; i < GS; i++)
          mag[i] = PACKED ? root_bits_small(ev[i]) : (u32)(int)(__builtin_amdgcn_sqrtf(ev[i]) * kRootGuard);
      } else {
        float ci[GS], cq[GS];
#pragma unroll
        for (int i = 0; i < GS; i++) {
          ci[i] = acc[0][j][r0 + i];
          cq[i] = acc[1][j][r0 + i];
        }
        mx_roots_exact<GS>(ci, cq, mag);
      }
      if constexpr (PACKED) {
        // (magnitudes in the low halves of mag[] -- the rounding add's 0x4B000000 sits above them --, two per v_perm_b32,
        //  then one saturating packed add per pair: 1 instruction per hypothesis instead of unpack, add, pack and witness)
        const SumRec &pr = pre[j * 4 + r0 / 4];
        SumRec nr;
        nr.w[0] = pk_add_sat_u16(pr.w[0], __builtin_amdgcn_perm(mag[1], mag[0], 0x05040100u));
        nr.w[1] = pk_add_sat_u16(pr.w[1], __builtin_amdgcn_perm(mag[3], mag[2], 0x05040100u));
        // (a lane's records of a tile are contiguous: the even group's waits for the odd one, both leave in one 16-byte store)
        if ((r0 / 4) & 1) {
          RecPairT<S16> both;
          both.a = held;
          both.b = nr;
          // ([offset][tile][pair][lane][2]: see mx_prefetch_sums)
#ifdef WALK_ABL_ALIAS
          const int t0r = t0 & WALK_ABL_ALIAS;
#else
          const int t0r = t0;
#endif
          SumRec *e_pair = reinterpret_cast<SumRec *>(energy) + (size_t)(t0r * kMxTiles + j) * 256 + (size_t)(r0 / 8) * 128 + (size_t)lane * 2;
          *reinterpret_cast<RecPairT<S16> *>(e_pair) = both;
        } else {
          held = nr;
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
#pragma unroll
      for (int i = 0; i < GS; i++) {
        const int r = r0 + i;
        u32 val = mag[i];
        if (MULTI)
          val += prev[i];
        out[i] = val;
        if (SEARCH && DIRECT) {
          const int p_off = ((r & 3) + 8 * (r >> 2)) * 64;   // PRN (r & 3) + 8 (r >> 2) + 4 h: 2 x 32 words per PRN
          atomicMax(slot + p_off, (val << 11) | key_lo);
          atomicAdd(slot + p_off + 32, val);
        } else if (SEARCH) {
          const u32 key = (val << 11) | key_lo;
          best[DIRECT ? 0 : r] = key > best[DIRECT ? 0 : r] ? key : best[DIRECT ? 0 : r];
          total[DIRECT ? 0 : r] += val;
        }
      }
      if (MULTI && !ms_last) {
#pragma unroll
        for (int g = 0; g < GS / 4; g++) {
          const u32 o4[4] = {out[4 * g], out[4 * g + 1], out[4 * g + 2], out[4 * g + 3]};
          rec_store<S16>(&e_rec[g], sums_pack<S16>(o4));
        }
      }
      if (SEARCH && !DIRECT) {   // (pinned in program order: left alone, the compiler sinks all 64 chains to the end and spills)
#pragma unroll
        for (int i = 0; i < GS; i++)
          asm volatile("" : "+v"(best[DIRECT ? 0 : r0 + i]), "+v"(total[DIRECT ? 0 : r0 + i]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  mx_round_to_nearest();
  if (SEARCH && !DIRECT) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int p_off = ((r & 3) + 8 * (r >> 2)) * 64;
      atomicMax(slot + p_off, best[DIRECT ? 0 : r]);
      atomicAdd(slot + p_off + 32, total[DIRECT ? 0 : r]);
    }
  }
}

// The single-block form's epilogue (n_ms == 1: what the headline sweep runs).  A wave that has its SIMD's vector ALU to
// itself issues an instruction every ~6.5 cycles whatever the instruction (tools/microbench/issue_mix.hip), so what counts
// here is their number: per hypothesis 2 clip-squares, 1 add, 5/8 for the group's radius test, the root of the scaled sum,
// 1 fma that (in round-toward-zero mode, root_bits_small) leaves floor(root) as an integer in the low mantissa bits, the
// key, and 1/2 + 1/2 for the running maximum and sum of its PRN (two tiles at a time: v_max3_u32 / v_add3_u32).
//   The f32 pattern of that fma is 0x4B000000 + floor(root): shifted left by 11 the exponent bits fall off the key; the
//   sums carry 0x4B000000 per term, four terms per PRN and sample offset: they start at -4 x 0x4B000000 (mod 2^32).
//   (Tried: taking the small path on trust and checking the best keys afterwards -- one test per 64 hypotheses, a second round
//   on the exact path for the PRN groups that show a radius >= 1024 -- saves the 5/8: 1 % faster on noise, 2.5 % slower
//   on the strong test signal, same-box A/B; not kept.)
constexpr u32 kRootBias = 0x4B000000u;
template <int NT>
__device__ __forceinline__ void mx_epilogue_single(MxShared &sh, int lane, const u32 (&kq)[NT], int t0,
                                                   const v16f (&acc)[2][NT], bool half_only = false, bool half_atomics = false,
                                                   int slots = -1)
{
  // (slots: which eighth of sh.part takes the results -- the bit shift's own, unless the pipelined byte-phase form says otherwise)
  const int n = lane & 31, h = lane >> 5;
  const int b = t0 & 7, half = t0 >> 3;
  u32 *slot = &sh.part[slots < 0 ? b : slots][4 * h][0][n];
  u32 best[16], total[16];
  // key = (magnitude << 11) | (2047 - byte offset), byte offset = 2 q + half: kq = 2047 - 2 q is the lane's own constant
  // (>= 1), the wave-uniform half comes off it here, once per tile
  u32 kqh[NT];
#pragma unroll
  for (int j = 0; j < NT; j++)
    kqh[j] = kq[j] - (u32)half;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    best[r] = 0;
    total[r] = 0u - (u32)NT * kRootBias;   // (one biased term per tile and PRN)
  }
  mx_round_toward_zero();
#pragma unroll
  for (int jp = 0; jp < NT; jp += 2) {
#ifdef GPSX_MX_ABLATIONS
    if (jp >= 2 && half_only)   // (timing ablation 128: half an epilogue)
      break;
#endif
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {
      // eight hypotheses: two tiles x four PRNs
      float ev[8];
      u32 e_max = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        ev[i] = clip_square_sum(acc[0][jp + (i >> 2)][r0 + (i & 3)], acc[1][jp + (i >> 2)][r0 + (i & 3)]);
        e_max = max(e_max, __float_as_uint(ev[i]));
      }
      const bool small = __builtin_amdgcn_ballot_w64(e_max >= 0x3C800000u /* 2^20 / 2^26 as f32 */) == 0;
      u32 bits[8];
      if (__builtin_expect(small, 1)) {
#pragma unroll
        for (int i = 0; i < 8; i++)
          bits[i] = root_bits_small(ev[i]);
      } else {
        float ci[8], cq[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          ci[i] = acc[0][jp + (i >> 2)][r0 + (i & 3)];
          cq[i] = acc[1][jp + (i >> 2)][r0 + (i & 3)];
        }
        mx_roots_exact<8>(ci, cq, bits);
#pragma unroll
        for (int i = 0; i < 8; i++)
          bits[i] += kRootBias;
      }
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int r = r0 + rr;
        const u32 k0 = (bits[rr] << 11) | kqh[jp], k1 = (bits[4 + rr] << 11) | kqh[jp + 1];
        best[r] = max(max(best[r], k0), k1);
        total[r] = total[r] + bits[rr] + bits[4 + rr];
      }
      // (pinned in program order: left alone, the compiler sinks all 64 chains to the end and spills)
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
        asm volatile("" : "+v"(best[r0 + rr]), "+v"(total[r0 + rr]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  mx_round_to_nearest();
#ifdef GPSX_MX_ABLATIONS
  if (half_atomics) {   // (timing ablation 512: half the LDS atomics)
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int p_off = ((r & 3) + 8 * (r >> 2)) * 64;
      atomicMax(slot + p_off, best[r] | best[r + 8]);
      atomicAdd(slot + p_off + 32, total[r] + total[r + 8]);
    }
    return;
  }
#endif
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int p_off = ((r & 3) + 8 * (r >> 2)) * 64;   // PRN (r & 3) + 8 (r >> 2) + 4 h: 2 x 32 words per PRN
    atomicMax(slot + p_off, best[r]);
    atomicAdd(slot + p_off + 32, total[r]);
  }
}

// Block-parallel form of a multi-block search (kMxStore): this workgroup handled ONE block; its magnitudes go out as
// u16 in the layout k_acq_vals_search (k_acq_poly.hip) sums and searches: per (search, block, PRN, Doppler) a plane of
// [sample offset][q & 3][q >> 2].
__device__ __forceinline__ void mx_epilogue_store(int lane, int q0_tile, int t0, const v16f (&acc)[2][kMxTiles], u32 group_mask,
                                                  uint16_t *__restrict__ plane0, size_t prn_stride, int slot0, int n_prn)
{
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < kMxTiles; j++) {
    const int q = 32 * (q0_tile + 2 * j) + n;
    uint16_t *v = plane0 + (size_t)t0 * 1024 + (size_t)(q & 3) * 256 + (size_t)(q >> 2);
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int r = 4 * g + rr;
        const int p = (r & 3) + 8 * (r >> 2) + 4 * h;
        const u32 val = mag8_f32(acc[0][j][r], acc[1][j][r]);
        if (((group_mask >> g) & 1u) && slot0 + p < n_prn)
          v[(size_t)p * prn_stride] = (uint16_t)val;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}



// max of a 64-bit value over the eight adjacent lanes of a PRN's bit shifts (quad permutes, then the mirrored half)
__device__ __forceinline__ unsigned long long mx_max8_u64(unsigned long long v)
{
  u32 lo = (u32)v, hi = (u32)(v >> 32);
#define MX_MAX8(ctrl)                                                                           \
  {                                                                                             \
    const u32 lo2 = (u32)__builtin_amdgcn_mov_dpp((int)lo, ctrl, 0xF, 0xF, true);               \
    const u32 hi2 = (u32)__builtin_amdgcn_mov_dpp((int)hi, ctrl, 0xF, 0xF, true);               \
    const bool g = hi2 > hi || (hi2 == hi && lo2 > lo);                                         \
    lo = g ? lo2 : lo;                                                                          \
    hi = g ? hi2 : hi;                                                                          \
  }
  MX_MAX8(0xB1)    // quad_perm [1, 0, 3, 2]
  MX_MAX8(0x4E)    // quad_perm [2, 3, 0, 1]
  MX_MAX8(0x141)   // row_half_mirror
#undef MX_MAX8
  return ((unsigned long long)hi << 32) | lo;
}

constexpr int kMxSingle = 0, kMxWalk = 1, kMxStore = 2, kMxWalk16 = 3, kMxByte = 4, kMxSplit = 5;   // k_acq_mx's MODE

}  // namespace

// mx_a [set][16][2][32][4]: the A fragments of a 32-slot cluster; mx_t [set][1032]: its transposed chip words
__global__ void k_build_mx_tables(const u32 *__restrict__ chipbits, int n_slots, u32 *__restrict__ mx_a, u32 *__restrict__ mx_t)
{
  const int n_sets = (n_slots + 31) / 32;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int per_set = 16 * 2 * 32 * 4 + 1032;
  if (idx >= n_sets * per_set)
    return;
  const int set = idx / per_set, r = idx - set * per_set;
  if (r < 16 * 2 * 32 * 4) {
    const int dw = r & 3, p = (r >> 2) & 31, h = (r >> 7) & 1, kappa = r >> 8;
    const int slot = 32 * set + p;
    const u32 word = slot < n_slots ? chipbits[(size_t)slot * 32 + 2 * kappa + h] : 0u;   // chips 64 kappa + 32 h ..
    mx_a[(size_t)set * (16 * 2 * 32 * 4) + r] = spread8((word >> (8 * dw)) & 0xFFu) << 1;   // chip 1 -> FP4 code 2 (= 1.0)
  } else {
    const int c = r - 16 * 2 * 32 * 4 - 1;   // -1 .. 1030
    u32 w = 0;
    if (c >= 0 && c < kChips)
      for (int p = 0; p < 32; p++) {
        const int slot = 32 * set + p;
        if (slot < n_slots)
          w |= ((chipbits[(size_t)slot * 32 + (c >> 5)] >> (c & 31)) & 1u) << p;
      }
    mx_t[(size_t)set * 1032 + (c + 1)] = w;
  }
}

void launch_build_mx_tables(hipStream_t s, const uint32_t *d_chipbits, int n_slots, uint32_t *d_mx_a, uint32_t *d_mx_t)
{
  const int n_sets = (n_slots + 31) / 32;
  const int n = n_sets * (16 * 2 * 32 * 4 + 1032);
  hipLaunchKernelGGL(k_build_mx_tables, dim3((n + 255) / 256), dim3(256), 0, s, d_chipbits, n_slots, d_mx_a, d_mx_t);
}

// MODE: kMxSingle (n_ms == 1), kMxWalk16 (the workgroup walks the blocks of its searches, running sums as 16-bit records in
// HBM scratch; `flags`[workgroup] tells whether a sum outgrew them), kMxWalk (the same with 24-bit records: launched behind
// kMxWalk16, a workgroup does its cluster again if its flag is up and leaves otherwise), kMxStore (a workgroup per block,
// magnitudes out as u16 for k_acq_vals_search: the form for few multi-block searches)
// One workgroup's work on one launch index `wg` (= blockIdx.x, except in the persistent byte-phase form, which walks them).
// `tables_set`: the PRN set whose tables are in LDS (-1: none); returns through it the set this call left there.
template <int MODE>
__device__ __forceinline__ void mx_unit(MxShared &sh, const AcqParams &prm, int wg, int &tables_set, int cluster_lo,
                                        const uint8_t *__restrict__ if_blocks, const u32 *__restrict__ mx_a,
                                        const u32 *__restrict__ mx_t, gpsx_peak_t *__restrict__ peaks, u32 *__restrict__ energy,
                                        u32 *__restrict__ flags)
{
  constexpr bool MULTI = MODE == kMxWalk || MODE == kMxWalk16, STORE = MODE == kMxStore, S16 = MODE == kMxWalk16;
  // SPLIT: the single-block fine grid for launches that leave most of the chip idle -- two workgroups per cluster, sample offsets
  // 0..7 and 8..15, the second one started directly at offset 8 (as the byte-phase form does); their search results meet in
  // two global u32 planes (`energy` = packed keys, behind them the sums: atomicMax / atomicAdd) that k_acq_finalize converts
  constexpr bool SPLIT = MODE == kMxSplit;
  typedef SumRecT<S16> SumRec;
  if constexpr (MODE == kMxWalk) {
    if (flags && flags[wg] == 0)   // (uniform: the 16-bit run of this cluster was exact)
      return;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform values in SGPRs)
#ifdef GPSX_MX_ABLATIONS
  const int ex = prm.experiment;
#else
  constexpr int ex = 0;            // (production builds carry none of it: its hoisted constants cost the walk form 16 registers)
#endif
                                   // timing ablations: 1 = no epilogue, 2 = no MFMA (noise-sized counts instead), 4 = both roles in
                                   // step, 8 = no vector building, 16 = raised priority for the MFMA passes, 32 = no MFMA pass in
                                   // role 1, 64 = no epilogue in role 0, 128 = half an epilogue, 256 = no steps, 512 = half the atomics
  const int role = (ex & 4) ? 0 : wave >> 2;             // waves w and w + 4 share a SIMD: half a step apart
  const int q0_tile = 8 * (wave >> 1) + (wave & 1);      // this wave owns q-tiles q0_tile + 2 j

  // ---- decode: cluster = (search, Doppler, set of 32 PRN slots); its four 8-PRN groups are sharding units -----------
  const int n_sets = (prm.n_groups + 3) / 4;
  const int ms_store = STORE ? wg % prm.n_ms : 0;
  const int n_seg = SPLIT ? prm.split_segs : 1;          // SPLIT: workgroups per cluster (2, 4 or 8) ...
  const int seg = SPLIT ? wg % n_seg : 0;   // ... and which run of 16 / n_seg sample offsets this one has
  const int cluster = cluster_lo + (STORE ? wg / prm.n_ms : SPLIT ? wg / n_seg : wg);
  const int set = cluster % n_sets;
  const int sd = cluster / n_sets;
  const int dopp = sd % prm.n_dopp;
  const int search = sd / prm.n_dopp;
  u32 group_mask = 0;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int group = 4 * set + g;
    const int unit = sd * prm.n_groups + group;
    if (group < prm.n_groups && unit >= prm.unit_lo && unit < prm.unit_hi)
      group_mask |= 1u << g;
  }
  group_mask = (u32)__builtin_amdgcn_readfirstlane((int)group_mask);
  if (group_mask == 0)
    return;
  const float freq_hz = (float)(prm.if_hz + prm.dopp_min_hz + dopp * prm.dopp_step_hz);   // PM/GPS/acquisition.c:285-289
  const u32 step_word = nco_step_per_word(freq_hz);

  // tables of the cluster (the persistent form keeps them while the set stays the same)
  if (tables_set != set) {
    const u32 *src_a = mx_a + (size_t)set * (16 * 2 * 32 * 4);
    u32 *dst_a = reinterpret_cast<u32 *>(&sh.chips_a[0][0][0]);
    for (int i = tid; i < 16 * 2 * 32 * 4; i += kMxThreads)
      dst_a[i] = src_a[i];
    const u32 *src_t = mx_t + (size_t)set * 1032;
    for (int i = tid; i < 1032; i += kMxThreads)
      sh.chip_t[i] = src_t[i];
    mx_fill_tables(sh, tid);
    tables_set = set;
  }
  for (int i = tid; i < 8 * 32 * 2 * 32 / 4; i += kMxThreads)
    reinterpret_cast<uint4 *>(&sh.part[0][0][0][0])[i] = make_uint4(0, 0, 0, 0);
  const size_t block_bytes = prm.if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : kBytes;
  const uint8_t *block0 = if_blocks + (size_t)(search * prm.search_stride_blocks + (STORE ? wg % prm.n_ms : 0)) * block_bytes;
  mx_load_block(sh, block0, prm.if_format, tid);

  __syncthreads();
  // A operand of the extra K step: column 0 of lane half 0 = chip 1022 of PRN (lane & 31), of half 1 = chip 1021
  const v4i a_corr = v4i{(int)(((sh.chip_t[(lane >> 5 ? 1021 : 1022) + 1] >> (lane & 31)) & 1u) << 1), 0, 0, 0};
  u32 kq[kMxTiles];   // 2047 - (even byte offset of the lane's chip offset in tile j): the low field of its search keys
#pragma unroll
  for (int j = 0; j < kMxTiles; j++) {
    kq[j] = (u32)(2047 - 2 * (32 * (q0_tile + 2 * j) + (lane & 31)));
    if constexpr (!MULTI && !STORE)
      asm volatile("" : "+v"(kq[j]));   // (kept in registers, not rebuilt per group; the other forms do not use them)
  }
  u32 *e_wave = MULTI ? energy + ((size_t)wg * 8 + wave) * (16 * kMxTiles * 4 * 64 * (S16 ? 2 : 3)) : nullptr;   // dwords per record
  const u32 *zero_recs = MULTI ? flags - kMxZeroRecBytes / sizeof(u32) : nullptr;   // (the launcher put them in front of the flags)
  u32 witness = 0;   // S16: OR of every sum this lane stored
  const int n_ms = MULTI ? prm.n_ms : 1;
  // SPLIT: two direct passes at sample offset t0s, then 16 / n_seg - 1 steps of the walk (local pass lp >= 2 is pass t0s + lp)
  const int t0s = SPLIT ? seg * (16 / n_seg) : 0;
  const int n_pass = SPLIT ? 16 / n_seg + 1 : kPasses;
  const int pbase = t0s;
#pragma unroll 1
  for (int ms = 0; ms < n_ms; ms++) {
    const bool ms_first = ms == 0, ms_last = ms == n_ms - 1;
    // (the walk forms: thread indices made opaque per block, so that the preamble's per-thread address arithmetic is redone
    //  per block instead of being hoisted out of this loop into registers that the step loop below then has to spill)
    int tid_p = tid, lane_p = lane;
    if constexpr (MULTI)
      asm volatile("" : "+v"(tid_p), "+v"(lane_p));
    if (ms > 0) {
      __syncthreads();   // the previous block's readers are done
      mx_load_block(sh, block0 + (size_t)ms * block_bytes, prm.if_format, tid_p);
      __syncthreads();
    }
    mx_wipe_block(sh, step_word, tid_p, lane_p);
    if (SPLIT && seg) {
      // not the first run: the first two vectors from the block sums of sample offset t0s (the planes' barrier is the loop's first)
      mx_vector_build_direct(sh, 0, t0s, &sh.e8[0][0][0][0], tid);
      mx_vector_build_direct(sh, 1, t0s, &sh.e8[1][0][0][0], tid);
      for (int i = tid; i < 2 * 2 * 2 * 128; i += kMxThreads)
        (&sh.corr[0][0][0][0])[i] = 0;
    } else {
      __syncthreads();
      // the first two vectors (from the popcounts of sample offset 0), by the two-phase builders
      mx_vector_phase1(sh, 0, 0, tid_p, kMxThreads);
      __syncthreads();
      mx_vector_phase2(sh, 0, tid_p, kMxThreads);
      __syncthreads();
      mx_vector_phase1(sh, 1, 1, tid_p, kMxThreads);
      __syncthreads();
      mx_vector_phase2(sh, 1, tid_p, kMxThreads);
    }

#ifdef GPSX_MX_TIMELINE   // (tools/experiments/single_timeline.py: cycle stamps of workgroup 1000's waves 0 and 4 behind the peaks)
    unsigned long long *tl1 = (MODE == kMxSingle || (MODE == kMxWalk16 && ms == 5)) && blockIdx.x == 1000 && lane == 0 && (wave & 3) == 0
                                  ? reinterpret_cast<unsigned long long *>(peaks + (size_t)gridDim.x * 256) + role * 512 : nullptr;
    int tl1i = 0;
#define MX_TL1() do { if (tl1 && tl1i < 512) tl1[tl1i++] = __builtin_readcyclecounter(); } while (0)
#else
#define MX_TL1() do { } while (0)
#endif
    MX_TL1();
    v16f acc[2][kMxTiles];
    mx_init_acc(sh.ones, lane_p, q0_tile, acc, prm.win_start, prm.win_stop);
    if ((ex & 2) || ((ex & 32) && role)) {   // (timing ablation without MFMAs: noise-sized counts, so that the epilogue takes its usual path)
#pragma unroll
      for (int j = 0; j < kMxTiles; j++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          acc[0][j][r] = acc[1][j][r] = (float)(lane + r) * kAccScale;
    }
    SumRec pre[MULTI ? 16 : 1];

    // Steps of two halves: role 0 runs pass p, then the epilogue of sample offset p - 1; role 1 the epilogue of sample
    // offset p - 2, then pass p -- one wave of a SIMD on the matrix pipe while the other has the vector ALU, with nothing
    // but their own pace between the halves: ONE barrier per step, where all eight waves build the vector of pass p + 1
    // into the buffer that both roles read during step p - 1.
#ifdef GPSX_MX_CYCLES   // (lab: where workgroup 1000's waves spend their cycles -- barrier, vector building, pass, epilogue)
    unsigned long long cy_bar = 0, cy_build = 0, cy_pass = 0, cy_epi = 0;
    const unsigned long long cy_begin = __builtin_readcyclecounter();
#define MX_CY(acc_var) do { const unsigned long long now = __builtin_readcyclecounter(); acc_var += now - cy_last; cy_last = now; } while (0)
#else
#define MX_CY(acc_var) do { } while (0)
#endif
#pragma unroll 1
    for (int hs = (ex & 256) ? 2 * n_pass + 1 : 0; hs <= 2 * n_pass; hs++) {   // (timing ablation 256: no steps at all)
#ifdef GPSX_MX_CYCLES
      unsigned long long cy_last = __builtin_readcyclecounter();
#endif
      MX_TL1();
      if ((hs & 1) == 0)
        __syncthreads();
      MX_CY(cy_bar);
      MX_TL1();
      // The vector of the next step: built behind the barrier by everybody (single-block forms), or behind this step's epilogue
      // by the role that just finished one (walk forms: each role's threads own one stream of the vector -- threads 0..255 =
      // waves 0..3 = I, 256..511 = Q --, the buffer it goes into was last read in the previous step, and the barrier that opens
      // the next step publishes it).  Same-box A/B: behind the epilogue is 1.5 % faster for the walk form (whose epilogue waits
      // on HBM anyway) and 1.8 % slower for the single-block form -- the step is bound by the SIMD's issue port, not by the
      // barrier: moving the work does not shorten it.
#ifdef MX_BUILD_BEHIND
      constexpr bool kBuildBehindEpilogue = true;
#else
      constexpr bool kBuildBehindEpilogue = MULTI;
#endif
      if (!kBuildBehindEpilogue && (hs & 1) == 0) {
        const int p_vec = (hs >> 1) + 1;
        if (p_vec >= 2 && p_vec < n_pass && !(ex & 8)) {
          int tid_v = tid;
          if constexpr (MULTI)
            asm volatile("" : "+v"(tid_v));   // (as above: the builder's addresses are not worth registers across the passes)
          mx_vector_build(sh, pbase + p_vec, tid_v);
        }
      }
      MX_TL1();
      MX_CY(cy_build);
      int lane_s = lane;         // (walk forms: opaque per half step, see tid_p -- record addresses are recomputed, not spilled)
      if constexpr (MULTI)
        asm volatile("" : "+v"(lane_s));
      const int x = hs - role;   // role-local half step: even = MFMA pass x / 2, odd = epilogue after pass (x - 1) / 2
      const bool active = x >= 0 && x < 2 * n_pass;
      const int p = x >> 1;
      if (active && (x & 1) == 0) {
        if constexpr (MULTI) {
          if (p >= 1)
            mx_prefetch_sums<0, 8, S16>(e_wave, zero_recs, lane_s, p - 1, ms_first, pre);
        }
        if (!(ex & 2) && !((ex & 32) && role)) {
          if (ex & 16)
            __builtin_amdgcn_s_setprio(3);
          mx_pass<!MULTI>(sh, p & 1, lane, q0_tile, acc, p == 1 ? kScaleEight : kScaleOne, a_corr, p >= 2 && (SPLIT || p != 9));
          if (ex & 16)
            __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (SPLIT) {
          if (p == 1 && seg) {
            int lane_d = lane;   // (opaque: the terms' per-lane addresses are not worth registers across the step loop)
            asm volatile("" : "+v"(lane_d));
            mx_direct_terms(sh, lane_d, q0_tile, acc, t0s, prm.win_start, prm.win_stop);
          }
        } else {
          if (p == 9)
            mx_half_switch(sh, lane_s, q0_tile, acc, prm.win_start, prm.win_stop);
        }
      }
      MX_TL1();
      MX_CY(cy_pass);
      if (STORE) {
        if (active && (x & 1) && p >= 1) {
          uint16_t *plane0 = reinterpret_cast<uint16_t *>(energy) +
                             ((size_t)((search * prm.n_ms + ms_store) * prm.n_prn + 32 * set) * prm.n_dopp + dopp) * (16 * 1024);
          mx_epilogue_store(lane, q0_tile, p - 1, acc, group_mask, plane0, (size_t)prm.n_dopp * (16 * 1024), 32 * set, prm.n_prn);
        }
      } else if (active && (x & 1) && p >= 1 && !(ex & 1) && !((ex & 64) && !role)) {
        if (!MULTI)
          mx_epilogue_single(sh, lane, kq, t0s + p - 1, acc, (ex & 128) != 0, (ex & 512) != 0);
        else if (!ms_last)
          mx_epilogue<MULTI, false, S16>(sh, lane_s, q0_tile, p - 1, acc, group_mask, e_wave, zero_recs, pre, ms_first, witness);
        else
          mx_epilogue<MULTI, true, S16>(sh, lane_s, q0_tile, p - 1, acc, group_mask, e_wave, zero_recs, pre, ms_first, witness);
      }
      MX_CY(cy_epi);
      if (kBuildBehindEpilogue && (x & 1) != 0) {
        const int p_vec = (hs >> 1) + 1;
        if (p_vec >= 2 && p_vec < n_pass && !(ex & 8)) {
          int tid_v = tid;
          if constexpr (MULTI)
            asm volatile("" : "+v"(tid_v));
          mx_vector_build(sh, pbase + p_vec, tid_v);
        }
      }
    }
#ifdef GPSX_MX_CYCLES
    if (blockIdx.x == 1000 && lane == 0 && ms == 0)
      printf("mode %d wave %d: barrier %llu build %llu pass %llu epilogue %llu, loop %llu cycles\n", MODE, wave, cy_bar, cy_build, cy_pass,
             cy_epi, __builtin_readcyclecounter() - cy_begin);
#endif
  }
  if (STORE)
    return;   // k_acq_vals_search sums the blocks and searches
  if constexpr (S16) {
    // did any stored sum need more than 16 bits?  (sh.ones is free after the last block's preamble)
    if (tid == 0)
      sh.ones[0] = 0;
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(witness >= 0xFFFFu) != 0 && lane == 0)
      atomicOr(&sh.ones[0], 1u);
  }
  __syncthreads();
  if constexpr (S16) {
    if (tid == 0)
      flags[wg] = sh.ones[0];
    if (sh.ones[0])
      return;   // (uniform) the second kernel does this cluster again and writes its triplets
  }
  // the finished triplets: one per (PRN, bit shift); threads 0..255 fold the 32 lane slots of the maxima and write
  // (max, phase), threads 256..511 those of the sums and write (sum, avr)
  {
    const int which = tid >> 8, p = (tid >> 3) & 31, b = tid & 7;
    const int slot = 32 * set + p;
    const u32 *row = sh.part[b][p][which];
    u32 vals[32];
#pragma unroll
    for (int l = 0; l < 32; l++)
      vals[l] = row[(l + tid) & 31];   // (rotated start: the threads of a wave spread over the banks)
    u32 k = 0, t = 0;
#pragma unroll
    for (int l = 0; l < 32; l++) {
      k = vals[l] > k ? vals[l] : k;
      t += vals[l];
    }
    if (((group_mask >> (p >> 3)) & 1u) && slot < prm.n_prn && b < prm.n_bits) {
      const size_t idx = ((size_t)(search * prm.n_prn + slot) * prm.n_dopp + dopp) * prm.n_bits + b;
      uint2 *pk = reinterpret_cast<uint2 *>(&peaks[idx]);
      if constexpr (SPLIT) {   // the runs of a cluster meet in the planes: keys [0, n), sums [n, 2 n)
        const size_t n_planes = (size_t)prm.n_planes;
        if (which == 0)
          atomicMax(&energy[idx], k);
        else
          atomicAdd(&energy[n_planes + idx], t);
      } else if (which == 0) {
        const u32 max_val = k >> 11;
        pk[0] = make_uint2(max_val, max_val ? 2047u - (k & 2047u) : 0u);   // gpsx_peak_t: max_val, phase
      } else {
        pk[1] = make_uint2(t, t / (2u * kChips));                          //              sum, avr
      }
    }
    // the packed key of (search, PRN, Doppler) -- (energy << 14) | (16383 - fine phase) of the best bit shift, what k_acq_keys
    // makes of the triplets -- while they are in registers (unsharded launches: prm.keys is null otherwise)
    if constexpr (!SPLIT) {
      if (prm.keys && which == 0) {   // (wave-uniform: waves 0..3; a PRN's eight bit shifts are eight adjacent lanes)
        unsigned long long key = 0;
        if (b < prm.n_bits) {
          const u32 max_val = k >> 11, phase = max_val ? 2047u - (k & 2047u) : 0u;
          key = ((unsigned long long)max_val << 14) | (unsigned long long)(16383u - (8u * phase + (u32)b));
        }
        key = mx_max8_u64(key);
        if (b == 0 && ((group_mask >> (p >> 3)) & 1u) && slot < prm.n_prn)
          prm.keys[(size_t)(search * prm.n_prn + slot) * prm.n_dopp + dopp] = (int64_t)key;
      }
    }
  }
}

// ---- the byte-phase grid as ONE software pipeline over the clusters of a persistent workgroup (k_acq_mx<4>) -----------------
// Per cluster and wave two stages -- sample offset 0, sample offset 8, each started from its own block sums: start values, two
// passes on the wave's four q-tiles, an epilogue of 64 hypotheses per lane -- the two waves of a SIMD half a stage apart, one
// barrier per stage.  The clusters follow each other WITHOUT a fill and a drain half stage and without a preamble between them:
// what cluster c + 1 and c + 2 need is made by all eight waves behind the barriers of cluster c's stages, each piece in a buffer
// nobody reads then (mx_byte_pipe).  Two copies of what a stage reads of its block (d), of the sums' bytes and of the result
// slots (sh.part[0] / [6]) by the cluster's parity; three of pop(D).
struct MxBlockRegs {
  u32 v[4];
};
__device__ __forceinline__ MxBlockRegs mx_block_request(const uint8_t *blk, int if_format, int tid)
{
  // thread t: 16-bit words 2 t and 2 t + 1 of the sign plane (1023 exist), as load_sign16 reads them
  MxBlockRegs r;
  const uint16_t *p = reinterpret_cast<const uint16_t *>(blk);
  const bool second = 2 * tid + 1 < kWords16;
  if (if_format == GPSX_IF_2BIT_SM) {
    r.v[0] = p[4 * tid];
    r.v[1] = p[4 * tid + 1];
    r.v[2] = second ? p[4 * tid + 2] : 0;
    r.v[3] = second ? p[4 * tid + 3] : 0;
  } else {
    r.v[0] = p[2 * tid];
    r.v[1] = second ? p[2 * tid + 1] : 0;
    r.v[2] = r.v[3] = 0;
  }
  return r;
}
__device__ __forceinline__ void mx_block_commit(MxShared &sh, const MxBlockRegs &r, int if_format, int tid)
{
  u32 lo = r.v[0], hi = r.v[1];
  if (if_format == GPSX_IF_2BIT_SM) {
    lo = even_bits16(r.v[0] | (r.v[1] << 16));
    hi = even_bits16(r.v[2] | (r.v[3] << 16));
  }
  reinterpret_cast<u32 *>(sh.x)[tid] = (lo & 0xFFFFu) | (hi << 16);
}

// FP4 codes of the two parts of a block sum S = 0 .. 16: -2 (S & 3) -> 0, C, E, F; -(S >> 2) -> 0, A, C, D, E
__device__ __forceinline__ u32 sum_code_low(u32 s) { return (0xFEC0u >> ((s << 2) & 0xCu)) & 0xFu; }
__device__ __forceinline__ u32 sum_code_high(u32 s) { return (0xEDCA0u >> (s & 0x1Cu)) & 0xFu; }

// Table for the wipe-off piece (in the recurrence's lookup tables' LDS, which this form does not use): two block sums
// (a | b << 5, each 0 .. 16) -> the byte of their low codes and, above it, the byte of their high codes
__device__ __forceinline__ void mx_byte_fill_code_table(MxShared &sh, int tid)
{
  uint16_t *lut = reinterpret_cast<uint16_t *>(sh.t_lut);
  static_assert(sizeof(sh.t_lut) >= 1024 * sizeof(uint16_t), "code table fits");
  for (int i = tid; i < 1024; i += kMxThreads) {
    const u32 sa = (u32)i & 31u, sb = (u32)i >> 5;
    lut[i] = (uint16_t)(sum_code_low(sa) | (sum_code_low(sb) << 4) | (sum_code_high(sa) << 8) | (sum_code_high(sb) << 12));
  }
}

// Wipe-off of the block in sh.x -> d[2][514] (word 511 = the wrap-around copy), pop(D) -> ones (zeroed beforehand), and copy 0
// of the four vectors of each stream -- entry k = the FP4 code of a part of the block sum S_t0[k mod 1023], k < 2056 -- as bytes
// of two entries: thread w has word w and wipes word w + 1 a second time (no barrier between the stream and its sums), i.e.
// sums 2 w, 2 w + 1, 2 w + 2 of either sample offset: byte w of the first period, byte 512 + w of the second (which starts at
// the odd entry 1023), and bytes 0..4 again as 1023..1027 (entries from 2046).  Entry 1023 = entry 0.  The four bytes of a
// thread (low / high vector, first / second period) are transposed over its quad, so that each lane writes ONE dword.
__device__ __forceinline__ void mx_byte_wipe_codes(const MxShared &sh, u32 *d, u32 *ones, u32 *base0, u32 *base8, u32 step_word,
                                                   int tid, int lane)
{
  const u32 *x32 = reinterpret_cast<const u32 *>(sh.x);
  const uint16_t *lut = reinterpret_cast<const uint16_t *>(sh.t_lut);
  const int w = tid, k = tid & 3;
  const u32 x_first = x32[0], x_cur = x32[w], x_next = x32[w < 511 ? w + 1 : 0];
  const u32 quad_cur = (step_word * (u32)w) >> 30, quad_next = (step_word * (u32)(w + 1)) >> 30;
  const u32 sel = (u32)k * 0x0101u + 0x0400u;   // v_perm_b32: byte k of the second source, byte k of the first
  // lane k of a quad writes item k: low / high vector (k & 1), first / second period (k >> 1), dword w / 4 of it
  const int item_dword = (k & 1) * 258 + (k >> 1) * 128 + (w >> 2);
  u32 cnt = 0;   // both streams' counts in one register (each below 2^16 per wave)
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const u32 first = (s ? carrier_q(0u) : carrier_i(0u)) ^ x_first;
    const u32 wrap = first << 16;   // samples 16352..16367 are zero, then sample 0 again
    const u32 cur = w < kWords32 ? (s ? carrier_q(quad_cur) : carrier_i(quad_cur)) ^ x_cur : wrap;
    const u32 nxt = w + 1 < kWords32 ? (s ? carrier_q(quad_next) : carrier_i(quad_next)) ^ x_next : (w + 1 == kWords32 ? wrap : 0u);
    d[s * 514 + w] = cur;
    cnt += (w < kWords32 ? (u32)__popc(cur) : 0u) << (16 * s);
    const u32 x8 = __builtin_amdgcn_alignbit(nxt, cur, 8u);
#pragma unroll
    for (int o = 0; o < 2; o++) {
      u32 s0 = pop16(o ? x8 : cur), s1 = (u32)__popc((o ? x8 : cur) >> 16);
      const u32 s2 = pop16(o ? nxt >> 8 : nxt);
      if (o && w == 511) {
        s1 = pop16(first >> 8);   // entry 1023 = entry 0 (offset 0: the wrap word's upper half already is D[0, 16))
        // Offset 8, entry 1022 of the FIRST period (the only one chip 1022 - q ever meets): the odd byte offsets skip the
        // replica word at the wrap (quirk Q3) -- - chip[1022 - q] beta_0 with beta_0 = 16 - 2 pop(W), and pop(W) IS this
        // entry's block sum S_8[1022] = pop(D[0, 8)): entry -2 S - beta_0 = -16 whatever the data, i.e. "S = 8".
        s0 = 8;
      }
      // bytes: [0] low vector, first period; [1] high, first; [2] low, second period; [3] high, second
      const u32 pk = (u32)lut[s0 | (s1 << 5)] | ((u32)lut[s1 | (s2 << 5)] << 16);
      u32 *base = (o ? base8 : base0) + s * (2 * 258);   // [stream][low / high][258 dwords]
      const u32 p0 = (u32)__builtin_amdgcn_mov_dpp((int)pk, 0x00, 0xF, 0xF, true), p1 = (u32)__builtin_amdgcn_mov_dpp((int)pk, 0x55, 0xF, 0xF, true);
      const u32 p2 = (u32)__builtin_amdgcn_mov_dpp((int)pk, 0xAA, 0xF, 0xF, true), p3 = (u32)__builtin_amdgcn_mov_dpp((int)pk, 0xFF, 0xF, 0xF, true);
      const u32 out = (__builtin_amdgcn_perm(p1, p0, sel) & 0xFFFFu) | (__builtin_amdgcn_perm(p3, p2, sel) << 16);
      if (w < 508 || k < 2)
        base[item_dword] = out;
      uint8_t *bytes = reinterpret_cast<uint8_t *>(base);
      if (w >= 508 && w < 511) {   // the last dword of the second period also holds byte 1023, which is entry 2046's
        bytes[512 + w] = (uint8_t)(pk >> 16);
        bytes[258 * 4 + 512 + w] = (uint8_t)(pk >> 24);
      }
      if (w < 5) {
        bytes[1023 + w] = (uint8_t)pk;
        bytes[258 * 4 + 1023 + w] = (uint8_t)(pk >> 8);
      }
    }
  }
  cnt = wave_sum_to_lane63(cnt);
  if (lane == 63) {
    atomicAdd(&ones[0], cnt & 0xFFFFu);
    atomicAdd(&ones[1], cnt >> 16);
  }
}

// the low and the high vector of one sample offset: the eight shifted copies of each from its copy 0
__device__ __forceinline__ void mx_byte_vector_pair(const u32 *base, u32 *dst_low, u32 *dst_high, int tid)
{
  const int iq = tid >> 8, j = tid & 255;
#pragma unroll
  for (int which = 0; which < 2; which++) {
    const u32 *v = base + (iq * 2 + which) * 258 + j;
    const u32 lo = v[0], hi = v[1];
    u32 *dst = (which ? dst_high : dst_low) + (iq * 8) * kCopyDwords + j;
#pragma unroll
    for (int c = 0; c < 8; c++)
      dst[c * kCopyDwords] = c ? __builtin_amdgcn_alignbit(hi, lo, 4u * (u32)c) : lo;
  }
}

// the triplets of one cluster from result slots `slots` of sh.part (bit shift 0 only), and the slots back to zero: thread
// (which, PRN, eighth) folds four lane slots, the eight threads of a PRN meet over DPP / permutes
__device__ __forceinline__ void mx_byte_fold(MxShared &sh, int slots, u32 group_mask, int set, int search, int dopp,
                                             const AcqParams &prm, gpsx_peak_t *__restrict__ peaks, int tid)
{
  const int which = tid >> 8, p = (tid >> 3) & 31, part = tid & 7;
  uint4 *row = reinterpret_cast<uint4 *>(&sh.part[slots][p][which][4 * part]);
  const uint4 v = *row;
  *row = make_uint4(0, 0, 0, 0);
  u32 k = max(max(v.x, v.y), max(v.z, v.w)), t = v.x + v.y + v.z + v.w;
  // the eight lanes of a PRN: neighbours, pairs (quad permutes), then the other half of the eight (mirrored: all four alike by then)
#define MX_FOLD8(ctrl)                                                                          \
  {                                                                                             \
    const u32 ko = (u32)__builtin_amdgcn_mov_dpp((int)k, ctrl, 0xF, 0xF, true);                 \
    const u32 to = (u32)__builtin_amdgcn_mov_dpp((int)t, ctrl, 0xF, 0xF, true);                 \
    k = ko > k ? ko : k;                                                                        \
    t += to;                                                                                    \
  }
  MX_FOLD8(0xB1)    // quad_perm [1, 0, 3, 2]
  MX_FOLD8(0x4E)    // quad_perm [2, 3, 0, 1]
  MX_FOLD8(0x141)   // row_half_mirror
#undef MX_FOLD8
  const int slot = 32 * set + p;
  if (part == 0 && ((group_mask >> (p >> 3)) & 1u) && slot < prm.n_prn) {
    const size_t idx = ((size_t)(search * prm.n_prn + slot) * prm.n_dopp + dopp) * prm.n_bits;
    uint2 *pk = reinterpret_cast<uint2 *>(&peaks[idx]);
    if (which == 0) {
      const u32 max_val = k >> 11, phase = max_val ? 2047u - (k & 2047u) : 0u;
      pk[0] = make_uint2(max_val, phase);                                // gpsx_peak_t: max_val, phase
      if (prm.keys)   // (the packed key k_acq_keys would make of it: one bit shift)
        prm.keys[idx] = (int64_t)(((unsigned long long)max_val << 14) | (unsigned long long)(16383u - 8u * phase));
    } else {
      pk[1] = make_uint2(t, t / (2u * kChips));                          //              sum, avr
    }
  }
}

__device__ __forceinline__ void mx_byte_pipe(MxShared &sh, const AcqParams &prm, int cluster_lo,
                                             const uint8_t *__restrict__ if_blocks, const u32 *__restrict__ mx_a,
                                             const u32 *__restrict__ mx_t, gpsx_peak_t *__restrict__ peaks)
{
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2;                            // waves w and w + 4 share a SIMD: half a stage apart
  const int q0_tile = 8 * (wave >> 1) + (wave & 1);      // this wave owns q-tiles q0_tile + 2 j
  const int n_sets = (prm.n_groups + 3) / 4;
  const int stride = (int)gridDim.x, first = cluster_lo + (int)blockIdx.x;
  const int n_my = (prm.n_clusters - (int)blockIdx.x + stride - 1) / stride;   // clusters first, first + stride, ...: >= 1
  if (n_my <= 0)                                         // (a grid larger than the launch's clusters: the launcher never makes one)
    return;
  const int set = first % n_sets;                        // (the launcher's grid is a multiple of n_sets: one PRN set per workgroup)
  const size_t block_bytes = prm.if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : kBytes;

  // LDS this form has to itself: the polyphase planes (second copy of d, three of ones), the lookup tables (code table, step
  // table), and of the result slots of bit shifts 1..7: the offset-8 vectors, behind them copy 0 of the vectors as the wipe-off
  // piece leaves them ([offset 0 | offset 8 of even / odd clusters][stream][low / high][258 dwords]), slot 7 = odd clusters' results
  u32 *d_alt = &sh.plane[0][0][0], *ones3 = d_alt + 2 * 514;   // ones3[3][2]
  static_assert(sizeof(sh.plane) >= (2 * 514 + 6) * sizeof(u32), "overlays fit");
  u32 *e8x = &sh.part[1][0][0][0];
  constexpr int kVec = 2 * 8 * kCopyDwords, kBase = 2 * 2 * 258, kSlotsEven = 0, kSlotsOdd = 7;
  u32 *bbase = e8x + 2 * kVec;
  static_assert((2 * kVec + 3 * kBase) * sizeof(u32) <= 6 * sizeof(sh.part[0]), "two vectors and three sets of their copy 0 below result slots 7");

  // (search, Doppler bin) of this workgroup's clusters c - 1 .. c + 2 around the cluster c the pieces are at: moved on by
  // additions, one division when the workgroup starts (a cluster's pieces need three decodes; divisions cost them a third)
  const int sd_step = stride / n_sets, search_step = sd_step / prm.n_dopp, dopp_step = sd_step % prm.n_dopp;
  int w_sd[4], w_search[4], w_dopp[4], w_at = 0;   // [k]: cluster w_at - 1 + k
  w_sd[1] = first / n_sets;
  w_search[1] = w_sd[1] / prm.n_dopp;
  w_dopp[1] = w_sd[1] % prm.n_dopp;
  w_sd[0] = w_sd[1], w_search[0] = w_search[1], w_dopp[0] = w_dopp[1];
#pragma unroll
  for (int k = 2; k < 4; k++) {
    w_sd[k] = w_sd[k - 1] + sd_step;
    w_dopp[k] = w_dopp[k - 1] + dopp_step;
    w_search[k] = w_search[k - 1] + search_step + (w_dopp[k] >= prm.n_dopp ? 1 : 0);
    w_dopp[k] -= w_dopp[k] >= prm.n_dopp ? prm.n_dopp : 0;
  }
  auto window_to = [&](int c) {   // (at most one step per call)
    if (w_at < c) {
#pragma unroll
      for (int k = 0; k < 3; k++)
        w_sd[k] = w_sd[k + 1], w_search[k] = w_search[k + 1], w_dopp[k] = w_dopp[k + 1];
      w_sd[3] = w_sd[2] + sd_step;
      w_dopp[3] = w_dopp[2] + dopp_step;
      w_search[3] = w_search[2] + search_step + (w_dopp[3] >= prm.n_dopp ? 1 : 0);
      w_dopp[3] -= w_dopp[3] >= prm.n_dopp ? prm.n_dopp : 0;
      w_at++;
    }
  };
  auto decode = [&](int i, int &search, int &dopp, u32 &mask) {   // i in w_at - 1 .. w_at + 2
    const int k = i - w_at + 1;
    const int sd = k == 0 ? w_sd[0] : k == 1 ? w_sd[1] : k == 2 ? w_sd[2] : w_sd[3];
    search = k == 0 ? w_search[0] : k == 1 ? w_search[1] : k == 2 ? w_search[2] : w_search[3];
    dopp = k == 0 ? w_dopp[0] : k == 1 ? w_dopp[1] : k == 2 ? w_dopp[2] : w_dopp[3];
    mask = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int group = 4 * set + g, unit = sd * prm.n_groups + group;
      if (group < prm.n_groups && unit >= prm.unit_lo && unit < prm.unit_hi)
        mask |= 1u << g;
    }
  };
  auto block_of = [&](int i) {
    int search, dopp;
    u32 mask;
    decode(i, search, dopp, mask);
    return if_blocks + (size_t)(search * prm.search_stride_blocks) * block_bytes;
  };
  // the carrier's step per 32-sample word for the first 256 Doppler bins (one correctly rounded division each: once per
  // workgroup instead of once per cluster, where everything in the wipe-off piece waits for it); behind the code table
  u32 *step_tab = sh.t_lut + 512;
  static_assert(sizeof(sh.t_lut) >= (512 + 256) * sizeof(u32), "step table fits");
  auto step_of_bin = [&](int dopp) {
    return nco_step_per_word((float)(prm.if_hz + prm.dopp_min_hz + dopp * prm.dopp_step_hz));   // PM/GPS/acquisition.c:285-289
  };
  auto step_of = [&](int i) {
    int search, dopp;
    u32 mask;
    decode(i, search, dopp, mask);
    return dopp < 256 ? step_tab[dopp] : step_of_bin(dopp);
  };
  auto d_of = [&](int i) { return i & 1 ? d_alt : &sh.d[0][0]; };
  auto ones_of = [&](int i) { return ones3 + 2 * (i % 3); };
  u32 *base0 = bbase;
  auto base8_of = [&](int i) { return bbase + (1 + (i & 1)) * kBase; };

  // ---- fill: tables of the PRN set, cluster 0 up to its offset-0 vectors, cluster 1's block in LDS ------------------------------
  {
    MxBlockRegs b0 = mx_block_request(block_of(0), prm.if_format, tid);
    MxBlockRegs b1 = n_my > 1 ? mx_block_request(block_of(1), prm.if_format, tid) : b0;
    const u32 *src_a = mx_a + (size_t)set * (16 * 2 * 32 * 4);
    u32 *dst_a = reinterpret_cast<u32 *>(&sh.chips_a[0][0][0]);
    for (int i = tid; i < 16 * 2 * 32 * 4; i += kMxThreads)
      dst_a[i] = src_a[i];
    const u32 *src_t = mx_t + (size_t)set * 1032;
    for (int i = tid; i < 1032; i += kMxThreads)
      sh.chip_t[i] = src_t[i];
    constexpr int kSlotVecs = (int)(sizeof(sh.part[0]) / sizeof(uint4));
    for (int i = tid; i < 2 * kSlotVecs; i += kMxThreads)
      reinterpret_cast<uint4 *>(&sh.part[i / kSlotVecs ? kSlotsOdd : kSlotsEven][0][0][0])[i % kSlotVecs] = make_uint4(0, 0, 0, 0);
    mx_byte_fill_code_table(sh, tid);
    if (tid < 256 && tid < prm.n_dopp)
      step_tab[tid] = step_of_bin(tid);
    if (tid < 6)
      ones3[tid] = 0;
    if (tid < 4) {   // the zero pad behind the wrap-around word, both copies
      sh.d[tid >> 1][512 + (tid & 1)] = 0;
      d_alt[(tid >> 1) * 514 + 512 + (tid & 1)] = 0;
    }
    mx_block_commit(sh, b0, prm.if_format, tid);
    __syncthreads();
    mx_byte_wipe_codes(sh, d_of(0), ones_of(0), base0, base8_of(0), step_of(0), tid, lane);
    __syncthreads();
    mx_byte_vector_pair(base0, &sh.e8[0][0][0][0], &sh.e8[1][0][0][0], tid);
    mx_block_commit(sh, b1, prm.if_format, tid);
  }
  u32 kq[kMxTiles];   // 2047 - (even byte offset of the lane's chip offset in tile j): the low field of its search keys
#pragma unroll
  for (int j = 0; j < kMxTiles; j++)
    kq[j] = (u32)(2047 - 2 * (32 * (q0_tile + 2 * j) + (lane & 31)));

  // Half stages: per cluster four -- passes of sample offset 0 (start values, two passes on the wave's four q-tiles), its epilogue
  // (64 hypotheses per lane), passes of offset 8 (with the odd offset's extra K step), its epilogue; role 1 one half stage
  // behind role 0.  One barrier per stage, i.e. two per cluster, and behind them by everybody (c = the cluster role 0 is in):
  //   start of the offset-0 stage:  the offset-8 vectors of c (read until the half stage before); wipe-off / pop(D) / sums of
  //                                 c + 1 into the copies c - 1 had; the request for c + 2's block (registers);
  //   start of the offset-8 stage:  the offset-0 vectors of c + 1; c + 2's block -> LDS, its pop(D) counters zeroed; the
  //                                 triplets of c - 1 (its last epilogue ran in the half stage before), its slots zeroed.
  MxBlockRegs next_block = {{0, 0, 0, 0}};
  v16f acc[2][kMxTiles];
  const int n_half = 4 * n_my;
#ifdef GPSX_MX_TIMELINE   // (tools/experiments/byte_timeline.py: cycle stamps of workgroup 0's waves 0 and 4 behind the peaks)
  unsigned long long *tl = blockIdx.x == 0 && lane == 0 && (wave & 3) == 0
                               ? reinterpret_cast<unsigned long long *>(peaks + (size_t)prm.n_clusters * 32) + role * 2048 : nullptr;
  int tli = 0;
  int tli2 = 1024;
#define MX_TL() do { if (tl && tli < 1024) tl[tli++] = __builtin_readcyclecounter(); } while (0)
#define MX_TL2() do { if (tl && tli2 < 2048) tl[tli2++] = __builtin_readcyclecounter(); } while (0)
#else
#define MX_TL() do { } while (0)
#define MX_TL2() do { } while (0)
#endif
  // the pieces behind the barrier of even half stage hs_even, thread t's share
  auto piece = [&](int hs_even, int t) {
    asm volatile("" : "+v"(t));   // (per-thread addresses of these pieces are recomputed, not kept across the stages)
    const int c = hs_even >> 2;
    window_to(c);
    const bool steady = c >= 1 && c + 2 < n_my;   // every piece exists: one straight run, their LDS round trips overlap
#ifdef GPSX_MX_NO_PIECES   // (timing ablation: results are then wrong)
    if (steady)
      return;
#endif
    if ((hs_even & 2) == 0) {
      if (steady) {
        const u32 step = step_of(c + 1);
        const uint8_t *blk = block_of(c + 2);
        next_block = mx_block_request(blk, prm.if_format, t);
        mx_byte_vector_pair(base8_of(c), e8x, e8x + kVec, t);
        mx_byte_wipe_codes(sh, d_of(c + 1), ones_of(c + 1), base0, base8_of(c + 1), step, t, t & 63);
      } else {
        if (c < n_my)
          mx_byte_vector_pair(base8_of(c), e8x, e8x + kVec, t);
        if (c + 1 < n_my)
          mx_byte_wipe_codes(sh, d_of(c + 1), ones_of(c + 1), base0, base8_of(c + 1), step_of(c + 1), t, t & 63);
        if (c + 2 < n_my)
          next_block = mx_block_request(block_of(c + 2), prm.if_format, t);
      }
    } else {
      int search, dopp;
      u32 mask;
      if (steady) {
        decode(c - 1, search, dopp, mask);
        mx_block_commit(sh, next_block, prm.if_format, t);
        if (t < 2)
          ones_of(c + 2)[t] = 0;
        mx_byte_fold(sh, (c - 1) & 1 ? kSlotsOdd : kSlotsEven, mask, set, search, dopp, prm, peaks, t);
        mx_byte_vector_pair(base0, &sh.e8[0][0][0][0], &sh.e8[1][0][0][0], t);
      } else {
        if (c + 1 < n_my)
          mx_byte_vector_pair(base0, &sh.e8[0][0][0][0], &sh.e8[1][0][0][0], t);
        if (c + 2 < n_my) {
          mx_block_commit(sh, next_block, prm.if_format, t);
          if (t < 2)
            ones_of(c + 2)[t] = 0;
        }
        if (c >= 1) {
          decode(c - 1, search, dopp, mask);
          mx_byte_fold(sh, (c - 1) & 1 ? kSlotsOdd : kSlotsEven, mask, set, search, dopp, prm, peaks, t);
        }
      }
    }
  };
#pragma unroll 1
  for (int hs = 0; hs <= n_half; hs++) {
    MX_TL();
    if ((hs & 1) == 0)
      __syncthreads();
    MX_TL();
    // A stage's pieces only have to be done before the NEXT barrier, and what they write nobody reads before it: role 1 does
    // its threads' share at once (pieces, epilogue, passes), role 0 at the end of its stage (passes, epilogue, pieces) -- the
    // two waves of a SIMD are then on the matrix pipe one after the other from the barrier on.
    if (role == 1 && (hs & 1) == 0)
      piece(hs, tid);
    MX_TL();
    const int x = hs - role;   // this role's half stage
    if (x >= 0 && x < n_half) {
      const int cc = x >> 2, o = (x >> 1) & 1;   // sample offset 8 o
      if ((x & 1) == 0) {
        const u32 *dd = d_of(cc), *ones = ones_of(cc);
        const u32 *va = o ? e8x : &sh.e8[0][0][0][0], *vb = o ? e8x + kVec : &sh.e8[1][0][0][0];
        if (o) {
          u32 b_low[2][kMxTiles], b_high[2][kMxTiles];
          mx_odd_tail_operands(va, vb, lane, q0_tile, b_low, b_high);
          mx_init_acc_odd(ones, dd, dd + 514, lane, q0_tile, acc, prm.win_start, prm.win_stop);
          mx_odd_tail_steps(sh, lane, b_low, b_high, acc);
        } else {
          mx_init_acc(ones, lane, q0_tile, acc, prm.win_start, prm.win_stop);
        }
        mx_pass2(sh, lane, q0_tile, acc, va, vb);
      } else {
        mx_epilogue_single(sh, lane, kq, 8 * o, acc, false, false, cc & 1 ? kSlotsOdd : kSlotsEven);
      }
    }
    if (role == 0 && (hs & 1) != 0)
      piece(hs - 1, tid);
  }
  __syncthreads();
  {
    int search, dopp;
    u32 mask;
    decode(n_my - 1, search, dopp, mask);
    mx_byte_fold(sh, (n_my - 1) & 1 ? kSlotsOdd : kSlotsEven, mask, set, search, dopp, prm, peaks, tid);
  }
}

template <int MODE>
__global__ __launch_bounds__(kMxThreads, 1) void k_acq_mx(const AcqParams prm, int cluster_lo, const uint8_t *__restrict__ if_blocks,
                                                          const u32 *__restrict__ mx_a, const u32 *__restrict__ mx_t,
                                                          gpsx_peak_t *__restrict__ peaks, u32 *__restrict__ energy,
                                                          u32 *__restrict__ flags)
{
  __shared__ MxShared sh;
  if constexpr (MODE == kMxByte) {
    // persistent: one workgroup per CU walks its clusters as one software pipeline
    mx_byte_pipe(sh, prm, cluster_lo, if_blocks, mx_a, mx_t, peaks);
  } else {
    int tables_set = -1;
    mx_unit<MODE>(sh, prm, (int)blockIdx.x, tables_set, cluster_lo, if_blocks, mx_a, mx_t, peaks, energy, flags);
  }
}

// clusters that intersect this shard's run of units
static void mx_cluster_range(const AcqParams &prm, int *c_lo, int *c_hi)
{
  const int n_sets = (prm.n_groups + 3) / 4;
  auto cluster_of = [&](int unit) { return (unit / prm.n_groups) * n_sets + (unit % prm.n_groups) / 4; };
  *c_lo = cluster_of(prm.unit_lo);
  *c_hi = cluster_of(prm.unit_hi - 1) + 1;
}

long acq_mx_clusters(const AcqParams &prm)
{
  if (prm.unit_hi <= prm.unit_lo)
    return 0;
  int c_lo, c_hi;
  mx_cluster_range(prm, &c_lo, &c_hi);
  return c_hi - c_lo;
}

const char *launch_acq_mx(hipStream_t s, const AcqParams &prm, const uint8_t *d_if, const uint32_t *d_mx_a,
                          const uint32_t *d_mx_t, gpsx_peak_t *d_peaks, uint32_t *d_energy, bool block_parallel, size_t n_peaks,
                          uint32_t *d_planes, int n_cus, bool *keys_done)
{
  *keys_done = false;   // true: every key of the launch was written by the kernels themselves (prm.keys)
  if (prm.unit_hi <= prm.unit_lo)
    return "";
  int c_lo, c_hi;
  mx_cluster_range(prm, &c_lo, &c_hi);
  if (prm.n_ms > 1 && block_parallel) {
    hipLaunchKernelGGL(k_acq_mx<kMxStore>, dim3((unsigned)((c_hi - c_lo) * prm.n_ms)), dim3(kMxThreads), 0, s, prm, c_lo, d_if,
                       d_mx_a, d_mx_t, d_peaks, d_energy, (u32 *)nullptr);
    launch_acq_vals_search(s, prm, reinterpret_cast<const uint16_t *>(d_energy), d_peaks, n_peaks);
    return "k_acq_mx<2>";
  }
  if (prm.n_ms > 1) {
    // 16-bit running sums first; where they cannot overflow (n_ms x 11573 < 2^16) that is all, otherwise the 24-bit form
    // follows and redoes the clusters whose flag went up.  The flags sit behind the records.
    const unsigned n_wg = (unsigned)(c_hi - c_lo);
    u32 *d_flags = d_energy + acq_mx_energy_bytes(n_wg) / sizeof(u32) - n_wg;
    (void)hipMemsetAsync(d_flags - kMxZeroRecBytes / sizeof(u32), 0, kMxZeroRecBytes, s);   // the first block's "previous sums"
    hipLaunchKernelGGL(k_acq_mx<kMxWalk16>, dim3(n_wg), dim3(kMxThreads), 0, s, prm, c_lo, d_if, d_mx_a, d_mx_t, d_peaks,
                       d_energy, d_flags);
    if (prm.n_ms * 11573 > 65535)
      hipLaunchKernelGGL(k_acq_mx<kMxWalk>, dim3(n_wg), dim3(kMxThreads), 0, s, prm, c_lo, d_if, d_mx_a, d_mx_t, d_peaks,
                         d_energy, d_flags);
    *keys_done = prm.keys != nullptr;
    return "k_acq_mx<3>";
  }
  if (prm.n_bits == 1) {   // byte-phase grid: sample offsets 0 and 8, each started from its own block sums
    AcqParams bp = prm;
    bp.n_clusters = c_hi - c_lo;
    // one software pipeline per persistent workgroup; a workgroup keeps ONE PRN set's tables: the grid is a multiple of n_sets
    const int n_sets = (prm.n_groups + 3) / 4;
    const int grid = bp.n_clusters < n_cus ? bp.n_clusters : n_cus - n_cus % n_sets;
    hipLaunchKernelGGL(k_acq_mx<kMxByte>, dim3((unsigned)grid), dim3(kMxThreads), 0, s, bp, c_lo, d_if, d_mx_a, d_mx_t, d_peaks,
                       (u32 *)nullptr, (u32 *)nullptr);
    *keys_done = prm.keys != nullptr;
    return "k_acq_mx<4>";
  }
  if (d_planes && 2 * (c_hi - c_lo) <= n_cus) {
    // fewer clusters than half the chip (a lone cold start is 21): two workgroups per cluster with eight sample offsets each,
    // four with four each from a quarter of the chip down
    AcqParams sp = prm;
    const int nc = c_hi - c_lo;
    sp.split_segs = 8 * nc <= n_cus ? 8 : 4 * nc <= n_cus ? 4 : 2;   // (a lone cold start: 21 clusters -> 168 workgroups)
    if (prm.split_segs)   // ($GPSX_ACQ_SPLIT, read when the context was created: 2, 4 or 8 whatever the launch size -- tests, A/B)
      sp.split_segs = prm.split_segs;
    sp.n_planes = n_peaks;   // (the planes are all-zero between launches: k_acq_finalize puts back what it reads)
    hipLaunchKernelGGL(k_acq_mx<kMxSplit>, dim3((unsigned)(sp.split_segs * (c_hi - c_lo))), dim3(kMxThreads), 0, s, sp, c_lo, d_if,
                       d_mx_a, d_mx_t, d_peaks, d_planes, (u32 *)nullptr);
    launch_acq_finalize(s, d_planes, d_planes + n_peaks, n_peaks, d_peaks, prm.n_bits == 8 ? prm.keys : nullptr);
    *keys_done = prm.n_bits == 8 && prm.keys != nullptr;
    return "k_acq_mx<5>";
  }
  // One workgroup per cluster and CU: a launch is rounds of n_cus clusters, and a last round that fills at most half of the
  // chip takes as long as a full one.  Then the full rounds go out as they are and the leftover clusters in the split form
  // behind them -- 2, 4 or 8 workgroups per cluster, as many as still fit ONE round, each with a run of sample offsets it
  // starts directly (16 captures = 336 clusters: 256 + 80 x 2; 64 captures = 1344: 1280 + 64 x 4).
  const int nc = c_hi - c_lo, tail = nc % n_cus;
  if (d_planes && nc > n_cus && tail > 0 && 2 * tail <= n_cus) {
    const int c_tail = c_hi - tail;
    hipLaunchKernelGGL(k_acq_mx<kMxSingle>, dim3((unsigned)(nc - tail)), dim3(kMxThreads), 0, s, prm, c_lo, d_if, d_mx_a, d_mx_t,
                       d_peaks, (u32 *)nullptr, (u32 *)nullptr);
    AcqParams sp = prm;
    sp.split_segs = 8 * tail <= n_cus ? 8 : 4 * tail <= n_cus ? 4 : 2;
    if (prm.split_segs && prm.split_segs * tail <= n_cus)   // $GPSX_ACQ_SPLIT here too, as long as the tail still fits one round
      sp.split_segs = prm.split_segs;
    sp.n_planes = n_peaks;
    // the planes of the tail's peaks only: from the first peak of the search the tail begins in
    const int n_sets = (prm.n_groups + 3) / 4;
    const size_t per_search = (size_t)prm.n_prn * prm.n_dopp * prm.n_bits;
    const size_t first = (size_t)(c_tail / (n_sets * prm.n_dopp)) * per_search;
    // (no memset: the planes are all-zero between launches, k_acq_finalize* puts back what it reads)
    hipLaunchKernelGGL(k_acq_mx<kMxSplit>, dim3((unsigned)(sp.split_segs * tail)), dim3(kMxThreads), 0, s, sp, c_tail, d_if,
                       d_mx_a, d_mx_t, d_peaks, d_planes, (u32 *)nullptr);
    launch_acq_finalize_from(s, d_planes, d_planes + n_peaks, first, n_peaks, d_peaks, prm.n_prn, prm.n_dopp, prm.n_bits, n_sets,
                             c_tail, prm.n_bits == 8 ? prm.keys : nullptr);
    *keys_done = prm.n_bits == 8 && prm.keys != nullptr;   // (the full rounds' workgroups wrote theirs in their folds)
    return "k_acq_mx<0>";
  }
  hipLaunchKernelGGL(k_acq_mx<kMxSingle>, dim3((unsigned)(c_hi - c_lo)), dim3(kMxThreads), 0, s, prm, c_lo, d_if, d_mx_a, d_mx_t,
                     d_peaks, (u32 *)nullptr, (u32 *)nullptr);
  *keys_done = prm.keys != nullptr;
  return "k_acq_mx<0>";
}

// =============================================================================================================================
// EXTENSION, not in the reference: the weighted two-bit grid (include/gpsx.h gpsx_acq_grid_weighted, k_acq_weighted.hip for the
// definition) on the matrix cores -- the same Toeplitz GEMM, with values where the sign-only grid has bits.
//   v(n) in {0, +-1, +-3}: the wiped sample's sign x its magnitude weight; the sixteen samples the carrier NCO never mixes: 0
//   I(16 q + t0) = sum_c s[c] S_t0[(q + c) mod 1023],  s = 1 - 2 chip,  S_t0[k] = sum of v over the window [16 k + t0, +16)
//                = T - 2 sum_c chip[c] S_t0[q + c],    T = sum of all v (every sample sits in exactly one window)
// A = chips (FP4 1.0, the tables of the sign-only grid, at block scale 2^0: the accumulators hold plain integers), start value T:
//   * sample offset 0 in THREE passes: y = -S_0 in [-48, 48] = y0 + 4 y1 + 16 y2 with balanced base-4 digits in [-2, 2]
//     (|y2| <= 3), the vector carries 2 y_i (FP4-exact: 0, +-2, +-4, +-6) at block scales 2^0, 2^2, 2^4;
//   * every further offset in one: S_{t0+1}[k] - S_t0[k] = v_t0(k + 1) - v_t0(k) in {0, +-1, +-2, +-3, +-4, +-6} (a difference
//     of 5 does not exist) with v_t0(i) = v(16 i + t0), i mod 1023, and v_t0(1022) = 0 (the unmixed samples, whatever t0):
//     the vector carries its negative at block scale 2^1.
// Every partial sum is an integer below 2^24 at a power-of-two scale: exact in f32 in any order, like the sign-only grid.
// The epilogue is this grid's own: no clipping (the correlation is signed), floor(sqrt(I^2 + Q^2)) exactly, the first fine phase
// reaching the maximum, the sum.  Work split, pipelining of the two roles and the result slots are k_acq_mx<0>'s.
namespace {

struct MxwShared {
  MxShared s;
  u32 mag[514];                     // the capture's magnitude plane, laid out as s.d (zero in the sign-only mode)
  u32 mplane[16][kPlaneWordsMx];    // polyphase magnitude planes, as s.plane
  int wsum[2];                      // sum over the mixed samples of (2 d - 1) m, per stream
};
#ifndef MXW_ABL
#define MXW_ABL 0   // (lab builds: timing ablations -- 1 no epilogue, 2 no vector building, 4 no MFMA pass, 8 raised priority for the
                    //  passes, 16 no magnitude test in front of the epilogue, 32 no pass in waves 4-7, 64 no epilogue in waves 0-3;
                    //  wrong results)
#endif
constexpr int kWPasses = 18;                 // 3 for the first offset + 15 recurrence steps
constexpr u32 kScaleTwo = 0x80808080u;       // E8M0 128 = 2^1
constexpr u32 kScaleFour = 0x81818181u;      // 2^2
constexpr u32 kScaleSixteen = 0x83838383u;   // 2^4

__device__ __forceinline__ void mxw_write_copies(u32 *dst, u32 lo, u32 hi)
{
#pragma unroll
  for (int c = 0; c < 8; c++)
    dst[c * kCopyDwords] = c ? __builtin_amdgcn_alignbit(hi, lo, 4u * (u32)c) : lo;
}

// digit `which` of the first offset's chip sums (thread (stream, j): entries 8 j .. 8 j + 15 of copy 0 -> dword j of the copies)
__device__ __forceinline__ void mxw_build_start(MxwShared &shw, int which, int buf, int tid)
{
  const int iq = tid >> 8, j = tid & 255;
  const u32 *dd = shw.s.d[iq], *mm = shw.mag;
  u32 w2[2] = {0, 0};
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k = wrap1023(8 * j + e);
    const u32 x = (dd[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, m = (mm[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
    int y = -((2 * (int)__popc(x) - 16) + 2 * (2 * (int)__popc(x & m) - (int)__popc(m)));
    y = k == kChips - 1 ? 0 : y;                       // window 1022 = the sixteen unmixed samples
    const int y0 = ((y + 2) & 3) - 2, r1 = (y - y0) >> 2;
    const int y1 = ((r1 + 2) & 3) - 2, y2 = (r1 - y1) >> 2;
    const int digit = which == 0 ? y0 : which == 1 ? y1 : y2;
    w2[e >> 3] |= fp4_code(2 * digit) << (4 * (e & 7));
  }
  mxw_write_copies(&shw.s.e8[buf][iq][0][j], w2[0], w2[1]);
}

// the vector that takes the accumulators from sample offset t0 to t0 + 1: entry k = v_t0(k) - v_t0(k + 1).
// Lookup table (in the sign-only grid's t_lut, which this kernel does not use otherwise): (sign, magnitude) pairs of five
// consecutive samples -- ten bits, sample i in bits 2 i, 2 i + 1 -- -> the FP4 codes of their four differences
__device__ __forceinline__ int mxw_val2(u32 sm) { return ((sm & 1u) ? 1 : -1) * ((sm & 2u) ? 3 : 1); }
__device__ void mxw_fill_table(MxShared &sh, int tid)
{
  uint16_t *lut = reinterpret_cast<uint16_t *>(sh.t_lut);
  static_assert(sizeof(sh.t_lut) >= 1024 * sizeof(uint16_t), "difference table fits");
  for (int i = tid; i < 1024; i += kMxThreads) {
    u32 codes = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
      codes |= fp4_code(mxw_val2(((u32)i >> (2 * k)) & 3u) - mxw_val2(((u32)i >> (2 * k + 2)) & 3u)) << (4 * k);
    lut[i] = (uint16_t)codes;
  }
}
// 16 bits -> the even bit positions of 32
__device__ __forceinline__ u32 spread16(u32 x)
{
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  return (x | (x << 1)) & 0x55555555u;
}
__device__ __forceinline__ void mxw_build_step(MxwShared &shw, int t0, int buf, int tid)
{
  const int iq = tid >> 8, j = tid & 255;
  const u32 *pl = shw.s.plane[iq][t0], *mp = shw.mplane[t0];
  const uint16_t *lut = reinterpret_cast<const uint16_t *>(shw.s.t_lut);
  const u32 sx = __builtin_amdgcn_alignbit(pl[(j >> 2) + 1], pl[j >> 2], 8u * (u32)(j & 3));   // plane bits 8 j .. 8 j + 31
  const u32 mx = __builtin_amdgcn_alignbit(mp[(j >> 2) + 1], mp[j >> 2], 8u * (u32)(j & 3));
  const u32 z_lo = spread16(sx & 0xFFFFu) | (spread16(mx & 0xFFFFu) << 1);                      // samples 0..15 of the window
  const u32 z_hi = ((sx >> 16) & 1u) | (((mx >> 16) & 1u) << 1);                                // sample 16
  u32 w2[2];
  w2[0] = (u32)lut[z_lo & 0x3FFu] | ((u32)lut[(z_lo >> 8) & 0x3FFu] << 16);
  w2[1] = (u32)lut[(z_lo >> 16) & 0x3FFu] | ((u32)lut[(z_lo >> 24) | ((z_hi & 3u) << 8)] << 16);
  // entries 1021, 1022 (dword 127, nibbles 5 and 6) and their wrap-around copies 2044, 2045 (dword 255, nibbles 4 and 5: 1023 is
  // odd) touch the unmixed samples, v(1022) = 0: they are v(1021) - 0 and 0 - v(0).  Every thread works the two codes out (four
  // broadcast reads) and patches by selection: a branch here put 300 instructions with dependent LDS reads on two waves' paths
  {
    const u32 c1 = fp4_code(mxw_val2(((pl[31] >> 29) & 1u) | (((mp[31] >> 29) & 1u) << 1)));
    const u32 c2 = fp4_code(-mxw_val2((pl[0] & 1u) | ((mp[0] & 1u) << 1)));
    const u32 at127 = (c1 << 20) | (c2 << 24), at255 = (c1 << 16) | (c2 << 20);
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int dword = j + k;
      u32 w = w2[k];
      w = dword == 127 ? (w & ~0x0FF00000u) | at127 : w;
      w = dword == 255 ? (w & ~0x00FF0000u) | at255 : w;
      w2[k] = w;
    }
  }
  mxw_write_copies(&shw.s.e8[buf][iq][0][j], w2[0], w2[1]);
}

__device__ __forceinline__ u32 mxw_root_exact(int i, int q)
{
  const u64 e = (u64)((long long)i * i) + (u64)((long long)q * q);
  u64 r = (u64)__builtin_sqrt((double)e);
  r = r * r > e ? r - 1 : r;
  r = (r + 1) * (r + 1) <= e ? r + 1 : r;
  return (u32)r;
}

// floor(sqrt(E)) for E = I^2 + Q^2 < 2^24 - 2 (an integer, exact in f32), in eight instructions: v_sqrt_f32 is good to one ulp,
// at most 2^-12 below 4096, and sqrt(E + 2) - sqrt(E) = 2 / (sqrt(E + 2) + sqrt(E)) > 2^-12 there: the root of E + 2 as the
// hardware returns it is not below floor(sqrt(E)) =: r and stays below r + 2 -- its truncation is r or r + 1, and
// (E + 2) - (r + 1)^2 < 2 (exact) tells which.  The + 2 rides in the first multiply-add.
__device__ __forceinline__ u32 mxw_root_small(float fi, float fq)
{
  const float e2 = __builtin_fmaf(fi, fi, __builtin_fmaf(fq, fq, 2.0f));
  const u32 r = (u32)__builtin_amdgcn_sqrtf(e2);
  const float rf = (float)r;
  return __builtin_fmaf(-rf, rf, e2) < 2.0f ? r - 1u : r;
}

// the epilogue of sample offset t0: 64 hypotheses per lane into the slots of bit shift t0 & 7 (byte offset 2 q + (t0 >> 3))
template <bool ALL_SMALL>
__device__ __forceinline__ void mxw_epilogue_body(MxShared &sh, int lane, int q0_tile, int t0, const v16f (&acc)[2][kMxTiles],
                                                  const bool (&small)[kMxTiles])
{
  const int n = lane & 31, h = lane >> 5;
  u32 key_lo[kMxTiles];
#pragma unroll
  for (int j = 0; j < kMxTiles; j++)
    key_lo[j] = (u32)(2047 - (2 * (32 * (q0_tile + 2 * j) + n) + (t0 >> 3)));
  const bool last_exists = 32 * (q0_tile + 2 * (kMxTiles - 1)) + n < kChips;   // chip offset 1023 (tile 31, lane 31) does not exist
  u32 *slot = &sh.part[t0 & 7][4 * h][0][n];
  if constexpr (ALL_SMALL) {
    // two PRNs (eight hypotheses) at a time, stage by stage: eight independent instructions between dependent ones -- a vector
    // instruction behind the one it depends on waits out its latency, next to the other wave's MFMAs even longer
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float e2[8], rf[8];
      u32 root[8];
#pragma unroll
      for (int i = 0; i < 8; i++)
        e2[i] = __builtin_fmaf(acc[1][i & 3][r + (i >> 2)], acc[1][i & 3][r + (i >> 2)], 2.0f);
#pragma unroll
      for (int i = 0; i < 8; i++)
        e2[i] = __builtin_fmaf(acc[0][i & 3][r + (i >> 2)], acc[0][i & 3][r + (i >> 2)], e2[i]);
#pragma unroll
      for (int i = 0; i < 8; i++)
        rf[i] = __builtin_amdgcn_sqrtf(e2[i]);
#pragma unroll
      for (int i = 0; i < 8; i++)
        root[i] = (u32)rf[i];
#pragma unroll
      for (int i = 0; i < 8; i++)
        rf[i] = (float)root[i];
#pragma unroll
      for (int i = 0; i < 8; i++)
        rf[i] = __builtin_fmaf(-rf[i], rf[i], e2[i]);
#pragma unroll
      for (int i = 0; i < 8; i++)
        root[i] = rf[i] < 2.0f ? root[i] - 1u : root[i];
      root[3] = last_exists ? root[3] : 0u;
      root[7] = last_exists ? root[7] : 0u;
      asm volatile("" : "+v"(root[0]), "+v"(root[1]), "+v"(root[2]), "+v"(root[3]), "+v"(root[4]), "+v"(root[5]), "+v"(root[6]), "+v"(root[7]));
#pragma unroll
      for (int k = 0; k < 2; k++) {
        u32 best = 0, total = 0;
#pragma unroll
        for (int j = 0; j < kMxTiles; j++) {
          const u32 key = (root[4 * k + j] << 11) | key_lo[j];
          best = key > best ? key : best;
          total += root[4 * k + j];
        }
        const int p = ((r + k) & 3) + 8 * ((r + k) >> 2);
        atomicMax(&slot[p * 64], best);
        atomicAdd(&slot[p * 64 + 32], total);
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; r++) {
    u32 best = 0, total = 0;
#pragma unroll
    for (int j = 0; j < kMxTiles; j++) {
      const float fi = acc[0][j][r], fq = acc[1][j][r];     // (plain integers: the A operand's block scale is 2^0 here)
      u32 m;
      if (small[j])
        m = mxw_root_small(fi, fq);
      else
        m = mxw_root_exact((int)fi, (int)fq);
      if (j == kMxTiles - 1)
        m = last_exists ? m : 0u;
      const u32 key = (m << 11) | key_lo[j];
      best = key > best ? key : best;
      total += m;
    }
    const int p = (r & 3) + 8 * (r >> 2);              // PRN p + 4 h of the cluster
    atomicMax(&slot[p * 64], best);
    atomicAdd(&slot[p * 64 + 32], total);
  }
}
// max(m, |a|, |b|) in ONE instruction (the source modifiers of v_max3_f32; written out because fmaxf() on fabsf() compiles to a
// canonicalising v_max_f32 |x|, |x| per operand in front of the maximum: 3.5 instructions per pair instead of one)
__device__ __forceinline__ float mxw_max_abs(float m, float a, float b)
{
  asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(a), "v"(b));
  return m;
}
__device__ __forceinline__ void mxw_epilogue(MxShared &sh, int lane, int q0_tile, int t0, const v16f (&acc)[2][kMxTiles])
{
  // (wave-uniform) every |I|, |Q| of the wave's 64 x 64 hypotheses below 2896: I^2 + Q^2 + 2 < 2^24 -- all but the tiles next to a
  // strong satellite's peak
  float lim[kMxTiles];
#pragma unroll
  for (int j = 0; j < kMxTiles; j++)
    lim[j] = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; r++)
#pragma unroll
    for (int j = 0; j < kMxTiles; j++)   // (four independent chains)
      lim[j] = mxw_max_abs(lim[j], acc[0][j][r], acc[1][j][r]);
  const float top = __builtin_fmaxf(__builtin_fmaxf(lim[0], lim[1]), __builtin_fmaxf(lim[2], lim[3]));
  bool small[kMxTiles];
  if (__builtin_amdgcn_ballot_w64(top >= 2896.0f) == 0 || (MXW_ABL & 16)) {
    mxw_epilogue_body<true>(sh, lane, q0_tile, t0, acc, small);
  } else {
#pragma unroll
    for (int j = 0; j < kMxTiles; j++)
      small[j] = __builtin_amdgcn_ballot_w64(lim[j] >= 2896.0f) == 0;
    mxw_epilogue_body<false>(sh, lane, q0_tile, t0, acc, small);
  }
}

}  // namespace

__global__ __launch_bounds__(kMxThreads, 1) void k_acq_mxw(const uint8_t *__restrict__ if_blocks, int stride_blocks, int n_prn,
                                                           const u32 *__restrict__ mx_a, int if_hz, int dopp_min_hz, int dopp_step_hz,
                                                           int n_dopp, int use_magnitude, gpsx_peak_t *__restrict__ peaks)
{
  __shared__ MxwShared shw;
  MxShared &sh = shw.s;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2;                            // waves w and w + 4 share a SIMD: half a step apart
  const int q0_tile = 8 * (wave >> 1) + (wave & 1);      // this wave owns q-tiles q0_tile + 2 j
  const int n_sets = (n_prn + 31) / 32;
  const int cluster = (int)blockIdx.x;
  const int set = cluster % n_sets, sd = cluster / n_sets, dopp = sd % n_dopp, search = sd / n_dopp;
  const u32 step_word = nco_step_per_word((float)(if_hz + dopp_min_hz + dopp * dopp_step_hz));
  const uint8_t *blk = if_blocks + (size_t)search * stride_blocks * GPSX_BYTES_PER_MS_2BIT;

  // ---- the cluster's chips, the capture's two bit planes -----------------------------------------------------------------
  {
    const u32 *src_a = mx_a + (size_t)set * (16 * 2 * 32 * 4);
    u32 *dst_a = reinterpret_cast<u32 *>(&sh.chips_a[0][0][0]);
    for (int i = tid; i < 16 * 2 * 32 * 4; i += kMxThreads)
      dst_a[i] = src_a[i];
  }
  for (int i = tid; i < 8 * 32 * 2 * 32 / 4; i += kMxThreads)
    reinterpret_cast<uint4 *>(&sh.part[0][0][0][0])[i] = make_uint4(0, 0, 0, 0);
  mxw_fill_table(sh, tid);
  mx_load_block(sh, blk, GPSX_IF_2BIT_SM, tid);
  for (int w = tid; w < 514; w += kMxThreads) {
    u32 m = 0;
    if (use_magnitude && w < 512) {
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        const int w16 = 2 * w + hh;
        if (w16 < kWords16) {
          const uint16_t *p = reinterpret_cast<const uint16_t *>(blk) + 2 * w16;
          m |= even_bits16(((u32)p[0] | ((u32)p[1] << 16)) >> 1) << (16 * hh);
        }
      }
    }
    shw.mag[w] = m;
  }
  if (tid < 2)
    shw.wsum[tid] = 0;
  __syncthreads();
  if (tid == 0)
    shw.mag[511] |= shw.mag[0] << 16;                    // the stream wraps to sample 0 (as s.d's word 511)
  mx_wipe_block(sh, step_word, tid, lane);
  // ---- magnitude planes (first period), the weighted part of the streams' totals, the first two vectors -----------------------
  for (int m = tid; m < 32 * 16; m += kMxThreads) {
    const int t0 = m & 15, w = m >> 4;
    const u32 *src = &shw.mag[16 * w];
    u32 bits = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const u32 sk = src[k];
      bits |= ((sk >> t0) & 1u) << (2 * k);
      bits |= ((sk >> (16 + t0)) & 1u) << (2 * k + 1);
    }
    shw.mplane[t0][w] = bits;
  }
  {
    int part_i = 0, part_q = 0;
    for (int w = tid; w < kWords32; w += kMxThreads) {
      const u32 m = shw.mag[w];
      part_i += 2 * (int)__popc(sh.d[0][w] & m) - (int)__popc(m);
      part_q += 2 * (int)__popc(sh.d[1][w] & m) - (int)__popc(m);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      part_i += __shfl_xor(part_i, off);
      part_q += __shfl_xor(part_q, off);
    }
    if (lane == 0) {
      atomicAdd(&shw.wsum[0], part_i);
      atomicAdd(&shw.wsum[1], part_q);
    }
  }
  mxw_build_start(shw, 0, 0, tid);
  mxw_build_start(shw, 1, 1, tid);
  __syncthreads();
  for (int m = tid; m < 16 * (kPlaneWordsMx - 32); m += kMxThreads) {   // circular extension, as mx_wipe_block's
    const int w = 32 + m % (kPlaneWordsMx - 32);
    const int r = m / (kPlaneWordsMx - 32);
    const u32 *pl = shw.mplane[r];
    const int pos = 32 * w - (w >= 64 ? 2 * kChips : kChips);
    const int lo = pos >> 5;
    u32 v = __builtin_amdgcn_alignbit(lo < 31 ? pl[lo + 1] : 0u, pl[lo], (u32)(pos & 31));
    if (pos + 32 > kChips) {
      const int k = kChips - pos;
      v = (v & ((1u << k) - 1u)) | (pl[0] << k);
    }
    shw.mplane[r][w] = v;
  }

  v16f acc[2][kMxTiles];
  {
    const float t_i = (float)(2 * (int)sh.ones[0] - 32 * kWords32 + 2 * shw.wsum[0]);
    const float t_q = (float)(2 * (int)sh.ones[1] - 32 * kWords32 + 2 * shw.wsum[1]);
#pragma unroll
    for (int j = 0; j < kMxTiles; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        acc[0][j][r] = t_i;
        acc[1][j][r] = t_q;
      }
  }
  const v4i no_corr = v4i{0, 0, 0, 0};
  // steps of two halves, as mx_unit: role 0 runs pass p, then the epilogue of the offset pass p - 1 finished; role 1 the
  // epilogue first, then the pass; one barrier per step.  The vector of pass p + 1 is built during step p by role 0 alone
#if MXW_ABL & 256   // (lab: where one workgroup's waves spend their cycles -- barrier, vector building, pass, epilogue)
  unsigned long long t_bar = 0, t_build = 0, t_pass = 0, t_epi = 0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
#define MXW_T(acc_var) do { const unsigned long long now = __builtin_readcyclecounter(); acc_var += now - t_last; t_last = now; } while (0)
#else
#define MXW_T(acc_var) do { } while (0)
#endif
#pragma unroll 1
  for (int hs = 0; hs <= 2 * kWPasses; hs++) {
#if MXW_ABL & 256
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    if ((hs & 1) == 0) {
      __syncthreads();
      MXW_T(t_bar);
    }
    MXW_T(t_build);
    const int x = hs - role;
    const bool active = x >= 0 && x < 2 * kWPasses;
    const int p = x >> 1;
    if (active && (x & 1) == 0) {
      if (!(MXW_ABL & 4) && !((MXW_ABL & 32) && role)) {
        if (MXW_ABL & 8)
          __builtin_amdgcn_s_setprio(2);
        mx_pass<true, kMxTiles, (MXW_ABL & 128) ? kScaleA : kScaleOne>(sh, p & 1, lane, q0_tile, acc, p == 0 ? kScaleOne : p == 1 ? kScaleFour : p == 2 ? kScaleSixteen : kScaleTwo,
                      no_corr, false);
        if (MXW_ABL & 8)
          __builtin_amdgcn_s_setprio(0);
      } else
#pragma unroll
        for (int j = 0; j < kMxTiles; j++)
          asm volatile("" : "+v"(acc[0][j]), "+v"(acc[1][j]));
    }
    MXW_T(t_pass);
    if (active && (x & 1) && p >= 2) {
      if (!(MXW_ABL & 1) && !((MXW_ABL & 64) && !role)) {
        mxw_epilogue(sh, lane, q0_tile, p - 2, acc);
      } else
#pragma unroll
        for (int j = 0; j < kMxTiles; j++)
          asm volatile("" ::"v"(acc[0][j]), "v"(acc[1][j]));
    }
    MXW_T(t_epi);
    // the vector of the NEXT step's pass, by the waves of role 0 alone, behind their epilogue: they are the ones that wait at the
    // step's barrier (role 1's epilogue runs beside a pass and takes half as long again); the buffer was last read in the
    // previous step
    if (role == 0 && (hs & 1)) {
      const int p_vec = (hs >> 1) + 1;
      if (p_vec == 2) {
        mxw_build_start(shw, 2, 0, tid);
        mxw_build_start(shw, 2, 0, tid + 256);
      } else if (p_vec > 2 && p_vec < kWPasses && !(MXW_ABL & 2)) {
        mxw_build_step(shw, p_vec - 3, p_vec & 1, tid);
        mxw_build_step(shw, p_vec - 3, p_vec & 1, tid + 256);
      }
    }
  }
#if MXW_ABL & 256
  if (blockIdx.x == 1000 && lane == 0)
    printf("wave %d: barrier %llu build %llu pass %llu epilogue %llu, loop %llu cycles\n", wave, t_bar, t_build, t_pass, t_epi,
           __builtin_readcyclecounter() - t_begin);
#endif
  __syncthreads();
  // ---- one triplet per (search, PRN, Doppler): the eight bit shifts' slots (32 lanes each) meet here --------------------------
  {
    const int which = tid >> 8, p = (tid >> 3) & 31, b = tid & 7;
    const int slot = 32 * set + p;
    const u32 *row = sh.part[b][p][which];
    u32 k = 0, t = 0;
#pragma unroll
    for (int l = 0; l < 32; l++) {
      const u32 v = row[(l + tid) & 31];
      k = v > k ? v : k;
      t += v;
    }
    const size_t idx = ((size_t)search * n_prn + slot) * n_dopp + dopp;
    if (which == 0) {   // (wave-uniform: waves 0..3; a PRN's eight bit shifts are eight adjacent lanes)
      const u32 max_val = k >> 11, fine = 8u * (2047u - (k & 2047u)) + (u32)b;
      unsigned long long key = max_val ? ((unsigned long long)max_val << 14) | (unsigned long long)(16383u - fine) : 0ull;
      key = mx_max8_u64(key);
      if (b == 0 && slot < n_prn) {
        peaks[idx].max_val = (u32)(key >> 14);
        peaks[idx].phase = key ? 16383u - (u32)(key & 16383u) : 0u;
      }
    } else {
      t += __shfl_xor(t, 1);
      t += __shfl_xor(t, 2);
      t += __shfl_xor(t, 4);
      if (b == 0 && slot < n_prn) {
        peaks[idx].sum = t;
        peaks[idx].avr = t / (u32)kSamples;
      }
    }
  }
}

void launch_acq_mxw(hipStream_t s, const uint8_t *d_if_blocks, int n_search, int stride_blocks, int n_prn, const uint32_t *d_mx_a,
                    int if_hz, int dopp_min_hz, int dopp_step_hz, int n_dopp, int use_magnitude, gpsx_peak_t *d_peaks)
{
  const int n_sets = (n_prn + 31) / 32;
  hipLaunchKernelGGL(k_acq_mxw, dim3((unsigned)(n_search * n_dopp * n_sets)), dim3(kMxThreads), 0, s, d_if_blocks, stride_blocks,
                     n_prn, d_mx_a, if_hz, dopp_min_hz, dopp_step_hz, n_dopp, use_magnitude, d_peaks);
}

}  // namespace gpsx
