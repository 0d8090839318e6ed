// ips_bitops.h -- per-lane bit arithmetic of the FLE kernels.
//
// One lane owns one HALF-BLOCK: 32 consecutive rows of a 64-row FLE block, i.e. one 32-bit half of
// each of the block's W plane words (fle-encoding.h:8338-8340: word i holds bit i of the 64
// values, value k at bit 63-k, so rows 0..31 live in the high dword and rows 32..63 in the low
// dword, both MSB-first).  Everything here is straight-line 32-bit integer code with compile-time
// trip counts so that hipcc keeps planes and values in VGPRs.
//
// The functions are __host__ __device__ so the exact code the kernels run can be unit-tested on
// the CPU build (tests/host_bitops_test.cpp); the product path never runs them on the host.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define IPS_HD __host__ __device__ __forceinline__
#else
#define IPS_HD inline
#endif

namespace ips {

// Comparison state of one half-block against a constant, MSB -> LSB (the Mlt/Meq recurrence of
// fle-encoding.h:8039-8042; Mgt of :8151-8154 is ~(Mlt|Meq) and is never materialised).
struct CmpState {
  uint32_t lt;
  uint32_t eq;
};

// One plane step.  x = this lane's 32 bits of plane i, c = all-ones if bit i of the constant is
// set else zero (wave-uniform).
IPS_HD void cmp_step(CmpState& s, uint32_t x, uint32_t c) {
  uint32_t below = ~x & c;     // constant has 1, value has 0           (one bit-select)
  s.lt |= s.eq & below;        // ... and equal so far                   (one and-or)
  s.eq &= ~(x ^ c);            // still equal after this plane           (one 3-input bitop)
}

// Final selection for Eq/Lt/Le/Gt/Ge from (lt, eq): sel_lt/sel_eq pick the terms, inv flips.
//   EQ: eq            LT: lt            LE: lt|eq
//   GT: ~(lt|eq)      GE: ~lt
IPS_HD uint32_t cmp_select(const CmpState& s, int op) {
  switch (op) {
    case 0: return s.eq;
    case 1: return s.lt;
    case 2: return s.lt | s.eq;
    case 3: return ~(s.lt | s.eq);
    default: return ~s.lt;
  }
}

// ---------------------------------------------------------------------------------------------
// The same comparisons in ONE op per plane (gfx950: v_bitop3_b32, any 3-input boolean function).
// Walking the planes LSB -> MSB, "value < constant so far" is the borrow of value - constant:
//     b_k = (~x_k & c_k) | (~(x_k ^ c_k) & b_{k-1})        b_{-1} = 0 -> x < c,  b_{-1} = ~0 -> x <= c
// so LT / LE are the final borrow with the matching start value and GT / GE its complement; EQ is
// its own one-op chain.  (The reference runs MSB -> LSB with two masks, fle-encoding.h:8039-8042;
// the result is the same predicate.)  Truth tables: operands (src0, src1, src2) <-> 0xF0, 0xCC, 0xAA.
// ---------------------------------------------------------------------------------------------
IPS_HD uint32_t borrow_step(uint32_t b, uint32_t x, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(b, x, c, 0xB2);  // (~x & c) | (~(x ^ c) & b)
#else
  return (~x & c) | (~(x ^ c) & b);
#endif
}
IPS_HD uint32_t eq_step(uint32_t eq, uint32_t x, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(eq, x, c, 0x90);  // eq & ~(x ^ c)
#else
  return eq & ~(x ^ c);
#endif
}
IPS_HD uint32_t ne_step(uint32_t ne, uint32_t x, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(ne, x, c, 0xF6);  // ne | (x ^ c)
#else
  return ne | (x ^ c);
#endif
}
// start value of the borrow chain for LT(1) / LE(2) / GT(3) / GE(4), and the final selection
IPS_HD uint32_t borrow_init(int op) { return (op == 2 || op == 3) ? ~0u : 0u; }
IPS_HD uint32_t borrow_select(uint32_t b, int op) { return op >= 3 ? ~b : b; }

IPS_HD uint32_t bitrev32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bitreverse32(v);
#else
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
  v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
  return (v >> 16) | (v << 16);
#endif
}

// ---------------------------------------------------------------------------------------------
// 32x32 bit-matrix transpose, in registers.
//
// In:  a[i] = this lane's 32 bits of plane i (i < W; rows i >= W are zero and are folded away by
//      the compiler because W and all indices are compile-time constants).
// Out: a[r] bit i = (input a[i]) bit r, i.e. a[r] is the W-bit value whose plane bits sit at bit
//      position r.  Row j of the half-block sits at bit 31-j, so value(row j) = a[31 - j].
//
// Five butterfly stages; a stage with distance J exchanges, for every pair (k, k+J), the bit
// groups selected by M: two shifts + two bit-selects (v_bfi_b32) per pair.
// ---------------------------------------------------------------------------------------------
template <int J, uint32_t M>
IPS_HD void transpose_stage(uint32_t (&a)[32]) {
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    if ((k & J) == 0) {
      uint32_t lo = a[k], hi = a[k + J];
#if defined(__HIP_DEVICE_COMPILE__)
      // the two byte-granular stages are single v_perm_b32 byte shuffles of {hi, lo}
      if (J == 16) {
        a[k] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);       // lo.b0 lo.b1 hi.b0 hi.b1
        a[k + J] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);   // lo.b2 lo.b3 hi.b2 hi.b3
        continue;
      }
      if (J == 8) {
        a[k] = __builtin_amdgcn_perm(hi, lo, 0x06020400u);       // lo.b0 hi.b0 lo.b2 hi.b2
        a[k + J] = __builtin_amdgcn_perm(hi, lo, 0x07030501u);   // lo.b1 hi.b1 lo.b3 hi.b3
        continue;
      }
#endif
      a[k] = (lo & M) | ((hi << J) & ~M);
      a[k + J] = ((lo >> J) & M) | (hi & ~M);
    }
  }
}

IPS_HD void transpose32(uint32_t (&a)[32]) {
  transpose_stage<16, 0x0000FFFFu>(a);
  transpose_stage<8, 0x00FF00FFu>(a);
  transpose_stage<4, 0x0F0F0F0Fu>(a);
  transpose_stage<2, 0x33333333u>(a);
  transpose_stage<1, 0x55555555u>(a);
}

// Narrow columns need fewer stages: with W <= R planes (R = 8 or 16) only the stages J < R are
// run, on R registers.  Every 32-bit register then behaves as 32/R independent R-bit lanes, and
// afterwards a[r] holds 32/R values side by side: lane q of a[r] is the value whose plane bits sit
// at bit position R*q + r.  Cost: 48 VALU ops for R = 8, 112 for R = 16 (256 for the full matrix).
template <int R>
IPS_HD void transpose_lanes(uint32_t (&a)[32]) {
  if (R >= 32) transpose_stage<16, 0x0000FFFFu>(a);
  if (R >= 16) transpose_stage<8, 0x00FF00FFu>(a);  // only pairs (k, k+8) with k < 8 hold data
  transpose_stage<4, 0x0F0F0F0Fu>(a);
  transpose_stage<2, 0x33333333u>(a);
  transpose_stage<1, 0x55555555u>(a);
}

template <int W>
struct LaneWidth { static constexpr int R = W <= 8 ? 8 : W <= 16 ? 16 : 32; };

// Planes -> values for one half-block.  p[0..W) in, v[j] = value of row j (j = 0..31) out.
// Row j sits at bit position 31-j.
template <int W>
IPS_HD void planes_to_values(const uint32_t (&p)[W], uint32_t (&v)[32]) {
  constexpr int R = LaneWidth<W>::R;
  uint32_t a[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = i < W ? p[i] : 0u;
  transpose_lanes<R>(a);  // rows >= R of 'a' are zero and stay untouched by the compiler
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int pos = 31 - j;
    if (R == 32) v[j] = a[pos];
    else v[j] = (a[pos % R] >> (R * (pos / R))) & ((1u << R) - 1u);
  }
}

// Planes -> the lane-packed image only (no per-row extraction): a[r], r < R, holds 32/R values side
// by side; the value of row j (bit position pos = 31 - j) is field pos / R of a[pos % R].  The
// scan's index-list path parks exactly these R registers in LDS (2 or 4 x 16 bytes per lane for
// W <= 8 / 16) and lets the selected rows be fetched as bytes / halfwords.
template <int W>
IPS_HD void planes_to_lanes(const uint32_t (&p)[W], uint32_t (&a)[32]) {
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = i < W ? p[i] : 0u;
  transpose_lanes<LaneWidth<W>::R>(a);
}

// Planes -> "quads" for the wide widths (R = 32): only the three coarse stages (16, 8, 4) of the
// 32x32 transpose -- 128 of its 256 ops.  Afterwards t[4*bh + kl] bit (4*kh + bl) = plane
// (4*kh + kl) bit (4*bh + bl): the value at bit position b = 4*bh + bl is spread over the four
// registers t[4*bh .. 4*bh+3] as a comb of every fourth bit,
//     value(b) = OR over kl of (((t[4*bh + kl] >> bl) & 0x11111111) << kl)         (11 ops),
// which the scan's index-list path evaluates for the SELECTED rows only (quads_value below): at
// 10 % selectivity that is 4 rounds of 64 rows instead of the two fine stages for all 2048.
template <int W>
IPS_HD void planes_to_quads(const uint32_t (&p)[W], uint32_t (&t)[32]) {
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i] = i < W ? p[i] : 0u;
  transpose_stage<16, 0x0000FFFFu>(t);
  transpose_stage<8, 0x00FF00FFu>(t);
  transpose_stage<4, 0x0F0F0F0Fu>(t);
}
// the remaining fine stages: quads -> values by position (a[b] = value at bit position b)
IPS_HD void quads_to_values(uint32_t (&t)[32]) {
  transpose_stage<2, 0x33333333u>(t);
  transpose_stage<1, 0x55555555u>(t);
}
// value at bit position 4*bh + bl from its four quad registers q0..q3 = t[4*bh .. 4*bh+3]
IPS_HD uint32_t quads_value(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t bl,
                            uint32_t m = 0x11111111u) {
  // three nested bit-selects (v_bfi_b32): every comb position is overwritten by the right term, so
  // the shifted registers need no masks of their own (10 ops)
  const uint32_t t0 = q0 >> bl, t1 = (q1 >> bl) << 1, t2 = (q2 >> bl) << 2, t3 = (q3 >> bl) << 3;
  const uint32_t m1 = m << 1, m2 = m << 2;
#if defined(__HIP_DEVICE_COMPILE__)
  // (the compiler would turn the portable form back into four ANDs and two ORs)
  uint32_t x23, x123, r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(x23) : "v"(m2), "v"(t2), "v"(t3));
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(x123) : "v"(m1), "v"(t1), "v"(x23));
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(t0), "v"(x123));
#else
  const uint32_t x23 = (t2 & m2) | (t3 & ~m2);
  const uint32_t x123 = (t1 & m1) | (x23 & ~m1);
  const uint32_t r = (t0 & m) | (x123 & ~m);
#endif
  return m == 0x11111111u ? r : (r & (m | m1 | m2 | (m << 3)));  // narrow lanes: drop the other lane
}
// The same for 9..16-bit columns (R = 16: two values side by side per register): only the stages
// 8 and 4 of transpose_lanes<16> (48 of 112 ops); the value at bit position 16*q + 4*bh + bl is
// quads_value(t[4*bh .. 4*bh+3], 16*q + bl, 0x1111).
template <int W>
IPS_HD void planes_to_lane_quads16(const uint32_t (&p)[W], uint32_t (&t)[32]) {
#pragma unroll
  for (int i = 0; i < 32; ++i) t[i] = i < W ? p[i] : 0u;
  transpose_stage<8, 0x00FF00FFu>(t);
  transpose_stage<4, 0x0F0F0F0Fu>(t);
}

// Values -> planes (the encoder direction): v[j] = value of row j, p[i] = plane i bits.
template <int W>
IPS_HD void values_to_planes(const uint32_t (&v)[32], uint32_t (&p)[W]) {
  constexpr int R = LaneWidth<W>::R;
  uint32_t a[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = 0u;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int pos = 31 - j;
    if (R == 32) a[pos] = v[j];
    else a[pos % R] |= (v[j] & ((1u << R) - 1u)) << (R * (pos / R));
  }
  transpose_lanes<R>(a);  // the lane-wise transpose is an involution
#pragma unroll
  for (int i = 0; i < W; ++i) p[i] = a[i];
}

// ---------------------------------------------------------------------------------------------
// Tile geometry shared by host launch code and kernels.
// A wave works on SUB-TILES of 32 blocks = 2048 rows (IPS_BATCH_ROWS).  In LDS each block's W
// plane words are stored with an odd word stride so that the 32 lanes of a ds_read group hit 32
// different banks ((block*stride) mod 16 is then a bijection over 16 consecutive blocks).
// ---------------------------------------------------------------------------------------------
constexpr int kBlocksPerTile = 32;
constexpr int kRowsPerTile = kBlocksPerTile * 64;  // 2048
constexpr int kRowTileStrideDw = 36;               // 32 values + 4 dwords pad per lane (144 B)
constexpr int kRowTileBytes = 64 * kRowTileStrideDw * 4;  // 9216
// Index list of the scan's sparse path: byte offsets (u16) of the selected rows' values inside the
// wave's LDS image, in row order; sub-tiles with more selected rows take the dense path.
constexpr int kIndexListMax = 512;
constexpr int kIndexListBytes = kIndexListMax * 2;
constexpr int kScanWaveBytes = kRowTileBytes + kIndexListBytes;  // 10240: 4 waves x 4 workgroups = 160 KiB
// lane stride (bytes) of the packed image: 32/64 payload bytes + pad so that the 16-byte stores of
// 8 consecutive lanes fall into 8 different bank quads
constexpr int packed_lane_stride(int r) { return r == 8 ? 48 : r == 16 ? 80 : 144; }

constexpr int plane_stride_words(int w) { return w | 1; }
constexpr int plane_tile_bytes(int w) { return kBlocksPerTile * plane_stride_words(w) * 8; }

// dword offset of row rho (0..2047) inside the padded row tile
IPS_HD int row_tile_dw(int rho) { return (rho >> 5) * kRowTileStrideDw + (rho & 31); }

}  // namespace ips
