// ips_rank_device.h -- the device side of the rank tiles that other kernels share: tile geometry,
// NOT-NULL root words, and the tile-count pass, which also rides on the data-predicate launch of
// the nullable leaf (PredArgs::aux_*: the first aux_blocks workgroups of that launch count the
// definition levels' tiles and exit, so the 33 MB pass costs no launch of its own).
#pragma once
#include "ips_device.h"

namespace ips {

constexpr int kRankThreads = 256;
constexpr int kRankWaves = kRankThreads / kWave;
constexpr int kRankRounds = 8;                                   // 16-byte loads per lane
constexpr int kRankWordsPerWave = kWave * 2 * kRankRounds;        // 1024 words = 65536 rows
constexpr int kRankWordsPerTile = kRankWordsPerWave * kRankWaves;  // 4096 words
// expand_kernel works on quarter tiles: a workgroup takes the 1024 words one counting WAVE covered
// (4 waves x 2 rounds), so the grid has 4x the blocks of the counting pass -- several generations
// of workgroups per CU, whose load / deposit / store phases overlap (with one generation of
// 4096-word blocks every wave of the chip was in the same phase: 38 -> 34 us for 2^28 rows)
constexpr int kExpRounds = IPS_EXP_ROUNDS;
constexpr int kExpWordsPerWave = kWave * 2 * kExpRounds;          // 256 words
constexpr int kExpWordsPerBlock = kExpWordsPerWave * kRankWaves;   // 1024 words
constexpr int kExpBlocksPerTile = kRankWordsPerTile / kExpWordsPerBlock;
static_assert(kExpBlocksPerTile >= 1 && kRankWaves % kExpBlocksPerTile == 0, "an expand block is whole counting waves");

typedef unsigned long long u64;

// Root word i as a NOT-NULL mask in bitmap order, rows >= n_rows cleared.
//   kRootBitmap: an ordinary bitmap word (LSB = first row)
//   kRootLevels1: a width-1 FLE block of definition levels with max_def_level 1 (row k at bit
//                 63-k, fle-encoding.h:8338-8340): the mask is the bit-reversed word
enum RootKind { kRootBitmap = 0, kRootLevels1 = 1 };

template <int ROOT>
__device__ __forceinline__ u64 root_mask(u64 w, int64_t word, int64_t n_rows) {
  if (ROOT == kRootLevels1) w = __builtin_bitreverse64(w);
  const int64_t valid = n_rows - word * 64;
  if (valid < 64) w = valid <= 0 ? 0ull : (w & ((1ull << valid) - 1ull));
  return w;
}

// the wave's R x 2 root words from word 'first' on (word index of (round r, lane, e) =
// first + r * 128 + 2 * lane + e), masked; words beyond n_words are zero
template <int ROOT, int R>
__device__ __forceinline__ void load_root(const u64* __restrict__ root, int64_t first,
                                          int64_t n_words, int64_t n_rows, int lane,
                                          u64 (&m)[R][2]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t w0 = first + r * 128 + 2 * lane;
    u64 a = 0, b = 0;
    if (w0 + 1 < n_words) {
      u32x4 t = stream_load<true>(reinterpret_cast<const u32x4*>(root + w0));
      a = ((u64)t.y << 32) | t.x;
      b = ((u64)t.w << 32) | t.z;
    } else if (w0 < n_words) {
      a = root[w0];
    }
    m[r][0] = root_mask<ROOT>(a, w0, n_rows);
    m[r][1] = root_mask<ROOT>(b, w0 + 1, n_rows);
  }
}

// the same for a wave whose R x 128 words are all whole words of rows: no bounds, no row masks
template <int ROOT, int R>
__device__ __forceinline__ void load_root_whole(const u64* __restrict__ root, int64_t first, int lane,
                                                u64 (&m)[R][2]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const u32x4 t = stream_load<true>(reinterpret_cast<const u32x4*>(root + first + r * 128 + 2 * lane));
    if (ROOT == kRootLevels1) {
      m[r][0] = ((u64)__builtin_bitreverse32(t.x) << 32) | __builtin_bitreverse32(t.y);
      m[r][1] = ((u64)__builtin_bitreverse32(t.z) << 32) | __builtin_bitreverse32(t.w);
    } else {
      m[r][0] = ((u64)t.y << 32) | t.x;
      m[r][1] = ((u64)t.w << 32) | t.z;
    }
  }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
  return __builtin_amdgcn_readlane(wave_inclusive_scan(x), 63);
}

// ---- deposit (pdep) through a 256-entry nibble table in LDS -----------------------------------
// pdep of the low popcount(mask) bits of src into the set positions of mask, four mask bits at a
// time: table[mask4 << 4 | src4].  Per nibble: two bit-field extracts (the second at the running
// rank of the nibble inside its 32-bit half), one table read, one shift-or, one popcount-add.
__device__ __forceinline__ uint32_t deposit32(uint32_t sh, uint32_t mh, const uint8_t* __restrict__ lut) {
  uint32_t o = 0, rank = 0;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const uint32_t m4 = (mh >> (4 * n)) & 15u;
    const uint32_t s4 = (sh >> rank) & 15u;  // rank <= 28
    o |= (uint32_t)lut[(m4 << 4) | s4] << (4 * n);
    rank += (uint32_t)__builtin_popcount(m4);
  }
  return o;
}

// entry i = (mask4 << 4 | src4) of the deposit table
constexpr uint32_t deposit_lut_entry(uint32_t i) {
  const uint32_t m = i >> 4, v = i & 15u;
  uint32_t d = 0, j = 0;
  for (uint32_t bit = 0; bit < 4; ++bit) {
    if (m & (1u << bit)) {
      if (v & (1u << j)) d |= 1u << bit;
      ++j;
    }
  }
  return d;
}

// ---- extract (pext), the inverse of deposit: table[mask4 << 4 | src4] -------------------------
constexpr uint32_t extract_lut_entry(uint32_t i) {
  const uint32_t m = i >> 4, v = i & 15u;
  uint32_t e = 0, j = 0;
  for (uint32_t bit = 0; bit < 4; ++bit) {
    if (m & (1u << bit)) {
      if (v & (1u << bit)) e |= 1u << j;
      ++j;
    }
  }
  return e;
}

// the src bits at the set positions of mask, packed from bit 0 up
__device__ __forceinline__ u64 extract64(u64 src, u64 mask, const uint8_t* __restrict__ lut) {
  u64 out = 0;
  uint32_t pos = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t mh = (uint32_t)(mask >> (32 * h)), sh = (uint32_t)(src >> (32 * h));
    uint32_t o = 0, rank = 0;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const uint32_t m4 = (mh >> (4 * n)) & 15u;
      const uint32_t s4 = (sh >> (4 * n)) & 15u;
      o |= (uint32_t)lut[(m4 << 4) | s4] << rank;  // rank <= 28, the piece has <= 4 bits
      rank += (uint32_t)__builtin_popcount(m4);
    }
    out |= (u64)o << pos;
    pos += rank;
  }
  return out;
}

// tile_counts[t] = set root bits of tile t; tile_counts[tiles + 4 t + w] = those of its wave w (the
// quarter tiles expand_kernel works on).  One workgroup of 256 threads per tile.
// ZERO: also clears the words of 'zero_out' that belong to the tile (the compress output must start
// as zeros where two waves share a word; clearing it here saves a memset launch)
template <int ROOT, bool ZERO>
__device__ __forceinline__ void rank_tile_counts_body(const u64* __restrict__ root, int64_t n_rows,
                                                      uint32_t* __restrict__ tile_counts, int64_t tile,
                                                      int64_t tiles, u64* __restrict__ zero_out) {
  __shared__ uint32_t wave_tot[kRankWaves];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t n_words = (n_rows + 63) / 64;
  const int64_t first = tile * kRankWordsPerTile + wave * kRankWordsPerWave;
  // popcounts only (bit order does not matter), accumulated load by load: the body also lives
  // inside the predicate kernels and must stay below their register count
  uint32_t c = 0;
  if ((first + kRankWordsPerWave) * 64 <= n_rows) {
#pragma unroll
    for (int r = 0; r < kRankRounds; ++r) {
      const u32x4 t = stream_load<true>(reinterpret_cast<const u32x4*>(root + first + r * 128 + 2 * lane));
      c += __builtin_popcount(t.x) + __builtin_popcount(t.y) + __builtin_popcount(t.z) + __builtin_popcount(t.w);
    }
  } else {
    for (int r = 0; r < kRankRounds; ++r) {
      const int64_t w0 = first + r * 128 + 2 * lane;
      if (w0 < n_words) c += __builtin_popcountll(root_mask<ROOT>(root[w0], w0, n_rows));
      if (w0 + 1 < n_words) c += __builtin_popcountll(root_mask<ROOT>(root[w0 + 1], w0 + 1, n_rows));
    }
  }
  if (ZERO) {
#pragma unroll
    for (int r = 0; r < kRankRounds; ++r) {
      const int64_t w0 = first + r * 128 + 2 * lane;
      if (w0 + 1 < n_words) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(zero_out + w0) = z;
      } else if (w0 < n_words) {
        zero_out[w0] = 0ull;
      }
    }
  }
  const uint32_t tot = wave_sum(c);
  if (lane == 0) {
    wave_tot[wave] = tot;
    tile_counts[tiles + tile * kRankWaves + wave] = tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) tile_counts[tile] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// the counting workgroups of a predicate launch (see PredArgs::aux_*)
__device__ __forceinline__ void rank_aux_counts(const PredArgs& args) {
  const u64* root = reinterpret_cast<const u64*>(args.aux_root);
  if (args.aux_kind == kRootLevels1)
    rank_tile_counts_body<kRootLevels1, false>(root, args.aux_rows, args.aux_counts, blockIdx.x, args.aux_blocks, nullptr);
  else
    rank_tile_counts_body<kRootBitmap, false>(root, args.aux_rows, args.aux_counts, blockIdx.x, args.aux_blocks, nullptr);
}

}  // namespace ips
