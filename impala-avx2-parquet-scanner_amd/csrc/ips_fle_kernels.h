// ips_fle_kernels.h -- the templated FLE kernels (compile-time bit width W): fused scan, late
// materialisation (select), full decode and encode.  Instantiated per W in ips_fle_*.hip.
//
// HBM roofline bounds every kernel here (there is no contraction, so no MFMA): per 2048-row
// sub-tile a wave reads 256*W encoded bytes once, and writes 256 B of bitmap plus 4 bytes per
// selected row (scan/select) or out_width bytes per row (decode).
#pragma once
#include "ips_device.h"
#include "ips_rank_device.h"
#include "ips_chunk_device.h"

// (nt loads in decode / encode pay once their output streams are nt stores as well, -3...-4 %;
// with plain stores they measured 4 % slower: IPS_DECODE_NT_LOADS / IPS_ENCODE_NT_LOADS, ips_knobs.h)

namespace ips {



// Dictionary gather applied while values leave LDS: G = 0 none (store the code / raw value),
// 4 / 8 = bytes per dictionary entry.
template <int G>
struct GatherT { using type = uint32_t; };
template <>
struct GatherT<8> { using type = uint64_t; };

// four consecutive dictionary entries of a batch (dst 16-byte aligned): one / two 16-byte stores
template <int G, typename GT>
__device__ __forceinline__ void store_entries4(GT* dst, GT a, GT b, GT c, GT d) {
  if constexpr (G == 8) {
    const u32x4 t0 = {(uint32_t)a, (uint32_t)((uint64_t)a >> 32), (uint32_t)b, (uint32_t)((uint64_t)b >> 32)};
    const u32x4 t1 = {(uint32_t)c, (uint32_t)((uint64_t)c >> 32), (uint32_t)d, (uint32_t)((uint64_t)d >> 32)};
    *reinterpret_cast<u32x4*>(dst) = t0;
    *reinterpret_cast<u32x4*>(dst + 2) = t1;
  } else {
    const u32x4 t = {(uint32_t)a, (uint32_t)b, (uint32_t)c, (uint32_t)d};
    *reinterpret_cast<u32x4*>(dst) = t;
  }
}

// A dictionary that fits 4 KiB (every code of the width addressable: 2^W entries of G bytes) is
// copied into LDS once per workgroup and the gather reads it there: a dependent global load per
// selected value is what bounded the narrow dictionary scans (w=8 IN scan + gather 107 -> 87 us).
// At 16 KiB (w=12) the per-workgroup copy and the lost occupancy cost more than they save.
// The full decode gathers EVERY row, so there the copy pays up to 32 KiB (w=12: 476 -> 2xx us).
template <int W, int G, int MAX_BYTES = 4096>
struct DictLds {
  static constexpr bool kUse = G != 0 && W <= 13 && ((1 << (W <= 13 ? W : 0)) * G) <= MAX_BYTES;
  static constexpr int kEntries = kUse ? (1 << (W <= 13 ? W : 0)) : 1;
};
constexpr int kDecodeDictLdsBytes = 32768;

// Long IN lists on narrow columns (every dictionary code width): membership becomes a 2^W-bit set
// in LDS, built once per workgroup, and each decoded value costs one LDS read -- independent of
// the list length -- instead of W bit-select steps per constant.
// Below these list lengths the K * W plane steps on registers are cheaper than decoding every row
// and looking it up (48..112 ops of transposition + 6 per row).  Crossovers measured at 2^28 rows
// at the end of round 2 (tools/ab/in_table_sweep.py), separately for the fused scan -- whose list
// variants run at 8 / 5 waves per SIMD while the table variant of w > 8 needs the large LDS layout
// -- and for the stand-alone predicate.
constexpr int in_table_min_consts(int w) { return w <= 8 ? 20 : w <= 12 ? 34 : 22; }        // fused scan
constexpr int in_table_min_consts_pred(int w) { return w <= 8 ? 20 : w <= 12 ? 22 : 14; }  // predicate only
template <int W>
struct InTable {
  static constexpr bool kUse = W <= 16;
  static constexpr int kDwords = kUse ? ((1 << (W <= 16 ? W : 0)) + 31) / 32 : 1;
};

template <int W>
__device__ __forceinline__ void in_table_build(uint32_t* table, const PredArgs& args) {
  if (args.in_table != nullptr) {  // an ips_inset: its table's first 2^W bits (members >= 2^W match no code of W bits)
    for (int i = threadIdx.x; i < InTable<W>::kDwords; i += kThreads) table[i] = args.in_table[i];
    __syncthreads();
    return;
  }
  for (int i = threadIdx.x; i < InTable<W>::kDwords; i += kThreads) table[i] = 0u;
  __syncthreads();
  for (int j = threadIdx.x; j < args.n_consts; j += kThreads) {
    const uint32_t c = args.consts[j];
    atomicOr(&table[c >> 5], 1u << (c & 31u));
  }
  __syncthreads();
}

// bit j of the result <-> row j of the half-block (LSB first)
__device__ __forceinline__ uint32_t in_table_lookup(const uint32_t* table, const uint32_t (&v)[32]) {
  uint32_t w[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) w[j] = table[v[j] >> 5];
  uint32_t bm = 0u;
#pragma unroll
  for (int j = 0; j < 32; ++j) bm |= ((w[j] >> (v[j] & 31u)) & 1u) << j;
  return bm;
}

// ---------------------------------------------------------------------------------------------
// Fused scan: predicate (or given bitmap) -> bitmap, selected rows decoded and written per batch.
// Replaces EvalSimplePredicates + bitmap->skip-list + ReadValue(skip) of one column
// (hdfs-parquet-scanner.cc:1837-1865, 1134-1181, 1006-1027; fle-encoding.h:8012-8066, 344-379).
// ---------------------------------------------------------------------------------------------
// LDS per wave.  Wide columns need the 9 KiB row tile (+ 1 KiB index list); a narrow column's plane
// image and its parked lane-packed values are far smaller, and what the scan then buys with a
// small footprint is OCCUPANCY: the fused scan is bound by VALU issue at 4 waves per SIMD (w = 8:
// 194 VALU per sub-tile = 58 % of the SIMD's cycles at 62 % of the HBM roofline), and more resident
// waves fill the gaps that LDS round trips and waitcnts leave (w = 8 @10 %: 82 -> 70 us with 4 KiB
// per wave and 8 waves per SIMD).  The small layouts have no room for the dense compaction image
// of 2048 dwords: beyond 512 selected rows they compact the lane-packed bytes / halfwords instead.
template <int W, int MODE>
struct ScanLds {
  // (w = 9..16 with 6 KiB: the 96 registers that 5 waves per SIMD leave spill, and without spills
  // the gain was 0-10 % at low selectivity against a 40 % loss above 25 %: left on the large layout)
  // IN-list scans of w = 9..16 (dictionary codes, a handful of rows selected) do take the small
  // layout, WITHOUT a dense path: beyond 512 selected rows of a sub-tile they work the index list
  // off in windows of 512 (kWindowed) -- slow there, but their registers then fit 5 waves per SIMD.
  static constexpr bool kWindowed = IPS_SCAN_SMALL_LDS && MODE == kScanInList && W > IPS_SCAN_SMALL_LDS_MAX_W && W <= 16;
  // (membership-table scans of w <= 8 as well: the decoded values are only needed for the lookup,
  // what is parked are the lane-packed bytes)
  static constexpr bool kSmall = IPS_SCAN_SMALL_LDS &&
                                 (MODE != kScanInTable ? (W <= IPS_SCAN_SMALL_LDS_MAX_W || kWindowed)
                                                       : W <= IPS_SCAN_SMALL_LDS_MAX_W);
  static constexpr int kBody = !kSmall ? kRowTileBytes : W <= 8 ? 64 * packed_lane_stride(8) : 64 * packed_lane_stride(16);
  static constexpr int kWaveBytes = kBody + kIndexListBytes;  // 4096 / 6144 / 10240
  static_assert(plane_tile_bytes(kSmall ? (W <= 8 ? 8 : 16) : 32) <= kBody, "the plane image fits the body");
  // waves per SIMD the register allocation is asked to allow (without the dense path's 32 value
  // registers the narrow scans need 46-61 (w <= 8) and 79-96 (w <= 16) VGPRs)
  static constexpr int kMinWaves = !kSmall ? IPS_MIN_WAVES_PER_EU : MODE == kScanInTable ? 6  // (32 decoded values live during the lookup)
                                   : W <= IPS_SCAN_SMALL_LDS_MAX_W ? 8 : 5;
};

// A page of a column chunk inside the chunk's row space (PAGED kernels: ips_chunk_device.h): where
// its bitmap dwords go, where a given selection is read from.
struct PageCtx {
  BitmapWindow win;
  ChunkPage page;
  int64_t total_dwords;  // dwords of the chunk-wide bitmap
};

// first_tile / stride: the sub-tiles this wave takes (tile = first_tile, first_tile + stride, ...)
// PAGED: enc / n_rows are one page of a chunk; the bitmap (written, or given) is the chunk's, the
// batch outputs start at the page's first batch slot.
template <int W, int MODE, int G, bool PAGED = false>
__device__ __forceinline__ void fle_scan_body(
    const uint64_t* __restrict__ enc, int64_t n_rows, const PredArgs& args,
    uint32_t* __restrict__ bitmap32, const uint32_t* __restrict__ given_bitmap32,
    typename GatherT<G>::type* __restrict__ batch_values, uint32_t* __restrict__ batch_counts,
    const typename GatherT<G>::type* __restrict__ dict, uint32_t dict_entries,
    int32_t* __restrict__ bad_index, int64_t first_tile, int64_t stride, const PageCtx* pc = nullptr,
    int64_t tile_end = -1) {
  constexpr int kWB = ScanLds<W, MODE>::kWaveBytes;
  constexpr bool kSmallLds = ScanLds<W, MODE>::kSmall;
  constexpr bool kWindowed = ScanLds<W, MODE>::kWindowed;
  __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerBlock * kWB / 4];
  using GT = typename GatherT<G>::type;
  __shared__ GT dict_lds[DictLds<W, G>::kEntries];
  constexpr int L = (16 * W + kWave - 1) / kWave;
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * (kWB / 4);
  if constexpr (DictLds<W, G>::kUse) {
    for (uint32_t i = threadIdx.x; i < dict_entries && i < (uint32_t)DictLds<W, G>::kEntries; i += kThreads)
      dict_lds[i] = dict[i];
    __syncthreads();
  }
  // dense sub-tiles store four gathered entries per lane at once while the dictionary sits in LDS or
  // in the L1 (<= 32 KiB); with larger ones four dependent L2 gathers per lane measured slower than
  // one (D = 16384, every row selected: 739 vs 900 us)
  const bool quad_stores = DictLds<W, G>::kUse || (uint64_t)dict_entries * G <= 32768u;
  auto lookup = [&](uint32_t code) -> GT {
    if constexpr (DictLds<W, G>::kUse) return dict_lds[code];
    else return dict[code];
  };
  constexpr bool kInTable = MODE == kScanInTable && InTable<W>::kUse;
  __shared__ uint32_t in_table[kInTable ? InTable<W>::kDwords : 1];
  if constexpr (kInTable) in_table_build<W>(in_table, args);

  // (PAGED: 'tiles' is the end of this wave's share of the page's sub-tiles)
  const int64_t tiles = PAGED ? tile_end : (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t total_words = ((n_rows + 63) / 64) * W;
  const int64_t bm_dwords = bitmap_dwords(n_rows);
  int64_t tile = first_tile;
  [[maybe_unused]] WindowCarry carry;

  // given bitmap: this lane's dword of the sub-tile, and which blocks hold a selected row
  auto given_dword = [&](int64_t t) -> uint32_t {
    if constexpr (PAGED) {
      return t < tiles ? window_fetch(given_bitmap32, pc->total_dwords, pc->page, t, lane) : 0u;
    } else {
      const int64_t gd = t * 64 + lane;
      return (t < tiles && gd < bm_dwords) ? given_bitmap32[gd] : 0u;
    }
  };
  // this lane's bitmap dword of sub-tile t (rows beyond n_rows cleared) -> the bitmap
  auto put_dword = [&](int64_t t, uint32_t dword) {
    if constexpr (PAGED) {
      window_emit(pc->win, carry, t * 64 + lane, dword, 0);
    } else {
      const int64_t dd = t * 64 + lane;
      if (dd < bm_dwords) IPS_BITMAP_STORE(bitmap32 + dd, dword);
    }
  };
  auto needed_blocks = [&](uint32_t g) -> uint64_t {
    const uint64_t any = __builtin_amdgcn_ballot_w64(g != 0u);  // bit l <-> half-block of lane l
    return any | (any >> 1);                                    // bit 2b <-> block b
  };
  u32x4 r[L];
  for (int i = 0; i < L; ++i) r[i] = u32x4{0u, 0u, 0u, 0u};
  uint32_t given_cur = 0u, given_nxt = 0u;
  if (MODE == kScanGivenBitmap) {
    given_cur = given_dword(tile);
    given_nxt = given_dword(tile + stride);
    if (tile < tiles) tile_load_needed<L, W>(enc, tile, total_words, lane, needed_blocks(given_cur), r);
  } else if (tile < tiles) {
    tile_load<L>(enc, tile, W, total_words, lane, r);
  }
  while (tile < tiles) {
    const int64_t next = tile + stride;
    if (MODE == kScanGivenBitmap) {
      // nothing selected in this sub-tile (nothing was loaded for it either): count 0, move on
      const int64_t row0 = tile * kRowsPerTile + (int64_t)lane * 32;
      const bool mine_live = given_cur != 0u && row0 < n_rows;
      if (__builtin_amdgcn_ballot_w64(mine_live) == 0ull) {
        if (next < tiles) tile_load_needed<L, W>(enc, next, total_words, lane, needed_blocks(given_nxt), r);
        given_cur = given_nxt;
        given_nxt = given_dword(next + stride);
        if (lane == 0) batch_counts[tile] = 0u;
        tile = next;
        continue;
      }
    }
    tile_to_lds<L>(lds32, W, lane, r);
    uint32_t given_nn = 0u;
    if (MODE == kScanGivenBitmap) {
      if (next < tiles) tile_load_needed<L, W>(enc, next, total_words, lane, needed_blocks(given_nxt), r);
      given_nn = given_dword(next + stride);  // bitmap prefetch, two sub-tiles ahead
    } else if (next < tiles) {
      tile_load<L>(enc, next, W, total_words, lane, r);  // register prefetch
    }
    wave_lds_fence();

    [[maybe_unused]] const int64_t d = tile * 64 + lane;
    uint32_t bm;
    if (MODE == kScanInList && W > 16) {  // wide IN: K passes over the planes in LDS, before they enter VGPRs
      bm = finish_bitmap_dword(pred_from_lds(lds32, W, lane, args), tile, lane, n_rows);
      put_dword(tile, bm);
    }
    uint32_t p[W];
    planes_from_lds<W>(lds32, lane, p);
    uint32_t v[32];
    if (kInTable) {  // long list: decode first, one set lookup per value
      planes_to_values<W>(p, v);
      bm = finish_bitmap_dword(bitrev32(in_table_lookup(in_table, v)), tile, lane, n_rows);
      put_dword(tile, bm);
    }
    if (MODE == kScanInList && W <= 16) {
      bm = finish_bitmap_dword(pred_in_from_regs<W>(p, args.consts, args.n_consts), tile, lane, n_rows);
      put_dword(tile, bm);
    }
    if (MODE == kScanPredicate) {
      const uint32_t sel = pred_from_regs<W>(p, args);
      if (!PAGED && (tile + 1) * kRowsPerTile <= n_rows) {  // wave-uniform: every row of the sub-tile exists
        bm = bitrev32(sel);
        IPS_BITMAP_STORE(bitmap32 + d, bm);
      } else {
        bm = finish_bitmap_dword(sel, tile, lane, n_rows);
        put_dword(tile, bm);
      }
    } else if (MODE == kScanGivenBitmap) {
      bm = bitrev32(given_cur);  // finish_bitmap_dword reverses back; only the row mask is wanted
      bm = finish_bitmap_dword(bm, tile, lane, n_rows);
      given_cur = given_nxt;
      given_nxt = given_nn;
    }

    uint32_t count = 0;
    if (IPS_ABLATE != 2 && __builtin_amdgcn_ballot_w64(bm != 0u) != 0ull) {  // wave-uniform: any row selected
      const uint32_t mine = (uint32_t)__builtin_popcount(bm);
      const uint32_t incl = wave_inclusive_scan(mine);
      count = __builtin_amdgcn_readlane(incl, 63);
      uint32_t P = incl - mine;  // this lane's first output slot inside the batch
      GT* dst = batch_values + tile * kRowsPerTile;
      // Narrow columns, few selected rows per lane: pick the W bits of each selected row straight
      // out of the plane registers (2 ops per plane and row) instead of transposing all 32 rows
      // and going through the row tile -- a narrow sub-tile is only 256*W bytes of HBM time, so
      // the transposes and the walk are what bounds it.
      constexpr uint32_t kGatherLaneMax = W <= 4 ? IPS_GATHER_MAX_4 : W <= 8 ? IPS_GATHER_MAX_8 : W <= 12 ? IPS_GATHER_MAX_12
                                        : W <= 16 ? IPS_GATHER_MAX_16 : IPS_GATHER_WIDE;
      if (!kInTable && kGatherLaneMax != 0 &&
          __builtin_amdgcn_ballot_w64(mine > kGatherLaneMax) == 0ull) {
        uint32_t m = bm;
        int bad = 0;
#pragma unroll 1
        for (uint32_t round = 0; round < kGatherLaneMax; ++round) {
          if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) break;
          const bool ok = m != 0u;
          const uint32_t sh = 31u - (ok ? (uint32_t)__builtin_ctz(m) : 0u);  // row j sits at bit 31-j
          m &= m - 1u;
          uint32_t val = 0u;
#pragma unroll
          for (int k = 0; k < W; ++k) val |= ((p[k] >> sh) & 1u) << k;
          if (ok) {
            if (G == 0) {
              dst[P] = (GT)val;
            } else if (val < dict_entries) {
              dst[P] = lookup(val);
            } else {
              bad = 1;
            }
            ++P;
          }
        }
        if (G != 0 && bad && bad_index) *bad_index = 1;
        if (lane == 0) batch_counts[tile] = count;
        wave_lds_fence();  // LDS region is reused by the next sub-tile
        tile = next;
        continue;
      }
      constexpr int R = LaneWidth<W>::R;
      constexpr bool kPacked = R < 32;  // values stay lane-packed (table mode too: its decoded dwords are only for the lookup)
      constexpr bool kQuads = IPS_QUADS && !kInTable && R == 32;  // half-transposed in LDS
      constexpr bool kQuads16 = IPS_QUADS16 && kPacked && !kInTable && R == 16;
      const bool index_path = kWindowed || count <= (uint32_t)kIndexListMax;
      if (index_path) {
        // Index-list path (up to 25 % selectivity).  The lane parks its 32 values in LDS -- as
        // bytes / halfwords for W <= 8 / 16 (the lane-packed registers as they are: 2 / 4 x 16
        // bytes), as dwords above -- and appends the LDS byte offset of each of ITS selected rows
        // to the sub-tile's index list at its prefix position (one ds_write_b16 per selected row:
        // the only per-lane, imbalanced loop, a handful of instructions per round).  The list is
        // then consumed 64 entries per round by all lanes: entry -> value -> batch slot, so the
        // stores to HBM are full 256-byte rows of consecutive lanes whatever the bitmap looks like.
        uint8_t* lds8 = reinterpret_cast<uint8_t*>(lds32);
        constexpr int kStride = packed_lane_stride(kPacked ? R : 32);
        if (kPacked) {
          uint32_t a[32];
          if (kQuads16) planes_to_lane_quads16<W>(p, a); else planes_to_lanes<W>(p, a);
          wave_lds_fence();  // all plane reads precede the overwrite of the same LDS region
#pragma unroll
          for (int i = 0; i < R / 4; ++i) {
            u32x4 t = {a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
            *reinterpret_cast<u32x4*>(lds8 + lane * kStride + 16 * i) = t;
          }
        } else if (kQuads) {
          // wide columns: park the half-transposed "quads" (ips_bitops.h); phase B finishes the
          // transposition for the selected rows only
          uint32_t t[32];
          planes_to_quads<W>(p, t);
          wave_lds_fence();
          values_to_row_tile(lds32, lane, t);
        } else {
          if (!kInTable) planes_to_values<W>(p, v);
          wave_lds_fence();
          values_to_row_tile(lds32, lane, v);
        }
        // phase A: one list entry (lane << 5 | row) per selected row.  Every lane runs the same
        // straight-line round -- lowest set bit, store under the exec mask, clear it -- until no
        // lane has a bit left; finished lanes keep clearing zero.
        uint16_t* list = reinterpret_cast<uint16_t*>(lds8 + (kWB - kIndexListBytes));
        const uint32_t lane5 = (uint32_t)lane << 5;
        int bad = 0;
        auto value_at = [&](uint32_t e) -> uint32_t {
          const uint32_t src = e >> 5, j = e & 31u;
          if (kQuads) {
            const uint32_t b = 31u - j;  // row j sits at bit position 31 - j
            const u32x4 q = *reinterpret_cast<const u32x4*>(lds32 + src * kRowTileStrideDw + (b & ~3u));
            return quads_value(q.x, q.y, q.z, q.w, b & 3u);
          } else if (!kPacked) {
            return lds32[src * kRowTileStrideDw + j];
          } else if (kQuads16) {
            const uint32_t b = 31u - j;
            const u32x4 q = *reinterpret_cast<const u32x4*>(lds8 + src * kStride + 4u * (b & 12u));
            return quads_value(q.x, q.y, q.z, q.w, (b & 16u) | (b & 3u), 0x1111u);
          } else if (R == 16) {
            return *reinterpret_cast<const uint16_t*>(lds8 + src * kStride + 4u * ((31u - j) & 15u) + 2u * ((31u - j) >> 4));
          } else {
            return lds8[src * kStride + 4u * ((31u - j) & 7u) + ((31u - j) >> 3)];
          }
        };
        auto put = [&](uint32_t i, uint32_t x) {
          if (IPS_ABLATE == 4) {  // dev: phase B without its value stores
            if (x == 0xFFFFFFFFu && i == 0x7FFFFFFFu) dst[0] = (GT)x;
          } else if (G == 0) {
            if (IPS_NT_VALUE_STORE) __builtin_nontemporal_store((GT)x, dst + i);
            else dst[i] = (GT)x;
          } else if (x < dict_entries) {
            if (IPS_NT_VALUE_STORE) __builtin_nontemporal_store(lookup(x), dst + i);
            else dst[i] = lookup(x);
          } else {
            bad = 1;
          }
        };
        constexpr uint32_t kGroup = IPS_PHASE_B_GROUP;
        // phase B over the entries [0, n_win) of the list = rows win0.. of the sub-tile's selection
        auto phase_b = [&](uint32_t win0, uint32_t n_win) {
          for (uint32_t g0 = 0; g0 < n_win; g0 += kGroup * kWave) {  // wave-uniform
            uint32_t e[kGroup], x[kGroup];
#pragma unroll
            for (uint32_t k = 0; k < kGroup; ++k) {
              const uint32_t i = g0 + k * kWave + lane;
              e[k] = list[i < n_win ? i : 0u];
            }
#pragma unroll
            for (uint32_t k = 0; k < kGroup; ++k) x[k] = value_at(e[k]);
#pragma unroll
            for (uint32_t k = 0; k < kGroup; ++k) {
              const uint32_t i = g0 + k * kWave + lane;
              if (i < n_win) put(win0 + i, x[k]);
            }
          }
        };
        if (!kWindowed || count <= (uint32_t)kIndexListMax) {  // wave-uniform: the whole list fits
          if (IPS_ABLATE != 1) {
            uint32_t m = bm;
            uint16_t* slot = list + P;
            // trip count = the wave's largest popcount (one DPP max): scalar loop control, nothing
            // waits for a ballot per round
            const uint32_t trips = wave_max(mine);
            for (uint32_t t = 0; t < trips; ++t) {
              if (m != 0u) *slot = (uint16_t)(lane5 | (uint32_t)__builtin_ctz(m));
              ++slot;
              m &= m - 1u;
            }
          }
          wave_lds_fence();
          phase_b(0u, (IPS_ABLATE == 1 || IPS_ABLATE == 3) ? 0u : count);
        } else {
          // windows of 512 entries, every lane resuming where it stopped
          uint32_t m = bm, pos = P;
          for (uint32_t win0 = 0; win0 < count; win0 += (uint32_t)kIndexListMax) {
            const uint32_t win1 = win0 + (uint32_t)kIndexListMax;
            while (__builtin_amdgcn_ballot_w64(m != 0u && pos < win1) != 0ull) {
              if (m != 0u && pos < win1) {
                list[pos - win0] = (uint16_t)(lane5 | (uint32_t)__builtin_ctz(m));
                m &= m - 1u;
                ++pos;
              }
            }
            wave_lds_fence();
            phase_b(win0, count - win0 < (uint32_t)kIndexListMax ? count - win0 : (uint32_t)kIndexListMax);
            wave_lds_fence();  // the list is rewritten by the next window
          }
        }
        if (G != 0 && bad && bad_index) *bad_index = 1;
      } else if constexpr (kSmallLds && !kWindowed) {
        // Dense path of the small layouts (more than 512 selected rows): the lane-packed values
        // (bytes for W <= 8, halfwords for W <= 16) are compacted as they are -- element e of the
        // sub-tile's selection at byte e * EB of the region -- and leave four per lane and round.
        constexpr int EB = R / 8;  // bytes per element: 1 / 2
        uint32_t a[32];
        planes_to_lanes<W>(p, a);
        wave_lds_fence();  // all plane reads precede the overwrite of the same LDS region
        uint8_t* img = reinterpret_cast<uint8_t*>(lds32);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int pos = 31 - j;
          const uint32_t field = a[pos % R] >> (R * (pos / R));
          const uint32_t e = P + (uint32_t)__builtin_popcount(bm & ((1u << j) - 1u));
          if ((bm >> j) & 1u) {
            if (EB == 1) img[e] = (uint8_t)field;
            else *reinterpret_cast<uint16_t*>(img + 2 * e) = (uint16_t)field;
          }
        }
        wave_lds_fence();
        int bad = 0;
        for (uint32_t e0 = 4u * lane; e0 < count; e0 += 4u * kWave) {
          uint32_t x[4];
          if (EB == 1) {
            const uint32_t w4 = *reinterpret_cast<const uint32_t*>(img + e0);
            x[0] = w4 & 0xFFu; x[1] = (w4 >> 8) & 0xFFu; x[2] = (w4 >> 16) & 0xFFu; x[3] = w4 >> 24;
          } else {
            const uint32_t lo = *reinterpret_cast<const uint32_t*>(img + 2 * e0);
            const uint32_t hi = *reinterpret_cast<const uint32_t*>(img + 2 * e0 + 4);
            x[0] = lo & 0xFFFFu; x[1] = lo >> 16; x[2] = hi & 0xFFFFu; x[3] = hi >> 16;
          }
          if (G == 0 && e0 + 3 < count) {
            const u32x4 t = {x[0], x[1], x[2], x[3]};
            *reinterpret_cast<u32x4*>(reinterpret_cast<uint32_t*>(dst) + e0) = t;  // batch slots are 8 KiB aligned
          } else if (G != 0 && quad_stores && e0 + 3 < count && (x[0] | x[1] | x[2] | x[3]) < dict_entries) {
            store_entries4<G>(dst + e0, lookup(x[0]), lookup(x[1]), lookup(x[2]), lookup(x[3]));
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (e0 + k < count) {
                if (G == 0) dst[e0 + k] = (GT)x[k];
                else if (x[k] < dict_entries) dst[e0 + k] = lookup(x[k]);
                else bad = 1;
              }
            }
          }
        }
        if (G != 0 && bad && bad_index) *bad_index = 1;
      } else {
        if (!kInTable) planes_to_values<W>(p, v);
        wave_lds_fence();
        compact_lane_values(lds32, bm, P, v);
        wave_lds_fence();
        if (G == 0) {
          store_compacted(lds32, count, reinterpret_cast<uint32_t*>(dst), lane);
        } else {
          int bad = 0;
          if (quad_stores) {  // wave-uniform
            for (uint32_t e0 = 4u * lane; e0 < count; e0 += 4u * kWave) {  // four entries per lane: whole 16-byte stores
              const uint32_t* src = lds32 + compact_dw(e0);              // (4 elements never straddle a pad)
              if (e0 + 3 < count && (src[0] | src[1] | src[2] | src[3]) < dict_entries) {
                store_entries4<G>(dst + e0, lookup(src[0]), lookup(src[1]), lookup(src[2]), lookup(src[3]));
              } else {
                for (uint32_t e = e0; e < count && e < e0 + 4; ++e) {
                  const uint32_t code = src[e - e0];
                  if (code < dict_entries) dst[e] = lookup(code); else bad = 1;
                }
              }
            }
          } else {
            for (uint32_t e = lane; e < count; e += kWave) {
              const uint32_t code = lds32[compact_dw(e)];
              if (code < dict_entries) dst[e] = lookup(code); else bad = 1;
            }
          }
          if (bad && bad_index) *bad_index = 1;
        }
      }
    }
    if (lane == 0) batch_counts[tile] = count;
    wave_lds_fence();  // LDS region is reused by the next sub-tile
    tile = next;
  }
  if constexpr (PAGED && MODE != kScanGivenBitmap) window_flush(pc->win, carry, 0);
}

template <int W, int MODE, int G>
__global__ __launch_bounds__(kThreads, (ScanLds<W, MODE>::kMinWaves)) void fle_scan_kernel(
    const uint64_t* __restrict__ enc, int64_t n_rows, PredArgs args,
    uint32_t* __restrict__ bitmap32, const uint32_t* __restrict__ given_bitmap32,
    typename GatherT<G>::type* __restrict__ batch_values, uint32_t* __restrict__ batch_counts,
    const typename GatherT<G>::type* __restrict__ dict, uint32_t dict_entries,
    int32_t* __restrict__ bad_index) {
  fle_scan_body<W, MODE, G>(enc, n_rows, args, bitmap32, given_bitmap32, batch_values, batch_counts,
                            dict, dict_entries, bad_index,
                            (int64_t)blockIdx.x * kWavesPerBlock + wave_id(),
                            (int64_t)gridDim.x * kWavesPerBlock);
}

// Many data pages (separate buffers) in ONE launch: blockIdx.y picks the page, blockIdx.x the
// share of its sub-tiles.  A scanner holds a column chunk as a list of pages (ReadDataPage /
// InitDataPage per page, hdfs-parquet-scanner.cc:882-916; row groups and pages are independent,
// :1056-1060): a 2^20-row page is 0.8 us of HBM time, far below a launch.
struct PageScan {
  const uint64_t* enc;
  int64_t n_rows;
  uint32_t* bitmap32;
  uint32_t* batch_values;
  uint32_t* batch_counts;
};
constexpr int kPagesPerLaunch = 64;  // 64 descriptors + PredArgs stay below the 4 KiB kernarg limit
struct PageBatch { PageScan pages[kPagesPerLaunch]; };

// The descriptors travel in the kernel argument itself (no page table in device memory, nothing to
// copy or synchronise, capturable in a hipGraph); the page's row is read through the
// constant-address-space kernarg pointer -- indexing the by-value argument with blockIdx.y would
// make the compiler copy all of it to scratch.
template <int W>
__global__ __launch_bounds__(kThreads, (ScanLds<W, kScanPredicate>::kMinWaves)) void fle_scan_pages_kernel(
    PageBatch batch, PredArgs args) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)batch;
  typedef __attribute__((address_space(4))) const PageScan* KargPages;
  KargPages pages = (KargPages)__builtin_amdgcn_kernarg_segment_ptr();  // 'batch' is argument 0
  const PageScan pg = pages[blockIdx.y];
  fle_scan_body<W, kScanPredicate, 0>(pg.enc, pg.n_rows, args, pg.bitmap32, nullptr,
                                      pg.batch_values, pg.batch_counts, nullptr, 0u, nullptr,
                                      (int64_t)blockIdx.x * kWavesPerBlock + wave_id(),
                                      (int64_t)gridDim.x * kWavesPerBlock);
#endif
}

// The fused scan (and, MODE = kScanGivenBitmap, the late materialisation against a selection) over
// the pages of a column chunk: blockIdx.y = page, the bitmap is the chunk's, batch slot
// page.batch0 + sub-tile holds the sub-tile's selected values (the chunk's batch list is the
// concatenation of the pages' batches: a page's last batch is partial).
template <int W, int MODE, int G>
__global__ __launch_bounds__(kThreads, (ScanLds<W, MODE>::kMinWaves)) void fle_scan_chunk_kernel(
    const ChunkPage* __restrict__ pages, int64_t chunk_rows, PredArgs args, uint32_t* __restrict__ bitmap32,
    const uint32_t* __restrict__ given_bitmap32, typename GatherT<G>::type* __restrict__ batch_values,
    uint32_t* __restrict__ batch_counts, const typename GatherT<G>::type* __restrict__ dict,
    uint32_t dict_entries, int32_t* __restrict__ bad_index) {
  PageCtx pc;
  pc.page = pages[blockIdx.y];
  pc.win = bitmap_window(bitmap32, pc.page, chunk_rows, args.done, MODE == kScanGivenBitmap ? nullptr : args.edges);
  pc.total_dwords = bitmap_dwords(chunk_rows);
  const TileShare sh = tile_share(pc.win, (pc.page.n_data + kRowsPerTile - 1) / kRowsPerTile,
                                  (int64_t)blockIdx.x * kWavesPerBlock + wave_id(), (int64_t)gridDim.x * kWavesPerBlock);
  fle_scan_body<W, MODE, G, true>(pc.page.data, pc.page.n_data, args, nullptr, given_bitmap32,
                                  batch_values + (int64_t)pc.page.batch0 * kRowsPerTile,
                                  batch_counts + pc.page.batch0, dict, dict_entries, bad_index, sh.first, sh.step,
                                  &pc, sh.end);
  page_done(args.done, args.done_page0, args.done_epoch);
}

// ---------------------------------------------------------------------------------------------
// Predicate only: FleDecoder::Eq/Lt/Le/Gt/Ge/In on the encoded planes (fle-encoding.h:7962-8313).
// Nothing is decoded: per 64 rows the wave reads W words and writes one.  args.combine and-s /
// or-s the result into the existing bitmap (a conjunct chain touches the bitmap once per conjunct
// instead of once more for the AND), args.join evaluates a second comparison on the same column
// in the same pass (BETWEEN).
// ---------------------------------------------------------------------------------------------
enum PredKind { kPredSingle = 0, kPredPair = 1, kPredInList = 2, kPredInTable = 3 };

template <int W, int KIND, bool PAGED>
__device__ __forceinline__ void fle_pred_body(const uint64_t* __restrict__ enc, int64_t n_rows, const PredArgs& args,
                                              uint32_t* __restrict__ bitmap32, const BitmapWindow* win) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerBlock * plane_tile_bytes(W) / 4];
  constexpr int L = (16 * W + kWave - 1) / kWave;
  if (!PAGED && (int)blockIdx.x < args.aux_blocks) {  // counting workgroups of a nullable leaf
    rank_aux_counts(args);
    return;
  }
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * (plane_tile_bytes(W) / 4);

  int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t total_words = ((n_rows + 63) / 64) * W;
  int64_t stride = (int64_t)((int)gridDim.x - args.aux_blocks) * kWavesPerBlock;
  const int64_t bm_dwords = bitmap_dwords(n_rows);
  int64_t tile = (int64_t)((int)blockIdx.x - args.aux_blocks) * kWavesPerBlock + wave;
  [[maybe_unused]] WindowCarry carry;
  if constexpr (PAGED) {  // this wave's share of the page's sub-tiles
    // (longer shares for narrow columns -- 32 sub-tiles at w <= 8, 16 at w <= 16 -- left the unaligned
    // Q6 plan where it was, 446 us: what the contiguous shares cost in streaming order they save in atomics)
    const TileShare sh = tile_share(*win, tiles, tile, stride);
    tile = sh.first;
    stride = sh.step;
    tiles = sh.end;
  }
  constexpr bool kInTable = KIND == kPredInTable && InTable<W>::kUse;
  __shared__ uint32_t in_table[kInTable ? InTable<W>::kDwords : 1];
  if constexpr (kInTable) in_table_build<W>(in_table, args);

  // the rows of the sub-tile in the wave's plane image that satisfy the predicate (MSB first per lane)
  auto evaluate = [&]() -> uint32_t {
    if (KIND == kPredSingle) {
      uint32_t p[W];
      planes_from_lds<W>(lds32, lane, p);
      return pred_from_regs<W>(p, args);
    } else if (KIND == kPredPair) {
      uint32_t r1, r2;
      pred_pair_from_lds(lds32, W, lane, args.op, args.consts[0], args.op2, args.const2, &r1, &r2);
      return args.join == 1 ? (r1 & r2) : (r1 | r2);
    } else if (kInTable) {
      uint32_t p[W];
      planes_from_lds<W>(lds32, lane, p);
      uint32_t v[32];
      planes_to_values<W>(p, v);
      return bitrev32(in_table_lookup(in_table, v));
    } else if (W <= 16) {
      uint32_t p[W];
      planes_from_lds<W>(lds32, lane, p);
      return pred_in_from_regs<W>(p, args.consts, args.n_consts);
    } else {
      return args.in_list ? pred_in_from_lds(lds32, W, lane, args.in_list, args.in_list_n)
                          : pred_in_from_lds(lds32, W, lane, args.consts, args.n_consts);
    }
  };

  // A page that starts inside a bitmap dword (and has no edge slots): STRIPES of 62 whole dwords of the chunk's
  // bitmap = 1984 rows.  Stripe t covers page rows [1984 t - shift, 1984 (t + 1) - shift): whatever the shift,
  // they lie inside the 32 blocks from block max(31 t - 1, 0) on, so the wave still evaluates one ordinary
  // sub-tile (consecutive stripes re-read one block) and moves its result dwords into place with two DPP moves
  // and a funnel shift.  Every dword but the page's first and last belongs to this wave alone: plain stores /
  // read-modify-writes, no edge slots, no fix-up launch; the two end dwords are shared with the neighbouring
  // pages and merged with masked atomics (window_put).  Against the edge-slot version (kept for the fused
  // scans, whose batches are sub-tiles, and the early-pruning w = 32 predicate): Q6 over pages cut differently
  // per column, three launches, 417-429 -> 388 us (the w = 12 BETWEEN launch 171 -> 155 us, three fix-up
  // launches of 5-14 us gone) -> 367 us with the AND-into operand fetched ahead (w = 6: 116 -> 102 us, w = 4:
  // 105 -> 92).
  if constexpr (PAGED) {
    if (win->shift != 0u && win->edges == nullptr) {  // wave-uniform
      constexpr int kRun = 62;
      const uint32_t s = win->shift, rs = 32u - s;
      const int64_t n_blocks = (n_rows + 63) / 64;
      const int64_t stripes = (n_rows + s + kRun * 32 - 1) / (kRun * 32);
      const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
      int64_t t = (int64_t)blockIdx.x * kWavesPerBlock + wave;
      auto load_stripe = [&](int64_t st, u32x4 (&rr)[L]) {
        const int64_t b0 = st == 0 ? 0 : 31 * st - 1;
        int64_t left = (n_blocks - b0) * W;  // words of the page from that block on
        left = left < 0 ? 0 : (left > kBlocksPerTile * W ? kBlocksPerTile * W : left);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint64_t*>(left > 0 ? enc + b0 * W : enc), 0, (int)(left * 8), kBufferRsrcDword3);
#pragma unroll
        for (int i = 0; i < L; ++i) rr[i] = buffer_load16<true>(rsrc, (uint32_t)(i * kWave + lane) * 16u);
      };
      // the dword an AND-into / OR-into launch combines with, fetched with the stripe's planes (fetched where it is
      // used it put a memory round trip at the end of every stripe: the narrow AND-into launches of a plan)
      auto operand = [&](int64_t st) -> uint32_t {
        const int64_t d = st * kRun + lane, first = d * 32 - (int64_t)s;
        return (args.combine != 0 && lane < kRun && first < n_rows) ? win->base[d] : 0u;
      };
      u32x4 rr[L];
      uint32_t old = 0u;
      if (t < stripes) {
        old = operand(t);
        load_stripe(t, rr);
      }
      while (t < stripes) {
        tile_to_lds<L>(lds32, W, lane, rr);
        const int64_t next = t + n_waves;
        const uint32_t old_now = old;
        if (next < stripes) {
          old = operand(next);
          load_stripe(next, rr);
        }
        wave_lds_fence();
        const uint32_t rows = bitrev32(evaluate());  // bit j <-> row 32 lane + j of the sub-tile
        uint32_t out;
        if (t == 0) {  // the page's first rows sit 'shift' bits into the first dword
          const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x138, 0xF, 0xF, true);  // wave_shr:1
          out = __builtin_amdgcn_alignbit(rows, below, rs);
        } else {       // the stripe starts 64 - shift rows into block 31 t - 1
          const uint32_t a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x130, 0xF, 0xF, true);  // wave_shl:1
          const uint32_t a2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a1, 0x130, 0xF, 0xF, true);
          out = __builtin_amdgcn_alignbit(a2, a1, rs);
        }
        if (lane < kRun) {
          const int64_t d = t * kRun + lane;            // dword of the window
          const int64_t first = d * 32 - (int64_t)s;    // page row of its bit 0
          uint32_t mask = ~0u;
          if (first < 0) mask &= ~0u << (uint32_t)(-first);
          const int64_t valid = n_rows - first;
          if (valid < 32) mask &= valid <= 0 ? 0u : ((1u << valid) - 1u);
          const uint32_t val = out & mask;  // (rows behind the page's last one evaluate to anything)
          const bool tail = win->own_tail && d == win->tail_dword && mask != 0u;  // zeros behind the chunk's last row
          if (tail) mask |= win->tail_mask;
          window_put(win->base + d, val, mask, args.combine, 0u, args.combine != 0, old_now);
          if (tail && win->tail_extra && args.combine != 2) window_store(win->base + d + 1, 0u, 0u);
        }
        wave_lds_fence();
        t = next;
      }
      return;
    }
  }

  // The dword an AND-into / OR-into launch combines with travels with the register prefetch of its
  // sub-tile.  (Loaded where it is used -- after the predicate -- it put a whole memory round trip,
  // and a vmcnt(0) that also waited out the prefetched planes, at the end of every sub-tile.)
  auto combine_operand = [&](int64_t t) -> uint32_t {
    const int64_t d = t * 64 + lane;
    if constexpr (PAGED) return window_has_operand(*win, args.combine) ? window_operand(*win, d) : 0u;
    return (args.combine != 0 && d < bm_dwords) ? bitmap32[d] : 0u;
  };
  u32x4 r[L];
  uint32_t old = 0u;
  if (tile < tiles) {
    old = combine_operand(tile);
    tile_load<L>(enc, tile, W, total_words, lane, r);
  }
  while (tile < tiles) {
    tile_to_lds<L>(lds32, W, lane, r);
    const int64_t next = tile + stride;
    const uint32_t old_now = old;
    if (next < tiles) {  // register prefetch
      old = combine_operand(next);
      tile_load<L>(enc, next, W, total_words, lane, r);
    }
    wave_lds_fence();
    const uint32_t sel = evaluate();
    uint32_t bm = finish_bitmap_dword(sel, tile, lane, n_rows);
    const int64_t d = tile * 64 + lane;
    if constexpr (PAGED) {
      window_emit(*win, carry, d, bm, args.combine, kWave - 1, window_has_operand(*win, args.combine), old_now);
    } else if (d < bm_dwords) {
      if (args.combine == 1) bm &= old_now;
      else if (args.combine == 2) bm |= old_now;
      IPS_BITMAP_STORE(bitmap32 + d, bm);
    }
    wave_lds_fence();  // LDS region is reused by the next sub-tile
    tile = next;
  }
  if constexpr (PAGED) window_flush(*win, carry, args.combine);
}

template <int W, int KIND>
__global__ __launch_bounds__(kThreads, IPS_MIN_WAVES_PER_EU) void fle_pred_w_kernel(
    const uint64_t* __restrict__ enc, int64_t n_rows, PredArgs args,
    uint32_t* __restrict__ bitmap32) {
  fle_pred_body<W, KIND, false>(enc, n_rows, args, bitmap32, nullptr);
}

// The same over the pages of a column chunk (blockIdx.y = page; ips_chunk_device.h): every page in
// its own block geometry, the result at the page's row offset of the chunk-wide bitmap.
template <int W, int KIND>
__global__ __launch_bounds__(kThreads, IPS_MIN_WAVES_PER_EU) void fle_pred_pages_kernel(
    const ChunkPage* __restrict__ pages, int64_t chunk_rows, PredArgs args, uint32_t* __restrict__ bitmap32) {
  const ChunkPage pg = pages[blockIdx.y];
  const BitmapWindow win = bitmap_window(bitmap32, pg, chunk_rows, args.done, args.edges);
  fle_pred_body<W, KIND, true>(pg.data, pg.n_data, args, nullptr, &win);
  page_done(args.done, args.done_page0, args.done_epoch);
}

// ---------------------------------------------------------------------------------------------
// The nullable leaf in one pass (ColumnReader's nullable branch, hdfs-parquet-scanner.cc:326-345:
// predicate over the data rows, then IntersectBitset into the NOT-NULL positions).  A workgroup
// owns a quarter rank tile of OUTPUT words (1024 words, a wave 256): it forms the rank of its
// words from the tile counts (as expand_kernel does), evaluates the data predicate on exactly the
// data blocks that hold rows [rank, rank + NOT-NULL rows of the wave) -- block granular, at most
// nine sub-tiles -- into a bitmap segment in LDS, and deposits that segment into the NOT-NULL
// positions.  The data-row bitmap never exists in HBM (the three-launch route writes and re-reads
// it: 60 MB of 520 on a 2^28-row column) and the levels are read twice instead of three times.
// Long IN lists on codes of <= 16 bits go through the 2^w-bit membership table, built per workgroup.
// args.aux_root / aux_kind / aux_rows / aux_counts: the NOT-NULL root, its kind, the row count
// and the tile counts (complete before this launch); args.combine: 0 store, 1 and, 2 or into out.
// ---------------------------------------------------------------------------------------------
constexpr int kLeafSubTiles = (63 + kExpWordsPerWave * 64 + 2047) / 2048;  // 9
constexpr int kLeafSegDwords = kLeafSubTiles * 64;
static_assert(kThreads == kRankThreads, "a leaf workgroup is an expand workgroup");

// PAGED: the same per page of a column chunk (blockIdx.y = page): the page's levels, data blocks and
// tile counts (args.aux_counts + page.rank0), the result deposited at the page's row offset of the
// chunk-wide bitmap 'out'.
template <int W, int KIND, bool PAGED>
__device__ __forceinline__ void fle_leaf_body(const uint64_t* __restrict__ enc, int64_t n_sub, const PredArgs& args,
                                              u64* __restrict__ out, const BitmapWindow* win,
                                              const u64* __restrict__ root, int root_kind, int64_t n_rows,
                                              const uint32_t* __restrict__ tile_counts) {
  constexpr bool kInTable = KIND == kPredInTable && InTable<W>::kUse;
  static_assert(KIND != kPredInTable || InTable<W>::kUse, "the membership table holds codes of <= 16 bits");
  __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerBlock * plane_tile_bytes(W) / 4];
  __shared__ uint32_t seg_all[kWavesPerBlock][kLeafSegDwords];
  __shared__ uint32_t in_table[kInTable ? InTable<W>::kDwords : 1];
  if constexpr (kInTable) in_table_build<W>(in_table, args);
  __shared__ uint8_t lut[256];
  __shared__ u64 part[kRankWaves];
  __shared__ uint32_t wave_tot[kRankWaves];
  constexpr int L = (16 * W + kWave - 1) / kWave;
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * (plane_tile_bytes(W) / 4);
  uint32_t* seg = seg_all[wave];
  lut[threadIdx.x] = (uint8_t)deposit_lut_entry(threadIdx.x);

  const int64_t n_words = (n_rows + 63) / 64;
  const int64_t tiles = (n_words + kRankWordsPerTile - 1) / kRankWordsPerTile;
  const int64_t first = (int64_t)blockIdx.x * kExpWordsPerBlock + wave * kExpWordsPerWave;
  const bool whole = (first + kExpWordsPerWave) * 64 <= n_rows;  // wave-uniform
  u64 m[kExpRounds][2];
  if (root_kind == kRootLevels1) {
    if (whole) load_root_whole<kRootLevels1, kExpRounds>(root, first, lane, m);
    else load_root<kRootLevels1, kExpRounds>(root, first, n_words, n_rows, lane, m);
  } else {
    if (whole) load_root_whole<kRootBitmap, kExpRounds>(root, first, lane, m);
    else load_root<kRootBitmap, kExpRounds>(root, first, n_words, n_rows, lane, m);
  }
  // rank of the workgroup's first word (see expand_kernel)
  const int64_t tile = (int64_t)blockIdx.x / kExpBlocksPerTile;
  const int part_waves = (int)(blockIdx.x % kExpBlocksPerTile) * (kRankWaves / kExpBlocksPerTile);
  uint32_t before = 0;
  for (int64_t i = threadIdx.x; i < tile; i += kRankThreads) before += tile_counts[i];
  if ((int)threadIdx.x < part_waves) before += tile_counts[tiles + tile * kRankWaves + threadIdx.x];
  {
    const uint32_t lo = wave_sum(before & 0xFFFFu), hi = wave_sum(before >> 16);
    if (lane == 0) part[wave] = (u64)lo + ((u64)hi << 16);
  }
  uint32_t excl[kExpRounds];
  uint32_t run = 0;
#pragma unroll
  for (int r = 0; r < kExpRounds; ++r) {
    const uint32_t c = (uint32_t)(__builtin_popcountll(m[r][0]) + __builtin_popcountll(m[r][1]));
    const uint32_t incl = wave_inclusive_scan(c);
    excl[r] = run + incl - c;
    run += __builtin_amdgcn_readlane(incl, 63);
  }
  if (lane == 0) wave_tot[wave] = run;
  __syncthreads();
  u64 base = part[0] + part[1] + part[2] + part[3];
  for (int w = 0; w < wave; ++w) base += wave_tot[w];

  // the data predicate over blocks [b0, b0 + n_blk): bitmap dword (k * 64 + lane) of the segment
  const int64_t b0 = (int64_t)(base >> 6);
  const uint32_t lead = (uint32_t)(base & 63);
  const int n_blk = run ? (int)((lead + run + 63u) >> 6) : 0;
  const int n_st = (n_blk + kBlocksPerTile - 1) / kBlocksPerTile;
  const int64_t total_words = ((n_sub + 63) / 64) * W;
  auto load = [&](int k, u32x4 (&r)[L]) {
    const int64_t w0 = (b0 + (int64_t)k * kBlocksPerTile) * W;
    int64_t left = total_words - w0;
    const int64_t need = (int64_t)(n_blk - k * kBlocksPerTile) * W;  // never read past the wave's blocks
    left = left < need ? left : need;
    left = left < 0 ? 0 : (left > kBlocksPerTile * W ? kBlocksPerTile * W : left);
    const uint64_t* bp = left > 0 ? enc + w0 : enc;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(bp), 0, (int)(left * 8), kBufferRsrcDword3);
#pragma unroll
    for (int i = 0; i < L; ++i) r[i] = buffer_load16<true>(rsrc, (uint32_t)(i * kWave + lane) * 16u);
  };
  constexpr bool kEarly = W == 32 && (KIND == kPredSingle || KIND == kPredPair);
  if constexpr (kEarly) {
    // w = 32 comparisons, early pruning as in fle_pred32_early_kernel: the high 16 planes of a
    // sub-tile are loaded and evaluated first; the low 16 only if a row is still equal to a
    // constant after them (and then prefetched while that keeps happening)
    auto rsrc_of = [&](int k) {
      const int64_t w0 = (b0 + (int64_t)k * kBlocksPerTile) * W;
      int64_t left = total_words - w0;
      const int64_t need = (int64_t)(n_blk - k * kBlocksPerTile) * W;
      left = left < need ? left : need;
      left = left < 0 ? 0 : (left > kBlocksPerTile * W ? kBlocksPerTile * W : left);
      const uint64_t* bp = left > 0 ? enc + w0 : enc;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(bp), 0, (int)(left * 8), kBufferRsrcDword3);
    };
    // half = 1: words 16..31 of every block (planes 31..16), half = 0: words 0..15
    auto load_half = [&](int k, int half, u32x4 (&r)[4]) {
      const __amdgpu_buffer_rsrc_t rsrc = rsrc_of(k);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = i * kWave + lane;  // 256 chunks of 16 bytes per half sub-tile
        r[i] = buffer_load16<true>(rsrc, (uint32_t)(((ch >> 3) * W + half * 16 + (ch & 7) * 2) * 8));
      }
    };
    auto stage_half = [&](int half, const u32x4 (&r)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = i * kWave + lane;
        uint32_t* dst = lds32 + 2 * ((ch >> 3) * plane_stride_words(W) + half * 16 + (ch & 7) * 2);
        dst[0] = r[i].x; dst[1] = r[i].y; dst[2] = r[i].z; dst[3] = r[i].w;
      }
    };
    struct Half { uint32_t b, eq; };
    auto planes_step = [&](int half, uint32_t cc, uint32_t b_in) -> Half {
      const uint32_t* p = lds32 + plane_base_dw(W, lane);
      uint32_t x[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] = p[2 * (half * 16 + e)];
      Half h{b_in, ~0u};
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        h.b = borrow_step(h.b, x[e], bit_mask(cc, half * 16 + e));
        h.eq = eq_step(h.eq, x[e], bit_mask(cc, half * 16 + e));
      }
      return h;
    };
    auto finish = [&](int op, const Half& hi, bool with_lo, const Half& lo) -> uint32_t {
      if (op == 0) return with_lo ? (hi.eq & lo.eq) : 0u;  // without the low half no row was still equal
      const uint32_t b = with_lo ? (hi.b | (hi.eq & lo.b)) : hi.b;
      return borrow_select(b, op);
    };
    const uint32_t c1 = args.consts[0], c2 = args.const2;
    bool with_low = false;  // wave-uniform: the previous sub-tile needed the low planes
    u32x4 rh[4], rl[4];
    if (n_st > 0) load_half(0, 1, rh);
    for (int k = 0; k < n_st; ++k) {
      stage_half(1, rh);
      const bool have_low = with_low;
      if (have_low) stage_half(0, rl);
      wave_lds_fence();
      const Half hi = planes_step(1, c1, 0u);
      Half hi2{0u, 0u}, lo{0u, 0u}, lo2{0u, 0u};
      if (KIND == kPredPair) hi2 = planes_step(1, c2, 0u);
      // data rows that do not exist are padding: never let them ask for the low planes
      const int64_t valid = n_sub - ((b0 + (int64_t)k * kBlocksPerTile) * 64 + (int64_t)lane * 32);
      const uint32_t live = valid >= 32 ? ~0u : valid <= 0 ? 0u : ~((1u << (32 - valid)) - 1u);  // MSB-first rows
      const uint32_t open_rows = (KIND == kPredPair ? (hi.eq | hi2.eq) : hi.eq) & live;
      const bool undecided = __builtin_amdgcn_ballot_w64(open_rows != 0u) != 0ull;
      if (k + 1 < n_st) {  // prefetch: the high half always, the low half while the column needs it
        load_half(k + 1, 1, rh);
        if (undecided) load_half(k + 1, 0, rl);
      }
      if (undecided) {
        if (!have_low) {  // demand fetch of this sub-tile's low lines
          u32x4 now[4];
          load_half(k, 0, now);
          stage_half(0, now);
          wave_lds_fence();
        }
        lo = planes_step(0, c1, borrow_init(args.op));
        if (KIND == kPredPair) lo2 = planes_step(0, c2, borrow_init(args.op2));
      }
      with_low = undecided;
      uint32_t sel = finish(args.op, hi, undecided, lo);
      if (KIND == kPredPair) {
        const uint32_t sel2 = finish(args.op2, hi2, undecided, lo2);
        sel = args.join == 1 ? (sel & sel2) : (sel | sel2);
      }
      seg[k * 64 + lane] = bitrev32(sel & live);
      wave_lds_fence();  // the plane image is reused by the next sub-tile; the segment is read below
    }
  } else {
    u32x4 r[L];
    if (n_st > 0) load(0, r);
    for (int k = 0; k < n_st; ++k) {
      tile_to_lds<L>(lds32, W, lane, r);
      if (k + 1 < n_st) load(k + 1, r);  // register prefetch
      wave_lds_fence();
      uint32_t sel;
      if (KIND == kPredSingle) {
        uint32_t p[W];
        planes_from_lds<W>(lds32, lane, p);
        sel = pred_from_regs<W>(p, args);
      } else if (KIND == kPredPair) {
        uint32_t r1, r2;
        pred_pair_from_lds(lds32, W, lane, args.op, args.consts[0], args.op2, args.const2, &r1, &r2);
        sel = args.join == 1 ? (r1 & r2) : (r1 | r2);
      } else if (kInTable) {  // long list: decode, one set lookup per value
        uint32_t p[W];
        planes_from_lds<W>(lds32, lane, p);
        uint32_t v[32];
        planes_to_values<W>(p, v);
        sel = bitrev32(in_table_lookup(in_table, v));
      } else if (W <= 16) {
        uint32_t p[W];
        planes_from_lds<W>(lds32, lane, p);
        sel = pred_in_from_regs<W>(p, args.consts, args.n_consts);
      } else {
        sel = args.in_list ? pred_in_from_lds(lds32, W, lane, args.in_list, args.in_list_n)
                           : pred_in_from_lds(lds32, W, lane, args.consts, args.n_consts);
      }
      uint32_t bm = bitrev32(sel);
      const int64_t valid = n_sub - ((b0 + (int64_t)k * kBlocksPerTile) * 64 + (int64_t)lane * 32);
      if (valid < 32) bm = valid <= 0 ? 0u : (bm & ((1u << valid) - 1u));  // data rows that do not exist select nothing
      seg[k * 64 + lane] = bm;
      wave_lds_fence();  // the plane image is reused by the next sub-tile; the segment is read below
    }
  }

  // deposit: the window of a word = three segment dwords from its rank on
  [[maybe_unused]] WindowCarry carry;
#pragma unroll
  for (int r2 = 0; r2 < kExpRounds; ++r2) {
    const int64_t w0 = first + r2 * 128 + 2 * lane;
    u64 res[2];
    uint32_t rel = lead + excl[r2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint32_t mlo = (uint32_t)m[r2][e], mhi = (uint32_t)(m[r2][e] >> 32);
      uint32_t lo = 0, src_hi = 0;
      if (m[r2][e] != 0ull) {  // (a word without NOT-NULL rows may sit past the evaluated blocks)
        const uint32_t i0 = rel >> 5, sh = rel & 31u;
        const uint32_t d0 = seg[i0], d1 = seg[i0 + 1], d2 = seg[i0 + 2];
        lo = __builtin_amdgcn_alignbit(d1, d0, sh);
        const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh);
        const uint32_t pcl = (uint32_t)__builtin_popcount(mlo);
        src_hi = pcl == 32u ? hi : __builtin_amdgcn_alignbit(hi, lo, pcl);
      }
      res[e] = (u64)deposit32(lo, mlo, lut) | ((u64)deposit32(src_hi, mhi, lut) << 32);
      rel += (uint32_t)__builtin_popcountll(m[r2][e]);
    }
    if constexpr (PAGED) {
      // whole words on a 16-byte boundary of the chunk's bitmap take the stores below; everything
      // else goes dword by dword through the window (shifted, shared dwords merged atomically)
      if (!(whole && win->shift == 0u && win->coherent == 0u && (reinterpret_cast<uintptr_t>(win->base) & 15u) == 0u)) {
        const uint32_t in[4] = {(uint32_t)res[0], (uint32_t)(res[0] >> 32), (uint32_t)res[1], (uint32_t)(res[1] >> 32)};
        window_emit_quad(*win, carry, 2 * w0, in, args.combine);
        continue;
      }
    }
    if (w0 + 1 < n_words) {
      u32x4* dst = reinterpret_cast<u32x4*>(out + w0);
      if (args.combine != 0) {  // wave-uniform
        const u32x4 old = *dst;
        const u64 o0 = ((u64)old.y << 32) | old.x, o1 = ((u64)old.w << 32) | old.z;
        res[0] = args.combine == 1 ? (res[0] & o0) : (res[0] | o0);
        res[1] = args.combine == 1 ? (res[1] & o1) : (res[1] | o1);
      }
      const u32x4 t = {(uint32_t)res[0], (uint32_t)(res[0] >> 32), (uint32_t)res[1], (uint32_t)(res[1] >> 32)};
      IPS_STREAM_STORE16(dst, t);
    } else if (w0 < n_words) {
      if (args.combine != 0) res[0] = args.combine == 1 ? (res[0] & out[w0]) : (res[0] | out[w0]);
      out[w0] = res[0];
    }
  }
  if constexpr (PAGED) window_flush(*win, carry, args.combine);
}

template <int W, int KIND>
__global__ __launch_bounds__(kThreads, IPS_MIN_WAVES_PER_EU) void fle_leaf_kernel(
    const uint64_t* __restrict__ enc, int64_t n_sub, PredArgs args, u64* __restrict__ out) {
  fle_leaf_body<W, KIND, false>(enc, n_sub, args, out, nullptr, reinterpret_cast<const u64*>(args.aux_root),
                                args.aux_kind, args.aux_rows, args.aux_counts);
}

template <int W, int KIND>
__global__ __launch_bounds__(kThreads, IPS_MIN_WAVES_PER_EU) void fle_leaf_pages_kernel(
    const ChunkPage* __restrict__ pages, int64_t chunk_rows, PredArgs args, uint32_t* __restrict__ bitmap32) {
  const ChunkPage pg = pages[blockIdx.y];
  if ((int64_t)blockIdx.x * kExpWordsPerBlock * 64 >= pg.n_rows) {  // (the grid is sized for the largest page)
    page_done(args.done, args.done_page0, args.done_epoch);
    return;
  }
  const BitmapWindow win = bitmap_window(bitmap32, pg, chunk_rows, args.done);
  fle_leaf_body<W, KIND, true>(pg.data, pg.n_data, args, reinterpret_cast<u64*>(win.base), &win,
                               reinterpret_cast<const u64*>(pg.levels), kRootLevels1, pg.n_rows,
                               args.aux_counts + pg.rank0);
  page_done(args.done, args.done_page0, args.done_epoch);
}

// ---------------------------------------------------------------------------------------------
// Late materialisation of an OPTIONAL column in one pass (ReadValue(skip) over a whole selection,
// hdfs-parquet-scanner.cc:1006-1038 + 927-979: ReadDefinitionLevel says which selected rows are
// NULL and which data row a NOT-NULL one decodes).  A workgroup owns a quarter rank tile of ROWS
// (1024 words; a wave 256 words = 16 384 rows).  From the three count tables of
// rank3_counts_kernel the wave knows the data row of its first NOT-NULL row (R), the index of its
// first selected row (S) and the dense output index of its first selected NOT-NULL row (RS).
// First the NOT-NULL flag of every selected row: the NOT-NULL bits at the selected positions
// (pext through the nibble table), assembled in an LDS segment and flushed at the wave's S-rank.
// Then it extracts the selection at the NOT-NULL positions into the same segment -- the selection
// over its OWN data rows, which the step-by-step route wrote to HBM as a bitmap over all data rows
// -- and walks the data sub-tiles that hold them: blocks without a selected row are not loaded,
// of the others the selected rows go on an index list and are decoded one per lane straight from
// the plane image (through the dictionary if there is one) into dense[RS index .. ), coalesced and
// in row order.  No per-batch buffers, no compaction pass.
// ---------------------------------------------------------------------------------------------
struct SelNullArgs {
  const unsigned long long* root;
  const unsigned long long* sel;
  const uint32_t* c_r;
  const uint32_t* c_s;
  const uint32_t* c_rs;
  unsigned long long* flags;  // one NOT-NULL bit per selected row, cleared by the counting pass
  int64_t* n_selected;        // popcount of the selection
  int64_t* bad_index;         // set non-zero when a selected code lies outside the dictionary
  int64_t n_rows;
  int32_t root_kind;
  // over a column chunk held as a page list (ips_chunk_select_nullable; blockIdx.y = page, else NULL): the page's
  // levels / data blocks / rank tables come from its descriptor, its selection from an aligned copy of the
  // chunk-wide one (sel_copy + 32 words per earlier batch), and the page's first selected row / first selected
  // NOT-NULL row continue the earlier pages' (page_s / page_rs, exclusive prefix sums over the pages)
  const ChunkPage* pages;
  int32_t n_pages;       // (grid.y of the launch; n_rows = rows of the largest page)
  int32_t reserved;
  const unsigned long long* sel_copy;
  const unsigned long long* page_s;
  const unsigned long long* page_rs;
};

template <int W, int G>
__global__ __launch_bounds__(kThreads, IPS_MIN_WAVES_PER_EU) void fle_select_nullable_kernel(
    const uint64_t* __restrict__ enc_arg, int64_t n_data_arg, SelNullArgs a,
    typename GatherT<G>::type* __restrict__ dense, const typename GatherT<G>::type* __restrict__ dict,
    uint32_t dict_entries, int64_t* __restrict__ n_values) {
  using GT = typename GatherT<G>::type;
  const uint64_t* __restrict__ enc = enc_arg;
  int64_t n_data = n_data_arg;
  constexpr int kRegionBytes = plane_tile_bytes(W);
  constexpr int kSegWords64 = kLeafSegDwords / 2 + 2;  // the wave's data-row selection, <= 257 words used
  constexpr uint32_t kListMax = 1024;                  // index-list window (entries of 16 bits)
  __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerBlock * kRegionBytes / 4];
  __shared__ __attribute__((aligned(16))) u64 seg_all[kWavesPerBlock][kSegWords64];
  __shared__ uint16_t list_all[kWavesPerBlock][kListMax];
  __shared__ uint8_t lut[256];
  __shared__ u64 part_r[kRankWaves], part_s[kRankWaves], part_rs[kRankWaves];
  __shared__ uint32_t tot_r[kRankWaves], tot_s[kRankWaves], tot_rs[kRankWaves];
  constexpr int L = (16 * W + kWave - 1) / kWave;
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * (kRegionBytes / 4);
  u64* seg = seg_all[wave];
  uint16_t* list = list_all[wave];
  lut[threadIdx.x] = (uint8_t)extract_lut_entry(threadIdx.x);
  for (int i = lane; i < kSegWords64; i += kWave) seg[i] = 0ull;

  const u64* __restrict__ root = a.root;
  const u64* __restrict__ sel = a.sel;
  const uint32_t* __restrict__ c_r = a.c_r;
  const uint32_t* __restrict__ c_s = a.c_s;
  const uint32_t* __restrict__ c_rs = a.c_rs;
  int64_t n_rows = a.n_rows;
  u64 page_s = 0, page_rs = 0;
  const bool paged = a.pages != nullptr;  // wave-uniform
  if (paged) {
    const ChunkPage pg = a.pages[blockIdx.y];
    if ((int64_t)blockIdx.x * kExpWordsPerBlock * 64 >= pg.n_rows) return;  // (the grid is sized for the largest page)
    enc = pg.data;
    n_data = pg.n_data;
    root = reinterpret_cast<const u64*>(pg.levels);
    sel = a.sel_copy + (size_t)pg.batch0 * (kRowsPerTile / 64);
    n_rows = pg.n_rows;
    c_r += pg.rank0;
    c_s += pg.rank0;
    c_rs += pg.rank0;
    page_s = a.page_s[blockIdx.y];
    page_rs = a.page_rs[blockIdx.y];
  }
  const int64_t n_words = (n_rows + 63) / 64;
  const int64_t tiles = (n_words + kRankWordsPerTile - 1) / kRankWordsPerTile;
  const int64_t first = (int64_t)blockIdx.x * kExpWordsPerBlock + wave * kExpWordsPerWave;
  u64 m[kExpRounds][2], sv[kExpRounds][2];
  if (a.root_kind == kRootLevels1) load_root<kRootLevels1, kExpRounds>(root, first, n_words, n_rows, lane, m);
  else load_root<kRootBitmap, kExpRounds>(root, first, n_words, n_rows, lane, m);
  load_root<kRootBitmap, kExpRounds>(sel, first, n_words, n_rows, lane, sv);

  // ranks of the workgroup's first word in R, S and RS (see expand_kernel)
  const int64_t tile = (int64_t)blockIdx.x / kExpBlocksPerTile;
  const int part_waves = (int)(blockIdx.x % kExpBlocksPerTile) * (kRankWaves / kExpBlocksPerTile);
  uint32_t before_r = 0, before_s = 0, before_rs = 0;
  for (int64_t i = threadIdx.x; i < tile; i += kRankThreads) {
    before_r += c_r[i];
    before_s += c_s[i];
    before_rs += c_rs[i];
  }
  if ((int)threadIdx.x < part_waves) {
    before_r += c_r[tiles + tile * kRankWaves + threadIdx.x];
    before_s += c_s[tiles + tile * kRankWaves + threadIdx.x];
    before_rs += c_rs[tiles + tile * kRankWaves + threadIdx.x];
  }
  {
    const uint32_t lo = wave_sum(before_r & 0xFFFFu), hi = wave_sum(before_r >> 16);
    const uint32_t lo1 = wave_sum(before_s & 0xFFFFu), hi1 = wave_sum(before_s >> 16);
    const uint32_t lo2 = wave_sum(before_rs & 0xFFFFu), hi2 = wave_sum(before_rs >> 16);
    if (lane == 0) {
      part_r[wave] = (u64)lo + ((u64)hi << 16);
      part_s[wave] = (u64)lo1 + ((u64)hi1 << 16);
      part_rs[wave] = (u64)lo2 + ((u64)hi2 << 16);
    }
  }
  uint32_t excl[kExpRounds], excl_s[kExpRounds];
  uint32_t run_r = 0, run_s = 0, mine_rs = 0;
#pragma unroll
  for (int r = 0; r < kExpRounds; ++r) {
    const uint32_t c = (uint32_t)(__builtin_popcountll(m[r][0]) + __builtin_popcountll(m[r][1]));
    const uint32_t incl = wave_inclusive_scan(c);
    excl[r] = run_r + incl - c;
    run_r += __builtin_amdgcn_readlane(incl, 63);
    const uint32_t cs = (uint32_t)(__builtin_popcountll(sv[r][0]) + __builtin_popcountll(sv[r][1]));
    const uint32_t incl_s = wave_inclusive_scan(cs);
    excl_s[r] = run_s + incl_s - cs;
    run_s += __builtin_amdgcn_readlane(incl_s, 63);
    mine_rs += (uint32_t)(__builtin_popcountll(m[r][0] & sv[r][0]) + __builtin_popcountll(m[r][1] & sv[r][1]));
  }
  const uint32_t run_rs = wave_sum(mine_rs);
  if (lane == 0) {
    tot_r[wave] = run_r;
    tot_s[wave] = run_s;
    tot_rs[wave] = run_rs;
  }
  __syncthreads();
  u64 base_r = part_r[0] + part_r[1] + part_r[2] + part_r[3];
  u64 base_s = page_s + part_s[0] + part_s[1] + part_s[2] + part_s[3];
  u64 base_rs = page_rs + part_rs[0] + part_rs[1] + part_rs[2] + part_rs[3];
  for (int w = 0; w < wave; ++w) {
    base_r += tot_r[w];
    base_s += tot_s[w];
    base_rs += tot_rs[w];
  }
  if (!paged && a.n_selected && blockIdx.x == gridDim.x - 1 && threadIdx.x == kRankThreads - 1)
    *a.n_selected = (int64_t)(base_s + run_s);  // last wave of the last block: popcount(selection)

  // the NOT-NULL flag of every selected row (the NULL indicator bit, hdfs-parquet-scanner.cc:
  // 1022-1026): the NOT-NULL bits at the selected positions, appended at the wave's rank in S.
  // As in compress_kernel the wave assembles its bits in the LDS segment; the two end words can be
  // shared with the neighbouring waves and leave as atomics (the counting pass cleared the words).
  {
    const uint32_t lead_s = (uint32_t)(base_s & 63);
#pragma unroll
    for (int r = 0; r < kExpRounds; ++r) {
      uint32_t o = lead_s + excl_s[r];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const u64 sk = sv[r][e];
        if ((sk & m[r][e]) != 0ull) {
          const u64 bits = extract64(m[r][e], sk, lut);
          const uint32_t sh = o & 63u;
          __hip_atomic_fetch_or(&seg[o >> 6], bits << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (sh && (bits >> (64 - sh)))
            __hip_atomic_fetch_or(&seg[(o >> 6) + 1], bits >> (64 - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        o += (uint32_t)__builtin_popcountll(sk);
      }
    }
    wave_lds_fence();
    const int64_t g0 = (int64_t)(base_s >> 6);
    const uint32_t nw = run_s ? (lead_s + run_s + 63) / 64 : 0;
    for (uint32_t i = lane; i < nw; i += kWave) {
      const u64 v = seg[i];
      if (i == 0 || i == nw - 1) {
        if (v) atomicOr(a.flags + g0 + i, (unsigned long long)v);
      } else {
        a.flags[g0 + i] = v;
      }
      seg[i] = 0ull;  // the segment is used again below
    }
    wave_lds_fence();
  }

  // the selection over the wave's data rows: bit (lead + rank inside the wave) of the segment
  const uint32_t lead = (uint32_t)(base_r & 63);
#pragma unroll
  for (int r = 0; r < kExpRounds; ++r) {
    uint32_t o = lead + excl[r];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const u64 mk = m[r][e];
      const uint32_t pc = (uint32_t)__builtin_popcountll(mk);
      if ((sv[r][e] & mk) != 0ull) {  // (a word without a selected NOT-NULL row adds nothing)
        const u64 bits = extract64(sv[r][e], mk, lut);
        const uint32_t sh = o & 63u;
        __hip_atomic_fetch_or(&seg[o >> 6], bits << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (sh && (bits >> (64 - sh)))
          __hip_atomic_fetch_or(&seg[(o >> 6) + 1], bits >> (64 - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      o += pc;
    }
  }
  wave_lds_fence();
  const uint32_t* seg32 = reinterpret_cast<const uint32_t*>(seg);

  // data blocks [b0, b0 + n_blk) hold the wave's data rows; dword (k * 64 + lane) of the segment is
  // this lane's half-block of sub-tile k
  const int64_t b0 = (int64_t)(base_r >> 6);
  const int n_blk = run_r ? (int)((lead + run_r + 63u) >> 6) : 0;
  const int n_st = (n_blk + kBlocksPerTile - 1) / kBlocksPerTile;
  const int64_t total_words = ((n_data + 63) / 64) * W;
  auto dword_of = [&](int k) -> uint32_t {  // selected rows of the half-block that exist in the data buffer
    if (k >= n_st) return 0u;
    uint32_t bm = seg32[k * 64 + lane];
    const int64_t valid = n_data - ((b0 + (int64_t)k * kBlocksPerTile) * 64 + (int64_t)lane * 32);
    if (valid < 32) bm = valid <= 0 ? 0u : (bm & ((1u << valid) - 1u));
    return bm;
  };
  // per 16-byte chunk of a sub-tile that this lane moves: the bits of the "needed blocks" mask
  // that decide whether it is loaded, and its place in the LDS image -- the same for every sub-tile
  uint32_t need_sh[L];  // 2 * first block | 2 * second block << 8; 0xFFFF: no such chunk
  int img_dw[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const int c = i * kWave + lane;
    const int blk0 = (2 * c) / W, blk1 = (2 * c + 1) / W;
    need_sh[i] = c < 16 * W ? (uint32_t)(2 * blk0) | ((uint32_t)(2 * blk1) << 8) : 0xFFFFu;
    // (odd W: block stride W, the image is linear; even W: both words belong to one block)
    img_dw[i] = 2 * (blk0 * (W | 1) + (2 * c - blk0 * W));
  }
  auto load = [&](int k, uint32_t bm, u32x4 (&r)[L]) {  // only the blocks that hold a selected row
    const uint64_t any = __builtin_amdgcn_ballot_w64(bm != 0u);  // bit l <-> half-block of lane l
    if (k >= n_st || any == 0ull) return;                        // (wave-uniform)
    const uint64_t need = any | (any >> 1);                      // bit 2b <-> block b
    const int64_t w0 = (b0 + (int64_t)k * kBlocksPerTile) * W;
    int64_t left = total_words - w0;
    left = left < 0 ? 0 : (left > kBlocksPerTile * W ? kBlocksPerTile * W : left);
    const uint64_t* bp = left > 0 ? enc + w0 : enc;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(bp), 0, (int)(left * 8), kBufferRsrcDword3);
#pragma unroll
    for (int i = 0; i < L; ++i) {
      if (need_sh[i] != 0xFFFFu && (((need >> (need_sh[i] & 63u)) | (need >> (need_sh[i] >> 8))) & 1ull))
        r[i] = buffer_load16<true>(rsrc, (uint32_t)(i * kWave + lane) * 16u);
    }
  };
  auto to_lds = [&](const u32x4 (&r)[L]) {
#pragma unroll
    for (int i = 0; i < L; ++i) {
      if (need_sh[i] != 0xFFFFu) {
        const u32x2 lo = {r[i].x, r[i].y}, hi = {r[i].z, r[i].w};
        *reinterpret_cast<u32x2*>(lds32 + img_dw[i]) = lo;
        *reinterpret_cast<u32x2*>(lds32 + img_dw[i] + 2) = hi;
      }
    }
  };
  u64 out_pos = base_rs;  // dense index of the wave's next value
  // sub-tile k: its bytes are in r (if any row is selected); r is refilled for sub-tile k + 2 as
  // soon as it has been copied to LDS -- two sub-tiles of a wave are in flight, the chain of nine
  // dependent steps per wave is what bounds this kernel
  auto step = [&](int k, u32x4 (&r)[L], uint32_t bm, uint32_t bm_ahead) {
    const uint32_t cnt = (uint32_t)__builtin_popcount(bm);
    const uint32_t incl = wave_inclusive_scan(cnt);
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    if (total != 0u) to_lds(r);  // wave-uniform
    load(k + 2, bm_ahead, r);
    if (total == 0u) return;
    // index list, in windows of kListMax: entry e of the sub-tile = (half-block << 5 | row) of its
    // e-th selected row; then lane l decodes entries l, l + 64, ... straight from the plane image
    // (one bit per plane: 2 ops) and stores them side by side -- nothing is transposed for the
    // rows that are not selected, and the output leaves coalesced whatever its alignment
    uint32_t mm = bm;
    uint32_t idx = incl - cnt;  // sub-tile index of the lane's next entry
    GT* dst = dense + out_pos;
    for (uint32_t win0 = 0; win0 < total; win0 += kListMax) {
      while (mm != 0u && idx < win0 + kListMax) {
        list[idx - win0] = (uint16_t)((uint32_t)lane << 5 | (uint32_t)__builtin_ctz(mm));
        mm &= mm - 1u;
        ++idx;
      }
      wave_lds_fence();  // the list, and (first window) the plane image
      const uint32_t n_win = total - win0 < kListMax ? total - win0 : kListMax;
      for (uint32_t e = lane; e < n_win; e += kWave) {
        const uint32_t entry = list[e];
        const uint32_t h = entry >> 5, sh = 31u - (entry & 31u);
        const uint32_t* pl = lds32 + 2 * ((h >> 1) * (W | 1)) + (1 - (h & 1));
        uint32_t val = 0;
#pragma unroll
        for (int b = 0; b < W; ++b) val |= ((pl[2 * b] >> sh) & 1u) << b;
        if (G == 0) dst[win0 + e] = (GT)val;
        else if (val < dict_entries) dst[win0 + e] = dict[val];
        else if (a.bad_index) *a.bad_index = 1;  // DictDecoder::GetValue returns false (dict-encoding.h:316)
      }
      wave_lds_fence();  // the list is rewritten by the next window, the image by the next sub-tile
    }
    out_pos += total;
  };
  u32x4 ra[L], rb[L];
  uint32_t bm0 = dword_of(0), bm1 = dword_of(1);
  load(0, bm0, ra);
  load(1, bm1, rb);
  for (int k = 0; k < n_st; k += 2) {
    const uint32_t bm2 = dword_of(k + 2), bm3 = dword_of(k + 3);
    step(k, ra, bm0, bm2);
    if (k + 1 < n_st) step(k + 1, rb, bm1, bm3);
    bm0 = bm2;
    bm1 = bm3;
  }

  // the number of values: the wave that holds the end of the data buffer knows it, else the last wave
  if (!paged && n_values && lane == 0) {  // (paged: the page prefix kernel knows both totals)
    const bool last = blockIdx.x == gridDim.x - 1 && wave == kRankWaves - 1;
    const u64 nd = (u64)n_data;
    if ((base_r <= nd && nd < base_r + run_r) || (last && nd >= base_r + run_r)) *n_values = (int64_t)out_pos;
  }
}

// ---------------------------------------------------------------------------------------------
// w = 32, single comparison, early pruning.  A block of width 32 is two 128-byte lines: planes
// 31..16 in the second, 15..0 in the first.  The MSB->LSB recurrence only needs the low planes for
// rows that are still EQUAL to the constant after the high ones -- on a column that uses its 32
// bits that is one row in 65536 -- so the wave first loads and evaluates the high lines only (half
// the bytes) and fetches the low lines of a sub-tile only if some row in it is still undecided.
// A wave that had to fetch them keeps prefetching both halves until a sub-tile is decided by the
// high planes again, which bounds the cost on columns whose high bits all equal the constant's.
// ---------------------------------------------------------------------------------------------
// PAIR: two comparisons on the column in the same pass (BETWEEN): a row needs the low planes if it
// is still equal to either constant.
template <int W, bool PAIR, bool PAGED>  // W = 32 only
__device__ __forceinline__ void fle_pred32_early_body(const uint64_t* __restrict__ enc, int64_t n_rows,
                                                      const PredArgs& args, uint32_t* __restrict__ bitmap32,
                                                      const BitmapWindow* win) {
  static_assert(W == 32, "two 128-byte lines per block");
  __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerBlock * plane_tile_bytes(W) / 4];
  if (!PAGED && (int)blockIdx.x < args.aux_blocks) {  // counting workgroups of a nullable leaf
    rank_aux_counts(args);
    return;
  }
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * (plane_tile_bytes(W) / 4);
  int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t total_words = ((n_rows + 63) / 64) * W;
  int64_t stride = (int64_t)((int)gridDim.x - args.aux_blocks) * kWavesPerBlock;
  const int64_t bm_dwords = bitmap_dwords(n_rows);
  const uint32_t c = args.consts[0];
  const uint32_t c2 = args.const2;

  // half = 1: words 16..31 of every block (planes 31..16), half = 0: words 0..15
  auto load_half = [&](int64_t tile, int half, u32x4 (&r)[4]) {
    const __amdgpu_buffer_rsrc_t rsrc = tile_rsrc(enc, tile, W, total_words);  // range-checked by the hardware
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = i * kWave + lane;                  // 256 chunks of 16 bytes per half tile
      r[i] = buffer_load16<true>(rsrc, (uint32_t)(((ch >> 3) * W + half * 16 + (ch & 7) * 2) * 8));
    }
  };
  auto stage_half = [&](int half, const u32x4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = i * kWave + lane;
      uint32_t* dst = lds32 + 2 * ((ch >> 3) * plane_stride_words(W) + half * 16 + (ch & 7) * 2);
      dst[0] = r[i].x; dst[1] = r[i].y; dst[2] = r[i].z; dst[3] = r[i].w;
    }
  };
  // One half (16 planes, taken LSB -> MSB) of the comparison against constant cc: the borrow
  // chain started from b0 and the equality chain (ips_bitops.h), one v_bitop3_b32 each per plane.
  // The halves compose as  borrow(31..0) = borrow_hi(from 0) | (eq_hi & borrow_lo(from init)).
  struct Half { uint32_t b, eq; };
  auto planes_step = [&](int half, uint32_t cc, uint32_t b0) -> Half {
    const uint32_t* p = lds32 + plane_base_dw(W, lane);
    uint32_t x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = p[2 * (half * 16 + e)];
    Half h{b0, ~0u};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      h.b = borrow_step(h.b, x[e], bit_mask(cc, half * 16 + e));
      h.eq = eq_step(h.eq, x[e], bit_mask(cc, half * 16 + e));
    }
    return h;
  };
  // result of op from the high half and (when it was needed) the low half
  auto finish = [&](int op, const Half& hi, bool with_lo, const Half& lo) -> uint32_t {
    if (op == 0) return with_lo ? (hi.eq & lo.eq) : 0u;  // without the low half no row was still equal
    const uint32_t b = with_lo ? (hi.b | (hi.eq & lo.b)) : hi.b;
    return borrow_select(b, op);
  };

  int64_t tile = (int64_t)((int)blockIdx.x - args.aux_blocks) * kWavesPerBlock + wave;
  [[maybe_unused]] WindowCarry carry;
  if constexpr (PAGED) {  // this wave's share of the page's sub-tiles
    // (longer shares for narrow columns -- 32 sub-tiles at w <= 8, 16 at w <= 16 -- left the unaligned
    // Q6 plan where it was, 446 us: what the contiguous shares cost in streaming order they save in atomics)
    const TileShare sh = tile_share(*win, tiles, tile, stride);
    tile = sh.first;
    stride = sh.step;
    tiles = sh.end;
  }
  bool with_low = false;  // wave-uniform: the previous sub-tile needed the low planes
  auto combine_operand = [&](int64_t t) -> uint32_t {  // see fle_pred_body
    const int64_t d = t * 64 + lane;
    if constexpr (PAGED) return window_has_operand(*win, args.combine) ? window_operand(*win, d) : 0u;
    return (args.combine != 0 && d < bm_dwords) ? bitmap32[d] : 0u;
  };
  u32x4 rh[4], rl[4];
  uint32_t old = 0u;
  if (tile < tiles) {
    old = combine_operand(tile);
    load_half(tile, 1, rh);
  }
  while (tile < tiles) {
    stage_half(1, rh);
    const uint32_t old_now = old;
    const bool have_low = with_low;
    if (have_low) stage_half(0, rl);
    const int64_t next = tile + stride;
    wave_lds_fence();
    const Half hi = planes_step(1, c, 0u);
    Half hi2{0u, 0u}, lo{0u, 0u}, lo2{0u, 0u};
    if (PAIR) hi2 = planes_step(1, c2, 0u);
    // rows beyond n_rows are padding: never let them ask for the low planes
    const int64_t valid = n_rows - (tile * kRowsPerTile + (int64_t)lane * 32);
    const uint32_t live = valid >= 32 ? ~0u : valid <= 0 ? 0u : ~((1u << (32 - valid)) - 1u);  // MSB-first rows
    const uint32_t open_rows = (PAIR ? (hi.eq | hi2.eq) : hi.eq) & live;
    const bool undecided = __builtin_amdgcn_ballot_w64(open_rows != 0u) != 0ull;
    if (next < tiles) {  // prefetch: the high half always, the low half while the column needs it
      old = combine_operand(next);
      load_half(next, 1, rh);
      if (undecided) load_half(next, 0, rl);
    }
    if (undecided) {
      if (!have_low) {  // demand fetch of this sub-tile's low lines
        u32x4 now[4];
        load_half(tile, 0, now);
        stage_half(0, now);
        wave_lds_fence();
      }
      lo = planes_step(0, c, borrow_init(args.op));
      if (PAIR) lo2 = planes_step(0, c2, borrow_init(args.op2));
    }
    with_low = undecided;
    uint32_t sel = finish(args.op, hi, undecided, lo);
    if (PAIR) {
      const uint32_t sel2 = finish(args.op2, hi2, undecided, lo2);
      sel = args.join == 1 ? (sel & sel2) : (sel | sel2);
    }
    uint32_t bm = finish_bitmap_dword(sel, tile, lane, n_rows);
    const int64_t d = tile * 64 + lane;
    if constexpr (PAGED) {
      window_emit(*win, carry, d, bm, args.combine, kWave - 1, window_has_operand(*win, args.combine), old_now);
    } else if (d < bm_dwords) {
      if (args.combine == 1) bm &= old_now;
      else if (args.combine == 2) bm |= old_now;
      IPS_BITMAP_STORE(bitmap32 + d, bm);
    }
    wave_lds_fence();  // LDS region is reused by the next sub-tile
    tile = next;
  }
  if constexpr (PAGED) window_flush(*win, carry, args.combine);
}

template <int W, bool PAIR>
__global__ __launch_bounds__(kThreads, IPS_MIN_WAVES_PER_EU) void fle_pred32_early_kernel(
    const uint64_t* __restrict__ enc, int64_t n_rows, PredArgs args, uint32_t* __restrict__ bitmap32) {
  fle_pred32_early_body<W, PAIR, false>(enc, n_rows, args, bitmap32, nullptr);
}

template <int W, bool PAIR>
__global__ __launch_bounds__(kThreads, IPS_MIN_WAVES_PER_EU) void fle_pred32_early_pages_kernel(
    const ChunkPage* __restrict__ pages, int64_t chunk_rows, PredArgs args, uint32_t* __restrict__ bitmap32) {
  const ChunkPage pg = pages[blockIdx.y];
  const BitmapWindow win = bitmap_window(bitmap32, pg, chunk_rows, args.done, args.edges);
  fle_pred32_early_body<W, PAIR, true>(pg.data, pg.n_data, args, nullptr, &win);
  page_done(args.done, args.done_page0, args.done_epoch);
}

// ---------------------------------------------------------------------------------------------
// Full decode: FleDecoder::Get x n via Unpack_w (fle-encoding.h:404-567, 569-7329), batch form.
// OW = bytes per stored value (1, 2, 4) when G == 0; with G != 0 every code is looked up in the
// dictionary and the G-byte entry is stored (DictDecoder::GetValue, dict-encoding.h:310-319).
// ---------------------------------------------------------------------------------------------
// Dictionary columns of up to 16 bits (codes -> entries) take the PACKED path:
// the lane parks its 32 values lane-packed (8 registers of four bytes / 16 of two halfwords: the
// transposition stops where the values are bytes / halfwords), and the output is produced in
// 16-byte pieces of four consecutive rows: piece P = 64 k + lane of the sub-tile is rows 4q..4q+3 of
// source lane s = P / 8, q = lane % 8 -- one field of four neighbouring parked registers, i.e. one
// ds_read_b128 and four bit-field extracts (8-byte entries: pieces of two rows, one ds_read_b64, so
// that every store instruction still writes 1 KiB of consecutive output).  Against the row tile of the general path (every value
// unpacked to a dword, 32 dword writes + reads per lane): 3 / 5 KiB of LDS per wave instead of 9
// and about half the VALU (dictionary decode D = 4096, w = 12: 525 -> 2xx per sub-tile).
template <int W, int OW, int G>
struct DecodeLds {
  static constexpr bool kPacked = IPS_DECODE_PACKED && W <= 16 && G != 0;  // (to plain dwords: no gain, write-bound)
  static constexpr int kWaveBytes = !kPacked ? kRowTileBytes : 64 * packed_lane_stride(W <= 8 ? 8 : 16);
  static_assert(!kPacked || plane_tile_bytes(W) <= kWaveBytes, "the plane image fits");
};

// BW = waves per workgroup.  The default (4) keeps a private copy of a dictionary of up to 32 KiB
// per workgroup.  SHARED is the variant for larger dictionaries: ONE workgroup of 16 / 8 / 4 waves
// per CU whose waves share a single copy in dynamic LDS (dict_entries * G bytes, up to ~140 KiB
// next to the waves' 3-5 KiB images) instead of gathering every row from L2 (D = 16384 int32:
// 741 -> 314 us).
template <int W, int OW, int G, int BW = kWavesPerBlock, bool SHARED = (BW != kWavesPerBlock), bool TAIL = false>
__global__ __launch_bounds__(BW * kWave, SHARED ? 1 : IPS_MIN_WAVES_PER_EU) void fle_decode_kernel(
    const uint64_t* __restrict__ enc, int64_t n_rows, void* __restrict__ out,
    const typename GatherT<G>::type* __restrict__ dict, uint32_t dict_entries,
    int32_t* __restrict__ bad_index, uint32_t lds_entries) {
  constexpr bool kPackedOut = DecodeLds<W, OW, G>::kPacked;
  constexpr int kDecWaveBytes = DecodeLds<W, OW, G>::kWaveBytes;
  constexpr bool kSharedDict = SHARED;
  __shared__ __attribute__((aligned(16))) uint32_t lds_all[BW * kDecWaveBytes / 4];
  using GT = typename GatherT<G>::type;
  constexpr bool kPrivateDict = !kSharedDict && DictLds<W, G, kDecodeDictLdsBytes>::kUse;
  __shared__ GT dict_lds[kPrivateDict ? DictLds<W, G, kDecodeDictLdsBytes>::kEntries : 1];
  extern __shared__ __attribute__((aligned(16))) uint8_t dict_dyn_bytes[];
  const GT* dict_dyn = reinterpret_cast<const GT*>(dict_dyn_bytes);
  constexpr int L = (16 * W + kWave - 1) / kWave;
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * (kDecWaveBytes / 4);
  if constexpr (kSharedDict) {
    // the first lds_entries entries (all of them when the dictionary fits: the reference's largest
    // int32 dictionary, 40000 entries = 160 000 bytes, leaves no room for a wave's image, so its
    // last tenth stays in L2)
    GT* dst = reinterpret_cast<GT*>(dict_dyn_bytes);
    for (uint32_t i = threadIdx.x; i < lds_entries; i += BW * kWave) dst[i] = dict[i];
    __syncthreads();
  } else if constexpr (kPrivateDict) {
    for (uint32_t i = threadIdx.x; i < dict_entries && i < (uint32_t)DictLds<W, G, kDecodeDictLdsBytes>::kEntries; i += kThreads)
      dict_lds[i] = dict[i];
    __syncthreads();
  }
  auto lookup = [&](uint32_t code) -> GT {
    if constexpr (kSharedDict && TAIL) {  // a dictionary that does not fit: its tail is gathered from L2
      GT v = dict_dyn[code < lds_entries ? code : 0u];
      if (code >= lds_entries) v = dict[code];
      return v;
    } else if constexpr (kSharedDict) {
      return dict_dyn[code];
    }
    else if constexpr (kPrivateDict) return dict_lds[code];
    else return dict[code];
  };

  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t total_words = ((n_rows + 63) / 64) * W;
  const int64_t stride = (int64_t)gridDim.x * BW;
  int64_t tile = (int64_t)blockIdx.x * BW + wave;

  u32x4 r[L];
  if (tile < tiles) tile_load<L, IPS_DECODE_NT_LOADS>(enc, tile, W, total_words, lane, r);
  while (tile < tiles) {
    tile_to_lds<L>(lds32, W, lane, r);
    const int64_t next = tile + stride;
    if (next < tiles) tile_load<L, IPS_DECODE_NT_LOADS>(enc, next, W, total_words, lane, r);
    wave_lds_fence();

    uint32_t p[W];
    planes_from_lds<W>(lds32, lane, p);
    const int64_t row_base = tile * kRowsPerTile;
    if constexpr (kPackedOut) {
      constexpr int R = LaneWidth<W>::R;             // 8 or 16 parked registers per lane
      constexpr int kStride = packed_lane_stride(R);  // 48 / 80 bytes
      uint32_t a[32];
      planes_to_lanes<W>(p, a);
      wave_lds_fence();  // all plane reads precede the overwrite of the same LDS region
      uint8_t* lds8 = reinterpret_cast<uint8_t*>(lds32);
#pragma unroll
      for (int i = 0; i < R / 4; ++i) {
        const u32x4 t = {a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
        *reinterpret_cast<u32x4*>(lds8 + lane * kStride + 16 * i) = t;
      }
      wave_lds_fence();
      constexpr uint32_t kMask = (1u << R) - 1u;
      int bad = 0;
      GT* dst_all = reinterpret_cast<GT*>(out);
      if constexpr (sizeof(GT) == 4) {
        // rows 4q..4q+3 of a lane sit at bit positions pos = 31-4q .. 28-4q: field pos / R of the
        // registers pos % R, four neighbouring registers of one aligned group
        const int q = lane & 7;
        const int pos0 = 31 - 4 * q;
        const int grp = (pos0 % R) / 4;     // registers 4 grp .. 4 grp + 3
        const uint32_t fsh = (uint32_t)(R * (pos0 / R));
        if constexpr (kSharedDict && TAIL) {
          // A dictionary whose tail stays in L2: with one wave per SIMD nothing but the wave's own
          // loads in flight hides an L2 round trip, and a look-up that falls into the tail under a
          // branch of its own makes the wave wait 32 times per sub-tile.  All 32 codes are looked up in
          // the LDS part first, then the tail loads of the whole sub-tile are issued together.
          uint32_t xs[8][4];
          GT ys[8][4];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int src = 8 * k + (lane >> 3);
            const u32x4 t = *reinterpret_cast<const u32x4*>(lds8 + src * kStride + 16 * grp);
            xs[k][0] = (t.w >> fsh) & kMask; xs[k][1] = (t.z >> fsh) & kMask;
            xs[k][2] = (t.y >> fsh) & kMask; xs[k][3] = (t.x >> fsh) & kMask;
            const int64_t valid = n_rows - (row_base + 4 * (64 * k + lane));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (e >= valid) xs[k][e] = 0u;                       // (rows that do not exist: entry 0, never stored)
              else if (xs[k][e] >= dict_entries) { bad = 1; xs[k][e] = 0u; }
              ys[k][e] = dict_dyn[xs[k][e] < lds_entries ? xs[k][e] : 0u];
            }
          }
#pragma unroll
          for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (xs[k][e] >= lds_entries) ys[k][e] = dict[xs[k][e]];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int64_t row = row_base + 4 * (64 * k + lane);
            const int64_t valid = n_rows - row;
            if (valid >= 4) {
              const u32x4 o = {(uint32_t)ys[k][0], (uint32_t)ys[k][1], (uint32_t)ys[k][2], (uint32_t)ys[k][3]};
              IPS_STREAM_STORE16(dst_all + row, o);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (e < valid) dst_all[row + e] = ys[k][e];
            }
          }
        } else
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int src = 8 * k + (lane >> 3);
          const u32x4 t = *reinterpret_cast<const u32x4*>(lds8 + src * kStride + 16 * grp);
          // descending registers = ascending rows
          const uint32_t x[4] = {(t.w >> fsh) & kMask, (t.z >> fsh) & kMask, (t.y >> fsh) & kMask, (t.x >> fsh) & kMask};
          const int64_t row = row_base + 4 * (64 * k + lane);
          const int64_t valid = n_rows - row;
          GT y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = e < valid;
            if (ok && x[e] >= dict_entries) bad = 1;
            y[e] = (ok && x[e] < dict_entries) ? lookup(x[e]) : GT(0);
          }
          if (valid >= 4) {
            const u32x4 o = {(uint32_t)y[0], (uint32_t)y[1], (uint32_t)y[2], (uint32_t)y[3]};
            IPS_STREAM_STORE16(dst_all + row, o);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < valid && x[e] < dict_entries) dst_all[row + e] = y[e];
          }
        }
      } else {
        // 8-byte entries: rows 2u, 2u+1 of a lane (u = lane % 16): positions 31-2u, 30-2u = the
        // same field of two neighbouring registers
        const int u = lane & 15;
        const int pos0 = 31 - 2 * u;
        const int reg_lo = (pos0 - 1) % R;  // even register of the pair
        const uint32_t fsh = (uint32_t)(R * (pos0 / R));
        if constexpr (kSharedDict && TAIL) {  // (as above: LDS look-ups first, the tail's loads together)
          uint32_t xs[16][2];
          GT ys[16][2];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int src = 4 * k + (lane >> 4);
            const u32x2 t = *reinterpret_cast<const u32x2*>(lds8 + src * kStride + 4 * reg_lo);
            xs[k][0] = (t.y >> fsh) & kMask;
            xs[k][1] = (t.x >> fsh) & kMask;
            const int64_t valid = n_rows - (row_base + 2 * (64 * k + lane));
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              if (e >= valid) xs[k][e] = 0u;
              else if (xs[k][e] >= dict_entries) { bad = 1; xs[k][e] = 0u; }
              ys[k][e] = dict_dyn[xs[k][e] < lds_entries ? xs[k][e] : 0u];
            }
          }
#pragma unroll
          for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e)
              if (xs[k][e] >= lds_entries) ys[k][e] = dict[xs[k][e]];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int64_t row = row_base + 2 * (64 * k + lane);
            const int64_t valid = n_rows - row;
            if (valid >= 2) {
              const u32x4 o = {(uint32_t)ys[k][0], (uint32_t)((uint64_t)ys[k][0] >> 32), (uint32_t)ys[k][1], (uint32_t)((uint64_t)ys[k][1] >> 32)};
              IPS_STREAM_STORE16(dst_all + row, o);
            } else if (valid == 1) {
              dst_all[row] = ys[k][0];
            }
          }
        } else
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int src = 4 * k + (lane >> 4);
          const u32x2 t = *reinterpret_cast<const u32x2*>(lds8 + src * kStride + 4 * reg_lo);
          const uint32_t x[2] = {(t.y >> fsh) & kMask, (t.x >> fsh) & kMask};
          const int64_t row = row_base + 2 * (64 * k + lane);
          const int64_t valid = n_rows - row;
          GT y[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const bool ok = e < valid;
            if (ok && x[e] >= dict_entries) bad = 1;
            y[e] = (ok && x[e] < dict_entries) ? lookup(x[e]) : GT(0);
          }
          if (valid >= 2) {
            const u32x4 o = {(uint32_t)y[0], (uint32_t)((uint64_t)y[0] >> 32), (uint32_t)y[1], (uint32_t)((uint64_t)y[1] >> 32)};
            IPS_STREAM_STORE16(dst_all + row, o);
          } else if (valid == 1 && x[0] < dict_entries) {
            dst_all[row] = y[0];
          }
        }
      }
      if (G != 0 && bad && bad_index) *bad_index = 1;
      wave_lds_fence();  // LDS region is reused by the next sub-tile
      tile = next;
      continue;
    }
    uint32_t v[32];
    planes_to_values<W>(p, v);
    wave_lds_fence();
    values_to_row_tile(lds32, lane, v);
    wave_lds_fence();

    if (G != 0) {
      typename GatherT<G>::type* dst = reinterpret_cast<typename GatherT<G>::type*>(out);
      int bad = 0;
#pragma unroll 4
      for (int i = 0; i < 32; ++i) {
        int rho = i * kWave + lane;
        if (row_base + rho < n_rows) {
          uint32_t code = lds32[row_tile_dw(rho)];
          if (code < dict_entries) dst[row_base + rho] = lookup(code); else bad = 1;
        }
      }
      if (bad && bad_index) *bad_index = 1;
    } else if (OW == 4) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(out) + row_base;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int rho = 4 * (i * kWave + lane);
        u32x4 t = *reinterpret_cast<const u32x4*>(lds32 + row_tile_dw(rho));
        int64_t valid = n_rows - (row_base + rho);
        if (valid >= 4) {
          IPS_STREAM_STORE16(dst + rho, t);
        } else {
          if (valid > 0) dst[rho] = t.x;
          if (valid > 1) dst[rho + 1] = t.y;
          if (valid > 2) dst[rho + 2] = t.z;
        }
      }
    } else if (OW == 2) {
      uint16_t* dst = reinterpret_cast<uint16_t*>(out) + row_base;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int rho = 8 * (i * kWave + lane);
        u32x4 a = *reinterpret_cast<const u32x4*>(lds32 + row_tile_dw(rho));
        u32x4 b = *reinterpret_cast<const u32x4*>(lds32 + row_tile_dw(rho + 4));
        int64_t valid = n_rows - (row_base + rho);
        if (valid >= 8) {
          u32x4 t = {a.x | (a.y << 16), a.z | (a.w << 16), b.x | (b.y << 16), b.z | (b.w << 16)};
          IPS_STREAM_STORE16(dst + rho, t);
        } else {
          uint32_t e[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < valid) dst[rho + j] = (uint16_t)e[j];
        }
      }
    } else {
      uint8_t* dst = reinterpret_cast<uint8_t*>(out) + row_base;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int rho = 16 * (i * kWave + lane);
        u32x4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          q[j] = *reinterpret_cast<const u32x4*>(lds32 + row_tile_dw(rho + 4 * j));
        int64_t valid = n_rows - (row_base + rho);
        if (valid >= 16) {
          u32x4 t;
          t.x = q[0].x | (q[0].y << 8) | (q[0].z << 16) | (q[0].w << 24);
          t.y = q[1].x | (q[1].y << 8) | (q[1].z << 16) | (q[1].w << 24);
          t.z = q[2].x | (q[2].y << 8) | (q[2].z << 16) | (q[2].w << 24);
          t.w = q[3].x | (q[3].y << 8) | (q[3].z << 16) | (q[3].w << 24);
          IPS_STREAM_STORE16(dst + rho, t);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (4 * j + 0 < valid) dst[rho + 4 * j + 0] = (uint8_t)q[j].x;
            if (4 * j + 1 < valid) dst[rho + 4 * j + 1] = (uint8_t)q[j].y;
            if (4 * j + 2 < valid) dst[rho + 4 * j + 2] = (uint8_t)q[j].z;
            if (4 * j + 3 < valid) dst[rho + 4 * j + 3] = (uint8_t)q[j].w;
          }
        }
      }
    }
    wave_lds_fence();
    tile = next;
  }
}

// ---------------------------------------------------------------------------------------------
// Encode: FleEncoder::Put x n + Flush (fle-encoding.h:8315-8365, 9806-9812; Pack_w :8367-9803).
// IW = bytes per input value.  Padding rows of the last block are encoded as zero.
// ---------------------------------------------------------------------------------------------
template <int W, int IW>
__global__ __launch_bounds__(kThreads, IPS_MIN_WAVES_PER_EU) void fle_encode_kernel(const void* __restrict__ values,
                                                              int64_t n_rows,
                                                              uint64_t* __restrict__ enc) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerBlock * kRowTileBytes / 4];
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * (kRowTileBytes / 4);

  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t total_words = ((n_rows + 63) / 64) * W;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  constexpr int RPL = 16 / IW;      // rows per 16-byte load
  constexpr int NL = 32 / RPL;      // loads per lane per sub-tile

  for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < tiles; tile += stride) {
    const int64_t row_base = tile * kRowsPerTile;
    // 1. values HBM -> row tile in LDS (coalesced 16-byte loads)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int rho = RPL * (i * kWave + lane);
      int64_t valid = n_rows - (row_base + rho);
      uint32_t e[RPL];
      if (valid >= RPL) {
        u32x4 t = stream_load<IPS_ENCODE_NT_LOADS>(reinterpret_cast<const u32x4*>(
            reinterpret_cast<const uint8_t*>(values) + (row_base + rho) * IW));
        uint32_t tw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
          if (IW == 4) e[j] = tw[j];
          else if (IW == 2) e[j] = (tw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
          else e[j] = (tw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        }
      } else {
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
          uint32_t x = 0;
          if (j < valid) {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(values) +
                                 (row_base + rho + j) * IW;
            if (IW == 4) x = *reinterpret_cast<const uint32_t*>(src);
            else if (IW == 2) x = *reinterpret_cast<const uint16_t*>(src);
            else x = *src;
          }
          e[j] = x;
        }
      }
#pragma unroll
      for (int j = 0; j < RPL; j += 4) {
        u32x4 t = {e[j], e[j + 1], e[j + 2], e[j + 3]};
        *reinterpret_cast<u32x4*>(lds32 + row_tile_dw(rho + j)) = t;
      }
    }
    wave_lds_fence();
    // 2. lane-per-half-block: 32 values -> W plane halves
    uint32_t v[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u32x4 t = *reinterpret_cast<const u32x4*>(lds32 + lane * kRowTileStrideDw + 4 * i);
      v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
    uint32_t p[W];
    values_to_planes<W>(v, p);
    wave_lds_fence();
    // 3. plane halves -> LDS plane image
    {
      uint32_t* dst = lds32 + plane_base_dw(W, lane);
#pragma unroll
      for (int k = 0; k < W; ++k) dst[2 * k] = p[k];
    }
    wave_lds_fence();
    // 4. LDS plane image -> HBM, linear 16-byte stores
    const int64_t w0 = tile * (int64_t)(kBlocksPerTile * W);
    const int64_t left = total_words - w0;
    constexpr int L = (16 * W + kWave - 1) / kWave;
    constexpr int STRIDE = W | 1;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      int c = i * kWave + lane;
      if (c < 16 * W && 2 * c < left) {
        u32x4 t;
        if (W & 1) {
          t = *reinterpret_cast<const u32x4*>(lds32 + 4 * c);
        } else {
          int wi = 2 * c;
          int blk = wi / W;
          int k = wi - blk * W;
          const uint32_t* src = lds32 + 2 * (blk * STRIDE + k);
          u32x2 lo = *reinterpret_cast<const u32x2*>(src);
          u32x2 hi = *reinterpret_cast<const u32x2*>(src + 2);
          t.x = lo.x; t.y = lo.y; t.z = hi.x; t.w = hi.y;
        }
        if (2 * c + 1 < left) {
          IPS_STREAM_STORE16(enc + w0 + 2 * c, t);
        } else {
          u32x2 h = {t.x, t.y};
          *reinterpret_cast<u32x2*>(enc + w0 + 2 * c) = h;
        }
      }
    }
    wave_lds_fence();
  }
}

}  // namespace ips
