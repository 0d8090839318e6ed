// ips_plain.hip -- PLAIN fixed-width pages: ParquetPlainEncoder::Eq/Lt/Le/Gt/Ge (parquet-common.h:
// 197-250, int8 :335-383, int16 :400-449), the fused scan (EvalSimplePredicates + ReadValue(skip)
// on one PLAIN column, hdfs-parquet-scanner.cc:1837-1865, 1006-1027) and ReadValue(skip) against
// a given selection (parquet-common.h:186-190).  bit = x OP literal (SQL order; the REFERENCE order
// is obtained by the caller swapping LT<->GT, LE<->GE).
//
// Round 2, second version.  The first one kept the rows where the coalesced 16-byte loads put them
// (lane l holds rows RPL*l.. of a 64*RPL-row chunk) and rebuilt row order with ballots: two ballots
// per comparison and chunk plus a second ballot round to re-interleave them into bitmap words, and
// one small store burst per chunk for the selected slots -- 611 / 763 VALU per 2048 rows (pred /
// scan) and a dependency chain through the scalar unit per chunk.  Now the page goes the way the
// FLE planes go:
//   HBM -> VGPR (8 x 16-byte nt loads per lane = an 8 KiB tile, the next tile prefetched in
//   registers) -> LDS, laid out so that lane l finds ITS 128 consecutive bytes (32 four-byte or 16
//   eight-byte rows) at l * 144 (the 16-byte pad makes the ds_read_b128 of 16 lanes hit 64
//   different banks) -> the lane compares its rows and holds their bits as one mask, LSB = first
//   row: the bitmap dword itself.  64 lanes store 256 (128) contiguous bitmap bytes.
// Materialisation is the index list of the FLE scan: a DPP prefix sum of the popcounts, phase A
// appends one 16-bit entry (lane, row) per selected row at its prefix position, phase B turns 64
// entries per round into one coalesced store of 64 slots read back from the tile image in LDS.
// Lists longer than 512 entries are worked off in windows of 512, each lane resuming where it
// stopped, so one path serves every selectivity.
#include <string.h>

#include "ips_chunk_host.h"

namespace ips {

template <typename T>
struct PlainLit {
  T v[16];
  int32_t n;
  int32_t combine;  // 0 set, 1 and-into, 2 or-into the bitmap
  int32_t join;     // 0 none; 1 / 2: AND / OR with (x op2 v2) in the same pass
  int32_t op2;
  T v2;
  uint32_t* edges;  // paged launches, 4-byte slots: the chunk's edge slots (edge mode) or NULL
};

// T = compared type, S = slot type (uint32_t for 4-byte slots, uint64_t for 8-byte slots)
template <typename T, typename S>
__device__ __forceinline__ T slot_value(S raw) {
  if constexpr (sizeof(T) == sizeof(S)) {
    T t;
    __builtin_memcpy(&t, &raw, sizeof(T));
    return t;
  } else {
    return (T)(int32_t)raw;  // int8/int16: low bytes of the 4-byte slot, sign-extended by the cast
  }
}

constexpr int kPlainTileBytes = 8192;
constexpr int kPlainLaneBytes = kPlainTileBytes / kWave;           // 128
constexpr int kPlainLaneStride = kPlainLaneBytes + 16;             // 144
constexpr int kPlainImageBytes = kWave * kPlainLaneStride;         // 9216 (also holds the 2048 + 64 dwords of the dense compaction image)
static_assert(kPlainImageBytes >= (2048 + 64) * 4, "the padded compaction image fits the tile image");
constexpr int kPlainListMax = 512;
constexpr int kPlainWaveBytes = kPlainImageBytes + 2 * kPlainListMax;  // 10240
constexpr int kPlainLoads = kPlainTileBytes / (16 * kWave);        // 8 x 16 bytes per lane
static_assert(kPlainLoads == 8, "owner / piece arithmetic below");

template <typename S>
struct PlainGeom {
  static constexpr int R = kPlainLaneBytes / (int)sizeof(S);  // rows per lane: 32 / 16
  static constexpr int RT = kWave * R;                          // rows per tile: 2048 / 1024
  static constexpr int TPB = kRowsPerTile / RT;                 // tiles per 2048-row batch: 1 / 2
  static constexpr int RPL = 16 / (int)sizeof(S);               // rows per 16-byte load
};

// the tile's bytes as the lane's 8 coalesced 16-byte loads (piece i of the wave = bytes
// [i * 1024, (i + 1) * 1024) of the tile); rows at or beyond n_rows read as 0
template <typename S>
__device__ __forceinline__ void plain_tile_load(const S* __restrict__ page, int64_t tile, int64_t n_rows,
                                                int lane, u32x4 (&r)[kPlainLoads]) {
  using G = PlainGeom<S>;
  const int64_t row0 = tile * G::RT;
  // a buffer resource over the tile's bytes that exist: dwords beyond them read as 0 (ips_device.h)
  int64_t left = (n_rows - row0) * (int64_t)sizeof(S);
  left = left < 0 ? 0 : (left > kPlainTileBytes ? kPlainTileBytes : left);
  const S* base = left > 0 ? page + row0 : page;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<S*>(base), 0, (int)left, kBufferRsrcDword3);
#pragma unroll
  for (int i = 0; i < kPlainLoads; ++i) r[i] = buffer_load16<true>(rsrc, (uint32_t)(i * kWave + lane) * 16u);
}

// 16-byte piece c of the tile belongs to lane c / 8, piece c % 8 of its 128 bytes
__device__ __forceinline__ void plain_tile_stage(uint8_t* lds, int lane, const u32x4 (&r)[kPlainLoads]) {
#pragma unroll
  for (int i = 0; i < kPlainLoads; ++i)
    *reinterpret_cast<u32x4*>(lds + (i * 8 + (lane >> 3)) * kPlainLaneStride + (lane & 7) * 16) = r[i];
}

template <typename T, int OP>
__device__ __forceinline__ bool plain_cmp1(T x, const T* __restrict__ v, int n) {
  if (OP == 0) return x == v[0];
  if (OP == 1) return x < v[0];
  if (OP == 2) return x <= v[0];
  if (OP == 3) return x > v[0];
  if (OP == 4) return x >= v[0];
  bool f = false;
  for (int j = 0; j < n; ++j) f = f || (x == v[j]);
  return f;
}

// bit k of the result: row k of the lane's rows satisfies (x OP v)
template <typename T, typename S, int OP>
__device__ __forceinline__ uint32_t plain_lane_mask_op(const uint8_t* lds, int lane, const T* __restrict__ v, int n) {
  using G = PlainGeom<S>;
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < kPlainLoads; ++k) {
    const u32x4 q = *reinterpret_cast<const u32x4*>(lds + lane * kPlainLaneStride + 16 * k);
    S e[G::RPL];
    __builtin_memcpy(e, &q, 16);
#pragma unroll
    for (int j = 0; j < G::RPL; ++j)
      m |= (plain_cmp1<T, OP>(slot_value<T, S>(e[j]), v, n) ? 1u : 0u) << (k * G::RPL + j);
  }
  return m;
}

template <typename T, typename S>
__device__ __forceinline__ uint32_t plain_lane_mask(const uint8_t* lds, int lane, int op, const T* __restrict__ v, int n) {
  switch (op) {  // wave-uniform
    case 0: return plain_lane_mask_op<T, S, 0>(lds, lane, v, n);
    case 1: return plain_lane_mask_op<T, S, 1>(lds, lane, v, n);
    case 2: return plain_lane_mask_op<T, S, 2>(lds, lane, v, n);
    case 3: return plain_lane_mask_op<T, S, 3>(lds, lane, v, n);
    case 4: return plain_lane_mask_op<T, S, 4>(lds, lane, v, n);
    default: return plain_lane_mask_op<T, S, 5>(lds, lane, v, n);
  }
}

// rows of the lane that exist: tile row0 + R * lane + k < n_rows
template <typename S>
__device__ __forceinline__ uint32_t plain_valid_mask(int64_t tile, int lane, int64_t n_rows) {
  using G = PlainGeom<S>;
  const int64_t valid = n_rows - (tile * G::RT + (int64_t)lane * G::R);
  const uint32_t all = G::R == 32 ? ~0u : ((1u << (G::R & 31)) - 1u);
  if (valid >= G::R) return all;
  return valid <= 0 ? 0u : ((1u << valid) - 1u);
}

// bitmap dword index and value of this lane for the tile: 4-byte slots: one dword per lane;
// 8-byte slots: the 16 bits of lanes 2k and 2k+1 make dword k (held by the even lane)
template <typename S>
__device__ __forceinline__ uint32_t plain_pair_dword(uint32_t m) {
  if (PlainGeom<S>::R == 32) return m;
  const uint32_t odd = (uint32_t)__builtin_amdgcn_mov_dpp((int)m, 0xF5, 0xF, 0xF, true);  // quad_perm [1,1,3,3]
  return m | (odd << 16);
}

// The selected rows of the tile (mask m per lane, P = rows selected in lower lanes, count = in the
// whole tile) leave for dst[0 .. count) in row order.
template <typename S>
__device__ __forceinline__ void plain_materialise(const uint8_t* lds, uint16_t* list, int lane, uint32_t m,
                                                  uint32_t P, uint32_t count, S* __restrict__ dst) {
  using G = PlainGeom<S>;
  if (count == (uint32_t)G::RT) {
    // every row of the tile selected (wave-uniform): slot i is row i, no list is needed
    // (int32, every row selected, 2^28 rows: 567 -> 397 us; phase A was 4 windows of 8+ rounds)
#pragma unroll 1  // (unrolled it takes 64 more registers and the kernel half its waves)
    for (uint32_t i = lane; i < (uint32_t)G::RT; i += kWave)
      dst[i] = *reinterpret_cast<const S*>(lds + (i / (uint32_t)G::R) * kPlainLaneStride + (i % (uint32_t)G::R) * sizeof(S));
    return;
  }
  if (count > (sizeof(S) == 8 ? (uint32_t)G::RT / IPS_PLAIN_DENSE8 : (uint32_t)G::RT / 4u)) {
    // Dense tiles (more than a quarter of the rows selected, wave-uniform): the lane takes its 128
    // bytes into registers, every lane appends the dwords of its selected rows behind those of the
    // lanes before it in the (padded) compaction image, and the image leaves as whole 16-byte
    // stores -- the FLE scan's dense path.  The index list would need three windows of up to 32
    // ballot-controlled rounds each at 60 % (int32 twin of configs[1]: 397 -> 3xx us).
    uint32_t v[32];
#pragma unroll
    for (int k = 0; k < kPlainLoads; ++k) {
      const u32x4 q = *reinterpret_cast<const u32x4*>(lds + lane * kPlainLaneStride + 16 * k);
      v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
    }
    constexpr uint32_t DPS = sizeof(S) / 4;  // dwords per slot
    uint32_t md = m;                          // one mask bit per dword of the lane
    if (DPS == 2) {                           // bit j -> bits 2j, 2j + 1
      uint32_t x = m & 0xFFFFu;
      x = (x | (x << 8)) & 0x00FF00FFu;
      x = (x | (x << 4)) & 0x0F0F0F0Fu;
      x = (x | (x << 2)) & 0x33333333u;
      x = (x | (x << 1)) & 0x55555555u;
      md = x | (x << 1);
    }
    wave_lds_fence();  // every lane has its rows in registers: the image becomes the compaction image
    uint32_t* lds32 = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(lds));
    compact_lane_values(lds32, md, P * DPS, v);
    wave_lds_fence();
    const uint32_t n_dw = count * DPS;
    uint32_t* out = reinterpret_cast<uint32_t*>(dst);
    if ((reinterpret_cast<uintptr_t>(out) & 15u) == 0u) {  // (wave-uniform: the second tile of an 8-byte batch starts anywhere)
      store_compacted(lds32, n_dw, out, lane);
    } else {
      for (uint32_t e = lane; e < n_dw; e += kWave) out[e] = lds32[compact_dw(e)];
    }
    wave_lds_fence();
    return;
  }
  uint32_t pos = P;
  const uint32_t lane5 = (uint32_t)lane << 5;
  for (uint32_t win0 = 0; win0 < count; win0 += kPlainListMax) {  // wave-uniform
    const uint32_t win1 = win0 + kPlainListMax;
    // phase A: the lane's entries whose positions fall into the window
    while (__builtin_amdgcn_ballot_w64(m != 0u && pos < win1) != 0ull) {
      if (m != 0u && pos < win1) {
        list[pos - win0] = (uint16_t)(lane5 | (uint32_t)__builtin_ctz(m));
        m &= m - 1u;
        ++pos;
      }
    }
    wave_lds_fence();
    // phase B: 64 entries per round -> 64 consecutive slots
    const uint32_t nwin = count - win0 < (uint32_t)kPlainListMax ? count - win0 : (uint32_t)kPlainListMax;
    const uint32_t rounds = (nwin + kWave - 1) / kWave;
    for (uint32_t rd = 0; rd < rounds; ++rd) {
      const uint32_t i = rd * kWave + lane;
      if (i >= nwin) continue;
      const uint32_t e = list[i];
      if (IPS_PLAIN_ABLATE != 1) dst[win0 + i] = *reinterpret_cast<const S*>(lds + (e >> 5) * kPlainLaneStride + (e & 31u) * sizeof(S));
    }
    wave_lds_fence();  // the list is rewritten by the next window
  }
}

// the tile that follows 'tile' in a wave's sequence: the wave takes whole 2048-row batches
template <typename S>
__device__ __forceinline__ int64_t plain_next_tile(int64_t tile, int64_t waves) {
  using G = PlainGeom<S>;
  if (G::TPB == 2 && (tile & 1) == 0) return tile + 1;
  return (tile / G::TPB + waves) * G::TPB;
}

// SCAN = false: predicate only (bitmap set / and-ed / or-ed).  SCAN = true: bitmap + the selected
// rows' slots per 2048-row batch + batch counts.
// PAGED: page / n_rows are one page of a column chunk, the bitmap is the chunk's (the lane dwords go
// through the page's window, ips_chunk_device.h), the batch outputs start at the page's first slot.
// GIVEN (paged scans only): no comparison -- the lane's rows are selected by a bitmap over the chunk's
// rows (ReadValue(skip) against a selection, parquet-common.h:186-190), nothing is written to it.
template <typename T, typename S, bool SCAN, bool PAGED, bool GIVEN = false>
__device__ __forceinline__ void plain_tile_body(const S* __restrict__ page, int64_t n_rows, int op,
                                                const PlainLit<T>& lit, uint32_t* __restrict__ bitmap32,
                                                S* __restrict__ batch_values, uint32_t* __restrict__ batch_counts,
                                                const BitmapWindow* win, const uint32_t* __restrict__ given32 = nullptr,
                                                const ChunkPage* pg = nullptr, int64_t total_dwords = 0) {
  using G = PlainGeom<S>;
  __shared__ __attribute__((aligned(16))) uint8_t lds_all[kWavesPerBlock * kPlainWaveBytes];
  const int lane = lane_id();
  const int wave = wave_id();
  uint8_t* lds = lds_all + wave * kPlainWaveBytes;
  uint16_t* list = reinterpret_cast<uint16_t*>(lds + kPlainImageBytes);
  const int64_t n_tiles = (n_rows + G::RT - 1) / G::RT;
  const int64_t waves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t bm_dwords = bitmap_dwords(n_rows);

  int64_t tile = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * G::TPB;
  int64_t batch_step = waves;
  int64_t tiles_end = n_tiles;
  [[maybe_unused]] WindowCarry carry;
  if constexpr (PAGED) {  // this wave's share of the page's batches (contiguous when the window is shifted)
    const TileShare sh = tile_share(*win, (n_rows + kRowsPerTile - 1) / kRowsPerTile,
                                    (int64_t)blockIdx.x * kWavesPerBlock + wave, waves);
    tile = sh.first * G::TPB;
    batch_step = sh.step;
    tiles_end = sh.end * G::TPB < n_tiles ? sh.end * G::TPB : n_tiles;
  }
  // the dword an AND-into / OR-into launch combines with is prefetched with its tile (fle_pred_body)
  auto combine_operand = [&](int64_t t) -> uint32_t {
    const int64_t d = G::R == 32 ? t * 64 + lane : t * 32 + (lane >> 1);
    if constexpr (GIVEN) return 0u;
    if constexpr (PAGED) return (G::R == 32 && window_has_operand(*win, lit.combine)) ? window_operand(*win, d) : 0u;
    return (lit.combine != 0 && (G::R == 32 || (lane & 1) == 0) && d < bm_dwords) ? bitmap32[d] : 0u;
  };
  u32x4 r[kPlainLoads];
  uint32_t old = 0u;
  if (tile < tiles_end) {
    old = combine_operand(tile);
    plain_tile_load<S>(page, tile, n_rows, lane, r);
  }
  uint32_t base = 0;  // rows of the batch selected in its earlier tile (8-byte slots)
  while (tile < tiles_end) {
    plain_tile_stage(lds, lane, r);
    const int64_t next = plain_next_tile<S>(tile, batch_step);
    const uint32_t old_now = old;
    if (next < tiles_end) {  // register prefetch
      old = combine_operand(next);
      plain_tile_load<S>(page, next, n_rows, lane, r);
    }
    wave_lds_fence();

    uint32_t m;
    if constexpr (GIVEN) {
      if (G::R == 32) {
        m = window_fetch_at(given32, total_dwords, *pg, tile * 64 + lane, true);
      } else {  // 1024-row tile: lanes 0..31 fetch its 32 dwords, lane l takes half of the dword of lane l / 2
        const uint32_t dw = window_fetch_at(given32, total_dwords, *pg, tile * 32 + lane, lane < 32);
        const uint32_t mine = (uint32_t)__builtin_amdgcn_ds_bpermute((lane >> 1) << 2, (int)dw);
        m = (mine >> (16 * (lane & 1))) & 0xFFFFu;
      }
    } else {
      m = plain_lane_mask<T, S>(lds, lane, op, lit.v, lit.n);
      if (lit.join != 0) {
        const uint32_t m2 = plain_lane_mask<T, S>(lds, lane, lit.op2, &lit.v2, 1);
        m = lit.join == 1 ? (m & m2) : (m | m2);
      }
    }
    m &= plain_valid_mask<S>(tile, lane, n_rows);

    uint32_t bm = plain_pair_dword<S>(m);
    const int64_t d = G::R == 32 ? tile * 64 + lane : tile * 32 + (lane >> 1);
    if constexpr (GIVEN) {
      (void)bm;
    } else if constexpr (PAGED) {
      if (G::R == 32) {
        window_emit(*win, carry, d, bm, lit.combine, kWave - 1, window_has_operand(*win, lit.combine), old_now);
      } else {  // dword k of the tile sits in lane 2k: bring it to lane k, 32 dwords per tile
        const uint32_t mine = (uint32_t)__builtin_amdgcn_ds_bpermute((2 * lane) << 2, (int)bm);
        window_emit(*win, carry, tile * 32 + lane, mine, lit.combine, 31);
      }
    } else if ((G::R == 32 || (lane & 1) == 0) && d < bm_dwords) {
      if (lit.combine == 1) bm &= old_now;
      else if (lit.combine == 2) bm |= old_now;
      IPS_BITMAP_STORE(bitmap32 + d, bm);
    }
    if (SCAN) {
      uint32_t count = 0;
      if (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {  // wave-uniform
        const uint32_t mine = (uint32_t)__builtin_popcount(m);
        const uint32_t incl = wave_inclusive_scan(mine);
        count = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (IPS_PLAIN_ABLATE != 2) plain_materialise<S>(lds, list, lane, m, incl - mine, count,
                             batch_values + (tile / G::TPB) * kRowsPerTile + base);
      }
      const bool last_of_batch = G::TPB == 1 || (tile & 1) != 0 || tile + 1 >= n_tiles;
      if (last_of_batch) {
        if (lane == 0) batch_counts[tile / G::TPB] = base + count;
        base = 0;
      } else {
        base = count;
      }
    }
    wave_lds_fence();  // the image is rewritten by the next tile
    tile = next;
  }
  if constexpr (PAGED) window_flush(*win, carry, lit.combine);
}

// (four waves per SIMD is what the 40-KiB workgroups allow; the dense path's 32 row registers must
// not cost the scan one of them)
template <typename T, typename S, bool SCAN>
__global__ __launch_bounds__(kThreads, 4) void plain_tile_kernel(const S* __restrict__ page, int64_t n_rows, int op,
                                                              PlainLit<T> lit, uint32_t* __restrict__ bitmap32,
                                                              S* __restrict__ batch_values,
                                                              uint32_t* __restrict__ batch_counts) {
  plain_tile_body<T, S, SCAN, false>(page, n_rows, op, lit, bitmap32, batch_values, batch_counts, nullptr);
}

template <typename T, typename S, bool SCAN>
__global__ __launch_bounds__(kThreads, 4) void plain_tile_pages_kernel(const ChunkPage* __restrict__ pages, int64_t chunk_rows,
                                                                    int op, PlainLit<T> lit, uint32_t* __restrict__ bitmap32,
                                                                    S* __restrict__ batch_values,
                                                                    uint32_t* __restrict__ batch_counts) {
  const ChunkPage pg = pages[blockIdx.y];
  const BitmapWindow win = bitmap_window(bitmap32, pg, chunk_rows, nullptr, sizeof(S) == 4 ? lit.edges : nullptr);
  plain_tile_body<T, S, SCAN, true>(reinterpret_cast<const S*>(pg.data), pg.n_rows, op, lit, nullptr,
                                    SCAN ? batch_values + (int64_t)pg.batch0 * kRowsPerTile : nullptr,
                                    SCAN ? batch_counts + pg.batch0 : nullptr, &win);
}

// Late materialisation of a PLAIN chunk against a selection over the chunk's rows (all pages, one launch)
template <typename S>
__global__ __launch_bounds__(kThreads, 4) void plain_select_pages_kernel(const ChunkPage* __restrict__ pages, int64_t chunk_rows,
                                                                         const uint32_t* __restrict__ given32,
                                                                         S* __restrict__ batch_values,
                                                                         uint32_t* __restrict__ batch_counts) {
  const ChunkPage pg = pages[blockIdx.y];
  BitmapWindow win = bitmap_window(nullptr, pg, chunk_rows);
  win.shift = 0u;  // (nothing is emitted: every wave takes every n-th batch, whatever the page's offset)
  PlainLit<S> lit;
  __builtin_memset(&lit, 0, sizeof(lit));
  plain_tile_body<S, S, true, true, true>(reinterpret_cast<const S*>(pg.data), pg.n_rows, 0, lit, nullptr,
                                          batch_values + (int64_t)pg.batch0 * kRowsPerTile, batch_counts + pg.batch0, &win,
                                          given32, &pg, bitmap_dwords(chunk_rows));
}

ips_status launch_plain_select_pages(int stride_bytes, const ChunkPage* d_pages, int n_pages, int64_t max_rows,
                                     int64_t chunk_rows, const uint64_t* bitmap, void* batch_values,
                                     uint32_t* batch_counts, hipStream_t s) {
  const int64_t max_batches = (max_rows + kRowsPerTile - 1) / kRowsPerTile;
  const uint32_t* g = reinterpret_cast<const uint32_t*>(bitmap);
  if (stride_bytes == 4) {
    auto kern = plain_select_pages_kernel<uint32_t>;
    const int gx = paged_grid_x(reinterpret_cast<const void*>(kern), max_batches, n_pages);
    if (gx <= 0) return IPS_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)n_pages), dim3(kThreads), 0, s, d_pages, chunk_rows, g,
                       reinterpret_cast<uint32_t*>(batch_values), batch_counts);
  } else {
    auto kern = plain_select_pages_kernel<uint64_t>;
    const int gx = paged_grid_x(reinterpret_cast<const void*>(kern), max_batches, n_pages);
    if (gx <= 0) return IPS_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)n_pages), dim3(kThreads), 0, s, d_pages, chunk_rows, g,
                       reinterpret_cast<uint64_t*>(batch_values), batch_counts);
  }
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

template <typename T, typename S, bool SCAN>
static ips_status launch_plain_tiles(const void* page, int64_t n_rows, int op, const void* literals,
                                     int n_literals, int combine, int join, int op2, const void* literal2,
                                     uint64_t* bitmap, void* batch_values, uint32_t* batch_counts,
                                     hipStream_t s) {
  PlainLit<T> lit;
  lit.n = n_literals;
  lit.combine = combine;
  lit.join = join;
  lit.op2 = op2;
  lit.v2 = literal2 ? *reinterpret_cast<const T*>(literal2) : T();
  lit.edges = nullptr;
  for (int i = 0; i < 16; ++i) lit.v[i] = i < n_literals ? reinterpret_cast<const T*>(literals)[i] : T();
  auto kern = plain_tile_kernel<T, S, SCAN>;
  const int64_t n_batches = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int grid = grid_for_tiles(reinterpret_cast<const void*>(kern), n_batches);  // a wave per batch (predicate only: 4x..64x measure alike)
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, s, reinterpret_cast<const S*>(page), n_rows, op, lit,
                     reinterpret_cast<uint32_t*>(bitmap), reinterpret_cast<S*>(batch_values), batch_counts);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

template <typename T, typename S, bool SCAN>
static ips_status launch_plain_tiles_pages(const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                           int op, const void* literals, int n_literals, int combine, int join,
                                           int op2, const void* literal2, uint64_t* bitmap, void* batch_values,
                                           uint32_t* batch_counts, hipStream_t s, uint32_t* edges) {
  PlainLit<T> lit;
  lit.edges = sizeof(S) == 4 ? edges : nullptr;
  lit.n = n_literals;
  lit.combine = combine;
  lit.join = join;
  lit.op2 = op2;
  lit.v2 = literal2 ? *reinterpret_cast<const T*>(literal2) : T();
  for (int i = 0; i < 16; ++i) lit.v[i] = i < n_literals ? reinterpret_cast<const T*>(literals)[i] : T();
  auto kern = plain_tile_pages_kernel<T, S, SCAN>;
  const int64_t max_batches = (max_rows + kRowsPerTile - 1) / kRowsPerTile;  // a wave per batch
  const int gx = paged_grid_x(reinterpret_cast<const void*>(kern), max_batches, n_pages);
  if (gx <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)n_pages), dim3(kThreads), 0, s, d_pages, chunk_rows, op, lit,
                     reinterpret_cast<uint32_t*>(bitmap), reinterpret_cast<S*>(batch_values), batch_counts);
  IPS_HIP_TRY(hipGetLastError());
  if (lit.edges)  // edge mode: the sub-tiles' end dwords are merged by a second launch
    return launch_window_fixup(d_pages, n_pages, max_rows, chunk_rows, reinterpret_cast<uint32_t*>(bitmap), lit.edges,
                               combine, s);
  return IPS_OK;
}

#define IPS_PLAIN_TYPES(CALL)                 \
  switch (type) {                             \
    case IPS_T_INT8: CALL(int8_t, uint32_t);   \
    case IPS_T_INT16: CALL(int16_t, uint32_t); \
    case IPS_T_INT32: CALL(int32_t, uint32_t); \
    case IPS_T_INT64: CALL(int64_t, uint64_t); \
    case IPS_T_FLOAT: CALL(float, uint32_t);   \
    case IPS_T_DOUBLE: CALL(double, uint64_t); \
  }

ips_status launch_plain_scan(int type, const void* page, int64_t n_rows, int op, const void* literals,
                             int n_literals, int join, int op2, const void* literal2,
                             uint64_t* bitmap, void* batch_values, uint32_t* batch_counts,
                             hipStream_t s) {
#define IPS_PS(T, S)                                                                                     \
  return launch_plain_tiles<T, S, true>(page, n_rows, op, literals, n_literals, 0, join, op2, literal2, \
                                        bitmap, batch_values, batch_counts, s)
  IPS_PLAIN_TYPES(IPS_PS)
#undef IPS_PS
  set_error("plain_scan: bad type %d", type);
  return IPS_ERR_INVALID_ARG;
}

ips_status launch_plain_pred(int type, const void* page, int64_t n_rows, int op,
                             const void* literals, int n_literals, uint64_t* bitmap,
                             hipStream_t s, int combine, int join, int op2, const void* literal2) {
#define IPS_PL(T, S)                                                                                            \
  return launch_plain_tiles<T, S, false>(page, n_rows, op, literals, n_literals, combine, join, op2, literal2, \
                                         bitmap, nullptr, nullptr, s)
  IPS_PLAIN_TYPES(IPS_PL)
#undef IPS_PL
  set_error("plain_pred: bad type %d", type);
  return IPS_ERR_INVALID_ARG;
}

ips_status launch_plain_pred_pages(int type, const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                   int op, const void* literals, int n_literals, uint64_t* bitmap, hipStream_t s,
                                   int combine, int join, int op2, const void* literal2, uint32_t* edges) {
#define IPS_PL(T, S)                                                                                               \
  return launch_plain_tiles_pages<T, S, false>(d_pages, n_pages, max_rows, chunk_rows, op, literals, n_literals,   \
                                               combine, join, op2, literal2, bitmap, nullptr, nullptr, s, edges)
  IPS_PLAIN_TYPES(IPS_PL)
#undef IPS_PL
  set_error("plain_pred_pages: bad type %d", type);
  return IPS_ERR_INVALID_ARG;
}

ips_status launch_plain_scan_pages(int type, const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                   int op, const void* literals, int n_literals, int join, int op2, const void* literal2,
                                   uint64_t* bitmap, void* batch_values, uint32_t* batch_counts, hipStream_t s,
                                   uint32_t* edges) {
#define IPS_PS(T, S)                                                                                             \
  return launch_plain_tiles_pages<T, S, true>(d_pages, n_pages, max_rows, chunk_rows, op, literals, n_literals, 0, \
                                              join, op2, literal2, bitmap, batch_values, batch_counts, s, edges)
  IPS_PLAIN_TYPES(IPS_PS)
#undef IPS_PS
  set_error("plain_scan_pages: bad type %d", type);
  return IPS_ERR_INVALID_ARG;
}

// =============================================================================================
// Late materialisation on a PLAIN page against a given selection: ReadValue(skip) ->
// ParquetPlainEncoder::Decode(buffer, size, &val, skip_rows) per selected row (parquet-common.h:
// 186-190, hdfs-parquet-scanner.cc:1006-1027).  One wave per 2048-row batch, same batch layout as
// ips_fle_select.  Per batch, wave-uniform: with fewer than 64 selected rows the lane walks the set
// bits of its bitmap dword, four slot loads in flight; above that nearly every 128-byte line of
// the batch holds a selected row (10 %: 81 % of the lines of an 8-byte column), so the batch is
// streamed through the tile image like the fused scan, with the bitmap's bits in place of the
// comparison.
// =============================================================================================
constexpr uint32_t kPlainSelectStreamMin = 64;  // selected rows per 2048-row batch (3 %)

template <typename S>
__global__ __launch_bounds__(kThreads, 4) void plain_select_kernel(
    const S* __restrict__ page, int64_t n_rows, const uint32_t* __restrict__ bitmap32,
    S* __restrict__ batch_values, uint32_t* __restrict__ batch_counts) {
  using G = PlainGeom<S>;
  __shared__ __attribute__((aligned(16))) uint8_t lds_all[kWavesPerBlock * kPlainWaveBytes];
  const int lane = lane_id();
  const int wave = wave_id();
  uint8_t* lds = lds_all + wave * kPlainWaveBytes;
  uint16_t* list = reinterpret_cast<uint16_t*>(lds + kPlainImageBytes);
  const int64_t n_batches = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t n_tiles = (n_rows + G::RT - 1) / G::RT;
  const int64_t bm_dwords = bitmap_dwords(n_rows);
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;

  // the lane's bitmap dword of a batch (rows beyond n_rows cleared), its rank and the batch total
  struct Sel { uint32_t m, excl, count; };
  auto batch_sel = [&](int64_t batch) -> Sel {
    const int64_t d = batch * 64 + lane;
    uint32_t m = (batch < n_batches && d < bm_dwords) ? bitmap32[d] : 0u;
    const int64_t valid = n_rows - (batch * kRowsPerTile + (int64_t)lane * 32);
    if (valid < 32) m = valid <= 0 ? 0u : (m & ((1u << valid) - 1u));
    const uint32_t mine = (uint32_t)__builtin_popcount(m);
    const uint32_t incl = wave_inclusive_scan(mine);
    return Sel{m, incl - mine, (uint32_t)__builtin_amdgcn_readlane((int)incl, 63)};
  };

  int64_t batch = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (batch >= n_batches) return;
  Sel cur = batch_sel(batch);
  u32x4 r[kPlainLoads];
  if (cur.count >= kPlainSelectStreamMin) plain_tile_load<S>(page, batch * G::TPB, n_rows, lane, r);
  while (batch < n_batches) {
    // one batch ahead: the next batch's bits decide whether its page bytes are prefetched
    const int64_t next = batch + stride;
    const Sel nxt = batch_sel(next);
    const bool next_streams = nxt.count >= kPlainSelectStreamMin;  // wave-uniform
    S* dst = batch_values + batch * kRowsPerTile;
    if (cur.count >= kPlainSelectStreamMin) {
      uint32_t base = 0;
#pragma unroll
      for (int t = 0; t < G::TPB; ++t) {
        const int64_t tile = batch * G::TPB + t;
        if (tile < n_tiles) {
          plain_tile_stage(lds, lane, r);
          if (t + 1 < G::TPB) {
            if (tile + 1 < n_tiles) plain_tile_load<S>(page, tile + 1, n_rows, lane, r);
          } else if (next_streams) {
            plain_tile_load<S>(page, next * G::TPB, n_rows, lane, r);
          }
          wave_lds_fence();
          // the bits of this lane's rows of the tile: 4-byte slots: its own dword; 8-byte slots:
          // half of dword 32 t + lane / 2 of the batch
          uint32_t mt, excl_t, count_t;
          if (G::R == 32) {
            mt = cur.m, excl_t = cur.excl, count_t = cur.count;
          } else {
            const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((32 * t + (lane >> 1)) << 2, (int)cur.m);
            mt = (w >> (16 * (lane & 1))) & 0xFFFFu;
            const uint32_t mine_t = (uint32_t)__builtin_popcount(mt);
            const uint32_t incl_t = wave_inclusive_scan(mine_t);
            excl_t = incl_t - mine_t;
            count_t = (uint32_t)__builtin_amdgcn_readlane((int)incl_t, 63);
          }
          plain_materialise<S>(lds, list, lane, mt, excl_t, count_t, dst + base);
          base += count_t;
          wave_lds_fence();
        }
      }
    } else {
      if (next_streams) plain_tile_load<S>(page, next * G::TPB, n_rows, lane, r);
      uint32_t m = cur.m;
      uint32_t P = cur.excl;
      const S* src = page + batch * kRowsPerTile + (int64_t)lane * 32;
      while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
        S x[4];
        bool ok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ok[e] = m != 0u;
          x[e] = ok[e] ? src[__builtin_ctz(m)] : (S)0;
          m &= m - 1u;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ok[e]) dst[P++] = x[e];
        }
      }
    }
    if (lane == 0) batch_counts[batch] = cur.count;
    cur = nxt;
    batch = next;
  }
}

ips_status launch_plain_select(int stride_bytes, const void* page, int64_t n_rows,
                               const uint64_t* bitmap, void* batch_values, uint32_t* batch_counts,
                               hipStream_t s) {
  const int64_t n_batches = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  if (n_batches <= 0) return IPS_OK;
  if (stride_bytes == 4) {
    auto kern = plain_select_kernel<uint32_t>;
    const int grid = grid_for_tiles(reinterpret_cast<const void*>(kern), n_batches);
    if (grid <= 0) return IPS_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, s, reinterpret_cast<const uint32_t*>(page), n_rows,
                       reinterpret_cast<const uint32_t*>(bitmap), reinterpret_cast<uint32_t*>(batch_values),
                       batch_counts);
  } else {
    auto kern = plain_select_kernel<uint64_t>;
    const int grid = grid_for_tiles(reinterpret_cast<const void*>(kern), n_batches);
    if (grid <= 0) return IPS_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, s, reinterpret_cast<const uint64_t*>(page), n_rows,
                       reinterpret_cast<const uint32_t*>(bitmap), reinterpret_cast<uint64_t*>(batch_values),
                       batch_counts);
  }
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

}  // namespace ips
