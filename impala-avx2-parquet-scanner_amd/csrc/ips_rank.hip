// ips_rank.hip -- rank-based bitmap kernels: ColumnReader::IntersectBitset (expand) for OPTIONAL
// columns and the fused nullable predicate leaf built on it.
//
// The reference evaluates a predicate on a nullable dictionary page in three steps
// (hdfs-parquet-scanner.cc:338-345): fle_def_levels_->Eq(n, bits, max_def_level) -> NOT-NULL
// bitmap, the data predicate over bits.count() values, and IntersectBitset (:326-331), which walks
// the root bits one at a time and replaces every 1 by the next data bit.  Here:
//   1. the data predicate runs over the whole data buffer with the stand-alone predicate kernels
//      (its row count is never needed on the host: the buffer's size bounds it),
//   2. rank_tile_counts_kernel counts the non-NULL rows of every 262144-row tile straight from the
//      definition levels (width 1, max_def 1: the level words ARE the NOT-NULL bits, MSB first),
//   3. expand_kernel (workgroups of a quarter tile) gives every output word its rank (base = sum
//      of the tile counts before its tile, re-summed per workgroup from L2, plus the counts of the
//      tile's earlier waves; inside the workgroup a wave-level DPP scan), fetches the window of
//      data bits at that rank and deposits it into the NOT-NULL positions (four mask bits at a
//      time through a 256-entry table in LDS), optionally AND-ing / OR-ing into an existing
//      bitmap (ips_eval_program's combine modes).
// HBM traffic: def levels twice (n/8 bytes each), the data bitmap once in, the result once out.
#include "ips_chunk_host.h"
#include "ips_rank_device.h"

namespace ips {

template <int ROOT, bool ZERO>
__global__ __launch_bounds__(kRankThreads) void rank_tile_counts_kernel(
    const u64* __restrict__ root, int64_t n_rows, uint32_t* __restrict__ tile_counts,
    u64* __restrict__ zero_out) {
  rank_tile_counts_body<ROOT, ZERO>(root, n_rows, tile_counts, blockIdx.x, gridDim.x, zero_out);
}

__device__ __forceinline__ u64 deposit64(u64 src, u64 mask, const uint8_t* __restrict__ lut) {
  const uint32_t lo = deposit32((uint32_t)src, (uint32_t)mask, lut);
  const uint32_t hi = deposit32((uint32_t)(src >> __builtin_popcount((uint32_t)mask)), (uint32_t)(mask >> 32), lut);
  return (u64)lo | ((u64)hi << 32);
}

// the 256-entry tables, built at compile time and copied from the code object's constant data
// into LDS by each workgroup (one byte per thread)
struct NibbleLut { uint8_t v[256]; };
constexpr NibbleLut make_deposit_lut() {
  NibbleLut t{};
  for (uint32_t i = 0; i < 256; ++i) {
    t.v[i] = (uint8_t)deposit_lut_entry(i);
  }
  return t;
}
constexpr NibbleLut make_extract_lut() {
  NibbleLut t{};
  for (uint32_t i = 0; i < 256; ++i) {
    t.v[i] = (uint8_t)extract_lut_entry(i);
  }
  return t;
}
__device__ const NibbleLut kDepositLut = make_deposit_lut();
__device__ const NibbleLut kExtractLut = make_extract_lut();

__device__ __forceinline__ void deposit_lut_init(uint8_t* lut) { lut[threadIdx.x] = kDepositLut.v[threadIdx.x]; }

// out word i = deposit(sub bits [rank(i), rank(i) + popcount(root_i)), root_i), where rank(i) is the
// number of set root bits before word i.  Data bits at or beyond n_sub_bits read as 0 (a page
// whose NOT-NULL count exceeds its data rows selects nothing there).  COMBINE: 0 store, 1 and,
// 2 or into out.
//
// A wave whose words are all whole and whose data bits (plus the 96 bits a window load may touch)
// all exist takes the fast path: 32-bit offsets relative to the wave's base, the data window of a
// word = three dwords at dword granularity (issued for the wave's four words before the first
// deposit) funnel-shifted with v_alignbit, no per-word bounds arithmetic.  The last waves of a
// launch take the general path.  (SQ counters, 2^28 rows: 169 -> 118 VALU per word.)
struct __attribute__((packed, aligned(4))) Window3 { uint32_t d0, d1, d2; };

template <int COMBINE>
__device__ __forceinline__ void expand_store(u64* __restrict__ out, int64_t w0, int64_t n_words, u64 (&res)[2]) {
  if (w0 + 1 < n_words) {
    u32x4* dst = reinterpret_cast<u32x4*>(out + w0);
    if (COMBINE) {
      const u32x4 old = *dst;
      const u64 o0 = ((u64)old.y << 32) | old.x, o1 = ((u64)old.w << 32) | old.z;
      res[0] = COMBINE == 1 ? (res[0] & o0) : (res[0] | o0);
      res[1] = COMBINE == 1 ? (res[1] & o1) : (res[1] | o1);
    }
    const u32x4 t = {(uint32_t)res[0], (uint32_t)(res[0] >> 32), (uint32_t)res[1], (uint32_t)(res[1] >> 32)};
    IPS_STREAM_STORE16(dst, t);
  } else {
    if (COMBINE) res[0] = COMBINE == 1 ? (res[0] & out[w0]) : (res[0] | out[w0]);
    out[w0] = res[0];
  }
}

template <int ROOT, int COMBINE>
__global__ __launch_bounds__(kRankThreads) void expand_kernel(
    const u64* __restrict__ root, const u64* __restrict__ sub, int64_t n_rows, int64_t n_sub_bits,
    const uint32_t* __restrict__ tile_counts, int64_t tiles, u64* __restrict__ out) {
  __shared__ uint8_t lut[256];
  __shared__ u64 part[kRankWaves];
  __shared__ uint32_t wave_tot[kRankWaves];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t n_words = (n_rows + 63) / 64;
  const int64_t n_sub_words = (n_sub_bits + 63) / 64;
  deposit_lut_init(lut);
  const int64_t first = (int64_t)blockIdx.x * kExpWordsPerBlock + wave * kExpWordsPerWave;
  const bool whole = (first + kExpWordsPerWave) * 64 <= n_rows;  // wave-uniform
  u64 m[kExpRounds][2];
  if (whole) load_root_whole<ROOT, kExpRounds>(root, first, lane, m);
  else load_root<ROOT, kExpRounds>(root, first, n_words, n_rows, lane, m);
  // block base: the non-NULL rows of all tiles before this block's tile and of the parts of the
  // tile before this block (the counts of the tile's waves).  A thread's partial sum fits 32 bits
  // (<= 2^18 per tile, n_rows < 2^40: launch_expand checks); the wave total is formed from two
  // 16-bit halves so that no 32-bit scan can overflow.
  const int64_t tile = (int64_t)blockIdx.x / kExpBlocksPerTile;
  const int part_waves = (int)(blockIdx.x % kExpBlocksPerTile) * (kRankWaves / kExpBlocksPerTile);
  uint32_t before = 0;
  for (int64_t i = threadIdx.x; i < tile; i += kRankThreads) before += tile_counts[i];
  if ((int)threadIdx.x < part_waves) before += tile_counts[tiles + tile * kRankWaves + threadIdx.x];
  {
    const uint32_t lo = wave_sum(before & 0xFFFFu), hi = wave_sum(before >> 16);
    if (lane == 0) part[wave] = (u64)lo + ((u64)hi << 16);
  }

  // exclusive rank of every pair of words inside the wave: per round a DPP scan over the lanes
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

  const bool fast = whole && base + run + 128 <= (u64)n_sub_bits;
  if (fast) {  // wave-uniform
    const uint32_t* __restrict__ sp = reinterpret_cast<const uint32_t*>(sub + (base >> 6));
    const uint32_t lead = (uint32_t)(base & 63);
    Window3 win[kExpRounds][2];
    uint32_t sh[kExpRounds][2];
#pragma unroll
    for (int r = 0; r < kExpRounds; ++r) {
      uint32_t rel = lead + excl[r];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        win[r][e] = *reinterpret_cast<const Window3*>(sp + (rel >> 5));
        sh[r][e] = rel & 31u;
        rel += (uint32_t)__builtin_popcountll(m[r][e]);
      }
    }
#pragma unroll
    for (int r = 0; r < kExpRounds; ++r) {
      u64 res[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t mlo = (uint32_t)m[r][e], mhi = (uint32_t)(m[r][e] >> 32);
        const uint32_t lo = __builtin_amdgcn_alignbit(win[r][e].d1, win[r][e].d0, sh[r][e]);
        const uint32_t hi = __builtin_amdgcn_alignbit(win[r][e].d2, win[r][e].d1, sh[r][e]);
        const uint32_t pcl = (uint32_t)__builtin_popcount(mlo);
        const uint32_t src_hi = pcl == 32u ? hi : __builtin_amdgcn_alignbit(hi, lo, pcl);
        res[e] = (u64)deposit32(lo, mlo, lut) | ((u64)deposit32(src_hi, mhi, lut) << 32);
      }
      expand_store<COMBINE>(out, first + r * 128 + 2 * lane, n_words, res);
    }
    return;
  }

#pragma unroll
  for (int r = 0; r < kExpRounds; ++r) {
    const int64_t w0 = first + r * 128 + 2 * lane;
    if (w0 >= n_words) continue;
    u64 res[2];
    u64 off = base + excl[r];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const u64 mk = m[r][e];
      const uint32_t pc = (uint32_t)__builtin_popcountll(mk);
      u64 bits = 0;
      if (pc) {
        const int64_t wi = (int64_t)(off >> 6);
        const int sh = (int)(off & 63);
        if (wi < n_sub_words) {
          bits = sub[wi] >> sh;
          if (sh + (int)pc > 64 && wi + 1 < n_sub_words) bits |= sub[wi + 1] << (64 - sh);
          const int64_t avail = n_sub_bits - (int64_t)off;
          if (avail < 64) bits = avail <= 0 ? 0ull : (bits & ((1ull << avail) - 1ull));
        }
      }
      res[e] = deposit64(bits, mk, lut);
      off += pc;
    }
    expand_store<COMBINE>(out, w0, n_words, res);
  }
}

int64_t rank_tiles(int64_t n_rows) {
  const int64_t n_words = (n_rows + 63) / 64;
  return (n_words + kRankWordsPerTile - 1) / kRankWordsPerTile;
}

// workspace: tile counts + the counts of each tile's four waves
size_t rank_workspace_bytes(int64_t n_rows) {
  return ((size_t)rank_tiles(n_rows) * (1 + kRankWaves) * 4 + 255) & ~(size_t)255;
}

// root_kind: kRootBitmap / kRootLevels1.  tile_counts: rank_tiles(n_rows) uint32 (workspace).
ips_status launch_rank_tile_counts(int root_kind, const uint64_t* root, int64_t n_rows,
                                   uint32_t* tile_counts, hipStream_t s) {
  const int64_t tiles = rank_tiles(n_rows);
  if (tiles <= 0) return IPS_OK;
  const u64* r = reinterpret_cast<const u64*>(root);
  if (root_kind == kRootLevels1)
    hipLaunchKernelGGL((rank_tile_counts_kernel<kRootLevels1, false>), dim3((unsigned)tiles), dim3(kRankThreads), 0, s, r, n_rows, tile_counts, (u64*)nullptr);
  else
    hipLaunchKernelGGL((rank_tile_counts_kernel<kRootBitmap, false>), dim3((unsigned)tiles), dim3(kRankThreads), 0, s, r, n_rows, tile_counts, (u64*)nullptr);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// the same for every page of an OPTIONAL column chunk: blockIdx.y = page, its table at rank0
__global__ __launch_bounds__(kRankThreads) void rank_tile_counts_pages_kernel(const ChunkPage* __restrict__ pages,
                                                                             uint32_t* __restrict__ counts) {
  const ChunkPage pg = pages[blockIdx.y];
  const int64_t n_words = (pg.n_rows + 63) / 64;
  const int64_t tiles = (n_words + kRankWordsPerTile - 1) / kRankWordsPerTile;
  if ((int64_t)blockIdx.x >= tiles) return;  // (the grid is sized for the largest page)
  rank_tile_counts_body<kRootLevels1, false>(reinterpret_cast<const u64*>(pg.levels), pg.n_rows, counts + pg.rank0,
                                             blockIdx.x, tiles, nullptr);
}

ips_status launch_rank_counts_pages(const ChunkPage* d_pages, int n_pages, int64_t max_rows, uint32_t* counts,
                                    hipStream_t s) {
  const int64_t tiles = rank_tiles(max_rows);
  if (tiles <= 0 || n_pages <= 0) return IPS_OK;
  hipLaunchKernelGGL(rank_tile_counts_pages_kernel, dim3((unsigned)tiles, (unsigned)n_pages), dim3(kRankThreads), 0, s,
                     d_pages, counts);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status launch_expand(int root_kind, const uint64_t* root, const uint64_t* sub, int64_t n_rows,
                         int64_t n_sub_bits, const uint32_t* tile_counts, uint64_t* out,
                         int combine, hipStream_t s) {
  const int64_t tiles = rank_tiles(n_rows);
  if (tiles <= 0) return IPS_OK;
  IPS_REQUIRE(n_rows < (1ll << 40), "expand: %lld rows (the rank kernels take fewer than 2^40)", (long long)n_rows);
  const u64* r = reinterpret_cast<const u64*>(root);
  const u64* sb = reinterpret_cast<const u64*>(sub);
  u64* o = reinterpret_cast<u64*>(out);
  const int64_t n_words = (n_rows + 63) / 64;
  const dim3 grid((unsigned)((n_words + kExpWordsPerBlock - 1) / kExpWordsPerBlock));
#define IPS_EXPAND(R, C) hipLaunchKernelGGL((expand_kernel<R, C>), grid, dim3(kRankThreads), 0, s, r, sb, n_rows, n_sub_bits, tile_counts, tiles, o)
  if (root_kind == kRootLevels1) {
    if (combine == 0) IPS_EXPAND(kRootLevels1, 0); else if (combine == 1) IPS_EXPAND(kRootLevels1, 1); else IPS_EXPAND(kRootLevels1, 2);
  } else {
    if (combine == 0) IPS_EXPAND(kRootBitmap, 0); else if (combine == 1) IPS_EXPAND(kRootBitmap, 1); else IPS_EXPAND(kRootBitmap, 2);
  }
#undef IPS_EXPAND
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// IntersectBitset for callers that hold two bitmaps (ips_bitmap_expand): two launches.
ips_status launch_bitmap_expand(const uint64_t* root, const uint64_t* sub, int64_t n_rows,
                                uint64_t* out, void* workspace, hipStream_t s) {
  if (n_rows <= 0) return IPS_OK;
  uint32_t* counts = reinterpret_cast<uint32_t*>(workspace);
  ips_status st = launch_rank_tile_counts(kRootBitmap, root, n_rows, counts, s);
  if (st != IPS_OK) return st;
  return launch_expand(kRootBitmap, root, sub, n_rows, n_rows, counts, out, 0, s);
}

// ---- compress: the inverse of IntersectBitset ------------------------------------------------
// out bit j = src bit at the position of the j-th set bit of mask.  Same tiles and ranks as expand;
// per word the selected source bits are extracted (pext, four mask bits at a time through the
// table) and OR-ed into the wave's output segment, which is assembled in LDS (ds_or_b64: the
// pieces of neighbouring lanes share words) and leaves as whole words; only the first and the last
// word of a wave's segment can be shared with another wave and go out as global atomics (the
// output was cleared by the counting pass).  Round 1 issued two global atomics per input word.
__device__ __forceinline__ void extract_lut_init(uint8_t* lut) { lut[threadIdx.x] = kExtractLut.v[threadIdx.x]; }

// Like expand_kernel, a workgroup takes a quarter tile (four waves of 256 words): several
// generations of workgroups per CU, 2 KiB of LDS segment per wave instead of 8.
constexpr int kSegWords = kExpWordsPerWave + 2;  // a wave's output segment: <= 16384 bits + misalignment

template <int MASK, int SRC>
__global__ __launch_bounds__(kRankThreads) void compress_kernel(
    const u64* __restrict__ mask, const u64* __restrict__ src, int64_t n_rows,
    const uint32_t* __restrict__ tile_counts, int64_t tiles, u64* __restrict__ out,
    int64_t* __restrict__ n_out) {
  __shared__ uint8_t lut[256];
  __shared__ u64 part[kRankWaves];
  __shared__ uint32_t wave_tot[kRankWaves];
  __shared__ __attribute__((aligned(16))) u64 seg_all[kRankWaves * kSegWords];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t n_words = (n_rows + 63) / 64;
  extract_lut_init(lut);
  u64* seg = seg_all + wave * kSegWords;
  for (int i = lane; i < kSegWords; i += kWave) seg[i] = 0ull;
  const int64_t first = (int64_t)blockIdx.x * kExpWordsPerBlock + wave * kExpWordsPerWave;
  const bool whole = (first + kExpWordsPerWave) * 64 <= n_rows;  // wave-uniform
  u64 m[kExpRounds][2], sv[kExpRounds][2];
  if (whole) {
    load_root_whole<MASK, kExpRounds>(mask, first, lane, m);
    load_root_whole<SRC, kExpRounds>(src, first, lane, sv);
  } else {
    load_root<MASK, kExpRounds>(mask, first, n_words, n_rows, lane, m);
    load_root<SRC, kExpRounds>(src, first, n_words, n_rows, lane, sv);
  }
  // block base, as in expand_kernel
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
  if (n_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == kRankThreads - 1)
    *n_out = (int64_t)(base + run);  // last wave of the last block: popcount(mask)

  const uint32_t lead = (uint32_t)(base & 63);  // the segment starts inside global word base >> 6
#pragma unroll
  for (int r = 0; r < kExpRounds; ++r) {
    uint32_t o = lead + excl[r];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const u64 mk = m[r][e];
      const uint32_t pc = (uint32_t)__builtin_popcountll(mk);
      if (pc) {
        const u64 bits = extract64(sv[r][e], mk, lut);
        const uint32_t sh = o & 63u;
        if (bits) {
          __hip_atomic_fetch_or(&seg[o >> 6], bits << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (sh && (bits >> (64 - sh)))
            __hip_atomic_fetch_or(&seg[(o >> 6) + 1], bits >> (64 - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      }
      o += pc;
    }
  }
  wave_lds_fence();
  // flush: global words [g0, g0 + nw) hold this wave's bits; the two end words may be shared
  const int64_t g0 = (int64_t)(base >> 6);
  const uint32_t nw = run ? (lead + run + 63) / 64 : 0;
  for (uint32_t i = lane; i < nw; i += kWave) {
    const u64 v = seg[i];
    if (i == 0 || i == nw - 1) {
      if (v) atomicOr(reinterpret_cast<unsigned long long*>(out + g0 + i), (unsigned long long)v);
    } else {
      out[g0 + i] = v;
    }
  }
}

template <int MASK>
static void launch_compress_src(int src_kind, int64_t tiles, hipStream_t s, const u64* mask, const u64* src,
                                int64_t n_rows, const uint32_t* counts, u64* out, int64_t* n_out) {
  const int64_t n_words = (n_rows + 63) / 64;
  const dim3 grid((unsigned)((n_words + kExpWordsPerBlock - 1) / kExpWordsPerBlock));
  if (src_kind == kRootLevels1)
    hipLaunchKernelGGL((compress_kernel<MASK, kRootLevels1>), grid, dim3(kRankThreads), 0, s, mask, src, n_rows, counts, tiles, out, n_out);
  else
    hipLaunchKernelGGL((compress_kernel<MASK, kRootBitmap>), grid, dim3(kRankThreads), 0, s, mask, src, n_rows, counts, tiles, out, n_out);
}

// mask_kind / src_kind: kRootBitmap or kRootLevels1 (width-1 definition levels used as NOT-NULL bits)
ips_status launch_compress(int mask_kind, const uint64_t* mask, int src_kind, const uint64_t* src,
                           int64_t n_rows, uint64_t* out, int64_t* n_out, uint32_t* tile_counts,
                           hipStream_t s) {
  const int64_t tiles = rank_tiles(n_rows);
  if (tiles <= 0) {
    if (n_out) IPS_HIP_TRY(hipMemsetAsync(n_out, 0, 8, s));
    return IPS_OK;
  }
  const u64* mk = reinterpret_cast<const u64*>(mask);
  const u64* sr = reinterpret_cast<const u64*>(src);
  u64* o = reinterpret_cast<u64*>(out);
  const dim3 grid((unsigned)tiles);
  if (mask_kind == kRootLevels1) {
    hipLaunchKernelGGL((rank_tile_counts_kernel<kRootLevels1, true>), grid, dim3(kRankThreads), 0, s, mk, n_rows, tile_counts, o);
    launch_compress_src<kRootLevels1>(src_kind, tiles, s, mk, sr, n_rows, tile_counts, o, n_out);
  } else {
    hipLaunchKernelGGL((rank_tile_counts_kernel<kRootBitmap, true>), grid, dim3(kRankThreads), 0, s, mk, n_rows, tile_counts, o);
    launch_compress_src<kRootBitmap>(src_kind, tiles, s, mk, sr, n_rows, tile_counts, o, n_out);
  }
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status launch_bitmap_compress(const uint64_t* mask, const uint64_t* src, int64_t n_rows,
                                  uint64_t* out, int64_t* n_out, void* workspace, hipStream_t s) {
  return launch_compress(kRootBitmap, mask, kRootBitmap, src, n_rows, out, n_out,
                         reinterpret_cast<uint32_t*>(workspace), s);
}

// ---- counts for the one-pass late materialisation of an OPTIONAL column ----------------------
// Per rank tile and per wave quarter: NOT-NULL rows (R), selected rows (S) and selected NOT-NULL
// rows (RS) in one pass over the two bitmaps; also clears the flag words of the tile (the flags
// leave fle_select_nullable's compress step through atomics at the segment ends).
// pages != NULL (ips_chunk_select_nullable): blockIdx.y = page; the page's levels, its selection in the aligned
// copy (32 words per earlier batch), its tables at rank0; nothing is cleared (the caller cleared the flags).
template <int ROOT>
__global__ __launch_bounds__(kRankThreads) void rank3_counts_kernel(
    const u64* __restrict__ root, const u64* __restrict__ sel, int64_t n_rows,
    uint32_t* __restrict__ c_r, uint32_t* __restrict__ c_s, uint32_t* __restrict__ c_rs,
    u64* __restrict__ zero_out, const ChunkPage* __restrict__ pages) {
  __shared__ uint32_t wave_tot[3][kRankWaves];
  const int lane = lane_id();
  const int wave = wave_id();
  int64_t tile = blockIdx.x, tiles = gridDim.x;
  if (pages) {  // wave-uniform
    const ChunkPage pg = pages[blockIdx.y];
    n_rows = pg.n_rows;
    tiles = (((n_rows + 63) / 64) + kRankWordsPerTile - 1) / kRankWordsPerTile;
    if (tile >= tiles) return;  // (the grid is sized for the largest page)
    root = reinterpret_cast<const u64*>(pg.levels);
    sel += (size_t)pg.batch0 * (kRowsPerTile / 64);
    c_r += pg.rank0;
    c_s += pg.rank0;
    c_rs += pg.rank0;
  }
  const int64_t n_words = (n_rows + 63) / 64;
  const int64_t first = tile * kRankWordsPerTile + wave * kRankWordsPerWave;
  uint32_t cr = 0, cs = 0, crs = 0;
  if ((first + kRankWordsPerWave) * 64 <= n_rows) {  // whole words only: popcounts do not care about the bit order
#pragma unroll
    for (int r = 0; r < kRankRounds; ++r) {
      const int64_t w0 = first + r * 128 + 2 * lane;
      const u32x4 a = stream_load<true>(reinterpret_cast<const u32x4*>(root + w0));
      const u32x4 b = stream_load<true>(reinterpret_cast<const u32x4*>(sel + w0));
      u32x4 ar = a;
      if (ROOT == kRootLevels1)  // level words are MSB first, 64 bits at a time
        ar = u32x4{__builtin_bitreverse32(a.y), __builtin_bitreverse32(a.x), __builtin_bitreverse32(a.w), __builtin_bitreverse32(a.z)};
      cr += __builtin_popcount(a.x) + __builtin_popcount(a.y) + __builtin_popcount(a.z) + __builtin_popcount(a.w);
      cs += __builtin_popcount(b.x) + __builtin_popcount(b.y) + __builtin_popcount(b.z) + __builtin_popcount(b.w);
      crs += __builtin_popcount(ar.x & b.x) + __builtin_popcount(ar.y & b.y) + __builtin_popcount(ar.z & b.z) + __builtin_popcount(ar.w & b.w);
      const u32x4 z = {0u, 0u, 0u, 0u};
      if (zero_out) *reinterpret_cast<u32x4*>(zero_out + w0) = z;
    }
  } else {
    for (int r = 0; r < kRankRounds; ++r) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int64_t w0 = first + r * 128 + 2 * lane + e;
        if (w0 < n_words) {
          const u64 m = root_mask<ROOT>(root[w0], w0, n_rows);
          const u64 sv = root_mask<kRootBitmap>(sel[w0], w0, n_rows);
          cr += (uint32_t)__builtin_popcountll(m);
          cs += (uint32_t)__builtin_popcountll(sv);
          crs += (uint32_t)__builtin_popcountll(m & sv);
          if (zero_out) zero_out[w0] = 0ull;
        }
      }
    }
  }
  const uint32_t tr = wave_sum(cr), ts = wave_sum(cs), trs = wave_sum(crs);
  if (lane == 0) {
    wave_tot[0][wave] = tr;
    wave_tot[1][wave] = ts;
    wave_tot[2][wave] = trs;
    c_r[tiles + tile * kRankWaves + wave] = tr;
    c_s[tiles + tile * kRankWaves + wave] = ts;
    c_rs[tiles + tile * kRankWaves + wave] = trs;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    uint32_t* dst = threadIdx.x == 0 ? c_r : threadIdx.x == 1 ? c_s : c_rs;
    const uint32_t* w = wave_tot[threadIdx.x];
    dst[tile] = w[0] + w[1] + w[2] + w[3];
  }
}

ips_status launch_rank3_counts(int root_kind, const uint64_t* root, const uint64_t* sel, int64_t n_rows,
                               uint32_t* c_r, uint32_t* c_s, uint32_t* c_rs, uint64_t* zero_out,
                               hipStream_t s) {
  const int64_t tiles = rank_tiles(n_rows);
  if (tiles <= 0) return IPS_OK;
  const u64* r = reinterpret_cast<const u64*>(root);
  const u64* sv = reinterpret_cast<const u64*>(sel);
  u64* z = reinterpret_cast<u64*>(zero_out);
  if (root_kind == kRootLevels1)
    hipLaunchKernelGGL((rank3_counts_kernel<kRootLevels1>), dim3((unsigned)tiles), dim3(kRankThreads), 0, s, r, sv, n_rows, c_r, c_s, c_rs, z,
                       (const ChunkPage*)nullptr);
  else
    hipLaunchKernelGGL((rank3_counts_kernel<kRootBitmap>), dim3((unsigned)tiles), dim3(kRankThreads), 0, s, r, sv, n_rows, c_r, c_s, c_rs, z,
                       (const ChunkPage*)nullptr);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// ---- late materialisation of an OPTIONAL chunk held as a page list (ips_chunk_select_nullable) ----
// 1. every page's rows of the chunk-wide selection as a bitmap of its own (row 0 of the page at bit 0,
//    32 words per 2048-row batch slot of the chunk), so that the counting and the selecting kernel read
//    whole aligned words whatever row the page starts at
__global__ __launch_bounds__(256) void selection_pages_kernel(const ChunkPage* __restrict__ pages,
                                                             const u64* __restrict__ sel, int64_t total_words,
                                                             u64* __restrict__ out) {
  const ChunkPage pg = pages[blockIdx.y];
  const int64_t n_words = (pg.n_rows + 63) / 64;
  u64* dst = out + (size_t)pg.batch0 * (kRowsPerTile / 64);
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_words; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bit0 = pg.row0 + 64 * j;
    const int64_t w = bit0 >> 6;
    const uint32_t sh = (uint32_t)(bit0 & 63);
    u64 v = sel[w];
    if (sh) v = (v >> sh) | ((w + 1 < total_words ? sel[w + 1] : 0ull) << (64 - sh));
    const int64_t valid = pg.n_rows - 64 * j;
    if (valid < 64) v &= (1ull << valid) - 1ull;
    dst[j] = v;
  }
}

// 3. selected rows / selected NOT-NULL rows in front of every page (exclusive prefix sums over the pages'
//    tile counts) and both totals: counts[0] = selected rows, counts[1] = values.  One workgroup.
__global__ __launch_bounds__(256) void selnull_page_bases_kernel(const ChunkPage* __restrict__ pages, int n_pages,
                                                                const uint32_t* __restrict__ c_s,
                                                                const uint32_t* __restrict__ c_rs,
                                                                u64* __restrict__ page_s, u64* __restrict__ page_rs,
                                                                int64_t* __restrict__ counts) {
  __shared__ u64 sh_s[256], sh_rs[256];
  u64 carry_s = 0, carry_rs = 0;
  for (int p0 = 0; p0 < n_pages; p0 += 256) {
    const int p = p0 + (int)threadIdx.x;
    u64 ts = 0, trs = 0;
    if (p < n_pages) {
      const ChunkPage pg = pages[p];
      const int64_t tiles = (((pg.n_rows + 63) / 64) + kRankWordsPerTile - 1) / kRankWordsPerTile;
      for (int64_t t = 0; t < tiles; ++t) {
        ts += c_s[pg.rank0 + t];
        trs += c_rs[pg.rank0 + t];
      }
    }
    sh_s[threadIdx.x] = ts;
    sh_rs[threadIdx.x] = trs;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // inclusive scan
      const u64 a = (int)threadIdx.x >= d ? sh_s[threadIdx.x - d] : 0ull;
      const u64 b = (int)threadIdx.x >= d ? sh_rs[threadIdx.x - d] : 0ull;
      __syncthreads();
      sh_s[threadIdx.x] += a;
      sh_rs[threadIdx.x] += b;
      __syncthreads();
    }
    if (p < n_pages) {
      page_s[p] = carry_s + sh_s[threadIdx.x] - ts;
      page_rs[p] = carry_rs + sh_rs[threadIdx.x] - trs;
    }
    carry_s += sh_s[255];
    carry_rs += sh_rs[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts[0] = (int64_t)carry_s;
    counts[1] = (int64_t)carry_rs;
  }
}

// launches 1-3 of ips_chunk_select_nullable (the fourth, per run of equally wide pages: fle_select_nullable_kernel)
ips_status launch_selnull_pages_prepare(const ChunkPage* d_pages, int n_pages, int64_t max_rows, const uint64_t* d_sel,
                                        int64_t chunk_rows, uint64_t* sel_copy, uint32_t* c_r, uint32_t* c_s,
                                        uint32_t* c_rs, uint64_t* page_s, uint64_t* page_rs, int64_t* counts,
                                        hipStream_t s) {
  if (n_pages <= 0) return IPS_OK;
  const int64_t max_words = (max_rows + 63) / 64;
  unsigned gx = (unsigned)((max_words + 255) / 256);
  gx = gx > 64u ? 64u : gx;
  hipLaunchKernelGGL(selection_pages_kernel, dim3(gx, (unsigned)n_pages), dim3(256), 0, s, d_pages,
                     reinterpret_cast<const u64*>(d_sel), (chunk_rows + 63) / 64, reinterpret_cast<u64*>(sel_copy));
  IPS_HIP_TRY(hipGetLastError());
  const int64_t tiles = rank_tiles(max_rows);
  hipLaunchKernelGGL((rank3_counts_kernel<kRootLevels1>), dim3((unsigned)tiles, (unsigned)n_pages), dim3(kRankThreads), 0, s,
                     (const u64*)nullptr, reinterpret_cast<const u64*>(sel_copy), (int64_t)0, c_r, c_s, c_rs, (u64*)nullptr, d_pages);
  IPS_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(selnull_page_bases_kernel, dim3(1), dim3(256), 0, s, d_pages, n_pages, c_s, c_rs,
                     reinterpret_cast<u64*>(page_s), reinterpret_cast<u64*>(page_rs), counts);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// compress with the mask's tile counts already in 'tile_counts' and 'out' already cleared
ips_status launch_compress_counted(int mask_kind, const uint64_t* mask, int src_kind, const uint64_t* src,
                                   int64_t n_rows, uint64_t* out, int64_t* n_out,
                                   const uint32_t* tile_counts, hipStream_t s) {
  const int64_t tiles = rank_tiles(n_rows);
  if (tiles <= 0) {
    if (n_out) IPS_HIP_TRY(hipMemsetAsync(n_out, 0, 8, s));
    return IPS_OK;
  }
  const u64* mk = reinterpret_cast<const u64*>(mask);
  const u64* sr = reinterpret_cast<const u64*>(src);
  u64* o = reinterpret_cast<u64*>(out);
  if (mask_kind == kRootLevels1) launch_compress_src<kRootLevels1>(src_kind, tiles, s, mk, sr, n_rows, tile_counts, o, n_out);
  else launch_compress_src<kRootBitmap>(src_kind, tiles, s, mk, sr, n_rows, tile_counts, o, n_out);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// ---- nullable predicate leaf: workspace + NOT-NULL root -------------------------------------
static size_t bitmap_slot_bytes(int64_t n_rows) {
  return ((size_t)((n_rows + 63) / 64) * 8 + 255) & ~(size_t)255;
}
size_t nullable_workspace_bytes(int64_t n_rows) {
  return rank_workspace_bytes(n_rows) + 2 * bitmap_slot_bytes(n_rows);
}
NullableWs nullable_workspace(void* d_workspace, int64_t n_rows) {
  uint8_t* p = reinterpret_cast<uint8_t*>(d_workspace);
  NullableWs ws;
  ws.tile_counts = reinterpret_cast<uint32_t*>(p);
  ws.sub = reinterpret_cast<uint64_t*>(p + rank_workspace_bytes(n_rows));
  ws.nonnull = reinterpret_cast<uint64_t*>(p + rank_workspace_bytes(n_rows) + bitmap_slot_bytes(n_rows));
  return ws;
}

ips_status nullable_prepare_root(const void* d_def_levels, int def_bit_width, int max_def_level,
                                 int64_t n_rows, const NullableWs& ws, int* root_kind,
                                 const uint64_t** root, hipStream_t s, bool count_tiles) {
  if (def_bit_width == 1 && max_def_level == 1) {
    // the usual flat OPTIONAL column: the level words are the NOT-NULL bits
    *root_kind = kRootLevels1;
    *root = reinterpret_cast<const uint64_t*>(d_def_levels);
  } else {
    // fle_def_levels_->Eq(n, bits, max_def_level), hdfs-parquet-scanner.cc:342
    PredArgs args;
    __builtin_memset(&args, 0, sizeof(args));
    args.op = IPS_OP_EQ;
    args.n_consts = 1;
    args.consts[0] = (uint32_t)max_def_level;
    ips_status st = launch_fle_pred(def_bit_width, reinterpret_cast<const uint64_t*>(d_def_levels),
                                    n_rows, args, reinterpret_cast<uint32_t*>(ws.nonnull), s);
    if (st != IPS_OK) return st;
    *root_kind = kRootBitmap;
    *root = ws.nonnull;
  }
  if (!count_tiles) return IPS_OK;
  return launch_rank_tile_counts(*root_kind, *root, n_rows, ws.tile_counts, s);
}

// the first rank_tiles(n_rows) workgroups of the predicate launch that takes 'args' count the tiles
void attach_rank_counts(PredArgs* args, int root_kind, const uint64_t* root, int64_t n_rows,
                        uint32_t* tile_counts) {
  args->aux_blocks = (int32_t)rank_tiles(n_rows);
  args->aux_kind = root_kind;
  args->aux_root = root;
  args->aux_rows = n_rows;
  args->aux_counts = tile_counts;
}

}  // namespace ips
