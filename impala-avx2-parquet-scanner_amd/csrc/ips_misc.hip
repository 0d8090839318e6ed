// ips_misc.hip -- bitmap algebra, batch concatenation, tuple assembly and the synthetic-column
// generator (PLAIN pages: ips_plain.hip; rank-based bitmap kernels: ips_rank.hip).  All HBM-bound
// streaming kernels.
#include <string.h>

#include "ips_host.h"

namespace ips {

// read-once column slots and the 16-byte tuple stream of assemble_tuples carry the nt hint
// (232 -> 207 us; IPS_AUX_NT, ips_knobs.h)
template <typename T>
__device__ __forceinline__ T aux_load(const T* p) {
  return IPS_AUX_NT ? __builtin_nontemporal_load(p) : *p;
}
template <typename T>
__device__ __forceinline__ void aux_store(T* p, T v) {
  if (IPS_AUX_NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// =============================================================================================
// Bitmap algebra: AndOperate / OrOperate (simple-predicates.h:145-163), resize(n, value),
// count().
// =============================================================================================
template <int OP>
__global__ void bitmap_binop_kernel(uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                    int64_t n_words) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
       i += (int64_t)gridDim.x * blockDim.x)
    a[i] = OP == 0 ? (a[i] & b[i]) : (a[i] | b[i]);
}

__global__ void bitmap_fill_kernel(uint64_t* __restrict__ a, int64_t n_rows, int value) {
  const int64_t n_words = (n_rows + 63) / 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t w = value ? ~0ull : 0ull;
    int64_t valid = n_rows - i * 64;
    if (valid < 64) w &= (1ull << valid) - 1ull;
    a[i] = w;
  }
}

__global__ __launch_bounds__(256) void bitmap_count_kernel(const uint64_t* __restrict__ a,
                                                           int64_t n_rows,
                                                           unsigned long long* __restrict__ count) {
  __shared__ unsigned long long part[4];
  const int64_t n_words = (n_rows + 63) / 64;
  const int64_t n_pairs = n_words / 2;  // 16-byte loads; the last (maybe partial) words go scalar
  unsigned long long c = 0;
  const ulonglong2* a2 = reinterpret_cast<const ulonglong2*>(a);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; 2 * (i + 3 * step) + 2 < n_words; i += 4 * step) {  // four loads in flight, all words full
    const ulonglong2 w0 = a2[i], w1 = a2[i + step], w2 = a2[i + 2 * step], w3 = a2[i + 3 * step];
    c += __builtin_popcountll(w0.x) + __builtin_popcountll(w0.y) + __builtin_popcountll(w1.x) + __builtin_popcountll(w1.y) +
         __builtin_popcountll(w2.x) + __builtin_popcountll(w2.y) + __builtin_popcountll(w3.x) + __builtin_popcountll(w3.y);
  }
  for (; i < n_pairs; i += step) {
    if (2 * i + 2 < n_words) {  // both words are full
      ulonglong2 w = a2[i];
      c += __builtin_popcountll(w.x) + __builtin_popcountll(w.y);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) {  // the <= 3 trailing words, with the row mask
    const int64_t first = n_pairs >= 1 ? 2 * (n_pairs - 1) : 0;
    const int64_t i = first + threadIdx.x;
    if (i < n_words && (i >= 2 * n_pairs || 2 * (i / 2) + 2 >= n_words)) {
      uint64_t w = a[i];
      int64_t valid = n_rows - i * 64;
      if (valid < 64) w &= (1ull << valid) - 1ull;
      c += __builtin_popcountll(w);
    }
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if (lane_id() == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    c = part[0] + part[1] + part[2] + part[3];
    if (c) atomicAdd(count, c);
  }
}

static int small_grid(int64_t items, int threads) {
  int64_t want = (items + threads - 1) / threads;
  int64_t cap = (int64_t)device_cus() * 8;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

ips_status launch_bitmap_binop(int op, uint64_t* a, const uint64_t* b, int64_t n_words,
                               hipStream_t s) {
  if (n_words <= 0) return IPS_OK;
  int grid = small_grid(n_words, 256);
  if (op == 0) hipLaunchKernelGGL(bitmap_binop_kernel<0>, dim3(grid), dim3(256), 0, s, a, b, n_words);
  else hipLaunchKernelGGL(bitmap_binop_kernel<1>, dim3(grid), dim3(256), 0, s, a, b, n_words);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status launch_bitmap_fill(uint64_t* a, int64_t n_rows, int value, hipStream_t s) {
  if (n_rows <= 0) return IPS_OK;
  hipLaunchKernelGGL(bitmap_fill_kernel, dim3(small_grid((n_rows + 63) / 64, 256)), dim3(256), 0,
                     s, a, n_rows, value);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status launch_bitmap_count(const uint64_t* a, int64_t n_rows, int64_t* count, hipStream_t s) {
  IPS_HIP_TRY(hipMemsetAsync(count, 0, 8, s));
  if (n_rows <= 0) return IPS_OK;
  int grid = small_grid((n_rows + 127) / 128, 256);
  if (grid > device_cus() * 4) grid = device_cus() * 4;  // one atomic per block
  hipLaunchKernelGGL(bitmap_count_kernel, dim3(grid), dim3(256), 0, s, a, n_rows,
                     reinterpret_cast<unsigned long long*>(count));
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// The selected rows of every 2048-row batch of a selection (what a fused scan writes next to its
// values): one wave per batch, lane l the dword l of the batch's 32 words.
__global__ __launch_bounds__(256) void bitmap_batch_counts_kernel(const uint32_t* __restrict__ bitmap32,
                                                                  int64_t n_rows, uint32_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int64_t n_batches = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t dwords = 2 * ((n_rows + 63) / 64);
  for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_batches; b += (int64_t)gridDim.x * 4) {
    const int64_t d = b * 64 + lane;
    uint32_t w = d < dwords ? bitmap32[d] : 0u;
    const int64_t valid = n_rows - d * 32;
    if (valid < 32) w = valid <= 0 ? 0u : (w & ((1u << valid) - 1u));
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan((uint32_t)__builtin_popcount(w)), 63);
    if (lane == 0) counts[b] = total;
  }
}

ips_status launch_bitmap_batch_counts(const uint64_t* bitmap, int64_t n_rows, uint32_t* counts, hipStream_t s) {
  if (n_rows <= 0) return IPS_OK;
  const int64_t n_batches = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  int grid = small_grid((n_batches + 3) / 4, 1);
  if (grid > device_cus() * 16) grid = device_cus() * 16;
  hipLaunchKernelGGL(bitmap_batch_counts_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(bitmap),
                     n_rows, counts);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// =============================================================================================
// Device-wide exclusive scan of small per-item counts (three launches), shared by
//  - IntersectBitset expand: items = bitmap words, count = popcount      (scanner.cc:326-331)
//  - batch concatenation:   items = batches, count = batch_counts[b]      (scanner.cc:1151-1181)
// A block handles kScanItems consecutive items; block totals are scanned by one block.
// =============================================================================================
constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 4;
constexpr int kScanItems = kScanThreads * kScanPerThread;

struct PopcItems {
  const uint64_t* words;
  int64_t n_rows;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const {
    uint64_t w = words[i];
    int64_t valid = n_rows - i * 64;
    if (valid < 64) w &= (1ull << valid) - 1ull;
    return __builtin_popcountll(w);
  }
};
struct ArrayItems {
  const uint32_t* counts;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const { return counts[i]; }
};

// exclusive prefix of this thread's value inside the block + block total (all threads)
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wave_sums[kScanThreads / kWave];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  uint32_t incl = v;
  for (int off = 1; off < kWave; off <<= 1) {
    uint32_t t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  __syncthreads();  // wave_sums may still be read from a previous call
  if (lane == kWave - 1) wave_sums[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int i = 0; i < kScanThreads / kWave; ++i) {
    if (i < wave) base += wave_sums[i];
    tot += wave_sums[i];
  }
  *total = tot;
  return base + incl - v;
}

template <typename Items>
__global__ __launch_bounds__(kScanThreads) void scan_block_totals_kernel(Items items, int64_t n,
                                                                         uint64_t* block_totals) {
  const int64_t base = (int64_t)blockIdx.x * kScanItems + (int64_t)threadIdx.x * kScanPerThread;
  uint32_t v = 0;
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e)
    if (base + e < n) v += items(base + e);
  uint32_t total;
  block_exclusive_scan(v, &total);
  if (threadIdx.x == 0) block_totals[blockIdx.x] = total;
}

// one block: block_totals[0..nb) -> exclusive prefixes in place; grand total -> *grand
__global__ __launch_bounds__(kScanThreads) void scan_of_totals_kernel(uint64_t* block_totals,
                                                                      int64_t nb,
                                                                      int64_t* grand) {
  __shared__ uint64_t part[kScanThreads];
  const int64_t per = (nb + kScanThreads - 1) / kScanThreads;
  const int64_t lo = (int64_t)threadIdx.x * per;
  const int64_t hi = lo + per < nb ? lo + per : nb;
  uint64_t sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += block_totals[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t run = 0;
    for (int i = 0; i < kScanThreads; ++i) {
      uint64_t t = part[i];
      part[i] = run;
      run += t;
    }
    if (grand) *grand = (int64_t)run;
  }
  __syncthreads();
  uint64_t run = part[threadIdx.x];
  for (int64_t i = lo; i < hi; ++i) {
    uint64_t t = block_totals[i];
    block_totals[i] = run;
    run += t;
  }
}

// (IntersectBitset -- expand -- and its inverse, compress, live in ips_rank.hip.)

// exclusive prefix popcount per bitmap word (rank support for the NULL flags of selected rows)
__global__ __launch_bounds__(kScanThreads) void word_prefix_kernel(
    const uint64_t* __restrict__ words, int64_t n_bits, const uint64_t* __restrict__ block_offsets,
    uint64_t* __restrict__ prefix) {
  const int64_t n_words = (n_bits + 63) / 64;
  const int64_t base = (int64_t)blockIdx.x * kScanItems + (int64_t)threadIdx.x * kScanPerThread;
  PopcItems items{words, n_bits};
  uint32_t pc[kScanPerThread];
  uint32_t v = 0;
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) {
    pc[e] = base + e < n_words ? items(base + e) : 0u;
    v += pc[e];
  }
  uint32_t total;
  uint64_t off = block_offsets[blockIdx.x] + block_exclusive_scan(v, &total);
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) {
    if (base + e < n_words) prefix[base + e] = off;
    off += pc[e];
  }
}

size_t scan_workspace_bytes(int64_t items) {
  return (size_t)((items + kScanItems - 1) / kScanItems + 1) * 8;
}

// ---- batch concatenation --------------------------------------------------------------------
// Exclusive offset of every batch among all selected rows (block scan of the counts on top of the
// scanned block totals); the per-batch kernels below then run one wave per batch.
__global__ __launch_bounds__(kScanThreads) void batch_offsets_kernel(
    const uint32_t* __restrict__ counts, int64_t n_batches,
    const uint64_t* __restrict__ block_offsets, uint64_t* __restrict__ batch_off) {
  const int64_t base = (int64_t)blockIdx.x * kScanItems + (int64_t)threadIdx.x * kScanPerThread;
  uint32_t c[kScanPerThread];
  uint32_t v = 0;
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) {
    c[e] = base + e < n_batches ? counts[base + e] : 0u;
    v += c[e];
  }
  uint32_t total;
  uint64_t off = block_offsets[blockIdx.x] + block_exclusive_scan(v, &total);
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) {
    if (base + e < n_batches) batch_off[base + e] = off;
    off += c[e];
  }
}

template <typename V>
__global__ __launch_bounds__(kThreads) void batches_compact_kernel(
    const V* __restrict__ batch_values, const uint32_t* __restrict__ counts, int64_t n_batches,
    const uint64_t* __restrict__ batch_off, V* __restrict__ dense) {
  const int lane = lane_id();
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  int64_t batch = (int64_t)blockIdx.x * kWavesPerBlock + wave_id();
  if (batch >= n_batches) return;
  // a batch is a few hundred bytes: the wave's time is the latency of count/offset -> loads ->
  // stores, so the next batch's count and offset are fetched one batch ahead and the first four
  // rounds (256 values) are loaded before the first is stored
  uint32_t cnt = counts[batch];
  uint64_t off = batch_off[batch];
  while (batch < n_batches) {
    const int64_t next = batch + stride;
    uint32_t cnt_n = 0;
    uint64_t off_n = 0;
    if (next < n_batches) {
      cnt_n = counts[next];
      off_n = batch_off[next];
    }
    const V* src = batch_values + batch * kRowsPerTile;
    V* dst = dense + off;
    {
      V x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = lane + k * kWave < cnt ? src[lane + k * kWave] : V(0);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (lane + k * kWave < cnt) dst[lane + k * kWave] = x[k];  // (nt here: 51 -> 88 us, the 4-byte stores need L2)
    }
    uint32_t i = lane + 4 * kWave;
    for (; i + 3 * kWave < cnt; i += 4 * kWave) {  // four independent loads in flight per lane
      V x0 = src[i], x1 = src[i + kWave], x2 = src[i + 2 * kWave], x3 = src[i + 3 * kWave];
      dst[i] = x0; dst[i + kWave] = x1; dst[i + 2 * kWave] = x2; dst[i + 3 * kWave] = x3;
    }
    for (; i < cnt; i += kWave) dst[i] = src[i];
    cnt = cnt_n;
    off = off_n;
    batch = next;
  }
}

size_t batches_workspace_bytes(int64_t n_batches) {
  return scan_workspace_bytes(n_batches) + (size_t)(n_batches > 0 ? n_batches : 0) * 8 + 64;
}

// totals + per-batch offsets into the workspace; *total = number of selected rows
static ips_status launch_batch_offsets(const uint32_t* counts, int64_t n_batches, int64_t* total,
                                       void* workspace, uint64_t** batch_off, hipStream_t s) {
  const int64_t nb = (n_batches + kScanItems - 1) / kScanItems;
  uint64_t* totals = reinterpret_cast<uint64_t*>(workspace);
  *batch_off = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(workspace) +
                                           ((scan_workspace_bytes(n_batches) + 63) & ~(size_t)63));
  hipLaunchKernelGGL((scan_block_totals_kernel<ArrayItems>), dim3((unsigned)nb),
                     dim3(kScanThreads), 0, s, ArrayItems{counts}, n_batches, totals);
  hipLaunchKernelGGL(scan_of_totals_kernel, dim3(1), dim3(kScanThreads), 0, s, totals, nb, total);
  hipLaunchKernelGGL(batch_offsets_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, s, counts,
                     n_batches, totals, *batch_off);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

static int per_batch_grid(int64_t n_batches) {
  int64_t want = (n_batches + kWavesPerBlock - 1) / kWavesPerBlock;
  int64_t cap = (int64_t)device_cus() * 10 * grid_mult();
  return (int)(want < cap ? want : cap);
}

ips_status launch_batches_compact(const void* batch_values, const uint32_t* counts,
                                  int64_t n_batches, int value_width, void* dense,
                                  int64_t* total, void* workspace, hipStream_t s) {
  if (n_batches <= 0) {
    IPS_HIP_TRY(hipMemsetAsync(total, 0, 8, s));
    return IPS_OK;
  }
  uint64_t* batch_off = nullptr;
  ips_status st = launch_batch_offsets(counts, n_batches, total, workspace, &batch_off, s);
  if (st != IPS_OK) return st;
  const int grid = per_batch_grid(n_batches);
  if (value_width == 4)
    hipLaunchKernelGGL((batches_compact_kernel<uint32_t>), dim3(grid), dim3(kThreads), 0, s,
                       reinterpret_cast<const uint32_t*>(batch_values), counts, n_batches,
                       batch_off, reinterpret_cast<uint32_t*>(dense));
  else
    hipLaunchKernelGGL((batches_compact_kernel<uint64_t>), dim3(grid), dim3(kThreads), 0, s,
                       reinterpret_cast<const uint64_t*>(batch_values), counts, n_batches,
                       batch_off, reinterpret_cast<uint64_t*>(dense));
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// ---- tuple assembly -------------------------------------------------------------------------
struct TupleCols {
  const void* values[IPS_TUPLE_MAX_COLS];       // batches (REQUIRED) or dense values (OPTIONAL)
  const uint64_t* flags[IPS_TUPLE_MAX_COLS];    // OPTIONAL: non-NULL flag per selected row
  const uint64_t* prefix[IPS_TUPLE_MAX_COLS];   // OPTIONAL: set flags before each flag word
  int32_t width[IPS_TUPLE_MAX_COLS];
  int32_t offset[IPS_TUPLE_MAX_COLS];
  int32_t null_byte[IPS_TUPLE_MAX_COLS];
  int32_t null_mask[IPS_TUPLE_MAX_COLS];
  int32_t dense[IPS_TUPLE_MAX_COLS];            // REQUIRED column given as ONE dense array: tuple i takes value i
  int32_t n_cols;
  int32_t tuple_size;
  const uint8_t* d_template;                    // wide tuples: device copy of the template tuple
  uint32_t tmpl[32];                            // tuples <= 128 bytes: the template itself
};

constexpr int kTupleImageMax = 128;  // tuples up to this size are staged through LDS

// Row-major tuples from per-column batches (AssembleRows, hdfs-parquet-scanner.cc:1151-1181).
// One wave per batch, 64 tuples per round: lane i builds tuple i of the round in the wave's LDS
// image (every column read is a coalesced 256/512-byte load of the batch), then the image -- 64
// consecutive tuples = one contiguous piece of the output -- leaves as dword stores of consecutive
// lanes.  Every tuple starts as a copy of the template tuple (InitTuple()).  IMAGE = false (tuples
// wider than 128 bytes or not a multiple of 4): each lane copies the template and writes its slots
// straight to HBM.
// FAST: every column REQUIRED, 4 bytes wide and 4-byte aligned in the tuple (the common layout):
// the column loop is unrolled with constant indices, all slot loads of a round are issued before
// the first one is consumed.  (Tuples of at most 16 bytes of this kind: assemble_small_kernel.)
// The image is padded by one dword per 64 (img()): lane i writes dword d of its tuple at logical
// dword i * ts / 4 + d, which for 16 / 32 / 64-byte tuples put 4 / 8 / 16 lanes on every bank
// (68 % of the LDS cycles of the 16-byte case were conflicts); with the pad, lanes whose logical
// dwords are 64 apart land on neighbouring banks, and the linear read-out stays conflict-free.
// Measured (2^28 rows @10 %): the pad costs the read-out its 16-byte LDS reads and an add per access;
// 64-byte tuples gain 25 % from it (637 -> 474 us), 24- and 32-byte tuples LOSE 11-19 %, so it is
// applied from 64 bytes up (kTuplePadMin).
constexpr int kTuplePadMin = 64;
__device__ __forceinline__ uint32_t img(uint32_t byte, bool pad) { return pad ? byte + ((byte >> 8) << 2) : byte; }

// FAST tuples of at most 16 bytes never touch LDS: the lane builds its tuple in registers and the
// 64 tuples of a round leave as one coalesced store (4 / 8 / 12 / 16 bytes per lane).  Its own
// kernel with a compact argument block: inside assemble_tuples_kernel the scalar registers that
// TupleCols occupies spilled.
struct SmallTupleCols {
  const uint32_t* values[IPS_TUPLE_MAX_COLS];
  int32_t dword[IPS_TUPLE_MAX_COLS];  // tuple_offset / 4
  uint32_t tmpl[4];
  int32_t n_cols;
  int32_t tuple_size;
};

template <int TS>
__global__ __launch_bounds__(kThreads) void assemble_small_kernel(
    SmallTupleCols tc, const uint32_t* __restrict__ counts, int64_t n_batches,
    const uint64_t* __restrict__ batch_off, uint8_t* __restrict__ tuples) {
  const int lane = lane_id();
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  int64_t batch = (int64_t)blockIdx.x * kWavesPerBlock + wave_id();
  if (batch >= n_batches) return;
  uint32_t cnt = counts[batch];
  uint64_t first = batch_off[batch];
  for (; batch < n_batches; batch += stride) {
    const int64_t next = batch + stride;  // its count and offset are fetched one batch ahead
    const uint32_t cnt_n = next < n_batches ? counts[next] : 0u;
    const uint64_t first_n = next < n_batches ? batch_off[next] : 0ull;
    const uint32_t cnt_c = cnt;
    const uint64_t first_c = first;
    cnt = cnt_n;
    first = first_n;
    for (uint32_t i0 = 0; i0 < cnt_c; i0 += kWave) {
      const uint32_t i = i0 + lane;
      if (i >= cnt_c) continue;
      const uint64_t src = (uint64_t)batch * kRowsPerTile + i;
      uint32_t w[4] = {tc.tmpl[0], tc.tmpl[1], tc.tmpl[2], tc.tmpl[3]};
      for (int col = 0; col < tc.n_cols; ++col) {  // wave-uniform trip count and slot positions
        const uint32_t x = aux_load(tc.values[col] + src);
        const int d = tc.dword[col];
        w[0] = d == 0 ? x : w[0];
        w[1] = d == 1 ? x : w[1];
        w[2] = d == 2 ? x : w[2];
        w[3] = d == 3 ? x : w[3];
      }
      uint8_t* o = tuples + (first_c + i) * (uint64_t)TS;
      if (TS == 16) {
        const u32x4 v = {w[0], w[1], w[2], w[3]};
        aux_store(reinterpret_cast<u32x4*>(o), v);
      } else if (TS == 8) {
        *reinterpret_cast<uint64_t*>(o) = ((uint64_t)w[1] << 32) | w[0];
      } else {
#pragma unroll
        for (int d = 0; d < TS / 4; ++d) reinterpret_cast<uint32_t*>(o)[d] = w[d];
      }
    }
  }
}

template <bool IMAGE, bool FAST, bool PAD>
__global__ __launch_bounds__(kThreads) void assemble_tuples_kernel(
    TupleCols tc, const uint32_t* __restrict__ counts, int64_t n_batches,
    const uint64_t* __restrict__ batch_off, uint8_t* __restrict__ tuples) {
  extern __shared__ __attribute__((aligned(16))) uint8_t image_all[];  // IMAGE: 4 waves x 64 tuples
  const int lane = lane_id();
  const int wave = wave_id();
  const int ts = tc.tuple_size;
  constexpr bool pad = PAD;  // tuples of kTuplePadMin bytes and more (chosen by the launcher)
  uint8_t* image = image_all + (IMAGE ? wave * (int)img((uint32_t)(kWave * ts), pad) : 0);
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t batch = (int64_t)blockIdx.x * kWavesPerBlock + wave; batch < n_batches;
       batch += stride) {
    const uint32_t cnt = counts[batch];
    const uint64_t first = batch_off[batch];
    for (uint32_t i0 = 0; i0 < cnt; i0 += kWave) {
      const uint32_t i = i0 + lane;
      const uint32_t in_round = cnt - i0 < (uint32_t)kWave ? cnt - i0 : (uint32_t)kWave;
      // the lane's tuple: image + img(lane * ts + byte) (IMAGE) or its place in the output
      const uint32_t t0 = (uint32_t)(lane * ts);
      uint8_t* t = IMAGE ? image : tuples + (first + i) * (uint64_t)ts;
      auto at = [&](int byte) -> uint8_t* { return IMAGE ? image + img(t0 + (uint32_t)byte, pad) : t + byte; };
      if (IMAGE) {
#pragma unroll
        for (int d = 0; d < kTupleImageMax / 4; ++d)
          if (4 * d < ts) *reinterpret_cast<uint32_t*>(at(4 * d)) = tc.tmpl[d];
      } else if (i < cnt) {
        for (int d = 0; d < ts; ++d) t[d] = tc.d_template[d];
      }
      if (FAST) {
        uint32_t x[IPS_TUPLE_MAX_COLS];
        const uint64_t src = (uint64_t)batch * kRowsPerTile + i;
#pragma unroll
        for (int col = 0; col < IPS_TUPLE_MAX_COLS; ++col)
          x[col] = (col < tc.n_cols && i < cnt) ? aux_load(reinterpret_cast<const uint32_t*>(tc.values[col]) + src) : 0u;
        if (i < cnt) {
#pragma unroll
          for (int col = 0; col < IPS_TUPLE_MAX_COLS; ++col)
            if (col < tc.n_cols) *reinterpret_cast<uint32_t*>(at(tc.offset[col])) = x[col];
        }
      } else if (i < cnt) {
        const uint64_t gi = first + i;  // index of this tuple among all selected rows
        for (int col = 0; col < tc.n_cols; ++col) {
          uint64_t src = (uint64_t)batch * kRowsPerTile + i;
          if (tc.dense[col]) {  // a dense REQUIRED column (materialised over page lists): value gi
            src = gi;
          } else if (tc.flags[col]) {  // OPTIONAL column: NULL bit, or the rank-th dense value
            const uint64_t fw = tc.flags[col][gi >> 6];
            if (!((fw >> (gi & 63)) & 1ull)) {
              *at(tc.null_byte[col]) |= (uint8_t)tc.null_mask[col];
              continue;
            }
            src = tc.prefix[col][gi >> 6] + __builtin_popcountll(fw & ((1ull << (gi & 63)) - 1ull));
          }
          const int so = tc.offset[col];
          const bool aligned = ((so | ts) & 3) == 0;  // slots are naturally aligned in Impala tuples
          if (tc.width[col] == 4) {
            const uint32_t x = reinterpret_cast<const uint32_t*>(tc.values[col])[src];
            if (aligned) {
              *reinterpret_cast<uint32_t*>(at(so)) = x;
            } else {
              for (int k = 0; k < 4; ++k) *at(so + k) = (uint8_t)(x >> (8 * k));
            }
          } else {
            const uint64_t x = reinterpret_cast<const uint64_t*>(tc.values[col])[src];
            if (aligned) {
              *reinterpret_cast<uint32_t*>(at(so)) = (uint32_t)x;
              *reinterpret_cast<uint32_t*>(at(so + 4)) = (uint32_t)(x >> 32);
            } else {
              for (int k = 0; k < 8; ++k) *at(so + k) = (uint8_t)(x >> (8 * k));
            }
          }
        }
      }
      if (IMAGE) {
        wave_lds_fence();
        uint8_t* dst = tuples + (first + i0) * (uint64_t)ts;
        const uint32_t bytes = in_round * (uint32_t)ts;
        if ((ts & 15) == 0) {  // 16-byte pieces: the destination is 16-byte aligned as well
          if (pad) {
            for (uint32_t o = lane * 16; o < bytes; o += kWave * 16) {
              const uint32_t* src4 = reinterpret_cast<const uint32_t*>(image + img(o, true));  // a piece never straddles a pad
              const u32x4 v = {src4[0], src4[1], src4[2], src4[3]};
              aux_store(reinterpret_cast<u32x4*>(dst + o), v);
            }
          } else {
            for (uint32_t o = lane * 16; o < bytes; o += kWave * 16)
              aux_store(reinterpret_cast<u32x4*>(dst + o), *reinterpret_cast<const u32x4*>(image + o));
          }
        } else {
          for (uint32_t o = lane * 4; o < bytes; o += kWave * 4)
            *reinterpret_cast<uint32_t*>(dst + o) = *reinterpret_cast<const uint32_t*>(image + img(o, pad));
        }
        wave_lds_fence();
      }
    }
  }
}

size_t scan_workspace_bytes(int64_t items);
size_t assemble_workspace_bytes(int64_t n_batches, int n_optional) {
  const int64_t flag_words = (n_batches * kRowsPerTile + 63) / 64;
  return batches_workspace_bytes(n_batches) +
         (size_t)n_optional * (scan_workspace_bytes(flag_words) + (size_t)flag_words * 8) + 64 +
         4096;  // + device copy of a wide template tuple
}

ips_status launch_assemble_tuples(const ips_tuple_column* cols, int n_cols, const uint32_t* counts,
                                  int64_t n_batches, int tuple_size, const void* h_template,
                                  void* tuples, int64_t* total, void* workspace, hipStream_t s) {
  if (n_batches <= 0) {
    IPS_HIP_TRY(hipMemsetAsync(total, 0, 8, s));
    return IPS_OK;
  }
  TupleCols tc;
  memset(&tc, 0, sizeof(tc));
  tc.n_cols = n_cols;
  tc.tuple_size = tuple_size;
  // workspace: [batch scan totals + batch offsets][per OPTIONAL column: flag-word scan totals +
  // per-word prefix]
  const int64_t n_rows_cap = n_batches * kRowsPerTile;
  const int64_t flag_words = (n_rows_cap + 63) / 64;
  uint8_t* ws = reinterpret_cast<uint8_t*>(workspace) + batches_workspace_bytes(n_batches);
  for (int i = 0; i < n_cols; ++i) {
    tc.width[i] = cols[i].value_width;
    tc.offset[i] = cols[i].tuple_offset;
    if (cols[i].d_nonnull_flags) {
      tc.values[i] = cols[i].d_dense_values;
      tc.flags[i] = cols[i].d_nonnull_flags;
      tc.null_byte[i] = cols[i].null_byte_offset;
      tc.null_mask[i] = cols[i].null_bit_mask;
      uint64_t* f_totals = reinterpret_cast<uint64_t*>(ws);
      uint64_t* f_prefix = reinterpret_cast<uint64_t*>(ws + scan_workspace_bytes(flag_words));
      ws += scan_workspace_bytes(flag_words) + (size_t)flag_words * 8;
      const int64_t fb = (flag_words + kScanItems - 1) / kScanItems;
      // the flags bitmap has one bit per selected row; bits beyond the last tuple are zero
      hipLaunchKernelGGL((scan_block_totals_kernel<PopcItems>), dim3((unsigned)fb),
                         dim3(kScanThreads), 0, s, PopcItems{tc.flags[i], n_rows_cap}, flag_words,
                         f_totals);
      hipLaunchKernelGGL(scan_of_totals_kernel, dim3(1), dim3(kScanThreads), 0, s, f_totals, fb,
                         (int64_t*)nullptr);
      hipLaunchKernelGGL(word_prefix_kernel, dim3((unsigned)fb), dim3(kScanThreads), 0, s,
                         tc.flags[i], n_rows_cap, f_totals, f_prefix);
      tc.prefix[i] = f_prefix;
    } else if (cols[i].d_dense_values) {
      tc.values[i] = cols[i].d_dense_values;
      tc.dense[i] = 1;
    } else {
      tc.values[i] = cols[i].d_batch_values;
    }
  }
  const bool image = tuple_size <= kTupleImageMax && tuple_size % 4 == 0;
  if (image) {
    if (h_template) memcpy(tc.tmpl, h_template, (size_t)tuple_size);
  } else {  // ws now points behind the last OPTIONAL column's scratch
    if (tuple_size > 4096) {
      set_error("ips_assemble_tuples: tuple_size %d > 4096", tuple_size);
      return IPS_ERR_UNSUPPORTED;
    }
    if (h_template) IPS_HIP_TRY(hipMemcpyAsync(ws, h_template, (size_t)tuple_size, hipMemcpyHostToDevice, s));
    else IPS_HIP_TRY(hipMemsetAsync(ws, 0, (size_t)tuple_size, s));
    tc.d_template = ws;
  }
  uint64_t* batch_off = nullptr;
  ips_status st = launch_batch_offsets(counts, n_batches, total, workspace, &batch_off, s);
  if (st != IPS_OK) return st;
  const int grid = per_batch_grid(n_batches);
  bool fast = image;
  for (int i = 0; i < n_cols; ++i)
    fast = fast && !cols[i].d_nonnull_flags && !cols[i].d_dense_values && cols[i].value_width == 4 && (cols[i].tuple_offset & 3) == 0;
  const size_t wave_image = (size_t)kWave * tuple_size;
  const size_t lds = image ? (size_t)kWavesPerBlock * (wave_image + (tuple_size >= kTuplePadMin ? ((wave_image >> 8) << 2) : 0)) : 0;
  uint8_t* out = reinterpret_cast<uint8_t*>(tuples);
  const bool pad = tuple_size >= kTuplePadMin;
  if (fast && tuple_size <= 16) {
    SmallTupleCols sc;
    memset(&sc, 0, sizeof(sc));
    for (int i = 0; i < n_cols; ++i) {
      sc.values[i] = reinterpret_cast<const uint32_t*>(tc.values[i]);
      sc.dword[i] = tc.offset[i] >> 2;
    }
    memcpy(sc.tmpl, tc.tmpl, (size_t)tuple_size);
    sc.n_cols = n_cols;
    sc.tuple_size = tuple_size;
#define IPS_SMALL(TS) hipLaunchKernelGGL((assemble_small_kernel<TS>), dim3(grid), dim3(kThreads), 0, s, sc, counts, n_batches, batch_off, out)
    if (tuple_size == 16) IPS_SMALL(16); else if (tuple_size == 12) IPS_SMALL(12); else if (tuple_size == 8) IPS_SMALL(8); else IPS_SMALL(4);
#undef IPS_SMALL
  } else if (fast && pad)
    hipLaunchKernelGGL((assemble_tuples_kernel<true, true, true>), dim3(grid), dim3(kThreads), lds, s, tc,
                       counts, n_batches, batch_off, out);
  else if (fast)
    hipLaunchKernelGGL((assemble_tuples_kernel<true, true, false>), dim3(grid), dim3(kThreads), lds, s, tc,
                       counts, n_batches, batch_off, out);
  else if (image && pad)
    hipLaunchKernelGGL((assemble_tuples_kernel<true, false, true>), dim3(grid), dim3(kThreads), lds, s,
                       tc, counts, n_batches, batch_off, out);
  else if (image)
    hipLaunchKernelGGL((assemble_tuples_kernel<true, false, false>), dim3(grid), dim3(kThreads), lds, s,
                       tc, counts, n_batches, batch_off, out);
  else
    hipLaunchKernelGGL((assemble_tuples_kernel<false, false, false>), dim3(grid), dim3(kThreads), 0, s, tc,
                       counts, n_batches, batch_off, out);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// =============================================================================================
// Synthetic column generator: x_i = splitmix64(seed + i), value = x_i & mask.
// =============================================================================================
__global__ void synth_kernel(uint64_t seed, int64_t n, uint32_t mask, uint32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = seed + (uint64_t)i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    out[i] = (uint32_t)z & mask;
  }
}

ips_status launch_synth(uint64_t seed, int64_t n, uint32_t mask, uint32_t* out, hipStream_t s) {
  if (n <= 0) return IPS_OK;
  hipLaunchKernelGGL(synth_kernel, dim3(small_grid(n, 256)), dim3(256), 0, s, seed, n, mask, out);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

}  // namespace ips
